// pipe.hip — batches in flight: LANES x DEPTH contexts on one device, every batch queued on the device the moment it is submitted.
//
// The reference parallelises over pages with a rayon pool: every worker is fed without a central gate (src/ncc.rs:839-847).  On the
// GPU the unit is a batch of pages, and what has to overlap is one batch's latency-bound small kernels (statistics, verify, sorts,
// ordering, process_hits) with another batch's MFMA scan: that needs the batches on different streams — the LANES (three by default).
//
// Rounds 2-4 gave a lane one context and one worker thread, and a lane held exactly ONE batch: the host's round trip (wait, read,
// release, submit, wake the worker) stood between every two batches of a lane, and the scans were put in ticket order by a host-side
// gate — a thread woken a millisecond late delayed every later ticket, and a 3 ms pause of the submitting thread drained all lanes
// (BENCH_r04: 26.5 Gpx/s where the same build gave 32 on a quieter host).  Round 5:
//   * a lane is one STREAM and a ring of DEPTH contexts (two): batch k runs in context A, batch k + LANES in context B, k + 2 LANES in
//     A again — each with its own pages, scratch, result buffers and size estimates, so batch k + LANES is queued behind batch k on the
//     lane's stream while k still runs, and k's results stay readable until the host releases them;
//   * ONE thread (enqueue_main) queues every batch of every lane, in ticket order, as soon as it is submitted: kernels, the
//     cross-stream hand-over of the persistent scan kernel (an event chain, launch_scan_mfma) and an event behind the batch's last
//     kernel.  The scans therefore run in ticket order by construction — no gate, no condition variable between two scan launches —
//     and the device holds up to LANES x DEPTH batches of queued work (9 ms at BASELINE configs[1]) whatever the host threads do;
//   * nobody waits for the device on the submission path.  A batch is completed by whoever asks for it (focr_pipe_wait /
//     _host_results / _release): that thread waits for the batch's event, reads its sizes (finish_results; a batch whose size
//     estimates proved too small is redone there, exactly), and copies results out on the lane's side stream.
// DESIGN.md section 5 ("The executor") has the measurements.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "common.h"

namespace focr {

struct PipeJob {
    const void *pages = nullptr;  // nullptr: rescan the pages already resident in the slot's context
    int on_device = 0, invert = 0, mode = 0;
    size_t n_pages = 0, r_w = 0, r_h = 0;
    float threshold = 0.f, anchor_threshold = 0.f;
    uint32_t cap = 0;
    int32_t overlap = 0;
    int post = 1;
    bool last = false;          // nothing follows this batch for now: its tail may take the whole chip
    void *chars_out = nullptr;  // device buffer that receives a copy of the batch's characters (focr_hit_t[])
    size_t chars_cap = 0;       // its size in bytes
};

struct PinBuf {  // grow-only page-locked host buffer
    void *p = nullptr;
    size_t bytes = 0;
    void *ensure(size_t want) {
        if (want <= bytes && p) return p;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        const size_t grow = want + want / 4 + 4096;
        if (hipHostMalloc(&p, grow, hipHostMallocDefault) != hipSuccess) return p = nullptr;
        bytes = grow;
        return p;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// One context of a lane's ring.  All fields are guarded by focr_pipe::mu except where a state hands them to one thread:
// TAKEN = the enqueue thread is queueing the batch; finishing = one consumer thread is completing it.
struct PipeSlot {
    enum State { IDLE, QUEUED, TAKEN, ENQUEUED, DONE };
    focr_ctx *ctx = nullptr;
    unsigned lane = 0;
    State state = IDLE;
    bool finishing = false;
    uint64_t ticket = 0;
    PipeJob job;
    int rc = FOCR_OK;
    hipEvent_t ev_done = nullptr;  // behind the batch's last kernel on the lane's stream: focr_pipe::done_ring[ticket % DONE_RING] (timing on)
    // focr_pipe_set_fetch: the batch's results in page-locked host memory
    PinBuf h_counts, h_page_off, h_line_off, h_chars;
    focr_host_results_t res{};
    bool fetched = false;
    bool chars_queued = false;  // the copy into job.chars_out was queued behind process_hits on the lane's stream (no copy at completion)
    // focr_pipe_prefetch: the slot's NEXT batch crosses PCIe into the lane's staging buffer and is ingested into the context's
    // ALTERNATE page set (pages_alt_ingest, ctx.hip), both on the lane's copy stream, under the batches in flight; when the
    // announced batch is queued, the context's two page sets change places
    hipEvent_t ev_prefetch = nullptr;  // copy + ingest done
    const void *pf_ptr = nullptr;      // host pages announced and on their way (consumed when the batch that brings the same pointer is queued)
    bool pf_issuing = false;           // a thread is queueing a copy + ingest into this slot's alternate page set
    size_t pf_n = 0, pf_w = 0, pf_h = 0;
    int pf_invert = 0;
    focr_ticket_times_t times{};
};

struct PipeLane {
    hipStream_t stream = nullptr;       // owned by the lane's first context
    hipStream_t copy_stream = nullptr;  // announced batches: host -> staging buffer -> alternate page set
    hipStream_t io_stream = nullptr;    // results of finished batches: device -> host, device -> chars_out
    void *pf_stage = nullptr;
    size_t pf_stage_bytes = 0;
    std::mutex stage_mu;                // the staging buffer's owner while a copy + ingest is being queued
};

}  // namespace focr

struct focr_pipe {
    int device = 0;
    unsigned n_lanes = 0, depth = 0;
    std::vector<focr::PipeLane *> lanes;
    std::vector<focr::PipeSlot *> slots;  // slot i: lane i % n_lanes, ring position i / n_lanes; ticket t -> slot (t - 1) % slots.size()
    std::mutex mu;
    std::condition_variable cv;
    uint64_t next_ticket = 1;   // the next focr_pipe_submit gets this one
    uint64_t next_enqueue = 1;  // the enqueue thread's next ticket
    uint64_t announced = 0;     // focr_pipe_prefetch calls not yet consumed by their focr_pipe_submit
    bool next_is_last = false;  // focr_pipe_announce_last: the next submit ends the stream
    bool stop = false, fetch = false, trace = false;
    std::thread enq;
    // completion events by ticket (ring: a ticket's event is recorded again DONE_RING tickets later, long after it has been retired):
    // the device-side interval between the last kernels of consecutive tickets comes from two neighbours of the ring
    static constexpr unsigned DONE_RING = 128;
    hipEvent_t done_ring[DONE_RING] = {};
    uint64_t done_ok[DONE_RING] = {};  // the ticket each event was recorded for without an error (guarded by mu)
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::vector<focr_ticket_times_t> traces;  // FOCR_PIPE_TRACE=1: every ticket's stamps, printed when the pipe is destroyed
    std::vector<uint64_t> trace_tickets;
    double now_us() const { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
    focr::PipeSlot *slot_of(uint64_t t) const { return slots[(t - 1) % slots.size()]; }
};

namespace focr {

// Queue a copy + ingest of host pages into the slot's alternate page set on its lane's copy stream (no lock held: HIP calls only;
// the caller has set S->pf_issuing).  Errors go to the process-wide message: the context's own belongs to whoever runs its batch.
static int issue_prefetch(focr_pipe *P, PipeSlot *S, const void *pages, size_t n_pages, size_t r_w, size_t r_h, int invert) {
    PipeLane *L = P->lanes[S->lane];
    focr_ctx *c = S->ctx;
    const size_t bytes = n_pages * r_w * r_h;
    FOCR_HIP((focr_ctx *)nullptr, hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(L->stage_mu);  // one staging buffer per lane; its users are in order on the copy stream
    if (L->pf_stage_bytes < bytes) {
        FOCR_HIP((focr_ctx *)nullptr, hipStreamSynchronize(L->copy_stream));  // the previous announcement's ingest has read the old buffer
        if (L->pf_stage) (void)hipFree(L->pf_stage);
        L->pf_stage = nullptr;
        L->pf_stage_bytes = 0;
        if (hipMalloc(&L->pf_stage, bytes) != hipSuccess) return fail(nullptr, FOCR_ERR_NOMEM, "focr_pipe: staging buffer: hipMalloc failed");
        L->pf_stage_bytes = bytes;
    }
    // copy, then ingest, in stream order (the lane's previous ingest, which read the staging buffer, is ahead of this copy)
    FOCR_HIP((focr_ctx *)nullptr, hipMemcpyAsync(L->pf_stage, pages, bytes, hipMemcpyHostToDevice, L->copy_stream));
    if (int rc = pages_alt_ingest(c, L->pf_stage, n_pages, r_w, r_h, invert, L->copy_stream)) return rc;
    FOCR_HIP((focr_ctx *)nullptr, hipEventRecord(S->ev_prefetch, L->copy_stream));
    return FOCR_OK;
}

// The one thread that queues batches on the device, in ticket order.
static void enqueue_main(focr_pipe *P) {
    (void)hipSetDevice(P->device);
    for (;;) {
        PipeSlot *S = nullptr;
        PipeJob job;
        uint64_t t = 0;
        bool prefetched = false, alt_free = false;
        {
            std::unique_lock<std::mutex> lk(P->mu);
            P->cv.wait(lk, [&] {
                if (P->stop) return true;
                PipeSlot *s = P->slot_of(P->next_enqueue);
                return s->state == PipeSlot::QUEUED && s->ticket == P->next_enqueue;
            });
            if (P->stop) return;
            t = P->next_enqueue;
            S = P->slot_of(t);
            S->state = PipeSlot::TAKEN;
            job = S->job;
            if (job.pages && !job.on_device) {
                const bool mine = S->pf_ptr == job.pages && !S->pf_issuing;
                prefetched = mine && S->pf_n == job.n_pages && S->pf_w == job.r_w && S->pf_h == job.r_h && S->pf_invert == job.invert;
                if (mine && !prefetched) S->pf_ptr = nullptr;  // announced with another geometry or inversion: the announcement is void
                // host pages that were not announced take the same road now (copy + ingest on the copy stream, into the alternate
                // page set) — unless a LATER batch of this slot has been announced into that set already
                alt_free = !prefetched && S->pf_ptr == nullptr && !S->pf_issuing;
                if (alt_free) S->pf_issuing = true;
            }
        }
        focr_ctx *c = S->ctx;
        S->times.enqueue_begin_us = P->now_us();
        int rc = FOCR_OK;
        c->tail_full_chip = job.last;
        if (job.pages && !job.on_device) {
            if (alt_free) {
                rc = issue_prefetch(P, S, job.pages, job.n_pages, job.r_w, job.r_h, job.invert);
                if (rc != FOCR_OK) c->err = focr_last_error_global();
            }
            if (rc == FOCR_OK && (prefetched || alt_free)) {  // the pages are in the alternate set (or on their way): behind one event, the sets change places
                hipError_t e = hipStreamWaitEvent(c->stream, S->ev_prefetch, 0);
                if (e != hipSuccess) rc = fail(c, FOCR_ERR_NO_DEVICE, std::string("focr_pipe: prefetch wait failed: ") + hipGetErrorString(e));
                if (rc == FOCR_OK) rc = pages_alt_swap(c, job.n_pages, job.r_w, job.r_h);
            } else if (rc == FOCR_OK) {  // the alternate set belongs to a later batch: upload on the lane's own stream
                rc = focr_pages_alloc(c, job.n_pages, job.r_w, job.r_h);
                if (rc == FOCR_OK) rc = focr_pages_upload(c, 0, job.n_pages, (const uint8_t *)job.pages, job.invert);
            }
            {
                std::lock_guard<std::mutex> lk(P->mu);
                if (prefetched) S->pf_ptr = nullptr;  // consumed: the slot may take its next announcement
                if (alt_free) S->pf_issuing = false;
            }
            P->cv.notify_all();
        } else if (job.pages) {
            rc = focr_pages_alloc(c, job.n_pages, job.r_w, job.r_h);
            if (rc == FOCR_OK) rc = focr_pages_upload_device(c, 0, job.n_pages, job.pages, job.invert);
        }
        if (rc == FOCR_OK) rc = focr_scan(c, job.threshold, job.cap, job.mode);
        S->times.scan_queued_us = P->now_us();
        if (rc == FOCR_OK && job.post) rc = focr_process_hits(c, job.anchor_threshold, job.overlap);
        // the characters' copy into the caller's device buffer rides on the lane's stream too (a kernel that reads their number on the device)
        S->chars_queued = rc == FOCR_OK && job.post && job.chars_out && post_queue_chars_copy(c, job.chars_out, job.chars_cap);
        if (rc == FOCR_OK) {
            S->ev_done = P->done_ring[t % focr_pipe::DONE_RING];
            hipError_t e = hipEventRecord(S->ev_done, c->stream);
            if (e != hipSuccess) rc = fail(c, FOCR_ERR_NO_DEVICE, std::string("focr_pipe: event record failed: ") + hipGetErrorString(e));
            else c->batch_event = S->ev_done;
        }
        S->times.enqueue_end_us = P->now_us();
        {
            std::lock_guard<std::mutex> lk(P->mu);
            S->rc = rc;
            P->done_ok[t % focr_pipe::DONE_RING] = rc == FOCR_OK ? t : 0;
            S->state = PipeSlot::ENQUEUED;
            P->next_enqueue = t + 1;
        }
        P->cv.notify_all();
    }
}

// Complete the batch of slot S (ticket t): called with P->mu held through `lk`; returns the batch's status with the lock held again.
// Exactly one thread does the work (finishing); the others wait for DONE.
static int complete(focr_pipe *P, PipeSlot *S, uint64_t t, std::unique_lock<std::mutex> &lk) {
    P->cv.wait(lk, [&] { return S->ticket != t || S->state == PipeSlot::DONE || S->state == PipeSlot::IDLE || (S->state == PipeSlot::ENQUEUED && !S->finishing); });
    if (S->ticket != t || S->state == PipeSlot::IDLE) return fail(S->ctx, FOCR_ERR_STATE, "focr_pipe: ticket is not outstanding");
    if (S->state == PipeSlot::DONE) return S->rc;
    S->finishing = true;
    int rc = S->rc;
    const PipeJob job = S->job;
    // the previous ticket's completion event, for the device-side interval between two batches' last kernels
    const bool prev_ok = t > 1 && P->done_ok[(t - 1) % focr_pipe::DONE_RING] == t - 1;
    hipEvent_t prev_ev = P->done_ring[(t - 1) % focr_pipe::DONE_RING];
    lk.unlock();
    focr_ctx *c = S->ctx;
    PipeLane *L = P->lanes[S->lane];
    const uint64_t redone_before = c->counters_redone;
    if (rc == FOCR_OK) rc = focr_sync(c);  // the batch's event, then its sizes (a batch whose estimates were too small is redone here)
    if (c->counters_redone != redone_before) S->chars_queued = false;  // ... and what was copied out behind its first attempt is void
    S->times.done_us = P->now_us();
    S->times.device_gap_ms = -1.f;
    if (rc == FOCR_OK && prev_ok) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, prev_ev, S->ev_done) == hipSuccess) S->times.device_gap_ms = ms;
        else (void)hipGetLastError();  // the previous batch is still running (batches retired out of order): no interval
    }
    if (rc == FOCR_OK && job.post && job.chars_out) {
        const size_t bytes = focr_total_chars(c) * sizeof(focr_hit_t);
        if (bytes > job.chars_cap) {
            rc = fail(c, FOCR_ERR_OVERFLOW, "focr_pipe_submit: chars_out is too small for this batch");
        } else if (bytes && !S->chars_queued) {  // (exact-size batches, a batch redone: the results were final before a copy could be queued) on the lane's side stream
            hipError_t e = hipMemcpyAsync(job.chars_out, focr_lines_device_chars(c), bytes, hipMemcpyDefault, L->io_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(L->io_stream);
            if (e != hipSuccess) rc = fail(c, FOCR_ERR_NO_DEVICE, std::string("focr_pipe: copy-out failed: ") + hipGetErrorString(e));
        }
    }
    if (rc == FOCR_OK && P->fetch) {  // results to page-locked host memory
        focr_host_results_t &R = S->res;
        R = focr_host_results_t{};
        R.n_pages = c->n_pages;
        R.n_templates = c->n_templates;
        R.n_matches = focr_total_matches(c);
        R.n_lines = job.post ? focr_total_lines(c) : 0;
        R.n_chars = job.post ? focr_total_chars(c) : 0;
        float ms[6] = {0};
        focr_last_timings(c, ms);
        R.device_ms = ms[5] + ms[4];
        uint32_t *hc = (uint32_t *)S->h_counts.ensure(R.n_pages * R.n_templates * 4 + 16);
        uint64_t *hp = (uint64_t *)S->h_page_off.ensure((R.n_pages + 1) * 8);
        uint64_t *hl = (uint64_t *)S->h_line_off.ensure((R.n_lines + 1) * 8);
        focr_hit_t *hh = (focr_hit_t *)S->h_chars.ensure((R.n_chars + 1) * sizeof(focr_hit_t));
        if (!hc || !hp || !hl || !hh) rc = fail(c, FOCR_ERR_NOMEM, "focr_pipe: page-locked result buffers: hipHostMalloc failed");
        if (rc == FOCR_OK) rc = focr_get_counts(c, hc);
        if (rc == FOCR_OK && job.post) rc = focr_get_lines_into(c, hp, hl, hh);
        R.counts = hc;
        R.page_line_off = job.post ? hp : nullptr;
        R.line_char_off = job.post ? hl : nullptr;
        R.chars = job.post ? hh : nullptr;
        S->fetched = rc == FOCR_OK;
    }
    lk.lock();
    if (P->trace) {
        P->traces.push_back(S->times);
        P->trace_tickets.push_back(t);
    }
    S->rc = rc;
    S->state = PipeSlot::DONE;
    S->finishing = false;
    P->cv.notify_all();
    return rc;
}

}  // namespace focr

using namespace focr;

extern "C" {

int focr_pipe_create2(int device, unsigned n_lanes, unsigned depth, focr_pipe_t **out) {
    if (!out || n_lanes < 1 || n_lanes > 8 || depth < 1 || depth > 4) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_create: 1..8 lanes of 1..4 contexts");
    *out = nullptr;
    focr_pipe *p = new focr_pipe();
    p->device = device;
    p->n_lanes = n_lanes;
    p->depth = depth;
    p->trace = getenv("FOCR_PIPE_TRACE") != nullptr;
    hipDeviceProp_t prop;
    const bool have_prop = hipGetDeviceProperties(&prop, device) == hipSuccess;
    auto bail = [&](int rc) {
        focr_pipe_destroy(p);
        return rc;
    };
    // the contexts side by side (a context is a stream, events, a few device and page-locked allocations: ~8 ms each, and an
    // executor has six of them in front of its first batch)
    {
        (void)hipSetDevice(device);
        (void)hipFree(nullptr);  // the runtime comes up once, here, not in six threads at the same time
        std::vector<focr_ctx *> made(n_lanes * depth, nullptr);
        std::vector<int> rcs(n_lanes * depth, FOCR_OK);
        std::vector<std::thread> th;
        for (unsigned i = 0; i < n_lanes * depth; i++) th.emplace_back([&, i] { rcs[i] = focr_ctx_create(device, &made[i]); });
        for (std::thread &t : th) t.join();
        int bad = FOCR_OK;
        for (unsigned i = 0; i < n_lanes * depth; i++)
            if (rcs[i] != FOCR_OK) bad = rcs[i];
        for (unsigned i = 0; i < n_lanes * depth; i++) {
            if (bad != FOCR_OK) {
                if (made[i]) focr_ctx_destroy(made[i]);
                continue;
            }
            PipeSlot *S = new PipeSlot();
            S->lane = i % n_lanes;
            S->ctx = made[i];
            p->slots.push_back(S);
        }
        if (bad != FOCR_OK) return bail(bad);
    }
    for (unsigned i = 0; i < n_lanes * depth; i++) {
        PipeSlot *S = p->slots[i];
        if (i < n_lanes) {  // the lane: its stream is the first context's; a copy stream and a side stream of its own
            PipeLane *L = new PipeLane();
            L->stream = S->ctx->stream;
            p->lanes.push_back(L);
            if (hipStreamCreateWithFlags(&L->copy_stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&L->io_stream, hipStreamNonBlocking) != hipSuccess)
                return bail(fail(nullptr, FOCR_ERR_NO_DEVICE, "focr_pipe_create: hipStreamCreate failed"));
            S->ctx->io_stream = L->io_stream;
        } else {
            ctx_share_stream(S->ctx, p->lanes[S->lane]->stream, p->lanes[S->lane]->io_stream);
        }
        if (hipEventCreateWithFlags(&S->ev_prefetch, hipEventDisableTiming) != hipSuccess)
            return bail(fail(nullptr, FOCR_ERR_NO_DEVICE, "focr_pipe_create: hipEventCreate failed"));
        // several lanes: the persistent scan kernel takes seven eighths of the CUs and leaves the rest to the other lanes' small
        // kernels (statistics, row tail, ordering, process_hits) — a scan workgroup fills its CU completely, so they run nowhere
        // else while a scan is on.  Measured at BASELINE configs[1], 3 lanes: flat optimum from 192 to 224 of 256 CUs.
        if (n_lanes > 1 && have_prop) focr_ctx_set_scan_cus(S->ctx, (unsigned)(prop.multiProcessorCount - prop.multiProcessorCount / 8));
    }
    for (hipEvent_t &e : p->done_ring)
        if (hipEventCreate(&e) != hipSuccess) return bail(fail(nullptr, FOCR_ERR_NO_DEVICE, "focr_pipe_create: hipEventCreate failed"));
    p->enq = std::thread(enqueue_main, p);
    *out = p;
    return FOCR_OK;
}

int focr_pipe_create(int device, unsigned n_lanes, focr_pipe_t **out) {
    unsigned depth = 2;
    if (const char *e = getenv("FOCR_PIPE_DEPTH")) depth = (unsigned)std::min<unsigned long>(4, std::max<unsigned long>(1, strtoul(e, nullptr, 10)));
    return focr_pipe_create2(device, n_lanes, depth, out);
}

void focr_pipe_destroy(focr_pipe_t *p) {
    if (!p) return;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        // everything submitted is queued and finished first
        p->cv.wait(lk, [&] {
            for (PipeSlot *S : p->slots)
                if (S->state == PipeSlot::QUEUED || S->state == PipeSlot::TAKEN) return false;
            return true;
        });
        for (PipeSlot *S : p->slots)
            if (S->state == PipeSlot::ENQUEUED) (void)complete(p, S, S->ticket, lk);
        p->stop = true;
    }
    p->cv.notify_all();
    if (p->enq.joinable()) p->enq.join();
    if (p->trace) {
        std::vector<size_t> order(p->traces.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return p->trace_tickets[a] < p->trace_tickets[b]; });
        for (size_t i : order) {
            const focr_ticket_times_t &t = p->traces[i];
            fprintf(stderr, "[pipe] ticket %llu slot %u submit %.0f queueing starts +%.0f scan queued +%.0f all queued +%.0f done seen +%.0f device gap %.3f ms\n",
                    (unsigned long long)p->trace_tickets[i], (unsigned)((p->trace_tickets[i] - 1) % p->slots.size()), t.submit_us, t.enqueue_begin_us - t.submit_us,
                    t.scan_queued_us - t.submit_us, t.enqueue_end_us - t.submit_us, t.done_us - t.submit_us, t.device_gap_ms);
        }
    }
    (void)hipSetDevice(p->device);
    for (PipeLane *L : p->lanes) {
        if (L->stream) (void)hipStreamSynchronize(L->stream);
        if (L->copy_stream) (void)hipStreamSynchronize(L->copy_stream);
        if (L->io_stream) (void)hipStreamSynchronize(L->io_stream);
    }
    // the contexts that borrowed a lane's stream go first, the lanes' first contexts (the streams' owners) last
    for (size_t i = p->slots.size(); i-- > 0;) {
        PipeSlot *S = p->slots[i];
        if (S->ev_prefetch) (void)hipEventDestroy(S->ev_prefetch);
        for (PinBuf *b : {&S->h_counts, &S->h_page_off, &S->h_line_off, &S->h_chars}) b->release();
        focr_ctx_destroy(S->ctx);
        delete S;
    }
    for (hipEvent_t e : p->done_ring)
        if (e) (void)hipEventDestroy(e);
    for (PipeLane *L : p->lanes) {
        if (L->copy_stream) (void)hipStreamDestroy(L->copy_stream);
        if (L->io_stream) (void)hipStreamDestroy(L->io_stream);
        if (L->pf_stage) (void)hipFree(L->pf_stage);
        delete L;
    }
    delete p;
}

unsigned focr_pipe_contexts(const focr_pipe_t *p) { return p ? (unsigned)p->slots.size() : 0; }
unsigned focr_pipe_lanes(const focr_pipe_t *p) { return p ? p->n_lanes : 0; }

focr_ctx_t *focr_pipe_context(focr_pipe_t *p, unsigned index) { return (p && index < p->slots.size()) ? p->slots[index]->ctx : nullptr; }

int focr_pipe_bank_upload(focr_pipe_t *p, const focr_template_t *templates, size_t n_templates, const uint8_t *needles, size_t needles_len) {
    if (!p) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_bank_upload: null pipe");
    for (PipeSlot *S : p->slots) {
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv.wait(lk, [&] { return S->state != PipeSlot::QUEUED && S->state != PipeSlot::TAKEN; });
        if (S->state == PipeSlot::ENQUEUED) (void)complete(p, S, S->ticket, lk);  // a batch still in flight finishes with the old bank
    }
    // every context prepares and uploads its own copy (host-side quantisation + a dozen small copies each): side by side
    std::vector<int> rcs(p->slots.size(), FOCR_OK);
    std::vector<std::thread> th;
    for (size_t i = 0; i < p->slots.size(); i++)
        th.emplace_back([&, i] { rcs[i] = focr_bank_upload(p->slots[i]->ctx, templates, n_templates, needles, needles_len); });
    for (std::thread &t : th) t.join();
    for (size_t i = 0; i < rcs.size(); i++)
        if (rcs[i] != FOCR_OK) return fail(nullptr, rcs[i], std::string("focr_pipe_bank_upload: ") + focr_last_error(p->slots[i]->ctx));
    return FOCR_OK;
}

int focr_pipe_submit(focr_pipe_t *p, const void *pages, int pages_on_device, size_t n_pages, size_t r_w, size_t r_h, int invert,
                     float threshold, uint32_t cap, int mode, int process_hits, float anchor_threshold, int32_t overlap,
                     void *chars_out, size_t chars_out_bytes, uint64_t *ticket) {
    if (!p || !ticket) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_submit: bad arguments");
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->announced) {  // batches announced with focr_pipe_prefetch are submitted in the order they were announced
        PipeSlot *Sn = p->slot_of(p->next_ticket);
        if (Sn->pf_ptr != pages || pages_on_device)
            return fail(nullptr, FOCR_ERR_STATE, "focr_pipe_submit: another batch was announced with focr_pipe_prefetch for this ticket");
        p->announced--;
    }
    const uint64_t t = p->next_ticket++;
    const bool last = p->next_is_last;
    p->next_is_last = false;
    PipeSlot *S = p->slot_of(t);
    p->cv.wait(lk, [&] { return S->state == PipeSlot::IDLE; });  // until the slot's previous batch (LANES x DEPTH tickets ago) is released
    S->job = PipeJob{};
    S->job.pages = pages;
    S->job.on_device = pages_on_device;
    S->job.n_pages = n_pages;
    S->job.r_w = r_w;
    S->job.r_h = r_h;
    S->job.invert = invert;
    S->job.threshold = threshold;
    S->job.cap = cap;
    S->job.mode = mode;
    S->job.post = process_hits;
    S->job.anchor_threshold = anchor_threshold;
    S->job.overlap = overlap;
    S->job.last = last;
    S->job.chars_out = chars_out;
    S->job.chars_cap = chars_out_bytes;
    S->ticket = t;
    S->rc = FOCR_OK;
    S->fetched = false;
    S->times = focr_ticket_times_t{};
    S->times.submit_us = p->now_us();
    S->times.device_gap_ms = -1.f;
    S->state = PipeSlot::QUEUED;
    lk.unlock();
    p->cv.notify_all();
    *ticket = t;
    return FOCR_OK;
}

int focr_pipe_prefetch(focr_pipe_t *p, const void *pages, size_t n_pages, size_t r_w, size_t r_h, int invert) {
    if (!p || !pages || !n_pages || !r_w || !r_h) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_prefetch: bad arguments");
    PipeSlot *S = nullptr;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        if (p->announced >= p->slots.size()) return fail(nullptr, FOCR_ERR_STATE, "focr_pipe_prefetch: every context already holds an announced batch");
        S = p->slot_of(p->next_ticket + p->announced);
        // the slot's previous announced batch has taken its pages (its page sets have changed places): the alternate set is free
        // again — it holds the pages of a batch that was released before that one could be submitted
        p->cv.wait(lk, [&] { return S->pf_ptr == nullptr && !S->pf_issuing; });
        S->pf_issuing = true;
        p->announced++;  // reserved: announcements and submits are counted in order
    }
    // (the pipe's lock is NOT held while the staging buffer grows or the copy and the ingest are queued: submit / wait / release of
    // other threads go on meanwhile)
    const int rc = issue_prefetch(p, S, pages, n_pages, r_w, r_h, invert);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        S->pf_issuing = false;
        if (rc == FOCR_OK) {
            S->pf_ptr = pages;
            S->pf_n = n_pages;
            S->pf_w = r_w;
            S->pf_h = r_h;
            S->pf_invert = invert;
        } else {
            p->announced--;
        }
    }
    p->cv.notify_all();
    return rc;
}

int focr_pipe_announce_last(focr_pipe_t *p) {
    if (!p) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_announce_last: null pipe");
    std::lock_guard<std::mutex> lk(p->mu);
    p->next_is_last = true;  // consumed by the next submit
    return FOCR_OK;
}

int focr_pipe_end_of_stream(focr_pipe_t *p) {
    if (!p) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_end_of_stream: null pipe");
    std::lock_guard<std::mutex> lk(p->mu);
    if (p->next_ticket > 1) {  // the newest batch, if the enqueue thread has not taken it yet (batches are queued the moment they are submitted)
        PipeSlot *S = p->slot_of(p->next_ticket - 1);
        if (S->ticket == p->next_ticket - 1 && S->state == PipeSlot::QUEUED) S->job.last = true;
    }
    return FOCR_OK;
}

int focr_pipe_set_fetch(focr_pipe_t *p, int on) {
    if (!p) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_set_fetch: null pipe");
    std::lock_guard<std::mutex> lk(p->mu);
    p->fetch = on != 0;  // read when a batch is completed: set it before submitting
    return FOCR_OK;
}

int focr_pipe_host_results(focr_pipe_t *p, uint64_t ticket, focr_host_results_t *out) {
    if (!p || !ticket || !out) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_host_results: bad arguments");
    PipeSlot *S = p->slot_of(ticket);
    std::unique_lock<std::mutex> lk(p->mu);
    const int rc = complete(p, S, ticket, lk);
    if (rc != FOCR_OK) return rc;
    if (!S->fetched) return fail(S->ctx, FOCR_ERR_STATE, "focr_pipe_host_results: call focr_pipe_set_fetch first");
    *out = S->res;
    return FOCR_OK;
}

int focr_pipe_wait(focr_pipe_t *p, uint64_t ticket, focr_ctx_t **ctx) {
    if (!p || !ticket) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_wait: bad arguments");
    PipeSlot *S = p->slot_of(ticket);
    std::unique_lock<std::mutex> lk(p->mu);
    const int rc = complete(p, S, ticket, lk);
    if (ctx) *ctx = S->ctx;
    return rc;
}

int focr_pipe_ticket_times(focr_pipe_t *p, uint64_t ticket, focr_ticket_times_t *out) {
    if (!p || !ticket || !out) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_ticket_times: bad arguments");
    PipeSlot *S = p->slot_of(ticket);
    std::lock_guard<std::mutex> lk(p->mu);
    if (S->ticket != ticket || S->state != PipeSlot::DONE) return fail(S->ctx, FOCR_ERR_STATE, "focr_pipe_ticket_times: wait for the ticket first (and before its release)");
    *out = S->times;
    return FOCR_OK;
}

int focr_pipe_release(focr_pipe_t *p, uint64_t ticket) {
    if (!p || !ticket) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_release: bad arguments");
    PipeSlot *S = p->slot_of(ticket);
    {
        std::unique_lock<std::mutex> lk(p->mu);
        if (S->ticket != ticket || S->state == PipeSlot::IDLE) return fail(S->ctx, FOCR_ERR_STATE, "focr_pipe_release: ticket is not outstanding");
        (void)complete(p, S, ticket, lk);  // a batch released unseen still finishes first
        if (S->ticket != ticket || S->state != PipeSlot::DONE) return fail(S->ctx, FOCR_ERR_STATE, "focr_pipe_release: ticket is not outstanding");
        S->state = PipeSlot::IDLE;
    }
    p->cv.notify_all();
    return FOCR_OK;
}

}  // extern "C"
