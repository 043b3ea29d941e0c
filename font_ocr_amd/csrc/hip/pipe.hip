// pipe.hip — batches in flight: n contexts on one device, one worker thread each, batches handed out round-robin.
//
// The reference parallelises over pages with a rayon pool (src/ncc.rs:839-847).  On the GPU the unit is a batch of
// pages, and what has to overlap is one batch's latency-bound small kernels (statistics, sorts, verify, ordering,
// process_hits) with another batch's MFMA scan: that needs the batches on different streams, driven by different
// host threads (a batch has one host wait in the steady state, three in a setup's first scan).  This file is that executor,
// so a host in any language gets the overlap from submit / wait / release without writing thread code.  The lanes queue their
// persistent scan kernels in TICKET order (TurnGate, common.h), so batches finish in the order they were submitted.
// DESIGN.md section 5 ("Batches in flight") has the measurements.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "common.h"

namespace focr {

struct PipeJob {
    const void *pages = nullptr;  // nullptr: rescan the pages already resident in the lane's context
    int on_device = 0, invert = 0, mode = 0;
    size_t n_pages = 0, r_w = 0, r_h = 0;
    float threshold = 0.f, anchor_threshold = 0.f;
    uint32_t cap = 0;
    int32_t overlap = 0;
    int post = 1;
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

struct PipeLane {
    enum State { IDLE, QUEUED, RUNNING, DONE };
    focr_ctx *ctx = nullptr;
    // focr_pipe_set_fetch: the batch's results copied to page-locked host memory by the lane itself
    PinBuf h_counts, h_page_off, h_line_off, h_chars;
    focr_host_results_t res{};
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    State state = IDLE;
    uint64_t ticket = 0;
    PipeJob job;
    int rc = FOCR_OK;
    bool stop = false;
    // focr_pipe_prefetch: the NEXT batch's host pages cross PCIe into a staging buffer of the lane's own and are ingested into the
    // context's ALTERNATE page set (pages_alt_ingest, ctx.hip), both on a copy stream, while the lane still works on its current
    // batch; when the announced batch starts, the two page sets change places
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_prefetch = nullptr;  // copy + ingest done
    void *pf_stage = nullptr;
    size_t pf_stage_bytes = 0;
    const void *pf_ptr = nullptr;  // host pages announced and on their way (consumed by the submit that brings the same pointer)
    size_t pf_n = 0, pf_w = 0, pf_h = 0;
    int pf_invert = 0;
};

}  // namespace focr

struct PipeTrace {  // FOCR_PIPE_TRACE=1: host-side time stamps of every job, printed when the pipe is destroyed (us since creation)
    uint64_t ticket;
    unsigned lane;
    double t_start, t_scan_queued, t_post_queued, t_synced, t_done;
};

struct focr_pipe {
    mutable focr::TurnGate gate;  // the lanes' scans are queued in ticket order (common.h)
    bool trace = false;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    mutable std::mutex trace_mu;
    mutable std::vector<PipeTrace> traces;
    double now_us() const { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
    bool fetch = false;
    std::vector<focr::PipeLane *> lanes;
    std::mutex mu;  // guards next_ticket, announced
    uint64_t next_ticket = 1;
    uint64_t announced = 0;  // focr_pipe_prefetch calls not yet consumed by their focr_pipe_submit
};

namespace focr {

static void lane_main(PipeLane *L, const focr_pipe *P) {
    for (;;) {
        PipeJob job;
        {
            std::unique_lock<std::mutex> lk(L->mu);
            L->cv.wait(lk, [&] { return L->stop || L->state == PipeLane::QUEUED; });
            if (L->stop) return;
            L->state = PipeLane::RUNNING;
            job = L->job;
        }
        focr_ctx *c = L->ctx;
        uint64_t ticket;
        {
            std::lock_guard<std::mutex> lk(L->mu);
            ticket = L->ticket;
        }
        c->turn_gate = &P->gate;
        c->turn_ticket = ticket;
        PipeTrace tr{ticket, 0, P->now_us(), 0, 0, 0, 0};
        int rc = FOCR_OK;
        if (job.pages) {
            bool prefetched = false;
            {
                std::lock_guard<std::mutex> lk(L->mu);
                const bool announced = !job.on_device && L->pf_ptr == job.pages;
                prefetched = announced && L->pf_n == job.n_pages && L->pf_w == job.r_w && L->pf_h == job.r_h && L->pf_invert == job.invert;
                // (announced with another geometry or inversion: the announcement is void and the lane uploads the batch itself)
            }
            if (prefetched) {  // the pages are in the context's alternate page set already (or on their way): no copy, no ingest on this stream
                hipError_t e = hipStreamWaitEvent(c->stream, L->ev_prefetch, 0);
                if (e != hipSuccess) rc = fail(c, FOCR_ERR_NO_DEVICE, std::string("focr_pipe: prefetch wait failed: ") + hipGetErrorString(e));
                if (rc == FOCR_OK) rc = pages_alt_swap(c, job.n_pages, job.r_w, job.r_h);
            }
            {
                std::lock_guard<std::mutex> lk(L->mu);
                if (L->pf_ptr == job.pages && !job.on_device) L->pf_ptr = nullptr;  // consumed (or void): the lane may take its next announcement
            }
            L->cv.notify_all();
            if (!prefetched) {
                rc = focr_pages_alloc(c, job.n_pages, job.r_w, job.r_h);
                if (rc == FOCR_OK)
                    rc = job.on_device ? focr_pages_upload_device(c, 0, job.n_pages, job.pages, job.invert)
                                       : focr_pages_upload(c, 0, job.n_pages, (const uint8_t *)job.pages, job.invert);
            }
        }
        if (rc == FOCR_OK) rc = focr_scan(c, job.threshold, job.cap, job.mode);
        P->gate.skip(ticket);  // no-op when the scan took its turn; a batch that ended before must not hold up the later tickets
        tr.t_scan_queued = P->now_us();
        if (rc == FOCR_OK && job.post) rc = focr_process_hits(c, job.anchor_threshold, job.overlap);
        tr.t_post_queued = P->now_us();
        if (rc == FOCR_OK) rc = focr_sync(c);  // the batch's one host wait: scan and process_hits queue everything without waiting
        tr.t_synced = P->now_us();
        if (rc == FOCR_OK && job.post && job.chars_out) {  // copy-out on the context's own stream: ordered, no other queue involved
            const size_t bytes = focr_total_chars(c) * sizeof(focr_hit_t);
            if (bytes > job.chars_cap) {
                rc = fail(c, FOCR_ERR_OVERFLOW, "focr_pipe_submit: chars_out is too small for this batch");
            } else if (bytes) {
                hipError_t e = hipMemcpyAsync(job.chars_out, focr_lines_device_chars(c), bytes, hipMemcpyDeviceToDevice, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess) rc = fail(c, FOCR_ERR_NO_DEVICE, std::string("focr_pipe: copy-out failed: ") + hipGetErrorString(e));
            }
        }
        if (rc == FOCR_OK && P->fetch) {  // results to page-locked host memory here, on the lane's thread and stream
            focr_host_results_t &R = L->res;
            R = focr_host_results_t{};
            R.n_pages = c->n_pages;
            R.n_templates = c->n_templates;
            R.n_matches = focr_total_matches(c);
            R.n_lines = job.post ? focr_total_lines(c) : 0;
            R.n_chars = job.post ? focr_total_chars(c) : 0;
            float ms[6] = {0};
            focr_last_timings(c, ms);
            R.device_ms = ms[5] + ms[4];
            uint32_t *hc = (uint32_t *)L->h_counts.ensure(R.n_pages * R.n_templates * 4 + 16);
            uint64_t *hp = (uint64_t *)L->h_page_off.ensure((R.n_pages + 1) * 8);
            uint64_t *hl = (uint64_t *)L->h_line_off.ensure((R.n_lines + 1) * 8);
            focr_hit_t *hh = (focr_hit_t *)L->h_chars.ensure((R.n_chars + 1) * sizeof(focr_hit_t));
            if (!hc || !hp || !hl || !hh) rc = fail(c, FOCR_ERR_NOMEM, "focr_pipe: page-locked result buffers: hipHostMalloc failed");
            if (rc == FOCR_OK) rc = focr_get_counts(c, hc);
            if (rc == FOCR_OK && job.post) rc = focr_get_lines_into(c, hp, hl, hh);
            R.counts = hc;
            R.page_line_off = job.post ? hp : nullptr;
            R.line_char_off = job.post ? hl : nullptr;
            R.chars = job.post ? hh : nullptr;
        }
        if (P->trace) {
            tr.t_done = P->now_us();
            for (size_t i = 0; i < P->lanes.size(); i++)
                if (P->lanes[i] == L) tr.lane = (unsigned)i;
            std::lock_guard<std::mutex> lk(P->trace_mu);
            P->traces.push_back(tr);
        }
        {
            std::lock_guard<std::mutex> lk(L->mu);
            L->rc = rc;
            L->state = PipeLane::DONE;
        }
        L->cv.notify_all();
    }
}

}  // namespace focr

using namespace focr;

extern "C" {

int focr_pipe_create(int device, unsigned n_contexts, focr_pipe_t **out) {
    if (!out || n_contexts < 1 || n_contexts > 8) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_create: 1..8 contexts");
    *out = nullptr;
    focr_pipe *p = new focr_pipe();
    p->trace = getenv("FOCR_PIPE_TRACE") != nullptr;
    hipDeviceProp_t prop;
    for (unsigned i = 0; i < n_contexts; i++) {
        PipeLane *L = new PipeLane();
        int rc = focr_ctx_create(device, &L->ctx);
        if (rc != FOCR_OK) {
            delete L;
            focr_pipe_destroy(p);
            return rc;
        }
        // several batches in flight: the persistent scan kernel takes seven eighths of the CUs and leaves the rest to the other
        // batches' small kernels (statistics, row tail, ordering, process_hits) — a scan workgroup fills its CU completely, so
        // they run nowhere else while a scan is on.  Measured at BASELINE configs[1], 3 batches in flight: round 3 (tail 0.77 ms of
        // full-chip time per batch) 176 / 192 / 208 / 224 / 240 / 256 CUs -> 27.7 / 28.5-29.8 / 28.3-29.1 / 28.9-29.4 / 25.5 / 26.1 Gpx/s
        // (gpurun_out/r3_g_*, r3_h_*, r3_j_*: flat from 192 to 224); round 2 (tail 1.3 ms) had a sharp optimum at 192.
        if (n_contexts > 1 && hipGetDeviceProperties(&prop, device) == hipSuccess)
            focr_ctx_set_scan_cus(L->ctx, (unsigned)(prop.multiProcessorCount - prop.multiProcessorCount / 8));
        L->worker = std::thread(lane_main, L, p);
        p->lanes.push_back(L);
    }
    *out = p;
    return FOCR_OK;
}

void focr_pipe_destroy(focr_pipe_t *p) {
    if (!p) return;
    if (p->trace) {
        std::sort(p->traces.begin(), p->traces.end(), [](const PipeTrace &a, const PipeTrace &b) { return a.ticket < b.ticket; });
        for (const PipeTrace &t : p->traces)
            fprintf(stderr, "[pipe] ticket %llu lane %u start %.0f scan queued +%.0f post queued +%.0f synced +%.0f done +%.0f\n", (unsigned long long)t.ticket, t.lane, t.t_start,
                    t.t_scan_queued - t.t_start, t.t_post_queued - t.t_start, t.t_synced - t.t_start, t.t_done - t.t_start);
    }
    for (PipeLane *L : p->lanes) {
        {
            std::unique_lock<std::mutex> lk(L->mu);
            L->cv.wait(lk, [&] { return L->state != PipeLane::QUEUED && L->state != PipeLane::RUNNING; });
            L->stop = true;
        }
        L->cv.notify_all();
        if (L->worker.joinable()) L->worker.join();
        (void)hipSetDevice(L->ctx->device);
        if (L->copy_stream) {
            (void)hipStreamSynchronize(L->copy_stream);
            (void)hipStreamDestroy(L->copy_stream);
        }
        if (L->ev_prefetch) (void)hipEventDestroy(L->ev_prefetch);
        if (L->pf_stage) (void)hipFree(L->pf_stage);
        for (PinBuf *b : {&L->h_counts, &L->h_page_off, &L->h_line_off, &L->h_chars}) b->release();
        focr_ctx_destroy(L->ctx);
        delete L;
    }
    delete p;
}

unsigned focr_pipe_contexts(const focr_pipe_t *p) { return p ? (unsigned)p->lanes.size() : 0; }

focr_ctx_t *focr_pipe_context(focr_pipe_t *p, unsigned index) {
    return (p && index < p->lanes.size()) ? p->lanes[index]->ctx : nullptr;
}

int focr_pipe_bank_upload(focr_pipe_t *p, const focr_template_t *templates, size_t n_templates, const uint8_t *needles,
                          size_t needles_len) {
    if (!p) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_bank_upload: null pipe");
    for (PipeLane *L : p->lanes) {
        {
            std::unique_lock<std::mutex> lk(L->mu);
            L->cv.wait(lk, [&] { return L->state == PipeLane::IDLE || L->state == PipeLane::DONE; });
        }
        int rc = focr_bank_upload(L->ctx, templates, n_templates, needles, needles_len);
        if (rc != FOCR_OK) return rc;
    }
    return FOCR_OK;
}

int focr_pipe_submit(focr_pipe_t *p, const void *pages, int pages_on_device, size_t n_pages, size_t r_w, size_t r_h, int invert,
                     float threshold, uint32_t cap, int mode, int process_hits, float anchor_threshold, int32_t overlap,
                     void *chars_out, size_t chars_out_bytes, uint64_t *ticket) {
    if (!p || !ticket) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_submit: bad arguments");
    uint64_t t;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (p->announced) {  // batches announced with focr_pipe_prefetch are submitted in the order they were announced
            PipeLane *Ln = p->lanes[(p->next_ticket - 1) % p->lanes.size()];
            std::lock_guard<std::mutex> lk2(Ln->mu);
            if (Ln->pf_ptr != pages || pages_on_device)
                return fail(nullptr, FOCR_ERR_STATE, "focr_pipe_submit: another batch was announced with focr_pipe_prefetch for this ticket");
            p->announced--;
        }
        t = p->next_ticket++;
        p->gate.newest.store(t, std::memory_order_relaxed);
        p->gate.closing.store(false, std::memory_order_relaxed);
    }
    PipeLane *L = p->lanes[(t - 1) % p->lanes.size()];
    {
        std::unique_lock<std::mutex> lk(L->mu);
        L->cv.wait(lk, [&] { return L->state == PipeLane::IDLE; });  // until the lane's previous batch is released
        L->job.pages = pages;
        L->job.on_device = pages_on_device;
        L->job.n_pages = n_pages;
        L->job.r_w = r_w;
        L->job.r_h = r_h;
        L->job.invert = invert;
        L->job.threshold = threshold;
        L->job.cap = cap;
        L->job.mode = mode;
        L->job.post = process_hits;
        L->job.anchor_threshold = anchor_threshold;
        L->job.overlap = overlap;
        L->job.chars_out = chars_out;
        L->job.chars_cap = chars_out_bytes;
        L->ticket = t;
        L->state = PipeLane::QUEUED;
    }
    L->cv.notify_all();
    *ticket = t;
    return FOCR_OK;
}

int focr_pipe_prefetch(focr_pipe_t *p, const void *pages, size_t n_pages, size_t r_w, size_t r_h, int invert) {
    if (!p || !pages || !n_pages || !r_w || !r_h) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_prefetch: bad arguments");
    std::lock_guard<std::mutex> plk(p->mu);  // announcements and submits are serialised
    if (p->announced >= p->lanes.size()) return fail(nullptr, FOCR_ERR_STATE, "focr_pipe_prefetch: every lane already holds an announced batch");
    PipeLane *L = p->lanes[(p->next_ticket + p->announced - 1) % p->lanes.size()];
    focr_ctx *c = L->ctx;
    const size_t bytes = n_pages * r_w * r_h;
    FOCR_HIP((focr_ctx *)nullptr, hipSetDevice(c->device));
    std::unique_lock<std::mutex> lk(L->mu);
    // the lane's previous announced batch has taken its pages (its page sets have changed places): the alternate set is free again —
    // it was the lane's current set two batches ago, and every batch ends with focr_sync
    L->cv.wait(lk, [&] { return L->pf_ptr == nullptr; });
    if (!L->copy_stream) {
        FOCR_HIP((focr_ctx *)nullptr, hipStreamCreateWithFlags(&L->copy_stream, hipStreamNonBlocking));
        FOCR_HIP((focr_ctx *)nullptr, hipEventCreateWithFlags(&L->ev_prefetch, hipEventDisableTiming));
    }
    if (L->pf_stage_bytes < bytes) {
        FOCR_HIP((focr_ctx *)nullptr, hipStreamSynchronize(L->copy_stream));  // the previous announcement's ingest has read the old buffer
        if (L->pf_stage) (void)hipFree(L->pf_stage);
        L->pf_stage = nullptr;
        L->pf_stage_bytes = 0;
        if (hipMalloc(&L->pf_stage, bytes) != hipSuccess) return fail(nullptr, FOCR_ERR_NOMEM, "focr_pipe_prefetch: hipMalloc failed");
        L->pf_stage_bytes = bytes;
    }
    // copy, then ingest, in stream order on the lane's copy stream (the previous announcement's ingest, which read the staging buffer,
    // is ahead of this copy on the same stream)
    FOCR_HIP((focr_ctx *)nullptr, hipMemcpyAsync(L->pf_stage, pages, bytes, hipMemcpyHostToDevice, L->copy_stream));
    if (int rc = pages_alt_ingest(c, L->pf_stage, n_pages, r_w, r_h, invert, L->copy_stream)) return rc;
    FOCR_HIP((focr_ctx *)nullptr, hipEventRecord(L->ev_prefetch, L->copy_stream));
    L->pf_ptr = pages;
    L->pf_n = n_pages;
    L->pf_w = r_w;
    L->pf_h = r_h;
    L->pf_invert = invert;
    p->announced++;
    return FOCR_OK;
}

int focr_pipe_end_of_stream(focr_pipe_t *p) {
    if (!p) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_end_of_stream: null pipe");
    std::lock_guard<std::mutex> lk(p->mu);
    p->gate.closing.store(true, std::memory_order_relaxed);  // until the next submit
    return FOCR_OK;
}

int focr_pipe_set_fetch(focr_pipe_t *p, int on) {
    if (!p) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_set_fetch: null pipe");
    p->fetch = on != 0;  // read by the lanes when a batch completes: set it before submitting
    return FOCR_OK;
}

int focr_pipe_host_results(focr_pipe_t *p, uint64_t ticket, focr_host_results_t *out) {
    if (!p || !ticket || !out) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_host_results: bad arguments");
    PipeLane *L = p->lanes[(ticket - 1) % p->lanes.size()];
    std::unique_lock<std::mutex> lk(L->mu);
    if (L->ticket != ticket || L->state == PipeLane::IDLE) return fail(L->ctx, FOCR_ERR_STATE, "focr_pipe_host_results: ticket is not outstanding");
    L->cv.wait(lk, [&] { return L->state == PipeLane::DONE; });
    if (L->rc != FOCR_OK) return L->rc;
    if (!p->fetch) return fail(L->ctx, FOCR_ERR_STATE, "focr_pipe_host_results: call focr_pipe_set_fetch first");
    *out = L->res;
    return FOCR_OK;
}

int focr_pipe_wait(focr_pipe_t *p, uint64_t ticket, focr_ctx_t **ctx) {
    if (!p || !ticket) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_wait: bad arguments");
    PipeLane *L = p->lanes[(ticket - 1) % p->lanes.size()];
    std::unique_lock<std::mutex> lk(L->mu);
    if (L->ticket != ticket || L->state == PipeLane::IDLE) return fail(L->ctx, FOCR_ERR_STATE, "focr_pipe_wait: ticket is not outstanding");
    L->cv.wait(lk, [&] { return L->state == PipeLane::DONE; });
    if (ctx) *ctx = L->ctx;
    return L->rc;
}

int focr_pipe_release(focr_pipe_t *p, uint64_t ticket) {
    if (!p || !ticket) return fail(nullptr, FOCR_ERR_INVALID, "focr_pipe_release: bad arguments");
    PipeLane *L = p->lanes[(ticket - 1) % p->lanes.size()];
    {
        std::unique_lock<std::mutex> lk(L->mu);
        if (L->ticket != ticket || L->state == PipeLane::IDLE) return fail(L->ctx, FOCR_ERR_STATE, "focr_pipe_release: ticket is not outstanding");
        L->cv.wait(lk, [&] { return L->state == PipeLane::DONE; });
        L->state = PipeLane::IDLE;
    }
    L->cv.notify_all();
    return FOCR_OK;
}

}  // extern "C"
