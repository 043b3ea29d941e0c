// fleet.hip — focr_fleet_*: the batches-in-flight executor over several devices (include/focr_ncc.h, "every GPU of the
// node"; SURVEY.md section 8e at the product level, src/ncc.rs:839-847).  One focr_pipe per device; batch k goes to device
// k % n_devices.  A device's pipe sees every n_devices-th batch in order, so the pipe's own ticket of fleet ticket T is
// (T - 1) / n_devices + 1 and nothing has to be looked up.
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

struct focr_fleet {
    std::vector<int> devices;
    std::vector<focr_pipe_t *> pipes;
    unsigned lanes = 0, slots_per_pipe = 0;  // contexts per device's executor: batches it can hold unreleased
    std::mutex mu;  // serialises submit: tickets are handed out in submission order
    uint64_t next_ticket = 1;
    std::atomic<uint64_t> released{0};  // tickets given back with focr_fleet_release
};

using focr::fail;

namespace {
inline focr_pipe_t *pipe_of(const focr_fleet *f, uint64_t t) { return f->pipes[(t - 1) % f->pipes.size()]; }
inline uint64_t pipe_ticket(const focr_fleet *f, uint64_t t) { return (t - 1) / f->pipes.size() + 1; }
}  // namespace

extern "C" {

int focr_fleet_create(const int *devices, unsigned n_devices, unsigned lanes_per_device, focr_fleet_t **out) {
    if (!out || lanes_per_device < 1 || lanes_per_device > 8) return fail(nullptr, FOCR_ERR_INVALID, "focr_fleet_create: bad arguments (1..8 lanes per device)");
    *out = nullptr;
    focr_fleet *f = new focr_fleet();
    if (devices && n_devices) {
        f->devices.assign(devices, devices + n_devices);
    } else {
        const int n = focr_device_count();
        if (n <= 0) {
            delete f;
            return fail(nullptr, FOCR_ERR_NO_DEVICE, "focr_fleet_create: no HIP device available; this library has no CPU fallback");
        }
        for (int d = 0; d < n; d++) f->devices.push_back(d);
    }
    f->lanes = lanes_per_device;
    f->pipes.assign(f->devices.size(), nullptr);
    // the HIP runtime initialises a device on first use (~0.1 s each): create the executors side by side
    std::vector<int> rcs(f->devices.size(), FOCR_OK);
    std::vector<std::string> errs(f->devices.size());
    std::vector<std::thread> th;
    for (size_t i = 0; i < f->devices.size(); i++)
        th.emplace_back([&, i] {
            rcs[i] = focr_pipe_create(f->devices[i], lanes_per_device, &f->pipes[i]);
            if (rcs[i] != FOCR_OK) errs[i] = focr_last_error_global();
        });
    for (std::thread &t : th) t.join();
    for (size_t i = 0; i < rcs.size(); i++)
        if (rcs[i] != FOCR_OK) {
            const int rc = rcs[i];
            const std::string msg = "focr_fleet_create: device " + std::to_string(f->devices[i]) + ": " + errs[i];
            focr_fleet_destroy(f);
            return fail(nullptr, rc, msg);
        }
    f->slots_per_pipe = focr_pipe_contexts(f->pipes[0]);
    *out = f;
    return FOCR_OK;
}

void focr_fleet_destroy(focr_fleet_t *f) {
    if (!f) return;
    for (focr_pipe_t *p : f->pipes)
        if (p) focr_pipe_destroy(p);
    delete f;
}

unsigned focr_fleet_devices(const focr_fleet_t *f) { return f ? (unsigned)f->pipes.size() : 0; }
unsigned focr_fleet_lanes(const focr_fleet_t *f) { return f ? f->lanes : 0; }
unsigned focr_fleet_slots(const focr_fleet_t *f) { return f ? (unsigned)(f->pipes.size() * f->slots_per_pipe) : 0; }
focr_pipe_t *focr_fleet_pipe(focr_fleet_t *f, unsigned index) { return f && index < f->pipes.size() ? f->pipes[index] : nullptr; }

int focr_fleet_device_of(const focr_fleet_t *f, uint64_t ticket) {
    if (!f || !ticket) return -1;
    return f->devices[(ticket - 1) % f->devices.size()];
}

int focr_fleet_bank_upload(focr_fleet_t *f, const focr_template_t *templates, size_t n_templates, const uint8_t *needles, size_t needles_len) {
    if (!f) return fail(nullptr, FOCR_ERR_INVALID, "focr_fleet_bank_upload: null fleet");
    std::vector<int> rcs(f->pipes.size(), FOCR_OK);
    std::vector<std::string> errs(f->pipes.size());
    std::vector<std::thread> th;
    for (size_t i = 0; i < f->pipes.size(); i++)
        th.emplace_back([&, i] {
            rcs[i] = focr_pipe_bank_upload(f->pipes[i], templates, n_templates, needles, needles_len);
            if (rcs[i] != FOCR_OK) errs[i] = focr_last_error_global();
        });
    for (std::thread &t : th) t.join();
    for (size_t i = 0; i < rcs.size(); i++)
        if (rcs[i] != FOCR_OK) return fail(nullptr, rcs[i], "focr_fleet_bank_upload: device " + std::to_string(f->devices[i]) + ": " + errs[i]);
    return FOCR_OK;
}

int focr_fleet_announce_last(focr_fleet_t *f) {
    if (!f) return fail(nullptr, FOCR_ERR_INVALID, "focr_fleet_announce_last: null fleet");
    std::lock_guard<std::mutex> lk(f->mu);  // not between a submit's ticket and its pipe
    for (focr_pipe_t *p : f->pipes) focr_pipe_announce_last(p);  // every device's next batch is its stream's last
    return FOCR_OK;
}

int focr_fleet_end_of_stream(focr_fleet_t *f) {
    if (!f) return fail(nullptr, FOCR_ERR_INVALID, "focr_fleet_end_of_stream: null fleet");
    std::lock_guard<std::mutex> lk(f->mu);  // not between a submit's ticket and its pipe
    for (focr_pipe_t *p : f->pipes) focr_pipe_end_of_stream(p);
    return FOCR_OK;
}

int focr_fleet_set_fetch(focr_fleet_t *f, int on) {
    if (!f) return fail(nullptr, FOCR_ERR_INVALID, "focr_fleet_set_fetch: null fleet");
    for (focr_pipe_t *p : f->pipes) focr_pipe_set_fetch(p, on);
    return FOCR_OK;
}

int focr_fleet_submit(focr_fleet_t *f, const void *pages, int pages_on_device, size_t n_pages, size_t r_w, size_t r_h, int invert, float threshold,
                      uint32_t cap, int mode, int process_hits, float anchor_threshold, int32_t overlap, uint64_t *ticket) {
    if (!f || !ticket) return fail(nullptr, FOCR_ERR_INVALID, "focr_fleet_submit: bad arguments");
    std::lock_guard<std::mutex> lk(f->mu);  // submissions are ordered by definition
    const uint64_t t = f->next_ticket;
    // Every context holding an unreleased batch: the context this batch maps to is the one with the OLDEST ticket, and waiting for it
    // here would be waiting for the caller's own focr_fleet_release — a consumer that submits and retires on one thread would
    // hang with no diagnostic.  Refuse instead (a consumer that releases from a second thread simply submits again).
    if (t - 1 - f->released.load() >= (uint64_t)f->pipes.size() * f->slots_per_pipe)
        return fail(nullptr, FOCR_ERR_STATE, "focr_fleet_submit: every context holds an unreleased batch; release the oldest ticket first");
    uint64_t pt = 0;
    const int rc = focr_pipe_submit(pipe_of(f, t), pages, pages_on_device, n_pages, r_w, r_h, invert, threshold, cap, mode, process_hits, anchor_threshold,
                                    overlap, nullptr, 0, &pt);
    if (rc != FOCR_OK) return rc;
    if (pt != pipe_ticket(f, t)) {  // somebody used the pipe behind the fleet's back: do not leave the stray job queued on it
        focr_ctx_t *ctx = nullptr;
        (void)focr_pipe_wait(pipe_of(f, t), pt, &ctx);
        (void)focr_pipe_release(pipe_of(f, t), pt);
        return fail(nullptr, FOCR_ERR_STATE, "focr_fleet_submit: a pipe of the fleet was submitted to directly");
    }
    f->next_ticket++;
    *ticket = t;
    return FOCR_OK;
}

int focr_fleet_wait(focr_fleet_t *f, uint64_t ticket, focr_ctx_t **ctx) {
    if (!f || !ticket) return fail(nullptr, FOCR_ERR_INVALID, "focr_fleet_wait: bad arguments");
    return focr_pipe_wait(pipe_of(f, ticket), pipe_ticket(f, ticket), ctx);
}

int focr_fleet_host_results(focr_fleet_t *f, uint64_t ticket, focr_host_results_t *out) {
    if (!f || !ticket) return fail(nullptr, FOCR_ERR_INVALID, "focr_fleet_host_results: bad arguments");
    return focr_pipe_host_results(pipe_of(f, ticket), pipe_ticket(f, ticket), out);
}

int focr_fleet_release(focr_fleet_t *f, uint64_t ticket) {
    if (!f || !ticket) return fail(nullptr, FOCR_ERR_INVALID, "focr_fleet_release: bad arguments");
    const int rc = focr_pipe_release(pipe_of(f, ticket), pipe_ticket(f, ticket));
    if (rc == FOCR_OK) f->released.fetch_add(1);
    return rc;
}

}  // extern "C"
