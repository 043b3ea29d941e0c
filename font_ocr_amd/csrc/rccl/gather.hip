// gather.hip — libfocr_rccl.so: the match-list gather over RCCL for a single process driving several GPUs
// (include/focr_rccl.h).  Payloads are small (about 34 KB of characters per 608x720 page), so this is
// latency-, not bandwidth-critical: one grouped send/recv per rank and batch.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <vector>

#include "focr_rccl.h"

struct focr_gather {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
};

namespace {
std::mutex g_mu;
std::string g_err;
int fail(const std::string &m) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_err = m;
    return 1;
}
}  // namespace

extern "C" {

const char *focr_gather_last_error(void) {
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(g_mu);
    copy = g_err;
    return copy.c_str();
}

int focr_gather_create(const int *devices, int n_devices, focr_gather_t **out) {
    if (!devices || n_devices < 1 || !out) return fail("focr_gather_create: bad arguments");
    *out = nullptr;
    focr_gather *g = new focr_gather();
    g->devices.assign(devices, devices + n_devices);
    g->comms.assign(n_devices, nullptr);
    g->streams.assign(n_devices, nullptr);
    ncclResult_t r = ncclCommInitAll(g->comms.data(), n_devices, g->devices.data());
    if (r != ncclSuccess) {
        delete g;
        return fail(std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
    }
    for (int i = 0; i < n_devices; i++) {
        if (hipSetDevice(g->devices[i]) != hipSuccess || hipStreamCreateWithFlags(&g->streams[i], hipStreamNonBlocking) != hipSuccess) {
            focr_gather_destroy(g);
            return fail("focr_gather_create: hipStreamCreate failed");
        }
    }
    *out = g;
    return 0;
}

void focr_gather_destroy(focr_gather_t *g) {
    if (!g) return;
    for (size_t i = 0; i < g->devices.size(); i++) {
        (void)hipSetDevice(g->devices[i]);
        if (g->streams[i]) {
            (void)hipStreamSynchronize(g->streams[i]);
            (void)hipStreamDestroy(g->streams[i]);
        }
        if (g->comms[i]) (void)ncclCommDestroy(g->comms[i]);
    }
    delete g;
}

int focr_gather_bytes(focr_gather_t *g, const void *const *d_src, const size_t *bytes, void *d_dst, size_t dst_bytes) {
    if (!g || !d_src || !bytes || !d_dst) return fail("focr_gather_bytes: bad arguments");
    const int n = (int)g->devices.size();
    size_t total = 0;
    for (int i = 0; i < n; i++) total += bytes[i];
    if (total > dst_bytes) return fail("focr_gather_bytes: destination too small");
    ncclResult_t r = ncclGroupStart();
    size_t off = 0;
    for (int i = 0; i < n && r == ncclSuccess; i++) {  // every rank sends its block to the root, the root posts the matching receive
        if (bytes[i]) {
            r = ncclSend(d_src[i], bytes[i], ncclChar, 0, g->comms[i], g->streams[i]);
            if (r == ncclSuccess) r = ncclRecv((char *)d_dst + off, bytes[i], ncclChar, i, g->comms[0], g->streams[0]);
        }
        off += bytes[i];
    }
    const ncclResult_t e = ncclGroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) return fail(std::string("focr_gather_bytes: ") + ncclGetErrorString(r));
    for (int i = 0; i < n; i++) {
        if (hipSetDevice(g->devices[i]) != hipSuccess || hipStreamSynchronize(g->streams[i]) != hipSuccess) return fail("focr_gather_bytes: stream synchronise failed");
    }
    return 0;
}

}  // extern "C"
