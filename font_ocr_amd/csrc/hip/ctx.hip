// ctx.hip — context, bank upload, resident pages, result read-back (include/focr_ncc.h layer 2).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace focr {

static std::mutex g_err_mu;
static std::string g_err;

// Size estimates shared between the contexts of a process (ctx.hip: focr_scan, finish_results), keyed by the setup's signature
// (bank content, device, geometry, threshold, cap, mode): bounds only — a count above its bound redoes the batch exactly.
struct SharedEstimate {
    size_t cand, hits;
    uint32_t row_max, seg_shift;
};
static std::mutex g_est_mu;
static std::unordered_map<uint64_t, SharedEstimate> g_est;
static hipEvent_t g_base_event[64] = {};  // per device: the origin of focr_debug_phase_stamps (guarded by g_est_mu)

void set_global_error(const std::string &s) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err = s;
}

int fail(focr_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    set_global_error(msg);
    return code;
}

// tight luma8 pages -> pitched ink-high pages (image_to_u8, src/ncc.rs:887-892, on the device) + their int8 copy.
// One workgroup of 64 threads per page row; 4 pixels per thread and step when the rows are dword-aligned, else bytes.
__global__ __launch_bounds__(64) void ingest_pages(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint8_t *__restrict__ dst_i8, uint32_t r_w,
                                                   uint32_t r_h, size_t pitch, size_t rows_alloc, size_t first, size_t n_rows, int invert, int dwords) {
    const uint32_t flip = invert ? 0xffffffffu : 0u;
    for (size_t row = blockIdx.x; row < n_rows; row += gridDim.x) {
        const size_t p = row / r_h, y = row % r_h;
        const uint8_t *s = src + row * r_w;
        const size_t o = ((first + p) * rows_alloc + y) * pitch;
        if (dwords) {  // r_w % 4 == 0 and src 4-byte aligned (pitch is a multiple of 64)
            for (uint32_t x = threadIdx.x; x < r_w / 4; x += 64) {
                const uint32_t v = reinterpret_cast<const uint32_t *>(s)[x] ^ flip;  // 255 - v per byte
                reinterpret_cast<uint32_t *>(dst + o)[x] = v;
                reinterpret_cast<uint32_t *>(dst_i8 + o)[x] = v ^ 0x80808080u;  // ink - 128 as int8: the prefilter's operand
            }
        } else {
            for (uint32_t x = threadIdx.x; x < r_w; x += 64) {
                const uint8_t v = (uint8_t)(s[x] ^ (uint8_t)flip);
                dst[o + x] = v;
                dst_i8[o + x] = v ^ 0x80;
            }
        }
    }
}

__global__ void debug_rnorm_kernel(const uint32_t *s, const uint64_t *s2, const uint32_t *n, size_t cnt, double *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) out[i] = window_rnorm(s[i], s2[i], (double)n[i]);
}

// split-batch mode: keep only the hits that survive their call's cap, appended in order
__global__ void append_kept_hits(const uint64_t *__restrict__ hkeys, const float *__restrict__ hsims, const uint8_t *__restrict__ keep,
                                 const uint64_t *__restrict__ pos, size_t n, uint64_t *__restrict__ out_keys,
                                 float *__restrict__ out_sims) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    out_keys[pos[i]] = hkeys[i];
    out_sims[pos[i]] = hsims[i];
}

__global__ void widen_u8_to_u64(const uint8_t *__restrict__ in, size_t n, uint64_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

__global__ void widen_u32_to_u64(const uint32_t *__restrict__ in, size_t n, uint64_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) out[i] = i < n ? in[i] : 0;
}

template <typename T>
static void free_dev(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

static void free_bank(focr_ctx *c) {
    free_dev(c->d_tconst);
    free_dev(c->d_direct_bank);
    free_dev(c->d_qbank);
    free_dev(c->d_tglobal);
    free_dev(c->d_order_of);
    c->mfma_c_scale.clear();
    c->mfma_e_max.clear();
    c->mfma_rho_max.clear();
    free_dev(c->d_needles);
    free_dev(c->d_needle_off);
    free_dev(c->d_needles16);
    free_dev(c->d_needle16_row);
    free_dev(c->d_vmeta);
    free_dev(c->d_vrows_t);
    free_dev(c->d_vmeta_t);
    c->h_vrow0_t.clear();
    c->vrow_bytes = 0;
    free_dev(c->d_t_w);
    free_dev(c->d_t_h);
    free_dev(c->d_t_letter);
    c->classes.clear();
    c->h_tconst.clear();
    c->h_templates.clear();
    c->direct_bank_off.clear();
    c->h_needle_off.clear();
    c->n_templates = 0;
}

static void free_results(focr_ctx *c) {
    free_dev(c->d_hit_keys);
    free_dev(c->d_hit_keys_alt);
    free_dev(c->d_hit_sims);
    free_dev(c->d_hit_sims_alt);
    free_dev(c->d_cand);
    free_dev(c->d_cand_alt);
    c->cand_alt_capacity = 0;
    c->scan_flags.release();
    c->scan_pos.release();
    c->scan_live.release();
    c->scan_live_list.release();
    for (auto *b : {&c->ord_k2, &c->ord_k2_alt, &c->ord_v, &c->ord_v_alt, &c->ord_keep, &c->acc_matches, &c->acc_seg_count,
                    &c->acc_hkeys, &c->acc_hsims, &c->rows_hits, &c->rows_hbase, &c->rows_big})
        b->release();
    free_dev(c->d_L);
    free_dev(c->d_planes);
    c->planes_bytes = 0;
    free_dev(c->d_sort_tmp);
    free_dev(c->d_seg_count);
    free_dev(c->d_seg_start);
    free_dev(c->d_seg_offset);
    free_dev(c->d_matches);
    c->post_line_be.release();
    for (auto *b : {&c->post_keep, &c->post_choice, &c->post_owner, &c->post_packed, &c->post_scanned, &c->post_page_off,
                    &c->post_line_off, &c->post_chars})
        b->release();
    c->hit_capacity = c->cand_capacity = c->L_bytes = c->sort_tmp_bytes = c->seg_alloc = c->matches_alloc = 0;
    c->scanned = c->processed = false;
}

}  // namespace focr

using namespace focr;

void *focr_ctx::DevBuf::ensure(focr_ctx *c, size_t want) {
    if (want <= bytes && p) return p;
    (void)hipStreamSynchronize(c->stream);
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    size_t grow = want + want / 4 + 256;
    if (hipMalloc(&p, grow) != hipSuccess) {
        p = nullptr;
        return nullptr;
    }
    bytes = grow;
    return p;
}

// grow while preserving the first `keep_bytes` bytes
void *focr_ctx::DevBuf::ensure_keep(focr_ctx *c, size_t want, size_t keep_bytes) {
    if (want <= bytes && p) return p;
    (void)hipStreamSynchronize(c->stream);
    void *q = nullptr;
    size_t grow = want + want / 2 + 256;
    if (hipMalloc(&q, grow) != hipSuccess) return nullptr;
    if (p && keep_bytes) (void)hipMemcpy(q, p, keep_bytes, hipMemcpyDeviceToDevice);
    if (p) (void)hipFree(p);
    p = q;
    bytes = grow;
    return p;
}

void focr_ctx::DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

void focr_ctx::launch_begin(const char *name, uint32_t n_t, uint64_t alg, uint64_t issued) {
    focr_launch_info_t li{};
    snprintf(li.name, sizeof li.name, "%s", name);
    li.n_templates = n_t;
    li.alg_macs = alg;
    li.issued_macs = issued;
    launches.push_back(li);
    while (launch_events.size() < 2 * launches.size()) {
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        launch_events.push_back(e);
    }
    (void)hipEventRecord(launch_events[2 * (launches.size() - 1)], stream);
}

void focr_ctx::launch_end() { (void)hipEventRecord(launch_events[2 * (launches.size() - 1) + 1], stream); }

void focr_ctx::launches_collect() {
    for (size_t i = 0; i < launches.size(); i++)
        if (hipEventElapsedTime(&launches[i].ms, launch_events[2 * i], launch_events[2 * i + 1]) != hipSuccess) launches[i].ms = 0.f;
}

extern "C" {

size_t focr_last_launches(focr_ctx_t *c, focr_launch_info_t *out, size_t cap) {
    if (!c || finish_results(c) != FOCR_OK) return 0;
    for (size_t i = 0; out && i < c->launches.size() && i < cap; i++) out[i] = c->launches[i];
    return c->launches.size();
}

const char *focr_last_error_global(void) {
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(g_err_mu);
    copy = g_err;
    return copy.c_str();
}

const char *focr_last_error(const focr_ctx_t *ctx) { return ctx ? ctx->err.c_str() : focr_last_error_global(); }

int focr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int focr_ctx_create(int device, focr_ctx_t **out) {
    if (!out) return fail(nullptr, FOCR_ERR_INVALID, "focr_ctx_create: null out");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, FOCR_ERR_NO_DEVICE,
                    std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0") +
                        "); this library has no CPU fallback");
    if (device < 0 || device >= n) return fail(nullptr, FOCR_ERR_INVALID, "focr_ctx_create: bad device index");
    focr_ctx *c = new focr_ctx();
    c->device = device;
    auto init = [&]() -> int {
        FOCR_HIP(c, hipSetDevice(device));
        int cus = 0;
        FOCR_HIP(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        c->n_cus = (unsigned)std::max(cus, 1);
        c->chunked_verify = getenv("FOCR_VERIFY_GLOBAL") == nullptr;
        FOCR_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->io_stream = c->stream;
        {
            std::lock_guard<std::mutex> lk(g_est_mu);
            hipEvent_t &b = g_base_event[(unsigned)device % 64];
            if (!b && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(b, c->stream);
        }
        for (auto &ev : c->ev) FOCR_HIP(c, hipEventCreate(&ev));
        FOCR_HIP(c, hipMalloc(&c->d_counter, COUNTER_BYTES));
        FOCR_HIP(c, hipMemsetAsync(c->d_counter, 0, COUNTER_BYTES, c->stream));
        FOCR_HIP(c, hipMalloc((void **)&c->d_res, 8 * sizeof(uint64_t)));
        FOCR_HIP(c, hipMemsetAsync(c->d_res, 0, 8 * sizeof(uint64_t), c->stream));
        FOCR_HIP(c, hipHostMalloc((void **)&c->h_res, 8 * sizeof(uint64_t), hipHostMallocDefault));
        FOCR_HIP(c, hipHostMalloc((void **)&c->h_live, 40 * sizeof(uint32_t), hipHostMallocDefault));
        memset(c->h_res, 0, 8 * sizeof(uint64_t));
        memset(c->h_live, 0, 40 * sizeof(uint32_t));
        return FOCR_OK;
    };
    int rc = init();
    if (rc != FOCR_OK) {  // the message is already in focr_last_error_global()
        focr_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return FOCR_OK;
}

void focr_ctx_destroy(focr_ctx_t *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_bank(c);
    free_results(c);
    free_dev(c->d_pages);
    free_dev(c->d_pages_i8);
    free_dev(c->alt.u8);
    free_dev(c->alt.i8);
    free_dev(c->d_stage);
    free_dev(c->d_counter);
    free_dev(c->d_res);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->h_live) (void)hipHostFree(c->h_live);
    for (auto &ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->launch_events)
        if (ev) (void)hipEventDestroy(ev);
    if (c->stream && c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int focr_ctx_set_scan_cus(focr_ctx_t *c, unsigned max_cus) {
    if (!c) return fail(c, FOCR_ERR_INVALID, "focr_ctx_set_scan_cus: null context");
    c->scan_cus = max_cus;
    return FOCR_OK;
}

int focr_ctx_set_prefilter(focr_ctx_t *c, int mode) {
    if (!c || (mode != FOCR_PREFILTER_AUTO && mode != FOCR_PREFILTER_ONE_STAGE && mode != FOCR_PREFILTER_LEGACY))
        return fail(c, FOCR_ERR_INVALID, "focr_ctx_set_prefilter: bad arguments");
    c->prefilter = mode;
    return FOCR_OK;
}

int focr_ctx_set_row_tail(focr_ctx_t *c, int on) {
    if (!c) return fail(c, FOCR_ERR_INVALID, "focr_ctx_set_row_tail: null context");
    if (on < 0 || on > 1) return fail(c, FOCR_ERR_INVALID, "focr_ctx_set_row_tail: 0 (legacy tail) or 1 (hits-first row tail, the default)");
    c->tail_mode = on;
    c->est_row_max = 0;
    c->est_cand = c->est_hits = 0;  // the next scan runs with exact sizes
    return FOCR_OK;
}

int focr_ctx_set_column_drop(focr_ctx_t *c, int on) {
    if (!c) return fail(c, FOCR_ERR_INVALID, "focr_ctx_set_column_drop: null context");
    c->column_drop = on != 0;
    return FOCR_OK;
}

int focr_debug_force_split(focr_ctx_t *c, int on) {
    if (!c) return FOCR_ERR_INVALID;
    c->force_split = on != 0;
    return FOCR_OK;
}

// Diagnostic: where the phases of the context's last batch lie on the DEVICE's clock — milliseconds since a per-device base event
// (recorded when the first context of the device is created): [0] statistics start, [1] statistics end, [2] scan kernels end,
// [3] verify end, [4] ordering end, [5] process_hits start, [6] process_hits end, [7] start of the dominant scan launch, [8] its end.
// What a kernel trace shows, without a profiler in the process (tools/r5_phase_dump: the two rhythms of DESIGN.md section 5).
int focr_debug_phase_stamps(focr_ctx_t *c, double out[9]) {
    if (!c || !out) return FOCR_ERR_INVALID;
    if (int rc = finish_results(c)) return rc;
    FOCR_HIP(c, hipSetDevice(c->device));
    hipEvent_t base = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_est_mu);
        base = g_base_event[(unsigned)c->device % 64];
    }
    for (int i = 0; i < 9; i++) out[i] = -1.0;
    if (!base) return FOCR_OK;
    for (int i = 0; i < 7; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, base, c->ev[i]) == hipSuccess) out[i] = ms;
        else (void)hipGetLastError();
    }
    size_t best = 0;
    for (size_t i = 1; i < c->launches.size(); i++)
        if (c->launches[i].alg_macs > c->launches[best].alg_macs) best = i;
    if (!c->launches.empty() && c->launch_events.size() >= 2 * c->launches.size()) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, base, c->launch_events[2 * best]) == hipSuccess) out[7] = ms;
        if (hipEventElapsedTime(&ms, base, c->launch_events[2 * best + 1]) == hipSuccess) out[8] = ms;
        (void)hipGetLastError();
    }
    return FOCR_OK;
}

int focr_debug_set_tail_grid(focr_ctx_t *c, uint32_t num, uint32_t den) {
    if (!c) return FOCR_ERR_INVALID;
    c->dbg_grid_num = num;
    c->dbg_grid_den = den;
    return FOCR_OK;
}

int focr_debug_set_stats_form(focr_ctx_t *c, int form) {
    if (!c || form < 0 || form > 1) return FOCR_ERR_INVALID;
    c->dbg_stats_form = form;
    return FOCR_OK;
}

int focr_debug_planes(focr_ctx_t *c, uint16_t *out, size_t capacity, size_t *n_values) {
    if (!c || !n_values) return FOCR_ERR_INVALID;
    FOCR_HIP(c, hipSetDevice(c->device));
    if (int rc = focr_sync(c)) return rc;
    *n_values = c->planes_bytes / 2;
    if (!out) return FOCR_OK;
    if (capacity < *n_values) return fail(c, FOCR_ERR_INVALID, "focr_debug_planes: buffer too small");
    if (*n_values) FOCR_HIP(c, hipMemcpy(out, c->d_planes, *n_values * 2, hipMemcpyDeviceToHost));
    return FOCR_OK;
}

int focr_sync(focr_ctx_t *c) {
    if (!c) return FOCR_ERR_INVALID;
    FOCR_HIP(c, hipSetDevice(c->device));
    if (c->sizes_pending || c->post_pending) return finish_results(c);  // waits for the batch itself
    return wait_batch(c);
}

int focr_bank_upload(focr_ctx_t *c, const focr_template_t *templates, size_t n_templates, const uint8_t *needles,
                     size_t needles_len) {
    if (!c || !templates || !n_templates || !needles) return fail(c, FOCR_ERR_INVALID, "focr_bank_upload: bad arguments");
    if (n_templates > 65535) return fail(c, FOCR_ERR_INVALID, "focr_bank_upload: more than 65535 templates");
    for (size_t t = 0; t < n_templates; t++) {
        const focr_template_t &d = templates[t];
        if (d.n_w == 0 || d.n_h == 0) return fail(c, FOCR_ERR_INVALID, "focr_bank_upload: empty template");
        if (d.n_w > 32)  // the reference panics above 16 ("not handled", src/ncc.rs:392); 17..32 is this build's extension
            return fail(c, FOCR_ERR_INVALID, "focr_bank_upload: template wider than 32 px is not handled");
        if (d.n_h > 255) return fail(c, FOCR_ERR_INVALID, "focr_bank_upload: template taller than 255 px is not handled");
        if ((size_t)d.offset + (size_t)d.n_w * d.n_h > needles_len)
            return fail(c, FOCR_ERR_INVALID, "focr_bank_upload: template offset out of range");
    }
    FOCR_HIP(c, hipSetDevice(c->device));
    FOCR_HIP(c, hipStreamSynchronize(c->stream));
    free_bank(c);
    c->scanned = c->processed = false;
    c->sizes_pending = c->post_pending = false;
    c->bank_gen++;
    {  // content id of the bank (FNV-1a over the records and the pixels): contexts that hold the same bank share their size estimates
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const void *p, size_t n) {
            const uint8_t *b = (const uint8_t *)p;
            for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
        };
        for (size_t t = 0; t < n_templates; t++) {
            const focr_template_t &d = templates[t];
            const uint32_t rec[3] = {d.letter, (uint32_t)d.n_w << 16 | d.n_h, d.offset};
            mix(rec, sizeof rec);
        }
        mix(needles, needles_len);
        c->bank_hash = h ^ (uint64_t)c->column_drop;
    }
    std::vector<uint32_t> direct;
    std::vector<uint8_t> dense;
    bank_host_prepare(c, templates, n_templates, needles, direct, dense);
    std::vector<uint32_t> tw(n_templates), th(n_templates), tl(n_templates);
    for (size_t t = 0; t < n_templates; t++) {
        tw[t] = templates[t].n_w;
        th[t] = templates[t].n_h;
        tl[t] = templates[t].letter;
    }
    auto up = [&](auto *&dptr, const void *src, size_t bytes) -> int {
        FOCR_HIP(c, hipMalloc((void **)&dptr, bytes ? bytes : 16));
        FOCR_HIP(c, hipMemcpy(dptr, src, bytes, hipMemcpyHostToDevice));
        return FOCR_OK;
    };
    int rc;
    if ((rc = up(c->d_tconst, c->h_tconst.data(), c->h_tconst.size() * sizeof(TemplateConst)))) return rc;
    if ((rc = up(c->d_direct_bank, direct.data(), direct.size() * 4))) return rc;
    if ((rc = up(c->d_needles, dense.data(), dense.size()))) return rc;
    if ((rc = up(c->d_needle_off, c->h_needle_off.data(), c->h_needle_off.size() * 4))) return rc;
    if ((rc = up(c->d_t_w, tw.data(), tw.size() * 4))) return rc;
    if ((rc = up(c->d_t_h, th.data(), th.size() * 4))) return rc;
    if ((rc = up(c->d_t_letter, tl.data(), tl.size() * 4))) return rc;
    return build_mfma_bank(c, dense.data());
}

}  // extern "C"

namespace focr {
// Host-only part of the bank upload: size classes, class order, per-template constants, dense / direct-kernel needles.
void bank_host_prepare(focr_ctx *c, const focr_template_t *templates, size_t n_templates, const uint8_t *needles,
                       std::vector<uint32_t> &direct, std::vector<uint8_t> &dense) {
    c->n_templates = n_templates;
    c->h_templates.assign(templates, templates + n_templates);

    // size classes in order of first appearance
    std::vector<std::vector<uint32_t>> members;
    for (size_t t = 0; t < n_templates; t++) {
        size_t k = 0;
        for (; k < c->classes.size(); k++)
            if (c->classes[k].n_w == templates[t].n_w && c->classes[k].n_h == templates[t].n_h) break;
        if (k == c->classes.size()) {
            SizeClass sc{};
            sc.n_w = templates[t].n_w;
            sc.n_h = templates[t].n_h;
            sc.ndw = (sc.n_w + 3) / 4;
            sc.tall = sc.n_h > 32 || sc.n_w > 16;
            sc.maxh = sc.tall ? sc.n_h : (sc.n_h <= 16 ? 16 : 32);
            c->classes.push_back(sc);
            members.emplace_back();
        }
        members[k].push_back((uint32_t)t);
    }

    // Inside a class, order templates by (letter, shift): the sub-pixel variants of one glyph fire on the
    // same windows, so they should share a 16-template MFMA N-tile (fewer prefilter slow-path visits).
    // Results are keyed by the global template index, so this order is invisible to callers.
    for (auto &mem : members)
        std::stable_sort(mem.begin(), mem.end(), [&](uint32_t a, uint32_t b) {
            if (templates[a].letter != templates[b].letter) return templates[a].letter < templates[b].letter;
            if (templates[a].shift_x != templates[b].shift_x) return templates[a].shift_x < templates[b].shift_x;
            return templates[a].shift_y < templates[b].shift_y;
        });

    uint32_t first = 0;
    for (size_t k = 0; k < c->classes.size(); k++) {
        SizeClass &sc = c->classes[k];
        sc.first = first;
        sc.n_templates = (uint32_t)members[k].size();
        first += sc.n_templates;
        c->direct_bank_off.push_back(direct.size());
        const uint32_t n = sc.n_w * sc.n_h;
        for (uint32_t t : members[k]) {
            const uint8_t *nd = needles + templates[t].offset;
            // reference kernel prologue, src/ncc.cpp:73-86 / 278-291 (host IEEE double)
            uint32_t s_n = 0, s2_n = 0;
            for (uint32_t i = 0; i < n; i++) {
                s_n += nd[i];
                s2_n += (uint32_t)nd[i] * nd[i];
            }
            double norm2_n = (double)s2_n - (double)((uint64_t)s_n * (uint64_t)s_n) / (double)n;
            TemplateConst tc{};
            tc.s_n = (double)s_n;
            tc.n_recip = 1. / (double)n;
            tc.rnorm_n = 1. / std::sqrt(norm2_n);
            tc.norm2_n = norm2_n;
            tc.index = t;
            tc.n_w = sc.n_w;
            tc.n_h = sc.n_h;
            c->h_tconst.push_back(tc);
            // direct-kernel layout: maxh rows of ndw dwords, zero padded
            for (uint32_t j = 0; j < sc.maxh; j++)
                for (uint32_t k4 = 0; k4 < sc.ndw; k4++) {
                    uint32_t w = 0;
                    for (uint32_t b = 0; b < 4; b++) {
                        uint32_t i = k4 * 4 + b;
                        if (j < sc.n_h && i < sc.n_w) w |= (uint32_t)nd[j * sc.n_w + i] << (8 * b);
                    }
                    direct.push_back(w);
                }
            c->h_needle_off.push_back((uint32_t)dense.size());
            dense.insert(dense.end(), nd, nd + n);
        }
    }
}
}  // namespace focr

extern "C" {

int focr_pages_alloc(focr_ctx_t *c, size_t n_pages, size_t r_w, size_t r_h) {
    if (!c || !n_pages || !r_w || !r_h) return fail(c, FOCR_ERR_INVALID, "focr_pages_alloc: bad arguments");
    if (r_w > 65535 || r_h > 65535)  // Match.x/y and start_end are u16, src/ncc.cpp:7-10, src/ncc.rs:313-314
        return fail(c, FOCR_ERR_INVALID, "focr_pages_alloc: page side above 65535 px");
    if (n_pages > 65535) return fail(c, FOCR_ERR_INVALID, "focr_pages_alloc: more than 65535 pages per batch");
    FOCR_HIP(c, hipSetDevice(c->device));
    if (c->d_pages && c->r_w == r_w && c->r_h == r_h && n_pages <= c->pages_capacity) {
        // same geometry, no more pages than before: keep the buffer (its zero padding is never written)
        c->scanned = c->processed = false;
        c->n_pages = n_pages;
        return FOCR_OK;
    }
    FOCR_HIP(c, hipStreamSynchronize(c->stream));
    free_dev(c->d_pages);
    free_dev(c->d_pages_i8);
    c->pages_capacity = 0;
    c->scanned = c->processed = false;
    c->n_pages = n_pages;
    c->r_w = r_w;
    c->r_h = r_h;
    c->pitch = (r_w + 64 + 63) / 64 * 64;  // >= 64 zero bytes right of every row
    c->rows_alloc = r_h + 48;              // >= 48 zero rows below every page
    size_t bytes = c->n_pages * c->rows_alloc * c->pitch;
    if (hipMalloc(&c->d_pages, bytes) != hipSuccess || hipMalloc(&c->d_pages_i8, bytes) != hipSuccess) {
        free_dev(c->d_pages);
        free_dev(c->d_pages_i8);
        c->n_pages = 0;
        return fail(c, FOCR_ERR_NOMEM, "focr_pages_alloc: hipMalloc failed");
    }
    c->pages_capacity = n_pages;
    FOCR_HIP(c, hipMemsetAsync(c->d_pages, 0, bytes, c->stream));
    FOCR_HIP(c, hipMemsetAsync(c->d_pages_i8, 0x80, bytes, c->stream));  // paper (0) as int8
    return FOCR_OK;
}

static int ingest(focr_ctx *c, const uint8_t *d_src, size_t first, size_t count, int invert) {
    const size_t n_rows = count * c->r_h;
    const unsigned blocks = (unsigned)std::min<size_t>(n_rows, (size_t)1 << 20);
    const int dwords = c->r_w % 4 == 0 && (reinterpret_cast<uintptr_t>(d_src) & 3) == 0;
    hipLaunchKernelGGL(ingest_pages, dim3(blocks), dim3(64), 0, c->stream, d_src, c->d_pages, c->d_pages_i8, (uint32_t)c->r_w, (uint32_t)c->r_h, c->pitch,
                       c->rows_alloc, first, n_rows, invert, dwords);
    FOCR_HIP(c, hipGetLastError());
    c->scanned = c->processed = false;
    c->sizes_pending = c->post_pending = false;  // results of the previous batch are gone with its pages
    return FOCR_OK;
}

}  // extern "C"

namespace focr {

// The executor's early ingest (pipe.hip): n_pages tight luma8 pages at d_luma (device memory) become the ALTERNATE page set of the
// context, on stream s — not the context's own: the context may be scanning d_pages meanwhile.  The caller orders s behind the
// arrival of d_luma and the context's stream behind s (an event) before pages_alt_swap makes the set current.  The alternate set
// is free whenever this is called: it was current two batches ago, and every batch of a lane ends with focr_sync.
int pages_alt_ingest(focr_ctx *c, const void *d_luma, size_t n_pages, size_t r_w, size_t r_h, int invert, hipStream_t s) {
    if (!c || !d_luma || !n_pages || !r_w || !r_h || r_w > 65535 || r_h > 65535 || n_pages > 65535)
        return fail(nullptr, FOCR_ERR_INVALID, "pages_alt_ingest: bad arguments");
    // (errors go to the process-wide message only: the context's own belongs to the lane's thread, which may be running a batch)
    focr_ctx::PageSet &a = c->alt;
    if (!a.u8 || a.r_w != r_w || a.r_h != r_h || a.capacity < n_pages) {
        free_dev(a.u8);  // (hipFree waits for the device: a change of geometry, not the steady state)
        free_dev(a.i8);
        a = focr_ctx::PageSet{};
        const size_t pitch = (r_w + 64 + 63) / 64 * 64, rows_alloc = r_h + 48;  // as focr_pages_alloc
        const size_t bytes = n_pages * rows_alloc * pitch;
        if (hipMalloc(&a.u8, bytes) != hipSuccess || hipMalloc(&a.i8, bytes) != hipSuccess) {
            free_dev(a.u8);
            free_dev(a.i8);
            return fail(nullptr, FOCR_ERR_NOMEM, "pages_alt_ingest: hipMalloc failed");
        }
        a.capacity = n_pages;
        a.r_w = r_w;
        a.r_h = r_h;
        a.pitch = pitch;
        a.rows_alloc = rows_alloc;
        FOCR_HIP((focr_ctx *)nullptr, hipMemsetAsync(a.u8, 0, bytes, s));
        FOCR_HIP((focr_ctx *)nullptr, hipMemsetAsync(a.i8, 0x80, bytes, s));  // paper (0) as int8
    }
    const size_t n_rows = n_pages * r_h;
    const unsigned blocks = (unsigned)std::min<size_t>(n_rows, (size_t)1 << 20);
    const int dwords = r_w % 4 == 0 && (reinterpret_cast<uintptr_t>(d_luma) & 3) == 0;
    hipLaunchKernelGGL(ingest_pages, dim3(blocks), dim3(64), 0, s, (const uint8_t *)d_luma, a.u8, a.i8, (uint32_t)r_w, (uint32_t)r_h, a.pitch, a.rows_alloc, (size_t)0,
                       n_rows, invert, dwords);
    FOCR_HIP((focr_ctx *)nullptr, hipGetLastError());
    return FOCR_OK;
}

// The alternate set becomes the context's pages (n_pages of r_w x r_h, as ingested by pages_alt_ingest), the previous pages the
// alternate set.  Host state only: the caller has ordered the context's stream behind the ingest.
int pages_alt_swap(focr_ctx *c, size_t n_pages, size_t r_w, size_t r_h) {
    focr_ctx::PageSet &a = c->alt;
    if (!a.u8 || a.r_w != r_w || a.r_h != r_h || a.capacity < n_pages) return fail(c, FOCR_ERR_STATE, "pages_alt_swap: no such alternate page set");
    focr_ctx::PageSet cur;
    cur.u8 = c->d_pages;
    cur.i8 = c->d_pages_i8;
    cur.capacity = c->pages_capacity;
    cur.r_w = c->r_w;
    cur.r_h = c->r_h;
    cur.pitch = c->pitch;
    cur.rows_alloc = c->rows_alloc;
    c->d_pages = a.u8;
    c->d_pages_i8 = a.i8;
    c->pages_capacity = a.capacity;
    c->r_w = a.r_w;
    c->r_h = a.r_h;
    c->pitch = a.pitch;
    c->rows_alloc = a.rows_alloc;
    c->n_pages = n_pages;
    a = cur;
    c->scanned = c->processed = false;
    c->sizes_pending = c->post_pending = false;  // results of the previous batch are gone with its pages
    return FOCR_OK;
}

}  // namespace focr

extern "C" {

int focr_pages_upload(focr_ctx_t *c, size_t first, size_t count, const uint8_t *luma, int invert) {
    if (!c || !luma) return fail(c, FOCR_ERR_INVALID, "focr_pages_upload: bad arguments");
    if (!c->d_pages) return fail(c, FOCR_ERR_STATE, "focr_pages_upload: call focr_pages_alloc first");
    if (first + count > c->n_pages) return fail(c, FOCR_ERR_INVALID, "focr_pages_upload: page range out of bounds");
    FOCR_HIP(c, hipSetDevice(c->device));
    const size_t page_bytes = c->r_w * c->r_h;
    const size_t chunk_pages = std::max<size_t>(1, (256u << 20) / page_bytes);
    size_t need = std::min(count, chunk_pages) * page_bytes;
    if (c->stage_bytes < need) {
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        free_dev(c->d_stage);
        FOCR_HIP(c, hipMalloc(&c->d_stage, need));
        c->stage_bytes = need;
    }
    for (size_t done = 0; done < count; done += chunk_pages) {
        size_t n = std::min(chunk_pages, count - done);
        FOCR_HIP(c, hipMemcpyAsync(c->d_stage, luma + done * page_bytes, n * page_bytes, hipMemcpyHostToDevice, c->stream));
        int rc = ingest(c, c->d_stage, first + done, n, invert);
        if (rc) return rc;
        if (done + chunk_pages < count) FOCR_HIP(c, hipStreamSynchronize(c->stream));  // staging buffer reuse
    }
    return FOCR_OK;
}

int focr_host_alloc(size_t bytes, void **out) {
    if (!out || !bytes) return fail(nullptr, FOCR_ERR_INVALID, "focr_host_alloc: bad arguments");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        *out = nullptr;
        return fail(nullptr, e == hipErrorOutOfMemory ? FOCR_ERR_NOMEM : FOCR_ERR_NO_DEVICE,
                    std::string("focr_host_alloc: ") + hipGetErrorString(e));
    }
    return FOCR_OK;
}

void focr_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int focr_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return fail(nullptr, FOCR_ERR_INVALID, "focr_host_register: bad arguments");
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) return fail(nullptr, FOCR_ERR_NO_DEVICE, std::string("focr_host_register: ") + hipGetErrorString(e));
    return FOCR_OK;
}

void focr_host_unregister(void *p) {
    if (p) (void)hipHostUnregister(p);
}

int focr_pages_upload_device(focr_ctx_t *c, size_t first, size_t count, const void *d_luma, int invert) {
    if (!c || !d_luma) return fail(c, FOCR_ERR_INVALID, "focr_pages_upload_device: bad arguments");
    if (!c->d_pages) return fail(c, FOCR_ERR_STATE, "focr_pages_upload_device: call focr_pages_alloc first");
    if (first + count > c->n_pages) return fail(c, FOCR_ERR_INVALID, "focr_pages_upload_device: page range out of bounds");
    FOCR_HIP(c, hipSetDevice(c->device));
    return ingest(c, (const uint8_t *)d_luma, first, count, invert);
}

}  // extern "C"

template <typename Run>
static int scan_split(focr_ctx *c, Run &run) {
    const size_t T = c->n_templates, n_seg_all = c->n_pages * T;
    size_t match_total = 0, hit_total = 0, raw_total = 0, cand_total = 0;
    float ms_acc[6] = {0, 0, 0, 0, 0, 0};
    uint64_t issued = 0;
    uint32_t *acc_cnt = (uint32_t *)c->acc_seg_count.ensure(c, (n_seg_all + 1) * 4);
    if (!acc_cnt) return fail(c, FOCR_ERR_NOMEM, "focr_scan: hipMalloc failed");
    size_t np = std::max<size_t>(1, c->n_pages / 2);
    for (size_t p0 = 0; p0 < c->n_pages;) {
        np = std::min(np, c->n_pages - p0);
        int rc = run(p0, np);
        if ((rc == FOCR_ERR_OVERFLOW || rc == FOCR_ERR_NOMEM) && np > 1) {
            np = (np + 1) / 2;  // still too much: halve and retry the same pages
            continue;
        }
        if (rc) return rc;
        // append: matches, per-call counts, kept hits
        const size_t nm = c->n_matches, nh = c->n_hits;
        focr_match_t *am = (focr_match_t *)c->acc_matches.ensure_keep(c, (match_total + nm + 1) * sizeof(focr_match_t), match_total * sizeof(focr_match_t));
        uint64_t *ak = (uint64_t *)c->acc_hkeys.ensure_keep(c, (hit_total + nm + 1) * 8, hit_total * 8);
        float *as = (float *)c->acc_hsims.ensure_keep(c, (hit_total + nm + 1) * 4, hit_total * 4);
        if (!am || !ak || !as) return fail(c, FOCR_ERR_NOMEM, "focr_scan: hipMalloc failed");
        if (nm) FOCR_HIP(c, hipMemcpyAsync(am + match_total, c->d_matches, nm * sizeof(focr_match_t), hipMemcpyDeviceToDevice, c->stream));
        FOCR_HIP(c, hipMemcpyAsync(acc_cnt + p0 * T, c->d_seg_count, np * T * 4, hipMemcpyDeviceToDevice, c->stream));
        if (nh) {
            uint64_t *f64 = (uint64_t *)c->scan_flags.ensure(c, (nh + 1) * 8), *pos = (uint64_t *)c->scan_pos.ensure(c, (nh + 1) * 8);
            if (!f64 || !pos) return fail(c, FOCR_ERR_NOMEM, "focr_scan: hipMalloc failed");
            const unsigned nb = (unsigned)((nh + 255) / 256);
            hipLaunchKernelGGL(widen_u8_to_u64, dim3(nb), dim3(256), 0, c->stream, (const uint8_t *)c->ord_keep.p, nh, f64);
            if ((rc = exclusive_scan_u64(c, f64, pos, nh))) return rc;
            hipLaunchKernelGGL(append_kept_hits, dim3(nb), dim3(256), 0, c->stream, c->d_hkeys, c->d_hsims, (const uint8_t *)c->ord_keep.p,
                               pos, nh, ak + hit_total, as + hit_total);
            FOCR_HIP(c, hipGetLastError());
        }
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        match_total += nm;
        hit_total += nm;  // kept hits == matches
        raw_total += c->n_hits_raw;
        cand_total += c->n_cand;
        issued += c->counters[3];
        for (int i = 0; i < 6; i++) ms_acc[i] += c->ms[i];
        p0 += np;
    }
    // install the accumulated results as the scan's results
    {
        uint64_t *count64 = c->d_seg_start + (n_seg_all + 1);  // seg arrays were sized for the whole batch by the sub-runs
        hipLaunchKernelGGL(widen_u32_to_u64, dim3((unsigned)((n_seg_all + 256) / 256)), dim3(256), 0, c->stream, acc_cnt, n_seg_all, count64);
        int rc = exclusive_scan_u64(c, count64, c->d_seg_offset, n_seg_all + 1);
        if (rc) return rc;
        FOCR_HIP(c, hipMemcpyAsync(c->d_seg_count, acc_cnt, n_seg_all * 4, hipMemcpyDeviceToDevice, c->stream));
        uint8_t *keep = (uint8_t *)c->ord_keep.ensure(c, hit_total + 1);
        if (!keep) return fail(c, FOCR_ERR_NOMEM, "focr_scan: hipMalloc failed");
        FOCR_HIP(c, hipMemsetAsync(keep, 1, hit_total + 1, c->stream));
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        std::swap(c->d_matches, *(focr_match_t **)&c->acc_matches.p);  // hand the accumulated list over (capacities swap too)
        size_t acc_cap = c->acc_matches.bytes / sizeof(focr_match_t);
        c->acc_matches.bytes = c->matches_alloc * sizeof(focr_match_t);
        c->matches_alloc = acc_cap;
        c->d_hkeys = (uint64_t *)c->acc_hkeys.p;
        c->d_hsims = (float *)c->acc_hsims.p;
        // the accumulated hit count as the device-side value process_hits reads
        c->n_hits_raw_u64 = hit_total;
        FOCR_HIP(c, hipMemcpyAsync(c->d_res + 7, &c->n_hits_raw_u64, 8, hipMemcpyHostToDevice, c->stream));
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        c->d_n_hits = c->d_res + 7;
        c->ub_hits = hit_total;
    }
    c->sub_p0 = 0;
    c->sub_np = c->n_pages;
    c->n_matches = match_total;
    c->n_hits = hit_total;
    c->n_hits_raw = raw_total;
    c->n_cand = cand_total;
    c->counters[0] = cand_total;
    c->counters[1] = raw_total;
    c->counters[3] = issued;
    for (int i = 0; i < 6; i++) c->ms[i] = ms_acc[i];
    return FOCR_OK;
}

namespace focr {

// The whole scan pipeline on the resident batch with the parameters stored in the context (focr_scan, and the redo of a
// batch whose estimated sizes turned out too small).
static int scan_now(focr_ctx *c) {
    const float threshold = c->scan_thr;
    const int mode = c->scan_mode;
    c->scanned = c->processed = false;
    c->sizes_pending = c->post_pending = false;
    for (auto &m : c->ms) m = 0.f;
    c->counters[3] = 0;
    auto run = [&](size_t p0, size_t np) -> int {  // the whole pipeline on pages [p0, p0 + np)
        c->sub_p0 = p0;
        c->sub_np = np;
        c->ordered = false;
        int r = mode == FOCR_SCAN_MFMA ? launch_scan_mfma(c, threshold) : launch_scan_direct(c, threshold, mode == FOCR_SCAN_RUST);
        if (r) return r;
        if (!c->ordered && (r = order_hits(c))) return r;
        c->sizes_pending = true;
        return c->estimated ? FOCR_OK : finish_results(c);  // exact sizes: the counts are read here, as they always were
    };
    // focr_debug_force_split (tests): take the split-batch path without waiting for an overflow
    int rc = c->force_split ? FOCR_ERR_OVERFLOW : run(0, c->n_pages);
    if (rc == FOCR_ERR_OVERFLOW || rc == FOCR_ERR_NOMEM) {
        // Too many candidates for one pass (very low thresholds): scan the batch in page sub-ranges and append the
        // results.  Only hits that survive the per-call cap are kept, so the totals stay bounded by pages x T x cap.
        c->estimated = false;
        c->sizes_pending = false;
        rc = scan_split(c, run);
        if (rc) return rc;
        // the sub-runs left the size estimates at the counts of the LAST page sub-range: a following scan of this setup must
        // not run "estimated" on them (it would overflow, redo exact, overflow again and only then split)
        c->est_cand = c->est_hits = 0;
        c->est_last_cand = c->est_last_hits = 0;
        c->est_row_max = 0;
    } else if (rc) {
        return rc;
    }
    c->scanned = true;
    return FOCR_OK;
}

void row_segments(const focr_ctx *c, uint32_t *seg_shift, uint32_t *n_seg);  // rows.hip

// The context joins a lane of an executor (pipe.hip): it works on the lane's stream from now on (its own, idle, is destroyed) and reads
// results back on the lane's side stream.
void ctx_share_stream(focr_ctx *c, hipStream_t lane_stream, hipStream_t io_stream) {
    (void)hipSetDevice(c->device);
    if (c->stream && c->owns_stream && c->stream != lane_stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
    }
    c->owns_stream = false;
    c->stream = lane_stream;
    c->io_stream = io_stream ? io_stream : lane_stream;
}

// Until the work this context has queued is done.  Inside an executor that is the context's OWN batch (the event the executor
// recorded behind its last kernel; consumed here) — the lane's stream already holds the next batch of another context; a context
// with a stream of its own waits for the stream.
int wait_batch(focr_ctx *c) {
    if (c->batch_event) {
        hipEvent_t e = c->batch_event;
        c->batch_event = nullptr;
        FOCR_HIP(c, hipEventSynchronize(e));
        return FOCR_OK;
    }
    FOCR_HIP(c, hipStreamSynchronize(c->stream));
    return FOCR_OK;
}

int finish_results(focr_ctx *c) {
    if (!c->sizes_pending && !c->post_pending) return FOCR_OK;
    FOCR_HIP(c, hipSetDevice(c->device));
    if (int rc = wait_batch(c)) return rc;
    if (c->sizes_pending) {
        c->sizes_pending = false;
        const uint64_t n_cand = c->h_res[0], n_hits = c->h_res[1], total = c->h_res[2];
        if (c->h_res[4] & 4) return fail(c, FOCR_ERR_STATE, "internal error: a candidate key outside the batch reached the verify stage");
        if (c->estimated && (c->h_res[4] & 3)) {  // bit 0: a count above its bound, bit 1: a page row above the row kernel's capacity
            // a count exceeded the bound taken from the previous scan: redo this batch with exact sizes (and its
            // process_hits, if that was queued behind it)
            const bool redo_post = c->post_pending;
            c->post_pending = false;
            c->estimated = false;
            c->est_cand = c->est_hits = 0;
            c->est_var = 0.0667;  // back to the 20 % margin
            c->est_last_cand = c->est_last_hits = 0;
            c->est_row_max = 0;
            c->counters_redone++;
            int rc = scan_now(c);
            if (rc) return rc;
            return redo_post ? focr_process_hits(c, c->post_anchor, c->post_overlap) : FOCR_OK;
        }
        if (c->scan_mode == FOCR_SCAN_MFMA) {
            c->n_cand = (size_t)n_cand;
            c->counters[0] = n_cand;
            FOCR_HIP(c, hipEventElapsedTime(&c->ms[0], c->ev[0], c->ev[1]));
            FOCR_HIP(c, hipEventElapsedTime(&c->ms[1], c->ev[1], c->ev[2]));
            FOCR_HIP(c, hipEventElapsedTime(&c->ms[2], c->ev[2], c->ev[3]));
            c->counters[3] = 0;
            for (focr_launch_info_t &li : c->launches) {  // issued MACs follow the number of live M-tiles (known only now)
                if (strncmp(li.name, "scan_mfma", 9) == 0 && (li.n_templates >> 24) < 40) {
                    li.issued_macs *= c->h_live[li.n_templates >> 24];
                    li.n_templates &= 0xffffff;
                }
                c->counters[3] += li.issued_macs;
            }
            c->launches_collect();
            // bounds for the next scan of the same setup: this scan's counts + a margin that follows how much the counts have
            // been moving (20 % after the first scan of a setup; 4 % once consecutive batches agree to ~1 %): every element of
            // margin is sorted, scanned and stepped over by all the later phases
            if (c->est_last_cand) {
                const auto rel = [](uint64_t a, uint64_t b) { return (double)(a > b ? a - b : b - a) / (double)std::max<uint64_t>(std::min(a, b), 1); };
                c->est_var = std::max(c->est_var * 0.75, std::max(rel(n_cand, c->est_last_cand), rel(n_hits, c->est_last_hits)));
            }
            c->est_last_cand = n_cand;
            c->est_last_hits = n_hits;
            const double margin = std::min(0.2, std::max(0.04, 3.0 * c->est_var));
            c->est_cand = (size_t)n_cand + (size_t)((double)n_cand * margin) + 8192;
            c->est_hits = (size_t)n_hits + (size_t)((double)n_hits * margin) + 8192;
            c->est_row_max = c->row_cap ? (uint32_t)std::max<uint64_t>(c->h_res[5], 1) : 0;  // 0: the last scan took the legacy tail
            {  // buckets still well above what a wave sorts in registers: halve the x-segments for the next scan of this setup
                uint32_t sh, ns;
                row_segments(c, &sh, &ns);
                c->row_seg_shift = sh;
                if (c->h_res[5] > 2048 && sh > 5) c->row_seg_shift = sh - 1;
            }
            if (c->est_sig) {  // for the other contexts that scan this setup (focr_scan): this batch's counts + the widest margin
                std::lock_guard<std::mutex> lk(g_est_mu);
                if (g_est.size() > 256) g_est.clear();
                g_est[c->est_sig] = SharedEstimate{(size_t)n_cand + (size_t)n_cand / 5 + 8192, (size_t)n_hits + (size_t)n_hits / 5 + 8192, c->est_row_max, c->row_seg_shift};
            }
        }
        c->counters[1] = n_hits;
        c->n_hits = c->n_hits_raw = (size_t)n_hits;
        c->n_matches = (size_t)total;
        FOCR_HIP(c, hipEventElapsedTime(&c->ms[3], c->ev[3], c->ev[4]));
        FOCR_HIP(c, hipEventElapsedTime(&c->ms[5], c->ev[0], c->ev[4]));
    }
    if (c->post_pending) {
        c->post_pending = false;
        const uint64_t tot = c->h_res[3];
        c->n_lines = (size_t)(tot >> 32);
        c->n_chars = (size_t)(tot & 0xffffffffu);
        FOCR_HIP(c, hipEventElapsedTime(&c->ms[4], c->ev[5], c->ev[6]));
    }
    return FOCR_OK;
}

}  // namespace focr

extern "C" {

int focr_scan(focr_ctx_t *c, float threshold, uint32_t cap, int mode) {
    if (!c) return FOCR_ERR_INVALID;
    if (!c->n_templates) return fail(c, FOCR_ERR_STATE, "focr_scan: no bank uploaded");
    if (!c->d_pages) return fail(c, FOCR_ERR_STATE, "focr_scan: no pages resident");
    if (cap == 0) return fail(c, FOCR_ERR_INVALID, "focr_scan: cap must be >= 1 (src/ncc.cpp:43-46)");
    if (mode != FOCR_SCAN_MFMA && mode != FOCR_SCAN_DIRECT && mode != FOCR_SCAN_RUST) return fail(c, FOCR_ERR_INVALID, "focr_scan: bad mode");
    if (std::isnan(threshold)) threshold = INFINITY;  // `sim > NaN` is never true in the reference (src/ncc.cpp:362-366): no hits
    FOCR_HIP(c, hipSetDevice(c->device));
    c->cap = cap;
    c->scan_thr = threshold;
    c->scan_mode = mode;
    // algorithmic MACs, SURVEY.md section 8(d): true template area x searched windows
    uint64_t macs = 0;
    for (const SizeClass &sc : c->classes) {
        if (sc.n_w > c->r_w || sc.n_h > c->r_h) continue;
        uint64_t wx = c->r_w - sc.n_w, wy = c->r_h - sc.n_h;  // x in [1, r_w-n_w], y in [1, r_h-n_h]
        macs += wx * wy * (uint64_t)sc.n_w * sc.n_h * sc.n_templates;
    }
    c->counters[2] = macs * c->n_pages;
    auto nbits = [](size_t n) {
        uint32_t b = 1;
        while (((size_t)1 << b) < n) b++;
        return b;
    };
    c->fmt = KeyFmt{nbits(c->n_templates), nbits(c->r_w), nbits(c->r_h), nbits(c->n_pages)};
    // Size estimates are reused only for the very same setup (bank, batch geometry, threshold, cap, prefilter)
    uint32_t tb;
    memcpy(&tb, &threshold, 4);
    uint64_t sig = 1469598103934665603ull;
    for (uint64_t v : {(uint64_t)c->bank_hash, (uint64_t)c->device, (uint64_t)c->tail_mode, (uint64_t)c->n_pages, (uint64_t)c->r_w, (uint64_t)c->r_h, (uint64_t)tb, (uint64_t)cap, (uint64_t)mode,
                       (uint64_t)c->prefilter})
        sig = (sig ^ v) * 1099511628211ull;
    if (sig != c->est_sig) {
        c->est_row_max = 0;
        c->row_seg_shift = 0;
        c->est_cand = c->est_hits = 0;
        c->est_var = 0.0667;
        c->est_last_cand = c->est_last_hits = 0;
    }
    c->est_sig = sig;
    if (c->est_cand == 0 && c->estimates_enabled && mode == FOCR_SCAN_MFMA) {
        // no estimate of its own yet: another context of this process may have scanned the same setup (an executor's contexts take
        // consecutive batches of one stream: only the stream's very first batch pays the exact-size scan with its host waits)
        std::lock_guard<std::mutex> lk(g_est_mu);
        auto it = g_est.find(sig);
        if (it != g_est.end()) {
            c->est_cand = it->second.cand;
            c->est_hits = it->second.hits;
            c->est_row_max = it->second.row_max;
            c->row_seg_shift = it->second.seg_shift;
            c->est_var = 0.0667;  // a neighbour's batch, not this context's: the widest margin was applied when it was published
        }
    }
    c->estimated = c->estimates_enabled && mode == FOCR_SCAN_MFMA && !c->force_split && c->est_cand != 0;
    return scan_now(c);
}

int focr_size_estimate_stats(focr_ctx_t *c, uint64_t *redone, double *margin, uint32_t *row_max) {
    if (!c) return FOCR_ERR_INVALID;
    if (int rc = finish_results(c)) return rc;
    if (redone) *redone = c->counters_redone;
    if (margin) *margin = std::min(0.2, std::max(0.04, 3.0 * c->est_var));
    if (row_max) *row_max = c->est_row_max;
    return FOCR_OK;
}

int focr_ctx_set_size_estimates(focr_ctx_t *c, int on) {
    if (!c) return FOCR_ERR_INVALID;
    c->estimates_enabled = on != 0;
    return FOCR_OK;
}

int focr_get_counts(focr_ctx_t *c, uint32_t *counts) {
    if (!c || !counts) return fail(c, FOCR_ERR_INVALID, "focr_get_counts: bad arguments");
    if (!c->scanned) return fail(c, FOCR_ERR_STATE, "focr_get_counts: no scan results");
    if (int rc = finish_results(c)) return rc;
    FOCR_HIP(c, hipSetDevice(c->device));
    // (finished results are read back on io_stream: inside an executor the context's own stream already holds the lane's next batch)
    FOCR_HIP(c, hipMemcpyAsync(counts, c->d_seg_count, c->n_pages * c->n_templates * 4, hipMemcpyDeviceToHost, c->io_stream));
    FOCR_HIP(c, hipStreamSynchronize(c->io_stream));
    return FOCR_OK;
}

size_t focr_total_matches(focr_ctx_t *c) { return (c && c->scanned && finish_results(c) == FOCR_OK) ? c->n_matches : 0; }

int focr_get_matches(focr_ctx_t *c, uint64_t *offsets, focr_match_t *matches) {
    if (!c) return FOCR_ERR_INVALID;
    if (!c->scanned) return fail(c, FOCR_ERR_STATE, "focr_get_matches: no scan results");
    if (int rc = finish_results(c)) return rc;
    FOCR_HIP(c, hipSetDevice(c->device));
    if (offsets)
        FOCR_HIP(c, hipMemcpyAsync(offsets, c->d_seg_offset, (c->n_pages * c->n_templates + 1) * 8, hipMemcpyDeviceToHost,
                                   c->io_stream));
    if (matches && c->n_matches)
        FOCR_HIP(c, hipMemcpyAsync(matches, c->d_matches, c->n_matches * sizeof(focr_match_t), hipMemcpyDeviceToHost,
                                   c->io_stream));
    FOCR_HIP(c, hipStreamSynchronize(c->io_stream));
    return FOCR_OK;
}

int focr_last_timings(focr_ctx_t *c, float ms[6]) {
    if (!c || !ms) return FOCR_ERR_INVALID;
    if (int rc = finish_results(c)) return rc;
    for (int i = 0; i < 6; i++) ms[i] = c->ms[i];
    return FOCR_OK;
}

int focr_last_counters(focr_ctx_t *c, uint64_t out[4]) {
    if (!c || !out) return FOCR_ERR_INVALID;
    if (int rc = finish_results(c)) return rc;
    for (int i = 0; i < 4; i++) out[i] = c->counters[i];
    return FOCR_OK;
}

int focr_debug_rnorm(focr_ctx_t *c, const uint32_t *s, const uint64_t *s2, const uint32_t *n, size_t n_items, double *out) {
    if (!c || !s || !s2 || !n || !out) return fail(c, FOCR_ERR_INVALID, "focr_debug_rnorm: bad arguments");
    FOCR_HIP(c, hipSetDevice(c->device));
    uint32_t *ds = nullptr, *dn = nullptr;
    uint64_t *ds2 = nullptr;
    double *dout = nullptr;
    auto run = [&]() -> int {
        FOCR_HIP(c, hipMalloc(&ds, n_items * 4));
        FOCR_HIP(c, hipMalloc(&dn, n_items * 4));
        FOCR_HIP(c, hipMalloc(&ds2, n_items * 8));
        FOCR_HIP(c, hipMalloc(&dout, n_items * 8));
        FOCR_HIP(c, hipMemcpyAsync(ds, s, n_items * 4, hipMemcpyHostToDevice, c->stream));
        FOCR_HIP(c, hipMemcpyAsync(dn, n, n_items * 4, hipMemcpyHostToDevice, c->stream));
        FOCR_HIP(c, hipMemcpyAsync(ds2, s2, n_items * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(debug_rnorm_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, c->stream, ds, ds2, dn, n_items,
                           dout);
        FOCR_HIP(c, hipGetLastError());
        FOCR_HIP(c, hipMemcpyAsync(out, dout, n_items * 8, hipMemcpyDeviceToHost, c->stream));
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        return FOCR_OK;
    };
    const int rc = run();
    for (void *p : {(void *)ds, (void *)dn, (void *)ds2, (void *)dout})
        if (p) (void)hipFree(p);
    return rc;
}

}  // extern "C"
