// prefilter_model.hip — host model of the MFMA prefilter's bound (focr_debug_prefilter, include/focr_ncc.h).
//
// No device call: the quantised bank is built by the very function focr_bank_upload uses (quantise_bank), the threshold of a
// window by the very inline functions the statistics and scan kernels use (mfma_common.h: dropped_column_W, threshold_f32,
// plane_value, prefilter_cin).  The CPU tests check the property the whole fast path rests on — the reference emits
// (sim > thr)  =>  the prefilter flags the pair (G + C-in > 0) — on text, noise, degenerate and adversarial windows, for
// positive and negative thresholds, with and without the column drop (tests/test_prefilter_host.py).
#include <cmath>
#include <cstring>

#include "mfma_common.h"

using namespace focr;

extern "C" int focr_debug_prefilter(const focr_template_t *templates, size_t n_templates, const uint8_t *needles, size_t needles_len, int column_drop,
                                    const uint8_t *windows, size_t n_windows, uint32_t frame_w, uint32_t frame_h, float threshold, double *sim,
                                    int64_t *d, double *info, size_t n_info) {
    if (!templates || !n_templates || !needles) return FOCR_ERR_INVALID;
    for (size_t t = 0; t < n_templates; t++)
        if (templates[t].n_w == 0 || templates[t].n_h == 0 || templates[t].n_w > 16 || templates[t].n_h > 32 ||
            (size_t)templates[t].offset + (size_t)templates[t].n_w * templates[t].n_h > needles_len ||
            (windows && (templates[t].n_w > frame_w || templates[t].n_h > frame_h)))
            return FOCR_ERR_INVALID;
    focr_ctx ctx;  // host state only
    focr_ctx *c = &ctx;
    c->column_drop = column_drop != 0;
    std::vector<uint32_t> direct, tglobal, order_of;
    std::vector<uint8_t> dense;
    std::vector<int8_t> qbank;
    bank_host_prepare(c, templates, n_templates, needles, direct, dense);
    if (int rc = quantise_bank(c, dense.data(), qbank, tglobal, order_of)) return rc;
    for (size_t k = 0; k < c->classes.size() && info && 4 * k + 3 < n_info; k++) {
        info[4 * k] = c->mfma_c_scale[k];
        info[4 * k + 1] = c->mfma_e_max[k];
        info[4 * k + 2] = c->mfma_rho_max[k];
        info[4 * k + 3] = c->classes[k].keep_w;
    }
    if (!windows || !n_windows || !sim || !d) return FOCR_OK;
    const double thr_d = (double)threshold;
    // int8 templates back out of the per-lane operand image
    std::vector<std::vector<int>> bq(c->h_tconst.size());
    for (size_t k = 0; k < c->classes.size(); k++) {
        const SizeClass &sc = c->classes[k];
        const uint32_t ksteps = sc.k_groups / 4;
        for (uint32_t i = 0; i < sc.n_templates; i++) {
            std::vector<int> &q = bq[sc.first + i];
            const uint32_t slot = c->mfma_slot[sc.first + i];
            q.assign((size_t)sc.keep_w * sc.n_h, 0);
            for (uint32_t j = 0; j < sc.n_h; j++)
                for (uint32_t x = 0; x < sc.keep_w; x++) {
                    uint32_t ks, g, byte;
                    kgroup_of(sc.layout, j, x, &ks, &g, &byte);
                    q[j * sc.keep_w + x] = qbank[sc.q_offset + ((size_t)((slot / 16) * ksteps + ks) * 64 + g * 16 + slot % 16) * 16 + byte];
                }
        }
    }
    for (size_t wi = 0; wi < n_windows; wi++) {
        const uint8_t *a = windows + wi * (size_t)frame_w * frame_h;
        for (size_t k = 0; k < c->classes.size(); k++) {
            const SizeClass &sc = c->classes[k];
            const PlaneParams p = plane_params(c, k, thr_d);
            const uint32_t n = sc.n_w * sc.n_h, kw = sc.keep_w, n_k = kw * sc.n_h;
            uint32_t s = 0, s2 = 0, q1 = 0, q2 = 0;
            for (uint32_t j = 0; j < sc.n_h; j++)
                for (uint32_t x = 0; x < sc.n_w; x++) {
                    const uint32_t v = a[j * frame_w + x];
                    s += v, s2 += v * v;
                    if (x >= kw) q1 += v, q2 += v * v;
                }
            const uint64_t V = (uint64_t)n * s2 - (uint64_t)s * s;
            const float Wf = kw != sc.n_w ? dropped_column_W_upper(n_k, n - n_k, s - q1, q1, q2) : 0.f;
            const float Lf = threshold_f32(p, (float)V, Wf);
            const int16_t plane = V != 0 ? plane_value(p, Lf) : PLANE_NEVER;
            const int cin = prefilter_cin(p.shift, plane);
            const double norm_p = std::sqrt((double)V / (double)n);
            for (uint32_t i = 0; i < sc.n_templates; i++) {
                const TemplateConst &tc = c->h_tconst[sc.first + i];
                const size_t o = wi * n_templates + tc.index;
                sim[o] = NAN;
                d[o] = INT64_MIN;  // dead templates (constant needles) never reach the candidate list
                if (tglobal[sc.tg_offset + c->mfma_slot[sc.first + i]] == 0xffffffffu) continue;
                long G = 0;
                for (uint32_t j = 0; j < sc.n_h; j++)
                    for (uint32_t x = 0; x < kw; x++) G += (long)((int)a[j * frame_w + x] - 128) * bq[sc.first + i][j * kw + x];
                d[o] = (int64_t)G + cin;
                if (V != 0 && std::isfinite(tc.rnorm_n)) {
                    const uint8_t *nd = dense.data() + c->h_needle_off[sc.first + i];
                    double num = 0;
                    for (uint32_t j = 0; j < sc.n_h; j++)
                        for (uint32_t x = 0; x < sc.n_w; x++) num += (double)a[j * frame_w + x] * nd[j * sc.n_w + x];
                    num -= tc.s_n * (double)s * tc.n_recip;
                    sim[o] = num * tc.rnorm_n / norm_p;
                }
            }
        }
    }
    return FOCR_OK;
}

// the plane value of a threshold L for a unit 2^shift (mfma_common.h: plane_value), host flavour (the device's is checked against it
// on the GPU: test_gpu_parity.py)
extern "C" void focr_debug_plane_value(const float *L, size_t n, uint32_t shift, int16_t *out) {
    PlaneParams p{};
    p.shift = shift;
    p.S = std::ldexp(1.0f, (int)shift);
    p.inv_S = std::ldexp(1.0f, -(int)shift);
    for (size_t i = 0; i < n; i++) out[i] = plane_value(p, L[i]);
}

// ... and the device flavour, for the GPU test that compares the two bit for bit
__global__ void plane_value_kernel(const float *__restrict__ x, size_t n, PlaneParams p, int16_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = plane_value(p, x[i]);
}
extern "C" int focr_debug_plane_value_device(focr_ctx_t *c, const float *x, size_t n, uint32_t shift, int16_t *out) {
    if (!c || !x || !out || !n) return fail(c, FOCR_ERR_INVALID, "focr_debug_plane_value_device: bad arguments");
    FOCR_HIP(c, hipSetDevice(c->device));
    PlaneParams p{};
    p.shift = shift;
    p.S = std::ldexp(1.0f, (int)shift);
    p.inv_S = std::ldexp(1.0f, -(int)shift);
    float *dx = nullptr;
    int16_t *dout = nullptr;
    auto run = [&]() -> int {
        FOCR_HIP(c, hipMalloc((void **)&dx, n * 4));
        FOCR_HIP(c, hipMalloc((void **)&dout, n * 2));
        FOCR_HIP(c, hipMemcpyAsync(dx, x, n * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(plane_value_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, dx, n, p, dout);
        FOCR_HIP(c, hipGetLastError());
        FOCR_HIP(c, hipMemcpyAsync(out, dout, n * 2, hipMemcpyDeviceToHost, c->stream));
        FOCR_HIP(c, hipStreamSynchronize(c->stream));
        return FOCR_OK;
    };
    const int rc = run();
    if (dx) (void)hipFree(dx);
    if (dout) (void)hipFree(dout);
    return rc;
}
