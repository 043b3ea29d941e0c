// mfma_common.h — declarations shared by the MFMA prefilter kernels (scan_mfma.hip, scan_mfma2.hip).
#pragma once
#include "common.h"

namespace focr {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int32_t REJECT = 0x3fffffff;  // |negL| of a window the reference never emits
constexpr uint32_t WBUF = 64;           // wave-private candidate staging entries in LDS (no atomics on the way in)

// one global atomic per flush: lane 0 reserves `count` slots, the wave copies its staged keys out coalesced
__device__ __forceinline__ void flush_wave_candidates(uint64_t *wbuf, uint32_t count, int lane, uint64_t *__restrict__ cand,
                                                      unsigned long long *__restrict__ cand_counter, unsigned long long cand_cap) {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(cand_counter, (unsigned long long)count);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    base = ((unsigned long long)hi << 32) | lo;
    if ((uint32_t)lane < count && base + lane < cand_cap) cand[base + lane] = wbuf[lane];
}

struct MfmaLaunch {
    const SizeClass *sc;
    uint32_t chunk_first, chunk_n;  // templates of the class covered by this launch (multiple of 16 except the last)
    const int32_t *negL;
    uint32_t Lpitch, Lrows;
};

// scan_mfma2.hip: barrier-free variant (one wave = one independent work item stream)
size_t mfma2_bank_budget();
int dispatch_mfma_v2(focr_ctx *c, const MfmaLaunch &L, unsigned n_cus);

}  // namespace focr
