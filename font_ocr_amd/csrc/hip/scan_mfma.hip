// scan_mfma.hip — i8 MFMA conservative prefilter + exact verify (under construction).
#include "common.h"

namespace focr {

int build_mfma_bank(focr_ctx *c, const uint8_t *needles) { return FOCR_OK; }

int launch_scan_mfma(focr_ctx *c, float threshold) {
    return fail(c, FOCR_ERR_INVALID, "FOCR_SCAN_MFMA is not built yet in this revision; use FOCR_SCAN_DIRECT");
}

}  // namespace focr
