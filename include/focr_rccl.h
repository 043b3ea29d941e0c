/*
 * focr_rccl.h — the one collective of the path: the gather of the post-processed match lists of
 * page shards that live on different GPUs of one node, over RCCL / xGMI, for a single-process host that drives several
 * devices (the layout of the reference's main(): one process, all cores — src/ncc.rs:839-847).  Plain C ABI.
 *
 * Separate library (libfocr_rccl.so) so that the scan library does not pull librccl into processes that never gather
 * (the `ncc` binary prints on the host: its results converge there without a collective).  bench.py's one-process-per-GPU
 * form uses torch.distributed's "nccl" backend (= RCCL) for the same exchange (font_ocr_amd/shard.py).
 */
#ifndef FOCR_RCCL_H
#define FOCR_RCCL_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct focr_gather focr_gather_t;

/* One RCCL communicator over the given devices of this node (ncclCommInitAll), one stream per device.  Rank i = devices[i];
 * rank 0 is the root every gather delivers to.  0 on success; focr_gather_last_error() explains a failure. */
int focr_gather_create(const int *devices, int n_devices, focr_gather_t **out);
void focr_gather_destroy(focr_gather_t *g);
const char *focr_gather_last_error(void);

/* Variable-length gather: d_src[i] (on devices[i], bytes[i] bytes — e.g. focr_lines_device_chars / the chars_out buffer of
 * focr_pipe_submit with focr_total_chars * sizeof(focr_hit_t)) -> one contiguous buffer d_dst on devices[0], in rank order
 * (= page order when ranks hold contiguous page blocks).  The sizes are known to the single host process, so no size
 * exchange is needed: one grouped ncclSend / ncclRecv per rank.  Blocks until the data is in d_dst.  dst_bytes must be >=
 * the sum of bytes[].  The gather runs on its own streams: the sources must be complete when it is called (they are once
 * the getter that returned the size — focr_total_chars, focr_pipe_wait — has returned). */
int focr_gather_bytes(focr_gather_t *g, const void *const *d_src, const size_t *bytes, void *d_dst, size_t dst_bytes);

#ifdef __cplusplus
}
#endif
#endif /* FOCR_RCCL_H */
