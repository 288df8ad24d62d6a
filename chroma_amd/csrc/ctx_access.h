// ctx_access.h -- what the other translation units of libchroma_hip.so may ask of a chroma_ctx (defined in
// chroma_hip.hip): its device, its stream, and the library's thread-local error message.  Internal: not part of
// include/chroma_hip.h.
#pragma once
#include <hip/hip_runtime.h>
struct chroma_ctx;
extern "C" {
hipStream_t chroma_internal_stream(chroma_ctx *ctx);
int chroma_internal_device(chroma_ctx *ctx);
int chroma_internal_set_error(int code, const char *fmt, ...);      // returns `code`
// hipMalloc for the library's own buffers and scratch: when the device is out of memory, the blocks parked in the pool
// behind chroma_malloc / chroma_free are given back and the allocation is tried once more (free with hipFree)
hipError_t chroma_internal_malloc(chroma_ctx *ctx, void **ptr, size_t bytes);
// copies of any size between host memory (pageable is fine) and the device, through the context's pinned staging ring;
// both return when the data has arrived
int chroma_internal_dtoh(chroma_ctx *ctx, void *h_dst, const void *d_src, size_t nbytes);
int chroma_internal_htod(chroma_ctx *ctx, void *d_dst, const void *h_src, size_t nbytes);
// (bvh_device.hip) d_order[n] = the photons 0..n-1 ordered by a 16-bit cell of their direction; queued on the context's stream
int chroma_internal_direction_order(chroma_ctx *ctx, const float *d_dir, uint32_t n, uint32_t *d_order);
}
