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
}
