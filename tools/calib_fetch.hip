// calib_fetch.hip -- calibrates rocprofv3's FETCH_SIZE for this engine's access pattern:
// every lane reads ONE uint4 (16 B) at a pseudo-random index of a buffer far larger than L2 and the
// Infinity Cache, so (almost) every read misses; also a coalesced uint4 stream for comparison.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o /tmp/calib && rocprofv3 --pmc FETCH_SIZE ... -- /tmp/calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void k_gather16(const uint4 *buf, uint64_t nelem, uint64_t nreads, uint4 *out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nreads) return;
    uint64_t h = i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    uint4 v = buf[h % nelem];
    if (v.x == 0xdeadbeef) out[0] = v;      // never true: keeps the load
}
__global__ void k_stream16(const uint4 *buf, uint64_t nreads, uint4 *out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nreads) return;
    uint4 v = buf[i];
    if (v.x == 0xdeadbeef) out[0] = v;
}
int main()
{
    const uint64_t nelem = (8ull << 30) / 16;      // 8 GiB buffer
    const uint64_t nreads = 1ull << 28;             // 268 M reads = 4 GiB of 16-B requests
    uint4 *buf, *out;
    hipMalloc(&buf, nelem * 16); hipMalloc(&out, 64);
    hipMemset(buf, 1, nelem * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_gather16, dim3((unsigned)(nreads / 256)), dim3(256), 0, 0, buf, nelem, nreads, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("gather16: %llu reads of 16 B in %.2f ms (%.1f G reads/s)\n", (unsigned long long)nreads, ms, nreads / ms / 1e6);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_stream16, dim3((unsigned)(nreads / 256)), dim3(256), 0, 0, buf, nreads, out);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("stream16: %llu B in %.2f ms (%.1f GB/s)\n", (unsigned long long)(nreads * 16), ms, nreads * 16 / ms / 1e6);
    }
    return 0;
}
