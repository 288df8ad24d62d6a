// ref_linalg_driver.hip -- TEST INFRASTRUCTURE.  The reference's OWN unit-test kernels for its float3 algebra and its
// rotation -- test/linalg_test.cu (over chroma/cuda/linalg.h) and test/rotate_test.cu (over chroma/cuda/rotate.h) --
// compiled for gfx950 from the sources where they lie (found through -I, see oracle/Makefile; nothing is copied), and a
// host driver that launches them the way test/linalg_test.py and test/rotate_test.py do.  tests/test_gpu_ref_headers.py
// restates those two reference tests on this GPU (the kernels against NumPy, as the reference asserts) and holds the
// engine's own float3 algebra (csrc/device_common.h, through chroma_probe) to the same kernels bit for bit.
// (test/matrix_test.cu is not driven: the engine has no matrix type -- its intersect_triangle and rotate are written out
//  and pinned on the reference's mesh.h / rotate.h as compiled here.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "linalg_test.cu"     // the reference's test kernels, by path, unmodified
#include "rotate_test.cu"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "ref_linalg: %s failed: %s\n", #x, hipGetErrorString(e_)); return (int)e_; } } while (0)

// op: the kernel of linalg_test.cu, in the order of test/linalg_test.py:18-37.  a, b: [n][3]; out: [n][3], or [n] for dot / norm.
// The "equal" kernels work in place on a copy of `a`, which is what comes back.  n must be a multiple of 256 (the test's block).
extern "C" int ref_linalg_run(int op, int n, const float *a, const float *b, float c, float *out)
{
    if (n <= 0 || n % 256) return -1;
    const bool scalar_out = op == 16 || op == 18;
    float3 *d_a = nullptr, *d_b = nullptr;
    void *d_out = nullptr;
    CK(hipMalloc(&d_a, (size_t)n * sizeof(float3)));
    CK(hipMalloc(&d_b, (size_t)n * sizeof(float3)));
    CK(hipMalloc(&d_out, (size_t)n * sizeof(float3)));
    CK(hipMemcpy(d_a, a, (size_t)n * sizeof(float3), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, b, (size_t)n * sizeof(float3), hipMemcpyHostToDevice));
    const dim3 grid(n / 256), block(256);
    float3 *o3 = (float3 *)d_out;
    float *o1 = (float *)d_out;
    bool in_place = false;
    switch (op) {
    case 0: hipLaunchKernelGGL(float3add, grid, block, 0, 0, d_a, d_b, o3); break;
    case 1: hipLaunchKernelGGL(float3addequal, grid, block, 0, 0, d_a, d_b); in_place = true; break;
    case 2: hipLaunchKernelGGL(float3sub, grid, block, 0, 0, d_a, d_b, o3); break;
    case 3: hipLaunchKernelGGL(float3subequal, grid, block, 0, 0, d_a, d_b); in_place = true; break;
    case 4: hipLaunchKernelGGL(float3addfloat, grid, block, 0, 0, d_a, c, o3); break;
    case 5: hipLaunchKernelGGL(float3addfloatequal, grid, block, 0, 0, d_a, c); in_place = true; break;
    case 6: hipLaunchKernelGGL(floataddfloat3, grid, block, 0, 0, d_a, c, o3); break;
    case 7: hipLaunchKernelGGL(float3subfloat, grid, block, 0, 0, d_a, c, o3); break;
    case 8: hipLaunchKernelGGL(float3subfloatequal, grid, block, 0, 0, d_a, c); in_place = true; break;
    case 9: hipLaunchKernelGGL(floatsubfloat3, grid, block, 0, 0, d_a, c, o3); break;
    case 10: hipLaunchKernelGGL(float3mulfloat, grid, block, 0, 0, d_a, c, o3); break;
    case 11: hipLaunchKernelGGL(float3mulfloatequal, grid, block, 0, 0, d_a, c); in_place = true; break;
    case 12: hipLaunchKernelGGL(floatmulfloat3, grid, block, 0, 0, d_a, c, o3); break;
    case 13: hipLaunchKernelGGL(float3divfloat, grid, block, 0, 0, d_a, c, o3); break;
    case 14: hipLaunchKernelGGL(float3divfloatequal, grid, block, 0, 0, d_a, c); in_place = true; break;
    case 15: hipLaunchKernelGGL(floatdivfloat3, grid, block, 0, 0, d_a, c, o3); break;
    case 16: hipLaunchKernelGGL(dot, grid, block, 0, 0, d_a, d_b, o1); break;
    case 17: hipLaunchKernelGGL(cross, grid, block, 0, 0, d_a, d_b, o3); break;
    case 18: hipLaunchKernelGGL(norm, grid, block, 0, 0, d_a, o1); break;
    case 19: hipLaunchKernelGGL(minusfloat3, grid, block, 0, 0, d_a, o3); break;
    default: hipFree(d_a); hipFree(d_b); hipFree(d_out); return -1;
    }
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, in_place ? (void *)d_a : d_out, (size_t)n * (scalar_out ? sizeof(float) : sizeof(float3)), hipMemcpyDeviceToHost));
    hipFree(d_a); hipFree(d_b); hipFree(d_out);
    return 0;
}

// the kernel of rotate_test.cu as test/rotate_test.py launches it: a [n][3], phi [n], one axis; out [n][3]
extern "C" int ref_rotate_test_run(int n, const float *a, const float *phi, const float axis[3], float *out)
{
    if (n <= 0 || n % 256) return -1;
    float3 *d_a = nullptr, *d_out = nullptr;
    float *d_phi = nullptr;
    CK(hipMalloc(&d_a, (size_t)n * sizeof(float3)));
    CK(hipMalloc(&d_out, (size_t)n * sizeof(float3)));
    CK(hipMalloc(&d_phi, (size_t)n * sizeof(float)));
    CK(hipMemcpy(d_a, a, (size_t)n * sizeof(float3), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_phi, phi, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rotate, dim3(n / 256), dim3(256), 0, 0, d_a, d_phi, make_float3(axis[0], axis[1], axis[2]), d_out);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, d_out, (size_t)n * sizeof(float3), hipMemcpyDeviceToHost));
    hipFree(d_a); hipFree(d_out); hipFree(d_phi);
    return 0;
}
