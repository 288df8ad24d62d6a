// ref_headers_driver.hip -- TEST INFRASTRUCTURE.  A host driver around three more of the reference's
// OWN device headers, compiled for gfx950 from the sources where they lie (found through -I, see
// oracle/Makefile; nothing is copied): chroma/cuda/rotate.h (rotate, :22-28), interpolate.h
// (interp_idx :5-29, interp :32-57) and geometry.h (interp_property<T>, :64-75, instantiated on the
// reference's own Material struct of geometry_types.h).  Each kernel below only loads arguments, calls
// the reference function and stores what it returned.
//
// Still unbuildable here (no stand-ins are written): photon.h / random.h (curand_kernel.h), cx.h
// (cuComplex.h), daq.cu (curand), bvh.cu (cuda.h) -- see DESIGN.md section 4.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "rotate.h"
#include "interpolate.h"
#include "geometry.h"
#include "transform.cu"       // the reference's three point-transform kernels (chroma/cuda/transform.cu:9-49), by path, unmodified

extern "C" __global__ void ref_k_interp_property(int n, const float *x, const float *fp, int wavelength_n,
                                                 float wavelength_start, float wavelength_step, float *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Material m;
    memset(&m, 0, sizeof m);
    m.wavelength_n = wavelength_n;
    m.wavelength_start = wavelength_start;
    m.wavelength_step = wavelength_step;
    out[i] = interp_property(&m, x[i], fp);
}

extern "C" __global__ void ref_k_interp_idx(int n, const float *x, int ntab, float *xp, float *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = interp_idx(x[i], ntab, xp);
}

extern "C" __global__ void ref_k_interp(int n, const float *x, int ntab, float *xp, float *fp, float *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = interp(x[i], ntab, xp, fp);
}

// in: 7 floats per element (a.xyz, phi, axis.xyz); out: 5 floats per element: rotate(a, phi, axis).xyz
// and the device library's cosf(phi), sinf(phi) -- the two values rotate() forms first, so that a test
// can tell "different cosine" from "different algebra"
extern "C" __global__ void ref_k_rotate(int n, const float *in, float *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = in + 7 * (size_t)i;
    float3 a = make_float3(p[0], p[1], p[2]), axis = make_float3(p[4], p[5], p[6]);
    float3 r = rotate(a, p[3], axis);
    float *o = out + 5 * (size_t)i;
    o[0] = r.x; o[1] = r.y; o[2] = r.z;
    o[3] = cosf(p[3]); o[4] = sinf(p[3]);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "ref_headers: %s failed: %s\n", #x, hipGetErrorString(e_)); return (int)e_; } } while (0)

// fn 0: interp_property   x[n], tab_f[ntab (+1 pad: the reference reads fp[jl+1] with zero weight at the top edge)]
// fn 1: interp_idx        x[n], tab_x[ntab]
// fn 2: interp            x[n], tab_x[ntab], tab_f[ntab]
// fn 3: rotate            x[7 n] -> out[5 n]
// fn 4..6: the kernels of transform.cu on n points x[3 n] -> out[3 n]; tab_x = v[3] (translate), {phi, axis[3]} (rotate),
//          {phi, axis[3], point[3]} (rotate_around_point)
extern "C" int ref_headers_run(int fn, int n, const float *x, const float *tab_x, const float *tab_f, int ntab,
                               float start, float step, float *out)
{
    if (fn >= 4 && fn <= 6) {
        if (!tab_x) return -1;
        float3 *d_a = nullptr;
        CK(hipMalloc(&d_a, (size_t)n * sizeof(float3)));
        CK(hipMemcpy(d_a, x, (size_t)n * sizeof(float3), hipMemcpyHostToDevice));
        const int block = 256, grid = (n + block - 1) / block;
        if (fn == 4) hipLaunchKernelGGL(translate, dim3(grid), dim3(block), 0, 0, n, d_a, make_float3(tab_x[0], tab_x[1], tab_x[2]));
        else if (fn == 5) hipLaunchKernelGGL(rotate, dim3(grid), dim3(block), 0, 0, n, d_a, tab_x[0], make_float3(tab_x[1], tab_x[2], tab_x[3]));
        else hipLaunchKernelGGL(rotate_around_point, dim3(grid), dim3(block), 0, 0, n, d_a, tab_x[0], make_float3(tab_x[1], tab_x[2], tab_x[3]),
                                make_float3(tab_x[4], tab_x[5], tab_x[6]));
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out, d_a, (size_t)n * sizeof(float3), hipMemcpyDeviceToHost));
        hipFree(d_a);
        return 0;
    }
    const size_t nin = (size_t)n * (fn == 3 ? 7 : 1), nout = (size_t)n * (fn == 3 ? 5 : 1);
    float *d_x = nullptr, *d_tx = nullptr, *d_tf = nullptr, *d_out = nullptr;
    CK(hipMalloc(&d_x, nin * 4));
    CK(hipMalloc(&d_out, nout * 4));
    CK(hipMemcpy(d_x, x, nin * 4, hipMemcpyHostToDevice));
    if (tab_x) { CK(hipMalloc(&d_tx, (size_t)(ntab + 1) * 4)); CK(hipMemset(d_tx, 0, (size_t)(ntab + 1) * 4)); CK(hipMemcpy(d_tx, tab_x, (size_t)ntab * 4, hipMemcpyHostToDevice)); }
    if (tab_f) { CK(hipMalloc(&d_tf, (size_t)(ntab + 1) * 4)); CK(hipMemset(d_tf, 0, (size_t)(ntab + 1) * 4)); CK(hipMemcpy(d_tf, tab_f, (size_t)ntab * 4, hipMemcpyHostToDevice)); }
    const int block = 256, grid = (n + block - 1) / block;
    switch (fn) {
    case 0: hipLaunchKernelGGL(ref_k_interp_property, dim3(grid), dim3(block), 0, 0, n, d_x, d_tf, ntab, start, step, d_out); break;
    case 1: hipLaunchKernelGGL(ref_k_interp_idx, dim3(grid), dim3(block), 0, 0, n, d_x, ntab, d_tx, d_out); break;
    case 2: hipLaunchKernelGGL(ref_k_interp, dim3(grid), dim3(block), 0, 0, n, d_x, ntab, d_tx, d_tf, d_out); break;
    case 3: hipLaunchKernelGGL(ref_k_rotate, dim3(grid), dim3(block), 0, 0, n, d_x, d_out); break;
    default: return -1;
    }
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, d_out, nout * 4, hipMemcpyDeviceToHost));
    hipFree(d_x); hipFree(d_out); if (d_tx) hipFree(d_tx); if (d_tf) hipFree(d_tf);
    return 0;
}
