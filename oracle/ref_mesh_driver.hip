// ref_mesh_driver.hip -- TEST INFRASTRUCTURE.  A host driver around the reference's OWN
// ray-cast code: it #includes /root/reference/chroma/cuda/mesh.h (found through -I, see
// oracle/Makefile) and launches the reference's distance_to_mesh kernel (mesh.h:124-151)
// and a one-line kernel that calls the reference's intersect_mesh (mesh.h:42-118) so
// that the hit triangle id is visible too.  No reference source is copied here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "mesh.h"
#include "render.cu"          // the reference's render kernel (chroma/cuda/render.cu:37-181), by path, unmodified

extern "C" __global__ void
ref_intersect_mesh_kernel(int nthreads, float3 *_origin, float3 *_direction, int *_last_hit,
                          Geometry *g, float *_distance, int *_triangle)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nthreads)
        return;
    float3 origin = _origin[id];
    float3 direction = _direction[id];
    direction /= norm(direction);
    float distance;
    int last = _last_hit ? _last_hit[id] : -1;
    int triangle_index = intersect_mesh(origin, direction, g, distance, last);
    _triangle[id] = triangle_index;
    if (triangle_index != -1)
        _distance[id] = distance;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "ref_mesh: %s failed: %s\n", #x, hipGetErrorString(e_)); return (int)e_; } } while (0)

// use_reference_kernel != 0: launch the reference's distance_to_mesh (distance only)
extern "C" int
ref_mesh_run(const float *vertices, uint32_t nvertices, const uint32_t *triangles, uint32_t ntriangles,
             const uint32_t *nodes, uint32_t nnodes, const float world_origin[3], float world_scale,
             int n, const float *origins, const float *directions, const int *last_hits,
             float *out_distance, int *out_triangle, int use_reference_kernel)
{
    float3 *d_vertices; uint3 *d_triangles; uint4 *d_nodes; Geometry *d_geo;
    float3 *d_o, *d_d; int *d_last = nullptr, *d_tri; float *d_dist;
    CK(hipMalloc(&d_vertices, (size_t)nvertices * 12));
    CK(hipMalloc(&d_triangles, (size_t)ntriangles * 12));
    CK(hipMalloc(&d_nodes, (size_t)nnodes * 16));
    CK(hipMalloc(&d_geo, sizeof(Geometry)));
    CK(hipMalloc(&d_o, (size_t)n * 12));
    CK(hipMalloc(&d_d, (size_t)n * 12));
    CK(hipMalloc(&d_tri, (size_t)n * 4));
    CK(hipMalloc(&d_dist, (size_t)n * 4));
    CK(hipMemcpy(d_vertices, vertices, (size_t)nvertices * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_triangles, triangles, (size_t)ntriangles * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_nodes, nodes, (size_t)nnodes * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_o, origins, (size_t)n * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_d, directions, (size_t)n * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dist, out_distance, (size_t)n * 4, hipMemcpyHostToDevice));
    if (last_hits) {
        CK(hipMalloc(&d_last, (size_t)n * 4));
        CK(hipMemcpy(d_last, last_hits, (size_t)n * 4, hipMemcpyHostToDevice));
    }
    Geometry geo;
    memset(&geo, 0, sizeof geo);
    geo.vertices = d_vertices;
    geo.triangles = d_triangles;
    geo.primary_nodes = d_nodes;
    geo.extra_nodes = d_nodes;
    geo.world_origin = make_float3(world_origin[0], world_origin[1], world_origin[2]);
    geo.world_scale = world_scale;
    geo.nprimary_nodes = (int)nnodes;
    CK(hipMemcpy(d_geo, &geo, sizeof geo, hipMemcpyHostToDevice));

    int block = 64, grid = (n + block - 1) / block;
    if (use_reference_kernel) {
        hipLaunchKernelGGL(distance_to_mesh, dim3(grid), dim3(block), 0, 0, n, d_o, d_d, d_geo, d_dist);
    } else {
        hipLaunchKernelGGL(ref_intersect_mesh_kernel, dim3(grid), dim3(block), 0, 0, n, d_o, d_d, d_last, d_geo, d_dist, d_tri);
    }
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out_distance, d_dist, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (!use_reference_kernel && out_triangle)
        CK(hipMemcpy(out_triangle, d_tri, (size_t)n * 4, hipMemcpyDeviceToHost));
    hipFree(d_vertices); hipFree(d_triangles); hipFree(d_nodes); hipFree(d_geo);
    hipFree(d_o); hipFree(d_d); hipFree(d_tri); hipFree(d_dist); if (d_last) hipFree(d_last);
    return 0;
}


// The reference's own `render` kernel over a ray bundle: pixels, and the per-ray alpha-depth lists it keeps
// (dx, dxlen, color) in and out, so that a second call with keep_last_render semantics can be checked too.
extern "C" int
ref_render_run(const float *vertices, uint32_t nvertices, const uint32_t *triangles, uint32_t ntriangles,
               const uint32_t *colors, const uint32_t *nodes, uint32_t nnodes, const float world_origin[3], float world_scale,
               int n, const float *origins, const float *directions, uint32_t alpha_depth, uint32_t bg_color,
               uint32_t *pixels, float *dx, uint32_t *dxlen, float *color)
{
    float3 *d_vertices; uint3 *d_triangles; uint4 *d_nodes; Geometry *d_geo; unsigned int *d_colors;
    float3 *d_o, *d_d; unsigned int *d_pix, *d_len; float *d_dx; float4 *d_col;
    const size_t nd = (size_t)n * alpha_depth;
    CK(hipMalloc(&d_vertices, (size_t)nvertices * 12));
    CK(hipMalloc(&d_triangles, (size_t)ntriangles * 12));
    CK(hipMalloc(&d_colors, (size_t)ntriangles * 4));
    CK(hipMalloc(&d_nodes, (size_t)nnodes * 16));
    CK(hipMalloc(&d_geo, sizeof(Geometry)));
    CK(hipMalloc(&d_o, (size_t)n * 12));
    CK(hipMalloc(&d_d, (size_t)n * 12));
    CK(hipMalloc(&d_pix, (size_t)n * 4));
    CK(hipMalloc(&d_len, (size_t)n * 4));
    CK(hipMalloc(&d_dx, nd * 4));
    CK(hipMalloc(&d_col, nd * 16));
    CK(hipMemcpy(d_vertices, vertices, (size_t)nvertices * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_triangles, triangles, (size_t)ntriangles * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_colors, colors, (size_t)ntriangles * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_nodes, nodes, (size_t)nnodes * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_o, origins, (size_t)n * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_d, directions, (size_t)n * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pix, pixels, (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_len, dxlen, (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dx, dx, nd * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_col, color, nd * 16, hipMemcpyHostToDevice));
    Geometry geo;
    memset(&geo, 0, sizeof geo);
    geo.vertices = d_vertices;
    geo.triangles = d_triangles;
    geo.colors = d_colors;
    geo.primary_nodes = d_nodes;
    geo.extra_nodes = d_nodes;
    geo.world_origin = make_float3(world_origin[0], world_origin[1], world_origin[2]);
    geo.world_scale = world_scale;
    geo.nprimary_nodes = (int)nnodes;
    CK(hipMemcpy(d_geo, &geo, sizeof geo, hipMemcpyHostToDevice));
    int block = 64, grid = n / block + 1;
    hipLaunchKernelGGL(render, dim3(grid), dim3(block), 0, 0, n, d_o, d_d, d_geo, alpha_depth, d_pix, d_dx, d_len, d_col, bg_color);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(pixels, d_pix, (size_t)n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(dxlen, d_len, (size_t)n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(dx, d_dx, nd * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(color, d_col, nd * 16, hipMemcpyDeviceToHost));
    hipFree(d_vertices); hipFree(d_triangles); hipFree(d_colors); hipFree(d_nodes); hipFree(d_geo);
    hipFree(d_o); hipFree(d_d); hipFree(d_pix); hipFree(d_len); hipFree(d_dx); hipFree(d_col);
    return 0;
}
