"""GPURays: ray bundles on the device and the `render` kernel (reference: chroma/gpu/render.py:7-66).

Same constructor, methods and arguments; the transforms and the render call go through the C ABI
(chroma_points_translate / _rotate / _rotate_around_point, chroma_render) instead of PyCUDA kernels looked
up by name.  ``nblocks`` (threads per block in the reference) is accepted and ignored.
"""
import ctypes

import numpy as np

from chroma_amd import _lib
from chroma_amd.gpu.tools import GPUArray, get_context, to_float3, to_gpu, zeros, empty, vec

float4 = np.dtype([('x', np.float32), ('y', np.float32), ('z', np.float32), ('w', np.float32)])


def _f3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


class GPURays(object):
    """The GPURays class holds arrays of ray positions and directions
    on the GPU that are used to render a geometry."""

    def __init__(self, pos, dir, max_alpha_depth=10, nblocks=64):
        self.ctx = get_context()
        self.pos = to_gpu(to_float3(pos), self.ctx)
        self.dir = to_gpu(to_float3(dir), self.ctx)
        self.max_alpha_depth = max_alpha_depth
        self.nblocks = nblocks
        self.dx = empty(max_alpha_depth * self.pos.size, np.float32, self.ctx)
        self.color = empty(self.dx.size, float4, self.ctx)
        self.dxlen = zeros(self.pos.size, np.uint32, self.ctx)

    def rotate(self, phi, n):
        "Rotate by an angle phi around the axis `n`."
        lib, h = self.ctx._lib, self.ctx.handle
        _lib.check(lib.chroma_points_rotate(h, self.pos.size, self.pos.ptr, float(phi), _f3(n)))
        _lib.check(lib.chroma_points_rotate(h, self.dir.size, self.dir.ptr, float(phi), _f3(n)))

    def rotate_around_point(self, phi, n, point):
        """"Rotate by an angle phi around the axis `n` passing through
        the point `point`."""
        lib, h = self.ctx._lib, self.ctx.handle
        _lib.check(lib.chroma_points_rotate_around_point(h, self.pos.size, self.pos.ptr, float(phi), _f3(n), _f3(point)))
        _lib.check(lib.chroma_points_rotate(h, self.dir.size, self.dir.ptr, float(phi), _f3(n)))

    def translate(self, v):
        "Translate the ray positions by the vector `v`."
        _lib.check(self.ctx._lib.chroma_points_translate(self.ctx.handle, self.pos.size, self.pos.ptr, _f3(v)))

    def render(self, gpu_geometry, pixels, alpha_depth=10, keep_last_render=False, bg_color=0x00000000):
        """Render `gpu_geometry` and fill the GPU array `pixels` with pixel
        colors."""
        if not keep_last_render:
            self.dxlen.fill(0)
        if alpha_depth > self.max_alpha_depth:
            raise Exception('alpha_depth > max_alpha_depth')
        if not isinstance(pixels, GPUArray):
            raise TypeError('`pixels` must be a %s instance.' % GPUArray)
        if pixels.size != self.pos.size:
            raise ValueError('`pixels`.size != number of rays')
        _lib.check(self.ctx._lib.chroma_render(self.ctx.handle, gpu_geometry.gpudata, self.pos.size, self.pos.ptr, self.dir.ptr,
                                               int(alpha_depth), pixels.ptr, self.dx.ptr, self.dxlen.ptr, self.color.ptr,
                                               int(bg_color) & 0xFFFFFFFF))

    def snapshot(self, gpu_geometry, alpha_depth=10):
        "Render `gpu_geometry` and return a numpy array of pixel colors."
        pixels = empty(self.pos.size, np.uint32, self.ctx)
        self.render(gpu_geometry, pixels, alpha_depth)
        return pixels.get()
