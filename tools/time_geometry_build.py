#!/usr/bin/env python
"""Where the seconds of bench.py's `geometry_build_s` go: tools/time_geometry_build.py [config]  (needs the GPU for the
device BVH builder; CHROMA_LOG=1 adds the wide builder's own phase times on stderr)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    import bench
    from chroma_amd import demo, gpu, _lib
    from chroma_amd.loader import create_geometry_from_obj
    from chroma_amd.gpu.geometry import pack_geometry
    config = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'c3'
    builder = bench.CONFIGS[config][0]
    ctx = gpu.create_cuda_context(0)
    t = [time.time()]
    obj = getattr(demo, builder)()
    t.append(time.time())
    import numpy as np
    from chroma_amd.geometry import Mesh
    obj.flatten_timing = {}
    t0 = time.time(); obj.flatten(); t_flat = time.time() - t0
    t0 = time.time(); geo = create_geometry_from_obj(obj); t_bvh = time.time() - t0
    t.append(time.time())
    packed = pack_geometry(geo)
    t.append(time.time())
    packed.attach_wide_tree()
    t.append(time.time())
    gg = gpu.GPUDetector.from_packed(packed)
    t.append(time.time())
    if '--check' in sys.argv:          # the device tree against its host twin (CHROMA_TREE=levels on the host cores)
        t0 = time.time()
        host = _lib.wide_build(packed.arrays['nodes'], packed.desc.ntriangles)
        same = all(np.array_equal(np.asarray(host[k]).reshape(-1), np.asarray(packed.arrays[a]).reshape(-1)) for k, a in (
            ('wnodes', 'wide_nodes'), ('tri_to_record', 'wide_tri_to_record'), ('record_to_tri', 'wide_record_to_tri'), ('rank', 'wide_rank')))
        print('device tree == host twin: %s (host build %.1f s)' % (same, time.time() - t0))
    names = ['demo.%s() (solids placed)' % builder, 'create_geometry_from_obj (flatten + a-20 BVH)', 'pack_geometry (tables, arrays)',
             'attach_wide_tree (SAH tree + collapse; device when a context is current)', 'GPUDetector.from_packed (validation + upload)']
    d = packed.desc
    print('%s: %d triangles, %d nodes, %d wide nodes' % (config, d.ntriangles, d.nnodes, d.nwide))
    for n, a, b in zip(names, t[:-1], t[1:]):
        print('  %-58s %7.2f s' % (n, b - a))
        if n.startswith('create_geometry'):
            print('      %-54s %7.2f s' % ('Geometry.flatten', t_flat))
            print('      %-54s %7.2f s' % ('make_recursive_grid_bvh (device) + result on the host', t_bvh))
    print('  %-58s %7.2f s' % ('total', t[-1] - t[0]))


if __name__ == '__main__':
    main()
