"""tests/golden/bvh_golden.json: layer sizes and SHA-256 of the packed node arrays of the recursive-grid BVH
(golden 4) for demo.tiny(), make.box(100), make.cube(1000) and a sphere, as built by
chroma_amd's builders.  The reference's builder needs a CUDA device (chroma/cuda/bvh.cu:2 includes <cuda.h>),
so these vectors pin the repository's two independent implementations (C++ and NumPy) against a committed
value -- a regression pin, not a pin on the reference's GPU build.
usage: python tools/gen_bvh_golden.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chroma_amd import demo, make
from chroma_amd.loader import create_geometry_from_obj

CASES = {'tiny': lambda: demo.tiny(), 'box100': lambda: make.box(100.0, 100.0, 100.0), 'cube1000': lambda: make.cube(1000.0),
         'sphere100_16': lambda: make.sphere(100.0, 16)}


def describe(geometry):
    nodes = np.ascontiguousarray(geometry.bvh.nodes).view(np.uint32).reshape(-1, 4)
    lo = [int(x) for x in geometry.bvh.layer_offsets]
    return {'ntriangles': int(len(geometry.mesh.triangles)), 'nnodes': int(len(nodes)),
            'layer_sizes': [b - a for a, b in zip(lo, lo[1:] + [len(nodes)])],
            'world_scale': float(np.float32(geometry.bvh.world_coords.world_scale)),
            'world_origin': [float(np.float32(x)) for x in geometry.bvh.world_coords.world_origin],
            'nodes_sha256': hashlib.sha256(nodes.tobytes()).hexdigest()}


if __name__ == '__main__':
    out = {name: describe(create_geometry_from_obj(build())) for name, build in CASES.items()}
    path = os.path.join(ROOT, 'tests', 'golden', 'bvh_golden.json')
    json.dump(out, open(path, 'w'), indent=1, sort_keys=True)
    print(open(path).read())
