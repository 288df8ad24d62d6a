import cProfile, pstats, sys, time
sys.path.insert(0, '/root/repo')
import bench
from chroma_amd import demo, gpu
from chroma_amd.loader import create_geometry_from_obj
ctx = gpu.create_cuda_context(0)
obj = demo.detector29k()
pr = cProfile.Profile()
pr.enable()
obj.flatten()
geo = create_geometry_from_obj(obj)
from chroma_amd.gpu.geometry import pack_geometry
packed = pack_geometry(geo)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
