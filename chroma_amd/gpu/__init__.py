"""GPU host wrappers over libchroma_hip.so (reference package: chroma/gpu/__init__.py:1-7)."""
from chroma_amd.gpu.tools import (create_cuda_context, get_context, get_rng_states, chunk_iterator,
                                  to_float3, to_uint3, GPUArray, RNGStates, vec, cuda_options,
                                  device_count, empty, zeros, to_gpu)
from chroma_amd.gpu.geometry import GPUGeometry, pack_geometry
from chroma_amd.gpu.detector import GPUDetector
from chroma_amd.gpu.photon import GPUPhotons, GPUPhotonsSlice, generate_bomb
from chroma_amd.gpu.daq import GPUDaq, GPUChannels
from chroma_amd.gpu.funcs import get_cu_module, GPUFuncs
from chroma_amd.gpu.render import GPURays
