"""chroma_amd -- MI355X-native photon propagation behind chroma's Python API.

The package mirrors the module layout of pennneutrinos/chroma for the propagate
path only: ``event``, ``geometry``, ``detector``, ``make``,
``pmt``, ``bvh``, ``gpu`` and ``sim``.  The GPU side is ``libchroma_hip.so``
(hand-written HIP for gfx950, C ABI in ``include/chroma_hip.h``) loaded through
ctypes by :mod:`chroma_amd._lib`; nothing here imports PyCUDA, CUDA headers or torch.
"""
__version__ = "0.1.0"
