"""The C-ABI library loads and exports every symbol include/chroma_hip.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'chroma_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(chroma_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound():
    from chroma_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), 'libchroma_hip.so does not export %s' % name
        assert name in _lib.SIGNATURES, '%s has no ctypes signature in chroma_amd/_lib.py' % name
    assert set(_lib.SIGNATURES) == set(names)


def test_struct_sizes_match_the_header():
    from chroma_amd import _lib
    # LP64 layout of the structs in include/chroma_hip.h
    assert ctypes.sizeof(_lib.PhotonArrays) == 10 * 8
    assert ctypes.sizeof(_lib.Rng) == 16
    assert ctypes.sizeof(_lib.PropagateStats) == 136         # (+ physics_*, packet_*, reordered: round 3)
    assert ctypes.sizeof(_lib.GeometryDesc) % 8 == 0
    # round 4: what one call does (chroma_propagate_options: 12 int32) and where its hits go (chroma_hits_request: two
    # uint32, four pointers, the count -- padded to a multiple of 8)
    assert ctypes.sizeof(_lib.PropagateOptions) == 48
    assert ctypes.sizeof(_lib.HitsRequest) == 48 and _lib.HitsRequest.nhits.offset == 40 and _lib.HitsRequest.dst.offset == 8


def test_missing_library_fails_loudly(monkeypatch):
    from chroma_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIBRARY_PATH', '/nonexistent/libchroma_hip.so')
    with pytest.raises(_lib.ChromaError, match='no CPU fallback'):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'chroma_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(import|from)\s+oracle\b', text, flags=re.M), f
                assert 'oracle/' not in text.replace('the CPU oracle', ''), f
