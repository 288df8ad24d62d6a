"""``import chroma`` for code written against the reference package.

The reference's tests and drivers say ``from chroma.sim import Simulation``, ``from chroma import gpu``,
``from chroma.event import Photons`` (chroma/test/test_propagation.py:4-8, bin/chroma-sim).  This package
makes those imports resolve to the MI355X engine: ``chroma`` IS ``chroma_amd`` and every ``chroma.X`` is the
very module object ``chroma_amd.X`` (one class identity, so ``isinstance`` holds across the two names).
Put the repository root on ``sys.path`` (or install it) and the reference's host code runs against
libchroma_hip.so; nothing from the reference tree is imported or needed.
"""
import importlib
import pkgutil
import sys

import chroma_amd as _real

_this = sys.modules[__name__]
for _name in ('__doc__',):
    pass
# every submodule of chroma_amd under its reference name (modules that need an absent optional dependency are skipped)
for _info in pkgutil.walk_packages(_real.__path__, prefix='chroma_amd.'):
    try:
        _mod = importlib.import_module(_info.name)
    except Exception:       # pragma: no cover
        continue
    sys.modules['chroma' + _info.name[len('chroma_amd'):]] = _mod
for _key, _value in vars(_real).items():
    if not _key.startswith('__'):
        setattr(_this, _key, _value)
__path__ = []          # nothing of its own below this package: every chroma.X was registered above
