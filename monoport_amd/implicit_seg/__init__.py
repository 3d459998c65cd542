"""Drop-in for the ``implicit_seg`` package the reference imports (RTL/main.py:28-29;
pip-from-git dependency ``implicit-seg``, requirements.txt:15, not vendored).

To let the reference's ``from implicit_seg.functional import Seg3dLossless`` resolve here:

    import sys, monoport_amd.implicit_seg as iseg
    sys.modules["implicit_seg"] = iseg
    sys.modules["implicit_seg.functional"] = iseg.functional
    sys.modules["implicit_seg.functional.utils"] = iseg.functional.utils
"""
from . import functional  # noqa: F401
