"""Run bench.py against a side build of the C-ABI library (A/B of one kernel on the same box):
   python tools/ab_lib.py <library under monoport_amd/lib/side/ | product> [bench.py flags ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monoport_amd import _lib  # noqa: E402

name = sys.argv[1]
if name != "product":
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "side", name)
import bench  # noqa: E402

bench.main(sys.argv[2:])
