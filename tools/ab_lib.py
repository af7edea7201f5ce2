"""Runs bench.py against another build of the library (A/B of two kernels on ONE box: boxes differ by several per cent):
   python tools/ab_lib.py <path/to/librtm3d_hip.so> <bench.py arguments...>"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtm3d_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
