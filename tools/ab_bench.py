"""A/B helper: run bench.py against another build of the library.  python tools/ab_bench.py <lib.so> [bench args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import beta_recsys_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = lib
runpy.run_path(sys.argv[0], run_name="__main__")
