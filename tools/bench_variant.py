"""Tuning aid: run bench.py against an ALTERNATIVE build of the library (A/B of a kernel change
without touching the product .so).
usage: python tools/bench_variant.py path/to/libpairnet_hip_variant.so [bench.py arguments]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pairnet_amd import hip
hip.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
