#!/usr/bin/env python3
"""Lab helper: run a Python script (bench.py, scripts/microbench.py ...) against a VARIANT build of the library.
    python scripts/lab/with_lib.py video_llava_amd/libpgv_nopipe.so bench.py --workload vision ...
Variant libraries come from video_llava_amd.build.build_variant / build(lab=True); the product never loads them."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_llava_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
assert os.path.exists(_lib.LIB_PATH), _lib.LIB_PATH
if sys.argv[2] == "-m":                      # python scripts/lab/with_lib.py lib.so -m pytest tests/test_gpu_vision.py ...
    mod = sys.argv[3]
    sys.argv = sys.argv[3:]
    runpy.run_module(mod, run_name="__main__", alter_sys=True)
else:
    script = sys.argv[2]
    sys.argv = sys.argv[2:]
    runpy.run_path(script, run_name="__main__")
