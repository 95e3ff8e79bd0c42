#!/usr/bin/env python3
"""Run bench.py against a diagnostic build of the library (audiomod_amd/lib/diag/<name>/): timing decomposition
experiments only, results of such builds are numerically meaningless.  usage: tools/diag_run.py <name> [bench args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiomod_amd.engine as E  # noqa: E402

E.LIB_PATH = os.path.join(ROOT, "audiomod_amd", "lib", "diag", sys.argv[1], "libaudiomod_pv.so")
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
