#!/bin/bash
# Diagnostic build of the library with extra -D switches: tools/build_variant.sh <name> [-DPV_EXP_...]...
# -> audiomod_amd/lib/diag/<name>/libaudiomod_pv.so (run with tools/diag_run.py <name> [bench args])
set -e
N=$1; shift
mkdir -p audiomod_amd/lib/diag/$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Iinclude -Iaudiomod_amd/csrc -Wno-unused-result "$@" -shared audiomod_amd/csrc/pv_kernels.hip audiomod_amd/csrc/pv_hostio.hip audiomod_amd/csrc/pv_engine.cc audiomod_amd/csrc/pv_plan.cc audiomod_amd/csrc/phasevocoder.cc -o audiomod_amd/lib/diag/$N/libaudiomod_pv.so 2>&1 | grep -E "error" || true
