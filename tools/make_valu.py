#!/usr/bin/env python3
"""profiles/valu.json from a rocprofv3 --pmc SQ_INSTS_VALU pass of the bench workload (tools/pmc_valu.sh):
vector instructions per slice of each kernel (mean launch / slices per launch) with the mean issue cost of the
kernel's instruction mix (priced once per kernel variant with tools/pk_probe.hip's table -- DESIGN.md section 3 -- and
kept here as constants), stamped with the source hash and arithmetic setting it was measured on.

usage: tools/make_valu.py <pmc dir> <slices_per_launch> [out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from make_traffic import canonical  # noqa: E402

# mean issue cost relative to a two-source v_add_f32 (static mix of the kernel's main path; fma-heavy free-form
# variants are priced at their fma share: v_fma_f32 1.43, transcendental 3.2, f64 / SGPR-operand forms 1.65)
MEAN_COST = {"exact": {"pv_analyze_kernel": 1.17, "pv_synth_ola_kernel": 1.3, "pv_ola_kernel": 1.85,
                       "pv_match_kernel": 1.4, "pv_seq_kernel": 1.5},
             "fast": {"pv_analyze_kernel": 1.17, "pv_synth_ola_kernel": 1.27, "pv_ola_kernel": 1.38,
                      "pv_match_kernel": 1.4, "pv_seq_kernel": 1.5}}


def main():
    d, spl = sys.argv[1], float(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else "profiles/valu.json"
    arith = "exact" if os.environ.get("AUDIOMOD_PV_EXACT", "0") not in ("", "0") else "fast"
    acc = defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "SQ_INSTS_VALU":
                continue
            k = r["Kernel_Name"].split("(")[0].replace("pv::", "").replace("void ", "")
            if k.startswith("pv_"):
                acc[canonical(k)].append(float(r["Counter_Value"]))
    kernels = {}
    for k, v in acc.items():
        med = sorted(v)[len(v) // 2]  # (the first and last launch of a run are shorter)
        kernels[k] = {"valu_insts_per_slice": round(med / spl, 1), "mean_cost": MEAN_COST[arith].get(k, 1.3)}
    json.dump({"source_sha16": bench.kernel_source_hash(), "arithmetic": arith,
               "note": "SQ_INSTS_VALU of the median launch / slices per launch (separate --pmc pass of the bench workload at "
                       "20 s per stream) and the mean issue cost of each kernel's instruction mix relative to a two-source "
                       "v_add_f32 (tools/pk_probe.hip's price list, DESIGN.md section 3).  chip_peak: two-source instructions "
                       "per ms over the chip's 1024 SIMDs as the probe measures it.",
               "slices_per_launch": spl, "chip_peak_M_per_ms": 972.0, "kernels": kernels}, open(out, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
