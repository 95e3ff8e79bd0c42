#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/traffic.json (HBM bytes per launch).

Collected as MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
(they do not fit one pass), values are KiB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced
streaming read, so it is doubled.  Calibration on a known byte count in this code's own access pattern: the
analysis kernel reads rows*(Tn*hop+N)*4 = 15.4 MB of unique input per launch and reports FETCH_SIZE = 7.44 MB
(x2 = 14.9 MB); WRITE_SIZE matches the 134.3 MB of mag+phase planes it writes exactly.

usage: tools/make_traffic.py <pmc_dir> [out.json]     (pmc_dir holds the FETCH_SIZE and WRITE_SIZE pass outputs)
"""
import csv
import glob
import json
import sys
from collections import defaultdict

def canonical(k):
    """pv_analyze_wave_kernel<1024, 1> and friends are reported under the stage's kernel name"""
    if k.startswith("pv_analyze_wave_kernel") or k.startswith("pv_analyze_split_kernel"):
        return "pv_analyze_kernel"
    if k.startswith("pv_synth_wave_kernel"):
        return "pv_synth_kernel"
    if k.startswith("pv_ola_kernel"):  # pv_ola_kernel<resampling mode>
        return "pv_ola_kernel"
    if k.startswith("pv_synth_chain_kernel"):  # fused synthesis + overlap-add
        return "pv_synth_ola_kernel"
    if k.startswith("pv_seq_ring_kernel") or k.startswith("pv_seq_kernel"):
        return "pv_seq_kernel"
    if k.startswith("pv_resample_kernel") or k.startswith("pv_resample_fast_kernel") or k.startswith("pv_frames_chain_kernel"):
        return "pv_ola_kernel"  # the fused path's second kernel, reported under the overlap-add stage's name
    return k


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else "profiles/traffic.json"
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            k = r["Kernel_Name"].split("(")[0].replace("pv::", "").replace("void ", "")
            if k.startswith("pv_"):
                acc[canonical(k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res, detail = {}, {}
    for k, c in acc.items():
        # the first and last chunk of a run are shorter: use the median launch
        med = lambda v: sorted(v)[len(v) // 2]
        f_kib, w_kib = med(c["FETCH_SIZE"]), med(c["WRITE_SIZE"])
        res[k] = int((2.0 * f_kib + w_kib) * 1024)
        detail[k] = {"FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib, "launches": len(c["FETCH_SIZE"])}
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    import bench
    import os
    arith = "exact" if os.environ.get("AUDIOMOD_PV_EXACT", "0") not in ("", "0") else "fast"
    json.dump({"source_sha16": bench.kernel_source_hash(), "arithmetic": arith, "note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes, median launch; "
                       "bench workload geometry (128 stereo streams; since round 2's last change 512-slice chunks = 131072 slices "
                       "per launch, 65536 before -- compare with the bench line's roofline.slices_per_launch)",
               "bytes_per_launch": res, "counters": detail}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
