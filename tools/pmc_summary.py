#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch, per kernel.
usage: tools/pmc_summary.py <dir-with-csvs> [...]"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("pv::", "").replace("void ", "")
                if not k.startswith("pv_"):
                    continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n {len(v)}")


if __name__ == "__main__":
    main()
