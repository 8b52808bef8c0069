"""Summarises rocprofv3 --pmc CSV output: per kernel name, mean of each counter
over dispatches.  usage: python tools/pmc_summary.py DIR [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "ocrt::"
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name", "")
                if want not in name:
                    continue
                short = name.split("(")[0]
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for kernel, counters in acc.items():
        print(kernel)
        for c in sorted(counters):
            v = counters[c]
            print(f"  {c:36s} mean {sum(v) / len(v):18.1f}  n={len(v)}")


if __name__ == "__main__":
    main()
