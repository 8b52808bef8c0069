#!/usr/bin/env python3
"""Copies what tools/final_measure.sh left under gpurun_out/<dir> into profiles/ (the tracked, judged copies):
pmc.json, the per-workload PMC summaries, the rocprofv3 kernel statistics of the default bench command and the
bench lines, all under the round's prefix.

    python3 tools/install_profiles.py gpurun_out/final_r02c r02
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    shutil.copy(os.path.join(src, "pmc", "pmc.json"), os.path.join(ROOT, "profiles", "pmc.json"))
    for path in glob.glob(os.path.join(src, "pmc", "pmc_*.txt")):
        shutil.copy(path, os.path.join(ROOT, "profiles", f"{prefix}_{os.path.basename(path)}"))
    stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{prefix}_kernel_stats_bench_bunny_1080p_ao.csv"))
    stats = glob.glob(os.path.join(src, "stats_one", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{prefix}_kernel_stats_bench_bunny_1080p_ao_one_at_a_time.csv"))
    lines = {}
    for path in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
        lines[os.path.basename(path)[len("bench_"):-len(".json")]] = json.loads(open(path).read())
    with open(os.path.join(ROOT, "profiles", f"{prefix}_bench_lines.json"), "w") as f:
        json.dump(lines, f, indent=1)
    for w, b in lines.items():
        r = b["roofline"]
        print(f"{w:24s} {b['value']:9.1f} Mrays/s  {b['ms_per_step']:8.4f} ms per frame, one at a time (pipelined {b['pipelined']['value']:9.1f}, "
              f"{b['pipelined']['ms_per_frame']['median']:8.4f} ms)  kernel {r['kernel_ms']:8.4f} ms  VALU {r['achieved']} "
              f"(frac {r['frac']}, of measured ceiling {r.get('frac_of_measured_ceiling')}, nominal issue share {r.get('valu_nominal_issue_share')})  "
              f"HBM {r['hbm'].get('measured_frac')}")


if __name__ == "__main__":
    main()
