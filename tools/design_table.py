#!/usr/bin/env python3
"""Prints the rows of DESIGN.md's measurement table from profiles/<round>_bench_lines.json (what tools/final_measure.sh +
tools/install_profiles.py left there), so that the table is the committed bench lines and nothing else.

    python3 tools/design_table.py r05
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = [  # workload, label, one-frame-at-a-time Mrays/s of rounds 4 / 3 / 2 / 1
    ("bunny_1080p_ao", "**bunny 1080p `-s 1 -a 3`** (headline)", "28 992 / 23 911 / 21 890 / 15 533"),
    ("bunny_1080p_primary", "bunny 1080p primary only", "12 308 / 7 526 / 6 730 / 7 900"),
    ("bunny_600_defaults", "bunny 600² CLI defaults", "27 323 / 23 720 / 20 177 / 13 500"),
    ("bunny_1080p_s64", "bunny 1080p `-s 64` (2.2 G rays)", "60 302 / 42 896 / 41 392 / 36 900"),
    ("interior_1080p_ao", "interior stand-in 1080p (synthetic)", "55 205 / 37 112 / 34 774 / 22 900"),
    ("interior_4k_ao", "interior stand-in 4K (synthetic)", "67 547 / 43 093 / 41 351 / 34 200"),
    ("interior_hard_1080p_ao", "HARDER interior stand-in 1080p (synthetic)", "new"),
    ("interior_hard_4k_ao", "HARDER interior stand-in 4K (synthetic)", "new"),
    ("terrain_2m_1080p_ao", "height field, 2.0 M triangles, 1080p (0.58 GB of scene)", "5 077"),
    ("terrain_20m_1080p_ao", "height field, 20.5 M triangles, 1080p (6.0 GB of scene)", "1 654"),
]


def thousands(x):
    return f"{x:,.0f}".replace(",", " ")


def main():
    prefix = sys.argv[1] if len(sys.argv) > 1 else "r05"
    lines = json.load(open(os.path.join(ROOT, "profiles", f"{prefix}_bench_lines.json")))
    print("| workload (1× MI355X, `bench.py --steps 20`) | Mrays/s, ONE FRAME AT A TIME (`value`; min–max of the blocks) | ms/frame | rounds 4 / 3 / 2 / 1 | "
          "Mrays/s, 3 frames in flight (`pipelined`) | ms/frame | dominant kernel alone, ms | its VALU/clk/SIMD (frac of 0.5; of measured ceiling) | HBM measured |")
    print("|---|---|---|---|---|---|---|---|---|")
    for key, label, earlier in ROWS:
        if key not in lines:
            continue
        b = lines[key]
        r = b["roofline"]
        blocks = b.get("blocks", {})
        spread = f" ({thousands(blocks['mrays_per_s_min'])}–{thousands(blocks['mrays_per_s_max'])})" if blocks.get("n", 0) > 1 else ""
        value = f"**{thousands(b['value'])}**" if key == "bunny_1080p_ao" else thousands(b["value"])
        valu = (f"{r['achieved']:.3f} ({r['frac']:.2f}; {r['frac_of_measured_ceiling']:.2f})" if r.get("achieved") is not None else "n/a")
        hbm = f"{100 * r['hbm']['measured_frac']:.1f} %" if r["hbm"].get("measured_frac") is not None else "n/a"
        print(f"| {label} | {value}{spread} | {b['ms_per_step']:.3f} | {earlier} | {thousands(b['pipelined']['value'])} | "
              f"{b['pipelined']['ms_per_frame']['median']:.3f} | {r['kernel_ms']:.3f} | {valu} | {hbm} |")
    c = lines["bunny_1080p_ao"].get("cpu_baseline")
    if c:
        print(f"| CPU beside it (same run; {c['cpu_model'].replace(' 64-Core Processor', '')}, {c['cores']} host cores) | reference kernel {c['value']:.1f}, "
              f"oracle {c['port_value']:.1f}; PGM byte-identical | | | | | | | |")


if __name__ == "__main__":
    main()
