#!/usr/bin/env python3
"""Folds the per-workload PMC summaries of tools/pmc_collect.sh (OUTDIR/pmc_<workload>.txt) into the JSON that bench.py
reads (profiles/pmc.json): per workload the counters of its dominant kernel -- one launch = the mean over the
dispatches of the run -- and the HBM bytes corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts
64 B per 128-B request: doubled; FETCH_SIZE / WRITE_SIZE are in KB).

    python3 tools/pmc_to_json.py OUTDIR WORKLOAD [WORKLOAD...]  > OUTDIR/pmc.json
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, kernel_source_sha  # noqa: E402


def parse(path):
    kernels, current = {}, None
    for line in open(path):
        if not line.startswith(" "):
            current = kernels.setdefault(line.strip(), {})
            continue
        m = re.match(r"\s+(\S+)\s+mean\s+([0-9.eE+-]+)\s+n=(\d+)", line)
        if m and current is not None:
            current[m.group(1)] = float(m.group(2))
            current["_dispatches"] = max(current.get("_dispatches", 0), int(m.group(3)))
    return kernels


def frame_kernels(kernels):
    """The kernels of ONE frame among what a run dispatched: of the ambient-occlusion pass's two forms (with / without
    look-ahead loads: a ring's calibration at upload launches both) the one the frames were rendered with, i.e. the one with
    the most dispatches; never what an upload runs once (entry_kernel) or a statistic asks for (occluded_sum_kernel)."""
    names = [n for n in kernels if "ocrt::" in n and "entry_kernel" not in n and "occluded_sum" not in n and "frame_kernel" not in n]
    ao = [n for n in names if "ao_kernel" in n]
    if len(ao) > 1:
        keep = max(ao, key=lambda n: kernels[n].get("_dispatches", 0))
        names = [n for n in names if n not in ao or n == keep]
    return names


def in_frame(name):
    """A kernel of every frame -- not the one an upload runs once (entry_kernel: the tiles' walk intervals)."""
    return "ocrt::" in name and "entry_kernel" not in name


def main():
    out_dir, workloads = sys.argv[1], sys.argv[2:]
    result = {
        "kernel_source_sha256": kernel_source_sha(),
        "source": "profiles/r05_pmc_<workload>.txt (rocprofv3 --pmc, one pass per counter group, tools/pmc_collect.sh)",
        "units": "per launch of the dominant kernel; SQ_*_CYCLES and SQ_WAIT_* count quad-cycles",
        "workloads": {},
    }
    for w in workloads:
        kernels = parse(os.path.join(out_dir, f"pmc_{w}.txt"))
        want = "ao_kernel" if WORKLOADS[w]["ao"] else "primary_kernel"
        names = [k for k in frame_kernels(kernels) if want in k]
        if not names:
            continue
        c = kernels[names[0]]
        fetch_kb, write_kb = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
        entry = {
            "kernel": names[0],
            "valu_insts": c.get("SQ_INSTS_VALU"), "salu_insts": c.get("SQ_INSTS_SALU"), "smem_insts": c.get("SQ_INSTS_SMEM"),
            # every kernel of a frame together (primary pass with the ordering step, ambient-occlusion pass, finishing kernel)
            "frame_valu_insts": sum(kernels[name].get("SQ_INSTS_VALU", 0.0) for name in frame_kernels(kernels)),
            "vmem_rd_insts": c.get("SQ_INSTS_VMEM_RD"), "lds_insts": c.get("SQ_INSTS_LDS"), "branch_insts": c.get("SQ_INSTS_BRANCH"),
            "waves": c.get("SQ_WAVES"), "wave_quad_cycles": c.get("SQ_WAVE_CYCLES"), "busy_cycles": c.get("SQ_BUSY_CYCLES"),
            "wait_any_quad_cycles": c.get("SQ_WAIT_ANY"), "wait_inst_any_quad_cycles": c.get("SQ_WAIT_INST_ANY"),
            "active_inst_valu_quad_cycles": c.get("SQ_ACTIVE_INST_VALU"), "gui_active_cycles": c.get("GRBM_GUI_ACTIVE"),
            "hbm_bytes": (2.0 * fetch_kb + write_kb) * 1024.0,
            "hbm_fetch_kb_raw": fetch_kb, "hbm_write_kb": write_kb,
            # bytes by level: scalar loads are 64 B (a node and its successor) except the 16- and 32-B triangle / record
            # loads of leaves tested on the spot, so 64 B per instruction is an upper bound; TCP / TCC count 64-B and
            # 128-B requests
            "levels": {
                "scalar_cache_bytes_upper": None if c.get("SQ_INSTS_SMEM") is None else c["SQ_INSTS_SMEM"] * 64.0,
                "vector_l1_accesses": c.get("TCP_TOTAL_CACHE_ACCESSES_sum"), "l1_to_l2_read_requests": c.get("TCP_TCC_READ_REQ_sum"),
                "l2_requests": c.get("TCC_REQ_sum"), "l2_hits": c.get("TCC_HIT_sum"), "l2_misses": c.get("TCC_MISS_sum"),
                "hbm_bytes": (2.0 * fetch_kb + write_kb) * 1024.0,
            },
            # What one SIMD sustains on the node test's own instruction mix with 8 waves resident
            # (tools/microbench/valu_rate_probe.hip, profiles/r04_valu_rate_probe.txt): 3.38 cycles per instruction for the
            # 12-instruction centre / half-extent test of the ambient-occlusion pass (rounds 2-3: the 9-instruction plane
            # form, 3.15), 3.24 for the 11-instruction one of the primary pass; only v_fma / v_mul / v_add / v_mov on
            # REGISTERS reach the guide's 2 cycles (2.3 measured) -- an fma with an SGPR operand, as every fma of a node
            # test is, issues at 3-4.2.
            "valu_ceiling_measured": round(1.0 / 3.38, 3) if WORKLOADS[w]["ao"] else round(1.0 / 3.24, 3),
        }
        shared_path = os.path.join(out_dir, f"pmc_{w}__shared.txt")
        if os.path.exists(shared_path):  # the same frame with the grid of a host that shares its GPU
            shared = parse(shared_path)
            names = [k for k in shared if want in k]
            if names:
                entry["shared_valu_insts"] = shared[names[0]].get("SQ_INSTS_VALU")
                entry["shared_waves"] = shared[names[0]].get("SQ_WAVES")
                entry["shared_frame_valu_insts"] = sum(shared[name].get("SQ_INSTS_VALU", 0.0) for name in frame_kernels(shared))
        result["workloads"][w] = entry
    json.dump(result, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
