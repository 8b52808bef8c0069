#!/usr/bin/env python3
"""Build gate for the hand-scheduled kernels (called by opencl_raytracer_amd/csrc/Makefile).

Reads hipcc's `-Rpass-analysis=kernel-resource-usage` remarks for kernels.hip and fails
the build when a hot kernel would spill vector registers to scratch or run at fewer than
8 waves per SIMD: walk_collect (kernels.hip) hard-codes its scratch registers inside
kernels pinned to a 64-VGPR budget (`amdgpu_waves_per_eu(8, 8)`), so growth in live VGPRs
would otherwise turn into silent scratch traffic in the hottest loop.

    check_kernel_resources.py <remarks.txt> [--table out.txt] [--isa kernels.s] [--report-only]

Hot kernels (must have VGPR spill 0, scratch 0, occupancy 8): the default path, i.e. the
shared-walk instantiations primary_kernel<true> and ao_kernel<1, true> (1 = UNIFORM).  The
first-generation instantiations (<.., false>, debug knob OCRT_NO_SHARED_WALK) and the RANDOM
mode (ao_kernel<2, ..>, outside the bit-exact contract) must keep the occupancy; their
spills are reported, not fatal.  SGPR spills go to VGPR lanes (v_writelane / v_readlane), not
to memory -- but those are vector instructions, the very resource the walk is bound by, so
WHERE they land matters: with `--isa` (the device assembly of the same translation unit,
`hipcc --cuda-device-only -S`) the tool locates every lane operation of the hot kernels by
loop depth and fails the build when one sits inside the hand-scheduled node loop (the asm
blocks between .Lw_miss_a and .Lw_out), when the loop around it -- one turn per leaf stop or
batch -- holds more than LANE_OPS_PER_WALK_TURN of them, or when the per-packet code holds more
than LANE_OPS_PER_PACKET.
"""
import re
import subprocess
import sys

LANE_OPS_PER_WALK_TURN = 10  # v_readlane / v_writelane per turn of the loop around walk_collect, i.e. per leaf stop or batch (measured: 6 primary;
                             # 9 AO: four in the batch block, five single reloads on paths that exclude each other)
LANE_OPS_PER_PACKET = 40     # ... in the per-packet code around that loop (measured: 38 of ~700 vector instructions)

FIELDS = ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill",
          "VGPRs Spill", "LDS Size [bytes/block]")


def demangle(name: str) -> str:
    try:
        out = subprocess.run(["c++filt", name], capture_output=True, text=True, check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        return name
    return re.sub(r"\(.*", "", out)  # drop the parameter list


def parse(text: str):
    kernels, current = [], None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            current = {"name": demangle(m.group(1))}
            kernels.append(current)
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\S+) \[-Rpass-analysis", line)
        if m and current is not None and m.group(1).strip() in FIELDS:
            current[m.group(1).strip()] = m.group(2)
    return kernels


def lane_ops_by_place(isa_text: str):
    """Per kernel of the device assembly: where its v_readlane / v_writelane (SGPR spill traffic) sit.  Returns
    {demangled name: {"total", "in_node_loop", "walk_turn", "per_packet", "depths": {depth: count}}}."""
    lines = isa_text.split("\n")
    out, name, start = {}, None, 0
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m and name is None:
            name, start = m.group(1), i
        if line.startswith(".Lfunc_end") and name is not None:
            body = lines[start:i]
            regions, begin = [], None  # the hand-scheduled node loops: .Lw_miss_a_N ... .Lw_out_N
            for j, text in enumerate(body):
                if re.search(r"\.Lw_miss_a_\d+:", text):
                    begin = j
                if re.search(r"\.Lw_out_\d+:", text) and begin is not None:
                    regions.append((begin, j))
                    begin = None
            if regions:
                # loop depth of a line = the depth noted at the last basic-block label before it
                # ("in Loop: Header=... Depth=N" on the label's own line or on a "; %bb.N:" comment; a loop header lists
                # its parents first, over several comment lines, and says "This Inner Loop Header: Depth=N" /
                # "This Loop Header: Depth=N" last: that one counts)
                depth, depths = 0, []
                for j, text in enumerate(body):
                    m = re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)(.*)", text)
                    if m:
                        note, k = m.group(2), j + 1
                        while k < len(body) and re.match(r"^\s+;", body[k]):  # the comment's continuation lines
                            note += body[k]
                            k += 1
                        own = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", note)
                        inside = re.search(r"in Loop: Header=\S+ Depth=(\d+)", note)
                        depth = int(own.group(1)) if own else int(inside.group(1)) if inside else 0
                    depths.append(depth)
                lane = [j for j, text in enumerate(body) if "v_writelane" in text or "v_readlane" in text]
                walk_depth = max(depths[a] for a, _ in regions)  # the loop that holds the asm blocks
                by_depth = {}
                for j in lane:
                    by_depth[depths[j]] = by_depth.get(depths[j], 0) + 1
                out[demangle(name)] = {
                    "total": len(lane), "in_node_loop": sum(1 for j in lane if any(a <= j <= b for a, b in regions)),
                    "walk_turn": by_depth.get(walk_depth, 0), "per_packet": by_depth.get(walk_depth - 1, 0),
                    "depths": dict(sorted(by_depth.items())), "walk_depth": walk_depth}
            name = None
    return out


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    kernels = parse(open(sys.argv[1]).read())
    if not kernels:
        sys.exit("check_kernel_resources: no kernel-resource-usage remarks found in " + sys.argv[1])
    header = f"{'kernel':44s} {'SGPR':>5s} {'VGPR':>5s} {'scratch':>8s} {'waves/SIMD':>10s} {'SGPR spill':>10s} {'VGPR spill':>10s} {'LDS B':>7s}"
    lines, errors = [header], []
    for k in kernels:
        name = k["name"].replace("ocrt::", "").replace("void ", "")
        row = (f"{name:44s} {k.get('TotalSGPRs', '?'):>5s} {k.get('VGPRs', '?'):>5s} "
               f"{k.get('ScratchSize [bytes/lane]', '?'):>8s} {k.get('Occupancy [waves/SIMD]', '?'):>10s} "
               f"{k.get('SGPRs Spill', '?'):>10s} {k.get('VGPRs Spill', '?'):>10s} {k.get('LDS Size [bytes/block]', '?'):>7s}")
        lines.append(row)
        # hot: the default path = shared-walk instantiations of the primary pass and of the UNIFORM ambient-occlusion pass
        hot = name.startswith("primary_kernel<true>") or name.startswith("ao_kernel<1, true")
        walker = hot or name.startswith("ao_kernel<") or name.startswith("primary_kernel<")
        if walker and k.get("Occupancy [waves/SIMD]") != "8":
            errors.append(f"{name}: occupancy {k.get('Occupancy [waves/SIMD]')} waves/SIMD, the walk is scheduled for 8")
        if hot and (k.get("VGPRs Spill") != "0" or k.get("ScratchSize [bytes/lane]") != "0"):
            errors.append(f"{name}: VGPR spill {k.get('VGPRs Spill')}, scratch {k.get('ScratchSize [bytes/lane]')} B/lane "
                          "in a hot kernel (walk_collect's fixed registers v56-v62 need the 64-VGPR budget to hold)")
    if "--isa" in sys.argv:
        places = lane_ops_by_place(open(sys.argv[sys.argv.index("--isa") + 1]).read())
        lines.append("")
        lines.append("SGPR spill traffic (v_readlane / v_writelane) by place: kernel, total, inside the hand-scheduled node loop, "
                     "per turn of the loop around it, per packet, by loop depth")
        for kname, p in places.items():
            short = kname.replace("ocrt::", "").replace("void ", "")
            lines.append(f"{short:44s} {p['total']:5d} {p['in_node_loop']:5d} {p['walk_turn']:5d} {p['per_packet']:5d}   {p['depths']}")
            if not (short.startswith("primary_kernel<true>") or short.startswith("ao_kernel<1, true")):
                continue
            if p["in_node_loop"]:
                errors.append(f"{short}: {p['in_node_loop']} SGPR spill instructions inside the node loop")
            if p["walk_turn"] > LANE_OPS_PER_WALK_TURN:
                errors.append(f"{short}: {p['walk_turn']} SGPR spill instructions per turn of the loop around the node loop (limit {LANE_OPS_PER_WALK_TURN})")
            if p["per_packet"] > LANE_OPS_PER_PACKET:
                errors.append(f"{short}: {p['per_packet']} SGPR spill instructions in the per-packet code (limit {LANE_OPS_PER_PACKET})")
    table = "\n".join(lines) + "\n"
    if "--table" in sys.argv:
        with open(sys.argv[sys.argv.index("--table") + 1], "w") as f:
            f.write(table)
    print(table, end="")
    if errors and "--report-only" in sys.argv:  # instrumented builds (-DOCRT_STAMPS) carry extra live values
        print("check_kernel_resources (report only): " + "; ".join(errors))
    elif errors:
        sys.exit("check_kernel_resources: " + "; ".join(errors))


if __name__ == "__main__":
    main()
