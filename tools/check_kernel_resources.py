#!/usr/bin/env python3
"""Build gate for the hand-scheduled kernels (called by opencl_raytracer_amd/csrc/Makefile).

Reads hipcc's `-Rpass-analysis=kernel-resource-usage` remarks for kernels.hip and fails
the build when a hot kernel would spill vector registers to scratch or run at fewer than
8 waves per SIMD: walk_collect (kernels.hip) hard-codes its scratch registers inside
kernels pinned to a 64-VGPR budget (`amdgpu_waves_per_eu(8, 8)`), so growth in live VGPRs
would otherwise turn into silent scratch traffic in the hottest loop.

    check_kernel_resources.py <remarks.txt> [--table out.txt] [--report-only]

Hot kernels (must have VGPR spill 0, scratch 0, occupancy 8): the default path, i.e. the
shared-walk instantiations primary_kernel<true> and ao_kernel<1, true> (1 = UNIFORM).  The
first-generation instantiations (<.., false>, debug knob OCRT_NO_SHARED_WALK) and the RANDOM
mode (ao_kernel<2, ..>, outside the bit-exact contract) must keep the occupancy; their
spills are reported, not fatal.  SGPR spills go to VGPR lanes
(v_writelane / v_readlane outside the loops), not to memory; they are reported too.
"""
import re
import subprocess
import sys

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
        hot = name.startswith("primary_kernel<true>") or name.startswith("ao_kernel<1, true>")
        walker = hot or name.startswith("ao_kernel<") or name.startswith("primary_kernel<")
        if walker and k.get("Occupancy [waves/SIMD]") != "8":
            errors.append(f"{name}: occupancy {k.get('Occupancy [waves/SIMD]')} waves/SIMD, the walk is scheduled for 8")
        if hot and (k.get("VGPRs Spill") != "0" or k.get("ScratchSize [bytes/lane]") != "0"):
            errors.append(f"{name}: VGPR spill {k.get('VGPRs Spill')}, scratch {k.get('ScratchSize [bytes/lane]')} B/lane "
                          "in a hot kernel (walk_collect's fixed registers v56-v62 need the 64-VGPR budget to hold)")
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
