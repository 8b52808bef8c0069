"""Reads a rocprofv3 --kernel-trace CSV and prints, frame by frame, when the ray-casting kernels began and ended (us from
the first one) and on which queue: shows how consecutive frames overlap.

    python3 tools/trace_overlap.py path/to/*_kernel_trace.csv [first_row] [rows]
"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    want = ("primary_kernel", "ao_kernel", "finish_kernel", "resize_kernel")
    ks = [r for r in rows if any(w in r["Kernel_Name"] for w in want)]
    ks.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(ks[0]["Start_Timestamp"])
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    for r in ks[first:first + count]:
        name = next(w for w in want if w in r["Kernel_Name"])
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"queue {r.get('Queue_Id', '?'):>3} {name:16s} {s:10.1f} -> {e:10.1f} us  ({e - s:7.1f})")


if __name__ == "__main__":
    main()
