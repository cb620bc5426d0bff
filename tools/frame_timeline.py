#!/usr/bin/env python3
"""Print the per-dispatch timeline of one steady-state frame from a rocprofv3 --kernel-trace CSV.
usage: tools/frame_timeline.py <..._kernel_trace.csv> [k | +k]     k (default 5): k-th frame from the end; +k: k-th from the start.
Note for bench.py traces: the frames at the END of a run belong to its per-stage timing loop, which records an event between
the three stages (a ~6 us bubble each); frames of the timed loop are free of them -- pick those with +k."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    arg = sys.argv[2] if len(sys.argv) > 2 else "5"
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "preprocess_kernel" in r["Kernel_Name"] or "preprocess_views_kernel" in r["Kernel_Name"]]
    if any("block_flags_kernel" in r["Kernel_Name"] for r in rows):  # scenes with block bounds: the flags kernel opens the frame
        starts = [i for i, r in enumerate(rows) if "block_flags_kernel" in r["Kernel_Name"]]
    span = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # how many frames (launch sequences) to print
    if arg.startswith("+"):
        s, e = starts[int(arg)], starts[int(arg) + span]
    else:
        s, e = starts[-int(arg)], starts[-int(arg) + span] if -int(arg) + span < 0 else len(rows)
    t0 = int(rows[s]["Start_Timestamp"])
    prev_end = None
    total = 0.0
    for r in rows[s:e]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (st - prev_end) / 1e3 if prev_end else 0.0
        total += (en - st) / 1e3
        name = r["Kernel_Name"].replace("void ", "").replace("gsr::", "")[:44]
        gy = int(r.get("Grid_Size_Y", 1) or 1)
        print(f"{(st - t0) / 1e3:9.1f} us  dur {(en - st) / 1e3:8.1f}  gap {gap:5.1f}  grid {int(r['Grid_Size_X']):>9d} x {gy}  wg {r['Workgroup_Size_X']:>4s}  {name}")
        prev_end = en
    print(f"frame span {(prev_end - t0) / 1e3:.1f} us, sum of kernels {total:.1f} us, {e - s} dispatches")


if __name__ == "__main__":
    main()
