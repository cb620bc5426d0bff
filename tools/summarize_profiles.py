#!/usr/bin/env python3
"""Condense a gpurun_out/prof_* directory (rocprofv3 csv output) into the small, judged files under profiles/.

usage: tools/summarize_profiles.py gpurun_out/prof_r2 profiles/r2 [workload]
writes <out>_kernel_stats.csv (the --stats summary, verbatim), <out>_frame_timeline.txt, <out>_pmc.json and
updates profiles/pmc_traffic.json (per-launch HBM traffic and pipe figures of the blend kernel for that workload, read by
bench.py).

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE
tallies 128-B requests at 64 B, i.e. reads come out at one half for wide coalesced streams — the corrected figure
doubles it; WRITE_SIZE is exact.  Both raw and corrected values are kept.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "bicycle"
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    shutil.copy(os.path.join(src, "trace_kernel_stats.csv"), out + "_kernel_stats.csv")
    tl = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "frame_timeline.py"),
                         os.path.join(src, "trace_kernel_trace.csv"), "+20"], capture_output=True, text=True).stdout  # a frame of the timed loop
    open(out + "_frame_timeline.txt", "w").write(tl)
    pmc = {}
    stage_sum = collections.defaultdict(float)  # (stage, counter) -> total over every launch of the run
    frames = {}
    BIN_SORT = ("radix_hist", "radix_rowscan", "radix_scatter", "pair_count", "pair_scan", "pair_emit", "tile_ranges", "tile_order", "shard_compact")
    for f in sorted(glob.glob(os.path.join(src, "pmc_*_counter_collection.csv"))):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            kern = "blend_kernel" if ("blend_walk_kernel" in r["Kernel_Name"] or "blend_kernel" in r["Kernel_Name"]) else ("preprocess_kernel" if "preprocess_kernel" in r["Kernel_Name"] else r["Kernel_Name"][:40])
            per[kern][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if any(k in r["Kernel_Name"] for k in BIN_SORT):
                stage_sum[("bin_sort", r["Counter_Name"])] += float(r["Counter_Value"])
        for c, v in per.get("blend_kernel", {}).items():
            frames[c] = len(v)  # one blend launch per frame
        for kern, cs in per.items():
            for c, v in cs.items():
                steady = v[len(v) // 2:]  # drop warm-up launches
                pmc.setdefault(kern, {})[c] = {"per_launch_mean": sum(steady) / len(steady), "launches": len(v)}
    derived = {}
    for kern, cs in pmc.items():
        d = {}
        if "FETCH_SIZE" in cs:
            d["fetch_bytes_raw"] = cs["FETCH_SIZE"]["per_launch_mean"] * 1024
            d["fetch_bytes_corrected_x2"] = 2 * d["fetch_bytes_raw"]
        if "WRITE_SIZE" in cs:
            d["write_bytes"] = cs["WRITE_SIZE"]["per_launch_mean"] * 1024
        if "fetch_bytes_raw" in d and "write_bytes" in d:
            d["hbm_bytes_per_launch"] = d["fetch_bytes_corrected_x2"] + d["write_bytes"]
        derived[kern] = d
    json.dump({"counters": pmc, "derived": derived}, open(out + "_pmc.json", "w"), indent=1)
    tfile = os.path.join(os.path.dirname(out) or ".", "pmc_traffic.json")
    traffic = json.load(open(tfile)) if os.path.exists(tfile) else {}
    if "blend_kernel" in derived and "hbm_bytes_per_launch" in derived["blend_kernel"]:
        t = traffic.setdefault(workload, {})
        t["blend_kernel_bytes_per_launch"] = derived["blend_kernel"]["hbm_bytes_per_launch"]
        t["source"] = os.path.basename(out) + "_pmc.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)"
        c = {k: v["per_launch_mean"] for k, v in pmc["blend_kernel"].items()}
        try:  # pipe figures: per evaluated (quadrant, entry) instruction counts need the frame's counter from the traced bench
            we = json.load(open(os.path.join(src, "bench_under_trace.json")))["stats"]["wave_entries"]
            t["wave_entries"] = we
            if "SQ_INSTS_VALU" in c:
                t["valu_insts_per_wave_entry"] = c["SQ_INSTS_VALU"] / we
            if "SQ_INSTS_SALU" in c:
                t["salu_insts_per_wave_entry"] = c["SQ_INSTS_SALU"] / we
        except Exception as e:  # noqa: BLE001
            print("no per-entry figures:", e)
        if "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
            rows = [r for r in csv.DictReader(open(out + "_kernel_stats.csv")) if "blend_walk_kernel" in r["Name"] or "blend_kernel" in r["Name"]]
            if rows:
                t["clock_ghz"] = cyc / float(rows[0]["AverageNs"])
            if "SQ_LDS_IDX_ACTIVE" in c:
                t["lds_busy_frac"] = c["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc  # LDS-array cycles per CU over the kernel's cycles
            if "SQ_INSTS_SALU" in c:
                t["salu_issue_frac"] = c["SQ_INSTS_SALU"] / 256.0 / cyc  # one scalar unit per CU
            if "SQ_INSTS_VALU" in c:
                t["valu_issue_frac_2cyc"] = c["SQ_INSTS_VALU"] * 2.0 / 1024.0 / cyc  # at the 2-cycle floor of a wave64 VALU op
    # per-stage HBM traffic for bench.py's stage_rooflines: the preprocess kernel per launch, the bin + sort stage per frame (all its
    # dispatches together; FETCH_SIZE x2 + WRITE_SIZE like the blend's)
    t = traffic.setdefault(workload, {})
    if "preprocess_kernel" in derived and "hbm_bytes_per_launch" in derived["preprocess_kernel"]:
        t["preprocess_bytes_per_launch"] = derived["preprocess_kernel"]["hbm_bytes_per_launch"]
    if ("bin_sort", "FETCH_SIZE") in stage_sum and ("bin_sort", "WRITE_SIZE") in stage_sum and frames.get("FETCH_SIZE") and frames.get("WRITE_SIZE"):
        t["bin_sort_bytes_per_frame"] = (2 * stage_sum[("bin_sort", "FETCH_SIZE")] / frames["FETCH_SIZE"] + stage_sum[("bin_sort", "WRITE_SIZE")] / frames["WRITE_SIZE"]) * 1024
    json.dump(traffic, open(tfile, "w"), indent=1)
    print(tl)
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    main()
