#!/usr/bin/env python3
"""Copy the judged summaries from gpurun_out/<round>/ into profiles/<round>/ and derive profiles/traffic_<cfg>.json
(HBM bytes per k_step launch from the PMC passes; KiB units, gfx950 FETCH_SIZE x2 correction per MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r2"
src, dst = f"gpurun_out/{R}", f"profiles/{R}"
os.makedirs(dst, exist_ok=True)
for f in glob.glob(f"{src}/*_bench.json"):
    lines = [l for l in open(f) if l.startswith("{")]
    if lines:
        open(os.path.join(dst, os.path.basename(f)), "w").write(lines[-1])
for extra in ("c3_fresh", "c3_rollout"):
    st = sorted(glob.glob(f"{src}/prof_{extra}/*/*kernel_stats.csv"), key=os.path.getmtime)
    if st:
        shutil.copy(st[-1], f"{dst}/{extra}_kernel_stats.csv")
for f in ("stagger.txt", "trajectory_one_launch_per_step.txt", "launch_times.txt"):
    if os.path.exists(f"{src}/{f}"):
        shutil.copy(f"{src}/{f}", f"{dst}/{f}")
for cfg in ("c2", "c3", "c4", "c5", "c3_16384", "c3_65536", "c4_16384"):
    st = sorted(glob.glob(f"{src}/prof_{cfg}/*/*kernel_stats.csv"), key=os.path.getmtime)  # newest run
    if st:
        shutil.copy(st[-1], f"{dst}/{cfg}_kernel_stats.csv")
    out = {}
    for name, tag in (("WRITE_SIZE", "pmcw"), ("FETCH_SIZE", "pmcf")):
        fs = sorted(glob.glob(f"{src}/{tag}_{cfg}/*/*counter_collection.csv"), key=os.path.getmtime)  # newest run
        if not fs:
            continue
        rows = [r for r in csv.DictReader(open(fs[-1])) if "k_step" in r["Kernel_Name"]]
        v = [float(r["Counter_Value"]) for r in rows]
        if v:
            out[name] = {"dispatches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)}
    if len(out) == 2:
        w = out["WRITE_SIZE"]["mean_KiB"] * 1024
        r = out["FETCH_SIZE"]["mean_KiB"] * 1024 * 2
        bench = json.load(open(f"{dst}/{cfg}_bench.json"))
        res = {"config": cfg, "kernel": "k_step (fused loop, auto-reset, full refresh, replayed queue)",
               "envs": bench["config"]["envs_per_gpu"], "kernel_ms": bench["roofline"]["kernel_ms"],
               "hbm_GBps_from_counters": (w + r) / (bench["roofline"]["kernel_ms"] * 1e-3) / 1e9, "write_bytes_per_launch": w,
               "fetch_bytes_per_launch_corrected_x2": r, "hbm_bytes_per_launch": w + r,
               "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_env_step"] * bench["roofline"]["units_per_launch"],
               "counters": out,
               "method": "rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE in separate passes (tools/profile_all.sh); KiB units; "
                         "gfx950 FETCH_SIZE x2 correction per MI355X_MICROARCH.md"}
        json.dump(res, open(f"profiles/traffic_{cfg}.json", "w"), indent=1)
        print(cfg, "traffic %.1f MB vs algorithmic %.1f MB" % ((w + r) / 1e6, res["algorithmic_bytes_per_launch"] / 1e6))
