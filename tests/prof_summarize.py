#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tests/prof_bench.sh <tag> (under gpurun_out/) into the small summaries
committed under profiles/<round>/:  bench_1gib.json, bench_1gib_kernel_stats.csv, pmc_fetch_write_by_kernel.json.

    python tests/prof_summarize.py <tag> <round>       e.g.  r1d r1
"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles", rnd)
os.makedirs(P, exist_ok=True)
shutil.copy(os.path.join(G, tag + "_bench.json"), os.path.join(P, "bench_1gib.json"))
def norm(kernel_name):
    """zada::k_name -- without return type, anonymous namespace, template arguments and parameter list (k_prev_links<true> is k_prev_links)."""
    n = kernel_name.replace("(anonymous namespace)::", "").split("(")[0]
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0]


def newest(pattern):
    """The files of the LAST run of a pass only: gpurun merges a call's output into gpurun_out/ next to what earlier calls left there, and two
    runs' counters must not be added up."""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]


shutil.copy(newest(os.path.join(G, tag + "_stats", "*", "*_kernel_stats.csv"))[0], os.path.join(P, "bench_1gib_kernel_stats.csv"))
out = {}
for ctr, sub in (("FETCH_SIZE", "_fetch"), ("WRITE_SIZE", "_write")):
    acc = {}
    for f in newest(os.path.join(G, tag + sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            name = norm(r["Kernel_Name"])
            a = acc.setdefault(name, {"sum": 0.0, "dispatches": 0})
            a["sum"] += float(r["Counter_Value"]); a["dispatches"] += 1
    out[ctr] = acc
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-checks --no-host-path --no-v1 "
                "--bzip2-mib 0 --lzma-entries 0` (ONE step of the 1 GiB Deflate_3 workload and nothing else, plus the box's copy-bandwidth probe); values in KB summed over the dispatches of each kernel; FETCH_SIZE is to be doubled on gfx950 "
                "(MI355X_MICROARCH.md, HBM section)")
import subprocess
try:
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "zip-ada_amd", "bench.py"], capture_output=True, text=True).stdout.strip())
    commit = commit + ("+uncommitted changes" if dirty else "")
except Exception:
    commit = None
out["_commit"] = commit          # the tree the passes were taken on (bench.py prints it beside the traffic figures it reads from here)
try:
    out["_corpus"] = json.load(open(os.path.join(P, "bench_1gib.json")))["config"]["workload"]
except Exception:
    out["_corpus"] = None
json.dump(out, open(os.path.join(P, "pmc_fetch_write_by_kernel.json"), "w"), indent=1)
# SQ counters: what the kernels that are nowhere near the HBM roofline are bound by
sq = {}
for f in newest(os.path.join(G, tag + "_sq", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = norm(r["Kernel_Name"]).replace("zada::", "")
        a = sq.setdefault(name, {})
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
bounds = {}
for k, c in sq.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    b = {"valu_active_over_wave_cycles": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4),
         "waiting_over_wave_cycles": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4),
         "lds_active_over_wave_cycles": round(lds / wc, 4),
         "lds_bank_conflict_share": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / lds, 4) if lds else 0.0,
         "insts": {"valu": c.get("SQ_INSTS_VALU", 0.0), "salu": c.get("SQ_INSTS_SALU", 0.0), "lds": c.get("SQ_INSTS_LDS", 0.0)},
         "counters": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_* SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT (own pass), sums over the kernel's dispatches"}
    b["resource"] = ("waiting on LDS / memory round trips" if b["waiting_over_wave_cycles"] > 0.6 else
                     "vector issue" if b["valu_active_over_wave_cycles"] > 0.15 else "mixed")
    bounds[k] = b
if bounds:
    json.dump(bounds, open(os.path.join(P, "sq_bounds_by_kernel.json"), "w"), indent=1)
rows = list(csv.DictReader(open(os.path.join(P, "bench_1gib_kernel_stats.csv"))))
for r in rows[:14]:
    k = norm(r["Name"])
    f = out["FETCH_SIZE"].get(k); w = out["WRITE_SIZE"].get(k)
    print("%-32s calls %3s avg %9.3f ms %6s%%  fetch(x2) %7.2f GB  write %7.2f GB per launch" % (
        k[:32], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"][:5],
        2 * f["sum"] / f["dispatches"] * 1024 / 1e9 if f else -1, w["sum"] / w["dispatches"] * 1024 / 1e9 if w else -1))
