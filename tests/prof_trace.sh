#!/bin/bash
# usage (GPU box): tests/prof_trace.sh <tag> [mib]  -> gpurun_out/<tag>_ktrace.csv (kernel name, start, end; last call only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$1 -- python3 $R/tests/gpu_perf.py ${2:-64} ${3:-1} > $R/gpurun_out/$1.log 2>&1
python3 - "$R/gpurun_out/$1" > $R/gpurun_out/$1_ktrace.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%-40s q%-3s %10.3f %10.3f ms" % (r["Kernel_Name"][:40], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6))
PY
rm -rf $R/gpurun_out/$1
