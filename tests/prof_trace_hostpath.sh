#!/bin/bash
# usage (GPU box): tests/prof_trace_hostpath.sh <tag>  -> gpurun_out/<tag>_ktrace.txt: the kernel timeline of the LAST of the four 1 GiB calls of
# tests/gpu_hostpath.py (host buffers in and out: the input arrives while the link stage runs), with the idle time in front of every kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/$1 -- python3 $R/tests/gpu_hostpath.py > $R/gpurun_out/$1.log 2>&1
python3 - "$R/gpurun_out/$1" > $R/gpurun_out/$1_ktrace.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: from the last k_pad_init on
last = max(i for i, r in enumerate(rows) if "k_pad_init" in r["Kernel_Name"])
rows = rows[last:]
t0 = int(rows[0]["Start_Timestamp"])
end = {}
for r in rows:
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - end[q]) / 1e6 if q in end else 0.0
    end[q] = e
    print("%-44s q%-3s %9.3f %9.3f ms  dur %8.3f  idle before %7.3f" % (r["Kernel_Name"][:44], q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, gap))
    if (s - t0) / 1e6 > 75: break
PY
rm -rf $R/gpurun_out/$1
