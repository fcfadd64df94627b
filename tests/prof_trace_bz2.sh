#!/bin/bash
# usage (GPU box): tests/prof_trace_bz2.sh <tag>  -> gpurun_out/<tag>_bz_ktrace.txt: the kernel timeline (name, queue, start, end in ms) of tests/gpu_bz2_perf.py
# at 256 MiB -- which launches of the rotation sort wait behind the entropy search's workgroups on the other stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BZ_MIBS=256 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$1_bz -- python3 $R/tests/gpu_bz2_perf.py > $R/gpurun_out/$1_bz.log 2>&1
python3 - "$R/gpurun_out/$1_bz" > $R/gpurun_out/$1_bz_ktrace.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d >= 1.0:
        print("%-44s q%-3s %10.3f %10.3f  %8.3f ms" % (r["Kernel_Name"].replace("zada::", "")[:44], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, d))
PY
rm -rf $R/gpurun_out/$1_bz
