#!/bin/bash
# usage (GPU box): tests/prof_timeline.sh <tag> [mib] [knob=value ...]  -> gpurun_out/<tag>_timeline.txt: every kernel of the LAST device-resident Deflate_3 call of
# tests/gpu_phase.py in launch order (start, duration, idle time of its queue in front of it), from k_match on
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf $R/gpurun_out/$tag
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tests/gpu_phase.py "$@" > $R/gpurun_out/$tag.log 2>&1
python3 - $R/gpurun_out/$tag > $R/gpurun_out/${tag}_timeline.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_pad_init" in r["Kernel_Name"])
rows = rows[last:]
t0 = int(rows[0]["Start_Timestamp"])
end = {}
for r in rows:
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - end[q]) / 1e6 if q in end else 0.0
    end[q] = e
    print("%-40s q%-3s start %9.3f  dur %8.3f  idle before %7.3f" % (r["Kernel_Name"].split("(")[0][-40:], q, (s - t0) / 1e6, (e - s) / 1e6, gap))
PY
rm -rf $R/gpurun_out/$tag
