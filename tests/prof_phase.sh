#!/bin/bash
# usage (GPU box): tests/prof_phase.sh <tag> [mib] [knob=value ...]  -> gpurun_out/<tag>_kstats.txt: per-kernel time of the device-resident Deflate_3 step (tests/gpu_phase.py: four calls)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf $R/gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tests/gpu_phase.py "$@" > $R/gpurun_out/$tag.log 2>&1
python3 - $R/gpurun_out/$tag > $R/gpurun_out/${tag}_kstats.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    print("%-50s calls %5s  avg %10.1f us  min %10.1f  max %10.1f  per call of four %9.3f ms" % (n[:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 4e6))
PY
rm -rf $R/gpurun_out/$tag
