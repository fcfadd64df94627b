#!/bin/bash
# usage (GPU box): tests/prof_stats.sh <tag> [mib]  -> gpurun_out/<tag>_kstats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$1 -- python3 $R/tests/gpu_perf.py ${2:-64} 2 > $R/gpurun_out/$1.log 2>&1
cp $R/gpurun_out/$1/*/*kernel_stats.csv $R/gpurun_out/$1_kstats.csv
