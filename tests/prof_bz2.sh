#!/bin/bash
# usage (GPU box): tests/prof_bz2.sh <tag> [mib]  -> gpurun_out/<tag>_kstats.csv (rocprofv3 kernel stats of two BZip2_3 runs, tests/gpu_bz2_perf.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export BZ_MIBS=${2:-256}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$1 -- python3 $R/tests/gpu_bz2_perf.py > $R/gpurun_out/$1.log 2>&1
cp $R/gpurun_out/$1/*/*kernel_stats.csv $R/gpurun_out/$1_kstats.csv
