#!/bin/bash
# usage: tests/prof_pmc.sh <outdir> <counters...>   (run on the GPU box; one --pmc pass per invocation)
cd /tmp && export TMPDIR=/tmp
out=$1; shift
rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/tests/gpu_perf.py 64 1 > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
