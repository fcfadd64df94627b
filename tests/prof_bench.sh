#!/bin/bash
# usage (on the GPU box): tests/prof_bench.sh <tag>   -> gpurun_out/<tag>_{stats,fetch,write}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
rm -rf $R/gpurun_out/${tag}_stats $R/gpurun_out/${tag}_fetch $R/gpurun_out/${tag}_write   # (stale runs would be summarised too)
python3 $R/bench.py --steps 2 --warmup 1 > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-checks --no-host-path --no-v1 --no-copy-probe --bzip2-mib 0 --lzma-entries 0 > $R/gpurun_out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-checks --no-host-path --no-v1 --no-copy-probe --bzip2-mib 0 --lzma-entries 0 > $R/gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-checks --no-host-path --no-v1 --no-copy-probe --bzip2-mib 0 --lzma-entries 0 > $R/gpurun_out/${tag}_write.log 2>&1
