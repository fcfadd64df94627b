#!/bin/bash
# usage (GPU box): tests/prof_round.sh <tag>  -> everything tests/prof_summarize.py needs for profiles/<round>/:
#   gpurun_out/<tag>_bench.json, <tag>_stats/ (kernel trace + stats), <tag>_fetch/, <tag>_write/ (HBM bytes, separate passes),
#   <tag>_sq/ (SQ counters of the same command: what bounds the kernels that are not HBM bound)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
bash $R/tests/prof_bench.sh $tag
rm -rf $R/gpurun_out/${tag}_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/${tag}_sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-checks --no-host-path --no-v1 --no-copy-probe --bzip2-mib 0 --lzma-entries 0 > $R/gpurun_out/${tag}_sq.log 2>&1
ls $R/gpurun_out/${tag}_sq/*/ 2>/dev/null | head
