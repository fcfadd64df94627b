#!/bin/bash
# usage (GPU box): tests/prof_lzma_pmc.sh  -> gpurun_out/lzma_pmc.txt : SQ counters of k_lzma_encode for one 64 KiB LZMA_3 stream (one wave)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/lzma_pmc.txt
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/lzma_p$i -- python3 $R/tests/gpu_lzma_one.py > $R/gpurun_out/lzma_p$i.log 2>&1
  python3 - "$R/gpurun_out/lzma_p$i" >> $R/gpurun_out/lzma_pmc.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
if not f:
    print("no counter file in", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    k = "k_lzma_encode" if "k_lzma_encode" in r["Kernel_Name"] else r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc):
    if "lzma" in k: print(k, " ".join("%s=%.4g" % (c, v) for c, v in sorted(acc[k].items())))
PY
  tail -2 $R/gpurun_out/lzma_p$i.log >> $R/gpurun_out/lzma_pmc.txt
  rm -rf $R/gpurun_out/lzma_p$i
done
cat $R/gpurun_out/lzma_pmc.txt
