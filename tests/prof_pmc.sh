#!/bin/bash
# usage (GPU box): tests/prof_pmc.sh <tag> <mib> "<counters pass 1>" ["<counters pass 2>" ...]
#   -> gpurun_out/<tag>_pmc.txt : per kernel, the sum of every counter over its dispatches (one rocprofv3 run per pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; mib=$2; shift 2
: > $R/gpurun_out/${tag}_pmc.txt
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/${tag}_p$i -- python3 $R/tests/gpu_perf.py $mib 1 > $R/gpurun_out/${tag}_p$i.log 2>&1
  python3 - "$R/gpurun_out/${tag}_p$i" >> $R/gpurun_out/${tag}_pmc.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    nd[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    print(k, " ".join("%s=%.4g(n=%d)" % (c, v, nd[(k, c)]) for c, v in sorted(acc[k].items())))
PY
  rm -rf $R/gpurun_out/${tag}_p$i
done
