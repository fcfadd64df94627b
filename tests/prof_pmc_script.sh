#!/bin/bash
# usage (GPU box): tests/prof_pmc_script.sh <tag> <script.py> "<counters pass 1>" ["<counters pass 2>" ...]
#   -> gpurun_out/<tag>_pmc.txt : per kernel, the sum of every counter over its dispatches (one rocprofv3 run of `python3 script.py` per pass;
#   the script reads its parameters from the environment).  FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is to be doubled on gfx950.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; script=$2; shift 2
: > $R/gpurun_out/${tag}_pmc.txt
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/${tag}_p$i -- python3 $R/$script > $R/gpurun_out/${tag}_p$i.log 2>&1
  python3 - "$R/gpurun_out/${tag}_p$i" >> $R/gpurun_out/${tag}_pmc.txt <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
if not fs:
    print("no counter file in", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("zada::", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    nd[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    print(k, " ".join("%s=%.5g(n=%d)" % (c, v, nd[(k, c)]) for c, v in sorted(acc[k].items())))
PY
  tail -3 $R/gpurun_out/${tag}_p$i.log >> $R/gpurun_out/${tag}_pmc.txt
  rm -rf $R/gpurun_out/${tag}_p$i
done
cat $R/gpurun_out/${tag}_pmc.txt
