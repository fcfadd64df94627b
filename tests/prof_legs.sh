#!/bin/bash
# usage (GPU box): tests/prof_legs.sh <tag>  -> gpurun_out/<tag>_legs/: what profiles/<round>/ keeps of the secondary legs --
#   lzma_kernel_stats.csv, bzip2_kernel_stats.csv (rocprofv3 --kernel-trace --stats of tests/gpu_lzma_perf.py: two batches of 4 096 entries of 16 KiB
#   and one 4 MiB stream; tests/gpu_bz2_perf.py: two runs of 256 MiB) and pmc_secondary_legs.json (separate FETCH_SIZE / WRITE_SIZE passes of the same
#   scripts, batch only: HBM bytes of k_lzma_encode per launch and of k_bz_entropy per run; KB, FETCH_SIZE to be doubled on gfx950)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
O=$R/gpurun_out/${tag}_legs
rm -rf $O; mkdir -p $O
LZ_ONE_KIB=4096 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lz_stats -- python3 $R/tests/gpu_lzma_perf.py > $O/lzma.log 2>&1
BZ_MIBS=256 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bz_stats -- python3 $R/tests/gpu_bz2_perf.py > $O/bzip2.log 2>&1
cp $O/lz_stats/*/*kernel_stats.csv $O/lzma_kernel_stats.csv
cp $O/bz_stats/*/*kernel_stats.csv $O/bzip2_kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  LZ_ONE_KIB=0 LZ_WARM=0 rocprofv3 --pmc $ctr --output-format csv -d $O/lz_$ctr -- python3 $R/tests/gpu_lzma_perf.py > $O/lz_$ctr.log 2>&1
  BZ_MIBS=256 rocprofv3 --pmc $ctr --output-format csv -d $O/bz_$ctr -- python3 $R/tests/gpu_bz2_perf.py > $O/bz_$ctr.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
def total(pat, kernel, ctr):
    s, n = 0.0, 0
    for f in glob.glob(O + "/" + pat + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and kernel in r["Kernel_Name"]:
                s += float(r["Counter_Value"]); n += 1
    return s, n
out = {}
f, nf = total("lz_FETCH_SIZE", "k_lzma_encode", "FETCH_SIZE"); w, nw = total("lz_WRITE_SIZE", "k_lzma_encode", "WRITE_SIZE")
out["k_lzma_encode"] = {"fetch_kb": f, "write_kb": w, "launches": nf, "dispatches": nf, "workload": "two batches of 4 096 entries of 16 KiB, LZMA_3 (tests/gpu_lzma_perf.py), per launch"}
f, nf = total("bz_FETCH_SIZE", "k_bz_entropy", "FETCH_SIZE"); w, nw = total("bz_WRITE_SIZE", "k_bz_entropy", "WRITE_SIZE")
out["k_bz_entropy"] = {"fetch_kb": f, "write_kb": w, "launches": 2, "dispatches": nf, "workload": "two BZip2_3 runs of 256 MiB (tests/gpu_bz2_perf.py), all launches of the kernel, per run"}
for k in ("k_bt4_walk(", "k_bt4_walk_lds<true>", "k_bt4_walk_lds<false>", "k_rs_scatter"):
    f, nf = total("lz_FETCH_SIZE", k, "FETCH_SIZE"); w, nw = total("lz_WRITE_SIZE", k, "WRITE_SIZE")
    out[k.rstrip("(")] = {"fetch_kb": f, "write_kb": w, "launches": 2, "dispatches": nf, "workload": "the BT4 producer of the same two LZMA_3 batches, per batch"}
out["_note"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; values in KB summed over the kernel's dispatches; FETCH_SIZE is to be doubled on gfx950 (MI355X_MICROARCH.md, HBM section)"
json.dump(out, open(O + "/pmc_secondary_legs.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/lz_stats $O/bz_stats $O/lz_FETCH_SIZE $O/lz_WRITE_SIZE $O/bz_FETCH_SIZE $O/bz_WRITE_SIZE
ls $O
