"""GPU script: more seeds of test_gpu_atsize.test_soak_seed_bzip2 (random small inputs through zada_bzip2_batch, streams of runs whose block
limits fall inside runs) against the oracle.  usage: python tests/gpu_bz2_soak.py [first_seed [count]]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_gpu_atsize as T
from _common import product
enc = product().Encoder(0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 21
count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
t0 = time.time()
for seed in range(first, first + count):
    T.test_soak_seed_bzip2(enc, seed)
    print("seed", seed, "ok", round(time.time() - t0, 1), "s", flush=True)
