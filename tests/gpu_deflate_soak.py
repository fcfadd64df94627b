"""GPU script: more seeds of test_gpu_atsize.test_soak_seed_deflate_cuts (random inputs through zada_deflate, one stream cut over 2 .. 8 contexts,
batches of entries) against the oracle.  usage: python tests/gpu_deflate_soak.py [first_seed [count]]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_gpu_atsize as T
from _common import product
enc = product().Encoder(0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 31
count = int(sys.argv[2]) if len(sys.argv) > 2 else 4
t0 = time.time()
for seed in range(first, first + count):
    T.test_soak_seed_deflate_cuts(enc, seed)
    print("seed %d ok (%.0f s)" % (seed, time.time() - t0), flush=True)
print("deflate soak done")
