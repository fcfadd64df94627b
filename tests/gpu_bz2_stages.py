"""GPU script: stages of the BZip2 path against the oracle on sub-blocks of the edge inputs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _common import edge_inputs, product
from _bzip2 import oracle_block, product_stages

Z = product()
enc = Z.Encoder(0)
bad = 0
t0 = time.time()
cases = [(k, v) for k, v in dict(edge_inputs()).items() if 0 < len(v) <= 1_100_000]
rng = np.random.default_rng(5)
cases.append(("zeros_3m", bytes(3_000_000)))
cases.append(("runs", bytes(np.repeat(rng.integers(0, 4, 40000, dtype=np.uint8), rng.integers(1, 700, 40000)).tobytes()[:900000])))
cases.append(("ab_period", (b"ab" * 300000)))
cases.append(("abc_runs4", (b"aaaabbbbcccc" * 50000)))
for name, data in cases:
    n = len(data)
    starts = [0]; lens = [n]
    if n >= 8:
        q = n // 4
        starts += [0, q, 2 * q, 3 * q]; lens += [q, q, q, n - 3 * q]
        starts += [n // 3]; lens += [n - n // 3]
    P1 = product_stages(enc, data, starts, lens, stages=1)
    P = product_stages(enc, data, starts, lens, stages=3)
    P["rle"] = P1["rle"]
    for i, (s, l) in enumerate(zip(starts, lens)):
        o = oracle_block(data[s:s + l], want_bits=True)
        ok = (int(P["rle_n"][i]) == o["info"].rle_n and np.array_equal(P["rle"][i], o["rle"]) and int(P["crc"][i]) == o["info"].block_crc
              and np.array_equal(P["bwt"][i], o["bwt"]) and int(P["bwt_index"][i]) == o["info"].bwt_index
              and int(P["mtf_n"][i]) == o["info"].mtf_n and np.array_equal(P["mtf"][i], o["mtf"]))
        oi = o["info"]
        ok3 = (int(P["res"][i, 0]) == oi.coders and int(P["res"][i, 1]) == oi.max_code_len and int(P["res"][i, 2]) == oi.sample_width
               and np.array_equal(P["selectors"][i], o["selectors"]) and np.array_equal(P["lens"][i], o["lens"]) and int(P["res"][i, 7]) == oi.bits
               and np.array_equal(P["bits"][i], o["bits"]))
        if ok and not ok3:
            bad += 1
            print("ENTROPY MISMATCH", name, i, s, l, "res", P["res"][i].tolist(), (oi.coders, oi.max_code_len, oi.sample_width, oi.selector_count, oi.bits),
                  "sel", np.array_equal(P["selectors"][i], o["selectors"]), "lens", np.array_equal(P["lens"][i], o["lens"]), "bits", np.array_equal(P["bits"][i], o["bits"]))
        if not ok:
            bad += 1
            print("MISMATCH", name, i, s, l, "rle_n", int(P["rle_n"][i]), o["info"].rle_n, "rle", np.array_equal(P["rle"][i], o["rle"]),
                  "crc", hex(int(P["crc"][i])), hex(o["info"].block_crc), "bwt", np.array_equal(P["bwt"][i], o["bwt"]), "idx", int(P["bwt_index"][i]), o["info"].bwt_index, "mtf_n", int(P["mtf_n"][i]), o["info"].mtf_n, "mtf", np.array_equal(P["mtf"][i], o["mtf"]))
    print(name, n, "rounds", int(P["info"][1]), "ok" if not bad else "BAD", round(time.time() - t0, 1), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
