"""GPU script: soak of the link stage in runs of segments (round 5) -- random inputs (both corpus versions, every class mix, periodic and constant stretches spliced in),
random run lengths, shard sizes and methods, device path and host path, each stream against the oracle's.  SOAK_SEED, SOAK_SECONDS."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _common import product, oracle_deflate, silesia_mix
Z = product(); enc = Z.Encoder(0)
seed = int(os.environ.get("SOAK_SEED", "1")); budget = float(os.environ.get("SOAK_SECONDS", "150"))
rng = np.random.default_rng(seed)
t0 = time.time(); cases = bad = 0
while time.time() - t0 < budget:
    n = int(rng.integers(40000, 6 << 20))
    parts, left = [], n
    while left > 0:
        k = int(min(left, rng.integers(1000, 2 << 20)))
        kind = rng.integers(0, 10)
        if kind < 7:
            parts.append(silesia_mix(k, class_mask=int(rng.integers(1, 32)), offset=int(rng.integers(0, 1 << 22)), seed=int(rng.integers(1, 1 << 30)), version=int(rng.integers(1, 3))))
        elif kind == 7:
            unit = bytes(rng.integers(0, 256, int(rng.integers(1, 40000)), dtype=np.uint8)); parts.append((unit * (k // len(unit) + 1))[:k])
        elif kind == 8:
            parts.append(bytes([int(rng.integers(0, 256))]) * k)
        else:
            parts.append(bytes(rng.integers(0, int(rng.integers(2, 256)), k, dtype=np.uint8)))
        left -= k
    d = b"".join(parts)
    method = int(rng.choice([10, 10, 9, 8, 7]))
    run = int(rng.choice([0, 1, 2, 4, 8, 16, 32]))
    shard = int(rng.choice([1 << 20, 1 << 20, 512, 1024, 4096]))
    budget_k = int(rng.choice([-1, -1, 1, 2, 3, 8]))              # round 6: small first-pass budgets (many guesses, many demand rounds) and both ways of parsing the flagged chunks again
    exact = int(rng.choice([32768, 32768, 1 << 30, 64, 0]))   # (lists up to this many chunks: one wave per chunk with the exact search inside)
    cdf = int(rng.choice([1, 1, 1, 0]))                           # (the cross-segment continuation with the Bloom filter and its second / third pass, or in one pass)
    enc.set_knob("link_run", run); enc.set_knob("shard_kib", shard); enc.set_knob("budget", budget_k); enc.set_knob("exact_respec", exact); enc.set_knob("cd_filter", cdf)
    rc, ref, crc = oracle_deflate(d, method)
    try:
        out, crc2 = enc.deflate(d, method); rc2 = 0
    except Z.CompressionInefficient:
        rc2, out, crc2 = 1, b"", crc
    ok = rc == rc2 and (rc != 0 or (out == ref and crc == crc2))
    cases += 1
    if not ok:
        bad += 1
        print("MISMATCH seed %d case %d: n %d method %d link_run %d shard_kib %d budget %d exact_respec %d cd_filter %d" % (seed, cases, n, method, run, shard, budget_k, exact, cdf), flush=True)
print("link-run soak seed %d: %d cases in %.0f s, %d mismatches" % (seed, cases, time.time() - t0, bad), flush=True)
sys.exit(1 if bad else 0)
