"""Perf / sanity probe on degenerate inputs (64 MiB each): zeros, short period, random bytes, rows."""
import importlib, os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
za = importlib.import_module("zip-ada_amd")
enc = za.Encoder(0)
n = 64 << 20
rng = np.random.default_rng(1)
row = (b"0001234,ABCD,some field,99\n" * 40 + b"0001235,ABCE,some field,98\n")
cases = {
    "zeros": bytes(n), "period 2": b"ab" * (n // 2), "period 37": (bytes(range(37)) * (n // 37 + 1))[:n],
    "random": rng.integers(0, 256, n, dtype=np.uint8).tobytes(), "rows": (row * (n // len(row) + 1))[:n],
    "text": za.silesia_mix(n, class_mask=1).tobytes(), "db": za.silesia_mix(n, class_mask=8).tobytes(),
    "2 symbols": (rng.integers(0, 2, n, dtype=np.uint8) + 65).tobytes(), "4 symbols": (rng.integers(0, 4, n, dtype=np.uint8) + 65).tobytes(),
    "16 symbols": (rng.integers(0, 16, n, dtype=np.uint8) + 65).tobytes(),
    "noisy zeros": bytes(np.where(rng.random(n) < 0.001, rng.integers(1, 256, n), 0).astype(np.uint8)),
    "xml": za.silesia_mix(n, class_mask=2).tobytes(),
}
only = sys.argv[1:] 
if only: cases = {k: v for k, v in cases.items() if k in only}
for name, d in cases.items():
    for _ in range(2):
        t0 = time.time()
        try:
            out, _c = enc.deflate(d, 10); rc = 0
        except za.CompressionInefficient:
            out, rc = b"", 1
        dt = time.time() - t0
    tm = enc.last_timing()
    dev = sum(v for k, v in tm if not k.startswith('#'))
    ok = rc == 1 or zlib.decompress(out, -15) == d
    print("%-10s rc %d device %8.1f ms (%7.1f MB/s) ratio %.4f ok %s  %s" % (name, rc, dev, n / dev / 1e3, len(out) / n, ok,
          [(k, round(v, 1)) for k, v in tm if k in ("prev_links", "cross_links", "match", "parse", "#demand_rounds")]), flush=True)
