"""Throughput of many small entries: zada_deflate_batch vs one zada_deflate call per entry."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
za = importlib.import_module("zip-ada_amd")
enc = za.Encoder(0)
mix = za.silesia_mix(64 << 20).tobytes()
for size, cnt in ((16 << 10, 512), (256 << 10, 128), (4 << 20, 16)):
    datas = [mix[(i * size) % (len(mix) - size):][:size] for i in range(cnt)]
    enc.deflate_batch(datas[:8], 10)                       # warm-up (worker contexts, workspaces)
    t0 = time.time(); single = [enc.deflate(d, 10)[0] for d in datas]; t1 = time.time()
    res = enc.deflate_batch(datas, 10); t2 = time.time()
    ok = all(r[1] == s for r, s in zip(res, single))
    print("%4d entries of %7d B: one call each %.1f MB/s, batch %.1f MB/s, identical %s" % (cnt, size, cnt * size / (t1 - t0) / 1e6, cnt * size / (t2 - t1) / 1e6, ok))
