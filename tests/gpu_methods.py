"""Perf probe: all methods at one size, then small inputs (per-call latency).  usage: gpu_methods.py [mib]"""
import importlib, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
za = importlib.import_module("zip-ada_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
enc = za.Encoder(0)
d = za.silesia_mix(mib << 20).tobytes()
for name, m in (("Deflate_Fixed", 6), ("Deflate_0", 7), ("Deflate_1", 8), ("Deflate_2", 9), ("Deflate_3", 10)):
    for _ in range(2):
        t0 = time.time(); out, _c = enc.deflate(d, m); dt = time.time() - t0
    dev = sum(v for _k, v in enc.last_timing() if not _k.startswith('#'))
    print("%-14s %d MiB: device %.1f ms (%.0f MB/s) wall %.3f s ratio %.4f roundtrip %s" % (name, mib, dev, len(d) / dev / 1e3, dt, len(out) / len(d), zlib.decompress(out, -15) == d))
for sz in (1 << 10, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22):
    dd = d[:sz]
    enc.deflate(dd, 10)
    t0 = time.time()
    for _ in range(5): out, _c = enc.deflate(dd, 10)
    dt = (time.time() - t0) / 5
    dev = sum(v for _k, v in enc.last_timing() if not _k.startswith('#'))
    print("Deflate_3 %8d B: wall %.3f ms device %.3f ms (%.1f MB/s wall)" % (sz, dt * 1e3, dev, sz / dt / 1e6))
