"""Perf probe: times Deflate_3 on a silesia_mix buffer of the given MiB and prints the phase timing."""
import importlib, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
za = importlib.import_module("zip-ada_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
enc = za.Encoder(0)
d = za.silesia_mix(mib << 20, version=int(os.environ.get("CORPUS", "2"))).tobytes()
for _ in range(reps):
    t0 = time.time(); out, _c = enc.deflate(d, 10); dt = time.time() - t0
    tm = enc.last_timing()
    dev = sum(v for _k, v in tm if not _k.startswith('#'))
    print("%d MiB deflate_3: wall %.3fs (%.1f MB/s) device %.1f ms (%.1f MB/s) ratio %.4f" % (mib, dt, len(d) / dt / 1e6, dev, len(d) / dev / 1e3, len(out) / len(d)))
    print("   ", [(k, round(v, 2)) for k, v in tm])
try:
    print("roundtrip", zlib.decompress(out, -15) == d)
except zlib.error as e:
    print("roundtrip FAILED", e)
