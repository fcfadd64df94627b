"""Quick smoke: sizes given on the command line (KiB), Deflate_3, round trip + oracle compare; flushes as it goes."""
import importlib, os, sys, zlib, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
za = importlib.import_module("zip-ada_amd")
enc = za.Encoder(0)
for kib in [int(a) for a in sys.argv[1:]] or [64]:
    d = za.silesia_mix(kib << 10).tobytes()
    print("n =", len(d), flush=True)
    out, _c = enc.deflate(d, 10)
    print("  ok", len(out), zlib.decompress(out, -15) == d, [(k, round(v, 2)) for k, v in enc.last_timing()], flush=True)
