"""GPU script: ONE LZMA_3 stream of LZ_ONE_KIB KiB (default 2048) of the benchmark stream under the stream's knobs -- the producer's segment size
("lzma_segment": log2 of the positions per segment) and the positions per launch ("lzma_chunk") --, one line per setting, the defaults first and last."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
Z = product()
one = bytes(Z.silesia_mix(int(os.environ.get("LZ_ONE_KIB", "2048")) << 10, version=2))
want = None
for knobs in ({}, {"lzma_segment": 18}, {"lzma_segment": 19}, {"lzma_chunk": 16384}, {"lzma_chunk": 262144}, {"lzma_chunk": 1 << 20}, {}):
    enc = Z.Encoder(0)
    for k, v in knobs.items():
        enc.set_knob(k, v)
    enc.lzma(one[:200000], 18)
    t = time.time(); rc, z, _ = enc.lzma(one, 18); dt = time.time() - t
    want = want or z
    print("%-28s %.2f s = %.4f MB/s, same bytes %s" % (knobs or "defaults", dt, len(one) / dt / 1e6, z == want), {a: round(b, 1) for a, b in enc.last_timing() if not a.startswith("#")}, flush=True)
    enc.close()
