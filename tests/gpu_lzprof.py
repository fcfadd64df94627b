import sys, os
sys.path.insert(0, "/root/repo/tests")
from _common import product
Z = product(); enc = Z.Encoder(0)
mix = Z.silesia_mix(4 << 20)
for off, kib in ((0, 16), (1 << 20, 16), (0, 64), (0, 1024)):
    d = bytes(mix[off:off + (kib << 10)])
    print("== offset", off, "KiB", kib, flush=True)
    r = enc.lzma_batch([d], 18)
    print("ratio %.3f" % (len(r[0][1]) / len(d)), flush=True)
