"""One-off (GPU box): a 2 GiB stream as two ranges of 1 GiB on two contexts of the same GPU, each range in 512 MiB shards,
against ONE call on the whole stream (1 GiB shards) and an independent inflater."""
import importlib, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_ranges import deflate_over_contexts
za = importlib.import_module("zip-ada_amd")
n = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 2 << 30
d = za.silesia_mix(n).tobytes()
t0 = time.time()
rc, out, crc, res = deflate_over_contexts(d, 2, 10, shard_kib=512 << 10)
print("two ranges: rc %d, %d bytes, %.1f s" % (rc, len(out), time.time() - t0), [(r["bit_begin"], r["bit_end"]) for r in res])
enc = za.Encoder(0)
t0 = time.time()
one, crc1 = enc.deflate(d, 10)
print("one call: %d bytes, %.1f s" % (len(one), time.time() - t0), "identical:", one == out, "crc equal:", crc == crc1)
dec = zlib.decompressobj(-15); c = 0; tot = 0
for off in range(0, len(out), 1 << 24):
    ch = dec.decompress(out[off:off + (1 << 24)]); c = zlib.crc32(ch, c); tot += len(ch)
ch = dec.flush(); c = zlib.crc32(ch, c); tot += len(ch)
print("inflates to the input:", tot == n and c == zlib.crc32(d) and (crc ^ 0xFFFFFFFF) == c)
