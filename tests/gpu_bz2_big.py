"""GPU script: a BZip2_3 stream beyond 4 GiB (several spans, 64-bit positions): decompresses with libbz2 to the input's CRC."""
import sys, os, time, bz2, zlib, ctypes
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _common import product
Z = product()
enc = Z.Encoder(0)
L = Z.load_library()
mib = int(os.environ.get("BZ_BIG_MIB", "4608"))
n = mib << 20
h = np.zeros(n, np.uint8)
step = 256 << 20
for o in range(0, n, step):
    L.zada_silesia_mix(0, 0x5A1E51A, o, min(step, n - o), h[o:].ctypes.data)
out = np.zeros(n // 2 + (64 << 20), np.uint8)
ol = ctypes.c_uint64(); crc = ctypes.c_uint32(0xFFFFFFFF)
t0 = time.time()
rc = L.zada_bzip2(enc.ctx, 14, h.ctypes.data, n, out.ctypes.data, out.size, ctypes.byref(ol), ctypes.byref(crc), None, None)
dt = time.time() - t0
print("rc", rc, "in", n, "out", ol.value, "ratio %.4f" % (ol.value / n), "%.1f s, %.1f MB/s (host buffers)" % (dt, n / dt / 1e6), "blocks", len(enc.bz2_last_blocks()), flush=True)
assert rc == 0
want = 0
for o in range(0, n, step):
    want = zlib.crc32(h[o:o + step], want)
dec = bz2.BZ2Decompressor()
c, tot, off = 0, 0, 0
view = memoryview(out)[:ol.value]
while off < ol.value:
    ch = dec.decompress(view[off:off + (8 << 20)])
    off += 8 << 20
    c = zlib.crc32(ch, c); tot += len(ch)
print("decompressed", tot, "eof", dec.eof, "crc ok", c == want, "zip crc ok", (crc.value ^ 0xFFFFFFFF) == want, round(time.time() - t0, 1), "s")
assert tot == n and dec.eof and c == want and (crc.value ^ 0xFFFFFFFF) == want
