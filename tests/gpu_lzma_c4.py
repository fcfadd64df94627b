"""GPU script: BASELINE config 4's shape -- ONE LZMA_3 stream of C4_MIB MiB (default 1024 = config 4 itself) of the benchmark stream through
zada_lzma (bounded launches, feedback), decoded by liblzma and compared with the input's CRC; C4_ORACLE=1 also codes it with the CPU port and
compares the bytes (1 GiB: several minutes on one core)."""
import lzma, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
Z = product(); enc = Z.Encoder(0)
mib = int(os.environ.get("C4_MIB", "1024"))
n = mib << 20
d = Z.silesia_mix(n, seed=0x5A1E51A).tobytes()
seen = []
t0 = time.time()
def fb(pct):
    if not seen or pct != seen[-1]:
        seen.append(pct)
        if pct % 10 == 0:
            print("  %3d %%  %.0f s" % (pct, time.time() - t0), flush=True)
    return False
rc, z, crc = enc.lzma(d, 18, feedback=fb)
dt = time.time() - t0
tim = {k: round(v, 1) for k, v in enc.last_timing() if not k.startswith("#")}
print("config 4 shape: one LZMA_3 stream of %d MiB: rc %d, %.1f s = %.3f MB/s, ratio %.4f, feedback calls %d" % (mib, rc, dt, n / dt / 1e6, len(z) / n, len(seen)), tim, flush=True)
ds = int.from_bytes(z[5:9], "little")
dec = lzma.LZMADecompressor(format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA1, "dict_size": ds, "lc": 3, "lp": 0, "pb": 2}])
c, tot, off = 0, 0, 9
while off < len(z):
    ch = dec.decompress(z[off:off + (4 << 20)])
    off += 4 << 20
    c = zlib.crc32(ch, c); tot += len(ch)
print("liblzma decodes it to the input: %s (dictionary %d MiB, end marker met: %s)" % (tot == n and c == zlib.crc32(d) and (crc ^ 0xFFFFFFFF) == c, ds >> 20, dec.eof), flush=True)
if os.environ.get("C4_ORACLE") == "1":
    from _lzmah import oracle_lzma
    t1 = time.time()
    o = oracle_lzma(d, 18)
    print("CPU port (one core): %.1f s = %.2f MB/s; payloads equal: %s" % (time.time() - t1, n / (time.time() - t1) / 1e6, o == (rc, z, crc)), flush=True)
