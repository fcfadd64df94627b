"""GPU script: BASELINE config 4's shape -- ONE LZMA_3 stream of C4_MIB MiB (default 1024 = config 4 itself) of the benchmark stream through
zada_lzma (bounded launches, feedback), decoded by liblzma and compared with the input's CRC; C4_ORACLE=1 also codes it with the CPU port and
compares the bytes (1 GiB: several minutes on one core)."""
import json, lzma, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
Z = product(); enc = Z.Encoder(0)
mib = int(os.environ.get("C4_MIB", "1024"))
n = mib << 20
ver = int(os.environ.get("C4_CORPUS", "2"))          # silesia_mix_v2 since round 5 (v1's segments were shifted copies: the stream folded to 2.7 %)
d = Z.silesia_mix(n, seed=0x5A1E51A, version=ver).tobytes()
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
decodes = bool(tot == n and c == zlib.crc32(d) and (crc ^ 0xFFFFFFFF) == c)
print("liblzma decodes it to the input: %s (dictionary %d MiB, end marker met: %s)" % (decodes, ds >> 20, dec.eof), flush=True)
rec = {"workload": "ONE LZMA_3 stream of %d MiB silesia_mix_v%d" % (mib, ver), "seconds": round(dt, 1), "MB/s": round(n / dt / 1e6, 3), "rc": rc, "compression_ratio": round(len(z) / n, 4),
       "liblzma_decodes_it_to_the_input": decodes, "phase_ms": tim}
if os.environ.get("C4_ORACLE") == "1":
    from _lzmah import oracle_lzma
    t1 = time.time()
    o = oracle_lzma(d, 18)
    dto = time.time() - t1
    print("CPU port (one core): %.1f s = %.2f MB/s; payloads equal: %s" % (dto, n / dto / 1e6, o == (rc, z, crc)), flush=True)
    rec.update(cpu_port_one_core_seconds=round(dto, 1), cpu_port_one_core_MBs=round(n / dto / 1e6, 3), equals_cpu_port=bool(o == (rc, z, crc)), gpu_over_cpu_one_core=round(dto / dt, 3))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(out, exist_ok=True)
json.dump(rec, open(os.path.join(out, "config4_lzma3_%dmib_silesia_mix_v%d.json" % (mib, ver)), "w"), indent=1)
