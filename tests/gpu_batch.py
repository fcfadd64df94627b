"""Perf probe (GPU box): a batch of 10 000 entries of 16 KiB through zada_deflate_batch against one call per entry and the
single-thread oracle."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _common import oracle_deflate
za = importlib.import_module("zip-ada_amd")
enc = za.Encoder(0)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
mix = za.silesia_mix(count * size, version=2).tobytes()
datas = [mix[i * size:(i + 1) * size] for i in range(count)]
enc.deflate_batch(datas[:64], 10)
t0 = time.time(); res = enc.deflate_batch(datas, 10); dt = time.time() - t0
print("batch: %d entries of %d bytes in %.3f s = %.1f MB/s; phases %s" % (count, size, dt, count * size / dt / 1e6, [(k, round(v, 2)) for k, v in enc.last_timing()]))
import numpy as np, ctypes
cnt = count
lens = np.full(cnt, size, dtype=np.uint64); caps = lens + 64
arena = np.empty(int(caps.sum()), dtype=np.uint8); offs = np.arange(cnt, dtype=np.uint64) * (size + 64)
outp = (arena.ctypes.data + offs).astype(np.uint64)
base = ctypes.cast(ctypes.c_char_p(mix), ctypes.c_void_p).value
ins = (base + np.arange(cnt, dtype=np.uint64) * size).astype(np.uint64)
ols = np.zeros(cnt, dtype=np.uint64); crcs = np.full(cnt, 0xFFFFFFFF, dtype=np.uint32); rcs = np.zeros(cnt, dtype=np.int32)
t0 = time.time()
enc.lib.zada_deflate_batch(enc.ctx, 10, cnt, ins.ctypes.data, lens.ctypes.data, outp.ctypes.data, caps.ctypes.data, ols.ctypes.data, crcs.ctypes.data, rcs.ctypes.data)
dtc = time.time() - t0
print("C entry point alone: %.3f s = %.1f MB/s" % (dtc, count * size / dtc / 1e6))
t0 = time.time()
for d in datas[:200]:
    enc.deflate(d, 10)
dt1 = (time.time() - t0) / 200
print("one call per entry: %.3f ms each = %.1f MB/s" % (dt1 * 1e3, size / dt1 / 1e6))
t0 = time.time()
for d in datas[:200]:
    oracle_deflate(d, 10)
dto = (time.time() - t0) / 200
print("oracle, one thread: %.3f ms each = %.1f MB/s  -> batch is %.0fx" % (dto * 1e3, size / dto / 1e6, (count * size / dt) / (size / dto)))
ok = all(oracle_deflate(datas[i], 10)[1] == res[i][1] for i in range(0, count, max(1, count // 50)))
print("sampled parity with the oracle:", ok)
