"""GPU script: device-resident Deflate_3 of MIB MiB of the benchmark stream, phase times of the best of three runs (no checks: for A/B variants via ZADA_LIB)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
za = importlib.import_module("zip-ada_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib << 20
enc = za.Encoder(0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); enc.set_knob(k, int(v))
d_in = torch.from_numpy(za.silesia_mix(n, version=2)).cuda()
d_out = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
enc.deflate_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, 10)
torch.cuda.synchronize()
best, tim = 1e9, None
for _ in range(3):
    t0 = time.perf_counter()
    rc, ol, crc = enc.deflate_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, 10)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dt < best:
        best, tim = dt, enc.last_timing()
print("%s %s: %.2f ms per step (%.0f MB/s), ratio %.4f" % (os.environ.get("ZADA_LIB", "product"), " ".join(sys.argv[2:]), best * 1e3, n / best / 1e6, ol / n), [(k, round(v, 2)) for k, v in tim], flush=True)
