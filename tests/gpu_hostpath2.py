import sys, time, os
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np, torch, importlib
za = importlib.import_module("zip-ada_amd")
enc = za.Encoder(0)
n = 1 << 30
host = za.silesia_mix(n)
out = np.zeros(n + 64, np.uint8)
d_in = torch.from_numpy(host).cuda(); d_out = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
for _ in range(2):
    t0 = time.perf_counter(); enc.deflate_into(host, out, 10); t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter(); enc.deflate_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, 10); t_dev = time.perf_counter() - t0
print("host path %.1f ms, device path %.1f ms" % (t_host * 1e3, t_dev * 1e3))
a = np.empty_like(host); t0 = time.perf_counter(); np.copyto(a, host); t = time.perf_counter() - t0
print("numpy copy 1 GiB: %.1f ms (%.1f GB/s)" % (t * 1e3, n / t / 1e9))
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
t0 = time.perf_counter(); d_in.copy_(pin, non_blocking=True); torch.cuda.synchronize(); t = time.perf_counter() - t0
print("H2D pinned 1 GiB: %.1f ms (%.1f GB/s)" % (t * 1e3, n / t / 1e9))
t0 = time.perf_counter(); pin.copy_(d_in, non_blocking=True); torch.cuda.synchronize(); t = time.perf_counter() - t0
print("D2H pinned 1 GiB: %.1f ms (%.1f GB/s)" % (t * 1e3, n / t / 1e9))
print("cpus", os.cpu_count())
