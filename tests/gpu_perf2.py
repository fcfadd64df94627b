"""Perf probe: Deflate_3 of two DIFFERENT 1 GiB inputs alternately on one context (what one call leaves in the workspace is stale for the next)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
za = importlib.import_module("zip-ada_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib << 20
enc = za.Encoder(0)
ins = [torch.from_numpy(za.silesia_mix(n, offset=o)).cuda() for o in (0, 3 << 30)]
out = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
for rep in range(3):
    for k, t in enumerate(ins):
        torch.cuda.synchronize(); t0 = time.time()
        rc, ol, crc = enc.deflate_device(t.data_ptr(), n, out.data_ptr(), n + 4096, 10)
        torch.cuda.synchronize(); dt = time.time() - t0
        tm = dict(enc.last_timing())
        print("input %d: %.1f ms (%.1f MB/s) parse %.1f match %.1f cross %.1f" % (k, dt * 1e3, n / dt / 1e6, tm.get("parse", 0), tm.get("match", 0), tm.get("cross_links", 0)), flush=True)
