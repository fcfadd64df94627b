"""One-off (GPU box): a stream longer than one range takes (default 5 GiB) through ONE context span after span (device-resident
entry point), against the same stream as ranges on three contexts and an independent inflater."""
import importlib, os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_ranges import deflate_over_contexts
za = importlib.import_module("zip-ada_amd")
n = (int(sys.argv[1]) << 20) if len(sys.argv) > 1 else 5 << 30
host = za.silesia_mix(n)
enc = za.Encoder(0)
t_in = torch.from_numpy(host).cuda()
t_out = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
t0 = time.time()
rc, ol, crc = enc.deflate_device(t_in.data_ptr(), n, t_out.data_ptr(), n + 4096, 10)
torch.cuda.synchronize()
dt = time.time() - t0
print("one context, spans: rc %d, %d bytes, %.2f s = %.1f MB/s" % (rc, ol, dt, n / dt / 1e6), [(k, round(v, 1)) for k, v in enc.last_timing() if k.startswith("#")])
out = bytes(t_out[:ol].cpu().numpy())
del t_in, t_out
enc.close(); torch.cuda.empty_cache()
dec = zlib.decompressobj(-15); c = 0; tot = 0
for off in range(0, len(out), 1 << 24):
    ch = dec.decompress(out[off:off + (1 << 24)]); c = zlib.crc32(ch, c); tot += len(ch)
ch = dec.flush(); c = zlib.crc32(ch, c); tot += len(ch)
print("inflates to the input:", tot == n and c == zlib.crc32(host) and (crc ^ 0xFFFFFFFF) == c)
if len(sys.argv) > 2:
    rc2, out2, crc2, _ = deflate_over_contexts(host.tobytes(), 3, 10)
    print("three ranges on three contexts give the same stream:", rc2 == rc and out2 == out and crc2 == crc)
