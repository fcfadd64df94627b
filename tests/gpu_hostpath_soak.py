"""GPU script: the host-buffer entry point with the input still arriving (64 MiB and more: the link stage runs piece by piece on what has come,
lz_shard "job.need") against the device-resident entry point on the same bytes -- sizes around the pieces' edges, the three levels that have a
link stage; the streams must be identical and inflate to the input."""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from _common import product
Z = product(); enc = Z.Encoder(0)
M = 1 << 20
sizes = [64 * M, 64 * M + 1, 64 * M + 32767, 64 * M + 32768 + 63, 96 * M - 5, 128 * M, 128 * M + 65, 191 * M + 12345, 257 * M + 3]
host_all = Z.silesia_mix(max(sizes) + 64)
bad = 0
for n in sizes:
    for method in ((10,) if n > 130 * M else (8, 9, 10)):            # Deflate_1 / _2 / _3 (Compression_Method'Pos)
        host = host_all[7:7 + n].copy()                              # (an odd start: nothing is aligned by accident)
        hout = np.zeros(n + 64, dtype=np.uint8)
        rc, ol, crc = enc.deflate_into(host, hout, method)
        t_in = torch.from_numpy(host).cuda(); t_out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        rc2, ol2, crc2 = enc.deflate_device(t_in.data_ptr(), n, t_out.data_ptr(), t_out.numel(), method)
        dev = t_out[:ol2].cpu().numpy()
        same = (rc, ol, crc) == (rc2, ol2, crc2) and np.array_equal(hout[:ol], dev)
        d = zlib.decompressobj(-15); o = d.decompress(bytes(hout[:ol])) + d.flush()
        ok = same and o == bytes(host) and (crc ^ 0xFFFFFFFF) == zlib.crc32(host)
        bad += 0 if ok else 1
        print("%s n %10d method %2d: %d bytes, host path == device path %s, inflates to the input %s" % ("ok       " if ok else "DIFFERENT", n, method, ol, same, o == bytes(host)), flush=True)
        del t_in, t_out
print("host path soak done: different", bad)
sys.exit(1 if bad else 0)
