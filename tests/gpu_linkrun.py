"""GPU script: the link stage with runs of R segments per workgroup of k_prev_links ("link_run"), device-resident Deflate_3 of MIB MiB of the
benchmark stream: per R the phase times, the step, and that the stream is the same bytes."""
import hashlib, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
za = importlib.import_module("zip-ada_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
runs = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,2,4,8,16,32".split(","))]
n = mib << 20
enc = za.Encoder(0)
d_in = torch.from_numpy(za.silesia_mix(n, version=2)).cuda()
d_out = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
ref = None
for R in runs:
    enc.set_knob("link_run", R)
    enc.deflate_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, 10)
    torch.cuda.synchronize()
    best, tim = 1e9, None
    for _ in range(3):
        t0 = time.perf_counter()
        rc, ol, crc = enc.deflate_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, 10)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dt < best:
            best, tim = dt, dict(enc.last_timing())
    h = hashlib.sha256(d_out[:ol].cpu().numpy().tobytes()).hexdigest()
    if ref is None:
        ref = h
    print("link_run %2d: %.2f ms per step (%.0f MB/s)  prev_links %.2f  cross_links %.2f  match %.2f  parse %.2f   same bytes as the first: %s" % (
        R, best * 1e3, n / best / 1e6, tim["prev_links"], tim["cross_links"], tim["match"], tim["parse"], h == ref), flush=True)
