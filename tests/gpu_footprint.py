"""GPU script: HBM a context holds after one device-resident Deflate_3 call of MIB MiB of the benchmark stream (free memory before the context and after the call,
torch's own tensors subtracted), per input byte; optional knob=value arguments."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
za = importlib.import_module("zip-ada_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib << 20
d_in = torch.from_numpy(za.silesia_mix(n, version=2)).cuda()
d_out = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
free0, _ = torch.cuda.mem_get_info()
enc = za.Encoder(0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); enc.set_knob(k, int(v))
rc, ol, crc = enc.deflate_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, 10)
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info()
t = dict(enc.last_timing())
print("context after one Deflate_3 call of %d MiB %s: %.2f GiB = %.1f bytes per input byte (rc %d, ratio %.4f, atoms grown %d, splice slots grown %d)" % (
    mib, " ".join(sys.argv[2:]), (free0 - free1) / 2 ** 30, (free0 - free1) / n, rc, ol / n, t.get("#atoms_grown", 0), t.get("#fix_grown", 0)), flush=True)
