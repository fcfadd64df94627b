import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mode = sys.argv[1] if len(sys.argv) > 1 else "driver"
if mode == "driver":
    for m in ("create", "small", "big", "tokens", "torchfirst"):
        r = subprocess.run([sys.executable, __file__, m], capture_output=True, text=True)
        print(m, "->", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "", "| err:", r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else "")
    sys.exit(0)
import importlib
za = importlib.import_module("zip-ada_amd")
if mode == "torchfirst":
    import torch; torch.zeros(1).cuda()
enc = za.Encoder(0)
if mode == "small":
    enc.deflate(b"hello hello hello hello" * 100, 10)
if mode == "big":
    enc.deflate(za.silesia_mix(8 << 20).tobytes(), 10)
if mode == "tokens":
    enc.lz77_tokens(b"hello hello hello hello" * 100, 10)
maps = open("/proc/self/maps").read()
libs = sorted(set(l.split()[-1] for l in maps.splitlines() if "amdhip" in l or "hsa-runtime" in l))
import torch
try:
    torch.zeros(1).cuda(); ok = True
except Exception as e:
    ok = repr(e)[:80]
maps = open("/proc/self/maps").read()
libs2 = sorted(set(l.split()[-1] for l in maps.splitlines() if "amdhip" in l or "hsa-runtime" in l))
print(mode, ok, libs, libs2)
