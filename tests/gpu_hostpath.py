import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, numpy as np
za = importlib.import_module("zip-ada_amd")
enc = za.Encoder(0)
n = 1 << 30
host = za.silesia_mix(n, seed=0x5A1E51A, version=2)
hout = np.zeros(n + 64, dtype=np.uint8)
for i in range(4):
    t = time.perf_counter(); rc, ol, crc = enc.deflate_into(host, hout, 10); dt = time.perf_counter() - t
    tim = [(k, round(v, 2)) for k, v in enc.last_timing() if not k.startswith('#')]
    print("%.1f ms" % (dt * 1e3), tim[:6], "sum %.1f" % sum(v for _, v in tim))
