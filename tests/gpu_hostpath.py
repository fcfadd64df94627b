"""Wall time of zada_deflate (host buffers in and out, PCIe included) with preallocated buffers."""
import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
za = importlib.import_module("zip-ada_amd")
enc = za.Encoder(0)
for mib in (64, 1024):
    n = mib << 20
    d = za.silesia_mix(n)
    out = np.empty(n + 64, dtype=np.uint8)
    ol = ctypes.c_uint64(0); crc = ctypes.c_uint32(0xFFFFFFFF)
    for rep in range(3):
        crc.value = 0xFFFFFFFF
        t0 = time.perf_counter()
        rc = enc.lib.zada_deflate(enc.ctx, 10, d.ctypes.data, n, out.ctypes.data, n + 64, ctypes.byref(ol), ctypes.byref(crc), None, None)
        dt = time.perf_counter() - t0
    dev = sum(v for k, v in enc.last_timing() if not k.startswith('#'))
    print("%5d MiB: zada_deflate wall %.1f ms (%.0f MB/s), device part %.1f ms, rc %d" % (mib, dt * 1e3, n / dt / 1e6, dev, rc))
