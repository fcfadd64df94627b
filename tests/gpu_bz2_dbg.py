import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _common import product
from _bzip2 import product_stages, _fetch
Z = product(); enc = Z.Encoder(0); L = Z.load_library()
n = 4 << 20
h = np.zeros(n, np.uint8); L.zada_silesia_mix(0, 0x5A1E51A, 0, n, h.ctypes.data)
data = h.tobytes()
starts = [0, 0, 221000, 442000, 663000, 885000]; lens = [885000, 221000, 221000, 221000, 222000, 885000]
P = product_stages(enc, data, starts, lens, stages=3)
d = _fetch(L, enc, "dbg", np.uint64, 8 * len(starts)).reshape(-1, 8)
for i in range(len(starts)):
    x = d[i]
    print(lens[i], "res", P["res"][i].tolist(), "clk(100MHz) hist %.1f llhc %.1f cost %.1f chain %.1f total %.1f ms; passes %d rounds %d upd %.1f" % (x[0] / 1e5, x[1] / 1e5, x[2] / 1e5, x[3] / 1e5, x[7] / 1e5, x[4], x[5], x[6] / 1e5))
