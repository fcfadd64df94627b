"""GPU script: where one LZMA stream's time goes (variant build with -DZADA_LZ_PROF prints clock counters of entry 0)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
from _lzmah import lz_inputs
Z = product(); enc = Z.Encoder(0)
d = lz_inputs()["mix_256k"][:65536]
for m in (17, 18):
    t = time.time(); rc, z, crc = enc.lzma(d, m); print("method", m, len(z), "%.3f s" % (time.time() - t), flush=True)
