"""GPU script: ONE LZMA_3 stream of 64 KiB (what the PMC pass of DESIGN.md 10 counts)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
from _lzmah import lz_inputs
Z = product(); enc = Z.Encoder(0)
d = lz_inputs()["mix_256k"][:65536]
t = time.time(); rc, z, crc = enc.lzma(d, 18); print("LZMA_3 64 KiB:", len(z), "%.3f s" % (time.time() - t), flush=True)
