"""GPU script (not a pytest file): first run of the LZMA path -- parity with the oracle and time per level."""
import sys, os, time
sys.path.insert(0, os.path.dirname(__file__))
from _common import product
from _lzmah import lz_inputs, oracle_lzma

Z = product(); enc = Z.Encoder(0)
cases = lz_inputs()
names = ["text_0", "text_1", "text_257", "text_4096", "text_32768", "text_65537", "az_77777", "random_66666", "mix_256k", "two_symbols_40k", "zeros_100k"]
bad = 0
for m in (15, 16, 17, 18):
    for nm in names:
        d = cases[nm]
        t = time.time()
        rc, z, crc = enc.lzma(d, m)
        dt = time.time() - t
        orc, oz, ocrc = oracle_lzma(d, m)
        ok = (rc, z, crc) == (orc, oz, ocrc)
        if not ok:
            bad += 1
            k = next((i for i in range(min(len(z or b""), len(oz))) if z[i] != oz[i]), -1)
            print("DIFF", m, nm, rc, orc, len(z or b""), len(oz), "first diff at", k, hex(crc), hex(ocrc))
        print("m%d %-16s %7d -> %7d  %.3fs  %s" % (m, nm, len(d), len(oz), dt, "ok" if ok else "DIFFERENT"), flush=True)
print("different:", bad)
