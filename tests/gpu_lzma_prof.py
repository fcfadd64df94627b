"""GPU script: where one LZMA_3 stream's time goes -- run with the variant build `make -C zip-ada_amd/csrc variant NAME=lzprof DEFS=-DZADA_LZ_PROF`
(ZADA_LIB=zip-ada_amd/variants/lzprof.so), which prints the clock counters of entry 0 of every launch: entries of 16 KiB (fresh data at two
places of the benchmark stream), 64 KiB and 1 MiB (mostly long repeats), each as a batch of one = one launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
Z = product(); enc = Z.Encoder(0)
mix = Z.silesia_mix(4 << 20)
for off, kib in ((0, 16), (1 << 20, 16), (0, 64), (0, 1024)):
    d = bytes(mix[off:off + (kib << 10)])
    print("== offset", off, "KiB", kib, flush=True)
    r = enc.lzma_batch([d], 18)
    print("ratio %.3f" % (len(r[0][1]) / len(d)), flush=True)
# the same entries as ONE stream each through zada_lzma (launches of 64 Ki positions: one line per launch): the chain's wave alone, and with three helpers
for waves in (1, 4):
    enc.set_knob("lzma_waves", waves)
    for off, kib in ((0, 16), (0, 64)):
        d = bytes(mix[off:off + (kib << 10)])
        print("== zada_lzma, waves", waves, "KiB", kib, flush=True)
        enc.lzma(d, 18)
enc.set_knob("lzma_waves", 0)
