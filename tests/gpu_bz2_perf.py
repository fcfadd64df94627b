"""GPU script: BZip2_3 throughput on silesia_mix with input and output resident in HBM, and the phases' shares."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from _common import product
Z = product()
enc = Z.Encoder(0)
L = Z.load_library()
for knob in ("bz_lists", "bz_pipeline", "bz_small_wg", "bz_batch_melems", "bz_tail_pct", "bz_split", "bz_list_rows", "bz_text_order", "bz_pipe_prio"):
    if os.environ.get(knob.upper()):
        enc.set_knob(knob, int(os.environ[knob.upper()]))
for mib in [int(x) for x in os.environ.get("BZ_MIBS", "64,256").split(",")]:
    n = mib << 20
    h = Z.silesia_mix(n, version=2)                      # the benchmark stream (seed 0x5A1E51A, all five classes), as bench.py's BZip2 leg takes it
    d_in = torch.from_numpy(h).cuda()
    d_out = torch.zeros(n + 4096, dtype=torch.uint8, device="cuda")
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        rc, ol, crc = enc.bzip2_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, 14)
        dt = time.time() - t0
    tim = {}
    for k, v in enc.last_timing():
        tim[k] = tim.get(k, 0.0) + v
    blocks = enc.bz2_last_blocks()
    tac = [0, 0, 0, 0]
    for b in blocks: tac[b[2]] += 1
    print("%d MiB: rc %d, %.1f ms, %.1f MB/s, ratio %.4f, blocks %d, tactics %s" % (mib, rc, dt * 1e3, n / dt / 1e6, ol / n, len(blocks), tac))
    print("   ", {k: round(v, 1) for k, v in tim.items() if not k.startswith("#")}, flush=True)
    if os.environ.get("BZ_DBG"):
        from _bzip2 import _fetch
        info = _fetch(L, enc, "info", np.uint32, 4)
        nsb = int(info[2])
        d = _fetch(L, enc, "dbg", np.uint64, 8 * nsb).reshape(-1, 8)
        res = _fetch(L, enc, "res", np.uint32, 8 * nsb).reshape(-1, 8)
        mh = _fetch(L, enc, "bwt_m", np.uint64, 64)
        print("    BWT rows per round (share of all %d): %s; rounds %d, groups sorted from lists %d" % (int(mh[0]), " ".join("%.3f" % (x / mh[0]) for x in mh[1:]), int(info[1]), int(info[3])))
        st = d[:, 6].astype(np.int64); en = st + d[:, 7].astype(np.int64); t0 = st.min(); span = (en.max() - t0) / 1e5
        grid = np.linspace(t0, en.max(), 41)[:-1]
        conc = [int(((st <= g) & (en > g)).sum()) for g in grid]
        print("    entropy kernel span %.1f ms; workgroups in flight over time: %s" % (span, conc))
        tot = d[:, 7] / 1e5
        print("    last batch: %d sub-blocks, WG time sum %.0f ms, max %.1f, mean %.1f; hist %.0f llhc %.0f cost %.0f chain %.0f; groups mean %.0f max %d" % (
            nsb, tot.sum(), tot.max(), tot.mean(), d[:, 0].sum() / 1e5, d[:, 1].sum() / 1e5, d[:, 2].sum() / 1e5, d[:, 3].sum() / 1e5, res[:, 3].mean(), res[:, 3].max()))
