"""GPU script: step time of 256 MiB degenerate inputs (zeros, random, three symbols, a period longer than the window, text only) under the
round-6 defaults and under the round-5 settings of the same knobs -- a check that the filter, the lists and the exact re-parse have no cliff.  Prints one line per input."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
za = importlib.import_module("zip-ada_amd")
n = 256 << 20
rng = np.random.default_rng(1)
inputs = {"zeros": np.zeros(n, np.uint8), "random": rng.integers(0, 256, n, dtype=np.uint8), "sym3": (rng.integers(0, 3, n, dtype=np.uint8) + 65).astype(np.uint8),
          "period_40001": np.resize(rng.integers(0, 256, 40001, dtype=np.uint8), n), "text_only": za.silesia_mix(n, class_mask=1, version=2)}
for name, d in inputs.items():
    d_in = torch.from_numpy(d).cuda(); d_out = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
    row = []
    for knobs in ({}, {"cd_filter": 0, "exact_respec": 0, "budget": 8}):
        enc = za.Encoder(0)
        for k, v in knobs.items(): enc.set_knob(k, v)
        enc.deflate_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, 10)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, ol, crc = enc.deflate_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, 10)
        torch.cuda.synchronize(); row.append(((time.perf_counter() - t0) * 1e3, rc, ol))
        enc.close()
    print("%-14s round-6 defaults %8.1f ms   round-5 settings %8.1f ms   (rc %d, %d bytes; same: %s)" % (name, row[0][0], row[1][0], row[0][1], row[0][2], row[0][1:] == row[1][1:]), flush=True)
