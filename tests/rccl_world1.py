"""GPU script (run by tests/test_ranges.py::test_rccl_backend_with_a_world_of_one): the exchanges of the multi-GPU path on torch.distributed's
"nccl" backend -- RCCL -- with a world of ONE rank, which is all a one-GPU box can hold.  It proves little about xGMI, but it does prove that
every call the protocol makes (the fixed-size int64 tensor all_gathers of the parser states and bit positions, the padded one of the BZip2 tables,
all_gather of device tensors, the asynchronous gathers of the payloads and of the stream's edge bytes, all_reduce, barrier -- NOT the point-to-point
calls: with one rank there is no peer, so the chooser chain's irecv / send and the payload receives at their offsets first run in
tests/test_multigpu.py, which turns itself on where two GPUs are visible) is accepted by the RCCL backend with the dtypes and devices used, and that one range through TorchComm gives the oracle's stream."""
import importlib, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist
from _common import oracle_deflate
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
za = importlib.import_module("zip-ada_amd")
sh = importlib.import_module("zip-ada_amd.sharding")
enc = za.Encoder(0)
comm = sh.TorchComm(dev)
import numpy as np
assert comm.all_gather_i64([1, 5, 1 << 40, 2]) == [[1, 5, 1 << 40, 2]]
tabs = comm.all_gather_i64_var(np.array([7, 0xFFFFFFFFFFFFFFFF, 3], dtype=np.uint64))
assert len(tabs) == 1 and tabs[0].tolist() == [7, 0xFFFFFFFFFFFFFFFF, 3] and comm.all_gather_i64_var(np.zeros(0, np.uint64))[0].size == 0
t = torch.arange(1000, dtype=torch.int32, device=dev)
g = comm.all_gather_dev(t)
assert len(g) == 1 and bool((g[0] == t).all())
n = (3 << 20) + 777
d = za.silesia_mix(n)
d_in = torch.from_numpy(d).to(dev)
ranges = sh.stream_ranges(n, 1)
res = sh.deflate_stream_rank(enc, comm, torch, n, ranges, d_in.data_ptr(), 10,
                             lambda k: torch.empty(k, dtype=torch.int32, device=dev), lambda k: torch.empty(k, dtype=torch.uint8, device=dev))
h = sh.gather_stream_begin(res["payload"], res["spans"], res["total_bits"], dst=0)
stream = bytes(h.finish().cpu().numpy())
h2 = sh.gather_payloads_begin(res["payload"], res["nbytes"], torch.zeros(1, dtype=torch.int64), dst=0)     # the entries' gather (archives of many entries)
payloads, _ = h2.finish()
assert bytes(payloads[0].cpu().numpy()) == stream
rc, ref, crc = oracle_deflate(d.tobytes(), 10)
assert rc == 0 and stream == ref and zlib.decompress(stream, -15) == d.tobytes()
assert sh.stream_crc(enc.crc32_combine, res["infos"]) == crc
x = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(x, op=dist.ReduceOp.MAX)
dist.barrier()
dist.destroy_process_group()
print("rccl world-of-one ok", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
