"""GPU script for >= 2 GPUs (started by tests/test_multigpu.py under `python -m torch.distributed.run --nproc-per-node N`): ONE Deflate_3 stream
over N ranks on torch.distributed's "nccl" backend -- RCCL over xGMI -- through the product's range calls, with every range's warm-up state
reported WRONG once, so that the re-run of the LZ stage (a collective every rank joins) and the early-posted receive of the chooser's 352 bytes
both run between real peers.  The stream rank 0 puts together (gather_stream_begin: one receive per peer at the peer's byte offset) must be the
oracle's, byte for byte; a second step reuses the warmed-up contexts (no allocation between the posted receive and its wait)."""
import importlib, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == world
za = importlib.import_module("zip-ada_amd")
sh = importlib.import_module("zip-ada_amd.sharding")
enc = za.Encoder(local)


class LiesOnce:
    """The encoder with its first warm-up state falsified: every rank then sees a mismatch for every range behind the first and the
    range's LZ stage runs again from the true state (sharding.deflate_stream_rank step a)."""

    def __init__(self, e):
        self.e, self.lied = e, False

    def __getattr__(self, k):
        return getattr(self.e, k)

    def range_lz(self, entry):
        info = self.e.range_lz(entry)
        if entry is None and not self.lied and rank > 0:
            self.lied = True
            info = dict(info)
            info["warm"] = (info["warm"][0] + 1, info["warm"][1])
        return info


n = world * (3 << 20) + 4321
ranges = sh.stream_ranges(n, world)
lo, ln = ranges[rank]
first, pre, post = sh.range_window(n, lo, ln)
host = za.silesia_mix(pre + ln + post, offset=first)
d_in = torch.from_numpy(host).to(dev)
comm = sh.TorchComm(dev)
for step, e in enumerate((LiesOnce(enc), enc)):
    res = sh.deflate_stream_rank(e, comm, torch, n, ranges, d_in.data_ptr(), 10,
                                 lambda k: torch.empty(k, dtype=torch.int32, device=dev), lambda k: torch.empty(k, dtype=torch.uint8, device=dev))
    assert not res["inefficient"]
    stream = sh.gather_stream_begin(res["payload"], res["spans"], res["total_bits"], dst=0).finish()
    if rank == 0:
        from _common import oracle_deflate
        whole = za.silesia_mix(n).tobytes()
        rc, ref, crc = oracle_deflate(whole, 10)
        got = bytes(stream.cpu().numpy())
        assert rc == 0 and got == ref, "step %d: the stream of %d ranks over RCCL differs from the oracle's" % (step, world)
        assert zlib.decompress(got, -15) == whole and sh.stream_crc(enc.crc32_combine, res["infos"]) == crc
    dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("rccl world-of-%d ok" % world)
