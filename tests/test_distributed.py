"""world_size-2 CPU (gloo) test of the multi-GPU layout: entries shard over ranks, payloads are
gathered onto rank 0 and stitched into one .zip stream that equals the single-process archive.
The compressor is injected (the oracle stands in for the GPU encoder here, which needs a GPU)."""
import io
import os
import sys
import zipfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _common import ROOT, oracle_zip, silesia_mix


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import importlib
    from _common import oracle
    import ctypes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sharding = importlib.import_module("zip-ada_amd.sharding")
    za = importlib.import_module("zip-ada_amd")
    total, entry = 5 * 300000 + 1234, 300000
    ranges = sharding.entry_ranges(total, entry)
    mine = sharding.entries_of_rank(len(ranges), rank, world)
    rounds = max(len(sharding.entries_of_rank(len(ranges), r, world)) for r in range(world))
    gathered = []
    pending = None

    def collect(res):
        if rank == 0:
            for p, m in zip(*res):
                if int(m[3]) >= 0:
                    gathered.append((int(m[3]), bytes(p.numpy()), int(m[0]), int(m[1]), int(m[2])))
    O = oracle()
    for i in range(rounds):
        if i < len(mine):
            off, ln = ranges[mine[i]]
            d = silesia_mix(ln, offset=off)
            out = ctypes.create_string_buffer(ln + 64)
            ol = ctypes.c_uint64(0); crc = ctypes.c_uint32(0); zt = ctypes.c_uint16(0)
            assert O.zo_compress_data(d, ln, 10, out, ln + 64, ctypes.byref(ol), ctypes.byref(crc), ctypes.byref(zt)) == 0
            payload = torch.frombuffer(bytearray(out.raw[:ol.value]), dtype=torch.uint8)
            meta = torch.tensor([crc.value, ln, zt.value, mine[i]], dtype=torch.int64)
            length = ol.value
        else:
            payload = torch.zeros(1, dtype=torch.uint8); meta = torch.tensor([0, 0, 0, -1], dtype=torch.int64); length = 0
        # as bench.py does it: the gather of this entry travels while the next entry is being compressed
        h = sharding.gather_payloads_begin(payload, length, meta, dst=0)
        if pending is not None:
            collect(pending.finish())
        pending = h
    collect(pending.finish())
    if rank == 0:
        gathered.sort()
        zc = za.ZipCreate(None, 10)
        for idx, payload, crc, usize, zt in gathered:
            zc.add_compressed("entry_%04d.bin" % idx, payload, crc, usize, zt)
        q.put(zc.finish())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_and_stitch_equals_single_process_archive():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total, entry = 5 * 300000 + 1234, 300000
    entries, off, i = [], 0, 0
    while off < total:
        ln = min(entry, total - off)
        entries.append(("entry_%04d.bin" % i, silesia_mix(ln, offset=off)))
        off += ln; i += 1
    assert got == oracle_zip(entries, 10)
    zf = zipfile.ZipFile(io.BytesIO(got))
    assert zf.testzip() is None and len(zf.infolist()) == len(entries)
