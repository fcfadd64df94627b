"""GPU tests of the range machinery (one stream in shards on one context; one stream over several contexts): the
output must be, bit for bit, the stream of the sequential reference encoder (oracle) whatever the cut."""
import importlib
import threading
import zlib

import numpy as np
import pytest

from _common import METHODS, edge_inputs, oracle_deflate, oracle_tokens, product, silesia_mix

pytestmark = pytest.mark.gpu


def gpu_deflate(enc, d, method):
    """(rc, stream, running CRC register) -- the register is delivered also when the stream is inefficient (the caller Stores
    the entry with it, zip-compress.adb:224-237)."""
    out = bytearray(len(d) + 64)
    rc, ol, crc = enc.deflate_into(d, out, method)
    return rc, bytes(out[:ol]) if rc == 0 else b"", crc


def test_shards_do_not_change_a_byte(encoder):
    """ZADA_SHARD_KIB: the match finder takes the stream in pieces (32 KiB halo, parser state handed over, one atom
    array for the entropy stage).  8 MiB in 1 MiB shards == unsharded == oracle; every edge input in 64 KiB shards."""
    d = silesia_mix(8 << 20)
    try:
        for method in (10, 8, 7, 6):
            rc, ref, crc = oracle_deflate(d, method)
            for kib in (1024, 3 * 64, 1 << 20):
                encoder.set_knob("shard_kib", kib)
                rc2, out, crc2 = gpu_deflate(encoder, d, method)
                assert rc == rc2 == 0 and out == ref and crc == crc2, (method, kib)
        encoder.set_knob("shard_kib", 64)
        for name, dd in edge_inputs().items():
            for method in (10, 9, 7):
                ob = []
                rc, ref, crc = oracle_deflate(dd, method, ob)
                rc2, out, crc2 = gpu_deflate(encoder, dd, method)
                assert rc == rc2 and (rc != 0 or (out == ref and crc == crc2)), (name, method)
                a = oracle_tokens(dd, method)
                b = encoder.lz77_tokens(dd, method)
                assert len(a) == len(b) and (a == b).all(), (name, method)
        # periodic data never re-synchronises: the parser state at a shard boundary is (position, kind) of a run of matches
        for dd in (bytes(700000), b"ab" * 300000, b"abc" * 100000 + b"x" + b"abcd" * 100000):
            for method in (10, 8):
                rc, ref, crc = oracle_deflate(dd, method)
                rc2, out, crc2 = gpu_deflate(encoder, dd, method)
                assert rc == rc2 and out == ref and crc == crc2
    finally:
        encoder.set_knob("shard_kib", 1 << 20)


class ThreadComm:
    """deflate_stream_rank's exchanges between threads of one process (one context per thread on the same GPU)."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world
            self.mail = {}
            self.cv = threading.Condition()

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared.world

    def all_gather_obj(self, obj):
        self.s.slots[self.rank] = obj
        self.s.barrier.wait()
        out = list(self.s.slots)
        self.s.barrier.wait()
        return out

    def bcast_obj(self, obj, src):
        if self.rank == src:
            self.s.slots[src] = obj
        self.s.barrier.wait()
        out = self.s.slots[src]
        self.s.barrier.wait()
        return out

    def all_gather_dev(self, t):
        import torch
        torch.cuda.synchronize()
        return self.all_gather_obj(t)

    def send_bytes(self, b, dst):
        with self.s.cv:
            self.s.mail[(self.rank, dst)] = bytes(b)
            self.s.cv.notify_all()

    def recv_bytes(self, n, src):
        with self.s.cv:
            while (src, self.rank) not in self.s.mail:
                self.s.cv.wait()
            return self.s.mail.pop((src, self.rank))


def deflate_over_contexts(data, world, method, shard_kib=None, ranges=None):
    """The stream `data` compressed by `world` contexts on cuda:0, one thread each, through sharding.deflate_stream_rank
    (ranges: [(lo, n)] instead of the even cut of sharding.stream_ranges)."""
    import torch
    za = product()
    sh = importlib.import_module("zip-ada_amd.sharding")
    n = len(data)
    ranges = ranges or sh.stream_ranges(n, world)
    shared = ThreadComm.Shared(world)
    dev = torch.device("cuda", 0)
    whole = (torch.from_numpy(data) if isinstance(data, np.ndarray) else torch.frombuffer(bytearray(data) if n else bytearray(1), dtype=torch.uint8)).to(dev)
    results, errors = [None] * world, []

    def run(r):
        try:
            torch.cuda.set_device(0)
            enc = za.Encoder(0)
            if shard_kib:
                enc.set_knob("shard_kib", shard_kib)
            win = None
            ptr = 0
            if r < len(ranges):
                lo, ln = ranges[r]
                first, pre, post = sh.range_window(n, lo, ln)
                win = whole[first:first + pre + ln + post].clone()     # this rank's window of the stream (16-byte aligned copy)
                ptr = win.data_ptr()
            res = sh.deflate_stream_rank(enc, ThreadComm(shared, r), torch, n, ranges, ptr, method,
                                         lambda k: torch.empty(k, dtype=torch.int32, device=dev),
                                         lambda k: torch.empty(k, dtype=torch.uint8, device=dev))
            torch.cuda.synchronize()
            res["blocks"] = [tuple(int(x) for x in b) for b in enc.last_blocks()] if r < len(ranges) else []
            results[r] = res
            enc.close()
        except BaseException as e:          # noqa: BLE001 -- a dead thread must not leave the others at a barrier
            errors.append((r, repr(e)))
            shared.barrier.abort()
            with shared.cv:
                shared.cv.notify_all()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    r0 = results[0]
    spans = r0["spans"]
    if r0["inefficient"]:
        return 1, b"", None, results
    out = sh.stitch_stream(torch, [x["payload"] for x in results], spans, r0["total_bits"], dev)
    enc0 = za.Encoder(0)
    crc = sh.stream_crc(enc0.crc32_combine, r0["infos"])
    enc0.close()
    return 0, bytes(out.cpu().numpy()), crc, results


@pytest.mark.parametrize("world", (2, 3, 8))
def test_one_stream_over_several_contexts(world):
    """SURVEY 8e primary mode: ranges of one stream on different contexts, boundary state exchanged, output == oracle.
    Sizes chosen so that range boundaries fall inside flushes, at exact flush boundaries, and in ranges that own no
    flush at all (fewer than 65 536 atoms)."""
    cases = [silesia_mix(6 << 20), silesia_mix((3 << 20) + 4567, class_mask=1), silesia_mix(1 << 20, class_mask=16),
             edge_inputs()["text_rand_text"], edge_inputs()["copies_1500k"], edge_inputs()["fixedlike_mix"], bytes(2 << 20), b"ab" * 600000]
    for d in cases:
        for method in (10, 8, 7, 6):
            ob = []
            rc, ref, crc = oracle_deflate(d, method, ob)
            rc2, out, crc2, results = deflate_over_contexts(d, world, method)
            assert rc == rc2, (len(d), method)
            if rc == 0:
                assert out == ref and crc2 == crc, (len(d), method, world)
                assert zlib.decompress(out, -15) == d
                if method != 6:
                    blocks = [b for res in results for b in res["blocks"]]
                    assert blocks == ob, (len(d), method, world)


def test_ranges_with_shards_and_tiny_tail():
    """Ranges cut into shards themselves, a last range of a few bytes, and a stream smaller than the number of ranks."""
    d = silesia_mix((4 << 20) + 3)
    rc, ref, crc = oracle_deflate(d, 10)
    rc2, out, crc2, _ = deflate_over_contexts(d, 4, 10, shard_kib=256)
    assert rc == rc2 == 0 and out == ref and crc == crc2
    for n in (0, 1, 65536, 65537, 200000):
        d = silesia_mix(n, class_mask=1)
        rc, ref, crc = oracle_deflate(d, 10)
        rc2, out, crc2, _ = deflate_over_contexts(d, 4, 10)
        assert rc == rc2 and (rc != 0 or (out == ref and crc == crc2)), n


def test_bench_multi_rank_path_on_one_gpu(tmp_path):
    """bench.py's N > 1 path (torch.distributed: one process per rank, sharding.TorchComm, gather + stitch on rank 0) with
    every rank on GPU 0 over gloo (BENCH_EMULATE=1): the stitched stream inflates to the input's CRC and the sample
    equals the CPU port's stream."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_EMULATE="1", MASTER_ADDR="127.0.0.1")
    # the driver's plain form: bench.py starts its own ranks when no launcher has (bench.launch_ranks)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1", "--mib", "48", "--cpu-sample-mib", "8"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 3 and res["config"]["rc"] == 0
    assert res["checks"]["stream_inflates_to_input_crc"] is True and res["checks"]["sample_stream_equals_cpu_port"] is True, res["checks"]
    assert res["checks"]["inflated_bytes"] == 3 * (48 << 20)
    # the line says what carried the exchange and where the ranks spent their time (what must be exchanged, zip-create.adb:194-297)
    assert res["backend"] == "gloo" and res["ranks_seen"] == 3
    assert 0 < res["rank_ms_per_step"]["min"] <= res["rank_ms_per_step"]["max"] <= res["ms_per_step"] * 1.05
    assert set(res["exchange_ms_per_step"]) == {"all_gather_state", "boundary_atoms", "carry_chain", "spans", "gather_stitch"}
    assert all(v["max"] >= v["rank0"] >= 0 for v in res["exchange_ms_per_step"].values())


def test_spans_on_one_context(encoder):
    """Streams longer than one pass takes ("span_mib") go through one context span after span: parser state, the atoms behind
    the last whole flush (with their bytes), the chooser's state and the shared output byte are kept from span to span.
    8 MiB in 1 MiB and 3 MiB spans == the oracle, for every method; zeros (the carried atoms span many spans: 4 000 atoms per
    MiB), incompressible stretches (stored pieces of carried atoms), and the device-resident entry point."""
    import torch
    za = product()
    rng = np.random.default_rng(3)
    mix = silesia_mix((8 << 20) + 12345)
    cases = [mix, bytes(5 << 20) + mix[:100000], mix[:1 << 20] + bytes(rng.integers(0, 256, 3 << 20, dtype=np.uint8)) + mix[:(1 << 20) + 7],
             edge_inputs()["copies_1500k"] * 3, silesia_mix(4 << 20, class_mask=1)]
    try:
        for d in cases:
            for method in (10, 8, 7, 6):
                ob = []
                rc, ref, crc = oracle_deflate(d, method, ob)
                for span in (1, 3):
                    encoder.set_knob("span_mib", span)
                    rc2, out, crc2 = gpu_deflate(encoder, d, method)
                    assert rc == rc2 and crc == crc2 and (rc != 0 or out == ref), (len(d), method, span)
        d = mix
        rc, ref, crc = oracle_deflate(d, 10)
        encoder.set_knob("span_mib", 2)
        t_in = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
        t_out = torch.zeros(len(d) + 4096, dtype=torch.uint8, device="cuda")
        rc2, ol, crc2 = encoder.deflate_device(t_in.data_ptr(), len(d), t_out.data_ptr(), len(d) + 4096, 10)
        assert rc2 == 0 and bytes(t_out[:ol].cpu().numpy()) == ref and crc2 == crc
        # an incompressible entry longer than a span is Stored with the CRC-32 of ALL its bytes (zip-compress.adb:224-237):
        # the spans behind the one that says "inefficient" still go through the CRC kernels (host and device entry points)
        import io, zipfile, zlib
        encoder.set_knob("span_mib", 1)
        rnd = bytes(rng.integers(0, 256, (5 << 20) + 333, dtype=np.uint8))
        for method in (10, 6):
            rc2, out, crc2 = gpu_deflate(encoder, rnd, method)
            assert rc2 == 1 and (crc2 ^ 0xFFFFFFFF) == zlib.crc32(rnd), method
        t_in = torch.frombuffer(bytearray(rnd), dtype=torch.uint8).cuda()
        rc2, ol, crc2 = encoder.deflate_device(t_in.data_ptr(), len(rnd), t_out.data_ptr(), len(rnd), 10)
        assert rc2 == 1 and (crc2 ^ 0xFFFFFFFF) == zlib.crc32(rnd)
        zc = za.ZipCreate(encoder, 10)
        zc.add_stream("rnd.bin", rnd)
        zc.add_streams(["a.bin", "mix.txt"], [rnd[:3 << 20], bytes(mix[:200000])])
        zf = zipfile.ZipFile(io.BytesIO(zc.finish()))
        assert zf.testzip() is None and zf.read("rnd.bin") == rnd and zf.getinfo("rnd.bin").compress_type == 0
    finally:
        encoder.set_knob("span_mib", 2048)


def test_rccl_backend_with_a_world_of_one():
    """The multi-GPU protocol's calls on torch.distributed's "nccl" backend (= RCCL) with a world of one rank: tests/rccl_world1.py.
    (Two ranks cannot share a GPU under RCCL, so this is as far as a one-GPU box goes; the protocol with 2 and 3 ranks runs over gloo
    in tests/test_distributed.py and test_bench_multi_rank_path_on_one_gpu.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_world1.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl world-of-one ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
