"""GPU parity tests of the BZip2 path (zada_bzip2*): the product, through the C ABI, against the
oracle's restatement of bzip2-encoding.adb and against the committed digests."""
import bz2
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

from _common import GOLDEN, product
from _bzip2 import bz_inputs, oracle_block, oracle_encode, product_stages

pytestmark = pytest.mark.gpu


def _sub_blocks(n):
    starts, lens = [0], [n]
    if n >= 8:
        q = n // 4
        starts += [0, q, 2 * q, 3 * q, n // 3]
        lens += [q, q, q, n - 3 * q, n - n // 3]
    return starts, lens


def test_every_stage_of_encode_block_bit_exact(encoder):
    """RLE_1, block CRC, BWT + index, MTF / RLE_2 symbols, coders / selectors / code lengths and the block's bits."""
    cases = {k: v for k, v in bz_inputs().items() if 0 < len(v) <= 450000}
    cases["zeros_3m"] = bytes(3_000_000)
    cases["ab_period"] = b"ab" * 300000
    for name, data in cases.items():
        starts, lens = _sub_blocks(len(data))
        P1 = product_stages(encoder, data, starts, lens, stages=1)
        P = product_stages(encoder, data, starts, lens, stages=3)
        for i, (s, l) in enumerate(zip(starts, lens)):
            o = oracle_block(data[s:s + l], want_bits=True)
            oi = o["info"]
            assert int(P["rle_n"][i]) == oi.rle_n and np.array_equal(P1["rle"][i], o["rle"]), (name, i, "RLE_1")
            assert int(P["crc"][i]) == oi.block_crc, (name, i, "CRC")
            assert np.array_equal(P["bwt"][i], o["bwt"]) and int(P["bwt_index"][i]) == oi.bwt_index, (name, i, "BWT")
            assert int(P["mtf_n"][i]) == oi.mtf_n and np.array_equal(P["mtf"][i], o["mtf"]), (name, i, "MTF / RLE_2")
            assert (int(P["res"][i, 0]), int(P["res"][i, 1]), int(P["res"][i, 2])) == (oi.coders, oi.max_code_len, oi.sample_width), (name, i, "coders")
            assert np.array_equal(P["selectors"][i], o["selectors"]) and np.array_equal(P["lens"][i], o["lens"]), (name, i, "selectors / lengths")
            assert int(P["res"][i, 7]) == oi.bits and np.array_equal(P["bits"][i], o["bits"]), (name, i, "bits")


@pytest.mark.parametrize("method", [12, 13, 14])
def test_streams_bit_exact_vs_oracle_and_digests(encoder, method):
    dig = json.load(open(os.path.join(GOLDEN, "bzip2_digests.json")))
    tactics = set()
    for name, data in bz_inputs().items():
        if method != 14 and len(data) > 400000:
            continue
        o, ev = oracle_encode(data, method - 12)
        rc, p, crc = encoder.bzip2(data, method, cap=len(data) * 2 + 100000)
        assert p == o, (name, method)
        assert encoder.bz2_last_blocks() == ev, (name, method)
        assert rc == (1 if len(o) >= len(data) else 0) and (crc ^ 0xFFFFFFFF) == zlib.crc32(data), (name, method)
        want = dig["%s|%d" % (name, method)]
        assert len(p) == want["size"] and hashlib.sha256(p).hexdigest() == want["sha256"], (name, method)
        if len(data):
            assert bz2.decompress(p) == data
        tactics |= {e[2] for e in ev}
    if method == 14:
        assert tactics == {0, 1, 2, 3}


@pytest.mark.parametrize("method", [12, 13])
def test_small_block_options_many_blocks(encoder, method):
    """BZip2_1 / BZip2_2 (100 000 / 400 000-byte blocks, no splitting tactics, one candidate list of the entropy search,
    bzip2-encoding.adb:900-925) on a stream of dozens of blocks, incl. the balanced last two blocks (:1406-1423)."""
    Z = product()
    data = Z.silesia_mix((3 << 20) + 77777).tobytes()
    o, ev = oracle_encode(data, method - 12)
    rc, p, crc = encoder.bzip2(data, method)
    assert rc == 0 and p == o and encoder.bz2_last_blocks() == ev and bz2.decompress(p) == data
    assert len(ev) > (30 if method == 12 else 7) and all(e[2] == 0 for e in ev)


def test_many_blocks_and_batches(encoder):
    """A stream of a few dozen blocks; the same stream when the blocks go through the stages a few at a time."""
    Z = product()
    data = Z.silesia_mix(24 << 20).tobytes()
    o, ev = oracle_encode(data, 2)
    rc, p, crc = encoder.bzip2(data, 14)
    assert rc == 0 and p == o and encoder.bz2_last_blocks() == ev and (crc ^ 0xFFFFFFFF) == zlib.crc32(data)
    encoder.set_knob("bz_batch_melems", 3)
    try:
        rc, p2, _ = encoder.bzip2(data, 14)
    finally:
        encoder.set_knob("bz_batch_melems", 640)
    assert rc == 0 and p2 == o
    encoder.set_knob("bz_span_mib", 24)                 # the block chain walked a stretch of the stream at a time (streams of 4 GiB and more need it)
    try:
        data2 = Z.silesia_mix(70 << 20).tobytes()
        rc, p3, _ = encoder.bzip2(data2, 14)
        blocks3 = encoder.bz2_last_blocks()
    finally:
        encoder.set_knob("bz_span_mib", 1024)
    rc, p4, _ = encoder.bzip2(data2, 14)
    assert rc == 0 and p3 == p4 and blocks3 == encoder.bz2_last_blocks() and bz2.decompress(p3) == data2


def test_device_entry_feedback_abort_and_inefficient(encoder):
    import torch
    Z = product()
    data = Z.silesia_mix(3 << 20)
    d_in = torch.from_numpy(data).cuda()
    d_out = torch.zeros(data.size + 4096, dtype=torch.uint8, device="cuda")
    rc, n, crc = encoder.bzip2_device(d_in.data_ptr(), data.size, d_out.data_ptr(), data.size + 4096, 14)
    o, _ = oracle_encode(data.tobytes(), 2)
    assert rc == 0 and bytes(d_out[:n].cpu().numpy()) == o and (crc ^ 0xFFFFFFFF) == zlib.crc32(data.tobytes())
    seen = []
    rc, p, _ = encoder.bzip2(data.tobytes(), 14, feedback=lambda pct: seen.append(pct) and False)
    assert rc == 0 and p == o and seen[0] == 0 and seen[-1] == 100 and seen == sorted(seen)
    with pytest.raises(Z.UserAbort):
        encoder.bzip2(data.tobytes(), 14, feedback=lambda pct: pct >= 6)
    rc, p, _ = encoder.bzip2(data.tobytes(), 14)                   # the context is usable after an abort
    assert rc == 0 and p == o
    rnd = os.urandom(200000)
    o2, _ = oracle_encode(rnd, 2)
    rc, p, _ = encoder.bzip2(rnd, 14)
    assert rc == 1 and p == o2                                      # compression_ok = False; the stream is still delivered when it fits
    rc, p, _ = encoder.bzip2(rnd, 14, cap=len(rnd))                 # ... and only announced when it does not
    assert rc == 1 and p is None
    with pytest.raises(Z.ZadaError, match="too small"):           # a compressible entry and a buffer it does not fit: an error, not "inefficient"
        encoder.bzip2(data.tobytes(), 14, cap=100000)


def test_zip_archive_with_bzip2_entries(encoder):
    """Zip.Create with a BZip2 method (zip-compress.adb:204-209: format code 12; needed version stays 10, zip-create.adb:131):
    archive bytes == the oracle's Zip.Create restatement; Python's zipfile (libbz2) reads every entry back, incl. stored ones."""
    import io
    import zipfile
    from _common import oracle_zip
    Z = product()
    rng = np.random.default_rng(21)
    mix = Z.silesia_mix(2 << 20)
    entries = [("a/text.txt", bytes(mix[:300000])), ("a/empty", b""), ("b/random.bin", bytes(rng.integers(0, 256, 5000, dtype=np.uint8))),
               ("b/tiny", b"x"), ("c/more.txt", bytes(mix[300000:1500000]))]
    zc = Z.ZipCreate(encoder, 14)
    for name, d in entries:
        zc.add_stream(name, d)
    got = zc.finish()
    assert got == oracle_zip(entries, 14)
    zf = zipfile.ZipFile(io.BytesIO(got))
    assert zf.testzip() is None and [zf.read(i) for i in zf.infolist()] == [d for _, d in entries]
    assert [i.compress_type for i in zf.infolist()] == [12, 0, 0, 0, 12]


def _bzip2_over_contexts(data, world, method=14, ranges=None):
    """The stream compressed by `world` contexts on cuda:0, one thread each, through sharding.bzip2_stream_rank
    (ranges: [(lo, n)] instead of the even cut of sharding.bzip2_ranges)."""
    import importlib
    import threading
    import torch
    from test_ranges import ThreadComm
    Z = product()
    sh = importlib.import_module("zip-ada_amd.sharding")
    n = len(data)
    ranges = ranges or sh.bzip2_ranges(n, world)
    shared = ThreadComm.Shared(world)
    dev = torch.device("cuda", 0)
    whole = (torch.from_numpy(data) if isinstance(data, np.ndarray) else torch.frombuffer(bytearray(data) if n else bytearray(1), dtype=torch.uint8)).to(dev)
    results, errors = [None] * world, []

    def run(r):
        try:
            torch.cuda.set_device(0)
            enc = Z.Encoder(0)
            comm = ThreadComm(shared, r)
            ptr = 0
            if r < len(ranges):
                off, _ = sh.bzip2_window(n, *ranges[r])
                ptr = whole.data_ptr() + off
            results[r] = sh.bzip2_stream_rank(enc, comm, n, ranges, ptr, method, lambda k: torch.zeros(k, dtype=torch.uint8, device=dev))
            torch.cuda.synchronize()
            enc.close()
        except Exception as e:                      # noqa: BLE001
            errors.append((r, repr(e)))
            shared.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    res0 = results[0]
    stream = sh.stitch_stream(torch, [x["payload"] for x in results], [results[k]["spans"][k] if k < len(ranges) else None for k in range(world)], res0["total_bits"], dev)
    blocks = [b for x in results for b in x["blocks"]]
    reg = 0xFFFFFFFF
    for x in results:                                   # the stream's Zip CRC-32 from the ranks' pieces
        if x["crc_raw"] is not None:
            reg = Z.load_library().zada_crc32_combine(reg, x["crc_raw"], x["n"])
    if n and all(x["crc_raw"] is not None for x in results[:len(ranges)]):
        assert (reg ^ 0xFFFFFFFF) == zlib.crc32(data)
    return bytes(stream.cpu().numpy()), blocks, len(ranges)


@pytest.mark.parametrize("world", [2, 3])
def test_one_stream_over_several_contexts(encoder, world):
    """BASELINE config 5 in small: one BZip2_3 stream cut over `world` contexts (block chain handed on, tables gathered, the
    choice replayed, payloads OR-ed at the joints) == the stream of one call == the oracle's."""
    Z = product()
    data = Z.silesia_mix(64 << 20).tobytes()
    rc, one, _ = encoder.bzip2(data, 14)
    want_blocks = encoder.bz2_last_blocks()
    got, blocks, nr = _bzip2_over_contexts(data, world)
    assert nr == world and rc == 0
    assert got == one and blocks == want_blocks
    small = data[:5 << 20]                                          # shorter than two halos per rank: fewer ranges than contexts
    o, ev = oracle_encode(small, 2)
    got, blocks, nr = _bzip2_over_contexts(small, world)
    assert nr == 1 and got == o and blocks == ev


def test_bench_bzip2_multi_rank_path_on_one_gpu():
    """bench.py --method bzip2 at N = 3 (torch.distributed, one process per rank, TorchComm hand-over and gathers, payload
    gather + stitch on rank 0) with every rank on GPU 0 over gloo (BENCH_EMULATE=1): the stitched stream decompresses to the input."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_EMULATE="1", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):                     # the plain form: bench.py starts its own ranks (bench.launch_ranks)
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--method", "bzip2", "--steps", "1", "--warmup", "1", "--mib", "40"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 3 and "3 block ranges" in res["config"]["workload"]
    assert res["checks"] == {"stream_decompresses": True, "rank0_window_crc": True}, res["checks"]


def test_batch_of_small_entries_is_bit_exact(encoder):
    """zada_bzip2_batch: entries of one block each through one launch sequence == one call per entry == the oracle; entries
    longer than a block, empty and incompressible ones in the same call; several launch sequences (bz_batch_mib)."""
    Z = product()
    rng = np.random.default_rng(33)
    mix = Z.silesia_mix(6 << 20)
    datas, off = [], 0
    for k in range(60):
        ln = int(rng.integers(0, 120000)) if k % 7 else int(rng.integers(0, 300))
        datas.append(bytes(mix[off:off + ln]) if k % 9 else bytes(rng.integers(0, 256, ln, dtype=np.uint8)))
        off = (off + ln) % (len(mix) - 130000)
    datas.append(b"")
    datas.append(bytes(mix[:800000]))                       # more than one block's worth: taken by itself
    datas.append(bytes(mix[1000:720000]))                   # just below the limit
    for method, mib in ((14, 256), (12, 256), (14, 1)):
        sel = datas if method == 14 else datas[:40]
        encoder.set_knob("bz_batch_mib", mib)
        try:
            res = encoder.bzip2_batch(sel, method)
        finally:
            encoder.set_knob("bz_batch_mib", 256)
        for d, (rc, p, crc) in zip(sel, res):
            o, _ = oracle_encode(d, method - 12)
            assert p == o and rc == (1 if len(o) >= len(d) else 0) and (crc ^ 0xFFFFFFFF) == zlib.crc32(d), (method, mib, len(d))
    # the container writer takes the batch path for BZip2 methods too
    from _common import oracle_zip
    entries = [("f%02d" % k, d) for k, d in enumerate(datas[:25])]
    zc = Z.ZipCreate(encoder, 14)
    zc.add_streams([n for n, _ in entries], [d for _, d in entries])
    assert zc.finish() == oracle_zip(entries, 14)


def test_batch_with_more_sub_blocks_than_a_list_entry_can_name(encoder):
    """A list entry of the rotation sort's late rounds packs its sub-block into 19 bits (GlEntry::sb_rows, zada_bz2.hip): 120 000 small
    entries in ONE launch sequence make 600 000 sub-blocks (BZip2_3: the single block and its four quarters each, bzip2-encoding.adb:
    1237-1253) -- beyond that, the batch keeps sweeping instead of listing (bz_transform).  Every stream is decoded by libbz2, a sample is
    compared with the oracle byte for byte."""
    Z = product()
    rng = np.random.default_rng(524288)
    mix = Z.silesia_mix(4 << 20, class_mask=1)
    offs = rng.integers(0, len(mix) - 400, 120000)
    lens = rng.integers(250, 351, 120000)
    datas = [bytes(mix[int(o):int(o) + int(n)]) for o, n in zip(offs, lens)]
    res = encoder.bzip2_batch(datas, 14)
    assert len(res) == len(datas)
    for k in range(0, len(datas), 500):
        o, _ = oracle_encode(datas[k], 2)
        assert res[k][1] == o, k
    for d, (rc, p, crc) in zip(datas, res):
        assert bz2.decompress(p) == d and (crc ^ 0xFFFFFFFF) == zlib.crc32(d)


def test_group_lists_of_the_rotation_sort(encoder):
    """Round 3: the late rounds of the rotation sort run from lists of the unsorted groups (a thread per group of up to 8 rows,
    sixteen lanes up to 16, a wave up to 64, a workgroup up to 8 192; classes in two arrays that take turns, zada_bz2.hip "Late rounds: group lists").  The
    knob `bz_lists` moves the round from which sub-blocks may leave the sweeps: whatever it is, the stream is the oracle's -- on
    periodic data whose period straddles the team sizes (rotations equal to the end: groups that never come apart and go off the
    lists with h >= n), few-symbol data (groups of every size), repeated chunks (long repeats: pairs that take ten more doublings)
    and the mix; with and without the two-batch pipeline and the small entropy workgroups."""
    Z = product()
    rng = np.random.default_rng(303)
    mix = Z.silesia_mix(2 << 20)
    cases = []
    for per in (1, 2, 7, 8, 9, 16, 17, 64, 65, 1000):
        pat = bytes(rng.integers(0, 256, per, dtype=np.uint8))
        n = int(rng.integers(3000, 90000))
        cases.append((pat * (n // per + 1))[:n])
    cases.append(bytes(rng.integers(0, 3, 400000, dtype=np.uint8)))
    chunk = bytes(mix[12345:12345 + 41000])
    cases.append((chunk * 30)[:1100000])
    parts = [bytes(mix[int(rng.integers(0, 1 << 20)):][:int(rng.integers(100, 30000))]) for _ in range(12)]
    cases.append(b"".join(parts + parts[::-1] + parts))
    cases.append(bytes(mix[:1500000]))
    want = [oracle_encode(d, 2) for d in cases]
    try:
        for lists, pipeline, small_wg, list_rows, text_order in ((16, 1, 1, 0, 1), (8, 1, 1, 0, 0), (1024, 0, 1, 0, 1), (0, 0, 0, 0, 1), (16, 1, 0, 64, 1), (8, 0, 1, 300, 1),
                                                                  (64, 1, 1, 8, 0), (16, 0, 1, 0, 0)):
            encoder.set_knob("bz_lists", lists); encoder.set_knob("bz_pipeline", pipeline); encoder.set_knob("bz_small_wg", small_wg)
            encoder.set_knob("bz_list_rows", list_rows)                   # (largest group a listed sub-block may have: 0 = 8 192, a workgroup's sort)
            encoder.set_knob("bz_text_order", text_order)                 # (the thread-per-group list in the order of the text, or of the sorted rows)
            encoder.set_knob("bz_split", small_wg)                        # (the long sub-blocks' search as four workgroups, or as one)
            for d, (o, ev) in zip(cases, want):
                rc, p, crc = encoder.bzip2(d, 14, cap=len(d) * 2 + 4096)
                assert p == o and encoder.bz2_last_blocks() == ev and (crc ^ 0xFFFFFFFF) == zlib.crc32(d), (lists, pipeline, small_wg, list_rows, text_order, len(d))
    finally:
        encoder.set_knob("bz_lists", -1); encoder.set_knob("bz_pipeline", 1); encoder.set_knob("bz_small_wg", 1); encoder.set_knob("bz_split", 1)
        encoder.set_knob("bz_list_rows", 0); encoder.set_knob("bz_text_order", 1)
