"""At-size checks and soak seeds, under pytest so that the driver's `-m gpu` run sees them (round 2 kept them as one-off scripts).

At BASELINE's sizes the oracle is out of reach (11 MB/s), so the checks are the size-independent ones: the stream of ONE call equals
the stream of the same input cut over several contexts (two different cuts of the sequential encoder's state), and an independent
decoder (zlib / libbz2 / zipfile) gives the input back.  The soak seeds compare random cases with the oracle byte for byte."""
import bz2
import io
import os
import zipfile
import zlib

import numpy as np
import pytest

from _common import oracle_deflate, product
from test_ranges import deflate_over_contexts

pytestmark = pytest.mark.gpu


@pytest.fixture()
def encoder():
    enc = product().Encoder(0)
    yield enc
    enc.close()


def _inflate_crc(stream):
    dec = zlib.decompressobj(-15)
    crc, tot = 0, 0
    view = memoryview(stream)
    for off in range(0, len(view), 1 << 24):
        ch = dec.decompress(view[off:off + (1 << 24)])
        crc = zlib.crc32(ch, crc); tot += len(ch)
    ch = dec.flush()
    return zlib.crc32(ch, crc), tot + len(ch), dec.eof


def test_deflate_stream_of_two_and_a_half_gib(encoder):
    """A 2.5 GiB Deflate_3 stream in ONE call on one context (device buffers: two spans of the default 2 GiB, the first in two 1 GiB
    shards) == the same stream as two ranges on two contexts (512 MiB shards, boundary state exchanged) == the input after zlib's
    inflate, with the CRC-32 the calls deliver."""
    import torch
    za = product()
    n = (5 << 29) + 12345
    host = za.silesia_mix(n)
    want_crc = zlib.crc32(host)
    t_in = torch.from_numpy(host).cuda()
    t_out = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
    rc, ol, crc = encoder.deflate_device(t_in.data_ptr(), n, t_out.data_ptr(), n + 4096, 10)
    assert rc == 0 and (crc ^ 0xFFFFFFFF) == want_crc
    one = bytes(t_out[:ol].cpu().numpy())
    del t_in, t_out
    torch.cuda.empty_cache()
    c, tot, eof = _inflate_crc(one)
    assert tot == n and c == want_crc and eof
    encoder.close()                                  # (its workspace: about 100 GiB of HBM)
    rc2, two, crc2, res = deflate_over_contexts(host, 2, 10, shard_kib=512 << 10)
    assert rc2 == 0 and crc2 == crc and two == one, (len(two), len(one))
    assert [r["bit_begin"] for r in res][1] == res[0]["bit_end"]


def test_zip64_entry_beyond_four_gib(encoder):
    """Zip.Create promotes an archive to Zip_64 when an entry does not fit 32 bits (zip-create.adb:161-179, local header extension
    :237-251, central extension and end records :682-752).  A 4.2 GiB entry through zada_compress_data (host buffers, span after
    span) and ZipCreate, read back by Python's zipfile, entries before and behind it included."""
    za = product()
    n = (4 << 30) + (200 << 20) + 77
    big = za.silesia_mix(n)
    zc = za.ZipCreate(encoder, za.Method.Deflate_1)
    first = za.silesia_mix(100000, class_mask=1).tobytes()
    zc.add_stream("small/first.txt", first)
    zc.add_stream("big.bin", big)
    zc.add_stream("small/last.txt", b"the end\n")
    arc = zc.finish()
    assert zc.zip64
    zf = zipfile.ZipFile(io.BytesIO(arc))
    infos = zf.infolist()
    assert [i.filename for i in infos] == ["small/first.txt", "big.bin", "small/last.txt"]
    assert infos[1].file_size == n and infos[1].compress_type == 8 and infos[1].compress_size < n // 2
    crc = 0
    with zf.open("big.bin") as f:
        while True:
            ch = f.read(1 << 24)
            if not ch:
                break
            crc = zlib.crc32(ch, crc)
    assert crc == zlib.crc32(big) == infos[1].CRC
    assert zf.read("small/first.txt") == first and zf.read("small/last.txt") == b"the end\n"


def test_bzip2_stream_across_a_span_boundary(encoder):
    """A 1.2 GiB BZip2_3 stream: the block chain is walked a span (1 GiB) at a time, a span starting where a block starts
    (bzip2-encoding.adb:1144-1209).  libbz2 decodes the stream to the input; the Zip CRC-32 is the input's."""
    import torch
    za = product()
    n = (1 << 30) + (200 << 20) + 4321
    host = za.silesia_mix(n)
    t_in = torch.from_numpy(host).cuda()
    t_out = torch.zeros(n // 2 + (64 << 20), dtype=torch.uint8, device="cuda")
    rc, ol, crc = encoder.bzip2_device(t_in.data_ptr(), n, t_out.data_ptr(), t_out.numel(), 14)
    assert rc == 0 and (crc ^ 0xFFFFFFFF) == zlib.crc32(host)
    stream = bytes(t_out[:ol].cpu().numpy())
    blocks = encoder.bz2_last_blocks()
    assert blocks[0][0] == 0 and sum(b[1] for b in blocks) == n and all(blocks[i][0] + blocks[i][1] == blocks[i + 1][0] for i in range(len(blocks) - 1))
    dec = bz2.BZ2Decompressor()
    c, tot, off = 0, 0, 0
    while off < len(stream):
        ch = dec.decompress(stream[off:off + (8 << 20)])
        off += 8 << 20
        c = zlib.crc32(ch, c); tot += len(ch)
    assert tot == n and dec.eof and c == zlib.crc32(host)


def test_c3_rank_shape_two_gib_range_between_its_neighbours(encoder):
    """BASELINE config 3 gives each of 8 GPUs a 2 GiB range of ONE 16 GiB Deflate_3 stream (bench.py --gpus 8: sharding.stream_ranges,
    32 KiB in front of the range and 1 MiB behind it resident as well).  No 8-GPU node has run it yet, so ONE rank's share runs here at
    its size, on the one GPU there is: a 2 GiB range cut from the middle of a longer stream, the boundary state it needs (parser
    states, edge atoms, the 352-byte chooser state) produced by contexts on the 64 MiB ranges on either side of it, all through
    sharding.deflate_stream_rank -- the code a rank of bench.py runs.  The stitched stream == the stream of ONE call on the whole input
    (another cut of the sequential encoder's state) == the input after zlib's inflate, with the combined CRC-32."""
    import torch
    za = product()
    side, mid = 64 << 20, 2 << 30
    n = side + mid + side
    host = za.silesia_mix(n)
    want_crc = zlib.crc32(host)
    t_in = torch.from_numpy(host).cuda()
    t_out = torch.empty(n // 2 + (64 << 20), dtype=torch.uint8, device="cuda")
    rc, ol, crc = encoder.deflate_device(t_in.data_ptr(), n, t_out.data_ptr(), t_out.numel(), 10)
    assert rc == 0 and (crc ^ 0xFFFFFFFF) == want_crc
    one = bytes(t_out[:ol].cpu().numpy())
    del t_in, t_out
    encoder.close()                                  # (its workspace goes back before the three range contexts take theirs)
    torch.cuda.empty_cache()
    ranges = [(0, side), (side, mid), (side + mid, side)]
    rc2, three, crc2, res = deflate_over_contexts(host, 3, 10, ranges=ranges)
    assert rc2 == 0 and crc2 == crc and three == one, (len(three), len(one))
    assert res[1]["infos"][1]["n"] == mid and res[1]["bit_begin"] == res[0]["bit_end"] and res[2]["bit_begin"] == res[1]["bit_end"]
    c, tot, eof = _inflate_crc(three)
    assert tot == n and c == want_crc and eof


def test_c5_rank_shape_one_gib_bzip2_range_between_its_neighbours(encoder):
    """BASELINE config 5 gives each of 8 GPUs a 1 GiB range of ONE 8 GiB BZip2_3 stream plus the 9 MB behind it that its last block
    may reach into (sharding.bzip2_window).  One rank's share at its size on the one GPU there is: a 1 GiB range between two 64 MiB
    ranges, through sharding.bzip2_stream_rank (block chain handed on, tables gathered, tactics replayed, bytes assembled and OR-ed) ==
    the stream of ONE call on the whole input; libbz2 decodes it to the input."""
    import torch
    from test_gpu_bzip2 import _bzip2_over_contexts
    za = product()
    side, mid = 64 << 20, 1 << 30
    n = side + mid + side
    host = za.silesia_mix(n)
    t_in = torch.from_numpy(host).cuda()
    t_out = torch.zeros(n // 2 + (64 << 20), dtype=torch.uint8, device="cuda")
    rc, ol, crc = encoder.bzip2_device(t_in.data_ptr(), n, t_out.data_ptr(), t_out.numel(), 14)
    assert rc == 0 and (crc ^ 0xFFFFFFFF) == zlib.crc32(host)
    one = bytes(t_out[:ol].cpu().numpy())
    one_blocks = [tuple(int(x) for x in b) for b in encoder.bz2_last_blocks()]
    del t_in, t_out
    encoder.close()
    torch.cuda.empty_cache()
    stream, blocks, nr = _bzip2_over_contexts(host, 3, 14, ranges=[(0, side), (side, mid), (side + mid, side)])
    assert nr == 3 and stream == one, (len(stream), len(one))
    assert [tuple(int(x) for x in b) for b in blocks] == one_blocks
    dec = bz2.BZ2Decompressor()
    c, tot, off = 0, 0, 0
    while off < len(stream):
        ch = dec.decompress(stream[off:off + (8 << 20)])
        off += 8 << 20
        c = zlib.crc32(ch, c); tot += len(ch)
    assert tot == n and dec.eof and c == zlib.crc32(host)


def test_two_contexts_at_config_2_size_on_one_gpu():
    """Two contexts, each compressing a 1 GiB entry (BASELINE config 2's size) at the same time on one GPU: a context's workspace for
    such an entry is about 75 GiB with the default 1 GiB shards of the match finder (60 bytes per byte of shard + 15 per atom slot), 45 GiB with
    512 MiB shards ("shard_kib": 3.5 % slower) -- both fit 288 GB twice.  One context of each kind here; both streams == the stream
    of a context alone, and inflate to the input."""
    import threading
    import torch
    za = product()
    n = 1 << 30
    host = za.silesia_mix(n)
    want_crc = zlib.crc32(host)
    t_in = torch.from_numpy(host).cuda()
    outs = [torch.empty(n // 2 + (64 << 20), dtype=torch.uint8, device="cuda") for _ in range(2)]
    got, errors = [None, None], []

    def run(k):
        try:
            torch.cuda.set_device(0)
            enc = za.Encoder(0)
            if k == 1:
                enc.set_knob("shard_kib", 512 << 10)
            for _ in range(2):
                got[k] = enc.deflate_device(t_in.data_ptr(), n, outs[k].data_ptr(), outs[k].numel(), 10)
            free, total = torch.cuda.mem_get_info()
            got[k] = got[k] + (total - free,)
            enc.close()
        except BaseException as e:          # noqa: BLE001
            errors.append((k, repr(e)))
    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    (rc0, ol0, crc0, used0), (rc1, ol1, crc1, used1) = got
    assert rc0 == rc1 == 0 and ol0 == ol1 and crc0 == crc1 and (crc0 ^ 0xFFFFFFFF) == want_crc
    a = bytes(outs[0][:ol0].cpu().numpy())
    assert a == bytes(outs[1][:ol1].cpu().numpy())
    c, tot, eof = _inflate_crc(a)
    assert tot == n and c == want_crc and eof
    assert max(used0, used1) < (200 << 30), (used0, used1)        # both workspaces, the input and the two outputs together


# ---- soak seeds (tests/gpu_soak2.py, gpu_bz2_soak.py, gpu_lzma_soak.py ran thousands of such cases in round 2; one seed each here) ----

def _soak_input(za, rng, n):
    mask = int(rng.integers(1, 32)); seed = int(rng.integers(0, 1 << 30))
    d = za.silesia_mix(n, seed=seed, class_mask=mask).tobytes()
    k = int(rng.integers(0, 6))
    if k == 0 and n > 1000:                          # long runs and repeats
        d = bytearray(d); o = int(rng.integers(0, n // 2)); ln = min(300000, n // 3); d[o:o + ln] = bytes(ln); d = bytes(d)
    elif k == 1 and n > 1000:                        # an incompressible stretch (stored blocks)
        d = bytearray(d); o = int(rng.integers(0, n // 2)); ln = min(200000, n // 3); d[o:o + ln] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes(); d = bytes(d)
    elif k == 2:
        d = bytes((rng.integers(0, int(rng.integers(2, 20)), n) + 65).astype(np.uint8))
    return d


@pytest.mark.parametrize("seed", (11,))
def test_soak_seed_deflate_cuts(encoder, seed):
    """Random inputs through the three cuts that must not change a byte -- shards inside one call, ranges over several contexts,
    batches of entries -- against the oracle (12 cases of the generator of tests/gpu_soak2.py)."""
    za = product()
    rng = np.random.default_rng(seed)
    for case in range(12):
        mode = case % 3
        method = int(rng.choice([10, 10, 9, 8, 7, 6]))
        if mode == 0:
            n = int(rng.integers(1, 3 << 20)); d = _soak_input(za, rng, n)
            kib = int(rng.choice([64, 128, 192, 1024]))
            encoder.set_knob("shard_kib", kib)
            try:
                out = bytearray(n + 64)
                rc2, ol, c2 = encoder.deflate_into(d, out, method)
            finally:
                encoder.set_knob("shard_kib", 1 << 20)
            rc, ref, crc = oracle_deflate(d, method)
            assert rc == rc2 and crc == c2 and (rc != 0 or bytes(out[:ol]) == ref), (case, "shards", kib, n, method)
        elif mode == 1:
            n = int(rng.integers(1, 3 << 20)); d = _soak_input(za, rng, n); world = int(rng.choice([2, 3, 5, 8]))
            rc, ref, crc = oracle_deflate(d, method)
            rc2, out, c2, _ = deflate_over_contexts(d, world, method, shard_kib=int(rng.choice([0, 128])) or None)
            assert rc == rc2 and (rc != 0 or (out == ref and crc == c2)), (case, "ranges", world, n, method)
        else:
            cnt = int(rng.integers(2, 40)); datas = [_soak_input(za, rng, int(rng.integers(0, 150000))) for _ in range(cnt)]
            m = method if method in (8, 9, 10) else 10
            for dd, (rc2, out, c2) in zip(datas, encoder.deflate_batch(datas, m)):
                rc, ref, crc = oracle_deflate(dd, m)
                assert rc == rc2 and crc == c2 and (rc != 0 or out == ref), (case, "batch", len(dd), m)


@pytest.mark.parametrize("seed", (5,))
def test_soak_seed_bzip2(encoder, seed):
    """Random small inputs of every kind through zada_bzip2_batch (all three methods) and streams of runs whose block limits fall
    inside runs (bzip2-encoding.adb:1161-1209), against the oracle (the generator of tests/gpu_bz2_soak.py)."""
    from _bzip2 import oracle_encode
    za = product()
    rng = np.random.default_rng(seed)
    mix = za.silesia_mix(2 << 20)

    def small_case():
        kind = int(rng.integers(0, 6))
        n = int(rng.integers(0, 120000)) if rng.random() < 0.8 else int(rng.integers(0, 600))
        if kind == 0:
            o = int(rng.integers(0, len(mix) - n - 1)); return bytes(mix[o:o + n])
        if kind == 1:
            return bytes(rng.integers(0, int(rng.integers(1, 256)) + 1, n, dtype=np.uint8))
        if kind == 2:
            return bytes(np.repeat(rng.integers(0, 256, n // 3 + 1, dtype=np.uint8), rng.integers(1, int(rng.integers(2, 600)), n // 3 + 1))[:n])
        if kind == 3:
            p = bytes(rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8)); return (p * (n // len(p) + 1))[:n]
        if kind == 4:
            return bytes(rng.integers(97, 101, n, dtype=np.uint8)) + bytes(mix[:n // 2])
        return bytes(n)
    cases = [small_case() for _ in range(24)]
    for method in (12, 13, 14):
        sel = cases if method == 14 else cases[::3]
        for d, (rc, p, crc) in zip(sel, encoder.bzip2_batch(sel, method)):
            o, _ = oracle_encode(d, method - 12)
            assert p == o and (crc ^ 0xFFFFFFFF) == zlib.crc32(d) and rc == (1 if len(o) >= len(d) else 0), (method, len(d))
    for k in range(2):
        n = int(rng.integers(950000, 1600000))
        maxrun = int(rng.choice([6, 300, 5000]))
        vals = rng.integers(0, int(rng.choice([2, 4, 16, 256])), n // 2 + 2, dtype=np.uint8)
        d = bytes(np.repeat(vals, rng.integers(1, maxrun + 1, n // 2 + 2))[:n])
        o, ev = oracle_encode(d, 2)
        rc, p, crc = encoder.bzip2(d, 14, cap=len(d) * 2 + 4096)
        assert p == o and encoder.bz2_last_blocks() == ev, (k, n, maxrun)


@pytest.mark.parametrize("seed", (1003,))
def test_soak_seed_lzma(encoder, seed):
    """Random entries (sizes 0 .. 24 KiB incl. the 161 / 162 / 273 edges of BT4's tail, lz77.adb:959, 1000-1017; random, few-symbol,
    periodic and corpus data) through zada_lzma_batch, every payload against the oracle and through liblzma (tests/gpu_lzma_soak.py)."""
    from _lzmah import oracle_lzma, lzma_decode
    za = product()
    rng = np.random.default_rng(seed)
    datas = []
    for i in range(90):
        # (16 222 / 16 384 / 16 385: the producer keeps an entry's trees in LDS up to 16 KiB, zada_bt4.hip; 162 / 163: nothing / one position inserted)
        n = int(rng.choice([0, 1, 2, 3, 161, 162, 163, 164, 273, 274, 4096, 16222, 16383, 16384, 16385, 33000, int(rng.integers(0, 24576)), int(rng.integers(0, 24576)), int(rng.integers(0, 2000))]))
        kind = int(rng.integers(0, 6))
        if kind == 0:
            d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        elif kind == 1:
            d = bytes(rng.integers(0, int(rng.integers(2, 5)), n, dtype=np.uint8) + 65)
        elif kind == 2:
            d = (bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)) * (n + 1))[:n]
        else:
            d = bytes(za.silesia_mix(n, class_mask=int(rng.integers(1, 32)), seed=int(rng.integers(1, 1 << 30)), offset=int(rng.integers(0, 1 << 20))))
        datas.append(d)
    for m in (15, 16, 17, 18):
        for i, (d, got) in enumerate(zip(datas, encoder.lzma_batch(datas, m))):
            assert got == oracle_lzma(d, m), (m, i, len(d))
            assert lzma_decode(got[1], 4) == d


@pytest.mark.parametrize("seed", (4242,))
def test_soak_seed_lzma_stream(encoder, seed):
    """ONE LZMA_3 stream per case, coded in launches with the match producer in segments: random sizes, segment sizes ("lzma_segment"), launch
    budgets ("lzma_chunk") and dictionaries below the stream's size (window fills, moves; knob "lzma_dict") on corpus, periodic, few-symbol and
    random data -- the first ten cases of tests/gpu_lzma_soak_stream.py (the script that found the reference's defect behind unprocessed pending
    bytes).  Every payload == the oracle's and decodes; an entry the product refuses (ZADA_E_REFERENCE) is one whose reference
    stream does not decode to the input, or that the sequential matcher's own sets show a match that is none for."""
    from _lzmah import oracle_lzma_encode, lzma_decode, lzma_symbols
    za = product()
    mix = za.silesia_mix(32 << 20)
    rng = np.random.default_rng(seed)
    coded = 0
    try:
        for k in range(10):
            n = int(rng.integers(20000, 2_500_000))
            kind = int(rng.integers(0, 7))
            if kind <= 3:
                o = int(rng.integers(0, len(mix) - n - 1)); d = bytes(mix[o:o + n])
            elif kind == 4:
                per = bytes(rng.integers(0, 256, int(rng.integers(1, 70000)), dtype=np.uint8)); d = (per * (n // len(per) + 1))[:n]
            elif kind == 5:
                d = bytes((rng.integers(0, int(rng.integers(2, 6)), n) + 65).astype(np.uint8))
            else:
                n = min(n, 400000); d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
            seg = int(rng.choice([-1, 0, 13, 14, 15, 16, 17, 18, 19, 20]))
            chunk = int(rng.choice([0, 0, 4096, 10000, 65536, 200000]))
            ds = int(rng.choice([0, 0, 0, 5000, 70000, 300000]))
            if ds >= n:
                ds = 0
            encoder.set_knob("lzma_segment", seg); encoder.set_knob("lzma_chunk", chunk); encoder.set_knob("lzma_dict", ds)
            want, _ = oracle_lzma_encode(d, 3, dictionary_size=ds or None)
            try:
                rc, z, crc = encoder.lzma(d, 18)
            except za.ReferenceDefect:
                assert ds and lzma_symbols(want)[0] != d, (k, n, ds)
                continue
            assert z == bytes([16, 2, 5, 0]) + want, (k, n, kind, seg, chunk, ds)
            assert lzma_decode(z, 4) == d
            coded += 1
    finally:
        for kn in ("lzma_segment", "lzma_chunk", "lzma_dict"):
            encoder.set_knob(kn, 0)
    assert coded >= 8


def test_host_path_with_the_input_still_arriving_equals_the_device_path(encoder):
    """zada_deflate on host buffers of 64 MiB and more copies the input in on background lanes while the link stage already runs on what has
    come, 64 MiB at a time (zada_api.hip Arrival, zada_lz.hip lz_shard `job.need`: k_prev_links and k_bucket_limits piece by piece,
    k_cross_links too when the pieces come slowly): sizes around the pieces' edges, Deflate_1 and Deflate_3 -- the stream == the
    device-resident entry point's on the same bytes and inflates to the input (tests/gpu_hostpath_soak.py has more sizes)."""
    import torch
    za = product()
    M = 1 << 20
    sizes = [64 * M + 1, 64 * M + 32768 + 63, 96 * M - 5, 128 * M + 65]
    base = za.silesia_mix(max(sizes) + 64, seed=77)
    for n in sizes:
        for method in (8, 10):
            host = base[7:7 + n].copy()
            hout = np.zeros(n + 64, dtype=np.uint8)
            got = encoder.deflate_into(host, hout, method)
            t_in = torch.from_numpy(host).cuda()
            t_out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
            dev = encoder.deflate_device(t_in.data_ptr(), n, t_out.data_ptr(), t_out.numel(), method)
            assert got == dev and got[0] == 0, (n, method, got, dev)
            assert np.array_equal(hout[:got[1]], t_out[:dev[1]].cpu().numpy()), (n, method)
            c, tot, eof = _inflate_crc(bytes(hout[:got[1]]))
            assert tot == n and eof and c == zlib.crc32(host) == (got[2] ^ 0xFFFFFFFF)
            del t_in, t_out
