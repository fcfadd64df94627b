"""world_size-2 CPU (gloo) test of the multi-GPU layout: entries shard over ranks, payloads are
gathered onto rank 0 and stitched into one .zip stream that equals the single-process archive.
The compressor is injected (the oracle stands in for the GPU encoder here, which needs a GPU)."""
import io
import os
import sys
import zipfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _common import ROOT, oracle_zip, silesia_mix


def _worker(rank, world, port, q, method=10):
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
            assert O.zo_compress_data(d, ln, method, out, ln + 64, ctypes.byref(ol), ctypes.byref(crc), ctypes.byref(zt)) == 0
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
        zc = za.ZipCreate(None, method)
        for idx, payload, crc, usize, zt in gathered:
            zc.add_compressed("entry_%04d.bin" % idx, payload, crc, usize, zt)
        q.put(zc.finish())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("method", [10, 18])
def test_two_rank_gather_and_stitch_equals_single_process_archive(method):
    """Entries of an archive over two ranks (independent objects: no collective on the data path, a gather of the payloads at the
    end) -- Deflate_3, and LZMA_3, whose only parallelism IS the entries: Zip format 14, end-marker flag."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + method * 7) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, method)) for r in range(2)]
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
    assert got == oracle_zip(entries, method)
    zf = zipfile.ZipFile(io.BytesIO(got))
    assert zf.testzip() is None and len(zf.infolist()) == len(entries)
    assert all(i.compress_type == (14 if method == 18 else 8) for i in zf.infolist())


# ----------------------------------------------------------------------------------------------------------------------
# ONE stream over several ranks (sharding.deflate_stream_rank) on gloo, with a CPU model of the range calls
# ----------------------------------------------------------------------------------------------------------------------
class OracleRangeModel:
    """CPU stand-in for the zada_range_* calls of one rank, built from ONE run of the oracle over the whole stream: the
    rank's atoms are the oracle's tokens that start inside its range, its bits those of the blocks of the flushes it owns.
    It checks everything the protocol hands it (atom offsets, neighbours' atoms, bit position) against the oracle, so the
    test covers sharding.py's bookkeeping: counts -> flush grid -> look-behind / look-ahead assembly -> chooser chain ->
    byte offsets and OR-merge of the payloads."""

    def __init__(self, data, method, lie_once=False):
        import numpy as np
        from _common import oracle_deflate, oracle_tokens, oracle
        self.np, self.data, self.method = np, data, method
        self.blocks, self.bitpos = [], []
        rc, self.stream, crc = oracle_deflate(data, method, self.blocks, None, None, self.bitpos)
        assert rc == 0
        self.tok = oracle_tokens(data, method)
        lens = np.where(self.tok & 0x80000000, (self.tok >> 16) & 0x1FF, 1).astype(np.int64)
        self.start = np.concatenate(([0], np.cumsum(lens)))          # start[i] = byte position of token i; start[-1] = n
        self.total_bits = len(self.stream) * 8
        self.lie_once = lie_once
        self.O = oracle()

    @staticmethod
    def _arr(ptr, n):
        import ctypes
        import numpy as np
        return np.ctypeslib.as_array((ctypes.c_uint32 * n).from_address(ptr)) if n else np.zeros(0, dtype=np.uint32)

    def range_open(self, d_in_ptr, stream_size, lo, n, pre, post, method):
        assert stream_size == len(self.data) and method == self.method
        assert lo % 65536 == 0 and pre == (32768 if lo else 0) and post == min(1 << 20, stream_size - lo - n)
        self.lo, self.n = lo, n

    def _first_token_at_or_after(self, p):
        return int(self.np.searchsorted(self.start[:-1], p, side="left"))

    def range_lz(self, entry):
        i0, i1 = self._first_token_at_or_after(self.lo), self._first_token_at_or_after(self.lo + self.n)
        self.i0, self.i1 = i0, i1
        warm = (int(self.start[i0]), 1)
        if entry is None and self.lie_once and self.lo > 0:           # a warm-up parse that did not synchronise
            warm = (warm[0] + 1, 2)
        if entry is not None:
            assert tuple(entry) == (int(self.start[i0]), 1), entry    # the true state of the range before
        raw = self.O.zo_crc32_update(0, self.data[self.lo:self.lo + self.n], self.n)
        return dict(atoms=i1 - i0, exit=(int(self.start[i1]), 1), warm=warm, crc_raw=raw, entry_known=entry is not None or self.lo == 0)

    def range_edges(self, ha, hp, ta, tp):
        nh, nt = min(self.i1 - self.i0, 65536), min(self.i1 - self.i0, 2048)
        self._arr(ha, 65536)[:nh] = self.tok[self.i0:self.i0 + nh]; self._arr(hp, 65536)[:nh] = self.start[self.i0:self.i0 + nh] & 0xFFFFFFFF
        self._arr(ta, 2048)[:nt] = self.tok[self.i1 - nt:self.i1]; self._arr(tp, 2048)[:nt] = self.start[self.i1 - nt:self.i1] & 0xFFFFFFFF
        return nh, nt

    def range_place(self, before, total, lba, lbp, n_lb, laa, lap, n_la):
        np = self.np
        assert before == self.i0 and total == len(self.tok)
        if n_lb:
            assert (self._arr(lba, n_lb) == self.tok[before - n_lb:before]).all() and (self._arr(lbp, n_lb) == (self.start[before - n_lb:before] & 0xFFFFFFFF)).all()
        if n_la:
            assert (self._arr(laa, n_la) == self.tok[self.i1:self.i1 + n_la]).all() and (self._arr(lap, n_la) == (self.start[self.i1:self.i1 + n_la] & 0xFFFFFFFF)).all()
        f0 = (before + 65535) // 65536 * 65536
        self.own_lo, self.own_hi = f0, f0
        if self.i1 > f0:
            nfl = (self.i1 - f0 + 65535) // 65536
            self.own_hi = f0 + nfl * 65536
            # every window the splitter looks at must be inside [before - n_lb, i1 + n_la)
            assert min(self.own_hi, total) <= self.i1 + n_la
            assert f0 == 0 or f0 - 2048 >= before - n_lb

    def range_analyze(self):
        pass

    def range_choose(self, carry):
        import struct
        pos_in = struct.unpack_from("<Q", carry, 0)[0] if carry is not None else 0
        mine = [k for k, b in enumerate(self.blocks) if self.own_lo <= b[0] < self.own_hi]
        begin, end = pos_in, pos_in
        if mine:
            assert self.bitpos[mine[0]][1] == pos_in, (self.bitpos[mine[0]], pos_in)
            nxt = mine[-1] + 1
            end = self.bitpos[nxt][1] if nxt < len(self.blocks) else self.total_bits
        elif len(self.tok) == 0 or (self.own_lo >= len(self.tok) and self.i1 == len(self.tok) and not self.blocks):
            end = self.total_bits
        self.span = (begin, end)
        out = bytearray(352)
        struct.pack_into("<Q", out, 0, end)
        return bytes(out), begin, end

    def range_emit(self, d_out_ptr, cap):
        import ctypes
        b0, b1 = self.span
        off, ln = b0 // 8, (b1 + 7) // 8 - b0 // 8
        buf = bytearray(self.stream[off:off + ln])
        if ln:
            buf[0] &= (0xFF << (b0 & 7)) & 0xFF                       # bits before bit_begin belong to the range before
            if b1 & 7:
                buf[-1] &= (1 << (b1 & 7)) - 1                        # ... and those from bit_end on to the next one
        assert ln <= cap
        ctypes.memmove(d_out_ptr, bytes(buf), ln)
        return ln


def _stream_worker(rank, world, port, q, n, method, lie):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sharding = importlib.import_module("zip-ada_amd.sharding")
    za = importlib.import_module("zip-ada_amd")
    data = silesia_mix(n)
    ranges = sharding.stream_ranges(n, world)
    enc = OracleRangeModel(data, method, lie_once=lie)
    comm = sharding.TorchComm(torch.device("cpu"))
    res = sharding.deflate_stream_rank(enc, comm, torch, n, ranges, 0, method,
                                       lambda k: torch.zeros(k, dtype=torch.int32), lambda k: torch.zeros(k, dtype=torch.uint8))
    payload = res["payload"] if res["payload"] is not None else torch.zeros(1, dtype=torch.uint8)
    out = sharding.gather_stream(payload, res["spans"], res["total_bits"], dst=0)
    if rank == 0:
        lib = za.load_library()
        crc = sharding.stream_crc(lib.zada_crc32_combine, res["infos"])
        q.put((bytes(out.numpy()), crc, [i["atoms"] for i in res["infos"] if i is not None]))
    dist.barrier()
    dist.destroy_process_group()


def _run_stream(world, n, method, lie=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + world * 7 + n) % 2000
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, q, n, method, lie)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_one_stream_over_two_and_three_ranks_protocol():
    """The exchange protocol of sharding.deflate_stream_rank over gloo: stitched stream == the oracle's stream, combined
    CRC == zlib's; range boundaries inside flushes (3 MiB, ~900 000 atoms at Deflate_3) and ranges without a flush of
    their own (Deflate_0 on 200 000 bytes over 3 ranks: 65 536, 65 536 and 68 928 atoms)."""
    import zlib
    from _common import oracle_deflate
    for world, n, method, lie in ((2, 3 << 20, 10, False), (3, (2 << 20) + 77, 10, True), (3, 200000, 7, False)):
        data = silesia_mix(n)
        rc, ref, _ = oracle_deflate(data, method)
        out, crc, atoms = _run_stream(world, n, method, lie)
        assert rc == 0 and out == ref, (world, n, method)
        assert crc ^ 0xFFFFFFFF == zlib.crc32(data)
        assert len(atoms) == len([r for r in importlib_sharding().stream_ranges(n, world)])


def test_one_stream_over_eight_ranks_protocol():
    """The world the benchmark's config 3 runs in: eight ranks.  Ranges of 10 x 64 KiB and less (several ranges inside one flush of 65 536 atoms,
    look-ahead atoms assembled from up to seven successors, look-behind across ranges without a flush of their own), and every warm-up state
    wrong once (seven re-runs, each followed by a fresh all_gather of the states) -- the stitched stream is the oracle's, the combined CRC zlib's.
    All exchanges are fixed-size tensor collectives (one 64-byte-per-rank and one 24-byte-per-rank all_gather per step) and the posted receive /
    send pair of the chooser's 352 bytes: no object collectives (sharding.TorchComm)."""
    import zlib
    from _common import oracle_deflate
    for n, method, lie in (((5 << 20) + 4321, 10, False), (900000, 8, True)):
        data = silesia_mix(n)
        rc, ref, _ = oracle_deflate(data, method)
        out, crc, atoms = _run_stream(8, n, method, lie)
        assert rc == 0 and out == ref, (n, method)
        assert crc ^ 0xFFFFFFFF == zlib.crc32(data)
        assert len(atoms) == 8


def test_no_object_collectives_in_the_exchange_path():
    """pickled objects (all_gather_object / broadcast_object_list) are host round trips with implicit synchronisations on
    the nccl backend; the path that runs between the ranks of a step must not use them."""
    src = open(os.path.join(ROOT, "zip-ada_amd", "sharding.py")).read()
    assert "all_gather_object" not in src and "broadcast_object_list" not in src
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert "all_gather_object" not in bench and "broadcast_object_list" not in bench


def importlib_sharding():
    import importlib
    return importlib.import_module("zip-ada_amd.sharding")


class OracleBz2RangeModel:
    """The BZip2 range methods of Encoder (zada_bz2_range_*) made of oracle calls: block limits from the oracle's trace of the
    whole stream, every piece of every tactic through zo_bz2_block, the choice through the product's host function
    zada_bz2_select (no GPU needed), the bytes put together with Python integers.  Stands in for the GPU encoder in the
    gloo test of sharding.bzip2_stream_rank."""

    def __init__(self, data, method):
        import importlib
        from _bzip2 import oracle_encode
        self.data, self.method, self.option = data, method, method - 12
        self.ref, self.trace = oracle_encode(data, self.option)
        self.lib = importlib.import_module("zip-ada_amd").load_library()

    def bz2_range_open(self, d_buf, buf_len, buf_off, stream_total, start, own_end, method=14):
        mine = [b for b in self.trace if start <= b[0] < own_end]
        assert all(b[0] + b[1] <= buf_off + buf_len for b in mine), "halo too short"
        assert not mine or mine[0][0] == start, "the hand-over must be the start of a block"
        self.blocks = mine
        return (mine[-1][0] + mine[-1][1]) if mine else start, len(mine)

    def bz2_range_encode(self):
        import numpy as np
        from _bzip2 import bz_oracle, oracle_block
        O = bz_oracle()
        self.pieces, self.bits = [], {}
        for (st, ln, _t, _k) in self.blocks:
            raw = self.data[st:st + ln]
            tac = {0: [(0, ln)]}
            if self.option == 2:
                size, stop, p4 = ln // 4, 0, []
                for count in range(1, 5):
                    s0 = stop + 1
                    stop = ln if count == 4 else count * size
                    p4.append((s0 - 1, stop - s0 + 1))
                tac[1] = p4
                for t in (2, 3):
                    seg = np.zeros(ln // 4000 + 4, np.int32)
                    k = O.zo_bz2_segments(raw, ln, t, seg.ctypes.data, seg.size)
                    idx, lst = 1, []
                    for e in seg[:k]:
                        lst.append((idx - 1, int(e) - idx + 1)); idx = int(e) + 1
                    tac[t] = lst
            for t in tac:
                for (o, l) in tac[t]:
                    if (st + o, l) not in self.bits:
                        ob = oracle_block(raw[o:o + l], self.option, want_bits=True)
                        self.bits[(st + o, l)] = (ob["info"].bits, ob["info"].block_crc, bytes(ob["bits"]))
            self.pieces.append({t: [(st + o, l) for (o, l) in v] for t, v in tac.items()})

    def bz2_range_table(self):
        import numpy as np
        tab = np.zeros((len(self.blocks), 4, 3), np.uint64)
        for q, tac in enumerate(self.pieces):
            for t in range(4):
                if t not in tac:
                    tab[q, t] = (0xFFFFFFFFFFFFFFFF, 0, 0)
                    continue
                bits, fold = 0, 0
                for key in tac[t]:
                    b, c, _ = self.bits[key]
                    bits += b
                    fold = (((fold << 1) | (fold >> 31)) & 0xFFFFFFFF) ^ c
                tab[q, t] = (bits, len(tac[t]), fold)
        return tab

    def bz2_select(self, tab, bitpos_in=32, crc_in=0):
        import ctypes
        import numpy as np
        tab = np.ascontiguousarray(tab, np.uint64)
        nb = tab.shape[0]
        choice = np.zeros(max(nb, 1), np.uint8)
        bp, crc = ctypes.c_uint64(0), ctypes.c_uint32(0)
        self.lib.zada_bz2_select(nb, tab.ctypes.data, bitpos_in, crc_in, choice.ctypes.data, ctypes.byref(bp), ctypes.byref(crc))
        return choice[:nb], bp.value, crc.value

    def bz2_range_assemble(self, choice, bit_begin, d_out, cap, header=False, footer_crc=None):
        import ctypes
        base = 0 if header else bit_begin // 8 * 8
        acc, nbits = 0, bit_begin - base                       # acc holds nbits bits, most significant first
        if header:
            acc, nbits = int.from_bytes(b"BZh" + bytes([48 + (1, 4, 9)[self.option]]), "big"), 32
        for q, tac in enumerate(self.pieces):
            for key in tac[int(choice[q])]:
                b, _c, by = self.bits[key]
                acc = (acc << b) | (int.from_bytes(by, "big") >> (len(by) * 8 - b))
                nbits += b
        if footer_crc is not None:
            acc = (acc << 80) | (0x177245385090 << 32) | footer_crc
            nbits += 80
        ln = (nbits + 7) // 8
        buf = (acc << (ln * 8 - nbits)).to_bytes(ln, "big") if ln else b""
        assert ln <= cap
        ctypes.memmove(d_out, buf, ln)
        self.last = [(b[0], b[1], int(choice[q]), len(self.pieces[q][int(choice[q])])) for q, b in enumerate(self.blocks)]
        return ln

    def bz2_last_blocks(self):
        return self.last


def _bz_stream_worker(rank, world, port, q, n, method):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sharding = importlib.import_module("zip-ada_amd.sharding")
    data = bytes(silesia_mix(n))
    ranges = sharding.bzip2_ranges(n, world, method)
    enc = OracleBz2RangeModel(data, method)
    comm = sharding.TorchComm(torch.device("cpu"))
    res = sharding.bzip2_stream_rank(enc, comm, n, ranges, 0, method, lambda k: torch.zeros(k, dtype=torch.uint8))
    payload = res["payload"] if res["payload"] is not None else torch.zeros(1, dtype=torch.uint8)
    out = sharding.gather_stream(payload, res["spans"], res["total_bits"], dst=0)
    blocks = [None] * world
    dist.all_gather_object(blocks, res["blocks"])
    if rank == 0:
        q.put((bytes(out.numpy()), [b for bl in blocks for b in bl], len(ranges), enc.ref, enc.trace))
    dist.barrier()
    dist.destroy_process_group()


def test_one_bzip2_stream_over_two_and_three_ranks_protocol():
    """sharding.bzip2_stream_rank over gloo (block-chain hand-over by send / recv, all_gather of the tables, the choice
    replayed on every rank, payload gather, OR at the joints): stitched stream == the oracle's stream, block for block."""
    import bz2
    for world, n, method in ((2, (5 << 20) + 333, 12), (3, 7 << 20, 12), (2, 1 << 20, 14)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 33500 + (os.getpid() + world * 11 + n) % 2000
        procs = [ctx.Process(target=_bz_stream_worker, args=(r, world, port, q, n, method)) for r in range(world)]
        for p in procs:
            p.start()
        out, blocks, nr, ref, trace = q.get(timeout=240)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert nr == (world if method == 12 else 1)
        assert out == ref and blocks == trace, (world, n, method)
        assert bz2.decompress(out) == bytes(silesia_mix(n))


# ----------------------------------------------------------------------------------------------------------------------
# gather_stream_begin: every range received at its byte offset, shared edge bytes OR-ed
# ----------------------------------------------------------------------------------------------------------------------
def _gather_worker(rank, world, port, q, spans, seed):
    sys.path.insert(0, ROOT)
    import importlib
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sharding = importlib.import_module("zip-ada_amd.sharding")
    total_bits = max(s[1] for s in spans if s is not None)
    bits = np.random.RandomState(seed).randint(0, 2, total_bits).astype(np.uint8)
    sp = spans[rank] if rank < len(spans) else None
    payload = torch.zeros(1, dtype=torch.uint8)
    if sp is not None:
        own = np.zeros((total_bits + 7) // 8 * 8, np.uint8)
        own[sp[0]:sp[1]] = bits[sp[0]:sp[1]]                        # this range's bits only, at their place in the stream
        by = np.packbits(own, bitorder="little")
        off, ln = sp[0] // 8, max(0, (sp[1] + 7) // 8 - sp[0] // 8)
        payload = torch.from_numpy(np.concatenate([by[off:off + ln], np.full(5, 0xAA, np.uint8)]))   # bytes behind ln are not the range's
    h = sharding.gather_stream_begin(payload, spans, total_bits, dst=0)
    out = h.finish()
    if rank == 0:
        full = np.zeros((total_bits + 7) // 8 * 8, np.uint8)
        full[:total_bits] = bits
        q.put(bytes(out.numpy()) == bytes(np.packbits(full, bitorder="little")))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("spans", [
    [(0, 100003), (100003, 250001), (250001, 250002), (250002, 250009), (250009, 400000)],      # ranges of 1 and 7 bits inside one byte shared by three
    [(0, 64), (64, 72), (72, 80), (80, 1000)],                                                  # byte-aligned joints, one-byte ranges
    [(0, 5), (5, 11), None, None],                                                               # tiny stream, two ranks without a range
    [(0, 3 * 8 * 1000 + 1), (3 * 8 * 1000 + 1, 3 * 8 * 1000 + 17), (3 * 8 * 1000 + 17, 50000)],  # a range of exactly two (shared) bytes
])
def test_stream_gather_puts_every_range_at_its_offset(spans):
    world = len(spans)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() + world * 13 + spans[0][1]) % 2000
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q, spans, 5)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
