"""CPU tests of the BZip2 half of the oracle (oracle/zada_oracle_bz2.c).

Parity against the Ada build is UNPINNED (no GNAT here).  What these tests hold on to: every stream decompresses with
libbz2 to its input (the reference's own tests are round trips too: test/test_za.hac:169-172), the committed digests
(tests/golden/bzip2_digests.json, made by tests/golden/make_golden.py), hand-derivable known answers, and the coverage of
the four splitting tactics."""
import bz2
import ctypes
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

from _common import GOLDEN
from _bzip2 import bz_inputs, bz_oracle, oracle_block, oracle_encode


def test_streams_round_trip_and_match_the_digests():
    dig = json.load(open(os.path.join(GOLDEN, "bzip2_digests.json")))
    cases = bz_inputs()
    tactics = set()
    checked = 0
    for key, want in sorted(dig.items()):
        name, m = key.split("|")
        m = int(m)
        d = cases.get(name)
        if d is None:
            d = open(os.path.join(GOLDEN, name), "rb").read()
        if len(d) > 120000 and not (m == 14 and name in ("seg1_az_digits", "seg2_alphabets", "copies_1500k", "mix_1m_off")):
            continue                                              # the large inputs are the GPU suite's; a few stay here for the tactics
        assert hashlib.sha256(d).hexdigest() == want["in_sha256"], key
        z, ev = oracle_encode(d, m - 12)
        assert len(z) == want["size"] and hashlib.sha256(z).hexdigest() == want["sha256"], key
        assert [list(e) for e in ev] == want["blocks"], key
        if len(d):
            assert bz2.decompress(z) == d, key
        tactics |= {e[2] for e in ev}
        checked += 1
    assert checked > 100
    assert tactics == {0, 1, 2, 3}, "every splitting tactic of bzip2-encoding.adb:1214-1345 must be kept somewhere in the matrix"


def test_known_answers():
    O = bz_oracle()
    # BZip2.CRC (bzip2.adb): the check value of CRC-32/BZIP2
    assert O.zo_bz2_crc(b"123456789", 9) == 0xFC891918
    # stream header and footer (bzip2.ads:118-123, bzip2-encoding.adb:1380-1403); an empty input still gives one (empty) block (:1411-1428)
    z, ev = oracle_encode(b"", 2)
    assert z[:4] == b"BZh9" and z[4:10] == b"1AY&SY" and ev == [(0, 0, 0, 1)]
    for opt, digit in ((0, b"1"), (1, b"4"), (2, b"9")):
        z, _ = oracle_encode(b"hello hello hello", opt)
        assert z[:3] == b"BZh" and z[3:4] == digit and bz2.decompress(z) == b"hello hello hello"
    # the block CRC of a one-block stream is also the combined CRC in the footer (:1023-1026)
    d = b"The quick brown fox jumps over the lazy dog. " * 50
    z, _ = oracle_encode(d, 2)
    crc = O.zo_bz2_crc(d, len(d))
    assert int.from_bytes(z[10:14], "big") == crc
    bits = int.from_bytes(z, "big")
    foot = (0x177245385090 << 32) | crc
    assert any(((bits >> sh) & ((1 << 80) - 1)) == foot for sh in range(8)), "footer magic + combined CRC end the stream"


def test_rle1_and_block_limits():
    # RLE_1 (:167-213): runs of 4 .. 259 become four bytes and a count
    for run, want in ((1, b"a"), (3, b"aaa"), (4, b"aaaa\x00"), (5, b"aaaa\x01"), (259, b"aaaa\xff"), (260, b"aaaa\xffa"), (263, b"aaaa\xffaaaa\x00")):
        assert bytes(oracle_block(b"a" * run)["rle"]) == want, run
    # data acquisition (:1161-1209): a block takes at most ten capacities of raw bytes; the last two blocks are balanced (:1406-1423)
    z, ev = oracle_encode(bytes(1_200_000), 0)
    assert [e[1] for e in ev] == [1_000_000, 200_000]
    rng = np.random.default_rng(3)
    d = bytes(rng.integers(0, 256, 1_000_000, dtype=np.uint8))     # 1.05 .. 1.30 capacities: two halves
    z, ev = oracle_encode(d, 2)
    assert len(ev) == 2 and abs(ev[0][1] - 500_000) < 3000 and ev[0][1] + ev[1][1] == len(d)
    assert bz2.decompress(z) == d


def test_entropy_stage_invariants():
    rng = np.random.default_rng(5)
    d = bytes(rng.integers(97, 105, 60000, dtype=np.uint8)) + b"abcabcabd" * 3000
    o = oracle_block(d, 2, want_bits=True)
    info = o["info"]
    assert 2 <= info.coders <= 6 and info.max_code_len in (15, 17) and info.sample_width in (3, 4)
    assert info.selector_count == 1 + (info.mtf_n - 1) // 50 and set(o["selectors"].tolist()) <= set(range(1, info.coders + 1))
    for c in range(info.coders):
        l = o["lens"][c, :info.alphabet].astype(np.int64)
        assert l.min() >= 1 and l.max() <= info.max_code_len
        assert sum(2.0 ** -x for x in l) <= 1.0 + 1e-12          # Kraft: a prefix code exists
    assert o["mtf"][-1] == info.alphabet - 1                       # EOB closes the block (:408)
    assert len(o["bits"]) == (info.bits + 7) // 8


def test_zip_entry_semantics():
    O = bz_oracle()
    d = b"zip-ada " * 4000
    out = ctypes.create_string_buffer(len(d) + 64)
    n = ctypes.c_uint64()
    crc = ctypes.c_uint32(0xFFFFFFFF)
    assert O.zo_bzip2(d, len(d), 14, out, len(d) + 64, ctypes.byref(n), ctypes.byref(crc)) == 0
    assert bz2.decompress(out.raw[:n.value]) == d and (crc.value ^ 0xFFFFFFFF) == zlib.crc32(d)
    d = os.urandom(3000)                                            # not smaller than the input: compression_ok := False
    assert O.zo_bzip2(d, len(d), 14, out, len(d) + 64, ctypes.byref(n), ctypes.byref(crc)) == 1
    assert O.zo_bzip2(d, len(d), 10, out, len(d) + 64, ctypes.byref(n), ctypes.byref(crc)) < 0


def test_tactic_choice_replay_and_stream_ranges():
    """zada_bz2_select (the product's host-side replay of bzip2-encoding.adb:1312-1318 and :1023-1026, which every rank of a
    multi-GPU run executes on the gathered tables) against the oracle: tables built from the oracle's own Encode_Block sizes of
    every piece of every tactic must lead to the oracle's choices, stream length and footer CRC.  Also the range geometry."""
    import importlib
    from _common import product
    Z = product()
    L = Z.load_library()                                             # loads without a GPU; zada_bz2_select is plain host arithmetic
    sh = importlib.import_module("zip-ada_amd.sharding")
    O = bz_oracle()
    cases = bz_inputs()
    for name in ("seg1_az_digits", "seg2_alphabets", "copies_1500k"):
        data = cases[name]
        z, ev = oracle_encode(data, 2)
        tab = np.zeros((len(ev), 4, 3), np.uint64)
        for q, (start, ln, _t, _k) in enumerate(ev):
            raw = data[start:start + ln]
            pieces = {0: [(0, ln)]}
            size, stop, p4 = ln // 4, 0, []
            for count in range(1, 5):
                s0 = stop + 1
                stop = ln if count == 4 else count * size
                p4.append((s0 - 1, stop - s0 + 1))
            pieces[1] = p4
            for t in (2, 3):
                seg = np.zeros(ln // 4000 + 4, np.int32)
                k = O.zo_bz2_segments(raw, ln, t, seg.ctypes.data, seg.size)
                idx, lst = 1, []
                for e in seg[:k]:
                    lst.append((idx - 1, int(e) - idx + 1)); idx = int(e) + 1
                pieces[t] = lst
            memo = {}
            for t in range(4):
                bits, fold = 0, 0
                for (o, l) in pieces[t]:
                    if (o, l) not in memo:
                        info = oracle_block(raw[o:o + l])["info"]
                        memo[(o, l)] = (info.bits, info.block_crc)
                    b, c = memo[(o, l)]
                    bits += b
                    fold = (((fold << 1) | (fold >> 31)) & 0xFFFFFFFF) ^ c
                tab[q, t] = (bits, len(pieces[t]), fold)
        choice = np.zeros(len(ev), np.uint8)
        bp, crc = ctypes.c_uint64(), ctypes.c_uint32()
        L.zada_bz2_select.restype = None
        L.zada_bz2_select.argtypes = [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
        L.zada_bz2_select(len(ev), tab.ctypes.data, 32, 0, choice.ctypes.data, ctypes.byref(bp), ctypes.byref(crc))
        assert choice.tolist() == [e[2] for e in ev], name
        assert (bp.value + 80 + 7) // 8 == len(z), name
        bits = int.from_bytes(z, "big")
        assert any(((bits >> sh_) & 0xFFFFFFFF) == crc.value for sh_ in range(8)), name
    # ranges: at least two halos each, together the stream; a window ends with the stream or a halo behind the range
    for total, world in ((0, 4), (5 << 20, 8), (300 << 20, 8), (8 << 30, 8)):
        r = sh.bzip2_ranges(total, world)
        assert sum(n for _, n in r) == total and all(r[k][0] + r[k][1] == r[k + 1][0] for k in range(len(r) - 1))
        assert len(r) == 1 or all(n >= 2 * sh.BZ_HALO for _, n in r[:-1])
        for lo, n in r:
            off, ln = sh.bzip2_window(total, lo, n)
            assert off == lo and ln == min(total - lo, n + sh.BZ_HALO)
