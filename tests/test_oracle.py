"""CPU tests of the ORACLE (the checker): pins it against everything available without the Ada
binary -- zlib's LZ77 decisions, independent inflaters, an independent optimal-code computation,
hand-derivable known answers and the committed golden fixtures."""
import ctypes
import hashlib
import io
import json
import os
import zipfile
import zlib

import numpy as np
import pytest

from _common import (GOLDEN, LEVEL, METHODS, ROOT, TUNE, edge_inputs, oracle, oracle_deflate, oracle_tokens, oracle_zip,
                     position_tokens, silesia_mix, zlibpin)

FIXTURE_FILES = ("sample.xls", "sample.jpg", "sample_pgm_100k.bin")


def _fixture(name):
    return open(os.path.join(GOLDEN, name), "rb").read()


# ---------------------------------------------------------------- LZ77 stage vs libz (independent pin)
@pytest.mark.parametrize("fname", FIXTURE_FILES)
@pytest.mark.parametrize("level", (6, 8, 10))
def test_lz77_tokens_equal_committed_zlib_tokens(fname, level):
    """lz77.adb:460-943 == zlib deflate_slow under deflateTune(row of lz77.adb:534-546)."""
    d = _fixture(fname)
    z = np.load(os.path.join(GOLDEN, "zlib_tokens_%s_L%d.npz" % (fname.replace(".", "_"), level)))["tokens"]
    method = {6: 8, 8: 9, 10: 10}[level]
    pt = position_tokens(oracle_tokens(d, method), len(d))
    known = z != 0xFFFFFFFF                       # zlib hides tokens inside its stored blocks
    assert known.any()
    assert (z[known] == pt[known]).all()


@pytest.mark.parametrize("level", (4, 6, 8, 9, 10))
def test_lz77_tokens_equal_live_zlib(level):
    """Same pin, run live against the libz of this image on generated data (crosses the 32 KiB slide)."""
    P = zlibpin()
    # (the last three: the benchmark stream since round 5, silesia_mix_v2 -- every class, across segment boundaries)
    for d in (silesia_mix(200000), silesia_mix(70000, class_mask=4), silesia_mix(66000, class_mask=2, offset=65536 * 7), b"ab" * 50000, bytes(70000),
              silesia_mix(300000, version=2), silesia_mix(140000, class_mask=0x19, offset=65536 * 3 - 7000, version=2), silesia_mix(100000, class_mask=6, offset=65536 * 11, version=2)):
        z = np.zeros(len(d), dtype=np.uint32)
        assert P.zp_zlib_position_tokens(d, len(d), *TUNE[level], z.ctypes.data) == 0
        t = np.zeros(len(d) + 8, dtype=np.uint32)
        k = oracle().zo_lz77_tokens(d, len(d), level, t.ctypes.data, len(t))
        pt = position_tokens(t[:k], len(d))
        known = z != 0xFFFFFFFF
        assert (z[known] == pt[known]).all()


# ---------------------------------------------------------------- entropy stage: independent decoders
@pytest.mark.parametrize("method", METHODS)
def test_roundtrip_edge_inputs(method):
    for name, d in edge_inputs().items():
        rc, z, crc = oracle_deflate(d, method)
        assert crc ^ 0xFFFFFFFF == zlib.crc32(d), name
        if rc == 0:
            assert len(z) < len(d)
            assert zlib.decompress(z, -15) == d, name
        else:
            assert rc == 1, name


def test_golden_digests():
    """The oracle still produces the committed (self-pinned) streams."""
    dig = json.load(open(os.path.join(GOLDEN, "deflate_digests.json")))
    cases = dict(edge_inputs())
    for f in FIXTURE_FILES:
        cases[f] = _fixture(f)
    assert len(dig) == len(cases) * len(METHODS)
    for key, want in dig.items():
        name, m = key.rsplit("|", 1)
        d = cases[name]
        assert hashlib.sha256(d).hexdigest() == want["in_sha256"], name
        rc, z, crc = oracle_deflate(d, int(m))
        assert rc == want["rc"] and len(z) == want["size"], key
        if rc == 0:
            assert hashlib.sha256(z).hexdigest() == want["sha256"], key


def test_token_stage_and_full_stage_agree():
    """zo_deflate == zo_deflate_from_tokens(zo_lz77_tokens): the two stages are separable."""
    O = oracle()
    for d in (silesia_mix(150000), silesia_mix(70000, class_mask=16)):
        for m in (8, 9, 10):
            tok = oracle_tokens(d, m)
            out = ctypes.create_string_buffer(len(d) + 64)
            ol = ctypes.c_uint64(0)
            rc = O.zo_deflate_from_tokens(d, len(d), tok.ctypes.data, len(tok), m, out, len(d) + 64, ctypes.byref(ol), None, None)
            rc2, z, _ = oracle_deflate(d, m)
            assert rc == rc2 and (rc != 0 or out.raw[:ol.value] == z)


# ---------------------------------------------------------------- hand-derivable known answers
def test_empty_input_is_a_fake_fixed_block_then_store():
    """zip-compress-deflate.adb:1618-1634: no block was marked final -> 1, 01, EOB(0000000) = 03 00;
    2 >= 0 -> Compression_inefficient -> stored (zip-compress.adb:479-486, 224-237)."""
    O = oracle()
    out = ctypes.create_string_buffer(64)
    ol = ctypes.c_uint64(0)
    crc = ctypes.c_uint32(0xFFFFFFFF)
    # run the encoder "as if" the size were unknown is not possible; inspect via a 3-byte input instead
    rc, z, _ = oracle_deflate(b"", 10)
    assert rc == 1
    ol2 = ctypes.c_uint64(0); c2 = ctypes.c_uint32(0); zt = ctypes.c_uint16(9)
    assert O.zo_compress_data(b"", 0, 10, out, 64, ctypes.byref(ol2), ctypes.byref(c2), ctypes.byref(zt)) == 0
    assert ol2.value == 0 and zt.value == 0 and c2.value == 0


def test_deflate_fixed_single_literal():
    """Deflate_Fixed on b'a': bits 1,01 then literal 0x61 (fixed code 0x30+0x61 = 10010001, MSB first),
    EOB 0000000 -> bytes 4B 04 00 (hand-assembled from RFC 1951 3.2.6)."""
    rc, z, _ = oracle_deflate(b"a" * 1, 6)
    # 1 byte input: 3 bytes out >= 1 -> inefficient; check the stream through a larger cap
    assert rc == 1
    rc, z, _ = oracle_deflate(b"a" * 40, 6)
    assert rc == 0 and zlib.decompress(z, -15) == b"a" * 40
    assert z[0] & 7 == 0b011          # BFINAL = 1, BTYPE = 01


def test_crc32_matches_zlib():
    O = oracle()
    d = silesia_mix(100001)
    assert (O.zo_crc32_update(0xFFFFFFFF, d, len(d)) ^ 0xFFFFFFFF) == zlib.crc32(d)


# ---------------------------------------------------------------- length-limited codes
def _optimal_cost(freq, limit):
    """Independent check: minimum of sum f*l over prefix codes with l <= limit (textbook
    package-merge on (weight) items, no tie-breaking needed for the COST)."""
    w = sorted(f for f in freq if f > 0)
    n = len(w)
    if n <= 1:
        return sum(w)
    packages = []
    for _ in range(limit):
        merged = sorted(w + packages)
        packages = [merged[i] + merged[i + 1] for i in range(0, len(merged) - 1, 2)]
    # cost = sum of the 2n-2 smallest items of the last merged list
    return sum(sorted(merged)[:2 * n - 2])


def test_llhc_golden_vectors():
    vec = json.load(open(os.path.join(GOLDEN, "llhc_vectors.json")))
    assert len(vec) == 4
    for v in vec:
        f = np.array(v["freq"], dtype=np.uint64)
        bl = np.zeros(len(f), dtype=np.int32)
        assert oracle().zo_llhc(f.ctypes.data, len(f), v["max_bits"], bl.ctypes.data) == 0
        assert bl.tolist() == v["lengths"]
        assert max(bl) <= v["max_bits"]
        assert sum(2.0 ** -int(x) for x in bl if x > 0) == 1.0
        assert int((f.astype(np.int64) * bl).sum()) == _optimal_cost(v["freq"], v["max_bits"])


def test_llhc_random_is_optimal_and_complete():
    rs = np.random.RandomState(11)
    for it in range(300):
        n, mb = ((288, 15), (32, 15), (19, 7))[it % 3]
        f = (rs.pareto(1.2, n) * 5).astype(np.uint64) * (rs.rand(n) < 0.8)
        bl = np.zeros(n, dtype=np.int32)
        assert oracle().zo_llhc(f.ctypes.data, n, mb, bl.ctypes.data) == 0
        nz = int((f > 0).sum())
        assert ((bl > 0) == (f > 0)).all() and bl.max() <= mb
        if nz >= 2:
            assert sum(2.0 ** -int(x) for x in bl if x > 0) == 1.0
            assert int((f.astype(np.int64) * bl).sum()) == _optimal_cost(f.tolist(), mb)
        elif nz == 1:
            assert bl.sum() == 1


# ---------------------------------------------------------------- container (Zip.Create) bytes
def test_zip_archive_is_readable_by_zipfile_and_has_reference_constants():
    entries = [("a/text.txt", silesia_mix(50000, class_mask=1)), ("b\\rand.bin", bytes(np.random.RandomState(3).randint(0, 256, 3000).astype(np.uint8))), ("empty", b"")]
    z = oracle_zip(entries, 10)
    zf = zipfile.ZipFile(io.BytesIO(z))
    assert zf.testzip() is None
    infos = zf.infolist()
    assert [i.filename for i in infos] == ["a/text.txt", "b/rand.bin", "empty"]          # Unixify, zip-create.adb:181-192
    assert [i.compress_type for i in infos] == [8, 0, 0]                                 # store fallback for random / empty
    for (name, data), i in zip(entries, infos):
        assert zf.read(i) == data
        assert i.create_version == 23 and i.extract_version == 10                        # zip-create.adb:126, 131
        assert i.flag_bits == 0x0800                                                     # UTF-8 names (tools/zipada.adb:131)
    assert z[10:14] == (16789 * 65536).to_bytes(4, "little")                             # zip_streams.ads:223 default time


# ---------------------------------------------------------------- the parity matrix reaches every decision
def test_every_block_format_is_byte_compared():
    """The GPU parity tests compare bytes only when rc = 0.  This keeps the matrix honest: over edge_inputs() x the
    Taillaule methods the ORACLE takes each of the five ways of Send_as_block (zip-compress-deflate.adb:1243-1268)
    in a stream with rc = 0, and the byte-changing quirks fire: a fixed block in mid stream followed by a
    recycled one (:1223-1226), a fixed block opened after a dynamic one (:1108-1121), stored blocks inside a
    stream and two of 65 536 atoms (the halving of :1024-1038), the null-slice cut at atom 750 of even
    flushes for Deflate_3 but not Deflate_2 (:1372), an atom count that is an exact multiple of
    65 536 (:1617-1621, fake final fixed block)."""
    seen = {}
    per_case = {}
    for name, d in edge_inputs().items():
        for m in (7, 8, 9, 10):
            ob, cuts = [], []
            rc, z, _ = oracle_deflate(d, m, ob, cuts)
            if rc != 0:
                continue
            per_case[(name, m)] = (ob, cuts, z)
            for b in ob:
                seen.setdefault(b[2], (name, m))
    assert sorted(seen) == [0, 1, 2, 3, 4], seen
    ob, cuts, z = per_case[("fixedlike_mix", 10)]
    seq = [b[2] for b in ob]
    assert any(a == 1 and b == 4 for a, b in zip(seq, seq[1:])), seq            # recycle after fixed, in mid stream
    for key in (("flush_tail_m7", 7), ("flush_tail_m8", 8), ("flush_tail_m9", 9), ("flush_tail_m9", 10)):
        ob = per_case[key][0]
        assert [(b[0], b[1], b[2]) for b in ob[-2:]] == [(0, 65536, 2), (65536, 7, 1)], (key, ob[-2:])
    for m in (9, 10):
        ob, _, z = per_case[("flush_exact_m9", m)]
        assert sum(b[1] for b in ob) == 65536 and z[-2:] != b"" and (z[-1] or z[-2])   # ends with the fake fixed block
    for m in (7, 8, 9, 10):
        ob = per_case[("text_rand_text", m)][0]
        assert sum(1 for b in ob if b[2] == 0 and b[1] == 65536) >= 2, (m, ob)    # > 65 535 bytes each: halved on emission
    c10 = [a for a, _ in per_case[("copies_1500k", 10)][1]]
    c9 = [a for a, _ in per_case[("copies_1500k", 9)][1]]
    evens = [131072 * k + 750 for k in range(0, 3)]
    assert all(e in c10 for e in evens) and not any(e in c9 for e in evens), (c10, c9)


def test_pin_harness_rehearsal_with_a_stub_zipada(tmp_path):
    """oracle/pin_with_gnat.sh is what would pin the oracle to the Ada binary on a box with GNAT.  Its second half (pin_compare.py:
    zipada's option letters, local-header parsing, stored entries, digest comparison) is rehearsed here against oracle/zipada_stub.py,
    a stand-in with zipada's command line that compresses with the oracle itself -- so that the harness is known to work the day a
    GNAT box exists.  Nothing is pinned by this (the script says so)."""
    import subprocess
    import sys
    env = dict(os.environ, PIN_LIMIT="3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "pin_compare.py"), "%s %s" % (sys.executable, os.path.join(ROOT, "oracle", "zipada_stub.py")),
                        str(tmp_path), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 different" in r.stdout and "REHEARSAL" in r.stdout and "PINNED" not in r.stdout
    assert "a-cgcaso.adb" in r.stdout
