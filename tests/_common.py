"""Shared helpers for the test-suite: loads the oracle (checker), the zlib pin tool, the hostcheck
library and -- for GPU tests -- the product.  Only tests / smoke / bench's cpu_baseline use the oracle."""
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LEVEL = {6: 4, 7: 0, 8: 6, 9: 8, 10: 10}            # LZ77_choice, zip-compress-deflate.adb:1573-1579
TUNE = {4: (4, 4, 16, 16), 6: (8, 16, 128, 128), 8: (32, 128, 258, 1024), 9: (32, 258, 258, 4096), 10: (34, 258, 258, 4096)}  # lz77.adb:534-546
METHODS = (6, 7, 8, 9, 10)
TRACE = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64)


def _make(path, target=None):
    cmd = ["make", "-s", "-C", path] + ([target] if target else [])
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)


_cache = {}


def oracle():
    if "o" not in _cache:
        p = os.path.join(ROOT, "oracle", "libzada_oracle.so")
        if not os.path.exists(p):
            _make(os.path.join(ROOT, "oracle"))
        O = ctypes.CDLL(p)
        O.zo_lz77_tokens.restype = ctypes.c_uint64
        O.zo_lz77_tokens.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64]
        O.zo_deflate.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64,
                                 ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        O.zo_deflate_from_tokens.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p, ctypes.c_void_p]
        O.zo_compress_data.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64,
                                       ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint16)]
        O.zo_llhc.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        O.zo_crc32_update.restype = ctypes.c_uint32
        O.zo_crc32_update.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint64]
        O.zo_zip_create.restype = ctypes.c_void_p
        O.zo_zip_create.argtypes = [ctypes.c_int]
        O.zo_zip_add.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int]
        O.zo_zip_finish.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
        O.zo_zip_add_compressed.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64,
                                            ctypes.c_int, ctypes.c_uint32, ctypes.c_int]
        O.zo_zip_set_bias.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        O.zo_zip_free.argtypes = [ctypes.c_void_p]
        _cache["o"] = O
    return _cache["o"]


def zlibpin():
    if "p" not in _cache:
        p = os.path.join(ROOT, "oracle", "libzada_zlibpin.so")
        if not os.path.exists(p):
            _make(os.path.join(ROOT, "oracle"))
        P = ctypes.CDLL(p)
        P.zp_zlib_position_tokens.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        P.zp_stream_position_tokens.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p]
        _cache["p"] = P
    return _cache["p"]


def hostcheck():
    if "h" not in _cache:
        d = os.path.join(ROOT, "tests", "hostcheck")
        p = os.path.join(d, "libzada_hostcheck.so")
        src = os.path.join(d, "hostcheck.cpp")
        hdr = os.path.join(ROOT, "zip-ada_amd", "csrc", "zada_logic.h")
        hdr2 = os.path.join(d, "hostcheck_logic.h")
        hdr3 = os.path.join(ROOT, "zip-ada_amd", "csrc", "zada_bt4.h")
        if not os.path.exists(p) or os.path.getmtime(p) < max(os.path.getmtime(f) for f in (src, hdr, hdr2, hdr3)):
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", p, src], check=True)
        H = ctypes.CDLL(p)
        H.hc_chunked_tokens.restype = ctypes.c_uint64
        H.hc_chunked_tokens.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int)]
        H.hc_header_bits.restype = ctypes.c_uint32
        H.hc_demand_loop_tokens.restype = ctypes.c_uint64
        H.hc_demand_loop_tokens.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32,
                                            ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        H.hc_bt4_sets.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32]
        H.hc_bt4_sets_segments.argtypes = H.hc_bt4_sets.argtypes + [ctypes.c_uint32]
        H.hc_bt4_reads_behind_a_gap.argtypes = [ctypes.c_uint64, ctypes.c_int64]
        _cache["h"] = H
    return _cache["h"]


def mixlib():
    """The synthetic-corpus generator alone (no HIP dependency), for CPU tests."""
    if "m" not in _cache:
        d = os.path.join(ROOT, "zip-ada_amd", "csrc")
        p = os.path.join(ROOT, "tests", "hostcheck", "libzada_mix.so")
        src = os.path.join(d, "silesia_mix.c")
        if not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(src):
            subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", p, src, "-lm"], check=True)
        M = ctypes.CDLL(p)
        M.zada_silesia_mix.argtypes = [ctypes.c_uint64, ctypes.c_uint, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        M.zada_silesia_mix_v2.argtypes = M.zada_silesia_mix.argtypes
        _cache["m"] = M
    return _cache["m"]


def product():
    if "z" not in _cache:
        _cache["z"] = importlib.import_module("zip-ada_amd")
    return _cache["z"]


def silesia_mix(n, class_mask=0x1F, offset=0, seed=0x5A1E51A, version=1):
    """version 1: what the golden digests were taken on; version 2: segments seeded independently (the benchmark stream since round 5)."""
    b = np.zeros(n, dtype=np.uint8)
    if n:
        M = mixlib()
        (M.zada_silesia_mix_v2 if version >= 2 else M.zada_silesia_mix)(seed, class_mask, offset, n, b.ctypes.data)
    return b.tobytes()


def oracle_tokens(data, method):
    t = np.zeros(len(data) + 8, dtype=np.uint32)
    k = oracle().zo_lz77_tokens(data, len(data), LEVEL[method], t.ctypes.data, len(t))
    return t[:k]


def oracle_deflate(data, method, blocks=None, cuts=None, similar=None, bitpos=None):
    """Returns (rc, stream bytes, running crc).  rc 1 = Compression_inefficient.
    blocks / cuts / similar collect the oracle's trace events (the reference's compile-time trace,
    zip-compress-deflate.adb:83-90): (first atom, atoms, format, bits) / (atom, level) / (atom, distance, threshold, similar)."""
    n = len(data)
    out = ctypes.create_string_buffer(n + 64)
    ol = ctypes.c_uint64(0)
    crc = ctypes.c_uint32(0xFFFFFFFF)
    cb = None
    if blocks is not None or cuts is not None or similar is not None or bitpos is not None:
        def tr(_u, kind, a, b, c, d):
            if kind == 2 and blocks is not None:
                blocks.append((a, b, c, d))
            elif kind == 1 and cuts is not None:
                cuts.append((a, b))
            elif kind == 3 and similar is not None:
                similar.append((a, b, c, d))
            elif kind == 4 and bitpos is not None:
                bitpos.append((a, b))
        cb = TRACE(tr)
    rc = oracle().zo_deflate(data, n, method, out, n + 64, ctypes.byref(ol), ctypes.byref(crc), None, None,
                             ctypes.cast(cb, ctypes.c_void_p) if cb else None, None)
    return rc, out.raw[:ol.value], crc.value


def oracle_zip(entries, method):
    """entries: list of (name, bytes).  Returns the archive bytes Zip.Create would write."""
    O = oracle()
    a = O.zo_zip_create(method)
    try:
        for name, data in entries:
            rc = O.zo_zip_add(a, name.encode("utf-8"), data, len(data), 16789 * 65536, 1)
            assert rc == 0, rc
        p = ctypes.c_void_p()
        ln = ctypes.c_uint64()
        assert O.zo_zip_finish(a, ctypes.byref(p), ctypes.byref(ln)) == 0
        return ctypes.string_at(p.value, ln.value)
    finally:
        O.zo_zip_free(a)


def oracle_zip_compressed(entries, bias=0):
    """entries: list of (name, payload, crc, uncompressed size, zip_type) made elsewhere.  The archive bytes
    Zip.Create would write for them (bias: pretended number of bytes in front of the archive)."""
    O = oracle()
    a = O.zo_zip_create(10)
    try:
        O.zo_zip_set_bias(a, bias)
        for name, payload, crc, usize, zt in entries:
            assert O.zo_zip_add_compressed(a, name.encode("utf-8"), payload, len(payload), crc, usize, zt, 16789 * 65536, 1) == 0
        p = ctypes.c_void_p()
        ln = ctypes.c_uint64()
        assert O.zo_zip_finish(a, ctypes.byref(p), ctypes.byref(ln)) == 0
        return ctypes.string_at(p.value, ln.value)
    finally:
        O.zo_zip_free(a)


def position_tokens(tokens, n):
    """Token stream -> array indexed by input position (0xFFFFFFFE where no token starts)."""
    out = np.full(n, 0xFFFFFFFE, dtype=np.uint32)
    lens = np.where(tokens & 0x80000000, (tokens >> 16) & 0x1FF, 1).astype(np.int64)
    pos = np.concatenate(([0], np.cumsum(lens)[:-1])) if len(tokens) else np.zeros(0, dtype=np.int64)
    assert (int(lens.sum()) if len(tokens) else 0) == n
    out[pos] = tokens
    return out


def edge_inputs():
    """The reference's own edge-case recipe: tiny files, random, restricted alphabet
    (test/test_za.hac:121-137), sizes around the window slide (test/several_sizes.adb:77-89)."""
    rs = np.random.RandomState(1)
    cases = {}
    for sz in (0, 1, 2, 3, 4, 5, 100, 257, 258, 259, 260, 4095, 4096, 4097, 32505, 32506, 32507, 32508, 32767, 32768, 32769,
               65273, 65274, 65275, 65276, 65535, 65536, 65537):
        cases["text_%d" % sz] = silesia_mix(sz, class_mask=1)
    cases["zeros_100000"] = bytes(100000)
    cases["az_77777"] = bytes(rs.randint(65, 91, 77777).astype(np.uint8))
    cases["random_66666"] = bytes(rs.randint(0, 256, 66666).astype(np.uint8))
    for k in range(0, 101, 17):
        cases["random_%d" % k] = bytes(rs.randint(0, 256, k).astype(np.uint8))
    cases["ab_80000"] = b"ab" * 40000
    cases["abc_x"] = b"abc" * 30000 + b"x" + b"abc" * 100
    cases["mix_300000"] = silesia_mix(300000)
    cases["mix_1m_off"] = silesia_mix((1 << 20) + 1, offset=3 * 65536)
    cases.update(format_inputs())
    return cases


_LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]


def _fixed_like(natoms, seed, pm=0.22):
    """LZ atoms drawn i.i.d. from the distribution the FIXED Huffman code is optimal for (literals 0..143 at 2^-8,
    144..255 at 2^-9, length codes at 2^-7 / 2^-8, distance codes uniform), realised as bytes: copies from a
    random history.  Returns (bytes, history)."""
    r = np.random.RandomState(seed)
    out = bytearray(r.randint(0, 256, 40000).astype(np.uint8).tobytes())
    start = len(out)
    pl = np.array([2.0 ** -8] * 144 + [2.0 ** -9] * 112); pl /= pl.sum()
    plen = np.array([2.0 ** -7] * 23 + [2.0 ** -8] * 6); plen /= plen.sum()
    for _ in range(natoms):
        if r.rand() < pm:
            c = r.choice(29, p=plen)
            L = min(_LEN_BASE[c] + (r.randint(0, 1 << _LEN_EXTRA[c]) if _LEN_EXTRA[c] else 0), 258)
            dc = r.randint(0, 30)
            if dc < 4:
                d = dc + 1
            else:
                e = (dc >> 1) - 1
                d = ((2 + (dc & 1)) << e) + 1 + r.randint(0, 1 << e)
            d = min(max(d, L + 1) if d < L else d, len(out), 32000)
            s = len(out) - d
            for i in range(L):
                out.append(out[s + i])
        else:
            out.append(int(r.choice(256, p=pl)))
    return bytes(out[start:]), bytes(out[:start])


def _copies(n, seed):
    """Random bytes with a 4..12-byte copy every ~20 bytes at a log-uniform distance 4..32000: ~230 distinct
    literals per window, so that the empty-statistics descriptor of the null-slice quirk
    (zip-compress-deflate.adb:1372) is NOT similar to the data and Deflate_3 cuts at atom 750 of
    every even flush."""
    r = np.random.RandomState(seed)
    out = bytearray(r.randint(0, 256, 64).astype(np.uint8).tobytes())
    while len(out) < n:
        out += r.randint(0, 256, int(r.randint(8, 32))).astype(np.uint8).tobytes()
        L = int(r.randint(4, 13))
        d = min(int(np.exp(r.uniform(np.log(4), np.log(32000)))), len(out))
        s = len(out) - d
        for i in range(L):
            out.append(out[s + i])
    return bytes(out[:n])


def format_inputs():
    """Inputs that make the chooser of Send_as_block (zip-compress-deflate.adb:1222-1268) take every one of its five
    ways in streams with rc = 0 (tests/test_oracle.py::test_every_block_format_is_byte_compared asserts it):
    fixed wins (tiny repetitive inputs; a tiny last flush of unseen symbols after a dynamic block; a fixed block
    in mid stream followed by recycled ones), stored blocks inside a stream incl. two of 65 536 atoms (the halving
    of :1024-1038), the null-slice cut (:1372), an atom count that is an exact multiple of 65 536 (:1617-1621)."""
    rs = np.random.RandomState(7)
    cases = {}
    cases["abc_40"] = b"abc" * 40
    cases["a_40"] = b"a" * 40
    cases["abcdefghij_12"] = b"abcdefghij" * 12
    text = silesia_mix(1 << 20, class_mask=1)
    cases["text_rand_text"] = text[:300000] + bytes(rs.randint(0, 256, 200000).astype(np.uint8)) + silesia_mix(300000, class_mask=1, offset=1 << 20)
    cases["copies_1500k"] = _copies(1500000, 3)
    blob = bytearray()
    toff = 0
    for k in range(8):
        a, h = _fixed_like(6000, k)
        if k == 0:
            blob += h
        blob += a
        blob += text[toff:toff + 24000]
        toff += 24000
    cases["fixedlike_mix"] = bytes(blob)
    tail = bytes([1, 2, 3, 4, 5, 6, 7])                       # symbols the text never uses: not recyclable
    cases["flush_tail_m7"] = text[:65536] + tail              # Deflate_0: 65 536 atoms, then a flush of 7
    cases["flush_tail_m8"] = text[:364356] + tail             # Deflate_1: the same (atom counts found with the oracle)
    cases["flush_tail_m9"] = text[:372589] + tail             # Deflate_2 and Deflate_3
    cases["flush_exact_m9"] = text[:372589]                   # exactly 65 536 atoms: no last flush, fake final fixed block
    # the only earlier occurrence lies exactly MAX_DIST = 32 506 back: the reference accepts it as the head of the hash chain
    # (lz77.adb:850) but not behind it (:820); one byte nearer / farther for contrast; text so that chains are not empty
    for per in (32505, 32506, 32507):
        blk = bytes(rs.randint(0, 256, per - 2000).astype(np.uint8)) + text[500000:502000]
        cases["period_%d" % per] = blk * 3 + blk[:1234]
    return cases


def few_symbol_inputs():
    """2-, 4- and 16-symbol uniform random data: every position has thousands of candidates
    (lz77.adb:715-825 at chain 4096) -- the demand path of the match finder at its worst."""
    rs = np.random.RandomState(5)
    return {"sym2_4m": bytes((rs.randint(0, 2, 4 << 20) + 65).astype(np.uint8)),
            "sym4_4m": bytes((rs.randint(0, 4, 4 << 20) + 65).astype(np.uint8)),
            "sym16_6m": bytes((rs.randint(0, 16, 6 << 20) + 65).astype(np.uint8))}
