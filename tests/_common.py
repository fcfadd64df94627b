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
        if not os.path.exists(p) or os.path.getmtime(p) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", p, src], check=True)
        H = ctypes.CDLL(p)
        H.hc_chunked_tokens.restype = ctypes.c_uint64
        H.hc_chunked_tokens.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int)]
        H.hc_header_bits.restype = ctypes.c_uint32
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
        _cache["m"] = M
    return _cache["m"]


def product():
    if "z" not in _cache:
        _cache["z"] = importlib.import_module("zip-ada_amd")
    return _cache["z"]


def silesia_mix(n, class_mask=0x1F, offset=0, seed=0x5A1E51A):
    b = np.zeros(n, dtype=np.uint8)
    if n:
        mixlib().zada_silesia_mix(seed, class_mask, offset, n, b.ctypes.data)
    return b.tobytes()


def oracle_tokens(data, method):
    t = np.zeros(len(data) + 8, dtype=np.uint32)
    k = oracle().zo_lz77_tokens(data, len(data), LEVEL[method], t.ctypes.data, len(t))
    return t[:k]


def oracle_deflate(data, method, blocks=None):
    """Returns (rc, stream bytes, running crc).  rc 1 = Compression_inefficient."""
    n = len(data)
    out = ctypes.create_string_buffer(n + 64)
    ol = ctypes.c_uint64(0)
    crc = ctypes.c_uint32(0xFFFFFFFF)
    cb = None
    if blocks is not None:
        def tr(_u, kind, a, b, c, d):
            if kind == 2:
                blocks.append((a, b, c, d))
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
    return cases
