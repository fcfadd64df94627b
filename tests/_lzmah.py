"""LZMA helpers for the tests: ctypes bindings of the oracle's LZMA half (the checker) and a liblzma decoder for its streams."""
import ctypes
import lzma

import numpy as np

from _common import oracle, edge_inputs, silesia_mix

LZMA_METHODS = (15, 16, 17, 18)          # Compression_Method'Pos of LZMA_0 .. LZMA_3 (zip-compress.ads:93-98)


def lz_inputs(limit=400000):
    """The parity matrix of the LZMA half: the Deflate edge set (inputs up to `limit` bytes) plus a mixed corpus
    on which every variant of a DL code (literal + DL, DL + literal, expansion, split) is taken at Level_3."""
    cases = {k: bytes(v) for k, v in dict(edge_inputs()).items() if len(v) <= limit}
    cases["mix_256k"] = bytes(silesia_mix(256 * 1024, seed=3))
    rng = np.random.default_rng(21)
    cases["two_symbols_40k"] = bytes(rng.integers(0, 2, 40000, dtype=np.uint8) + 65)
    cases["zeros_100k"] = bytes(100000)
    return cases


def lz_oracle():
    O = oracle()
    if not getattr(O, "_lz_ready", False):
        vp = ctypes.c_void_p
        O.zo_lzma_encode.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int64, vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), vp]
        O.zo_lzma.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
        O.zo_bt4_match_sets.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int64, vp, vp, vp, ctypes.c_int]
        O._lz_ready = True
    return O


def oracle_lzma_encode(data, level, lc=3, lp=0, pb=2, end_marker=True, dictionary_size=None):
    """LZMA.Encoding.Encode -> (stream incl. the 5-byte header, the 8 choice counters)."""
    O = lz_oracle()
    data = bytes(data)
    cap = len(data) * 2 + 4096
    out = ctypes.create_string_buffer(cap)
    ol = ctypes.c_uint64()
    st = (ctypes.c_uint64 * 8)()
    rc = O.zo_lzma_encode(data, len(data), level, lc, lp, pb, int(end_marker), len(data) if dictionary_size is None else dictionary_size,
                          out, cap, ctypes.byref(ol), st)
    assert rc == 0, rc
    return out.raw[:ol.value], list(st)


def oracle_lzma(data, method):
    """Zip.Compress.LZMA_E -> (rc, Zip payload, CRC register)."""
    O = lz_oracle()
    data = bytes(data)
    cap = len(data) * 2 + 4096
    out = ctypes.create_string_buffer(cap)
    ol = ctypes.c_uint64()
    crc = ctypes.c_uint32(0xFFFFFFFF)
    rc = O.zo_lzma(data, len(data), method, out, cap, ctypes.byref(ol), ctypes.byref(crc))
    assert rc >= 0, rc
    return rc, out.raw[:ol.value], crc.value


def lzma_decode(stream, skip=0):
    """liblzma on a stream as LZMA.Encoding.Encode writes it (`skip` = 4 for the Zip payload's prefix)."""
    s = stream[skip:]
    props, ds = s[0], int.from_bytes(s[1:5], "little")
    lc, lp, pb = props % 9, (props // 9) % 5, props // 45
    d = lzma.LZMADecompressor(format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA1, "dict_size": ds, "lc": lc, "lp": lp, "pb": pb}])
    out = d.decompress(s[5:])
    assert d.eof, "no end-of-stream marker met"
    return out


BT4_SET = 50                             # most matches of one position (zada_bt4.h): two hash matches + Depth_Limit tree matches


def oracle_bt4_sets(data, dictionary_size=None):
    """The match sets of BT4_Algo.Read_One_and_Get_Matches at every position (oracle stage export) -> cnt [n], len [n, 50], dist [n, 50]."""
    O = lz_oracle()
    data = bytes(data)
    n = len(data)
    cnt = np.zeros(n, np.uint8); ln = np.zeros((n, BT4_SET), np.uint16); ds = np.zeros((n, BT4_SET), np.uint32)
    rc = O.zo_bt4_match_sets(data, n, n if dictionary_size is None else dictionary_size, cnt.ctypes.data, ln.ctypes.data, ds.ctypes.data, BT4_SET)
    assert rc == 0, rc
    return cnt, ln, ds


def sets_equal(a, b):
    """Two (cnt, len, dist) triples hold the same sets (slots beyond cnt are not compared)."""
    if not np.array_equal(a[0], b[0]):
        return False
    live = np.arange(BT4_SET)[None, :] < a[0][:, None]
    return bool(np.array_equal(a[1][live], b[1][live]) and np.array_equal(a[2][live], b[2][live]))
