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


def lzma_symbols(stream):
    """A plain LZMA decoder that keeps the symbols: (decoded bytes, [(position, kind, distance, length)]) of a stream as the oracle / the
    product write it (5-byte header, end marker); kind 'L' literal (distance = the byte), 'S' short repeat, 'R' repeat match, 'M' match,
    'E' end marker.  Test tool: it says WHICH symbol of a stream is wrong, liblzma only says that one is."""
    buf = bytes(stream)
    props = buf[0]; lc = props % 9; lp = (props // 9) % 5; pb = props // 45
    st = {"rng": 0xFFFFFFFF, "code": int.from_bytes(buf[6:10], "big"), "pos": 10}

    def norm():
        if st["rng"] < (1 << 24):
            st["rng"] = (st["rng"] << 8) & 0xFFFFFFFF
            st["code"] = ((st["code"] << 8) | (buf[st["pos"]] if st["pos"] < len(buf) else 0)) & 0xFFFFFFFF
            st["pos"] += 1

    def bit(probs, i):
        p = probs[i]; bound = (st["rng"] >> 11) * p
        if st["code"] < bound:
            st["rng"] = bound; probs[i] = p + ((2048 - p) >> 5); norm(); return 0
        st["rng"] -= bound; st["code"] -= bound; probs[i] = p - (p >> 5); norm(); return 1

    def direct(n):
        r = 0
        for _ in range(n):
            st["rng"] >>= 1
            st["code"] = (st["code"] - st["rng"]) & 0xFFFFFFFF
            t = (0 - (st["code"] >> 31)) & 0xFFFFFFFF
            st["code"] = (st["code"] + (st["rng"] & t)) & 0xFFFFFFFF
            norm(); r = (r << 1) + ((t + 1) & 1)
        return r

    def tree(probs, off, nb):
        m = 1
        for _ in range(nb):
            m = (m << 1) + bit(probs, off + m)
        return m - (1 << nb)

    def rtree(probs, off, nb):
        m, v = 1, 0
        for i in range(nb):
            b = bit(probs, off + m); m = (m << 1) + b; v |= b << i
        return v

    P = lambda n: [1024] * n
    is_match, is_rep, g0, g1, g2, rep0long = P(12 * 16), P(12), P(12), P(12), P(12), P(12 * 16)
    slot, spec, align, lit = P(4 * 64), P(115), P(16), P(0x300 << (lc + lp))
    lens = [{"c": P(2), "low": P(16 * 8), "mid": P(16 * 8), "high": P(256)} for _ in range(2)]

    def dlen(L, ps):
        if bit(L["c"], 0) == 0:
            return tree(L["low"], ps * 8, 3)
        if bit(L["c"], 1) == 0:
            return 8 + tree(L["mid"], ps * 8, 3)
        return 16 + tree(L["high"], 0, 8)

    out, state, reps, syms = bytearray(), 0, [0, 0, 0, 0], []
    while True:
        ps = len(out) & ((1 << pb) - 1)
        if bit(is_match, state * 16 + ps) == 0:
            prev = out[-1] if out else 0
            off = 0x300 * (((len(out) & ((1 << lp) - 1)) << lc) + (prev >> (8 - lc)))
            sym = 1
            if state >= 7:
                mb = out[len(out) - reps[0] - 1]
                while sym < 0x100:
                    mbit = (mb >> 7) & 1; mb = (mb << 1) & 0xFF
                    b = bit(lit, off + ((1 + mbit) << 8) + sym); sym = (sym << 1) | b
                    if mbit != b:
                        break
            while sym < 0x100:
                sym = (sym << 1) | bit(lit, off + sym)
            syms.append((len(out), "L", sym & 0xFF, 1)); out.append(sym & 0xFF)
            state = 0 if state < 4 else state - 3 if state < 10 else state - 6
            continue
        if bit(is_rep, state):
            if bit(g0, state) == 0:
                if bit(rep0long, state * 16 + ps) == 0:
                    state = 9 if state < 7 else 11
                    syms.append((len(out), "S", reps[0] + 1, 1)); out.append(out[len(out) - reps[0] - 1])
                    continue
                d = reps[0]
            else:
                if bit(g1, state) == 0:
                    d = reps[1]; reps[1] = reps[0]
                else:
                    if bit(g2, state) == 0:
                        d = reps[2]
                    else:
                        d = reps[3]; reps[3] = reps[2]
                    reps[2] = reps[1]; reps[1] = reps[0]
                reps[0] = d
            ln = dlen(lens[1], ps) + 2; state = 8 if state < 7 else 11; kind = "R"
        else:
            ln = dlen(lens[0], ps) + 2; state = 7 if state < 7 else 10
            s = tree(slot, (ln - 2 if ln - 2 < 3 else 3) * 64, 6)
            if s < 4:
                d = s
            else:
                nb = (s >> 1) - 1; d = (2 | (s & 1)) << nb
                d += rtree(spec, d - s - 1, nb) if s < 14 else (direct(nb - 4) << 4) + rtree(align, 0, 4)
            if d == 0xFFFFFFFF:
                syms.append((len(out), "E", 0, 0))
                break
            reps[3] = reps[2]; reps[2] = reps[1]; reps[1] = reps[0]; reps[0] = d; kind = "M"
        if reps[0] + 1 > len(out):
            raise ValueError("distance %d at position %d" % (reps[0] + 1, len(out)))
        syms.append((len(out), kind, reps[0] + 1, ln))
        for _ in range(ln):
            out.append(out[len(out) - reps[0] - 1])
    return bytes(out), syms
