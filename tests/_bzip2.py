"""BZip2 helpers for the tests: ctypes bindings of the oracle's BZip2 half (checker) and of the product's stage hook."""
import ctypes

import numpy as np

from _common import oracle, product, edge_inputs


class BlockInfo(ctypes.Structure):
    _fields_ = [("rle_n", ctypes.c_int32), ("bwt_index", ctypes.c_int32), ("mtf_n", ctypes.c_int32), ("selector_count", ctypes.c_int32),
                ("coders", ctypes.c_int32), ("max_code_len", ctypes.c_int32), ("sample_width", ctypes.c_int32), ("alphabet", ctypes.c_int32),
                ("block_crc", ctypes.c_uint32), ("pad", ctypes.c_uint32), ("bits", ctypes.c_uint64)]


BZ_TRACE = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int)


def bz_inputs():
    """The parity matrix of the BZip2 half: the Deflate edge set plus inputs on which the reference keeps each of its
    splitting tactics (bzip2-encoding.adb:1214-1345): parts_4 wins on `copies_1500k` (edge set), segmented_1 on
    `seg1_az_digits`, segmented_2 on `seg2_alphabets` (the oracle's trace says so; test_bzip2_oracle.py asserts it)."""
    cases = dict(edge_inputs())
    rng = np.random.default_rng(11)
    text = cases["text_rand_text"][:300000]

    def letters(r, n, k, base=97):
        return r.integers(base, base + k, n, dtype=np.uint8)
    cases["seg1_az_digits"] = bytes(np.concatenate([letters(rng, 150000, 26), rng.integers(48, 58, 150000, dtype=np.uint8), letters(rng, 100000, 4)]))
    rng = np.random.default_rng(12)
    v = None
    for _trial in range(3):
        parts = []
        for _j in range(rng.integers(3, 7)):
            k = int(rng.choice([2, 3, 4, 6, 8, 12, 16, 24, 26]))
            n = int(rng.integers(30000, 120000))
            parts.append(letters(rng, n, k, int(rng.choice([48, 65, 97]))))
        v = bytes(np.concatenate(parts))
    cases["seg2_alphabets"] = v
    del text
    return cases


def bz_oracle():
    O = oracle()
    if not getattr(O, "_bz_ready", False):
        vp = ctypes.c_void_p
        O.zo_bzip2_encode.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int64, vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), vp, vp]
        O.zo_bzip2.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
        O.zo_bz2_crc.restype = ctypes.c_uint32
        O.zo_bz2_crc.argtypes = [ctypes.c_char_p, ctypes.c_uint64]
        O.zo_bz2_block.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.POINTER(BlockInfo), vp, ctypes.c_uint64]
        O.zo_bz2_segments.restype = ctypes.c_int32
        O.zo_bz2_segments.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int, vp, ctypes.c_int32]
        O._bz_ready = True
    return O


def oracle_encode(data, option=2, known=True):
    """-> (stream bytes, [(raw_start, raw_len, tactic, sub_blocks)])"""
    O = bz_oracle()
    cap = len(data) * 2 + 2_000_000
    out = ctypes.create_string_buffer(cap)
    n = ctypes.c_uint64()
    ev = []
    cb = BZ_TRACE(lambda u, a, b, t, s: ev.append((a, b, t, s)))
    rc = O.zo_bzip2_encode(data, len(data), option, len(data) if known else -1, out, cap, ctypes.byref(n), ctypes.cast(cb, ctypes.c_void_p), None)
    assert rc == 0, rc
    return out.raw[:n.value], ev


def oracle_block(raw, option=2, want_bits=False):
    """Stages of one Encode_Block -> dict(rle, bwt, mtf, selectors, lens, info, bits)."""
    O = bz_oracle()
    n = len(raw)
    rle = np.zeros(n + n // 4 + 16, np.uint8)
    bwt = np.zeros(n + n // 4 + 16, np.uint8)
    mtf = np.zeros(2 * (n + n // 4) + 16, np.uint16)
    sel = np.zeros((n + n // 4) // 50 + 8, np.uint8)
    lens = np.zeros(6 * 258, np.uint8)
    info = BlockInfo()
    bits = np.zeros(2 * n + 1_000_000 if want_bits else 1, np.uint8)
    rc = O.zo_bz2_block(bytes(raw), n, option, rle.ctypes.data, bwt.ctypes.data, mtf.ctypes.data, sel.ctypes.data, lens.ctypes.data, ctypes.byref(info),
                        bits.ctypes.data if want_bits else None, bits.size)
    assert rc == 0, rc
    return dict(rle=rle[:info.rle_n], bwt=bwt[:info.rle_n], mtf=mtf[:info.mtf_n], selectors=sel[:info.selector_count], lens=lens.reshape(6, 258),
                info=info, bits=bits[:(info.bits + 7) // 8] if want_bits else None)


def _fetch(L, enc, name, dtype, cap_items):
    L.zada_bz2_fetch.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    buf = np.zeros(max(int(cap_items), 4), dtype)
    n = ctypes.c_uint64()
    rc = L.zada_bz2_fetch(enc.ctx, name.encode(), buf.ctypes.data, buf.nbytes, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError("zada_bz2_fetch(%s) rc=%d (needs %d bytes, has %d)" % (name, rc, n.value, buf.nbytes))
    return buf[:n.value // buf.itemsize]


def product_stages(enc, data, starts, lens, option=2, stages=3):
    """Sub-blocks of `data` through the product's stage hooks: 1 = RLE_1 / CRC / BWT, 2 = + MTF / RLE_2, 3 = + entropy coders and bits."""
    Z = product()
    L = Z.load_library()
    vp = ctypes.c_void_p
    L.zada_bz2_run.argtypes = [vp, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, vp, vp, ctypes.c_int, ctypes.c_int]
    L.zada_bz2_fetch.argtypes = [vp, ctypes.c_char_p, vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    nsb = len(starts)
    st = np.asarray(starts, np.uint64)
    ln = np.asarray(lens, np.uint32)
    rc = L.zada_bz2_run(enc.ctx, bytes(data), len(data), nsb, st.ctypes.data, ln.ctypes.data, option, stages)
    if rc != 0:
        raise RuntimeError("zada_bz2_run rc=%d: %s" % (rc, L.zada_last_error(enc.ctx).decode()))
    cap = int(sum(int(x) + int(x) // 4 + 8 for x in lens)) + 64
    rle_n = _fetch(L, enc, "n", np.uint32, nsb)
    off = np.concatenate([[0], np.cumsum(rle_n)]).astype(np.int64)
    bwt = _fetch(L, enc, "bwt", np.uint8, cap)
    R = dict(rle_n=rle_n, bwt_index=_fetch(L, enc, "bwt_index", np.uint32, nsb), crc=_fetch(L, enc, "crc", np.uint32, nsb),
             inuse=_fetch(L, enc, "inuse", np.uint32, 8 * nsb).reshape(nsb, 8), info=_fetch(L, enc, "info", np.uint32, 4),
             bwt=[bwt[off[i]:off[i + 1]] for i in range(nsb)])
    if stages == 1:
        rle = _fetch(L, enc, "rle", np.uint8, cap)
        R["rle"] = [rle[off[i]:off[i + 1]] for i in range(nsb)]
    if stages >= 2:
        R["mtf_n"] = _fetch(L, enc, "mtf_n", np.uint32, nsb)
        soff = _fetch(L, enc, "soff", np.uint32, nsb).astype(np.int64)
        sym = _fetch(L, enc, "sym", np.uint16, cap + nsb + 16)
        R["mtf"] = [sym[soff[i]:soff[i] + int(R["mtf_n"][i])] for i in range(nsb)]
    if stages >= 3:
        res = _fetch(L, enc, "res", np.uint32, 8 * nsb).reshape(nsb, 8)
        so = _fetch(L, enc, "sel_off", np.uint32, nsb + 1).astype(np.int64)
        sel = _fetch(L, enc, "sel", np.uint8, cap // 50 + 2 * nsb + 64)
        lens_t = _fetch(L, enc, "lens", np.uint8, 6 * 260 * nsb).reshape(nsb, 6, 260)
        woff = _fetch(L, enc, "woff", np.uint32, nsb + 1).astype(np.int64)
        words = _fetch(L, enc, "words", np.uint32, int(cap) // 2 + 300000 * nsb // 100 + 4096 * nsb + 1024)
        R.update(res=res, selectors=[sel[so[i]:so[i] + int(res[i, 3])] for i in range(nsb)], lens=lens_t[:, :, :258],
                 bits=[words[woff[i]:woff[i + 1]].byteswap().view(np.uint8)[:(int(res[i, 7]) + 7) // 8] for i in range(nsb)])
    return R
