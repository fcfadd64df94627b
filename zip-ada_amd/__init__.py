"""zip-ada_amd -- MI355X-native Deflate encoder behind the Zip-Ada `Zip.Compress` interface.

Host-side mirror of the reference's interface for the hot path (the reference is Ada; no Ada
toolchain exists in the build image, so the host side above the C ABI is written here and in
`csrc/`; see INTEGRATION.md for the Ada shim a maintainer would add):

    Compression_Method        zip_lib/zip-compress.ads:59-122   -> `Method`
    Zip.Compress.Deflate      zip_lib/zip-compress-deflate.ads:36-46 -> `Encoder.deflate`
    Zip.Compress.Compress_Data zip_lib/zip-compress.ads:169-180 -> `Encoder.compress_data`
    Zip.Create (Create_Archive / Add_Stream / Finish) zip_lib/zip-create.ads:75-211 -> `ZipCreate`

All compute runs in libzada_hip.so (hand-written HIP for gfx950).  There is no CPU fallback:
loading fails loudly when the library or a GPU is missing.
"""
import ctypes
import os
import struct

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZADA_LIB", os.path.join(_HERE, "libzada_hip.so"))   # ZADA_LIB: A/B builds of the same library


class Method:
    """Compression_Method'Pos, zip-compress.ads:59-122 (Deflation_Method subset)."""
    Store = 0
    Deflate_Fixed = 6
    Deflate_0 = 7
    Deflate_1 = 8
    Deflate_2 = 9
    Deflate_3 = 10
    BZip2_1, BZip2_2, BZip2_3 = 12, 13, 14
    LZMA_0, LZMA_1, LZMA_2, LZMA_3 = 15, 16, 17, 18


E_REFERENCE = -6     # zada.h ZADA_E_REFERENCE: LZMA_3, the reference's own matcher reports a match that is none on this entry


class ZadaError(RuntimeError):
    pass


class ReferenceDefect(ZadaError):
    """ZADA_E_REFERENCE: on this LZMA_3 entry the reference's BT4 matcher reports a match that is none (include/zada.h); the reference's own
    stream would not decode to the input, nothing was written."""


class CompressionInefficient(Exception):
    """zip-compress.ads:237 -- compressed size >= uncompressed size."""


class UserAbort(Exception):
    """zip-compress.ads:149."""


_lib = None


def load_library():
    """Loads libzada_hip.so.  Raises ZadaError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZadaError("libzada_hip.so is missing: run __graft_entry__.build() "
                        "(make -C zip-ada_amd/csrc). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 / libhsa-runtime64.  If this library pulled in
    # /opt/rocm's copy first, a later `import torch` would bring up a SECOND runtime in the same process, which on some
    # nodes cannot open the GPU any more ("No HIP GPUs are available").  Loaded after torch, libzada_hip.so binds to the
    # runtime torch has loaded (same soname).  PyTorch is only plumbing here (device buffers, torch.distributed).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, u64, u32p, u64p, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64), ctypes.c_int
    L.zada_create.restype = vp
    L.zada_create.argtypes = [i32]
    L.zada_destroy.argtypes = [vp]
    L.zada_last_error.restype = ctypes.c_char_p
    L.zada_last_error.argtypes = [vp]
    L.zada_version.restype = ctypes.c_char_p
    L.zada_deflate.argtypes = [vp, i32, vp, u64, vp, u64, u64p, u32p, vp, vp]
    L.zada_deflate_device.argtypes = [vp, i32, vp, u64, vp, u64, u64p, u32p]
    L.zada_deflate_batch.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.zada_compress_data.argtypes = [vp, i32, vp, u64, vp, u64, u64p, u32p, ctypes.POINTER(ctypes.c_uint16)]
    L.zada_lz77_tokens.argtypes = [vp, i32, vp, u64, vp, u64, u64p]
    L.zada_last_blocks.argtypes = [vp, vp, u64, u64p]
    L.zada_last_timing.argtypes = [vp, vp, vp, i32]
    L.zada_last_trace.argtypes = [vp, vp, u64, u64p]
    L.zada_silesia_mix.argtypes = [u64, ctypes.c_uint, u64, u64, vp]
    L.zada_silesia_mix_v2.argtypes = [u64, ctypes.c_uint, u64, u64, vp]
    L.zada_set_knob.argtypes = [vp, ctypes.c_char_p, i32]
    L.zada_range_open.argtypes = [vp, i32, vp, u64, u64, u64, u64, u64]
    L.zada_range_lz.argtypes = [vp, vp, vp]
    L.zada_range_edges.argtypes = [vp, vp, vp, u32p, vp, vp, u32p]
    L.zada_range_place.argtypes = [vp, u64, u64, vp, vp, ctypes.c_uint32, vp, vp, ctypes.c_uint32]
    L.zada_range_analyze.argtypes = [vp]
    L.zada_range_choose.argtypes = [vp, vp, vp, u64p, u64p]
    L.zada_range_emit.argtypes = [vp, vp, u64, u64p]
    L.zada_bzip2.argtypes = [vp, i32, vp, u64, vp, u64, u64p, u32p, vp, vp]
    L.zada_bzip2_device.argtypes = [vp, i32, vp, u64, vp, u64, u64p, u32p]
    L.zada_bz2_range_open.argtypes = [vp, i32, vp, u64, u64, u64, u64, u64, u64p, u64p]
    L.zada_bz2_range_encode.argtypes = [vp]
    L.zada_bz2_range_table.restype = ctypes.c_uint64
    L.zada_bz2_range_table.argtypes = [vp, vp, u64]
    L.zada_bz2_select.restype = None
    L.zada_bz2_select.argtypes = [u64, vp, u64, ctypes.c_uint32, vp, u64p, u32p]
    L.zada_bz2_range_assemble.argtypes = [vp, vp, u64, u64, i32, ctypes.c_uint32, vp, u64, u64p]
    L.zada_crc32_device.argtypes = [vp, vp, u64, u32p]
    L.zada_bzip2_batch.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.zada_lzma.argtypes = [vp, i32, vp, u64, vp, u64, u64p, u32p, vp, vp]
    L.zada_lzma_device.argtypes = [vp, i32, vp, u64, vp, u64, u64p, u32p]
    L.zada_lzma_export_state.argtypes = [vp, vp, u64, u64p, vp, u64, u64p, u64p]
    L.zada_lzma_import_state.argtypes = [vp, vp, u64]
    L.zada_lzma_batch.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.zada_lzma_match_sets.argtypes = [vp, vp, u64, vp, vp, vp, i32]
    L.zada_bz2_last_blocks.restype = ctypes.c_uint64
    L.zada_bz2_last_blocks.argtypes = [vp, vp, u64]
    L.zada_crc32_combine.restype = ctypes.c_uint32
    L.zada_crc32_combine.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u64]
    _lib = L
    return L


FEEDBACK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.c_void_p)
CARRY_BYTES = 352          # ZADA_CARRY_BYTES
RANGE_ALIGN = 65536        # range boundaries
RANGE_PRE = 32768          # bytes resident before a range that does not start the stream
RANGE_POST = 1 << 20       # ... and behind one that does not end it (or the rest of the stream, if shorter)
EDGE_HEAD, EDGE_TAIL = 65536, 2048


class _ParseState(ctypes.Structure):
    _fields_ = [("pos", ctypes.c_uint64), ("kind", ctypes.c_uint32), ("pad", ctypes.c_uint32)]


class _RangeInfo(ctypes.Structure):
    _fields_ = [("atoms", ctypes.c_uint64), ("exit", _ParseState), ("warm", _ParseState), ("crc_raw", ctypes.c_uint32), ("entry_known", ctypes.c_uint32)]


def _addr(buf):
    """Address of a bytes / bytearray / numpy array / ctypes buffer without copying."""
    if isinstance(buf, bytes):
        return ctypes.cast(ctypes.c_char_p(buf), ctypes.c_void_p).value
    if hasattr(buf, "ctypes"):
        return buf.ctypes.data
    return ctypes.addressof((ctypes.c_char * len(buf)).from_buffer(buf))


class Encoder:
    """One context = one GPU + stream + workspace (single owner, like one Ada task)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.ctx = self.lib.zada_create(device)
        if not self.ctx:
            raise ZadaError("zada_create(%d) failed: no usable gfx950 device (no CPU fallback)" % device)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.zada_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_knob(self, name, value):
        """Tuning / test knobs ("budget", "max_demand_rounds", "inner_budget", "shard_kib", "span_mib", "batch_mib"); none changes a byte of output.
        "lzma_dict" is the reference's dictionary_size parameter for LZMA_3 (0 = the entry's size, what Zip.Compress.LZMA_E passes)."""
        if self.lib.zada_set_knob(self.ctx, name.encode(), int(value)) != 0:
            raise ZadaError("unknown knob %r" % name)

    def _err(self, rc, what):
        if rc == E_REFERENCE:
            raise ReferenceDefect("%s: rc=%d (%s)" % (what, rc, self.lib.zada_last_error(self.ctx).decode()))
        raise ZadaError("%s failed: rc=%d (%s)" % (what, rc, self.lib.zada_last_error(self.ctx).decode()))

    def deflate(self, data, method=Method.Deflate_3, crc=0xFFFFFFFF, feedback=None):
        """Zip.Compress.Deflate.  Returns (raw deflate bytes, running CRC register).
        Raises CompressionInefficient when compression_ok would be False."""
        n = len(data)
        out = ctypes.create_string_buffer(n + 64)
        ol = ctypes.c_uint64(0)
        c = ctypes.c_uint32(crc)
        cb = FEEDBACK_FN(lambda pct, _u: 1 if feedback(pct) else 0) if feedback else None
        rc = self.lib.zada_deflate(self.ctx, method, _addr(data) if n else None, n, ctypes.addressof(out), n + 64,
                                   ctypes.byref(ol), ctypes.byref(c), ctypes.cast(cb, ctypes.c_void_p) if cb else None, None)
        if rc == 1:
            raise CompressionInefficient()
        if rc == 2:
            raise UserAbort()
        if rc != 0:
            self._err(rc, "zada_deflate")
        return out.raw[:ol.value], c.value

    def bzip2(self, data, method=14, crc=0xFFFFFFFF, feedback=None, cap=None):
        """Zip.Compress.BZip2_E (method 12 / 13 / 14 = BZip2_1 / _2 / _3).  Returns (rc, BZip2 stream, running CRC register);
        rc 1 = not smaller than the input (the stream is still returned when it fits `cap`, default len(data) * 5 // 4 + 4096)."""
        n = len(data)
        cap = int(cap if cap is not None else n + n // 4 + 4096)
        out = ctypes.create_string_buffer(cap)
        ol = ctypes.c_uint64(0)
        c = ctypes.c_uint32(crc)
        cb = FEEDBACK_FN(lambda pct, _u: 1 if feedback(pct) else 0) if feedback else None
        rc = self.lib.zada_bzip2(self.ctx, method, _addr(data) if n else None, n, ctypes.addressof(out), cap, ctypes.byref(ol), ctypes.byref(c),
                                 ctypes.cast(cb, ctypes.c_void_p) if cb else None, None)
        if rc == 2:
            raise UserAbort()
        if rc < 0:
            self._err(rc, "zada_bzip2")
        return rc, (out.raw[:ol.value] if ol.value <= cap else None), c.value

    def bzip2_batch(self, datas, method=14, crc=0xFFFFFFFF):
        """Independent BZip2 streams (one per Zip entry) in one call: zada_bzip2_batch takes the entries that are one block each
        (up to 0.8 block capacities) through ONE launch sequence.  Returns a list of (rc, stream or None, running CRC register);
        rc 1 = not smaller than the input (the stream is still there when it fits len + len // 4 + 1024 bytes)."""
        import numpy as np
        cnt = len(datas)
        if cnt == 0:
            return []
        lens = np.fromiter((len(d) for d in datas), dtype=np.uint64, count=cnt)
        caps = lens + lens // 4 + 1024
        offs = np.concatenate(([0], np.cumsum(caps)[:-1])).astype(np.uint64)
        arena = np.empty(int(caps.sum()), dtype=np.uint8)
        outp = (arena.ctypes.data + offs).astype(np.uint64)
        keep = [d if len(d) else b"\0" for d in datas]
        ins = np.fromiter((_addr(d) for d in keep), dtype=np.uint64, count=cnt)
        ols = np.zeros(cnt, dtype=np.uint64)
        crcs = np.full(cnt, crc, dtype=np.uint32)
        rcs = np.zeros(cnt, dtype=np.int32)
        worst = self.lib.zada_bzip2_batch(self.ctx, method, cnt, ins.ctypes.data, lens.ctypes.data, outp.ctypes.data, caps.ctypes.data,
                                          ols.ctypes.data, crcs.ctypes.data, rcs.ctypes.data)
        if worst < 0:
            self._err(worst, "zada_bzip2_batch")
        mv = memoryview(arena)
        return [(int(rcs[i]), bytes(mv[int(offs[i]):int(offs[i]) + int(ols[i])]) if rcs[i] >= 0 and ols[i] <= caps[i] else None, int(crcs[i])) for i in range(cnt)]

    def lzma(self, data, method=18, crc=0xFFFFFFFF, cap=None, feedback=None):
        """Zip.Compress.LZMA_E (method 15 .. 18 = LZMA_0 .. LZMA_3).  Returns (rc, Zip payload, running CRC register); rc 1 = not
        smaller than the input (the payload is still returned when it fits `cap`, default len(data) * 9 // 8 + 4096).
        feedback(percent) is called between the launches of the stream; a true return raises UserAbort."""
        n = len(data)
        cap = int(cap if cap is not None else n + n // 8 + 4096)
        out = ctypes.create_string_buffer(cap)
        ol = ctypes.c_uint64(0)
        c = ctypes.c_uint32(crc)
        cb = FEEDBACK_FN(lambda pct, _u: 1 if feedback(pct) else 0) if feedback else None
        rc = self.lib.zada_lzma(self.ctx, method, _addr(data) if n else None, n, ctypes.addressof(out), cap, ctypes.byref(ol), ctypes.byref(c),
                                ctypes.cast(cb, ctypes.c_void_p) if cb else None, None)
        if rc == 2:
            raise UserAbort()
        if rc < 0:
            self._err(rc, "zada_lzma")
        return rc, (out.raw[:ol.value] if ol.value <= cap else None), c.value

    def lzma_export_state(self, out_cap):
        """After lzma() raised UserAbort (its feedback stopped the stream between two launches): (state bytes, the stream bytes written so far,
        input positions coded).  zada_lzma_export_state."""
        sl = ctypes.c_uint64(0)
        self.lib.zada_lzma_export_state(self.ctx, None, 0, ctypes.byref(sl), None, 0, None, None)
        state = ctypes.create_string_buffer(sl.value)
        out = ctypes.create_string_buffer(int(out_cap))
        ob, pos = ctypes.c_uint64(0), ctypes.c_uint64(0)
        rc = self.lib.zada_lzma_export_state(self.ctx, ctypes.addressof(state), sl.value, ctypes.byref(sl), ctypes.addressof(out), int(out_cap), ctypes.byref(ob), ctypes.byref(pos))
        if rc != 0:
            self._err(rc, "zada_lzma_export_state")
        return state.raw[:sl.value], out.raw[:ob.value], pos.value

    def lzma_import_state(self, state):
        """The NEXT lzma() call of this encoder -- same input, same method -- goes on from `state` (lzma_export_state of any context or process); its
        payload is valid from the exported stream bytes' length on."""
        rc = self.lib.zada_lzma_import_state(self.ctx, _addr(state), len(state))
        if rc != 0:
            self._err(rc, "zada_lzma_import_state")

    def lzma_batch(self, datas, method=18, crc=0xFFFFFFFF):
        """Independent LZMA payloads (one per Zip entry) in one call: every entry is a workgroup of ONE launch of the coder.
        Returns a list of (rc, payload or None, running CRC register)."""
        import numpy as np
        cnt = len(datas)
        if cnt == 0:
            return []
        lens = np.fromiter((len(d) for d in datas), dtype=np.uint64, count=cnt)
        caps = lens + lens // 8 + 128
        offs = np.concatenate(([0], np.cumsum(caps)[:-1])).astype(np.uint64)
        arena = np.empty(int(caps.sum()), dtype=np.uint8)
        outp = (arena.ctypes.data + offs).astype(np.uint64)
        keep = [d if len(d) else b"\0" for d in datas]
        ins = np.fromiter((_addr(d) for d in keep), dtype=np.uint64, count=cnt)
        ols = np.zeros(cnt, dtype=np.uint64)
        crcs = np.full(cnt, crc, dtype=np.uint32)
        rcs = np.zeros(cnt, dtype=np.int32)
        worst = self.lib.zada_lzma_batch(self.ctx, method, cnt, ins.ctypes.data, lens.ctypes.data, outp.ctypes.data, caps.ctypes.data,
                                         ols.ctypes.data, crcs.ctypes.data, rcs.ctypes.data)
        if worst < 0 and not (worst == E_REFERENCE and all(r >= 0 or r == E_REFERENCE for r in rcs)):     # (refused entries: rc -6, no payload)
            self._err(worst, "zada_lzma_batch")
        mv = memoryview(arena)
        return [(int(rcs[i]), bytes(mv[int(offs[i]):int(offs[i]) + int(ols[i])]) if rcs[i] >= 0 and ols[i] <= caps[i] else None, int(crcs[i])) for i in range(cnt)]

    def lzma_match_sets(self, data, stride=50):
        """Test hook: the match sets of LZMA_3's BT4 matcher at every position of `data`, as the producer kernels leave them for the coder
        (lz77.adb:1234-1361).  Returns (cnt [n] u8, len [n, stride] u16, dist [n, stride] u32)."""
        import numpy as np
        n = len(data)
        cnt = np.zeros(n, np.uint8); ln = np.zeros((n, stride), np.uint16); ds = np.zeros((n, stride), np.uint32)
        rc = self.lib.zada_lzma_match_sets(self.ctx, _addr(data) if n else None, n, cnt.ctypes.data, ln.ctypes.data, ds.ctypes.data, stride)
        if rc < 0:
            self._err(rc, "zada_lzma_match_sets")
        return cnt, ln, ds

    def lzma_device(self, d_in, n, d_out, cap, method=18, crc=0xFFFFFFFF):
        """LZMA payload of n bytes at device address d_in into d_out (cap bytes).  Returns (rc, length, running CRC register)."""
        ol = ctypes.c_uint64(0)
        c = ctypes.c_uint32(crc)
        rc = self.lib.zada_lzma_device(self.ctx, method, d_in, n, d_out, cap, ctypes.byref(ol), ctypes.byref(c))
        if rc < 0:
            self._err(rc, "zada_lzma_device")
        return rc, ol.value, c.value

    def bzip2_device(self, d_in, n, d_out, cap, method=14, crc=0xFFFFFFFF):
        """BZip2 stream of n bytes at device address d_in into d_out (cap bytes).  Returns (rc, length, running CRC register)."""
        ol = ctypes.c_uint64(0)
        c = ctypes.c_uint32(crc)
        rc = self.lib.zada_bzip2_device(self.ctx, method, d_in, n, d_out, cap, ctypes.byref(ol), ctypes.byref(c))
        if rc < 0:
            self._err(rc, "zada_bzip2_device")
        return rc, ol.value, c.value

    def crc32_device(self, d_ptr, n):
        """Raw CRC-32 register (started from 0) of n bytes at a 16-byte aligned device address (see crc32_combine)."""
        raw = ctypes.c_uint32(0)
        rc = self.lib.zada_crc32_device(self.ctx, d_ptr, n, ctypes.byref(raw))
        if rc != 0:
            self._err(rc, "zada_crc32_device")
        return raw.value

    # ---- one BZip2 stream over several contexts (zada_bz2_range_*, include/zada.h) ----
    def bz2_range_open(self, d_buf, buf_len, buf_off, stream_total, start, own_end, method=14):
        """Block limits of the blocks that start in [start, own_end).  Returns (next_start, number of blocks)."""
        nxt, nb = ctypes.c_uint64(0), ctypes.c_uint64(0)
        rc = self.lib.zada_bz2_range_open(self.ctx, method, d_buf, buf_len, buf_off, stream_total, start, own_end, ctypes.byref(nxt), ctypes.byref(nb))
        if rc != 0:
            self._err(rc, "zada_bz2_range_open")
        return nxt.value, nb.value

    def bz2_range_encode(self):
        rc = self.lib.zada_bz2_range_encode(self.ctx)
        if rc != 0:
            self._err(rc, "zada_bz2_range_encode")

    def bz2_range_table(self):
        """numpy uint64 [blocks, 4 tactics, 3]: bits (all ones: the block has no such tactic), pieces, folded CRC."""
        import numpy as np
        nb = self.lib.zada_bz2_range_table(self.ctx, None, 0)
        tab = np.zeros((max(int(nb), 1), 4, 3), np.uint64)
        self.lib.zada_bz2_range_table(self.ctx, tab.ctypes.data, nb)
        return tab[:nb]

    def bz2_select(self, tab, bitpos_in=32, crc_in=0):
        """Tactic per block along the stream.  Returns (choices uint8 array, bit position behind, combined CRC behind)."""
        import numpy as np
        tab = np.ascontiguousarray(tab, np.uint64)
        nb = tab.shape[0]
        choice = np.zeros(max(nb, 1), np.uint8)
        bp, crc = ctypes.c_uint64(0), ctypes.c_uint32(0)
        self.lib.zada_bz2_select(nb, tab.ctypes.data, bitpos_in, crc_in, choice.ctypes.data, ctypes.byref(bp), ctypes.byref(crc))
        return choice[:nb], bp.value, crc.value

    def bz2_range_assemble(self, choice, bit_begin, d_out, cap, header=False, footer_crc=None):
        """The range's bytes of the stream (from byte bit_begin // 8 on) into d_out.  Returns their number."""
        import numpy as np
        choice = np.ascontiguousarray(choice, np.uint8)
        nbytes = ctypes.c_uint64(0)
        flags = (1 if header else 0) | (2 if footer_crc is not None else 0)
        rc = self.lib.zada_bz2_range_assemble(self.ctx, choice.ctypes.data if len(choice) else None, len(choice), bit_begin, flags, footer_crc or 0, d_out, cap, ctypes.byref(nbytes))
        if rc != 0:
            self._err(rc, "zada_bz2_range_assemble")
        return nbytes.value

    def bz2_last_blocks(self):
        """[(raw start, raw length, tactic, sub-blocks)] of the last bzip2 call (bzip2-encoding.adb:1144, :1312-1318)."""
        import numpy as np
        k = self.lib.zada_bz2_last_blocks(self.ctx, None, 0)
        buf = np.zeros(max(int(k), 1), np.uint64)
        self.lib.zada_bz2_last_blocks(self.ctx, buf.ctypes.data, k)
        return [tuple(int(x) for x in buf[i:i + 4]) for i in range(0, int(k), 4)]

    def deflate_into(self, data, out, method=Method.Deflate_3, crc=0xFFFFFFFF):
        """Zip.Compress.Deflate into a caller-owned buffer (bytearray / numpy uint8 / ctypes, len(out) >= len(data) + 64):
        no allocation on the way.  Returns (rc, out_len, running CRC register); rc 1 = inefficient."""
        n = len(data)
        ol = ctypes.c_uint64(0)
        c = ctypes.c_uint32(crc)
        rc = self.lib.zada_deflate(self.ctx, method, _addr(data) if n else None, n, _addr(out), len(out), ctypes.byref(ol), ctypes.byref(c), None, None)
        if rc < 0 or rc == 2:
            self._err(rc, "zada_deflate")
        return rc, ol.value, c.value

    def deflate_batch(self, datas, method=Method.Deflate_3, crc=0xFFFFFFFF):
        """Independent streams (one per Zip entry, Zip.Create.Add_Stream is per entry) in one call: zada_deflate_batch takes
        the entries of up to 4 MiB through ONE launch sequence.  Returns a list of (rc, raw deflate bytes or None, running
        CRC register); rc 1 = inefficient (the caller Stores the entry)."""
        import numpy as np
        cnt = len(datas)
        if cnt == 0:
            return []
        lens = np.fromiter((len(d) for d in datas), dtype=np.uint64, count=cnt)
        caps = lens + 64
        offs = np.concatenate(([0], np.cumsum(caps)[:-1])).astype(np.uint64)
        arena = np.empty(int(caps.sum()), dtype=np.uint8)                       # one output arena instead of one buffer per entry
        outp = (arena.ctypes.data + offs).astype(np.uint64)
        keep = [d if len(d) else b"\0" for d in datas]
        ins = np.fromiter((_addr(d) for d in keep), dtype=np.uint64, count=cnt)
        ols = np.zeros(cnt, dtype=np.uint64)
        crcs = np.full(cnt, crc, dtype=np.uint32)
        rcs = np.zeros(cnt, dtype=np.int32)
        worst = self.lib.zada_deflate_batch(self.ctx, method, cnt, ins.ctypes.data, lens.ctypes.data, outp.ctypes.data, caps.ctypes.data,
                                            ols.ctypes.data, crcs.ctypes.data, rcs.ctypes.data)
        if worst < 0:
            self._err(worst, "zada_deflate_batch")
        mv = memoryview(arena)
        return [(int(rcs[i]), bytes(mv[int(offs[i]):int(offs[i]) + int(ols[i])]) if rcs[i] == 0 else None, int(crcs[i])) for i in range(cnt)]

    def deflate_device(self, d_in_ptr, n, d_out_ptr, cap, method=Method.Deflate_3, crc=0xFFFFFFFF):
        """Device-resident variant (pointers are HBM addresses, e.g. torch tensor .data_ptr()).
        Returns (rc, out_len, crc); rc 1 = inefficient."""
        ol = ctypes.c_uint64(0)
        c = ctypes.c_uint32(crc)
        rc = self.lib.zada_deflate_device(self.ctx, method, d_in_ptr, n, d_out_ptr, cap, ctypes.byref(ol), ctypes.byref(c))
        if rc < 0:
            self._err(rc, "zada_deflate_device")
        return rc, ol.value, c.value

    def compress_data(self, data, method=Method.Deflate_3):
        """Zip.Compress.Compress_Data (single method, no password): returns
        (payload bytes, final CRC-32, zip_type) with the Store fallback applied."""
        n = len(data)
        if method == Method.Store:
            import zlib  # CRC of stored data only; not on the Deflate path
            return bytes(data), zlib.crc32(data) & 0xFFFFFFFF, 0
        out = ctypes.create_string_buffer(n + 64)
        ol = ctypes.c_uint64(0)
        c = ctypes.c_uint32(0)
        zt = ctypes.c_uint16(0)
        rc = self.lib.zada_compress_data(self.ctx, method, _addr(data) if n else None, n, ctypes.addressof(out), n + 64,
                                         ctypes.byref(ol), ctypes.byref(c), ctypes.byref(zt))
        if rc != 0:
            self._err(rc, "zada_compress_data")
        return out.raw[:ol.value], c.value, zt.value

    def lz77_tokens(self, data, method=Method.Deflate_3):
        import numpy as np
        n = len(data)
        tok = np.zeros(n + 8, dtype=np.uint32)
        nt = ctypes.c_uint64(0)
        rc = self.lib.zada_lz77_tokens(self.ctx, method, _addr(data) if n else None, n, tok.ctypes.data, n + 8, ctypes.byref(nt))
        if rc != 0:
            self._err(rc, "zada_lz77_tokens")
        return tok[:nt.value]

    # ---- one stream over several contexts (include/zada.h "One stream over several contexts"; driver: sharding.py) ----
    def range_open(self, d_in_ptr, stream_size, lo, n, pre, post, method=Method.Deflate_3):
        rc = self.lib.zada_range_open(self.ctx, method, d_in_ptr, stream_size, lo, n, pre, post)
        if rc != 0:
            self._err(rc, "zada_range_open")

    def range_lz(self, entry=None):
        """entry: (pos, kind) the range before ended in, or None.  Returns dict(atoms, exit, warm, crc_raw, entry_known)."""
        info = _RangeInfo()
        e = _ParseState(entry[0], entry[1], 0) if entry is not None else None
        rc = self.lib.zada_range_lz(self.ctx, ctypes.addressof(e) if e is not None else None, ctypes.addressof(info))
        if rc != 0:
            self._err(rc, "zada_range_lz")
        return dict(atoms=info.atoms, exit=(info.exit.pos, info.exit.kind), warm=(info.warm.pos, info.warm.kind),
                    crc_raw=info.crc_raw, entry_known=bool(info.entry_known))

    def range_edges(self, head_atoms_ptr, head_pos_ptr, tail_atoms_ptr, tail_pos_ptr):
        nh, nt = ctypes.c_uint32(0), ctypes.c_uint32(0)
        rc = self.lib.zada_range_edges(self.ctx, head_atoms_ptr, head_pos_ptr, ctypes.byref(nh), tail_atoms_ptr, tail_pos_ptr, ctypes.byref(nt))
        if rc != 0:
            self._err(rc, "zada_range_edges")
        return nh.value, nt.value

    def range_place(self, atoms_before, atoms_total, lb_atoms_ptr=None, lb_pos_ptr=None, n_lb=0, la_atoms_ptr=None, la_pos_ptr=None, n_la=0):
        rc = self.lib.zada_range_place(self.ctx, atoms_before, atoms_total, lb_atoms_ptr, lb_pos_ptr, n_lb, la_atoms_ptr, la_pos_ptr, n_la)
        if rc != 0:
            self._err(rc, "zada_range_place")

    def range_analyze(self):
        rc = self.lib.zada_range_analyze(self.ctx)
        if rc != 0:
            self._err(rc, "zada_range_analyze")

    def range_choose(self, carry_in=None):
        """carry_in: the 352-byte state of the range before (None: the stream starts here).
        Returns (carry_out bytes, bit_begin, bit_end)."""
        cin = ctypes.create_string_buffer(bytes(carry_in), CARRY_BYTES) if carry_in is not None else None
        cout = ctypes.create_string_buffer(CARRY_BYTES)
        b0, b1 = ctypes.c_uint64(0), ctypes.c_uint64(0)
        rc = self.lib.zada_range_choose(self.ctx, ctypes.addressof(cin) if cin is not None else None, ctypes.addressof(cout),
                                        ctypes.byref(b0), ctypes.byref(b1))
        if rc != 0:
            self._err(rc, "zada_range_choose")
        return cout.raw, b0.value, b1.value

    def range_emit(self, d_out_ptr, cap):
        nb = ctypes.c_uint64(0)
        rc = self.lib.zada_range_emit(self.ctx, d_out_ptr, cap, ctypes.byref(nb))
        if rc != 0:
            self._err(rc, "zada_range_emit")
        return nb.value

    def crc32_combine(self, reg, raw, length):
        return self.lib.zada_crc32_combine(reg, raw, length)

    def last_blocks(self):
        import numpy as np
        nb = ctypes.c_uint64(0)
        self.lib.zada_last_blocks(self.ctx, None, 0, ctypes.byref(nb))
        rec = np.zeros((max(nb.value, 1), 4), dtype=np.uint64)
        self.lib.zada_last_blocks(self.ctx, rec.ctypes.data, nb.value, ctypes.byref(nb))
        return rec[:nb.value]

    def last_trace(self):
        """Similarity tests of the block splitter in the last call: array of (atom, L1 distance, cut level or 0)."""
        import numpy as np
        k = ctypes.c_uint64(0)
        self.lib.zada_last_trace(self.ctx, None, 0, ctypes.byref(k))
        rec = np.zeros((max(k.value, 1), 3), dtype=np.uint64)
        self.lib.zada_last_trace(self.ctx, rec.ctypes.data, k.value, ctypes.byref(k))
        return rec[:k.value]

    def last_timing(self):
        names = (ctypes.c_char_p * 64)()
        ms = (ctypes.c_float * 64)()
        k = self.lib.zada_last_timing(self.ctx, ctypes.cast(names, ctypes.c_void_p), ctypes.cast(ms, ctypes.c_void_p), 64)
        return [(names[i].decode(), ms[i]) for i in range(k)]


def silesia_mix(nbytes, seed=0x5A1E51A, class_mask=0x1F, offset=0, version=1):
    """Deterministic synthetic corpus (csrc/silesia_mix.c), as a numpy uint8 array.  version 1 = "silesia_mix_v1" (what the committed
    golden digests were taken on; its 64 KiB segments are shifted copies of one stream of draws, which only encoders that look
    further back than 32 KiB can see), version 2 = "silesia_mix_v2" (segments seeded independently: the benchmark stream)."""
    import numpy as np
    L = load_library()
    b = np.zeros(nbytes, dtype=np.uint8)
    if nbytes:
        (L.zada_silesia_mix_v2 if version >= 2 else L.zada_silesia_mix)(seed, class_mask, offset, nbytes, b.ctypes.data)
    return b


class ZipCreate:
    """Zip.Create on a memory stream: Create_Archive / Add_Stream / Finish
    (zip_lib/zip-create.adb:36-58, 194-297, 645-756; headers zip-headers.adb:168-195, 244-276, 494-511), incl. the
    promotion to Zip_64 (Check_Size zip-create.adb:161-179; local header extension :237-251, 283-289 and
    zip-headers.adb:197-210, 336-355; central extension and Zip64 end records zip-create.adb:682-752,
    zip-headers.adb:534-579)."""

    DEFAULT_TIME = 16789 * 65536  # zip_streams.ads:223
    _MARGIN = 22 + 56 + 20 + 2 ** 16 + 10   # Check_Size, zip-create.adb:165-169

    def __init__(self, encoder, method=Method.Deflate_3, _offset_bias=0):
        self.enc, self.method = encoder, method
        self.buf = bytearray()
        self.entries = []
        self.zip64 = False
        self._bias = _offset_bias       # test hook: pretend that this many bytes precede the buffer

    def _check_size(self, value):
        if not self.zip64 and value >= 2 ** 32 - self._MARGIN:
            self.zip64 = True

    @staticmethod
    def _needs_zip64(csize, usize, offset):      # Needs_Local_Zip_64_Header_Extension, zip-headers.adb:197-210
        return csize >= 0xFFFFFFFF or usize >= 0xFFFFFFFF or offset >= 0xFFFFFFFF

    def add_stream(self, name, data, file_time=None, unicode_name=True):
        payload, crc, zt = self.enc.compress_data(data, self.method)
        return self.add_compressed(name, payload, crc, len(data), zt, file_time, unicode_name)

    def add_streams(self, names, datas, file_time=None, unicode_name=True):
        """Add_Stream for many entries at once: the entries are compressed as one batch (zada_deflate_batch: one launch
        sequence for all the small ones), with Compress_Data's Store fallback (zip-compress.adb:224-237) and CRC Init / Final
        (:144, 218) per entry.  The archive is the one Add_Stream after Add_Stream writes."""
        import zlib
        if self.method == Method.Store:
            res = [(1, None, 0)] * len(datas)
        elif 12 <= self.method <= 14:
            res = self.enc.bzip2_batch(datas, self.method)
        elif 15 <= self.method <= 18:
            res = self.enc.lzma_batch(datas, self.method)
        else:
            res = self.enc.deflate_batch(datas, self.method)
        for name, data, (rc, payload, crc) in zip(names, datas, res):
            if rc == 0:
                self.add_compressed(name, payload, crc ^ 0xFFFFFFFF, len(data), 12 if 12 <= self.method <= 14 else 14 if 15 <= self.method <= 18 else 8, file_time, unicode_name)
            else:
                self.add_compressed(name, bytes(data), zlib.crc32(data) if self.method == Method.Store else crc ^ 0xFFFFFFFF, len(data), 0, file_time, unicode_name)

    def add_compressed(self, name, payload, crc, usize, zt, file_time=None, unicode_name=True):
        """Entry whose payload was compressed elsewhere (another rank / GPU): the bytes written
        are those Add_Stream would have written for the same payload."""
        nm = name.replace("\\", "/").encode("utf-8")
        e = dict(name=nm, flag=(0x0800 if unicode_name else 0) | (0x0002 if zt == 14 else 0),   # LZMA_EOS_Flag_Bit, zip-create.adb:266-278
                 zip_type=zt, time=self.DEFAULT_TIME if file_time is None else file_time,
                 crc=crc, csize=len(payload), usize=usize, offset=len(self.buf) + self._bias)
        self._check_size(usize)
        # the local header's form is decided before compression, on the provisional sizes (:231-241)
        z64 = self._needs_zip64(usize, usize, e["offset"])
        if z64:
            hdr = struct.pack("<4sHHHIIIIHH", b"PK\x03\x04", 10, e["flag"], zt, e["time"], crc, 0xFFFFFFFF, 0xFFFFFFFF, len(nm), 20)
            ext = struct.pack("<HHQQ", 1, 16, usize, e["csize"])
        else:
            hdr = struct.pack("<4sHHHIIIIHH", b"PK\x03\x04", 10, e["flag"], zt, e["time"], crc, e["csize"], usize, len(nm), 0)
            ext = b""
        self.buf += hdr + nm + ext + payload
        self.entries.append(e)
        return e["csize"], zt

    def finish(self):
        cd_off = len(self.buf) + self._bias
        if not self.zip64 and len(self.entries) >= 0xFFFF:
            self.zip64 = True
        cd_size = 0
        for e in self.entries:
            z64 = self._needs_zip64(e["csize"], e["usize"], e["offset"])
            if z64:
                self.zip64 = True
            m = 0xFFFFFFFF
            self.buf += struct.pack("<4sHHHHIIIIHHHHHII", b"PK\x01\x02", 23, 10, e["flag"], e["zip_type"], e["time"], e["crc"],
                                    m if z64 else e["csize"], m if z64 else e["usize"], len(e["name"]), 28 if z64 else 0, 0, 0, 0, 0,
                                    m if z64 else e["offset"]) + e["name"]
            if z64:
                self.buf += struct.pack("<HHQQQ", 1, 24, e["usize"], e["csize"], e["offset"])
            cd_size += 46 + len(e["name"]) + (28 if z64 else 0)
        if self.entries:
            self._check_size(len(self.buf) + self._bias + 1)
        n = len(self.entries)
        if self.zip64:
            e64_off = len(self.buf) + self._bias
            self.buf += struct.pack("<4sQHHIIQQQQ", b"PK\x06\x06", 44, 0x2D, 0x2D, 0, 0, n, n, cd_size, cd_off)
            self.buf += struct.pack("<4sIQI", b"PK\x06\x07", 0, e64_off, 1)
            self.buf += struct.pack("<4sHHHHIIH", b"PK\x05\x06", 0, 0, 0xFFFF, 0xFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0)
        else:
            self.buf += struct.pack("<4sHHHHIIH", b"PK\x05\x06", 0, 0, n, n, cd_size, cd_off, 0)
        return bytes(self.buf)
