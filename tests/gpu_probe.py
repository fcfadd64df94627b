"""Bring-up probe (run on the GPU box): stage-by-stage comparison of the HIP path with the oracle."""
import ctypes, importlib, os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
za = importlib.import_module("zip-ada_amd")
O = ctypes.CDLL(os.path.join(ROOT, "oracle", "libzada_oracle.so"))
O.zo_lz77_tokens.restype = ctypes.c_uint64
O.zo_lz77_tokens.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64]
O.zo_deflate.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
LEVEL = {6: 4, 7: 0, 8: 6, 9: 8, 10: 10}
TRACE = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64)

def otok(d, method):
    t = np.zeros(len(d) + 8, dtype=np.uint32)
    k = O.zo_lz77_tokens(d, len(d), LEVEL[method], t.ctypes.data, len(t))
    return t[:k]

def odeflate(d, method, blocks=None):
    n = len(d); out = ctypes.create_string_buffer(n + 64); ol = ctypes.c_uint64(0); crc = ctypes.c_uint32(0xFFFFFFFF)
    cb = None
    if blocks is not None:
        def tr(u, kind, a, b, c, dd):
            if kind == 2: blocks.append((a, b, c, dd))
        cb = TRACE(tr)
    rc = O.zo_deflate(d, n, method, out, n + 64, ctypes.byref(ol), ctypes.byref(crc), None, None, ctypes.cast(cb, ctypes.c_void_p) if cb else None, None)
    return rc, out.raw[:ol.value], crc.value

def main():
    enc = za.Encoder(0)
    print("version", za.load_library().zada_version().decode())
    cases = {}
    cases["mix64k"] = za.silesia_mix(65536).tobytes()
    cases["mix1m"] = za.silesia_mix(1 << 20).tobytes()
    cases["mix3m"] = za.silesia_mix(3 << 20).tobytes()
    cases["text300k"] = za.silesia_mix(300000, class_mask=1).tobytes()
    cases["zeros100k"] = bytes(100000)
    cases["rand66666"] = bytes(np.random.RandomState(2).randint(0, 256, 66666).astype(np.uint8))
    cases["az77777"] = bytes(np.random.RandomState(1).randint(65, 91, 77777).astype(np.uint8))
    for sz in (0, 1, 2, 3, 4, 100, 258, 259, 4096, 4097, 32768, 65537):
        cases["t%d" % sz] = za.silesia_mix(sz, class_mask=1).tobytes()
    nbad = 0
    for name, d in cases.items():
        for method in (10, 8, 9, 7, 6):
            a = otok(d, method)
            try:
                b = enc.lz77_tokens(d, method)
            except Exception as e:
                print("TOKENS EXC", name, method, e); nbad += 1; continue
            tok_ok = len(a) == len(b) and bool((a == b).all())
            if not tok_ok:
                i = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
                print("TOKEN MISMATCH", name, method, len(a), len(b), "first diff", i, [hex(x) for x in a[i:i+3]], [hex(x) for x in b[i:i+3]])
                nbad += 1
            ob = []
            rc, ref, crc = odeflate(d, method, ob)
            try:
                out, c2 = enc.deflate(d, method)
                rc2 = 0
            except za.CompressionInefficient:
                rc2, out, c2 = 1, b"", None
            except Exception as e:
                print("DEFLATE EXC", name, method, e); nbad += 1; continue
            ok = (rc == rc2) and (rc != 0 or out == ref) and (c2 is None or c2 == crc)
            if not ok:
                nbad += 1
                gb = enc.last_blocks()
                j = next((i for i in range(min(len(out), len(ref))) if out[i] != ref[i]), min(len(out), len(ref)))
                print("DEFLATE MISMATCH", name, method, "rc", rc, rc2, "len", len(ref), len(out), "first diff byte", j, "crc", crc, c2)
                print("   oracle blocks", ob[:6], "gpu blocks", gb[:6].tolist())
                try:
                    print("   gpu roundtrip", zlib.decompress(out, -15) == d)
                except Exception as e:
                    print("   gpu stream invalid:", e)
            else:
                print("ok", name, method, len(d), "->", len(ref), "rc", rc, "tokens_ok", tok_ok)
    print("TOTAL BAD", nbad)
    d = za.silesia_mix(64 << 20).tobytes()
    for _ in range(2):
        t0 = time.time(); out, _c = enc.deflate(d, 10); dt = time.time() - t0
        print("64MiB deflate_3: %.3fs  %.1f MB/s  ratio %.4f" % (dt, len(d) / dt / 1e6, len(out) / len(d)))
        print("   ", [(k, round(v, 2)) for k, v in enc.last_timing()])
    print("roundtrip 64MiB", zlib.decompress(out, -15) == d)

if __name__ == "__main__":
    main()
