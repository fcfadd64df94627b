"""Generates the committed fixtures under tests/golden/ (run in the BUILD container, where
/root/reference exists; the fixtures -- data only -- travel to the GPU box, the reference does not).

  llhc_vectors.json   the three frequency vectors the reference's own LLHC test program feeds to
                      Length_Limited_Coding (test/test_llhc.adb:15, 46, 69-128; that program prints
                      the lengths and asserts nothing) + the oracle's code lengths ("self-pinned")
  sample.xls, sample.jpg, sample_pgm_100k.bin   data files of the reference's test corpus
                      (test/test_data/), the edge cases test/test_za.hac zips (sample.jpg is the
                      file behind lz77.adb:740-744)
  deflate_digests.json  SHA-256 + size of the oracle's stream for every (input, method) of the
                      parity matrix ("self-pinned": detects any later drift of the oracle and gives
                      the GPU tests a second, oracle-free comparison)
  lzma_digests.json   the same for the LZMA half (methods 15 .. 18), with the counters of the DL-code variants taken
  bzip2_digests.json  the same for the BZip2 half (methods 12 .. 14), with the block / tactic trace; every stream was
                      decompressed with libbz2 when the file was made
  zlib_tokens_*.npz   position-indexed LZ77 tokens made by libz 1.2.11 deflateTune for the
                      fixture files (the independent pin of the LZ77 stage)
"""
import ctypes
import hashlib
import json
import os
import re
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _common import METHODS, LEVEL, TUNE, oracle, zlibpin, oracle_deflate, edge_inputs  # noqa: E402

REF = "/root/reference"


def llhc_vectors():
    src = open(os.path.join(REF, "test", "test_llhc.adb")).read()
    v1 = [int(x) for x in re.search(r"freq := \(([\d, ]+)\);\s*\n\s*for m in 4 \.\. 5", src).group(1).split(",")]
    v2 = [int(x) for x in re.search(r"\n\s*freq := \(([\d, ]+)\);\s*--  OK after fixing LLHC", src).group(1).split(",")]
    body = src[src.index("procedure Test_3"):]
    pairs = re.findall(r"(\d+)\s*=>\s*(\d+)", body[:body.index("New_Line")])
    v3 = [0] * 288
    for k, v in pairs:
        v3[int(k)] = int(v)
    out = []
    for name, freq, mbs in (("test_1", v1, (4, 5)), ("test_2", v2, (7,)), ("test_3", v3, (15,))):
        for mb in mbs:
            f = np.array(freq, dtype=np.uint64)
            bl = np.zeros(len(freq), dtype=np.int32)
            assert oracle().zo_llhc(f.ctypes.data, len(freq), mb, bl.ctypes.data) == 0
            out.append(dict(name=name, max_bits=mb, freq=freq, lengths=bl.tolist(), source="test/test_llhc.adb", pinned="self (oracle)"))
    json.dump(out, open(os.path.join(HERE, "llhc_vectors.json"), "w"))
    return out


def data_files():
    td = os.path.join(REF, "test", "test_data")
    shutil.copyfile(os.path.join(td, "sample.xls"), os.path.join(HERE, "sample.xls"))
    shutil.copyfile(os.path.join(td, "sample.jpg"), os.path.join(HERE, "sample.jpg"))
    open(os.path.join(HERE, "sample_pgm_100k.bin"), "wb").write(open(os.path.join(td, "sample.pgm"), "rb").read()[:100000])
    for f in ("sample.xls", "sample.jpg", "sample_pgm_100k.bin"):
        os.chmod(os.path.join(HERE, f), 0o644)


def zlib_tokens():
    for f in ("sample.xls", "sample.jpg", "sample_pgm_100k.bin"):
        d = open(os.path.join(HERE, f), "rb").read()
        for lvl in (6, 8, 10):
            z = np.zeros(len(d), dtype=np.uint32)
            rc = zlibpin().zp_zlib_position_tokens(d, len(d), *TUNE[lvl], z.ctypes.data)
            assert rc == 0, rc
            np.savez_compressed(os.path.join(HERE, "zlib_tokens_%s_L%d.npz" % (f.replace(".", "_"), lvl)), tokens=z)


def digests():
    cases = dict(edge_inputs())
    for f in ("sample.xls", "sample.jpg", "sample_pgm_100k.bin"):
        cases[f] = open(os.path.join(HERE, f), "rb").read()
    out = {}
    for name, d in sorted(cases.items()):
        for m in METHODS:
            rc, z, crc = oracle_deflate(d, m)
            out["%s|%d" % (name, m)] = dict(rc=rc, size=len(z), sha256=hashlib.sha256(z).hexdigest() if rc == 0 else None,
                                           in_sha256=hashlib.sha256(d).hexdigest(), crc=crc ^ 0xFFFFFFFF)
    json.dump(out, open(os.path.join(HERE, "deflate_digests.json"), "w"), indent=0, sort_keys=True)


def bzip2_digests():
    """bzip2_digests.json: SHA-256 + size of the oracle's BZip2 stream per (input, method 12..14), the block trace (raw start,
    raw length, tactic, sub-blocks) and whether libbz2 gives the input back ("self-pinned" + round trip)."""
    import bz2
    from _bzip2 import oracle_encode, bz_inputs
    cases = bz_inputs()
    for f in ("sample.xls", "sample.jpg", "sample_pgm_100k.bin"):
        cases[f] = open(os.path.join(HERE, f), "rb").read()
    out = {}
    for name, d in sorted(cases.items()):
        for m in (12, 13, 14):
            if m != 14 and len(d) > 400000:
                continue
            z, ev = oracle_encode(d, m - 12)
            if len(d):
                assert bz2.decompress(z) == d, (name, m)
            out["%s|%d" % (name, m)] = dict(size=len(z), sha256=hashlib.sha256(z).hexdigest(), in_sha256=hashlib.sha256(d).hexdigest(), blocks=ev)
    json.dump(out, open(os.path.join(HERE, "bzip2_digests.json"), "w"), indent=0, sort_keys=True)


def lzma_digests():
    """lzma_digests.json: SHA-256 + size of the oracle's Zip LZMA payload per (input, method 15..18) and the counters of the
    choices taken ("self-pinned"); liblzma gives the input back for every one of them."""
    from _lzmah import oracle_lzma, oracle_lzma_encode, lz_inputs, lzma_decode
    cases = lz_inputs()
    for f in ("sample.xls", "sample.jpg", "sample_pgm_100k.bin"):
        cases[f] = open(os.path.join(HERE, f), "rb").read()
    out = {}
    for name, d in sorted(cases.items()):
        for m in (15, 16, 17, 18):
            rc, z, crc = oracle_lzma(d, m)
            assert lzma_decode(z, 4) == d, (name, m)
            st = oracle_lzma_encode(d, m - 15)[1]
            out["%s|%d" % (name, m)] = dict(rc=rc, size=len(z), sha256=hashlib.sha256(z).hexdigest(), in_sha256=hashlib.sha256(d).hexdigest(), choices=st)
    json.dump(out, open(os.path.join(HERE, "lzma_digests.json"), "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "bzip2":
        bzip2_digests()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lzma":
        lzma_digests()
        sys.exit(0)
    llhc_vectors()
    data_files()
    zlib_tokens()
    digests()
    bzip2_digests()
    lzma_digests()
    print("golden fixtures written to", HERE)
