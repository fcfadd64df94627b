"""Second half of pin_with_gnat.sh: runs the reference's zipada on every input of the parity matrix and compares the raw
Deflate, BZip2 and LZMA streams with the committed digests of the oracle's streams (tests/golden/deflate_digests.json, bzip2_digests.json,
lzma_digests.json)."""
import hashlib
import json
import os
import struct
import subprocess
import sys

zipada, work, root = sys.argv[1:4]
zipada = zipada.split()                                       # ("python3 oracle/zipada_stub.py" for the rehearsal without GNAT)
sys.path.insert(0, os.path.join(root, "tests"))
from _common import GOLDEN, edge_inputs  # noqa: E402

LIMIT = int(os.environ.get("PIN_LIMIT", "0"))                 # rehearsals: at most this many inputs per half (0 = all)
# What the Ada binary's bytes depend on OUTSIDE the zip-ada tree (exactness notes): worth knowing when a difference shows up.
print("note: BZip2 streams depend on GNAT's Ada.Containers.Generic_Constrained_Array_Sort (a-cgcaso.adb: heap sort, its order of equal keys)")
print("note: BZip2 segmentation depends on the C library's log() behind Ada.Numerics.Elementary_Functions.Log (data_segmentation.adb)")
print("note: LZMA estimates are `digits 15` floating point: the oracle assumes IEEE doubles without fused multiply-add")


def limited(items):
    items = sorted(items)
    return items[:LIMIT] if LIMIT else items

OPT = {6: "-edf", 7: "-ed0", 8: "-ed1", 9: "-ed2", 10: "-ed3"}
cases = dict(edge_inputs())
for f in ("sample.xls", "sample.jpg", "sample_pgm_100k.bin"):
    cases[f] = open(os.path.join(GOLDEN, f), "rb").read()
dig = json.load(open(os.path.join(GOLDEN, "deflate_digests.json")))
bad = checked = 0
for name, data in limited(cases.items()):
    src = os.path.join(work, "in.bin")
    open(src, "wb").write(data)
    for m, opt in OPT.items():
        want = dig["%s|%d" % (name, m)]
        arc = os.path.join(work, "out.zip")
        if os.path.exists(arc):
            os.remove(arc)
        subprocess.run(zipada + [opt, arc, src], check=True, stdout=subprocess.DEVNULL, cwd=work)
        z = open(arc, "rb").read()
        sig, ver, flag, method, tm, crc, csize, usize, nl, xl = struct.unpack("<4sHHHIIIIHH", z[:30])
        assert sig == b"PK\x03\x04"
        payload = z[30 + nl + xl:30 + nl + xl + csize]
        checked += 1
        if want["rc"] == 1:                                   # Compression_inefficient: the entry must have been stored
            ok = method == 0 and payload == data
        else:
            ok = method == 8 and csize == want["size"] and hashlib.sha256(payload).hexdigest() == want["sha256"]
        if not ok:
            bad += 1
            print("DIFFERENT: %s method %d: zipada wrote %d bytes (zip method %d), the oracle %s" % (name, m, csize, method, want["size"]))
# ---- the BZip2 half (tests/golden/bzip2_digests.json): zipada -eb1 / -eb2 / -eb3 ----
from _bzip2 import bz_inputs  # noqa: E402
bdig = json.load(open(os.path.join(GOLDEN, "bzip2_digests.json")))
bcases = bz_inputs()
for f in ("sample.xls", "sample.jpg", "sample_pgm_100k.bin"):
    bcases[f] = open(os.path.join(GOLDEN, f), "rb").read()
for key, want in limited(bdig.items()):
    name, m = key.split("|")
    data = bcases[name]
    src = os.path.join(work, "in.bin")
    open(src, "wb").write(data)
    arc = os.path.join(work, "out.zip")
    if os.path.exists(arc):
        os.remove(arc)
    subprocess.run(zipada + [{"12": "-eb1", "13": "-eb2", "14": "-eb3"}[m], arc, src], check=True, stdout=subprocess.DEVNULL, cwd=work)
    z = open(arc, "rb").read()
    sig, ver, flag, method, tm, crc, csize, usize, nl, xl = struct.unpack("<4sHHHIIIIHH", z[:30])
    payload = z[30 + nl + xl:30 + nl + xl + csize]
    checked += 1
    if want["size"] >= len(data):                             # compression_ok = False: stored
        ok = method == 0 and payload == data
    else:
        ok = method == 12 and csize == want["size"] and hashlib.sha256(payload).hexdigest() == want["sha256"]
    if not ok:
        bad += 1
        print("DIFFERENT: %s BZip2 method %s: zipada wrote %d bytes (zip method %d), the oracle %s" % (name, m, csize, method, want["size"]))
# ---- the LZMA half (tests/golden/lzma_digests.json): zipada -el0 / -el1 / -el2 / -el3 ----
from _lzmah import lz_inputs  # noqa: E402
ldig = json.load(open(os.path.join(GOLDEN, "lzma_digests.json")))
lcases = lz_inputs()
for f in ("sample.xls", "sample.jpg", "sample_pgm_100k.bin"):
    lcases[f] = open(os.path.join(GOLDEN, f), "rb").read()
for key, want in limited(ldig.items()):
    name, m = key.split("|")
    data = lcases[name]
    src = os.path.join(work, "in.bin")
    open(src, "wb").write(data)
    arc = os.path.join(work, "out.zip")
    if os.path.exists(arc):
        os.remove(arc)
    subprocess.run(zipada + [{"15": "-el0", "16": "-el1", "17": "-el2", "18": "-el3"}[m], arc, src], check=True, stdout=subprocess.DEVNULL, cwd=work)
    z = open(arc, "rb").read()
    sig, ver, flag, method, tm, crc, csize, usize, nl, xl = struct.unpack("<4sHHHIIIIHH", z[:30])
    payload = z[30 + nl + xl:30 + nl + xl + csize]
    checked += 1
    if want["rc"] == 1:                                       # compression_ok = False: stored
        ok = method == 0 and payload == data
    else:
        ok = method == 14 and csize == want["size"] and hashlib.sha256(payload).hexdigest() == want["sha256"]
    if not ok:
        bad += 1
        print("DIFFERENT: %s LZMA method %s: zipada wrote %d bytes (zip method %d), the oracle %s" % (name, m, csize, method, want["size"]))
print("%d streams compared, %d different" % (checked, bad))
if any("zipada_stub" in a for a in zipada):
    print("REHEARSAL with oracle/zipada_stub.py: the harness works end to end; nothing is pinned (the oracle was compared with itself)")
else:
    print("PINNED: the oracle's streams are the Ada binary's" if bad == 0 else "NOT pinned")
sys.exit(1 if bad else 0)
