#!/usr/bin/env python3
"""zipada_stub.py -- a stand-in with the command line of the reference's tools/zipada.adb (`zipada -ed3 archive.zip file`), used
ONLY to rehearse oracle/pin_compare.py where no GNAT exists: it maps zipada's method options (tools/zipada.adb:150-214: -edf, -ed0 ..
-ed3, -eb1 .. -eb3, -el0 .. -el3) to Compression_Method'Pos, compresses the file with the ORACLE and writes the one-entry archive with
the oracle's Zip.Create restatement.  Running pin_compare.py against it proves the harness (option mapping, local-header parsing,
stored-entry handling, digest comparison) works end to end; it pins nothing -- the oracle is compared with itself."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
from _common import oracle_zip  # noqa: E402

OPTIONS = {"-edf": 6, "-ed0": 7, "-ed1": 8, "-ed2": 9, "-ed3": 10, "-eb1": 12, "-eb2": 13, "-eb3": 14, "-el0": 15, "-el1": 16, "-el2": 17, "-el3": 18}


def main(argv):
    method = 8                                    # zipada's default is Deflate_1 (tools/zipada.adb:123)
    names = []
    for a in argv:
        if a in OPTIONS:
            method = OPTIONS[a]
        elif a.startswith("-"):
            raise SystemExit("zipada_stub: option %s is not one the pin harness uses" % a)
        else:
            names.append(a)
    if len(names) < 2:
        raise SystemExit("usage: zipada_stub.py [-edf|-ed0..3|-eb1..3|-el0..3] archive.zip file ...")
    arc = names[0] if names[0].lower().endswith(".zip") else names[0] + ".zip"
    entries = [(os.path.relpath(f).replace(os.sep, "/"), open(f, "rb").read()) for f in names[1:]]
    open(arc, "wb").write(oracle_zip(entries, method))


if __name__ == "__main__":
    main(sys.argv[1:])
