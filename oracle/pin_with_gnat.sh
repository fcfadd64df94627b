#!/bin/sh
# pin_with_gnat.sh -- closes the one gap this repository cannot close by itself: byte-level parity of the ORACLE
# (oracle/zada_oracle.c) with the real Zip-Ada encoder.  No Ada toolchain exists in the build image ("gcc -c x.adb" ->
# "cannot execute 'gnat1'"), so every "bit-exact" statement here is "with the CPU restatement"; run this wherever GNAT
# and a checkout of zertovitch/zip-ada are available:
#
#     oracle/pin_with_gnat.sh /path/to/zip-ada
#
# It builds the reference's own zipada tool (gprbuild -P zipada.gpr), compresses every input of the parity matrix
# (tests/_common.py:edge_inputs() + the fixture files) with zipada -edf / -ed0 / -ed1 / -ed2 / -ed3, cuts the raw Deflate
# stream out of each archive and compares size and SHA-256 with tests/golden/deflate_digests.json -- the digests of the
# oracle's streams, which the GPU path is tested against bit for bit; the same for zipada -eb1 .. -eb3 against bzip2_digests.json
# (oracle/zada_oracle_bz2.c) and zipada -el0 .. -el3 against lzma_digests.json (oracle/zada_oracle_lzma.c).  Exit 0: the oracle is
# pinned to the Ada binary.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
REF=${1:-/root/reference}
if ! command -v gprbuild >/dev/null 2>&1 && ! command -v gnatmake >/dev/null 2>&1; then
  echo "GNAT not found (gprbuild / gnatmake) -- parity with the Ada binary stays unpinned"
  if [ "$2" = "--rehearse" ] || [ "$1" = "--rehearse" ]; then      # the second half against a stand-in zipada (the oracle itself): proves the harness only
    WORK=$(mktemp -d)
    trap 'rm -rf "$WORK"' EXIT
    PIN_LIMIT=${PIN_LIMIT:-6} python3 "$HERE/pin_compare.py" "python3 $HERE/zipada_stub.py" "$WORK" "$ROOT" || exit 1
  fi
  exit 3
fi
if [ ! -f "$REF/zipada.gpr" ]; then
  echo "no zipada.gpr under $REF -- pass the path of a zertovitch/zip-ada checkout"
  exit 2
fi
WORK=$(mktemp -d)
trap 'rm -rf "$WORK"' EXIT
cp -r "$REF" "$WORK/ref"
( cd "$WORK/ref" && if command -v gprbuild >/dev/null 2>&1; then gprbuild -q -p -P zipada.gpr -XZip_Build_Mode=Fast zipada.adb; else gnatmake -q -O2 -Izip_lib -Itools tools/zipada.adb; fi )
ZIPADA=$(find "$WORK/ref" -type f -name 'zipada*' -perm -u+x | head -1)
[ -n "$ZIPADA" ] || { echo "zipada was not built"; exit 2; }
# The reference's defect behind unprocessed pending bytes: its own encoder on tests/golden/lzma_defect_input_12000.bin with a
# 5 000-byte dictionary must write the stream the oracle writes (tests/golden/lzma_defect.json) -- which does not decode to the input.
( cd "$WORK/ref" && cp "$HERE/pin_lzma_defect.adb" . && gnatmake -q -O2 -Izip_lib pin_lzma_defect.adb 2>/dev/null ) && {
  "$WORK/ref/pin_lzma_defect" "$ROOT/tests/golden/lzma_defect_input_12000.bin" "$WORK/defect.lzma" 5000
  python3 - "$WORK/defect.lzma" "$ROOT/tests/golden/lzma_defect.json" <<'PY'
import hashlib, json, sys
z = open(sys.argv[1], "rb").read(); want = json.load(open(sys.argv[2]))
same = hashlib.sha256(z).hexdigest() == want["stream_sha256"]
print("LZMA_3, dictionary 5000 on 12 000 bytes: the Ada encoder's stream %s the oracle's (%d bytes; the oracle's does not decode to the input)" % ("EQUALS" if same else "DIFFERS FROM", len(z)))
PY
} || echo "pin_lzma_defect.adb was not built (the defect check is skipped)"
exec python3 "$HERE/pin_compare.py" "$ZIPADA" "$WORK" "$ROOT"
