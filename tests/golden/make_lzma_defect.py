"""Makes tests/golden/lzma_defect_input_12000.bin and lzma_defect.json: the input and the ORACLE's stream (digest) for
LZMA.Encoding.Encode (Level_3, dictionary_size 5000) -- the regime in which the reference's BT4 matcher reads positions behind pending bytes
that no window fill took up and reports matches that are none.  oracle/pin_with_gnat.sh runs the reference's own encoder
(oracle/pin_lzma_defect.adb) on the same input and compares.  Run from the repository root: python tests/golden/make_lzma_defect.py"""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _lzmah import lz_inputs, oracle_lzma_encode, lzma_symbols
x = bytes(lz_inputs()["mix_256k"][:12000])
open(os.path.join(HERE, "lzma_defect_input_12000.bin"), "wb").write(x)
z, _ = oracle_lzma_encode(x, 3, dictionary_size=5000)
out, syms = lzma_symbols(z)
json.dump({"input": "lzma_defect_input_12000.bin (the first 12 000 bytes of the parity matrix's mix_256k)", "input_sha256": hashlib.sha256(x).hexdigest(),
           "call": "LZMA.Encoding.Encode (level => Level_3, dictionary_size => 5000, end_marker => True), lc 3, lp 0, pb 2, no size info",
           "stream_bytes": len(z), "stream_sha256": hashlib.sha256(z).hexdigest(),
           "decodes_to_input": out == x, "decoded_sha256": hashlib.sha256(out).hexdigest(),
           "first_wrong_byte": next(i for i in range(len(x)) if out[i] != x[i]),
           "made_by": "tests/golden/make_lzma_defect.py (oracle/zada_oracle_lzma.c); oracle/pin_with_gnat.sh compares the Ada encoder's stream with stream_sha256"},
          open(os.path.join(HERE, "lzma_defect.json"), "w"), indent=1)
