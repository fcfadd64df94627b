"""CPU tests of the oracle's LZMA half (row f4): oracle/zada_oracle_lzma.c restates lzma-encoding.adb + lz77.adb's BT4.
Parity with the Ada binary is UNPINNED (no GNAT here; oracle/pin_with_gnat.sh closes it where one exists); what these tests
hold: liblzma decodes every stream to the input, liblzma's own encoder writes the same bytes where the coding is forced,
the committed digests, and the reference's documented behaviours."""
import hashlib
import io
import json
import lzma
import os
import zipfile
import zlib

from _common import oracle_zip
from _lzmah import oracle_bt4_sets, lz_inputs, oracle_lzma, oracle_lzma_encode, lzma_decode, LZMA_METHODS

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_digests_and_round_trip():
    dig = json.load(open(os.path.join(GOLDEN, "lzma_digests.json")))
    cases = lz_inputs()
    for f in ("sample.xls", "sample.jpg", "sample_pgm_100k.bin"):
        cases[f] = open(os.path.join(GOLDEN, f), "rb").read()
    seen = 0
    for key, want in sorted(dig.items()):
        name, m = key.split("|")
        d = cases[name]
        assert hashlib.sha256(d).hexdigest() == want["in_sha256"], name
        rc, z, crc = oracle_lzma(d, int(m))
        assert (rc, len(z), hashlib.sha256(z).hexdigest()) == (want["rc"], want["size"], want["sha256"]), key
        assert z[:4] == bytes([16, 2, 5, 0])                       # zip-compress-lzma_e.adb:155-158
        assert lzma_decode(z, 4) == d, key
        assert crc ^ 0xFFFFFFFF == zlib.crc32(d)
        seen += 1
    assert seen == len(cases) * 4


def test_forced_codings_equal_liblzma():
    """Where the coding leaves no choice (no input; literals only, no byte equal to its predecessor) liblzma's encoder must
    write the very bytes: pins the range coder, the literal coder, the end marker and the flush on an independent encoder."""
    for d in (b"", b"a", b"abcdefgh", bytes(range(256)), b"The quick brown fx jumps."):
        for level in (0, 1, 2, 3):
            z, st = oracle_lzma_encode(d, level)
            ds = int.from_bytes(z[1:5], "little")
            ref = lzma.compress(d, format=lzma.FORMAT_ALONE, filters=[{"id": lzma.FILTER_LZMA1, "dict_size": max(ds, 4096), "lc": 3, "lp": 0, "pb": 2}])
            assert ref[0] == z[0] and (ref[1:5] == z[1:5] or level == 0) and ref[5:13] == b"\xff" * 8       # size unknown => end marker
            assert ref[13:] == z[5:], (d, level)


def test_header_and_dictionary_size():
    """lzma-encoding.adb:137-149, 1513-1522: properties byte, String_buffer_size per level."""
    d = bytes(lz_inputs()["mix_256k"][:70000])
    for level, want in ((0, 16), (1, 1 << 15), (2, 1 << 15), (3, 1 << 17)):     # 70000 + 338 -> 131072
        z, _ = oracle_lzma_encode(d, level)
        assert z[0] == 3 + 9 * 0 + 45 * 2 and int.from_bytes(z[1:5], "little") == want
    assert int.from_bytes(oracle_lzma_encode(b"abc", 3)[0][1:5], "little") == 4096            # Min_dictionary_size
    for lc, lp, pb in ((0, 0, 0), (4, 0, 0), (0, 2, 2), (3, 1, 2), (0, 4, 4), (1, 3, 4)):     # liblzma decodes lc + lp <= 4 only
        for level in (1, 2, 3):
            z, _ = oracle_lzma_encode(d, level, lc, lp, pb)
            assert z[0] == lc + 9 * lp + 45 * pb and lzma_decode(z) == d


def test_bt4_tail_quirk_and_windows():
    """lz77.adb:959, 1000-1017: `finishing` is the constant False, so the last Nice_Length - 1 positions of a window fill get no
    tree matches; an input shorter than that is coded like Level_0 (plus short repeats).  Dictionaries smaller than the data
    (window moves, cyclic tree) still give valid streams when every fill brings more than keepSizeAfter bytes."""
    d = b"abc" * 40
    assert oracle_lzma_encode(d, 3)[0][5:] == oracle_lzma_encode(d, 0)[0][5:]
    assert len(oracle_lzma_encode(d, 1)[0]) < len(oracle_lzma_encode(d, 0)[0])
    big = bytes(lz_inputs()["mix_256k"]) * 3
    for ds in (20000, 70000, 300000):
        z, st = oracle_lzma_encode(big, 3, dictionary_size=ds)
        assert lzma_decode(z) == big


def test_reference_defect_behind_pending_bytes_no_fill_took_up():
    """A defect of the reference that the oracle keeps (and the product refuses with ZADA_E_REFERENCE).  Move_Pos leaves lzPos behind readPos for
    pending bytes (lz77.adb:1000-1017, "GdM: This causes cyclicPos and lzpos not being in sync with readPos"); Fill_Window catches up only if the
    fill brought more than keepSizeAfter = 4 369 bytes (:1397-1406, 1431-1437).  After a shorter fill -- the last one of a stream, or every one at
    String_buffer_size 4096 -- the positions that follow are inserted with lzPos short by the gap, a distance into the text before the gap is short
    by as much, and the hash-2 / hash-3 matches compare only one byte at it (:1262-1290): matches that are none get coded (or expanded into the
    encoder's text buffer ahead of the coder), and the stream -- self-consistent, right length -- decodes to something else than the input.
    Zip.Compress.LZMA_E asks for a dictionary of the entry's size (zip-compress-lzma_e.adb:165): the whole entry arrives in the first fill, nothing
    is read behind a gap -- unless the entry is beyond 256 MiB (String_buffer_size is capped at 2 ** 28) and its last fill brings 163 .. 4 368 bytes."""
    from _common import hostcheck
    from _lzmah import lzma_symbols
    H = hostcheck()
    x = bytes(lz_inputs()["mix_256k"][:12000])         # dictionary 5000 -> String_buffer_size 8192: fills of 8 192 and 3 808 bytes
    z, _ = oracle_lzma_encode(x, 3, dictionary_size=5000)
    out, syms = lzma_symbols(z)
    # (the committed vector oracle/pin_with_gnat.sh holds against the Ada encoder itself: oracle/pin_lzma_defect.adb)
    gold = json.load(open(os.path.join(GOLDEN, "lzma_defect.json")))
    assert open(os.path.join(GOLDEN, "lzma_defect_input_12000.bin"), "rb").read() == x and hashlib.sha256(x).hexdigest() == gold["input_sha256"]
    assert (len(z), hashlib.sha256(z).hexdigest(), out == x, hashlib.sha256(out).hexdigest()) == (gold["stream_bytes"], gold["stream_sha256"], gold["decodes_to_input"], gold["decoded_sha256"])
    assert len(out) == len(x) and out != x and syms[-1][1] == "E"
    p = next(i for i in range(len(x)) if out[i] != x[i])
    assert p >= 8192                                                                              # behind the gap (the first fill's last 162 positions)
    # the root: the sequential matcher's own sets hold matches that are none there (and nowhere else)
    cnt, ln, ds = oracle_bt4_sets(x, 5000)
    none = [q for q in range(len(x)) for k in range(cnt[q]) if x[q - ds[q, k]:q - ds[q, k] + ln[q, k]] != x[q:q + ln[q, k]]]
    assert none and min(none) >= 8192 and min(none) <= p
    assert H.hc_bt4_reads_behind_a_gap(len(x), 5000) == 1
    # the entry's size as the dictionary, or fills that all bring more than keepSizeAfter bytes: fine, and the schedule says so
    assert lzma_symbols(oracle_lzma_encode(x, 3)[0])[0] == x and H.hc_bt4_reads_behind_a_gap(len(x), len(x)) == 0
    big = bytes(lz_inputs()["mix_256k"]) * 3
    for dsz in (20000, 70000, 300000):
        assert H.hc_bt4_reads_behind_a_gap(len(big), dsz) == 0
    assert H.hc_bt4_reads_behind_a_gap(1 << 30, 1 << 30) == 0                                       # BASELINE config 4: no fill is short
    assert H.hc_bt4_reads_behind_a_gap((1 << 30) + 1605632, (1 << 30) + 1605632) == 1               # an entry beyond 256 MiB whose last fill is


def test_every_variant_is_taken():
    """The matrix exercises each way of writing a DL code (lzma-encoding.adb:765-829) and each kind of match."""
    dig = json.load(open(os.path.join(GOLDEN, "lzma_digests.json")))
    tot = {17: [0] * 8, 18: [0] * 8}
    for key, v in dig.items():
        m = int(key.split("|")[1])
        if m in tot:
            tot[m] = [a + b for a, b in zip(tot[m], v["choices"])]
    assert all(x > 0 for x in tot[17][:4]) and tot[17][4] == 0 and all(x > 0 for x in tot[17][5:])    # Simple: no split
    assert all(x > 0 for x in tot[18][1:])                                                              # Splitting


def test_zip_semantics():
    """Compress_Data with an LZMA method: Zip format 14, general purpose bit 1 (end marker, zip-create.adb:266-278), stored when
    not smaller; Python's zipfile (liblzma) reads the archive."""
    cases = lz_inputs()
    names = ["text_4096", "mix_256k", "random_66666", "text_0"]
    for m in LZMA_METHODS:
        arc = oracle_zip([(nm, cases[nm]) for nm in names], m)
        with zipfile.ZipFile(io.BytesIO(arc)) as zf:
            assert zf.testzip() is None
            for nm in names:
                info = zf.getinfo(nm)
                if nm in ("random_66666", "text_0"):
                    assert info.compress_type == 0 and info.flag_bits & 2 == 0
                else:
                    assert info.compress_type == 14 and info.flag_bits & 2 == 2
                assert zf.read(nm) == cases[nm]
