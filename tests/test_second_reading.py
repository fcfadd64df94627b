"""The C oracle's entropy stage against a SECOND, independent reading of the same Ada lines
(oracle/second_reading/deflate_entropy.py, pure Python, written from the reference's text without
looking at oracle/zada_oracle.c): streams, block decisions (first atom, atoms, format, exact bits),
cuts and every similarity distance of the Taillaule scan must be equal.

This pins NOTHING to the Ada binary (parity stays unpinned: no GNAT here).  It halves the chance that
one misreading of zip-compress-deflate.adb:238-318, 549-704, 1105-1269, 1327-1408 or of
huffman-encoding-length_limited_coding.adb:46-280 sits under every digest of tests/golden/."""
import os
import sys

import numpy as np
import pytest

from _common import ROOT, METHODS, edge_inputs, oracle, oracle_deflate, oracle_tokens, silesia_mix

sys.path.insert(0, os.path.join(ROOT, "oracle", "second_reading"))
import deflate_entropy as second  # noqa: E402

FORMAT = {"stored": 0, "fixed": 1, "dynamic": 2, "dynamic_rle": 3, "recycled": 4}


def _compare(name, d, m):
    ob, cuts, sim = [], [], []
    rc, z, _ = oracle_deflate(d, m, ob, cuts, sim)
    tr = []
    z2 = second.deflate_from_tokens(d, oracle_tokens(d, m), m, tr)
    if rc != 0:
        # Compression_inefficient: the oracle stops where the reference's Write_Block does (zip-compress.adb:479-486),
        # the second reading has no such rule -- its stream must be no shorter than the input either
        assert rc == 1 and len(z2) >= len(d), (name, m, rc, len(z2), len(d))
        return None
    assert z2 == z, "%s method %d: the two readings write different streams (%d / %d bytes)" % (name, m, len(z), len(z2))
    assert [(a, b, FORMAT[f], bits) for k, a, b, f, bits in (t for t in tr if t[0] == "block")] == ob, (name, m)
    assert [(t[1], t[2]) for t in tr if t[0] == "cut"] == cuts, (name, m)
    assert [(t[1], t[2], t[3]) for t in tr if t[0] == "similar"] == [s[:3] for s in sim], (name, m)
    return ob


def test_streams_blocks_cuts_and_distances_on_the_edge_matrix():
    """Every edge input up to 400 KB x the five methods (the larger format inputs: next test)."""
    seen = set()
    n = 0
    for name, d in edge_inputs().items():
        if len(d) > 400000:
            continue
        for m in METHODS:
            if m == 7 and len(d) > 120000:            # (Deflate_0: one atom per byte, the slowest in Python -- the small inputs and two format inputs below cover it)
                continue
            ob = _compare(name, d, m)
            if ob is not None:
                n += 1
                seen.update(b[2] for b in ob)
    assert n > 140 and seen >= {1, 2, 3, 4}, (n, seen)


@pytest.mark.parametrize("name,methods", [("text_rand_text", (7, 10)), ("fixedlike_mix", (8, 10)), ("mix_1m_off", (7, 9)), ("copies_1500k", (10,))])
def test_the_format_inputs(name, methods):
    """Stored blocks in mid stream incl. the halving of 65 536-atom blocks (text_rand_text), fixed / recycled in mid stream
    (fixedlike_mix), the null-slice cut at atom 750 of even flushes (copies_1500k, :1372), several ring laps (mix_1m_off)."""
    d = edge_inputs()[name]
    formats = set()
    for m in methods:
        ob = _compare(name, d, m)
        if ob is not None:
            formats.update(b[2] for b in ob)
    if name == "text_rand_text":
        assert 0 in formats
    if name == "fixedlike_mix":
        assert {1, 4} <= formats


def test_length_limited_coding_against_the_oracle():
    """The reference's three test_llhc.adb input vectors (tests/golden/llhc_vectors.json) and random count vectors with many ties
    (where the quicksort's order decides): same lengths from both readings, for the three instantiations 288/15, 32/15, 19/7."""
    import ctypes
    import json
    O = oracle()

    def oracle_llhc(freq, limit):
        f = np.array(freq, dtype=np.uint64)
        bl = np.zeros(len(freq), dtype=np.int32)
        O.zo_llhc(f.ctypes.data, len(freq), limit, bl.ctypes.data)
        return bl.tolist()
    with open(os.path.join(ROOT, "tests", "golden", "llhc_vectors.json")) as f:
        vec = json.load(f)
    for v in vec:
        assert second.length_limited_coding(list(v["freq"]), v["max_bits"]) == oracle_llhc(v["freq"], v["max_bits"]) == v["lengths"]
    rs = np.random.RandomState(11)
    for trial in range(200):
        n, limit = ((288, 15), (32, 15), (19, 7))[trial % 3]
        hi = (2, 4, 40, 100000)[trial % 4]
        freq = rs.randint(0, hi, n).tolist()
        if trial % 5 == 0:
            freq = [x if rs.rand() < 0.3 else 0 for x in freq]
        assert second.length_limited_coding(list(freq), limit) == oracle_llhc(freq, limit), (trial, n, limit)


def test_tweak_for_better_rle_known_answers():
    """Hand-checkable cases of zip-compress-deflate.adb:238-318 (independent of the oracle)."""
    c = [0] * 10
    second.tweak_for_better_rle(c)
    assert c == [0] * 10                                  # all zeros: length runs down to 0, nothing touched
    c = [5, 5, 6, 5, 0, 0]                                # trailing zeros untouched; stride of 4 near 5 collapses to the rounded mean
    second.tweak_for_better_rle(c)
    assert c == [5, 5, 5, 5, 0, 0]
    c = [1, 0, 0, 0, 9]                                   # limit 1, the 9 ends a stride of four with sum 1: upper-rounded mean 0 -> at least 1
    second.tweak_for_better_rle(c)
    assert c == [1, 1, 1, 1, 9]
    c = [0, 0, 0, 7]                                      # a 3-stride of zeros stays zero (sum = 0 -> new_count 0), nothing upgraded to 1
    second.tweak_for_better_rle(c)
    assert c == [0, 0, 0, 7]
