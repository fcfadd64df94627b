"""The synthetic benchmark corpus (zip-ada_amd/csrc/silesia_mix.c; SURVEY.md 8d: "64 KiB segments drawn i.i.d.").

silesia_mix_v1 seeded segment i with (seed + i) * gamma + c, gamma being SplitMix64's own increment: consecutive segments were
shifted copies of one stream of draws (round-4 review, VERDICT.md "What's weak" 2).  v2 seeds every segment with a finalised
value.  These tests pin v1's bytes (the golden digests were taken on v1 inputs) and assert what v1 failed: independence of v2's
segments, and compression ratios of the mix in the range of the Silesia corpus for the three encoder families."""
import bz2
import hashlib
import lzma
import zlib

import numpy as np

from _common import silesia_mix


def test_v1_bytes_are_what_the_golden_digests_were_taken_on():
    assert hashlib.sha256(silesia_mix(1 << 20)).hexdigest() == V1_FIRST_MIB
    assert hashlib.sha256(silesia_mix(200001, class_mask=1, offset=65536 * 9 + 5)).hexdigest() == V1_TEXT_SLICE


def test_both_versions_are_seekable_per_segment():
    for version in (1, 2):
        whole = silesia_mix(300000, version=version)
        assert silesia_mix(100001, offset=65537, version=version) == whole[65537:65537 + 100001]
        assert silesia_mix(0, version=version) == b""


def _recurrences_near_a_segment(d, win=48, slack=64, step=97):
    """How many of the sampled win-byte windows of d occur again 65 536 +- slack bytes further on."""
    hits = tot = 0
    for p in range(0, len(d) - 65536 - slack - win, step):
        w = d[p:p + win]
        tot += 1
        if d.find(w, p + 65536 - slack, p + 65536 + slack + win) >= 0:
            hits += 1
    return hits, tot


def test_v2_segments_are_independent_where_v1_segments_were_shifted_copies():
    v1 = silesia_mix(4 << 20, class_mask=1, version=1)
    v2 = silesia_mix(4 << 20, class_mask=1, version=2)
    h1, t1 = _recurrences_near_a_segment(v1)
    h2, t2 = _recurrences_near_a_segment(v2)
    assert h1 > t1 // 50, (h1, t1)        # the defect this test exists for: v1 repeats itself one segment on
    assert h2 == 0, (h2, t2)
    # neighbouring segments do not start alike either, and equal seeds + different indices never meet
    segs = [v2[i * 65536:i * 65536 + 256] for i in range(64)]
    assert len(set(segs)) == 64


def test_v2_ratios_of_the_mix_are_in_silesia_range():
    """Silesia (doc/sae01_za_pres_1_deflate_v3.pdf slide 7: deflate 31.8 %; LZMA about 23 %, BZip2 about 26 % of 211 938 580 bytes)."""
    d = silesia_mix(16 << 20, version=2)
    rz = len(zlib.compress(d, 9)) / len(d)
    rb = len(bz2.compress(d, 9)) / len(d)
    rx = len(lzma.compress(d, preset=6)) / len(d)
    assert 0.30 <= rz <= 0.38, rz
    assert 0.22 <= rb <= 0.32, rb
    assert 0.20 <= rx <= 0.30, rx
    # every class on its own: nothing that an encoder with a long memory folds away (v1: text 0.0055 under xz)
    for mask in (1, 2, 4, 8, 16):
        c = silesia_mix(2 << 20, class_mask=mask, version=2)
        assert len(lzma.compress(c, preset=6)) / len(c) > 0.12, mask


def test_class_shares_of_v2_follow_the_recipe():
    """45 / 20 / 15 / 10 / 10 per cent of the segments, told apart by their first bytes."""
    d = silesia_mix(2048 * 65536 // 8, version=2)    # 256 segments
    heads = [d[i:i + 16] for i in range(0, len(d), 65536)]
    xml = sum(h.startswith(b"  <record id=") for h in heads)
    db = sum(h[:8].isdigit() and h[8:9] == b"," for h in heads)
    assert 0.10 <= xml / len(heads) <= 0.30 and 0.03 <= db / len(heads) <= 0.18, (xml, db, len(heads))


V1_FIRST_MIB = "3a9a10298dcc1e56b911223de7f4c69590bc5618a18d736e2ccfa784f010eb86"
V1_TEXT_SLICE = "85852e8660c00f455134909e597f7091505cb744b601c5b26c922e3e7af42296"


def test_oracles_round_trip_on_the_benchmark_stream():
    """The three CPU oracles on silesia_mix_v2 itself (their golden digests were taken on v1 inputs): every stream goes back to its input through an
    independent decoder -- zlib (raw inflate), libbz2, liblzma -- and the BZip2 oracle keeps more than one splitting tactic on it."""
    import bz2
    import lzma as _lzma
    import zlib as _zlib
    from _common import oracle_deflate
    from _bzip2 import oracle_encode
    from _lzmah import oracle_lzma, lzma_decode
    d = silesia_mix((1 << 20) + 4321, offset=65536 * 5 - 1000, version=2)
    for method in (10, 8):
        rc, z, crc = oracle_deflate(d, method)
        assert rc == 0 and _zlib.decompress(z, -15) == d and crc ^ 0xFFFFFFFF == _zlib.crc32(d)
    z, blocks = oracle_encode(d[:1 << 20], 2)
    assert bz2.decompress(z) == d[:1 << 20] and len(blocks) >= 1
    small = d[:300000]
    for method in (15, 16, 17, 18):
        rc, z, crc = oracle_lzma(small, method)
        assert rc == 0 and lzma_decode(z, 4) == small and crc ^ 0xFFFFFFFF == _zlib.crc32(small)
