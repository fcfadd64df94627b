"""GPU parity tests of the LZMA path (zada_lzma*): the product, through the C ABI, against the oracle's
restatement of lzma-encoding.adb / lz77.adb's BT4 and against the committed digests.  A stream is one chain of dependent steps
(one workgroup codes it), so the matrix goes through zada_lzma_batch -- all its streams at once -- and single calls cover the
other entry points."""
import hashlib
import io
import json
import os
import zipfile
import zlib

import numpy as np
import pytest

from _common import GOLDEN, product, oracle_zip, silesia_mix
from _lzmah import lz_inputs, oracle_lzma, oracle_lzma_encode, lzma_decode, LZMA_METHODS, oracle_bt4_sets, sets_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method", LZMA_METHODS)
def test_streams_bit_exact_vs_oracle_and_digests(encoder, method):
    """Every input of the matrix, one batch per method: payload, return code and CRC equal the oracle's; digests as committed.
    (LZMA_3 on the GPU takes tens of microseconds per byte of one stream: the longest inputs set the time of the launch.)"""
    dig = json.load(open(os.path.join(GOLDEN, "lzma_digests.json")))
    cases = {k: v for k, v in lz_inputs().items() if len(v) <= (110000 if method >= 17 else 400000)}
    names = sorted(cases)
    res = encoder.lzma_batch([cases[k] for k in names], method)
    for name, (rc, z, crc) in zip(names, res):
        d = cases[name]
        orc, oz, ocrc = oracle_lzma(d, method)
        assert (rc, crc) == (orc, ocrc), (name, method)
        assert z == oz, (name, method, len(z or b""), len(oz))
        want = dig["%s|%d" % (name, method)]
        assert (want["rc"], want["size"], want["sha256"]) == (rc, len(z), hashlib.sha256(z).hexdigest()), name
        assert crc ^ 0xFFFFFFFF == zlib.crc32(d)


def test_match_sets_of_the_bt4_producer_equal_the_sequential_matchers(encoder):
    """The stage between the BT4 producer (zada_bt4.hip: one lane per hash-4 bucket, sorted hash-2 / hash-3 predecessors, replayed
    window fills) and the coder: the match set of EVERY position == what the oracle's sequential BT4 (lz77.adb:1234-1361, every position
    read) returns there -- with the entry's size as the dictionary and with smaller ones (several fills, window moves, the pending
    bytes that String_buffer_size 4096 never catches up with)."""
    cases = lz_inputs()
    try:
        for name in sorted(cases):
            d = cases[name]
            if len(d) < 5 or len(d) > 300000:
                continue
            for ds in (0, 3000, 5000, 20000):
                if ds and (ds >= len(d) or len(d) > 120000) and ds != 3000:
                    continue
                encoder.set_knob("lzma_dict", ds)
                got = encoder.lzma_match_sets(d)
                assert sets_equal(oracle_bt4_sets(d, ds or None), got), (name, len(d), ds)
    finally:
        encoder.set_knob("lzma_dict", 0)


def test_one_stream_of_four_mib_and_more(encoder):
    """BASELINE config 4's shape at a size the oracle finishes in seconds: ONE LZMA_3 stream of 4 MiB + 12 345 bytes of the benchmark stream
    (not a repeated slice) through zada_lzma -- the producer's kernels on one big entry (its long buckets, nodes in HBM), the coder in bounded
    launches of 64 Ki positions with feedback -- == the oracle's payload; liblzma decodes it."""
    Z = product()
    d = bytes(Z.silesia_mix((4 << 20) + 12345, seed=0x5A1E51A))
    seen = []
    rc, z, crc = encoder.lzma(d, 18, feedback=lambda pct: seen.append(pct) and False)
    assert (rc, z, crc) == oracle_lzma(d, 18)
    assert lzma_decode(z, 4) == d and seen[0] == 0 and seen[-1] == 100 and len(seen) > 60


def test_one_stream_with_the_producer_in_segments(encoder):
    """ONE LZMA_3 stream coded in launches takes its match sets segment by segment ("lzma_segment": 2 ** k positions; by default 2 ** 20 for
    streams of two segments and more): the walks of segment k + 1 run beside the coder of segment k, a bucket's root travelling from segment to
    segment through a table like the reference's hash4Table (lz77.adb:1247-1251).  The stage: the sets of every position == the sequential
    matcher's, for segments of 8 Ki .. 64 Ki positions, the entry's size as the dictionary and smaller ones (window moves inside segments).
    The stream: payload == the oracle's with segments smaller than, equal to and larger than a launch; an abort leaves the context usable; a pool
    that is too small for the segments sends the stream back to the unsegmented way (same payload, one more rerun counted)."""
    Z = product()
    cases = lz_inputs()
    d = cases["mix_256k"][:200000]
    rep = (cases["mix_256k"][:30000] * 6)[:170001]                # long buckets that span segments
    try:
        for data, ds in ((d, 0), (d, 20000), (rep, 0), (rep, 5000), (d[:70000], 3000)):
            encoder.set_knob("lzma_dict", ds)
            want = oracle_bt4_sets(data, ds or None)
            for seg in (13, 16):
                encoder.set_knob("lzma_segment", seg)
                assert sets_equal(want, encoder.lzma_match_sets(data)), (len(data), ds, seg)
        encoder.set_knob("lzma_dict", 0)
        want = oracle_lzma(d, 18)
        for seg, chunk in ((13, 20000), (14, 16384), (15, 5000), (16, 0)):
            encoder.set_knob("lzma_segment", seg)
            encoder.set_knob("lzma_chunk", chunk)
            assert encoder.lzma(d, 18) == want, (seg, chunk)
        assert dict(encoder.last_timing()).get("#lzma_launches", 0) >= 4
        encoder.set_knob("lzma_dict", 10000)
        encoder.set_knob("lzma_segment", 14)
        z, _ = oracle_lzma_encode(rep, 3, dictionary_size=10000)
        assert encoder.lzma(rep, 18)[1] == bytes([16, 2, 5, 0]) + z
        encoder.set_knob("lzma_dict", 0)
        with pytest.raises(Z.UserAbort):
            encoder.lzma(d, 18, feedback=lambda pct: pct >= 30)
        assert encoder.lzma(d[:50000], 18) == oracle_lzma(d[:50000], 18)
        # an entry small enough for the LDS walks (up to 16 384 bytes) with segments forced on it -- the segmented path returns
        # before those walks are launched and drops the buckets of "small" entries, so the stream was coded without a single tree match; such an
        # entry is no longer marked small when segments are asked for
        for nsmall in (16384, 16383, 9000):
            encoder.set_knob("lzma_segment", 13)
            small = d[777:777 + nsmall]
            assert sets_equal(oracle_bt4_sets(small, None), encoder.lzma_match_sets(small)), nsmall
            assert encoder.lzma(small, 18) == oracle_lzma(small, 18), nsmall
        # Round 5: the overflow pool of the match sets GROWS between the segments (a 512 MiB stream of the benchmark corpus ran out of the pool it
        # started with at 70 % and was coded a second time, all match sets first).  A pool that holds the first segments' long sets (those of more
        # than seven matches book a block each) but not the stream's: no rerun, the pool enlarged at least once, the oracle's payload.
        code = silesia_mix(600000, class_mask=6, version=2)     # (tagged records and source-like lines: one position in a hundred has a long set)
        longsets = oracle_bt4_sets(code, None)[0] > 7
        pool = int(longsets[:3 << 13].sum()) + 1200
        assert int(longsets.sum()) > 2 * pool, "the stream must need more blocks than the pool it starts with"
        want_code = oracle_lzma(code, 18)
        encoder.set_knob("lzma_segment", 13)
        encoder.set_knob("lzma_pool", pool)
        r0, g0 = (dict(encoder.last_timing()).get(k, 0) for k in ("#bt4_reruns", "#bt4_pool_grown"))
        assert encoder.lzma(code, 18) == want_code
        r1, g1 = (dict(encoder.last_timing()).get(k, 0) for k in ("#bt4_reruns", "#bt4_pool_grown"))
        assert r1 == r0 and g1 > g0, (r0, r1, g0, g1)
        encoder.set_knob("lzma_pool_fixed", 1)                 # the round-4 behaviour: run out, start again -- the same payload
        assert encoder.lzma(code, 18) == want_code
        assert dict(encoder.last_timing()).get("#bt4_reruns", 0) > r1
        encoder.set_knob("lzma_pool", 0)
        encoder.set_knob("lzma_segment", 14)
        before = dict(encoder.last_timing()).get("#bt4_reruns", 0)
        encoder.set_knob("lzma_pool", 1)                       # (still with a pool that may not grow)
        assert encoder.lzma(d, 18) == want
        assert dict(encoder.last_timing()).get("#bt4_reruns", 0) >= before + 2    # (out of the segments, then the unsegmented producer's own)
    finally:
        for k in ("lzma_dict", "lzma_chunk", "lzma_segment", "lzma_pool", "lzma_pool_fixed"):
            encoder.set_knob(k, 0)
    # without segments: the same stream
    try:
        encoder.set_knob("lzma_segment", -1)
        assert encoder.lzma(d, 18) == want
    finally:
        encoder.set_knob("lzma_segment", 0)


def test_one_stream_alone_on_four_waves(encoder):
    """A stream coded by zada_lzma has a workgroup of four waves: the one that walks the chain and three helpers that take shares of its forks (the
    literal / shortened-code pair and the cuts of a code to write, Scoring's candidates; zada_lzma.hip "one stream on four waves", knob "lzma_waves").
    Every input of the parity matrix up to 100 KB as a single LZMA_3 stream on four waves and on one: both == the oracle (which the batches' one-wave
    kernel is held against as well); launches of a few thousand positions in between (the helpers are released and taken up again at every launch)."""
    cases = lz_inputs()
    try:
        for name in sorted(cases):
            d = cases[name]
            if len(d) > 100000:
                continue
            want = oracle_lzma(d, 18)
            for waves, chunk in ((4, 0), (1, 0), (4, 3000)):
                encoder.set_knob("lzma_waves", waves)
                encoder.set_knob("lzma_chunk", chunk)
                assert encoder.lzma(d, 18) == want, (name, len(d), waves, chunk)
        # LZMA_2 (no BT4, no helpers) is untouched by the knob
        encoder.set_knob("lzma_waves", 4)
        d = cases["mix_256k"][:60000]
        assert encoder.lzma(d, 17) == oracle_lzma(d, 17)
    finally:
        encoder.set_knob("lzma_waves", 0)
        encoder.set_knob("lzma_chunk", 0)


def test_overflow_pool_of_the_match_sets_too_small(encoder):
    """The producer keeps seven matches of a position next to it and longer sets in blocks of an overflow pool sized by a guess; a pool that
    is too small is counted, not overrun, and the walk runs again with a pool of the counted size (the trees are rebuilt from nothing).
    Forced here with a pool of ONE block ("lzma_pool"): same sets, same payloads."""
    cases = lz_inputs()
    d = cases["mix_256k"][:150000]
    small = [cases["mix_256k"][i * 9000:i * 9000 + 8000 + 37 * i] for i in range(20)]
    want_sets = oracle_bt4_sets(d)
    assert int((want_sets[0] > 7).sum()) > 50                  # (sets of more than seven matches exist: the pool is needed)
    before = dict(encoder.last_timing()).get("#bt4_reruns", 0)
    try:
        encoder.set_knob("lzma_pool", 1)
        assert sets_equal(want_sets, encoder.lzma_match_sets(d))
        assert encoder.lzma(d, 18) == oracle_lzma(d, 18)
        for e, got in zip(small, encoder.lzma_batch(small, 18)):
            assert got == oracle_lzma(e, 18)
    finally:
        encoder.set_knob("lzma_pool", 0)
    encoder.lzma(d[:1000], 18)
    assert dict(encoder.last_timing()).get("#bt4_reruns", 0) >= before + 2     # (the sets and the stream of d; the small entries may have no long set)


def test_single_calls_host_and_device_entry(encoder):
    """zada_lzma and zada_lzma_device on the mixed corpus (every variant of a DL code is taken there), all four methods;
    liblzma decodes what the product wrote."""
    import torch
    Z = product()
    d = lz_inputs()["mix_256k"][:98304]
    for method in LZMA_METHODS:
        rc, z, crc = encoder.lzma(d, method)
        assert (rc, z, crc) == oracle_lzma(d, method), method
        assert lzma_decode(z, 4) == d
    t = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    out = torch.empty(len(d) + 4096, dtype=torch.uint8, device="cuda")
    for method in (16, 18):
        rc, ln, crc = encoder.lzma_device(t.data_ptr(), len(d), out.data_ptr(), out.numel(), method)
        orc, oz, ocrc = oracle_lzma(d, method)
        assert (rc, crc, bytes(out[:ln].cpu().numpy())) == (orc, ocrc, oz), method
    # an input that is not 16-byte aligned in device memory (the entry point takes a copy)
    rc, ln, crc = encoder.lzma_device(t.data_ptr() + 3, 20000, out.data_ptr(), out.numel(), 17)
    assert (rc, bytes(out[:ln].cpu().numpy()), crc) == oracle_lzma(d[3:20003], 17)
    del Z


def test_inefficient_and_small_buffers(encoder):
    """compression_ok = False (zip-compress.adb:479-486) when the payload is not smaller; an output buffer that cannot hold an
    efficient payload is an error, one that cannot hold an inefficient one is not."""
    rng = np.random.default_rng(5)
    rnd = bytes(rng.integers(0, 256, 30000, dtype=np.uint8))
    for method in (15, 16, 18):
        rc, z, crc = encoder.lzma(rnd, method)
        assert rc == 1 and (rc, z, crc) == oracle_lzma(rnd, method)
        rc, z, _ = encoder.lzma(rnd, method, cap=len(rnd))
        assert rc == 1 and z is None
    text = lz_inputs()["text_32768"]
    with pytest.raises(Exception):
        encoder.lzma(text, 16, cap=100)


def test_zip_archive_with_lzma_entries(encoder):
    """Zip.Create with an LZMA method (zip-compress.adb:211-216: format code 14; general purpose bit 1 for the end marker,
    zip-create.adb:266-278): archive bytes == the oracle's Zip.Create restatement, entry by entry and as one batch;
    Python's zipfile (liblzma) reads every entry back."""
    Z = product()
    rng = np.random.default_rng(21)
    mix = Z.silesia_mix(1 << 20)
    entries = [("a/text.txt", bytes(mix[:60000])), ("a/empty", b""), ("b/random.bin", bytes(rng.integers(0, 256, 5000, dtype=np.uint8))),
               ("b/tiny", b"x"), ("c/more.txt", bytes(mix[300000:390000]))]
    for method in (16, 18):
        want = oracle_zip(entries, method)
        zc = Z.ZipCreate(encoder, method)
        for name, d in entries:
            zc.add_stream(name, d)
        assert zc.finish() == want
        zb = Z.ZipCreate(encoder, method)
        zb.add_streams([n for n, _ in entries], [d for _, d in entries])
        got = zb.finish()
        assert got == want
        zf = zipfile.ZipFile(io.BytesIO(got))
        assert zf.testzip() is None and [zf.read(i) for i in zf.infolist()] == [d for _, d in entries]
        assert [i.compress_type for i in zf.infolist()] == [14, 0, 0, 0, 14]
        assert [i.flag_bits & 2 for i in zf.infolist()] == [2, 0, 0, 0, 2]


def test_batch_of_many_small_entries(encoder):
    """zipada's usual workload: many small files.  2 000 entries of 1 .. 12 KiB through one launch; a sample is compared with
    the oracle, all of them are decoded."""
    Z = product()
    rng = np.random.default_rng(33)
    mix = Z.silesia_mix(8 << 20)
    datas = []
    for i in range(2000):
        n = int(rng.integers(1024, 12289))
        o = int(rng.integers(0, len(mix) - n))
        datas.append(bytes(mix[o:o + n]))
    for method in (16, 18):
        res = encoder.lzma_batch(datas, method)
        for i in range(0, len(datas), 40):
            assert res[i] == oracle_lzma(datas[i], method), (i, method)
        for d, (rc, z, crc) in zip(datas, res):
            assert lzma_decode(z, 4) == d and crc ^ 0xFFFFFFFF == zlib.crc32(d)


def test_dictionary_smaller_than_the_entry(encoder):
    """Entries beyond 256 MiB get a dictionary smaller than themselves (lzma-encoding.adb:137-149): BT4's window then moves
    (lz77.adb:1375-1386), its cyclic tree wraps and pending positions are caught up after every fill (:1397-1406).  Exercised at a
    small scale through the reference's own dictionary_size parameter (knob "lzma_dict"): payload == the oracle's for the same value."""
    m = lz_inputs()["mix_256k"]
    try:
        for ds, d in ((20000, m + m[:150000]), (10000, m + m[:47856])):     # 32 KiB / 16 KiB dictionaries: the window moves at 315 / 291 KB
            encoder.set_knob("lzma_dict", ds)
            rc, z, crc = encoder.lzma(d, 18)
            want, _ = oracle_lzma_encode(d, 3, dictionary_size=ds)
            assert z == bytes([16, 2, 5, 0]) + want, ds
            assert lzma_decode(z, 4) == d
    finally:
        encoder.set_knob("lzma_dict", 0)


def test_reference_defect_is_refused_not_written(encoder):
    """Where the reference's own matcher reports matches that are none (positions read behind pending bytes no window fill took up:
    test_lzma_oracle.py::test_reference_defect_behind_pending_bytes_no_fill_took_up) its stream decodes to something else than the input, and the
    product -- whose coder reads the text itself, not a buffer such matches have been copied into -- could only write a third thing.  It verifies
    the match sets it reads on such entries (bt4_reads_behind_a_gap) and refuses at the first match that is none: ZADA_E_REFERENCE, nothing written,
    per entry in a batch, the context usable afterwards.  An entry in the same regime whose sets are all true is coded, == the oracle, and decodes."""
    Z = product()
    m = lz_inputs()["mix_256k"]
    x = bytes(m[:12000])
    try:
        encoder.set_knob("lzma_dict", 5000)
        with pytest.raises(Z.ReferenceDefect):
            encoder.lzma(x, 18)
        with pytest.raises(Z.ReferenceDefect):
            encoder.lzma(bytes(m[50000:70000]), 18, feedback=lambda pct: False)
        # (the sets themselves are the sequential matcher's, matches that are none included)
        assert sets_equal(oracle_bt4_sets(x, 5000), encoder.lzma_match_sets(x))
        # levels below 3 have no BT4: coded as ever
        assert encoder.lzma(x, 17) == oracle_lzma(x, 17)
        # in the regime, but nothing wrong among the sets read: no byte behind the gap (positions 8 030 .. 8 191 are never inserted) occurs in
        # front of it, so no distance goes across it -- 3 275 positions behind it have matches, all of them true
        r = bytes(np.concatenate([np.frombuffer(m[:8030], np.uint8) & 0x7F, np.frombuffer(m[20000:23970], np.uint8) | 0x80]).astype(np.uint8))
        rc, z, crc = encoder.lzma(r, 18)
        want, _ = oracle_lzma_encode(r, 3, dictionary_size=5000)
        assert z == bytes([16, 2, 5, 0]) + want and lzma_decode(z, 4) == r
        # refused in the middle of a stream whose producer works in segments (the walks of the next segment are under way when the coder gives
        # up): the call after it finds the context's buffers its own
        big = bytes(m[:250000])
        encoder.set_knob("lzma_dict", 70000)                  # String_buffer_size 131 072: the second (last) fill brings 118 928 bytes -- no gap
        encoder.set_knob("lzma_segment", 13)
        want_big, _ = oracle_lzma_encode(big, 3, dictionary_size=70000)
        assert encoder.lzma(big, 18)[1] == bytes([16, 2, 5, 0]) + want_big
        tail = bytes(m[:135000])                              # ... 3 928 bytes: read behind a gap, in the 17th segment
        encoder.set_knob("lzma_dict", 70000)
        with pytest.raises(Z.ReferenceDefect):                # (the oracle's stream for it does not decode to the input)
            encoder.lzma(tail, 18)
        assert encoder.lzma(big, 18)[1] == bytes([16, 2, 5, 0]) + want_big
        encoder.set_knob("lzma_segment", 0)
        encoder.set_knob("lzma_dict", 5000)
        # a batch: the refused entries say so, the others are coded
        res = encoder.lzma_batch([x, r, x[:5000], bytes(m[50000:70000])], 18)
        assert [t[0] for t in res] == [Z.E_REFERENCE, 0, 0, Z.E_REFERENCE] and res[0][1] is None and res[3][1] is None
        assert res[1][1] == z and res[2] == encoder.lzma(x[:5000], 18)
    finally:
        encoder.set_knob("lzma_dict", 0)
        encoder.set_knob("lzma_segment", 0)
    assert encoder.lzma(x, 18) == oracle_lzma(x, 18)


def test_stream_in_bounded_launches_feedback_and_abort(encoder):
    """Zip.Compress.LZMA_E drives `feedback` and raises User_abort (zip-compress-lzma_e.adb:78-92).  A stream is coded as a sequence
    of bounded launches with the coder's state (model, match sets, range coder, BT4) parked in device memory in between ("lzma_chunk"
    positions per launch): chunked == one launch == the oracle for every method, whatever the chunk; feedback is monotone 0 .. 100 and
    called between the launches; an abort returns ZADA_ABORTED (UserAbort) and leaves the context usable."""
    Z = product()
    d = bytes(Z.silesia_mix(90000))
    try:
        for method in LZMA_METHODS:
            want = oracle_lzma(d, method)
            for chunk in (-1, 777, 20000):
                encoder.set_knob("lzma_chunk", chunk)
                assert encoder.lzma(d, method) == want, (method, chunk)
        encoder.set_knob("lzma_chunk", 5000)
        for method in (15, 16, 18):
            seen = []
            got = encoder.lzma(d, method, feedback=lambda pct: seen.append(pct) and False)
            assert got == oracle_lzma(d, method)
            assert seen[0] == 0 and seen[-1] == 100 and seen == sorted(seen) and len(seen) >= 90000 // 5000, seen
            with pytest.raises(Z.UserAbort):
                encoder.lzma(d, method, feedback=lambda pct: pct >= 40)
            assert encoder.lzma(d[:30000], method) == oracle_lzma(d[:30000], method)      # usable after an abort
        # the window-move case (dictionary smaller than the entry) across launch boundaries
        m = lz_inputs()["mix_256k"]
        d2 = m + m[:47856]
        encoder.set_knob("lzma_dict", 10000)
        encoder.set_knob("lzma_chunk", 30011)
        rc, z, crc = encoder.lzma(d2, 18)
        want, _ = oracle_lzma_encode(d2, 3, dictionary_size=10000)
        assert z == bytes([16, 2, 5, 0]) + want
    finally:
        encoder.set_knob("lzma_dict", 0)
        encoder.set_knob("lzma_chunk", 0)


def test_a_stream_stopped_by_its_feedback_goes_on_in_another_context():
    """zada_lzma_export_state / zada_lzma_import_state (round 6): a stream that its feedback stops between two launches (User_abort,
    zip-compress-lzma_e.adb:78-92) is taken up by ANOTHER context from the exported state -- same input, same method; the match sets are made
    again, the launches up to the state's position code nothing -- and the exported stream bytes followed by the resumed call's bytes from that length on
    are the oracle's payload.  LZMA_3 (BT4 producer in segments, helper waves) and LZMA_2 (tokens of the Info-Zip matcher), stops at two places, and a
    state handed to the wrong input is refused by the coder's own end check or gives a different stream -- never silently the right one."""
    from _common import product, silesia_mix
    from _lzmah import oracle_lzma
    Z = product()
    d = silesia_mix((1 << 20) + 4321, version=2)
    for method, chunk in ((18, 20000), (17, 60000)):
        want = oracle_lzma(d, method)
        for stop_pct in (30, 80):
            a = Z.Encoder(0)
            a.set_knob("lzma_chunk", chunk)
            seen = []
            with pytest.raises(Z.UserAbort):
                a.lzma(d, method, feedback=lambda pct: seen.append(pct) or pct >= stop_pct)
            state, head, pos = a.lzma_export_state(len(d) + 4096)
            a.close()
            assert 0 < pos < len(d) and seen[-1] >= stop_pct
            b = Z.Encoder(0)                                            # another context: nothing of the first one's device memory is there
            b.set_knob("lzma_chunk", chunk)
            b.lzma_import_state(state)
            rc, z, crc = b.lzma(d, method)
            assert (rc, head + z[len(head):], crc) == want, (method, stop_pct, pos, len(head))
            rc2, z2, crc2 = b.lzma(d, method)                           # (the state was for one call: the next one starts at the first byte)
            assert (rc2, z2, crc2) == want
            # a state is checked against the stream it is handed to before the kernel takes its counters as they are: another length, another method's kind
            # of state, a damaged blob -- refused (ZADA_E_INVALID), the state is spent, and the context codes the next stream as ever
            for wrong_input, wrong_method in ((d[:-1], method), (d, 35 - method)):
                b.lzma_import_state(state)
                with pytest.raises(Z.ZadaError, match="imported state"):
                    b.lzma(wrong_input, wrong_method)
            for damage in (lambda s: b"\x02" + s[1:], lambda s: s[:-1], lambda s: b""):
                with pytest.raises(Z.ZadaError, match="not the state of a stopped stream"):
                    b.lzma_import_state(damage(state))
            assert b.lzma(d, method) == want
            b.close()
