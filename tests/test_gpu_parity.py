"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
the oracle on the same inputs -- bit-exact (integer/byte work) -- against the committed golden
digests, and through size-independent properties at the benchmark's full size."""
import hashlib
import io
import json
import os
import zipfile
import zlib

import numpy as np
import pytest

from _common import GOLDEN, METHODS, edge_inputs, few_symbol_inputs, oracle_deflate, oracle_tokens, oracle_zip, product, silesia_mix

pytestmark = pytest.mark.gpu
FIXTURE_FILES = ("sample.xls", "sample.jpg", "sample_pgm_100k.bin")


def gpu_deflate(enc, d, method):
    za = product()
    try:
        out, crc = enc.deflate(d, method)
        return 0, out, crc
    except za.CompressionInefficient:
        return 1, b"", None


@pytest.mark.parametrize("method", METHODS)
def test_tokens_bit_exact_vs_oracle(encoder, method):
    """LZ77 stage alone (lz77.adb:460-943 semantics): identical token streams."""
    for name, d in edge_inputs().items():
        a = oracle_tokens(d, method)
        b = encoder.lz77_tokens(d, method)
        assert len(a) == len(b) and (a == b).all(), name


@pytest.mark.parametrize("method", METHODS)
def test_stream_bit_exact_vs_oracle(encoder, method):
    """Whole path: same bytes, same CRC register, same compression_ok, same block decisions."""
    for name, d in edge_inputs().items():
        ob = []
        rc, ref, crc = oracle_deflate(d, method, ob)
        rc2, out, crc2 = gpu_deflate(encoder, d, method)
        assert rc == rc2, name
        if rc == 0:
            assert out == ref, name
            assert crc2 == crc, name
        if method != 6:
            # block decisions and their exact bit costs (zip-compress-deflate.adb:1244-1266).  When the stream is not
            # smaller than the input the reference stops at the first 1 MiB flush that says so (zip-compress.adb:479-486):
            # its trace is then a prefix of the complete list.
            gb = [(int(a), int(b), int(c), int(e)) for a, b, c, e in encoder.last_blocks()]
            assert gb == ob if rc == 0 else (len(ob) > 0 or len(d) == 0) and gb[:len(ob)] == ob, name


def test_golden_digests(encoder):
    """Oracle-free comparison with the committed fixtures (tests/golden/deflate_digests.json)."""
    dig = json.load(open(os.path.join(GOLDEN, "deflate_digests.json")))
    cases = dict(edge_inputs())
    for f in FIXTURE_FILES:
        cases[f] = open(os.path.join(GOLDEN, f), "rb").read()
    for key, want in dig.items():
        name, m = key.rsplit("|", 1)
        d = cases[name]
        rc, out, crc = gpu_deflate(encoder, d, int(m))
        assert rc == want["rc"], key
        if rc == 0:
            assert len(out) == want["size"] and hashlib.sha256(out).hexdigest() == want["sha256"], key
            assert crc ^ 0xFFFFFFFF == want["crc"], key


def test_multi_flush_and_long_streams(encoder):
    """More than 65536 atoms (several Flush_half_buffer rounds, odd/even halves, look-behind
    windows) and all five synthetic classes."""
    for mask, n in ((0x1F, 6 << 20), (1, 3 << 20), (16, 1 << 20), (8, 2 << 20), (0x1F, (4 << 20) + 12345)):
        d = silesia_mix(n, class_mask=mask)
        for method in (10, 9, 8):
            rc, ref, crc = oracle_deflate(d, method)
            rc2, out, crc2 = gpu_deflate(encoder, d, method)
            assert rc == rc2 and (rc != 0 or (out == ref and crc == crc2)), (mask, n, method)


def test_splitter_trace_equals_the_oracles(encoder):
    """The reference's trace events (zip-compress-deflate.adb:83-90): similarity distance at every test point
    (:480-488), cut position and step level (:1384-1390), from the product (zada_last_trace) == from the oracle."""
    cases = dict(edge_inputs())
    for name in ("mix_1m_off", "text_rand_text", "copies_1500k", "fixedlike_mix"):
        d = cases[name]
        for method in (10, 9, 8):
            cuts, sim = [], []
            rc, ref, crc = oracle_deflate(d, method, None, cuts, sim)
            rc2, out, crc2 = gpu_deflate(encoder, d, method)
            assert rc == rc2 == 0
            tr = [(int(a), int(b), int(c)) for a, b, c in encoder.last_trace()]
            dist = {a: b for a, b, _ in tr}
            assert len(sim) > 0 and all(dist.get(w) == dd for w, dd, thr, _ in sim), (name, method)
            assert sorted(set(w for w, _, _, _ in sim)) == sorted(dist), (name, method)   # the same test points
            assert [(a, c) for a, _, c in tr if c] == cuts, (name, method)


def test_few_symbol_random_data(encoder):
    """2-, 4- and 16-symbol uniform random data at 4-6 MiB with the default budget: every position has thousands of
    candidates (lz77.adb:715-825 at chain 4096 / 1024), the demand-driven match finder at its worst."""
    for name, d in few_symbol_inputs().items():
        for method in ((10,) if name == "sym2_4m" else (10, 9)):
            ob = []
            rc, ref, crc = oracle_deflate(d, method, ob)
            rc2, out, crc2 = gpu_deflate(encoder, d, method)
            assert rc == rc2 == 0 and out == ref and crc == crc2, (name, method)
            assert [(int(a), int(b), int(c), int(e)) for a, b, c, e in encoder.last_blocks()] == ob, (name, method)


def test_incompressible_entries_are_stored(encoder):
    """Compression_inefficient (zip-compress.adb:479-486, 224-237): 2-4 MiB of random bytes, every method.  The
    fixed-code stream of random bytes is 5 % LARGER than the input -- larger than the encoder's output workspace --
    and the Taillaule methods end in stored blocks + headers: rc 1, then Store with the CRC of the data."""
    za = product()
    rs = np.random.RandomState(9)
    for n in ((2 << 20) + 17, 4 << 20):
        d = bytes(rs.randint(0, 256, n).astype(np.uint8))
        for method in METHODS:
            rc, _, crc = oracle_deflate(d, method)
            assert rc == 1
            with pytest.raises(za.CompressionInefficient):
                encoder.deflate(d, method)
            payload, c2, zt = encoder.compress_data(d, method)
            assert zt == 0 and payload == d and c2 == zlib.crc32(d)
    # near-random data with a long match in every block: stored format impossible (a match longer than 14), the dynamic
    # header makes the stream larger than the input
    blk = bytes(rs.randint(0, 256, 300).astype(np.uint8))
    parts = []
    for _ in range(40):
        parts.append(bytes(rs.randint(0, 256, 60000).astype(np.uint8)) + blk[:20])
    d = b"".join(parts)
    for method in (8, 10):
        rc, ref, crc = oracle_deflate(d, method)
        rc2, out, crc2 = gpu_deflate(encoder, d, method)
        assert rc == rc2 and (rc != 0 or out == ref)


def test_never_resynchronising_inputs(encoder):
    """Periodic data: the speculative chunk parses never meet the true parse; the splice must still
    converge to the sequential result."""
    for d in (bytes(300000), b"ab" * 150000, b"abc" * 100000 + b"x" + b"abcd" * 1000, bytes(range(256)) * 1200):
        for method in (10, 8):
            rc, ref, crc = oracle_deflate(d, method)
            rc2, out, crc2 = gpu_deflate(encoder, d, method)
            assert rc == rc2 and out == ref and crc == crc2


def test_degenerate_inputs_converge_quickly(encoder):
    """Constant and exactly periodic data: the speculative parses never meet the true one, so the splice would
    advance one chunk per round (65 536 rounds for 64 MiB).  k_fix_forward carries it through the runs of
    maximal matches, and a crawling splice triggers the exact search of everything still guessed (lz_stage).
    Parity with the oracle, and a bound on the number of rounds."""
    row = b"0001234,ABCD,some field,99\n" * 40 + b"0001235,ABCE,some field,98\n"
    n = 8 << 20
    for d in (bytes(n), b"ab" * (n // 2), (bytes(range(37)) * (n // 37 + 1))[:n], (row * (n // len(row) + 1))[:n]):
        rc, ref, crc = oracle_deflate(d, 10)
        rc2, out, crc2 = gpu_deflate(encoder, d, 10)
        assert rc == rc2 and out == ref and crc == crc2
        rounds = dict(encoder.last_timing())
        assert rounds["#splice_rounds"] <= 80 and rounds["#demand_rounds"] <= 3, rounds


def test_lds_atomics_return_in_lane_order(tmp_path):
    """The radix passes of k_prev_links take an element's rank from the value an LDS atomicAdd returns, which is
    stable only if lanes of one instruction that hit the same counter are served in lane order.  That is a property
    of the hardware, not of the ISA document: check it here (65 M samples per digit count)."""
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "lds_atomic_order.hip")
    exe = str(tmp_path / "lds_atomic_order")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", src, "-o", exe], check=True, capture_output=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    lines = [l for l in out.splitlines() if l.startswith("digits")]
    assert len(lines) == 7 and all(" results 0 of " in l for l in lines), out


def test_demand_driven_matching_is_budget_independent(encoder):
    """The first pass gives every position a bounded search and the exact search only happens where a parse
    lands.  The stream must not depend on the bound: tiny budget (nearly everything is a guess
    and gets demanded), default, and unbounded (no guesses, the former single-pass pipeline) -- all equal
    to the oracle, on every synthetic class and on data built to have very long chains."""
    rep = (b"0001234,ABCD,some field,99\n" * 40 + b"0001235,ABCE,some field,98\n") * 900
    cases = [silesia_mix(3 << 20), silesia_mix(1 << 20, class_mask=8), silesia_mix(1 << 20, class_mask=2), rep[: (1 << 20) + 77],
             bytes(200000), b"abcdefgh" * 50000]
    try:
        for d in cases:
            rc, ref, crc = oracle_deflate(d, 10)
            for budget in (1, 2, 6, 0):
                encoder.set_knob("budget", budget)
                rc2, out, crc2 = gpu_deflate(encoder, d, 10)
                assert rc == rc2 and out == ref and crc == crc2, (len(d), budget)
            # safety valve of the demand loop: after one round everything that is still a guess is searched
            encoder.set_knob("budget", 1); encoder.set_knob("max_demand_rounds", 1)
            rc2, out, crc2 = gpu_deflate(encoder, d, 10)
            encoder.set_knob("max_demand_rounds", 12)
            assert rc == rc2 and out == ref and crc == crc2 and dict(encoder.last_timing())["#demand_rounds"] <= 2, len(d)
            for method in (9, 8):
                encoder.set_knob("budget", 1)
                rc, ref, crc = oracle_deflate(d, method)
                rc2, out, crc2 = gpu_deflate(encoder, d, method)
                assert rc == rc2 and out == ref and crc == crc2, (len(d), method)
    finally:
        encoder.set_knob("max_demand_rounds", 12)
        encoder.set_knob("budget", -1)


def test_compress_data_store_fallback_and_archive_bytes(encoder):
    """Zip.Compress.Compress_Data + Zip.Create bytes == the oracle's archive; readable by zipfile."""
    za = product()
    entries = [("a/text.txt", silesia_mix(200000, class_mask=1)), ("b\\rand.bin", bytes(np.random.RandomState(3).randint(0, 256, 3000).astype(np.uint8))),
               ("empty", b""), ("mix.bin", silesia_mix(1 << 20))]
    for method in (10, 8):
        zc = za.ZipCreate(encoder, method)
        for name, data in entries:
            zc.add_stream(name, data)
        got = zc.finish()
        assert got == oracle_zip(entries, method)
        zf = zipfile.ZipFile(io.BytesIO(got))
        assert zf.testzip() is None
        for (name, data), info in zip(entries, zf.infolist()):
            assert zf.read(info) == data


def test_feedback_and_abort(encoder):
    za = product()
    d = silesia_mix(1 << 20)
    seen = []
    encoder.deflate(d, 10, feedback=lambda pct: seen.append(pct) or False)
    assert seen[0] == 0 and seen[-1] == 100 and seen == sorted(seen)
    with pytest.raises(za.UserAbort):
        encoder.deflate(d, 10, feedback=lambda pct: pct >= 5)


def test_device_resident_entry_point(encoder):
    """zada_deflate_device (HBM in, HBM out) == host-buffer entry point."""
    import torch
    za = product()
    d = silesia_mix(2 << 20)
    t_in = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    t_out = torch.zeros(len(d) + 4096, dtype=torch.uint8, device="cuda")
    rc, ol, crc = encoder.deflate_device(t_in.data_ptr(), len(d), t_out.data_ptr(), len(d) + 4096, 10)
    ref, crc2 = encoder.deflate(d, 10)
    assert rc == 0 and bytes(t_out[:ol].cpu().numpy()) == ref and crc == crc2


def test_batch_of_entries_equals_single_calls(encoder):
    """zada_deflate_batch (entries compressed several at a time on separate streams) == one zada_deflate per entry
    == the oracle, in entry order, including empty, stored-fallback and multi-segment entries."""
    rng = np.random.default_rng(11)
    mix = silesia_mix(3 << 20)
    datas = [b"", b"a", bytes(rng.integers(0, 256, 5000, dtype=np.uint8))]          # empty, tiny, incompressible
    off = 0
    for k in range(40):
        ln = int(rng.integers(1, 200000))
        datas.append(bytes(mix[off:off + ln])); off = (off + ln) % (len(mix) - 200000)
    datas.append(bytes(mix[:2 << 20]))                                                 # several segments
    res = encoder.deflate_batch(datas, 10)
    assert len(res) == len(datas)
    for i, (d, (rc, out, crc)) in enumerate(zip(datas, res)):
        rc1, ref, crc1 = gpu_deflate(encoder, d, 10)
        assert rc == rc1, i
        if rc == 0:
            assert out == ref and crc == crc1, i
        if i % 7 == 0:
            rc2, oref, ocrc = oracle_deflate(d, 10, [])
            assert rc2 == rc and (rc != 0 or (oref == out and ocrc == crc)), i


def test_batch_is_one_launch_sequence_and_bit_exact(encoder):
    """zada_deflate_batch: many small entries through ONE launch sequence (per-entry layout of the LZ buffer, flush grid,
    chooser state and output).  Every entry's stream, CRC and return code == the oracle's, for entry sizes around the
    segment (32 KiB) and flush boundaries, empty and 1-byte entries, incompressible ones and ones with stored blocks."""
    rng = np.random.default_rng(5)
    mix = silesia_mix(6 << 20)
    text = silesia_mix(1 << 20, class_mask=1)
    datas = [b"", b"a", b"ab" * 5, bytes(rng.integers(0, 256, 5000, dtype=np.uint8)), text[:32768], text[:32767], text[:32769], text[:65536],
             text[:65537], text[:3], bytes(40000), b"abc" * 20000, mix[:300000], text[:400000],
             text[:150000] + bytes(rng.integers(0, 256, 100000, dtype=np.uint8)) + text[150000:300000]]
    off = 0
    for k in range(300):
        ln = int(rng.integers(1, 60000))
        datas.append(bytes(mix[off:off + ln])); off = (off + ln) % (len(mix) - 70000)
    datas.append(b"")
    for method, batch_mib in ((10, 512), (8, 512), (9, 1), (7, 512), (6, 512), (6, 1)):   # (1 MiB: the entries go through many small batches)
        encoder.set_knob("batch_mib", batch_mib)
        try:
            res = encoder.deflate_batch(datas, method)
        finally:
            encoder.set_knob("batch_mib", 512)
        assert len(res) == len(datas)
        for i, (d, (rc, out, crc)) in enumerate(zip(datas, res)):
            rc2, ref, crc2 = oracle_deflate(d, method)
            assert rc == rc2, (i, len(d), method)
            assert crc == crc2, (i, len(d), method)
            if rc == 0:
                assert out == ref, (i, len(d), method)


def test_archive_of_many_small_files_equals_the_oracles(encoder):
    """zipada's usual workload (tools/zipada.adb:126-134): many small files into one archive.  ZipCreate.add_streams
    (one batch) writes the bytes Zip.Create writes entry after entry, incl. stored entries."""
    za = product()
    rng = np.random.default_rng(8)
    mix = silesia_mix(4 << 20)
    entries, off = [], 0
    for k in range(120):
        ln = int(rng.integers(0, 50000))
        d = bytes(mix[off:off + ln]) if k % 9 else bytes(rng.integers(0, 256, ln, dtype=np.uint8))
        entries.append(("dir/f%03d.txt" % k, d)); off = (off + ln) % (len(mix) - 60000)
    zc = za.ZipCreate(encoder, 10)
    zc.add_streams([n for n, _ in entries], [d for _, d in entries])
    got = zc.finish()
    assert got == oracle_zip(entries, 10)
    zf = zipfile.ZipFile(io.BytesIO(got))
    assert zf.testzip() is None and [zf.read(i) for i in zf.infolist()] == [d for _, d in entries]


def test_full_size_properties(encoder):
    """BASELINE config C2 size (1 GiB, Deflate_3): properties that do not need the oracle at full size --
    the stream inflates back to the input (independent inflater), CRC equals zlib's, and the first
    16 MiB compressed alone equal the oracle's stream (the encoder is a pure function).  The benchmark stream (silesia_mix_v2)."""
    za = product()
    n = 1 << 30
    d = za.silesia_mix(n, version=2)
    out, crc = encoder.deflate(d, 10)
    dec = zlib.decompressobj(-15)
    h = hashlib.sha256()
    total = 0
    view = memoryview(out)
    for off in range(0, len(out), 1 << 24):
        chunk = dec.decompress(view[off:off + (1 << 24)])
        h.update(chunk); total += len(chunk)
    tail = dec.flush(); h.update(tail); total += len(tail)
    assert total == n and h.digest() == hashlib.sha256(d).digest()
    assert crc ^ 0xFFFFFFFF == zlib.crc32(d)
    assert 0.30 < len(out) / n < 0.42
    head = d[:16 << 20].tobytes()
    rc, ref, _ = oracle_deflate(head, 10)
    got, _ = encoder.deflate(head, 10)
    assert rc == 0 and got == ref
    # The size-dependent defaults, bit for bit at this size (an inflate round trip cannot see a missed cross link): the host path's
    # pieces of 2 048 segments with runs of 8, the device path's runs of 16, no runs at all, and the flagged chunks parsed again with guesses in
    # every round (rounds 1-5) or by the exact parse in every round instead of for short lists only (round 6), the cross-segment continuation in
    # one pass without its Bloom filter -- six ways through the link stage and the demand loop, one stream.
    import torch
    want = hashlib.sha256(out).digest()
    d_in = torch.from_numpy(d).cuda()
    d_out = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
    try:
        for knobs in ({}, {"link_run": 1}, {"exact_respec": 0}, {"exact_respec": 1 << 30}, {"cd_filter": 0}):
            for k, v in knobs.items():
                encoder.set_knob(k, v)
            rc2, ol, crc2 = encoder.deflate_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, 10)
            assert rc2 == 0 and crc2 == crc and ol == len(out), knobs
            assert hashlib.sha256(d_out[:ol].cpu().numpy().tobytes()).digest() == want, knobs
            for k in knobs:
                encoder.set_knob(k, {"link_run": 0, "exact_respec": 32768, "cd_filter": 1}[k])
    finally:
        encoder.set_knob("link_run", 0); encoder.set_knob("exact_respec", 32768); encoder.set_knob("cd_filter", 1)


def test_link_stage_in_runs_of_segments(encoder):
    """Round 5: a workgroup of k_prev_links takes a RUN of segments one after the other and makes the cross links of all but the run's first
    segment itself (tails read in key order while the sorted order is in LDS, occupancy bit maps instead of initialised tables, the searches that
    end at a cross link settled against the previous segment's bytes); k_cross_links is left with the runs' first segments.  By default only from
    64 MiB on ("link_run" 0): here forced to 2, 4 and 16 segments on small inputs -- tokens of every edge input, streams of several MiB of every
    class of both corpus versions, different inputs one after the other on one context (stale tables under fresh bit maps), batches (entries'
    first segments inside runs) and an input in shards -- all equal to the oracle / to the run-less path."""
    rng = np.random.default_rng(5)
    a = silesia_mix(3 << 20)
    b = silesia_mix((5 << 20) + 4321, version=2)
    cth = bytes((rng.integers(0, 3, 1 << 20) + 65).astype(np.uint8))
    streams = [a, b, cth, silesia_mix(2 << 20, class_mask=16, version=2), a[:1 << 20], bytes(300000) + b[:1 << 20], silesia_mix(1 << 20, class_mask=8, version=2), b"abcdefgh" * 40000 + a[:200000]]
    refs = {(i, m): oracle_deflate(x, m) for i, x in enumerate(streams) for m in (10, 8)}
    mix = silesia_mix(3 << 20, version=2)
    datas = [b"", b"a", bytes(mix[:40000]), bytes(mix[40000:240000]), bytes(mix[:(1 << 20) + 17]), bytes(mix[100000:100000 + 70000]), bytes(mix[1 << 20:(1 << 20) + 32768]), bytes(mix[:65536])]
    encoder.set_knob("link_run", 1)
    batch_ref = encoder.deflate_batch(datas, 10)
    try:
        for run in (2, 4, 16):
            encoder.set_knob("link_run", run)
            for method in (10, 8):
                for name, d in edge_inputs().items():
                    if len(d) >= 32768:
                        ta = oracle_tokens(d, method)
                        tb = encoder.lz77_tokens(d, method)
                        assert len(ta) == len(tb) and (ta == tb).all(), (run, method, name)
                for i, x in enumerate(streams):
                    rc, ref, crc = refs[(i, method)]
                    rc2, out, crc2 = gpu_deflate(encoder, x, method)
                    assert rc == rc2 and (rc != 0 or (out == ref and crc == crc2)), (run, method, i)
            assert encoder.deflate_batch(datas, 10) == batch_ref, run
            encoder.set_knob("shard_kib", 1024)
            rc, ref, crc = refs[(1, 10)]
            rc2, out, crc2 = gpu_deflate(encoder, streams[1], 10)
            encoder.set_knob("shard_kib", 1 << 20)
            assert rc == rc2 and out == ref and crc == crc2, run
    finally:
        encoder.set_knob("link_run", 0)
        encoder.set_knob("shard_kib", 1 << 20)


def test_stale_workspace_between_calls(encoder):
    """The tails tables of the link stage are not initialised (round 3): only occupied buckets are written, and their consumers
    (k_cross_links, k_match_demand) tell a tail from what an earlier call left behind by hashing the position an entry names.
    Different inputs one after the other on ONE context -- every bucket of the second finds the first one's entries where it has
    none of its own -- each the oracle's stream, for the methods with long and short chains."""
    rng = np.random.default_rng(77)
    a = silesia_mix(3 << 20)
    b = silesia_mix(3 << 20, seed=12345, class_mask=0x1B)
    cth = bytes((rng.integers(0, 3, 2 << 20) + 65).astype(np.uint8))
    d = bytes(rng.integers(0, 256, 100000, dtype=np.uint8)) + a[:1 << 20]
    for method in (10, 8):
        for x in (a, b, cth, a, d, b[: (1 << 20) + 5], a):
            rc, ref, crc = oracle_deflate(x, method)
            rc2, out, crc2 = gpu_deflate(encoder, x, method)
            assert rc == rc2 and (rc != 0 or (out == ref and crc == crc2)), (len(x), method)


def test_workspace_guesses_that_turn_out_too_small(encoder):
    """Round 6 sizes two arrays by what streams usually need instead of by the worst case: the atom arrays (room for 0.5 atoms per input byte, "atoms_pct";
    one per byte is the worst case) and the parse splice's token slots (128 per 512-byte chunk, "fix_stride"; 1152 is the worst case).  Both grow when a
    call needs more -- the atoms of the shards before are kept, the parse starts again --, and the stream is the oracle's either way: forced here with a
    1 % guess and slots of two tokens, on inputs in several shards, and without forcing on random bytes (one atom per byte)."""
    rng = np.random.default_rng(21)
    inputs = [silesia_mix((3 << 20) + 123, version=2), bytes(rng.integers(0, 256, 5 << 20, dtype=np.uint8)), silesia_mix(5 << 20, class_mask=1) + bytes(rng.integers(0, 256, 1 << 20, dtype=np.uint8))]
    refs = [oracle_deflate(d, 10) for d in inputs]
    try:
        for pct, stride, shard in ((50, 0, 1 << 20), (1, 2, 1 << 20), (1, 2, 1024), (3, 7, 2048)):
            encoder.set_knob("atoms_pct", pct); encoder.set_knob("fix_stride", stride); encoder.set_knob("shard_kib", shard)
            grown = [0, 0]
            for d, (rc, ref, crc) in zip(inputs, refs):
                rc2, out, crc2 = gpu_deflate(encoder, d, 10)
                assert rc == rc2 and (rc != 0 or (out == ref and crc == crc2)), (pct, stride, shard, len(d))
                t = dict(encoder.last_timing())
                grown[0] += t.get("#atoms_grown", 0); grown[1] += t.get("#fix_grown", 0)
            if pct == 1:
                assert grown[0] >= 1 and grown[1] >= 1, grown            # both fall-backs really ran
            else:
                assert grown[1] == 0 or stride, grown
        # ... and in a stream that goes through one context span after span (the atoms carried from span to span live in the arrays that grow)
        d = inputs[2] + inputs[0]
        encoder.set_knob("span_mib", 4); encoder.set_knob("shard_kib", 1 << 20)
        for method in (10, 7):
            rc, ref, crc = oracle_deflate(d, method)
            encoder.set_knob("atoms_pct", 1)
            rc2, out, crc2 = gpu_deflate(encoder, d, method)
            assert rc == rc2 and (rc != 0 or (out == ref and crc == crc2)), ("spans", method)
            assert dict(encoder.last_timing()).get("#atoms_grown", 0) >= 0
    finally:
        encoder.set_knob("atoms_pct", 50); encoder.set_knob("fix_stride", 0); encoder.set_knob("shard_kib", 1 << 20); encoder.set_knob("span_mib", 2048)


def test_cross_segment_walks_with_lists_that_overflow(encoder):
    """k_cross_dist (round 6) puts the level-4 walks its sweep leaves open on a list for a second pass, and the longest of those on a second list for a
    wave that reads the text backwards; a list that is full leaves the walk where it is.  With lists of 1, 7 and 300 entries ("cd_list_cap") nearly every
    open walk takes that path -- streams of every kind still the oracle's; the few-symbol inputs are where walks are long."""
    rng = np.random.default_rng(31)
    inputs = [silesia_mix((5 << 20) + 77, version=2), bytes((rng.integers(0, 3, 3 << 20) + 65).astype(np.uint8)), silesia_mix(3 << 20, class_mask=16, version=2) + bytes(100000) + silesia_mix(1 << 20, class_mask=2)]
    refs = [oracle_deflate(d, 10) for d in inputs]
    try:
        for cap in (1, 7, 300, 0):
            encoder.set_knob("cd_list_cap", cap)
            for d, (rc, ref, crc) in zip(inputs, refs):
                rc2, out, crc2 = gpu_deflate(encoder, d, 10)
                assert rc == rc2 and (rc != 0 or (out == ref and crc == crc2)), (cap, len(d))
    finally:
        encoder.set_knob("cd_list_cap", 0)
