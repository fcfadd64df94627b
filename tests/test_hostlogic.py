"""CPU tests of the product's host/device logic (zip-ada_amd/csrc/zada_logic.h compiled for the host
by tests/hostcheck -- a TEST library) against the oracle.  These are the per-lane routines the HIP
kernels run; the kernels themselves are covered by the -m gpu tests."""
import ctypes
import re
import os

import numpy as np
import pytest

from _common import ROOT, edge_inputs, hostcheck, oracle, oracle_tokens, silesia_mix


def test_llhc_lane_serial_equals_oracle():
    """llhc_serial (explicit-stack boundary package-merge + the reference's quicksort) ==
    oracle's literal restatement, ties included."""
    O, H = oracle(), hostcheck()
    rs = np.random.RandomState(5)
    for it in range(3000):
        n, mb = ((288, 15), (32, 15), (19, 7))[it % 3]
        kind = it % 7
        if kind == 0: f = rs.randint(0, 4, n)
        elif kind == 1: f = rs.randint(0, 50, n)
        elif kind == 2: f = (rs.pareto(1.0, n) * 10).astype(np.int64)
        elif kind == 3: f = rs.randint(0, 2, n) * rs.randint(1, 100000, n)
        elif kind == 4: f = np.where(rs.rand(n) < 0.1, rs.randint(1, 5, n), 0)
        elif kind == 5: f = (2 ** rs.randint(0, 17, n)) * (rs.rand(n) < 0.5)
        else: f = rs.randint(1, 3, n)
        f = np.minimum(f, 1 << 24)
        f64, f32 = f.astype(np.uint64), f.astype(np.uint32)
        a = np.zeros(n, dtype=np.int32)
        b = np.zeros(n, dtype=np.uint8)
        assert O.zo_llhc(f64.ctypes.data, n, mb, a.ctypes.data) == 0
        H.hc_llhc(f32.ctypes.data_as(ctypes.c_void_p), n, mb, b.ctypes.data_as(ctypes.c_void_p))
        assert (a == b).all(), (it, n, kind)


def test_symbol_tables_equal_rfc1951():
    H = hostcheck()
    lbase = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
    lext = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
    dbase = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
    dext = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
    for L in range(3, 259):
        s = 28 if L == 258 else max(i for i in range(28) if lbase[i] <= L)
        assert H.hc_len_symbol(L) == 257 + s and H.hc_len_extra_bits(L) == lext[s] and H.hc_len_extra_val(L) == L - lbase[s]
    for D in range(1, 32769):
        s = max(i for i in range(30) if dbase[i] <= D)
        assert H.hc_dist_symbol(D) == s and H.hc_dist_extra_bits(D) == dext[s] and H.hc_dist_extra_val(D) == D - dbase[s]


def test_demand_loop_model_with_lengths_to_beat():
    """Round 6's rules of the demand loop (zada_lz.hip, lz_shard) as a sequential model over the product's own chunk logic (zada_logic.h: run_parser's
    on_guess with the state's match length, parse_fix_chunk's capped sink): a first pass of `budget` chain steps leaves guesses; the parses that land on
    them demand them; a changed value flags its chunk -- from the second round on only if its length exceeds the smallest length to beat a parse left --;
    short lists are parsed with the exact look-up inside; a splice that overflows its small slot starts the parse again.  Whatever the budget, the list
    threshold, the slots, and with or without the lengths to beat: the oracle's tokens (lz77.adb:460-943 run sequentially).  And the rule is worth having:
    over the matrix it leaves changed values unflagged, and never parses more chunks again than the plain rule."""
    H = hostcheck()
    cases = {k: v for k, v in edge_inputs().items() if 3000 < len(v) <= 310000}
    rs = np.random.RandomState(5)
    cases["runs_and_noise"] = b"".join(bytes([rs.randint(97, 101)]) * int(rs.randint(1, 400)) if rs.rand() < 0.5 else bytes(rs.randint(97, 103, int(rs.randint(1, 300))).astype(np.uint8)) for _ in range(500))
    saved = respec = 0
    for name, d in cases.items():
        for method, level in ((10, 10), (8, 6)):
            a = oracle_tokens(d, method)
            got = {}
            for budget, use_beat, exact_max, fix_cap, chunk in ((1, 1, 0, 0xFFFFFFFF, 512), (1, 0, 0, 0xFFFFFFFF, 512), (4, 1, 8, 128, 512), (4, 0, 8, 128, 512),
                                                                (16, 1, 1 << 30, 2, 1024), (2, 1, 0, 3, 4096), (64, 1, 4, 128, 512)):
                t = np.zeros(len(d) + 8, dtype=np.uint32)
                st = np.zeros(6, dtype=np.uint64)
                k = H.hc_demand_loop_tokens(d, len(d), level, chunk, budget, use_beat, exact_max, fix_cap, t.ctypes.data, len(t), st.ctypes.data)
                assert k == len(a) and (t[:k] == a).all(), (name, method, budget, use_beat, exact_max, fix_cap, chunk, st.tolist())
                got[(budget, use_beat, exact_max, fix_cap, chunk)] = st.tolist()
            for budget, ex, cap, chunk in ((1, 0, 0xFFFFFFFF, 512), (4, 8, 128, 512)):
                with_beat, plain = got[(budget, 1, ex, cap, chunk)], got[(budget, 0, ex, cap, chunk)]
                assert with_beat[4] == plain[4] and plain[2] == 0
                assert with_beat[1] <= plain[1], (name, method, with_beat, plain)
                saved += with_beat[2]; respec += plain[1] - with_beat[1]
            assert got[(16, 1, 1 << 30, 2, 1024)][3] <= 1                      # (slots of two tokens: at most one restart, and only where a splice needs a third)
    assert saved > 0 and respec > 0, (saved, respec)


@pytest.mark.parametrize("method", (6, 8, 9, 10))
def test_chunked_speculative_parse_equals_sequential_reference(method):
    """The GPU's parse = per-chunk speculative parse + splice to a fixpoint (parse_spec_chunk /
    parse_fix_chunk) over all-position match tables.  Emulated sequentially here; must give the
    oracle's (= the reference's sequential) token stream, also on inputs that never resynchronise."""
    H = hostcheck()
    level = {6: 4, 8: 6, 9: 8, 10: 10}[method]
    cases = edge_inputs()
    for name, d in cases.items():
        if len(d) > 400000:
            continue
        a = oracle_tokens(d, method)
        for chunk in (4096, 1024):
            t = np.zeros(len(d) + 8, dtype=np.uint32)
            r = ctypes.c_int(0)
            k = H.hc_chunked_tokens(d, len(d), level, chunk, t.ctypes.data, len(t), ctypes.byref(r))
            assert k == len(a) and (t[:k] == a).all(), (name, chunk)


def test_c_abi_exports_every_declared_symbol():
    """The C-ABI library loads and exports every function include/zada.h declares (no compute)."""
    hdr = open(os.path.join(ROOT, "include", "zada.h")).read()
    names = set(re.findall(r"\b(zada_[a-z0-9_]+)\s*\(", hdr)) - {"zada_feedback_fn"}
    assert len(names) >= 12
    lib = ctypes.CDLL(os.path.join(ROOT, "zip-ada_amd", "libzada_hip.so"))
    for nme in sorted(names):
        assert hasattr(lib, nme), nme
    lib.zada_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.zada_version()


def test_product_never_references_the_oracle():
    """The product path must not import, link or execute anything under oracle/."""
    base = os.path.join(ROOT, "zip-ada_amd")
    for dp, _dn, fn in os.walk(base):
        for f in fn:
            if f.endswith((".hip", ".h", ".c", ".cpp", ".py", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "zada_oracle" not in txt and "libzada_oracle" not in txt and "zo_deflate" not in txt and "zo_bz" not in txt and "zo_lzma" not in txt, f


def test_zip64_promotion_of_the_container_writer():
    """Zip_64 in ZipCreate (zip-create.adb:161-179, 237-251, 682-752; zip-headers.adb:197-210, 534-579) without 4 GiB of
    data: entries made elsewhere with recorded sizes beyond 4 GiB, a pretended archive offset beyond 4 GiB, and 65 535
    entries.  Bytes == the oracle's Zip.Create restatement; Python's zipfile (independent reader) accepts what it can see."""
    import io
    import zipfile
    import zlib
    from _common import oracle_zip_compressed
    from _common import product
    za = product()
    small = zlib.compress(b"hello " * 100, 9)[2:-4]
    crc = zlib.crc32(b"hello " * 100)

    def both(entries, bias=0):
        zc = za.ZipCreate(None, 10, _offset_bias=bias)
        for name, payload, c, usize, zt in entries:
            zc.add_compressed(name, payload, c, usize, zt)
        got = zc.finish()
        assert got == oracle_zip_compressed(entries, bias)
        return got

    # (a) plain Zip_32 stays Zip_32
    got = both([("a.txt", small, crc, 600, 8), ("b\\c.txt", b"xyz", zlib.crc32(b"xyz"), 3, 0)])
    assert b"PK\x06\x06" not in got and zipfile.ZipFile(io.BytesIO(got)).read("a.txt") == b"hello " * 100
    # (b) an uncompressed size beyond 4 GiB: local extension (20 bytes), central extension (28), Zip64 end records
    big = 5 * 2 ** 30 + 123
    got = both([("a.txt", small, crc, 600, 8), ("big.bin", small, crc, big, 8), ("c.txt", small, crc, 600, 8)])
    zf = zipfile.ZipFile(io.BytesIO(got))
    assert [i.file_size for i in zf.infolist()] == [600, big, 600] and zf.read("c.txt") == b"hello " * 100
    assert got.count(b"PK\x06\x06") == 1 and got.count(b"PK\x06\x07") == 1
    # (c) Check_Size: a size just under 4 GiB promotes the ARCHIVE (end records) but needs no extension
    got = both([("edge.bin", small, crc, 2 ** 32 - 65644, 8)])
    assert b"PK\x06\x06" in got and got[28:30] == b"\x00\x00"
    got = both([("edge.bin", small, crc, 2 ** 32 - 65645, 8)])
    assert b"PK\x06\x06" not in got
    # (d) offsets beyond 4 GiB (pretended): the extension is decided by the offset alone
    got = both([("late.bin", small, crc, 600, 8), ("later.bin", b"", 0, 0, 0)], bias=2 ** 32 + 5)
    assert got[28:30] == b"\x14\x00"
    # (e) 65 534 entries stay Zip_32, 65 535 need Zip_64 (:682-687)
    for count, z64 in ((65534, False), (65535, True), (65540, True)):
        entries = [("e%05d" % i, b"", 0, 0, 0) for i in range(count)]
        got = both(entries)
        assert (b"PK\x06\x06" in got[-200:]) == z64
        assert len(zipfile.ZipFile(io.BytesIO(got)).infolist()) == count


def test_bench_starts_its_own_ranks_and_fails_loudly():
    """`python bench.py --gpus N` (no launcher) starts N ranks through torch.distributed.run before touching a GPU and relays
    their exit code; a launcher whose WORLD_SIZE disagrees with --gpus is refused.  (No GPU here: the ranks stop at "needs a
    GPU", which is the loud failure the product path promises.)"""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    # the parent counts the devices before it starts anything -- from sysfs, without loading torch or HIP; with every GPU hidden
    # from it the answer is 0 on any box
    import re
    hidden = dict(env, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0", "--mib", "1"], env=hidden, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and re.search(r"--gpus 2 but only 0 GPU\(s\) are visible", r.stderr), r.stderr[-2000:]
    src = open(bench).read()
    assert "import torch" not in src[src.index("def launch_ranks"):src.index("def main")], "the launching parent must not load torch / HIP"
    # BENCH_EMULATE=1 (every rank on GPU 0) skips that count: the two ranks are started and stop at "needs a GPU"
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0", "--mib", "1"], env=dict(hidden, BENCH_EMULATE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and r.stderr.count("needs a GPU") >= 2, r.stderr[-2000:]
    r = subprocess.run([sys.executable, bench, "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "--gpus 4 but the launcher started 2 ranks" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "at least 1" in r.stderr


def test_every_knob_is_documented_in_the_header():
    """zada_set_knob's names (zada_api.hip) all appear in include/zada.h, where an integrator looks for them."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    api = open(os.path.join(root, "zip-ada_amd", "csrc", "zada_api.hip")).read()
    hdr = open(os.path.join(root, "include", "zada.h")).read()
    names = set(re.findall(r'strcmp\(name, "([a-z0-9_]+)"\)', api))
    assert len(names) >= 15
    missing = sorted(n for n in names if '"%s"' % n not in hdr)
    assert not missing, missing


def test_return_codes_of_the_header_and_of_the_python_mirror_agree():
    """The negative codes of include/zada.h are distinct, and the one the Python mirror names (E_REFERENCE: an LZMA_3 entry on which the
    reference's own matcher leaves the format) has the header's value; INTEGRATION.md tells the shim what to do with it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "zada.h")).read()
    codes = {k: int(v) for k, v in re.findall(r"\b(ZADA_[A-Z_]+)\s*=\s*(-?\d+)", hdr)}
    assert codes["ZADA_OK"] == 0 and codes["ZADA_INEFFICIENT"] == 1 and codes["ZADA_ABORTED"] == 2
    neg = [v for k, v in codes.items() if k.startswith("ZADA_E_")]
    assert len(neg) == len(set(neg)) >= 6 and all(v < 0 for v in neg)
    src = open(os.path.join(root, "zip-ada_amd", "__init__.py")).read()
    assert int(re.search(r"^E_REFERENCE\s*=\s*(-?\d+)", src, re.M).group(1)) == codes["ZADA_E_REFERENCE"] == -6
    assert "ZADA_E_REFERENCE" in open(os.path.join(root, "INTEGRATION.md")).read()


def test_bench_reads_committed_records_and_carries_no_constants():
    """every figure on the bench line that the run did not measure itself comes from a committed file under profiles/ and is
    named with that file (and the commit it was taken at); config 4's shape is the record of the largest committed run on the round's corpus."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    c4 = b.config_4_log()
    assert c4 is not None and c4["workload"].endswith(b.CORPUS_NAME) and c4["equals_cpu_port"] is True and c4["source"].startswith("profiles/")
    assert c4["is_config_4_itself"] == ("1024 MiB" in c4["workload"])
    t, src = b.pmc_traffic(1 << 30, "k_match")
    assert isinstance(t, int) and t > (1 << 30) and src["file"].startswith("profiles/") and src["taken_at_commit"]
    assert b.pmc_traffic(1 << 20, "k_match") == (None, None)                  # (only valid for the workload the passes were taken on)
    pt = b.pipeline_traffic(1 << 30)
    assert isinstance(pt, int) and pt > t
    tr, tsrc = b.leg_traffic("k_bz_entropy")
    assert tr > 0 and "profiles/" in tsrc and "commit" in tsrc
    src_text = open(os.path.join(ROOT, "bench.py")).read()
    assert "788.6" not in src_text and "1.362" not in src_text                 # (round 4's hard-coded config-4 result)


def test_fused_cross_links_model():
    """The scheme of k_prev_links' fused cross links (round 5) in numpy, against the plain definition "distance to the nearest earlier
    position with the same key, if within MAX_DIST": per segment the links inside the segment from a stable sort; a bucket's first member carries
    its KEY in the link array, a bit map says which entries are keys; the previous segment's tails table is consulted in two rounds, round r taking
    the keys with bit 15 = r -- a head linked in round 0 holds a distance <= MAX_DIST < 2^15 (or 0) and is never mistaken for a key of round 1."""
    MAX_DIST, SEG = 32506, 32768
    rng = np.random.default_rng(9)
    for trial in range(6):
        nkeys = [40, 3000, 65536, 65536, 20000, 7][trial]
        keys = rng.integers(0, nkeys, 3 * SEG).astype(np.int64)
        if trial >= 2:
            keys = (keys * 40503 + 12345) % 65536                  # spread over all 16 bits, bit 15 included
        n = len(keys)
        # the plain definition
        want = np.zeros(n, np.int64)
        last = {}
        for p in range(n):
            q = last.get(int(keys[p]))
            if q is not None and p - q <= MAX_DIST:
                want[p] = p - q
            last[int(keys[p])] = p
        prev_tails = None
        for s in range(3):
            k = keys[s * SEG:(s + 1) * SEG]
            order = np.argsort(k, kind="stable")                    # the segment's sorted order: (key, position)
            P = np.zeros(SEG, np.int64)
            head = np.zeros(SEG, bool)
            ks, es = k[order], order
            first = np.ones(SEG, bool); first[1:] = ks[1:] != ks[:-1]
            lastm = np.ones(SEG, bool); lastm[:-1] = ks[1:] != ks[:-1]
            P[es[~first]] = (es[1:] - es[:-1])[~first[1:]]          # links inside the segment
            P[es[first]] = ks[first]; head[es[first]] = True        # heads: the key in the link's place
            tails = np.full(65536, 0xFFFF, np.int64); tails[ks[lastm]] = es[lastm]
            if prev_tails is None:
                P[head] = 0                                         # (a stream's first segment: nothing before it)
            else:
                for r in (0, 1):
                    for e in np.nonzero(head)[0]:
                        v = P[e]
                        if (v >> 15) != r:
                            continue                                # a key of the other round -- or, in round 1, a head that round 0 has linked
                        if r == 1:
                            assert v >= 32768
                        t = prev_tails[v]
                        d = e + SEG - t
                        P[e] = d if (t != 0xFFFF and d <= MAX_DIST) else 0
                        assert P[e] < 32768
            got = P.copy()
            got[got > MAX_DIST] = 0                                 # (inside a segment a link is at most 32 767; the match kernels bound it)
            w = want[s * SEG:(s + 1) * SEG]
            assert np.array_equal(got, w), (trial, s, int(np.argmax(got != w)))
            prev_tails = tails
