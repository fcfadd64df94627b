#!/usr/bin/env python3
"""bench.py -- Deflate encode throughput of the MI355X-native encoder (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json configs):
  N = 1   C2: Deflate_3 on 1 GiB of the synthetic "silesia_mix_v2" stream, one entry, input resident in HBM.
  N > 1   C3: Deflate_3 on ONE logical stream of N x 2 GiB (16 GiB at N = 8), cut into N ranges, one per GPU
          (SURVEY.md 8e primary mode).  The ranks exchange the sequential encoder's state at the range boundaries over
          RCCL / xGMI (zip-ada_amd/sharding.py: parser states and atom counts by all_gather, boundary atoms by all_gather,
          the block chooser's 352-byte state from rank to rank) and the compressed ranges are gathered onto rank 0 and
          OR-ed into one stream -- all inside the timed region.  The stream is, bit for bit, what one call on the whole
          input produces (tests/test_ranges.py).  "weak" scaling: per-GPU work is fixed for N >= 2.

One "step" = one pass of the whole hot path (CRC-32, LZ77 match finding + lazy parse, Taillaule block split, Huffman coding,
bit emission) over the workload.  value = input bytes of all ranks / max-over-ranks time, MB/s (10^6 bytes), with the input
already in HBM when the timed region starts; "host_path" on the same line is the drop-in entry point zada_deflate on host
buffers (PCIe both ways included), measured after the timed region.

Also on the JSON line:
  roofline     dominant kernel: algorithmic bytes per launch = N_in + N_out / its duration measured with HIP
               events on the encoder's own stream; "bound2": the resource that really bounds it, from the committed SQ
               counter passes (profiles/).
  cpu_baseline the oracle (single-threaded C port of the reference encoder, kind "port") on a bounded sample of the same
               stream on this box's host cores: one core, all cores (one stream per hardware thread), and libz tuned to
               the reference's IZ_10 row as a second anchor.
  checks       after the run: the stream inflates back to the input (independent inflater) and the first 64 MiB compressed
               alone equal the CPU port's stream byte for byte.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_SPEC_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
HBM_PEAK_GBS = HBM_SPEC_GBS    # the roofline's peak (every "frac" divides by it); the copy bandwidth measured on the box is carried beside it
HBM_MEASURED = None
SEED = 0x5A1E51A
CORPUS = 2                     # silesia_mix_v2 (segments seeded independently; v1's were shifted copies of one stream of draws -- csrc/silesia_mix.c)
CORPUS_NAME = "silesia_mix_v2"


def mix(za, nbytes, offset=0, version=None):
    return za.silesia_mix(nbytes, seed=SEED, offset=offset, version=CORPUS if version is None else version)


def host_cpus():
    """Hardware threads this process may use and, when the container has a CPU quota (cgroup cpu.max), the cores' worth of CPU
    time it actually gets -- the all-cores baselines scale with the latter, not with the thread count."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            quota = round(int(a) / int(b), 2)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = round(q / per, 2)
        except Exception:
            pass
    return cores, quota


def all_cores(fn, parts, one_core_mbs):
    """fn(part) on every part at once, one thread per part (the oracle is called through ctypes, which releases the GIL; the
    reference is single-threaded per stream).  Returns the all-cores object of a cpu_baseline."""
    from concurrent.futures import ThreadPoolExecutor
    cores, quota = host_cpus()
    nbytes = sum(len(p) for p in parts)
    t = time.perf_counter()
    c0 = time.process_time()
    with ThreadPoolExecutor(len(parts)) as ex:
        list(ex.map(fn, parts))
    dt = time.perf_counter() - t
    busy = (time.process_time() - c0) / dt if dt > 0 else 0.0
    v = nbytes / dt / 1e6
    return {"value": round(v, 2), "unit": "MB/s", "cores": cores, "threads": len(parts), "cpu_quota_cores": quota,
            "effective_parallelism": round(busy, 1), "speedup_over_one_core": round(v / one_core_mbs, 1) if one_core_mbs else None,
            "sample": "%d independent streams of %d KiB at once, one per thread, %.1f s (effective_parallelism = CPU seconds per wall second)" % (len(parts), len(parts[0]) >> 10, dt)}


def hbm_copy_gbs(torch, dev, mib=1024, reps=5):
    """Measured HBM bandwidth of this box: a device-to-device copy of `mib` MiB (read + write), best of `reps`, HIP events.
    SURVEY.md 8d: the roofline's peak is the lower of this and the 8 TB/s of the data sheet."""
    a = torch.empty(mib << 20, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    a.zero_(); b.copy_(a)
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); e1.synchronize()
        best = max(best, 2.0 * (mib << 20) / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return round(best, 1)


def _oracle():
    import subprocess
    so = os.path.join(ROOT, "oracle", "libzada_oracle.so")
    pin = os.path.join(ROOT, "oracle", "libzada_zlibpin.so")
    if not os.path.exists(so) or not os.path.exists(pin):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    O = ctypes.CDLL(so)
    O.zo_deflate.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64,
                             ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    P = ctypes.CDLL(pin)
    P.zp_zlib_tuned_size.restype = ctypes.c_int64
    P.zp_zlib_tuned_size.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return O, P


def _oracle_deflate(O, d):
    n = len(d)
    out = ctypes.create_string_buffer(n + 64)
    ol = ctypes.c_uint64(0)
    crc = ctypes.c_uint32(0xFFFFFFFF)
    rc = O.zo_deflate(d, n, 10, out, n + 64, ctypes.byref(ol), ctypes.byref(crc), None, None, None, None)
    assert rc == 0
    return out.raw[:ol.value]


def cpu_baseline(za, sample_mib):
    """The CPU restatement of zip-compress-deflate.adb (oracle/, kind "port") timed on the host cores of this box, on the
    first sample_mib MiB of the benchmark stream.  The oracle is used here only as the measured baseline and checker."""
    O, P = _oracle()
    n = sample_mib << 20
    d = mix(za, n).tobytes()
    t0 = time.perf_counter()
    ref = _oracle_deflate(O, d)
    dt = time.perf_counter() - t0
    # all cores: one independent stream per hardware thread (the reference is single-threaded per stream); every thread
    # takes its own 8 MiB of the stream
    cores, _ = host_cpus()
    per = 8 << 20
    parts = [mix(za, per, (i + 1) * (64 << 20)).tobytes() for i in range(cores)]
    ac = all_cores(lambda p: len(_oracle_deflate(O, p)), parts, n / dt / 1e6)
    # libz with the reference's IZ_10 tuple (lz77.adb:546): the same LZ77 decisions, zlib's own block splitting
    t2 = time.perf_counter()
    zt = P.zp_zlib_tuned_size(d, n, 34, 258, 258, 4096)
    dzt = time.perf_counter() - t2
    t3 = time.perf_counter()
    z9 = len(zlib.compress(d, 9)) - 6
    dz9 = time.perf_counter() - t3
    return {"value": round(n / dt / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": "first %d MiB of the same %s stream, Deflate_3, oracle/zada_oracle.c single thread, %.1f s" % (sample_mib, CORPUS_NAME, dt),
            "ratio": round(len(ref) / n, 4),
            "all_cores": ac,
            "zlib_tuned_34_258_258_4096": {"ratio": round(zt / n, 4), "MB/s": round(n / dzt / 1e6, 2)},
            "zlib9": {"ratio": round(z9 / n, 4), "MB/s": round(n / dz9 / 1e6, 2)}}, ref


def _latest_profile(name):
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for rnd in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        p = os.path.join(pdir, rnd, name)
        if os.path.exists(p):
            best = p
    return best


def pmc_traffic(n, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (separate FETCH_SIZE and
    WRITE_SIZE runs of this same command, profiles/<round>/pmc_fetch_write_by_kernel.json; counters are
    in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM).  Only valid for the 1 GiB workload
    the profile was taken on; null otherwise."""
    best = _latest_profile("pmc_fetch_write_by_kernel.json")
    if n != (1 << 30) or not best:
        return None, None
    d = json.load(open(best))
    try:
        f = d["FETCH_SIZE"]["zada::" + kernel]; w = d["WRITE_SIZE"]["zada::" + kernel]
        src = {"file": os.path.relpath(best, ROOT), "taken_at_commit": d.get("_commit"), "corpus": d.get("_corpus"),
               "note": "NOT measured in this run: rocprofv3 --pmc passes of this same command, committed under profiles/"}
        return int((2 * f["sum"] / f["dispatches"] + w["sum"] / w["dispatches"]) * 1024), src
    except Exception:
        return None, None


def pipeline_traffic(n):
    """HBM bytes of one whole step (all kernels) from the same committed PMC passes: sum over kernels of (2 x FETCH + WRITE), for a
    --steps 1 --warmup 0 run of the 1 GiB workload."""
    best = _latest_profile("pmc_fetch_write_by_kernel.json")
    if n != (1 << 30) or not best:
        return None
    try:
        d = json.load(open(best))
        return int((2 * sum(v["sum"] for v in d["FETCH_SIZE"].values()) + sum(v["sum"] for v in d["WRITE_SIZE"].values())) * 1024)
    except Exception:
        return None


def leg_traffic(key):
    """HBM bytes per launch / per run of a secondary leg's dominant kernel from the committed PMC passes of that leg's script
    (profiles/<round>/pmc_secondary_legs.json: {key: {"fetch_kb", "write_kb", "launches", "workload"}}, FETCH_SIZE doubled)."""
    best = _latest_profile("pmc_secondary_legs.json")
    if not best:
        return None, None
    try:
        d = json.load(open(best))[key]
        whole = json.load(open(best))
        return (int((2 * d["fetch_kb"] + d["write_kb"]) * 1024 / d["launches"]),
                d["workload"] + " (" + os.path.relpath(best, ROOT) + (", taken at commit %s" % whole["_commit"] if whole.get("_commit") else "") + "; not measured in this run)")
    except Exception:
        return None, None


def second_bound(kernel):
    """What really bounds the dominant kernel (VALU issue slots / LDS / waiting), from the committed SQ counter pass."""
    best = _latest_profile("sq_bounds_by_kernel.json")
    if not best:
        return None
    try:
        d = json.load(open(best))[kernel]
        d["source"] = os.path.relpath(best, ROOT) + " (committed SQ counter pass, not measured in this run)"
        return d
    except Exception:
        return None


def inflate_check(stream, n, crc_expected):
    """Independent inflater (zlib, raw): the stream decodes to n bytes with the input's CRC-32."""
    dec = zlib.decompressobj(-15)
    crc, total = 0, 0
    view = memoryview(stream)
    for off in range(0, len(view), 1 << 24):
        chunk = dec.decompress(view[off:off + (1 << 24)])
        crc = zlib.crc32(chunk, crc); total += len(chunk)
    tail = dec.flush()
    crc = zlib.crc32(tail, crc); total += len(tail)
    return total == n and crc == crc_expected


def bzip2_leg(za, enc, mib, with_cpu, with_checks):
    """Secondary measurement (BASELINE config 5's method): BZip2_3 of `mib` MiB of the same synthetic
    stream, input and output resident in HBM; the oracle on a 2 MiB sample beside it; the stream through libbz2."""
    import bz2
    import numpy as np
    import torch
    n = mib << 20
    host = mix(za, n)
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.zeros(n + 4096, dtype=torch.uint8, device="cuda")
    enc.bzip2_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, 14)
    torch.cuda.synchronize()
    runs = []
    for _ in range(3):                                   # three timed runs; the value is their median
        t0 = time.perf_counter()
        rc, ol, crc = enc.bzip2_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, 14)
        runs.append(time.perf_counter() - t0)
    dt = sorted(runs)[1]
    blocks = enc.bz2_last_blocks()
    # the phases one by one (one batch in flight at a time, "bz_pipeline" = 0: with two in flight the entropy stage runs on a worker's
    # stream and has no marks): one extra run, not part of the value
    enc.set_knob("bz_pipeline", 0)
    enc.bzip2_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, 14)
    enc.set_knob("bz_pipeline", 1)
    tim = {}
    for k, v in enc.last_timing():
        if not k.startswith("#") and k != "bz:end":
            tim[k] = round(tim.get(k, 0.0) + v, 1)
    out = {"metric": "BZip2_3 encode MB/s (stream bit-exact with the CPU restatement of the reference, Ada parity unpinned)", "value": round(n / dt / 1e6, 2),
           "unit": "MB/s", "workload": "%d MiB %s, one stream, input and output resident in HBM" % (mib, CORPUS_NAME), "ms": round(dt * 1e3, 1), "runs_ms": [round(r * 1e3, 1) for r in runs], "rc": rc,
           "compression_ratio": round(ol / n, 4), "blocks": len(blocks), "tactics_kept": [sum(1 for b in blocks if b[2] == t) for t in range(4)], "phase_ms": tim}
    # dominant KERNEL (by kernel time, profiles/<round>/bzip2_256mib_kernel_stats.csv): k_bz_entropy, the search for the tables and selectors --
    # the "bz:entropy" phase is its launches alone (the longest PHASE, bz:bwt, is some 400 launches of a dozen kernels, none of them as long)
    dom = "bz:entropy"
    ach = (n + ol) / (tim[dom] * 1e-3) / 1e9 if tim.get(dom) else 0.0
    traffic, tsrc = leg_traffic("k_bz_entropy") if mib == 256 else (None, None)
    out["phases_are_of"] = "one extra run with one batch in flight at a time (bz_pipeline = 0), not part of the value"
    out["roofline"] = {"bound": "hbm", "kernel": "k_bz_entropy", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6),
                       "traffic": traffic, "traffic_source": tsrc, "launch_ms": tim.get(dom),
                       "note": "algorithmic bytes = N_in + N_out over the kernel's launches of one run (HIP events of the context's stream); the kernel is bound by LDS round trips "
                               "(cost sums, package-merge) and by how many sub-blocks are in flight, not by HBM (DESIGN.md 9)"}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from _bzip2 import oracle_encode
        sample = host[:2 << 20].tobytes()
        t1 = time.perf_counter()
        ref, _ = oracle_encode(sample, 2)
        dtc = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(len(sample) / dtc / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "port",
                               "sample": "first 2 MiB of the workload, oracle/zada_oracle_bz2.c (its BWT is a prefix-doubling sort: faster than the reference's comparison sort)"}
        cores, _ = host_cpus()
        per = 1 << 20                                     # (more than one 900 k block each)
        parts = [host[(i * per) % (n - per):(i * per) % (n - per) + per].tobytes() for i in range(cores)]
        out["cpu_baseline"]["all_cores"] = all_cores(lambda p: len(oracle_encode(p, 2)[0]), parts, len(sample) / dtc / 1e6)
        if with_checks:
            _, g, _ = enc.bzip2(sample, 14)
            out["sample_stream_equals_cpu_port"] = bool(g == ref)
    if with_checks and rc == 0:
        stream = bytes(d_out[:ol].cpu().numpy())
        dec = bz2.BZ2Decompressor()
        c, tot, off = 0, 0, 0
        while off < len(stream):
            ch = dec.decompress(stream[off:off + (1 << 22)])
            off += 1 << 22
            c = zlib.crc32(ch, c)
            tot += len(ch)
        out["stream_decompresses_to_input_crc"] = bool(tot == n and dec.eof and c == zlib.crc32(host) and (crc ^ 0xFFFFFFFF) == c)
    return out


def config_4_log():
    """BASELINE config 4 (ONE LZMA_3 stream of 1 GiB) is a run of an hour and more and not part of this command: what tests/gpu_lzma_c4.py
    wrote when it was last run on this round's corpus -- the largest such run whose record is committed under profiles/ (the full size does
    not fit the pool's one-hour limit per GPU call at 0.25 MB/s; its size is in the record) -- or nothing."""
    import glob
    import re
    best, best_mib = None, 0
    for p in glob.glob(os.path.join(ROOT, "profiles", "*", "config4_lzma3_*mib_%s.json" % CORPUS_NAME)):
        m = re.search(r"config4_lzma3_(\d+)mib_", os.path.basename(p))
        if m and (int(m.group(1)), p) > (best_mib, best or ""):
            best, best_mib = p, int(m.group(1))
    if not best:
        return None
    try:
        d = json.load(open(best))
        d["source"] = os.path.relpath(best, ROOT) + (": tests/gpu_lzma_c4.py, ONE stream coded in TWO GPU calls on one MI355X each (the first stopped by its feedback, its state exported "
                                                      "and imported: the pool ends a call after an hour) -- NOT part of this run" if "first_call" in d else
                                                      ": tests/gpu_lzma_c4.py, one run on one MI355X -- NOT part of this run")
        d["is_config_4_itself"] = bool(best_mib == 1024)
        return d
    except Exception:
        return None


def lzma_leg(za, enc, entries, kib, with_cpu, with_checks, one_mib=4):
    """Secondary measurement (BASELINE config 4's method): LZMA_3 of `entries` Zip entries of `kib` KiB
    (slices of the same synthetic stream) through ONE launch of the coder -- a stream is a chain of dependent steps, so entries
    are what runs in parallel -- and ONE stream alone beside it (config 4's shape).  Host buffers in, host buffers out."""
    import lzma
    size = kib << 10
    host = mix(za, entries * size)
    datas = [host[i * size:(i + 1) * size].tobytes() for i in range(entries)]
    enc.lzma_batch(datas, 18)                            # warm-up at full size (the producer's and the coder's buffers are allocated here)
    runs = []
    for _ in range(3):                                   # three timed calls; the value is their median
        t0 = time.perf_counter()
        res = enc.lzma_batch(datas, 18)
        runs.append(time.perf_counter() - t0)
    dt = sorted(runs)[1]
    tim = {k: round(v, 1) for k, v in enc.last_timing() if not k.startswith("#")}
    out_bytes = sum(len(z) for _, z, _ in res)
    out = {"metric": "LZMA_3 encode MB/s over a batch of Zip entries (payloads bit-exact with the CPU restatement of the reference, Ada parity unpinned)",
           "value": round(entries * size / dt / 1e6, 3), "unit": "MB/s", "workload": "%d entries of %d KiB, %s, one launch" % (entries, kib, CORPUS_NAME),
           "ms": round(dt * 1e3, 1), "runs_ms": [round(r * 1e3, 1) for r in runs], "compression_ratio": round(out_bytes / (entries * size), 4), "phase_ms": tim}
    kms = tim.get("lzma:end", dt * 1e3)
    ach = (entries * size + out_bytes) / (kms * 1e-3) / 1e9
    traffic, tsrc = leg_traffic("k_lzma_encode") if (entries, kib) == (4096, 16) else (None, None)
    out["roofline"] = {"bound": "hbm", "kernel": "k_lzma_encode", "achieved": round(ach, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 7),
                       "traffic": traffic, "traffic_source": tsrc, "launch_ms": kms, "producer_ms": tim.get("lzma:bt4"),
                       "note": "algorithmic bytes = N_in + N_out over the coder's launch (lzma:end; the BT4 match producer's kernels run before it: lzma:bt4); the bound is neither HBM nor MFMA "
                               "but the latency of one dependent instruction stream per entry (adaptive probabilities), 2 048 entries in flight (DESIGN.md 10)"}
    # BASELINE config 4's shape: ONE stream, `one_mib` MiB of the stream itself (not a repeated slice)
    one = host[:one_mib << 20].tobytes() if (one_mib << 20) <= len(host) else mix(za, one_mib << 20).tobytes()
    t1 = time.perf_counter()
    rc1, z1, _ = enc.lzma(one, 18)
    d1 = time.perf_counter() - t1
    tim1 = {k: round(v, 1) for k, v in enc.last_timing() if not k.startswith("#")}
    cnt1 = {k: v for k, v in enc.last_timing() if k.startswith("#")}
    out["one_stream"] = {"value": round(len(one) / d1 / 1e6, 4), "unit": "MB/s", "bytes": len(one), "compression_ratio": round(len(z1) / len(one), 4), "phase_ms": tim1,
                         "launches": int(cnt1.get("#lzma_launches", 0)), "producer": "in segments of 2**20 positions on a second stream, beside the coder (knob lzma_segment)",
                         "coder": "one workgroup of four waves: the chain's and three helpers for its forks (knob lzma_waves)",
                         "seconds_for_config_4_extrapolated": round((1 << 30) / (len(one) / d1), 0),
                         "note": "config 4 is ONE 1 GiB stream: it runs at this rate -- the match sets come from the producer's parallel kernels, segment k + 1 while the coder -- one "
                                 "wave walking the chain of adaptive probabilities, the independent simulations of a step on teams of its lanes -- codes segment k"}
    c4 = config_4_log()
    if c4:
        out["one_stream"]["config_4_shape_measured"] = c4
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from _lzmah import oracle_lzma
        k = min(entries, max(1, (2 << 20) // size))
        t2 = time.perf_counter()
        ref = [oracle_lzma(d, 18) for d in datas[:k]]
        dtc = time.perf_counter() - t2
        out["cpu_baseline"] = {"value": round(k * size / dtc / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "port",
                               "sample": "the first %d entries, oracle/zada_oracle_lzma.c" % k}
        cores, _ = host_cpus()
        per = max(1, (512 << 10) // size)                 # entries per thread: 512 KiB of them
        groups = [b"".join(datas[(i * per + j) % entries] for j in range(per)) for i in range(cores)]
        out["cpu_baseline"]["all_cores"] = all_cores(lambda g: [oracle_lzma(g[o:o + size], 18) for o in range(0, len(g), size)] and len(g), groups, k * size / dtc / 1e6)
        # what an unthrottled host of this box would do: the one-core rate times its hardware threads (the quota above is the container's)
        ht = os.cpu_count() or cores
        one_core = k * size / dtc / 1e6
        out["cpu_baseline"]["extrapolated_all_hardware_threads"] = {"value": round(one_core * ht, 1), "unit": "MB/s", "threads": ht,
                                                                    "gpu_batch_over_it": round(out["value"] / (one_core * ht), 2) if one_core else None,
                                                                    "note": "one-core rate x os.cpu_count(): an upper bound for the host (no memory or SMT contention counted)"}
        if with_checks:
            t3 = time.perf_counter()
            ref1 = oracle_lzma(one[:1 << 20], 18)
            c1 = (1 << 20) / (time.perf_counter() - t3) / 1e6
            out["one_stream"]["cpu_one_core"] = {"value": round(c1, 3), "unit": "MB/s", "sample": "the first MiB of the same stream, oracle/zada_oracle_lzma.c, one thread"}
            out["one_stream"]["gpu_over_cpu_one_core"] = round(out["one_stream"]["value"] / c1, 3) if c1 else None
            out["one_stream"]["first_MiB_alone_equals_cpu_port"] = bool(enc.lzma(one[:1 << 20], 18) == ref1)
        if with_checks:
            out["sample_payloads_equal_cpu_port"] = bool(all(res[i] == ref[i] for i in range(k)))
    if with_checks:
        ok = True
        for d, (rc, z, crc) in zip(datas, res):
            ds = int.from_bytes(z[5:9], "little")
            dec = lzma.LZMADecompressor(format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA1, "dict_size": ds, "lc": 3, "lp": 0, "pb": 2}])
            ok = ok and dec.decompress(z[9:]) == d and (crc ^ 0xFFFFFFFF) == zlib.crc32(d)
        out["payloads_decode_to_input_crc"] = bool(ok)
    return out


def bzip2_main(args, za, sharding, enc, torch, dist, rank, world, dev, emulate):
    """BASELINE config 5: ONE BZip2_3 stream of world x mib MiB, the blocks sharded over the GPUs (sharding.bzip2_stream_rank):
    same contract as the Deflate line (W warm-up steps, K timed steps between barriers, max over ranks, one JSON line)."""
    import bz2
    mib = args.mib or 1024
    n = mib << 20
    total = n * world
    ranges = sharding.bzip2_ranges(total, world)
    active = rank < len(ranges)
    off, blen = sharding.bzip2_window(total, *ranges[rank]) if active else (0, 1)
    host = mix(za, blen, off)
    d_buf = torch.from_numpy(host).to(dev)

    class Solo:                                       # one GPU: the same protocol without a process group
        rank, world = 0, 1

        def all_gather_obj(self, obj):
            return [obj]
    comm = sharding.TorchComm(torch.device("cpu") if emulate else dev) if world > 1 else Solo()
    state = {}

    def step():
        res = sharding.bzip2_stream_rank(enc, comm, total, ranges, d_buf.data_ptr(), 14, lambda k: torch.zeros(k, dtype=torch.uint8, device=dev))
        if world > 1:
            payload = res["payload"] if res["payload"] is not None else torch.zeros(1, dtype=torch.uint8, device=dev)
            if emulate:
                torch.cuda.synchronize()
                payload = payload.cpu()
            got = sharding.gather_stream(payload, res["spans"], res["total_bits"], dst=0)     # every range straight to its byte offset on rank 0
            if got is not None:
                state["stream"] = got
        else:
            state["stream"] = res["payload"][:res["nbytes"]]
        state["res"] = res

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if emulate else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    res = state["res"]
    nbytes = (res["total_bits"] + 7) // 8
    out = {"metric": "BZip2_3 encode MB/s (one stream, blocks sharded over the GPUs; bit-exact with the CPU restatement of the reference, Ada parity unpinned)",
           "value": round(total * args.steps / dt / 1e6, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "C5: BZip2_3, one stream of %d x %d MiB synthetic %s, %d block ranges, input resident in HBM" % (world, mib, CORPUS_NAME, len(ranges)),
                      "bytes_per_gpu": n, "stream_bytes": total, "compression_ratio": round(nbytes / total, 4), "blocks_rank0": len(res["blocks"])}}
    if not args.no_checks and total <= (3 << 30):
        stream = bytes(state["stream"].cpu().numpy())
        dec = bz2.BZ2Decompressor()
        c, tot, o = 0, 0, 0
        while o < len(stream):
            ch = dec.decompress(stream[o:o + (1 << 22)])
            o += 1 << 22
            if tot < len(host):
                c = zlib.crc32(ch[:max(0, len(host) - tot)], c)
            tot += len(ch)
        out["checks"] = {"stream_decompresses": bool(tot == total and dec.eof), "rank0_window_crc": bool(c == zlib.crc32(host[:min(len(host), total)]))}
    print(json.dumps(out))


def visible_gpus():
    """GPUs this process' children would see, counted WITHOUT loading torch or the HIP runtime: the KFD topology nodes that have
    SIMDs (CPUs have none), cut down by a *_VISIBLE_DEVICES list when one is set."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(f):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except OSError:
            pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n):
    """Starts `python -m torch.distributed.run --nproc-per-node n bench.py <the same arguments>` as a child process, relays its
    output (rank 0 prints the JSON line) and returns its exit code."""
    import socket
    import subprocess
    if os.environ.get("BENCH_EMULATE") != "1":
        have = visible_gpus()                         # (read from sysfs: this process never loads the HIP runtime)
        if have < n:
            print("bench.py: --gpus %d but only %d GPU(s) are visible" % (n, have), file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mib", type=int, default=0, help="input MiB per GPU (default: 1024 at one GPU = BASELINE C2, 2048 otherwise = C3 at 8 GPUs)")
    ap.add_argument("--cpu-sample-mib", type=int, default=128, help="MiB of the workload the one-core CPU baseline (the oracle) is timed on: about 12 s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-checks", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--no-v1", action="store_true", help="skip the extra Deflate measurement on silesia_mix_v1")
    ap.add_argument("--no-copy-probe", action="store_true", help="skip the device-to-device copy that measures the box's HBM bandwidth (the PMC passes: their counters then hold the step alone)")
    ap.add_argument("--method", choices=("deflate", "bzip2"), default="deflate", help="bzip2: BASELINE config 5 -- ONE BZip2_3 stream of N x --mib (default 1024) MiB over N GPUs")
    ap.add_argument("--bzip2-mib", type=int, default=256, help="input MiB of the secondary BZip2_3 measurement at one GPU (0 = skip)")
    ap.add_argument("--lzma-entries", type=int, default=4096, help="entries of the secondary LZMA_3 batch measurement at one GPU (0 = skip)")
    ap.add_argument("--lzma-kib", type=int, default=16, help="KiB per entry of the LZMA_3 batch")
    ap.add_argument("--lzma-one-mib", type=int, default=4, help="MiB of the ONE LZMA_3 stream measured beside the batch (BASELINE config 4's shape)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process only starts the N ranks (one per GPU) and waits for them.  It never
        # touches the GPU itself, so nothing that has initialised HIP is ever replaced or forked.
        raise SystemExit(launch_ranks(args.gpus))
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    # BENCH_EMULATE=1: every rank on GPU 0 over gloo -- exercises the N > 1 code path on a one-GPU box (not a measurement)
    emulate = os.environ.get("BENCH_EMULATE") == "1"
    if not emulate and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible" % (world, torch.cuda.device_count()))
    if emulate:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emulate:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the encoder has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    global HBM_PEAK_GBS, HBM_MEASURED
    HBM_MEASURED = None if args.no_copy_probe else hbm_copy_gbs(torch, dev)
    za = importlib.import_module("zip-ada_amd")
    sharding = importlib.import_module("zip-ada_amd.sharding")
    enc = za.Encoder(local_rank)
    if args.method == "bzip2":
        bzip2_main(args, za, sharding, enc, torch, dist, rank, world, dev, emulate)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    mib = args.mib or (1024 if world == 1 else 2048)
    n = mib << 20                                     # bytes per GPU
    total = n * world                                 # the stream
    phase_ms = {}

    def add_timing():
        for k, v in enc.last_timing():
            phase_ms[k] = phase_ms.get(k, 0.0) + v    # (names starting with '#' are counters, e.g. rounds of the demand loop)

    if world == 1:
        host = mix(za, n)
        d_in = torch.from_numpy(host).to(dev)
        d_out = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        last = {}

        def step():
            rc, out_len, crc = enc.deflate_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, za.Method.Deflate_3)
            last.update(rc=rc, out_len=out_len, crc=crc)

        def drain():
            pass
    else:
        # this rank's window of the ONE stream: range [rank * n, (rank + 1) * n) with 32 KiB before and 1 MiB behind it
        ranges = sharding.stream_ranges(total, world)
        lo, ln = ranges[rank]
        first, pre, post = sharding.range_window(total, lo, ln)
        host = mix(za, pre + ln + post, first)
        d_in = torch.from_numpy(host).to(dev)
        comm = sharding.TorchComm(torch.device("cpu") if emulate else dev)
        torch.cuda.synchronize()
        last = {}
        state = {"pending": None, "stream": None}

        xt = {}                                        # wall-clock seconds this rank spent in the exchange steps (all timed steps)

        def finish_pending():
            p = state["pending"]
            if p is None:
                return
            h, res = p
            tg = time.perf_counter()
            got = h.finish() if h is not None else None
            if got is not None:                       # rank 0: the stream, every range received at its byte offset, shared edge bytes OR-ed
                state["stream"] = got
            state["pending"] = None
            xt["gather_stitch"] = xt.get("gather_stitch", 0.0) + time.perf_counter() - tg

        def step():
            res = sharding.deflate_stream_rank(enc, comm, torch, total, ranges, d_in.data_ptr(), za.Method.Deflate_3,
                                               lambda k: torch.empty(k, dtype=torch.int32, device=dev),
                                               lambda k: torch.empty(k, dtype=torch.uint8, device=dev))
            payload = res["payload"] if res["payload"] is not None else torch.empty(1, dtype=torch.uint8, device=dev)
            if emulate:
                torch.cuda.synchronize()
                payload = payload.cpu()
            # the payload of this step travels to rank 0 while the next step is compressed
            for k, v in res.get("exchange_s", {}).items():
                xt[k] = xt.get(k, 0.0) + v
            h = None if res["inefficient"] else sharding.gather_stream_begin(payload, res["spans"], res["total_bits"], dst=0)
            finish_pending()
            state["pending"] = (h, res)
            last.update(rc=1 if res["inefficient"] else 0, out_len=(res["total_bits"] + 7) // 8,
                        crc=sharding.stream_crc(enc.crc32_combine, res["infos"]))

        def drain():
            finish_pending()

    for _ in range(args.warmup):
        step()
    drain()
    if world > 1:
        xt.clear()                                        # (the warm-up's exchange times are not the timed region's)
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        add_timing()
    drain()                                               # every payload has arrived on rank 0 and is stitched, inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    multi = None
    if world > 1:
        # what the N > 1 line says about itself: the backend that really carried the exchange, the ranks the process group saw, every rank's
        # own time for the K steps and where it spent its exchange time (so that a scaling loss can be attributed)
        keys = ["all_gather_state", "boundary_atoms", "carry_chain", "spans", "gather_stitch"]
        mine = torch.tensor([dt] + [xt.get(k, 0.0) for k in keys], dtype=torch.float64, device=torch.device("cpu") if emulate else dev)
        allr = torch.empty(world * mine.numel(), dtype=torch.float64, device=mine.device)
        dist.all_gather_into_tensor(allr, mine)                # (one fixed-size tensor collective, as in the timed region: no object collectives anywhere)
        allr = allr.cpu().view(world, -1).tolist()
        every = [{"s": v[0], "exchange_s": dict(zip(keys, v[1:]))} for v in allr]
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cpu") if emulate else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        multi = {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(),
                 "rank_ms_per_step": {"min": round(min(e["s"] for e in every) * 1e3 / args.steps, 3), "max": round(max(e["s"] for e in every) * 1e3 / args.steps, 3)},
                 "exchange_ms_per_step": {k: {"rank0": round(every[0]["exchange_s"].get(k, 0.0) * 1e3 / args.steps, 3),
                                              "max": round(max(e["exchange_s"].get(k, 0.0) for e in every) * 1e3 / args.steps, 3)} for k in keys},
                 "exchange_note": "wall clock a rank spends inside each exchange step of sharding.deflate_stream_rank (waiting for its neighbours included): "
                                  "all_gather_state = parser states and atom counts, one 64-byte-per-rank tensor all_gather (a), boundary_atoms = edge atoms by all_gather (c), carry_chain = the 352-byte chooser state "
                                  "from rank to rank, its receive posted before the range's analysis (d), spans = bit positions, one 24-byte-per-rank tensor all_gather (d), gather_stitch = payloads to rank 0, each received at its byte offset in the stream, shared edge bytes OR-ed (e)"}

    if rank == 0:
        rc, out_len, crc = last["rc"], last["out_len"], last["crc"]
        ms_per_step = dt * 1e3 / args.steps
        value = total * args.steps / dt / 1e6
        # dominant kernel = the single-kernel phase with the largest time (each of these phases is ONE launch per shard)
        kernels = {"prev_links": "k_prev_links", "match": "k_match", "window_descr": "k_window_descr", "block_analyze": "k_block_analyze"}
        dom = max(kernels, key=lambda k: phase_ms.get(k, 0.0))
        t_dom = phase_ms.get(dom, 0.0) / args.steps * 1e-3
        alg_bytes = n + out_len / world                  # N_in + N_out of what this GPU's launches process
        achieved = alg_bytes / t_dom / 1e9 if t_dom > 0 else 0.0
        traffic, traffic_src = pmc_traffic(n, kernels[dom]) if world == 1 else (None, None)
        ptraffic = pipeline_traffic(n) if world == 1 else None
        if world == 1:
            wl = "C2: Deflate_3, %d MiB synthetic %s, one entry on one GPU, input resident in HBM" % (mib, CORPUS_NAME)
        else:
            wl = ("C3: Deflate_3, ONE logical stream of %d MiB synthetic %s cut into %d ranges of %d MiB, one per GPU; "
                  "boundary state over RCCL, payloads gathered and stitched on rank 0" % (total >> 20, CORPUS_NAME, world, mib))
        res = {
            "metric": "Deflate encode MB/s (Deflate_3; stream bit-exact with the CPU restatement of the reference, Ada parity unpinned)",
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "backend": multi["backend"] if multi else "none (one process, no process group)", "ranks_seen": multi["ranks_seen"] if multi else 1,
            "config": {"workload": wl, "bytes_per_gpu": n, "stream_bytes": total, "compression_ratio": round(out_len / total, 4), "rc": rc,
                       "value_is": "device-resident input (host buffers: see host_path)",
                       "phase_ms_per_step": {k: round(v / args.steps, 3) for k, v in phase_ms.items()}},
            "value_kind": "device_resident: input and output in HBM when the clock starts (the measurement contract's `value`); value_host_buffers on this line is "
                          "SURVEY 8d's end-to-end figure through zada_deflate on pageable host buffers, H2D and D2H included",
            "roofline": {"bound": "hbm", "kernel": kernels[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "peak_measured_copy": HBM_MEASURED, "frac_of_measured_copy": round(achieved / HBM_MEASURED, 6) if HBM_MEASURED else None,
                         "launch_ms": round(t_dom * 1e3, 3), "traffic": traffic, "traffic_source": traffic_src,
                         "bound2": second_bound(kernels[dom]),
                         "note": "algorithmic bytes = N_in + N_out per launch / the kernel's duration (HIP events on the encoder's stream, this run); peak = the data sheet's 8 TB/s (a 1 GiB device-to-device "
                                 "copy measured on this box, read + write, is peak_measured_copy); the LZ kernels are latency / issue bound (radix sort of positions, chain walk), not HBM bound"},
            "roofline_pipeline": {"bound": "hbm", "achieved": round(alg_bytes / (ms_per_step * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": ptraffic,
                                  "traffic_over_algorithmic": round(ptraffic / alg_bytes, 1) if ptraffic else None,
                                  "note": "the whole step priced like one kernel: (N_in + N_out) / ms_per_step / peak; traffic = HBM bytes of all kernels of one step from the same committed PMC passes as roofline.traffic"},
        }
        if multi:
            res["rank_ms_per_step"] = multi["rank_ms_per_step"]
            res["exchange_ms_per_step"] = multi["exchange_ms_per_step"]
            res["exchange_note"] = multi["exchange_note"]
        stream = b""
        if world == 1 and args.no_host_path:
            args.no_checks = True
        elif world == 1:
            # the drop-in entry point on host buffers (PCIe both ways), same input
            import numpy as np
            hout = np.empty(n + 64, dtype=np.uint8)
            hout[:] = 0                                   # (pages touched before the clock starts)
            enc.deflate_into(host, hout, za.Method.Deflate_3)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                _, ol_h, crc_h = enc.deflate_into(host, hout, za.Method.Deflate_3)
            dth = (time.perf_counter() - t1) / args.steps
            res["host_path"] = {"value": round(n / dth / 1e6, 2), "unit": "MB/s", "ms_per_step": round(dth * 1e3, 3), "steps": args.steps,
                                "entry": "zada_deflate (pageable host buffers in and out, PCIe both ways included), the same %d steps" % args.steps}
            res["value_host_buffers"] = res["host_path"]["value"]
            stream = hout[:ol_h].tobytes()
        else:
            stream = bytes(state["stream"].cpu().numpy()) if state["stream"] is not None else b""
        ref = None
        if not args.no_cpu_baseline:
            cb, ref = cpu_baseline(za, args.cpu_sample_mib)
            res["cpu_baseline"] = cb
        if not args.no_checks and rc == 0:
            checks = {}
            if world == 1:
                checks["stream_inflates_to_input_crc"] = bool(inflate_check(stream, total, zlib.crc32(host)) and (crc ^ 0xFFFFFFFF) == zlib.crc32(host))
            else:
                # Rank 0 holds range 0 of the input only.  The stream is inflated (independent inflater) through range 0, across
                # the first joint and 64 MiB into range 1 -- all of it when it is at most 3 GiB, then also against the CRC-32
                # the ranks computed and combined; range 0 of the decoded data must be rank 0's input.
                want = total if total <= (3 << 30) else n + (64 << 20)
                dec = zlib.decompressobj(-15)
                c_all, c0, tot = 0, 0, 0
                view = memoryview(stream)
                off = 0
                while tot < want and off < len(view):
                    ch = dec.decompress(view[off:off + (1 << 22)], want - tot)
                    off += 1 << 22                          # (what the call leaves unconsumed is drained by the inner loop)
                    while True:
                        c_all = zlib.crc32(ch, c_all)
                        c0 = zlib.crc32(ch[:max(0, min(len(ch), n - tot))], c0)
                        tot += len(ch)
                        if not dec.unconsumed_tail or tot >= want:
                            break
                        ch = dec.decompress(dec.unconsumed_tail, want - tot)
                ok = tot == want and c0 == zlib.crc32(host[pre:pre + n])
                if want == total:
                    ok = ok and dec.eof and c_all == (crc ^ 0xFFFFFFFF)
                checks["stream_inflates_to_input_crc"] = bool(ok)
                checks["inflated_bytes"] = int(tot)
            if ref is not None:
                head = mix(za, args.cpu_sample_mib << 20).tobytes()
                g, _ = enc.deflate(head, za.Method.Deflate_3)
                checks["sample_stream_equals_cpu_port"] = bool(g == ref)
            res["checks"] = checks
        if world == 1 and not args.no_v1:
            # round 5: the same workload on silesia_mix_v1 (the stream of rounds 1-4) beside it, so that the change of corpus is visible
            # on one line; Deflate's 32 KiB window cannot see what was wrong with v1
            h1 = mix(za, n, version=1)
            d_in.copy_(torch.from_numpy(h1).to(dev))
            rc1, ol1, _ = enc.deflate_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, za.Method.Deflate_3)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                enc.deflate_device(d_in.data_ptr(), n, d_out.data_ptr(), n + 4096, za.Method.Deflate_3)
            torch.cuda.synchronize()
            d1 = (time.perf_counter() - t1) / args.steps
            res["deflate_on_silesia_mix_v1"] = {"value": round(n / d1 / 1e6, 2), "unit": "MB/s", "ms_per_step": round(d1 * 1e3, 3), "steps": args.steps, "rc": rc1,
                                                "compression_ratio": round(ol1 / n, 4), "note": "the workload of rounds 1-4, device-resident, same steps"}
            del h1
        if world == 1 and (args.bzip2_mib > 0 or args.lzma_entries > 0):
            # the secondary legs get a context of their own: the Deflate context's workspace (80 GiB for the 1 GiB entry) goes back first
            del d_in, d_out
            enc.close()
            torch.cuda.empty_cache()
            enc = za.Encoder(local_rank)
        if world == 1 and args.bzip2_mib > 0:
            res["bzip2"] = bzip2_leg(za, enc, args.bzip2_mib, not args.no_cpu_baseline, not args.no_checks)
        if world == 1 and args.lzma_entries > 0:
            res["lzma"] = lzma_leg(za, enc, args.lzma_entries, args.lzma_kib, not args.no_cpu_baseline, not args.no_checks, args.lzma_one_mib)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
