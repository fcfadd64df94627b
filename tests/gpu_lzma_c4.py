"""GPU script: BASELINE config 4's shape -- ONE LZMA_3 stream of C4_MIB MiB (default 1024 = config 4 itself) of the benchmark stream through
zada_lzma (bounded launches, feedback), decoded by liblzma and compared with the input's CRC; C4_ORACLE=1 also codes it with the CPU port and
compares the bytes (1 GiB: several minutes on one core)."""
import json, lzma, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import product
Z = product(); enc = Z.Encoder(0)
mib = int(os.environ.get("C4_MIB", "1024"))
n = mib << 20
ver = int(os.environ.get("C4_CORPUS", "2"))          # silesia_mix_v2 since round 5 (v1's segments were shifted copies: the stream folded to 2.7 %)
d = Z.silesia_mix(n, seed=0x5A1E51A, version=ver).tobytes()
seen = []
t0 = time.time()
def fb(pct):
    if not seen or pct != seen[-1]:
        seen.append(pct)
        if pct % 10 == 0:
            print("  %3d %%  %.0f s" % (pct, time.time() - t0), flush=True)
    return False
# C4_STOP_PCT=p: the stream is stopped between two launches once p % are coded, the coder's state and the stream bytes so far go to
#   gpurun_out/c4_state.bin / c4_head.sha256 (the state is small; the bytes are hashed), and the run ends: the pool ends a GPU call after an hour, config 4 takes 70 minutes.
# C4_RESUME=path: the state is imported and the same stream goes on from it in this process (tests/ckpt/ is where a state travels to the next call).
# Either way C4_ORACLE=1 codes the whole stream with the CPU port on a host thread meanwhile and compares the bytes this call produced with its slice.
import hashlib, threading
stop_pct = int(os.environ.get("C4_STOP_PCT", "0"))
resume = os.environ.get("C4_RESUME", "")
outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(outdir, exist_ok=True)
oracle_box = {}
if os.environ.get("C4_ORACLE") == "1" and (stop_pct or resume):
    from _lzmah import oracle_lzma
    def run_oracle():
        t1 = time.time(); oracle_box["res"] = oracle_lzma(d, 18); oracle_box["s"] = time.time() - t1
    th = threading.Thread(target=run_oracle); th.start()
head_len = 0
if resume:
    st = open(resume, "rb").read()
    meta = json.load(open(resume + ".json"))
    head_len = meta["stream_bytes"]
    enc.lzma_import_state(st)
    print("resuming at position %d (%.1f %%), %d stream bytes written by the call before" % (meta["positions"], 100.0 * meta["positions"] / n, head_len), flush=True)
stopped = False
try:
    rc, z, crc = enc.lzma(d, 18, feedback=(lambda pct: fb(pct) or (stop_pct and pct >= stop_pct)) if stop_pct else fb)
except Z.UserAbort:
    stopped = True
dt = time.time() - t0
if stopped:
    state, head, pos = enc.lzma_export_state(n + 4096)
    open(os.path.join(outdir, "c4_state.bin"), "wb").write(state)
    rec = {"workload": "ONE LZMA_3 stream of %d MiB silesia_mix_v%d, first call: stopped by its feedback at %d %%" % (mib, ver, stop_pct), "seconds": round(dt, 1), "positions": pos,
           "stream_bytes": len(head), "stream_sha256": hashlib.sha256(head).hexdigest(), "MB/s": round(pos / dt / 1e6, 3)}
    print("stopped at position %d after %.1f s = %.3f MB/s; %d stream bytes, state of %d bytes" % (pos, dt, pos / dt / 1e6, len(head), len(state)), flush=True)
    if os.environ.get("C4_ORACLE") == "1":
        th.join()
        o = oracle_box["res"]
        rec.update(cpu_port_one_core_seconds=round(oracle_box["s"], 1), head_equals_cpu_port=bool(o[1][:len(head)] == head))
        print("CPU port (one core, beside the GPU): %.1f s; the first %d stream bytes equal: %s" % (oracle_box["s"], len(head), rec["head_equals_cpu_port"]), flush=True)
    json.dump(rec, open(os.path.join(outdir, "c4_state.bin.json"), "w"), indent=1)
    sys.exit(0)
if resume:
    tail = z[head_len:]
    rec = {"workload": "ONE LZMA_3 stream of %d MiB silesia_mix_v%d" % (mib, ver), "calls": "two GPU calls: this record is the second call's, from the state the first exported (first_call)",
           "seconds": round(dt, 1), "rc": rc, "stream_bytes_total": len(z),
           "tail_bytes": len(tail), "tail_sha256": hashlib.sha256(tail).hexdigest(), "compression_ratio": round(len(z) / n, 4), "first_call": meta}
    print("second call: %.1f s, rc %d, stream of %d bytes (%d of them from this call), ratio %.4f" % (dt, rc, len(z), len(tail), len(z) / n), flush=True)
    if os.environ.get("C4_ORACLE") == "1":
        th.join()
        o = oracle_box["res"]
        ok = bool(o[0] == rc and o[2] == crc and len(o[1]) == len(z) and o[1][head_len:] == tail and hashlib.sha256(o[1][:head_len]).hexdigest() == meta["stream_sha256"])
        rec.update(cpu_port_one_core_seconds=round(oracle_box["s"], 1), equals_cpu_port=ok, is_config_4_itself=bool(mib == 1024),
                   total_gpu_seconds=round(dt + meta["seconds"], 1), MBs=round(n / (dt + meta["seconds"]) / 1e6, 3),
                   note="one stream, one chain of adaptive probabilities, coded in TWO GPU calls: the first stopped by its feedback, its state (%d bytes) exported and imported by the second" % len(st))
        print("CPU port (one core): %.1f s; payload == exported head + this call's tail: %s" % (oracle_box["s"], ok), flush=True)
    json.dump(rec, open(os.path.join(outdir, "config4_lzma3_%dmib_silesia_mix_v%d.json" % (mib, ver)), "w"), indent=1)
    sys.exit(0)
tim = {k: round(v, 1) for k, v in enc.last_timing() if not k.startswith("#")}
print("config 4 shape: one LZMA_3 stream of %d MiB: rc %d, %.1f s = %.3f MB/s, ratio %.4f, feedback calls %d" % (mib, rc, dt, n / dt / 1e6, len(z) / n, len(seen)), tim, flush=True)
ds = int.from_bytes(z[5:9], "little")
dec = lzma.LZMADecompressor(format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA1, "dict_size": ds, "lc": 3, "lp": 0, "pb": 2}])
c, tot, off = 0, 0, 9
while off < len(z):
    ch = dec.decompress(z[off:off + (4 << 20)])
    off += 4 << 20
    c = zlib.crc32(ch, c); tot += len(ch)
decodes = bool(tot == n and c == zlib.crc32(d) and (crc ^ 0xFFFFFFFF) == c)
print("liblzma decodes it to the input: %s (dictionary %d MiB, end marker met: %s)" % (decodes, ds >> 20, dec.eof), flush=True)
rec = {"workload": "ONE LZMA_3 stream of %d MiB silesia_mix_v%d" % (mib, ver), "seconds": round(dt, 1), "MB/s": round(n / dt / 1e6, 3), "rc": rc, "compression_ratio": round(len(z) / n, 4),
       "liblzma_decodes_it_to_the_input": decodes, "phase_ms": tim}
if os.environ.get("C4_ORACLE") == "1":
    from _lzmah import oracle_lzma
    t1 = time.time()
    o = oracle_lzma(d, 18)
    dto = time.time() - t1
    print("CPU port (one core): %.1f s = %.2f MB/s; payloads equal: %s" % (dto, n / dto / 1e6, o == (rc, z, crc)), flush=True)
    rec.update(cpu_port_one_core_seconds=round(dto, 1), cpu_port_one_core_MBs=round(n / dto / 1e6, 3), equals_cpu_port=bool(o == (rc, z, crc)), gpu_over_cpu_one_core=round(dto / dt, 3))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(out, exist_ok=True)
json.dump(rec, open(os.path.join(out, "config4_lzma3_%dmib_silesia_mix_v%d.json" % (mib, ver)), "w"), indent=1)
