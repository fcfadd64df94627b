"""Multi-GPU layout of the Deflate path: independent Zip entries shard across ranks
("Across Zip entries: yes, perfectly"; each entry is its own Deflate stream,
Zip.Create.Add_Stream is per entry, zip-create.adb:194-297).  One process per GPU; the only
exchange is the final gather of the per-entry compressed payloads to rank 0, which stitches them
into the .zip stream.  With torch.distributed's "nccl" backend this is RCCL over xGMI; the same
code runs on "gloo" for the CPU tests.
"""
import torch
import torch.distributed as dist


def entry_ranges(total_bytes, entry_bytes):
    """Splits a logical stream into fixed-size entries: [(offset, length), ...]."""
    out, off = [], 0
    while off < total_bytes:
        ln = min(entry_bytes, total_bytes - off)
        out.append((off, ln))
        off += ln
    return out


def entries_of_rank(n_entries, rank, world):
    """Contiguous block partition of entry indices over ranks (rank r owns a contiguous run, so
    the gathered archive keeps the entry order)."""
    base, rem = divmod(n_entries, world)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


class PendingGather:
    """A gather of per-rank payloads in flight (gather_payloads_begin).  finish() waits for it and returns, on dst,
    (list of uint8 tensors trimmed to their lengths, list of meta tensors); None elsewhere.  The payload tensor
    given to gather_payloads_begin must not be overwritten before finish()."""

    def __init__(self, work, bufs, lens, hdrs, send):
        self.work, self.bufs, self.lens, self.hdrs, self.send = work, bufs, lens, hdrs, send

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            # NCCL's wait() only orders the current stream behind the collective: the caller (another stream, or a
            # library with its own stream) is about to overwrite the send buffer, so wait on the host as well
            if self.send.is_cuda:
                torch.cuda.current_stream(self.send.device).synchronize()
        if self.bufs is None:
            return None
        return [b[:ln] for b, ln in zip(self.bufs, self.lens)], [h[1:] for h in self.hdrs]


def gather_payloads_begin(payload, length, meta, dst=0, group=None):
    """Starts gathering one variable-length payload per rank onto `dst` and returns a PendingGather.

    payload : 1-D uint8 tensor (device of the backend), valid in [0, length)
    meta    : 1-D int64 tensor of per-entry metadata (e.g. crc, usize, zip_type)
    The lengths are exchanged first (a small all_gather), then the payloads travel asynchronously, so that a rank
    can compress its next entry meanwhile.  Traffic = sum of compressed sizes: at ratio 0.37 that is ~0.37 x input,
    far below one xGMI link per peer, so a direct gather to the root is used, not a ring."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    gdst = dst if group is None else dist.get_global_rank(group, dst)      # (dst is a rank of the group; torch takes global ranks)
    dev = payload.device
    hdr = torch.cat([torch.tensor([length], dtype=torch.int64, device=dev), meta.to(dev)])
    hdrs = [torch.empty_like(hdr) for _ in range(world)]
    dist.all_gather(hdrs, hdr, group=group)
    lens = [int(h[0].item()) for h in hdrs]
    maxlen = max(max(lens), 1)
    send = payload[:maxlen] if payload.numel() >= maxlen else torch.cat([payload, torch.zeros(maxlen - payload.numel(), dtype=torch.uint8, device=dev)])
    send = send.contiguous()
    if rank == dst:
        bufs = [torch.empty(maxlen, dtype=torch.uint8, device=dev) for _ in range(world)]
        work = dist.gather(send, bufs, dst=gdst, group=group, async_op=True)
        return PendingGather(work, bufs, lens, hdrs, send)
    work = dist.gather(send, None, dst=gdst, group=group, async_op=True)
    return PendingGather(work, None, lens, hdrs, send)


def gather_payloads(payload, length, meta, dst=0, group=None):
    """gather_payloads_begin(...).finish(): the blocking form."""
    return gather_payloads_begin(payload, length, meta, dst=dst, group=group).finish()


class PendingStream:
    """A gather of the ranges' bytes of ONE stream in flight (gather_stream_begin).  finish() waits for it and returns, on dst, the
    stream as a 1-D uint8 tensor of ceil(total_bits / 8) bytes; None elsewhere.  The payload handed to gather_stream_begin must not be
    overwritten before finish()."""

    def __init__(self, works, out, edges, layout, keep):
        self.works, self.out, self.edges, self.layout, self.keep = works, out, edges, layout, keep

    def finish(self):
        for w in self.works:
            w.wait()
        if self.works and self.keep is not None and self.keep.is_cuda:
            # NCCL's wait() only orders the current stream behind the transfer; the caller is about to reuse the payload from another stream
            torch.cuda.current_stream(self.keep.device).synchronize()
        self.works = []
        if self.out is None:
            return None
        if self.edges is not None:
            # the only bytes two ranges can share are a range's first and last: OR them together on the host (2 bytes per range) and
            # write them with one indexed store; every other byte of the stream was received (or copied) straight into its place
            e = torch.stack(self.edges).cpu().tolist()
            merged = {}
            for (off, ln), (first, last) in zip(self.layout, (x[:2] for x in e)):
                if ln >= 1:
                    merged[off] = merged.get(off, 0) | first
                if ln >= 2:
                    merged[off + ln - 1] = merged.get(off + ln - 1, 0) | last
            if merged:
                idx = torch.tensor(sorted(merged), dtype=torch.int64, device=self.out.device)
                val = torch.tensor([merged[k] for k in sorted(merged)], dtype=torch.uint8, device=self.out.device)
                self.out[idx] = val
            self.edges = None
        return self.out


def stream_layout(spans, world):
    """Byte offset and byte count of every rank's payload inside the stream, from the bit spans every rank knows: [(off, ln), ...]
    ((0, 0) for a rank without a range)."""
    out = []
    for k in range(world):
        sp = spans[k] if k < len(spans) else None
        if sp is None:
            out.append((0, 0))
        else:
            b0, b1 = sp
            out.append((b0 // 8, max(0, (b1 + 7) // 8 - b0 // 8)))
    return out


def gather_stream_begin(payload, spans, total_bits, dst=0, group=None):
    """Starts gathering the ranges' bytes of ONE stream onto `dst`, every range straight into its place ("Gather-v of
    bit-strings to rank 0 ... then a shift-merge"; here the ranges are emitted at their final bit phase, so there is nothing to shift):

      * every rank knows every range's bit span (the `spans` all_gather of deflate_stream_rank / the replayed choice of bzip2_stream_rank),
        hence every payload's byte offset and length: no length exchange, no padding to the longest payload;
      * `dst` allocates the stream ONCE and posts one receive per peer for the peer's INTERIOR bytes (all but its first and last byte),
        directly at stream[off + 1 : off + ln - 1] -- a byte strictly inside a range's span holds that range's bits only;
      * the first and last byte of a range may be shared with its neighbours: they travel as one small fixed-size gather (8 bytes per
        rank) and are OR-ed together on the host in finish() -- at most 2 x world bytes.

    payload : this rank's 1-D uint8 tensor, valid in [0, ln) with (off, ln) = stream_layout(spans)[rank]; bits outside its span zero.
    Peak memory on dst: the stream plus its own payload (the padded gather held world x longest + the stream)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)

    def g(k):
        return k if group is None else dist.get_global_rank(group, k)
    layout = stream_layout(spans, world)
    off, ln = layout[rank]
    dev = payload.device
    edge = torch.zeros(8, dtype=torch.uint8, device=dev)
    if ln >= 1:
        edge[0] = payload[0]
        edge[1] = payload[ln - 1]
    works = []
    if rank == dst:
        out = torch.empty((total_bits + 7) // 8, dtype=torch.uint8, device=dev)
        edges = [torch.empty(8, dtype=torch.uint8, device=dev) for _ in range(world)]
        works.append(dist.gather(edge, edges, dst=g(dst), group=group, async_op=True))
        ops = []
        for k in range(world):
            ko, kl = layout[k]
            if kl <= 2:
                continue
            if k == rank:
                out[ko + 1:ko + kl - 1].copy_(payload[1:kl - 1])
            else:
                ops.append(dist.P2POp(dist.irecv, out[ko + 1:ko + kl - 1], g(k), group))
        if ops:
            works.extend(dist.batch_isend_irecv(ops))
        return PendingStream(works, out, edges, layout, payload)
    works.append(dist.gather(edge, None, dst=g(dst), group=group, async_op=True))
    if ln > 2:
        body = payload[1:ln - 1]
        works.extend(dist.batch_isend_irecv([dist.P2POp(dist.isend, body, g(dst), group)]))
    return PendingStream(works, None, None, layout, payload)


def gather_stream(payload, spans, total_bits, dst=0, group=None):
    """gather_stream_begin(...).finish(): the blocking form."""
    return gather_stream_begin(payload, spans, total_bits, dst=dst, group=group).finish()


# ======================================================================================================================
# ONE stream over several GPUs (primary mode): the stream is cut into ranges at multiples of 64 KiB, rank r
# compresses range r with the zada_range_* calls of libzada_hip.so and the ranks exchange exactly the state the
# reference's sequential encoder carries through the stream (include/zada.h "One stream over several contexts"):
#
#   a. parser state at the range boundaries   ONE all_gather of a fixed-size int64 tensor (atoms, exit, warm, CRC, bytes) -- 64 bytes per rank.  A range whose
#                                             warm-up parse did not meet its neighbour's exit state re-runs its LZ stage
#                                             from that state (degenerate data only, e.g. one long run)
#   b. atom counts -> position of every range on the stream's 65 536-atom flush grid (zip-compress-deflate.adb:1424-1432)
#   c. boundary atoms                         all_gather of each range's first 65 536 and last 2 048 atoms (540 KB per
#                                             rank): a range completes its last flush with its successors' atoms and
#                                             reads 2 048 atoms back into its predecessors' (:1338-1360, 1372)
#   d. Send_as_block's state                  352 bytes handed from rank to rank (send / recv), the one sequential step
#   e. payloads                               received by rank 0 straight at their byte offsets in the stream (gather_stream_begin); only a
#                                             range's first and last byte can be shared with a neighbour: those are OR-ed (2 bytes per rank)
#
# With torch.distributed's "nccl" backend these are RCCL collectives over xGMI; TorchComm runs the same code on "gloo"
# (CPU tests), ThreadComm inside one process (several contexts on one GPU: tests/test_ranges.py).
# ======================================================================================================================
RANGE_ALIGN, RANGE_PRE, RANGE_POST, EDGE_HEAD, EDGE_TAIL, FLUSH, HALF_SLIDER = 65536, 32768, 1 << 20, 65536, 2048, 65536, 2048


def stream_ranges(stream_size, world):
    """Cuts [0, stream_size) into `world` ranges at multiples of 64 KiB, as even as possible; trailing ranges may be
    empty for tiny streams (they are dropped: fewer ranks take part).  Returns [(lo, n), ...] of the non-empty ranges
    (at least one, possibly (0, 0))."""
    units = (stream_size + RANGE_ALIGN - 1) // RANGE_ALIGN
    base, rem = divmod(units, world)
    out, lo = [], 0
    for r in range(world):
        u = base + (1 if r < rem else 0)
        n = min(u * RANGE_ALIGN, stream_size - lo)
        if n > 0:
            out.append((lo, n))
        lo += n
    return out or [(0, 0)]


def range_window(stream_size, lo, n):
    """Bytes of the stream a rank must hold for range [lo, lo + n): (first byte, pre, post)."""
    pre = RANGE_PRE if lo > 0 else 0
    post = min(RANGE_POST, stream_size - (lo + n))
    return lo - pre, pre, post


def needed_neighbour_atoms(before, own, total, fixed_only=False):
    """How many atoms of its predecessors (look-behind) and successors (look-ahead) a range needs, given the atoms of
    the stream before it, its own and the stream's total.  Mirrors range_place in csrc/zada_api.hip."""
    if fixed_only:
        return 0, 0
    f0 = (before + FLUSH - 1) // FLUSH * FLUSH
    nflush = (before + own - f0 + FLUSH - 1) // FLUSH if before + own > f0 else 0
    if nflush == 0:
        return 0, 0
    last_end = min(f0 + nflush * FLUSH, total)
    n_la = max(0, last_end - (before + own))
    n_lb = HALF_SLIDER - (f0 - before) if (f0 > 0 and f0 - before < HALF_SLIDER) else 0
    return n_lb, n_la


class TorchComm:
    """The exchanges of deflate_stream_rank over torch.distributed (RCCL on the GPU box, gloo in the CPU tests)."""

    def __init__(self, device, group=None):
        self.group, self.device = group, device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def _g(self, k):
        """Rank k of the group as torch.distributed addresses it (src / dst are GLOBAL ranks also when a group is given)."""
        return k if self.group is None else dist.get_global_rank(self.group, k)

    def all_gather_i64(self, values):
        """Every rank's vector of `len(values)` integers (the same length on every rank), as a list of lists: ONE all_gather of a fixed-size
        int64 tensor (64 bytes per rank for the parser states, 24 for the bit positions) -- no pickling, no object collectives, one
        device-to-host copy of world x len(values) numbers on the nccl backend."""
        t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=self.device)
        out = torch.empty(self.world * t.numel(), dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        return out.cpu().view(self.world, t.numel()).tolist()

    def all_gather_i64_var(self, arr):
        """Every rank's 1-D array of unsigned 64-bit numbers, lengths differing (the BZip2 block tables): the lengths by all_gather_i64, then ONE
        all_gather of the arrays padded to the longest.  Returns a list of numpy uint64 arrays."""
        import numpy as np
        a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
        lens = [v[0] for v in self.all_gather_i64([a.size])]
        width = max(max(lens), 1)
        pad = np.zeros(width, dtype=np.int64)
        pad[:a.size] = a.view(np.int64)
        t = torch.from_numpy(pad).to(self.device)
        out = torch.empty(self.world * width, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        h = out.cpu().numpy().reshape(self.world, width)
        return [h[k, :lens[k]].view(np.uint64).copy() for k in range(self.world)]

    def all_gather_dev(self, t):
        if t.is_cuda and dist.get_backend(self.group) == "gloo":     # CPU-only transport (tests, BENCH_EMULATE): stage through the host
            torch.cuda.synchronize()
            h = t.cpu()
            out = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(out, h, group=self.group)
            return [o.to(t.device) for o in out]
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return out

    def send_bytes(self, b, dst):
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(self.device)
        dist.send(t, dst=self._g(dst), group=self.group)

    def recv_bytes(self, n, src):
        return self.recv_bytes_begin(n, src)()

    def recv_bytes_begin(self, n, src):
        """Posts the receive now and returns a function that waits for it and returns the bytes: the 352-byte chooser state is asked for BEFORE
        the range's analysis, so that the hop from the rank before costs its own latency only if that rank is late.  (Posted after the last
        collective in front of it: operations of one communicator run in the order they were issued on every rank.)"""
        t = torch.empty(n, dtype=torch.uint8, device=self.device)
        work = dist.irecv(t, src=self._g(src), group=self.group)

        def wait():
            work.wait()
            return bytes(t.cpu().numpy())
        return wait

    def gather_stream(self, payload, spans, total_bits, dst=0):
        return gather_stream(payload, spans, total_bits, dst=dst, group=self.group)


def _gather_i64(comm, values):
    """comm.all_gather_i64, or the same through a comm double that only gathers Python objects (threads inside one process)."""
    if hasattr(comm, "all_gather_i64"):
        return comm.all_gather_i64(values)
    return [list(v) for v in comm.all_gather_obj([int(x) for x in values])]


def _gather_i64_var(comm, arr):
    import numpy as np
    if hasattr(comm, "all_gather_i64_var"):
        return comm.all_gather_i64_var(arr)
    return [np.frombuffer(b, np.uint64) for b in comm.all_gather_obj(np.ascontiguousarray(arr, np.uint64).tobytes())]


def _info_vec(info):
    """A range's parser state as eight integers (the all_gather of step a): active, atoms, exit, warm-up state, raw CRC register, bytes."""
    if info is None:
        return [0] * 8
    return [1, info["atoms"], info["exit"][0], info["exit"][1], info["warm"][0], info["warm"][1], info["crc_raw"], info["n"]]


def _info_of(v):
    return None if not v[0] else dict(atoms=v[1], exit=(v[2], v[3]), warm=(v[4], v[5]), crc_raw=v[6], n=v[7])


def deflate_stream_rank(enc, comm, tensors, stream_size, ranges, d_in_ptr, method, alloc_u32, alloc_out):
    """Rank `comm.rank`'s part of compressing ONE stream cut into `ranges` (stream_ranges); ranks beyond len(ranges)
    only take part in the collectives.

    enc        Encoder (or a test double with the same range_* methods)
    d_in_ptr   device address of stream byte lo - pre of this rank's window (range_window)
    alloc_u32  n -> 1-D uint32 tensor of n elements on the encoder's device (exchange buffers)
    alloc_out  n -> 1-D uint8 tensor (this rank's payload)
    tensors    module with the tensor ops the protocol needs (torch)
    Returns dict(bit_begin, bit_end, total_bits, payload (uint8 tensor holding stream bytes [bit_begin // 8,
    ceil(bit_end / 8))), nbytes, crc_raw, n, infos, inefficient)."""
    import time
    r, W = comm.rank, comm.world
    nr = len(ranges)
    active = r < nr
    fixed_only = method == 6
    info = None
    xs = {"all_gather_state": 0.0, "boundary_atoms": 0.0, "carry_chain": 0.0, "spans": 0.0}     # seconds inside the exchange steps (waiting included)
    if active:
        lo, n = ranges[r]
        _, pre, post = range_window(stream_size, lo, n)
        enc.range_open(d_in_ptr, stream_size, lo, n, pre, post, method)
        info = enc.range_lz(None)
        info["n"] = n
    # ---- a. parser states at the boundaries
    t_x = time.perf_counter()
    infos = [_info_of(v) for v in _gather_i64(comm, _info_vec(info))]
    for k in range(1, nr):
        if tuple(infos[k]["warm"]) != tuple(infos[k - 1]["exit"]):
            # the warm-up parse of range k did not meet the true one: run it again from the true state, and everybody hears the new one (rare:
            # data like one long run; every rank sees the same states, so every rank comes here)
            if r == k:
                info = enc.range_lz(tuple(infos[k - 1]["exit"]))
                info["n"] = ranges[k][1]
            infos = [_info_of(v) for v in _gather_i64(comm, _info_vec(info))]
    xs["all_gather_state"] += time.perf_counter() - t_x
    # ---- b. the ranges on the flush grid
    counts = [infos[k]["atoms"] for k in range(nr)]
    total = sum(counts)
    before = [sum(counts[:k]) for k in range(nr)]
    # ---- c. boundary atoms
    head_a, head_p, tail_a, tail_p = alloc_u32(EDGE_HEAD), alloc_u32(EDGE_HEAD), alloc_u32(EDGE_TAIL), alloc_u32(EDGE_TAIL)
    if active and not fixed_only:
        enc.range_edges(head_a.data_ptr(), head_p.data_ptr(), tail_a.data_ptr(), tail_p.data_ptr())
    need = [needed_neighbour_atoms(before[k], counts[k], total, fixed_only) for k in range(nr)]
    lb_a = lb_p = la_a = la_p = None
    n_lb = n_la = 0
    if not fixed_only and any(a or b for a, b in need):
        t_x = time.perf_counter()
        heads_a, heads_p = comm.all_gather_dev(head_a), comm.all_gather_dev(head_p)
        tails_a, tails_p = comm.all_gather_dev(tail_a), comm.all_gather_dev(tail_p)
        xs["boundary_atoms"] += time.perf_counter() - t_x
        if active:
            n_lb, n_la = need[r]
            if n_lb:                                  # the last n_lb (<= 2 048) atoms before this range: tails of r-1, r-2, ...
                parts_a, parts_p, left, k = [], [], n_lb, r - 1
                while left > 0:
                    have = min(counts[k], EDGE_TAIL)  # tails[k] holds the last `have` atoms of range k
                    take = min(left, have)
                    parts_a.insert(0, tails_a[k][have - take:have]); parts_p.insert(0, tails_p[k][have - take:have])
                    left -= take
                    k -= 1
                lb_a, lb_p = tensors.cat(parts_a).contiguous(), tensors.cat(parts_p).contiguous()
            if n_la:                                  # the first n_la (< 65 536) atoms behind this range: heads of r+1, r+2, ...
                parts_a, parts_p, left, k = [], [], n_la, r + 1
                while left > 0:
                    take = min(left, counts[k], EDGE_HEAD)
                    parts_a.append(heads_a[k][:take]); parts_p.append(heads_p[k][:take])
                    left -= take
                    k += 1
                la_a, la_p = tensors.cat(parts_a).contiguous(), tensors.cat(parts_p).contiguous()
    # ---- d. (first half) the receive of the chooser's state from the rank before is posted before this range is analysed
    carry_wait = None
    if active and r > 0:
        carry_wait = comm.recv_bytes_begin(352, r - 1) if hasattr(comm, "recv_bytes_begin") else (lambda: comm.recv_bytes(352, r - 1))
    if active:
        if (n_lb or n_la) and hasattr(tensors, "cuda") and (lb_a if n_lb else la_a).is_cuda:
            tensors.cuda.current_stream().synchronize()      # the slices were put together on torch's stream, the encoder reads them on its own
        enc.range_place(before[r], total, lb_a.data_ptr() if n_lb else None, lb_p.data_ptr() if n_lb else None, n_lb,
                        la_a.data_ptr() if n_la else None, la_p.data_ptr() if n_la else None, n_la)
        enc.range_analyze()
    # ---- d. the block decisions: the one sequential step, 352 bytes from rank to rank
    bit_begin = bit_end = 0
    if active:
        t_x = time.perf_counter()
        carry = carry_wait() if carry_wait is not None else None
        xs["carry_chain"] += time.perf_counter() - t_x
        carry_out, bit_begin, bit_end = enc.range_choose(carry)
        if r + 1 < nr:
            t_x = time.perf_counter()
            comm.send_bytes(carry_out, r + 1)
            xs["carry_chain"] += time.perf_counter() - t_x
    t_x = time.perf_counter()
    spans = [(v[1], v[2]) if v[0] else None for v in _gather_i64(comm, [1 if active else 0, bit_begin, bit_end])]
    xs["spans"] += time.perf_counter() - t_x
    total_bits = spans[nr - 1][1]
    inefficient = (total_bits + 7) // 8 >= stream_size            # Compression_inefficient, zip-compress.adb:479-486
    # ---- e. this rank's bytes of the stream
    payload, nbytes = None, 0
    if active and not inefficient:
        cap = (bit_end + 7) // 8 - bit_begin // 8 + 64
        payload = alloc_out(cap)
        nbytes = enc.range_emit(payload.data_ptr(), cap)
    return dict(bit_begin=bit_begin, bit_end=bit_end, total_bits=total_bits, payload=payload, nbytes=nbytes, spans=spans,
                infos=infos, inefficient=inefficient, crc_raw=info["crc_raw"] if active else 0, exchange_s=xs)


def stitch_stream(tensors, payloads, spans, total_bits, device):
    """The ranges' bytes OR-ed into one stream (a byte shared by two ranges holds bits of both) -- for payloads that are already in one
    process (several contexts on one GPU: tests/test_ranges.py).  Between ranks gather_stream_begin puts every range in its place."""
    out = tensors.zeros((total_bits + 7) // 8, dtype=tensors.uint8, device=device)
    for p, span in zip(payloads, spans):
        if span is None or p is None:
            continue
        b0, b1 = span
        off, ln = b0 // 8, (b1 + 7) // 8 - b0 // 8
        if ln > 0:
            out[off:off + ln] |= p[:ln].to(device)
    return out


def stream_crc(enc_or_lib_combine, infos, reg=0xFFFFFFFF):
    """CRC-32 register after the whole stream from the ranges' raw registers (zada_crc32_combine)."""
    for i in infos:
        if i is not None:
            reg = enc_or_lib_combine(reg, i["crc_raw"], i["n"])
    return reg


# ------------------------------------------------------------------------------------------------------------------
#  One BZip2 stream over the GPUs of a node (BASELINE config 5).  A rank takes the blocks that START inside its range; the
#  block chain (where the next block starts) and the bit phase are the only things that run along the stream:
#    a. 8 bytes from rank to rank: where the range's first block starts (block limits only: milliseconds per GiB)
#    b. every rank encodes every piece of every tactic of its blocks (the work, no communication)
#    c. all_gather of the per-block tables (12 values per block); every rank replays the choice of the tactics
#    d. every rank assembles its bytes of the stream; the payloads are gathered and OR-ed at the joints (stitch_stream)
# ------------------------------------------------------------------------------------------------------------------
def bzip2_halo(method=14):
    """Bytes of the following ranges a rank needs behind its own: a block takes at most ten capacities (bzip2-encoding.adb:1156-1158)."""
    return 10 * {12: 100_000, 13: 400_000, 14: 900_000}[method] + 512


BZ_HALO = bzip2_halo(14)


def bzip2_ranges(stream_size, world, method=14):
    """Ranges of at least two halos each (so that a rank's halo lies in the next range); fewer ranges than ranks for short streams."""
    nr = max(1, min(world, stream_size // (2 * bzip2_halo(method))))
    step = (stream_size + nr - 1) // nr if stream_size else 0
    return [(k * step, min(step, stream_size - k * step)) for k in range(nr)] if stream_size else [(0, 0)]


def bzip2_window(stream_size, lo, n, method=14):
    """The bytes a rank holds: its range and the halo behind it.  Returns (buffer offset in the stream, buffer length)."""
    return lo, min(stream_size, lo + n + bzip2_halo(method)) - lo


def bzip2_stream_rank(enc, comm, stream_size, ranges, d_buf_ptr, method, alloc_out):
    """Rank `comm.rank`'s part of ONE BZip2 stream cut into `ranges` (bzip2_ranges); d_buf_ptr = device address of the rank's
    window (bzip2_window).  Returns dict(bit_begin, bit_end, total_bits, payload, nbytes, spans, blocks, inefficient)."""
    import numpy as np
    r, nr = comm.rank, len(ranges)
    active = r < nr
    tab = np.zeros((0, 4, 3), np.uint64)
    crc_raw = None
    if active:
        lo, n = ranges[r]
        off, blen = bzip2_window(stream_size, lo, n, method)
        start = int.from_bytes(comm.recv_bytes(8, r - 1), "little") if r > 0 else 0
        nxt, _nb = enc.bz2_range_open(d_buf_ptr, blen, off, stream_size, start, lo + n if r + 1 < nr else stream_size, method)
        if r + 1 < nr:
            comm.send_bytes(int(nxt).to_bytes(8, "little"), r + 1)
        enc.bz2_range_encode()
        tab = enc.bz2_range_table()
        if hasattr(enc, "crc32_device") and n and d_buf_ptr % 16 == 0:      # the rank's piece of the Zip CRC-32 (its own range, not the halo)
            crc_raw = enc.crc32_device(d_buf_ptr, n)
    tabs = _gather_i64_var(comm, tab if active else np.zeros(0, np.uint64))
    bp, crc = 32, 0
    spans, mine = [], None
    for k in range(nr):
        tk = tabs[k].reshape(-1, 4, 3)
        ch, bp2, crc2 = enc.bz2_select(tk, bp, crc)
        b0 = 0 if k == 0 else bp
        b1 = bp2 + (80 if k == nr - 1 else 0)
        spans.append((b0, b1))
        if k == r:
            mine = (ch, bp)
        bp, crc = bp2, crc2
    total_bits = bp + 80
    inefficient = (total_bits + 7) // 8 >= stream_size
    payload, nbytes = None, 0
    if active:
        b0, b1 = spans[r]
        cap = (b1 + 7) // 8 - b0 // 8 + 64
        payload = alloc_out(cap)
        nbytes = enc.bz2_range_assemble(mine[0], mine[1], payload.data_ptr(), cap, header=(r == 0), footer_crc=crc if r == nr - 1 else None)
    return dict(bit_begin=spans[r][0] if active else 0, bit_end=spans[r][1] if active else 0, total_bits=total_bits, payload=payload, nbytes=nbytes,
                spans=spans, blocks=enc.bz2_last_blocks() if active else [], inefficient=inefficient, combined_crc=crc,
                crc_raw=crc_raw, n=ranges[r][1] if active else 0)
