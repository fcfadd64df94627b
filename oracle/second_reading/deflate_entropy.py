"""SECOND READING of the reference's entropy stage -- TEST INFRASTRUCTURE, pins nothing.

A plain-Python transcription of the Ada text, written in round 6 from the reference's sources alone
(`oracle/zada_oracle.c` was not opened while writing it), so that the C oracle and this file are two
independent readings of the same lines.  `tests/test_second_reading.py` compares the two stream for
stream; agreement does NOT pin the oracle to the Ada binary (no GNAT in the image) -- it only makes
it less likely that one misreading of a tie-break sits under every digest.

What is transcribed (all in /root/reference/zip_lib/):
  zip-compress-deflate.adb:101-160    Put_byte / Flush_bit_buffer / Put_Bits (32-bit accumulator)
  zip-compress-deflate.adb:238-318    Tweak_for_better_RLE
  zip-compress-deflate.adb:324-371    Build_descriptors + Patch_statistics_for_buggy_decoders
  zip-compress-deflate.adb:382-508    Convert, L1_tweaked, Similar, Recyclable
  zip-compress-deflate.adb:549-704    Put_Compression_Structure
  zip-compress-deflate.adb:709-728    fixed descriptors, Put_literal_byte
  zip-compress-deflate.adb:757-918    length / distance code tables, Put_DL_code
  zip-compress-deflate.adb:953-1062   Get_statistics, Put_LZ_buffer, Mark_new_block, Expand_LZ_buffer
  zip-compress-deflate.adb:1105-1269  Send_as_block
  zip-compress-deflate.adb:1294-1432  Scan_and_send_from_main_buffer, Flush_half_buffer, Push
  zip-compress-deflate.adb:1593-1635  Encode's prologue and epilogue
  huffman-encoding.adb:34-80          Invert, Prepare_Codes
  huffman-encoding-length_limited_coding.adb:46-280   the whole procedure, pool and garbage collection included

What is NOT transcribed: the LZ77 front end (lz77.adb) -- the atoms come in as arguments (the tests take
them from the oracle's LZ77 stage, which zlib pins) -- and the Compression_inefficient rule of
zip-compress.adb:479-486 (the caller compares lengths).

Pure-Python loops: meant for inputs up to a few MiB.
"""

BIG = (1 << 63) // 2 - 2          # Count_type'Last (zip-compress-deflate.adb:225)
NULL = -1                          # null_index (length_limited_coding.adb:52): any value that is no pool index


# ------------------------------------------------------------------ huffman-encoding-length_limited_coding.adb:46-280
def length_limited_coding(frequencies, max_bits):
    n_alpha = len(frequencies)
    bit_lengths = [0] * n_alpha
    pool_size = 2 * max_bits * (max_bits + 1)
    # Node = [weight, count, tail, in_use]
    pool = [[0, 0, NULL, False] for _ in range(pool_size)]
    st = {"pool_next": 0}
    lists = [[NULL, NULL] for _ in range(max_bits)]
    leaves = []                    # (weight, symbol)
    for a in range(n_alpha):       # :228-235
        if frequencies[a] > 0:
            leaves.append((frequencies[a], a))
    num_symbols = len(leaves)
    if num_symbols > 2 ** max_bits:
        raise ValueError("too_many_symbols_for_length_limit")
    if num_symbols == 0:
        return bit_lengths
    if num_symbols == 1:
        bit_lengths[leaves[0][1]] = 1
        return bit_lengths

    def init_node(weight, count, tail, node_idx):          # :86-92
        nd = pool[node_idx]
        nd[0] = weight
        nd[1] = count
        nd[2] = tail
        nd[3] = True

    def get_free_node(use_lists):                          # :97-122
        while True:
            if st["pool_next"] > pool_size - 1:
                for nd in pool:
                    nd[3] = False
                if use_lists:
                    for i in range(max_bits * 2):
                        node_idx = lists[i // 2][i % 2]
                        while node_idx != NULL:
                            pool[node_idx][3] = True
                            node_idx = pool[node_idx][2]
                st["pool_next"] = 0
            if not pool[st["pool_next"]][3]:
                break
            st["pool_next"] += 1
        st["pool_next"] += 1
        return st["pool_next"] - 1

    def boundary_pm(index, final):                         # :131-165
        lastcount = pool[lists[index][1]][1]
        if index == 0 and lastcount >= num_symbols:
            return
        newchain = get_free_node(True)
        oldchain = lists[index][1]
        lists[index] = [oldchain, newchain]
        if index == 0:
            init_node(leaves[lastcount][0], lastcount + 1, NULL, newchain)
        else:
            s = pool[lists[index - 1][0]][0] + pool[lists[index - 1][1]][0]
            if lastcount < num_symbols and s > leaves[lastcount][0]:
                init_node(leaves[lastcount][0], lastcount + 1, pool[oldchain][2], newchain)
            else:
                init_node(s, lastcount, lists[index - 1][1], newchain)
                if not final:
                    boundary_pm(index - 1, False)
                    boundary_pm(index - 1, False)

    def quick_sort(first, last):                           # :196-223 on the slice leaves(first .. last)
        n = last - first + 1
        if n < 2:
            return
        p = leaves[n // 2 + first]
        i = 0
        j = n - 1
        while True:
            while leaves[i + first][0] < p[0]:
                i += 1
            while p[0] < leaves[j + first][0]:
                j -= 1
            if i >= j:
                break
            leaves[i + first], leaves[j + first] = leaves[j + first], leaves[i + first]
            i += 1
            j -= 1
        quick_sort(first, first + i - 1)
        quick_sort(first + i, last)

    quick_sort(0, num_symbols - 1)
    # Init_Lists :169-176
    node0 = get_free_node(False)
    node1 = get_free_node(False)
    init_node(leaves[0][0], 1, NULL, node0)
    init_node(leaves[1][0], 2, NULL, node1)
    for k in range(max_bits):
        lists[k] = [node0, node1]
    runs = 2 * num_symbols - 4
    for i in range(1, runs + 1):
        boundary_pm(max_bits - 1, i == runs)
    node_idx = lists[max_bits - 1][1]                      # Extract_Bit_Lengths :182-191
    while node_idx != NULL:
        for i in range(pool[node_idx][1]):
            bit_lengths[leaves[i][1]] += 1
        node_idx = pool[node_idx][2]
    return bit_lengths


# ------------------------------------------------------------------ huffman-encoding.adb:34-80
def prepare_codes(bls, max_huffman_bits=15, invert_bit_order=True):
    bl_count = [0] * (max_huffman_bits + 1)
    next_code = [0] * (max_huffman_bits + 1)
    for bl in bls:
        bl_count[bl] += 1                 # bit length 0 is counted too (:56-59)
    code = 0
    for bits in range(1, max_huffman_bits + 1):
        code = (code + bl_count[bits - 1]) * 2
        next_code[bits] = code
    codes = []
    for bl in bls:
        if bl > 0:
            codes.append(next_code[bl])
            next_code[bl] += 1
        else:
            codes.append(0)
    if invert_bit_order:
        for k, bl in enumerate(bls):      # Invert :34-43
            a = codes[k]
            b = 0
            for _ in range(bl):
                b = b * 2 + a % 2
                a //= 2
            codes[k] = b
    return codes


# ------------------------------------------------------------------ tables, zip-compress-deflate.adb:709-716, 757-803, 886-918, 1066-1091
DEFAULT_LIT_LEN_BL = [8] * 144 + [9] * 112 + [7] + [7] * 23 + [8] * 8
DEFAULT_DIS_BL = [5] * 32
END_OF_BLOCK = 256


def _code_for_length():
    t = {}
    for L in range(3, 11):
        t[L] = 257 + (L - 3)
    rows = ((11, 12, 265), (13, 14, 266), (15, 16, 267), (17, 18, 268), (19, 22, 269), (23, 26, 270), (27, 30, 271), (31, 34, 272),
            (35, 42, 273), (43, 50, 274), (51, 58, 275), (59, 66, 276), (67, 82, 277), (83, 98, 278), (99, 114, 279), (115, 130, 280),
            (131, 162, 281), (163, 194, 282), (195, 226, 283), (227, 257, 284), (258, 258, 285))
    for lo, hi, c in rows:
        for L in range(lo, hi + 1):
            t[L] = c
    return t


CODE_FOR_LENGTH = _code_for_length()


def _len_extra(L):                    # :789-803 -> (offset, bits)
    if 3 <= L <= 10 or L == 258:
        return (None, 0)
    if L <= 18:
        return (11, 1)
    if L <= 34:
        return (19, 2)
    if L <= 66:
        return (35, 3)
    if L <= 130:
        return (67, 4)
    return (131, 5)


_DIST_ROWS = ((1, 4, 0, 1, 0), (5, 8, 4, 2, 1), (9, 16, 6, 4, 2), (17, 32, 8, 8, 3), (33, 64, 10, 16, 4), (65, 128, 12, 32, 5),
              (129, 256, 14, 64, 6), (257, 512, 16, 128, 7), (513, 1024, 18, 256, 8), (1025, 2048, 20, 512, 9),
              (2049, 4096, 22, 1024, 10), (4097, 8192, 24, 2048, 11), (8193, 16384, 26, 4096, 12), (16385, 32768, 28, 8192, 13))


def dist_code(distance):              # :886-918 / :841-883 -> (code, extra value, extra bits)
    for lo, hi, base, div, eb in _DIST_ROWS:
        if lo <= distance <= hi:
            return (base + (distance - lo) // div, (distance - lo) % div, eb)
    raise ValueError(distance)


EXTRA_LEN_CODE = {c: e for c, e in zip(range(257, 286), [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0])}   # :1066-1074
EXTRA_DIS_CODE = [0, 0, 0, 0] + [e for e in range(1, 14) for _ in (0, 1)]                                                  # :1076-1091
MAX_EXPAND = 14
CODE_FOR_MAX_EXPAND = 266
TWEAK = (None, 100, 255, 379, 490, 594, 694, 791, 885, 978, 1069, 1159, 1249, 1338, 1426, 1513, 1600)                     # :417-421

MIN_STEP = 750
STEP_CHOICE = ((8 * MIN_STEP, 420), (4 * MIN_STEP, 430), (MIN_STEP, 2050))      # :1304-1308, all L1_tweaked
SLIDER_SIZE = 4096
HALF_SLIDER = SLIDER_SIZE // 2
SLIDER_MAX = SLIDER_SIZE - 1
LZ_BUFFER_SIZE = 1 << 17
RING = LZ_BUFFER_SIZE - 1

# Deflation_Method: 6 = Deflate_Fixed, 7 = Deflate_0, 8 = Deflate_1, 9 = Deflate_2, 10 = Deflate_3 (zip-compress.ads:78-83 in the order the
# repo's C ABI numbers them); max_choice :1310-1311
MAX_CHOICE = {7: 3, 8: 1, 9: 2, 10: 3}


def tweak_for_better_rle(counts):     # :238-318, in place
    length = len(counts)
    good_for_rle = [False] * len(counts)
    while True:
        if length == 0:
            return
        if counts[length - 1] != 0:
            break
        length -= 1
    symbol = counts[0]
    stride = 0
    for i in range(0, length + 1):
        if i == length or counts[i] != symbol:
            if (symbol == 0 and stride >= 5) or (symbol != 0 and stride >= 7):
                for k in range(stride):
                    good_for_rle[i - k - 1] = True
            stride = 1
            if i != length:
                symbol = counts[i]
        else:
            stride += 1
    stride = 0
    limit = counts[0]
    s = 0
    for i in range(0, length + 1):
        if (i == length or good_for_rle[i] or (i > 0 and good_for_rle[i - 1])
                or abs(counts[i] - limit) >= 4):
            if stride >= 4 or (stride >= 3 and s == 0):
                new_count = max(1, (s + stride // 2) // stride)
                if s == 0:
                    new_count = 0
                for k in range(stride):
                    counts[i - k - 1] = new_count
            stride = 0
            s = 0
            if i < length - 3:
                limit = (counts[i] + counts[i + 1] + counts[i + 2] + counts[i + 3] + 2) // 4
            elif i < length:
                limit = counts[i]
            else:
                limit = 0
        stride += 1
        if i != length:
            s += counts[i]


def build_descriptors_from_stats(stats_lit_len, stats_dis):      # :324-371 -> (bl lit_len[288], bl dis[32])
    dis = list(stats_dis)
    used = sum(1 for c in dis if c != 0)
    if used == 0:
        dis[0] = 1
        dis[1] = 1
    elif used == 1:
        if dis[0] == 0:
            dis[0] = 1
        else:
            dis[1] = 1
    return (length_limited_coding(stats_lit_len, 15), length_limited_coding(dis, 15))


def convert(descr):                   # :382-403
    return [16 if bl == 0 else bl for bl in descr[0]] + [16 if bl == 0 else bl for bl in descr[1]]


def similar(h1, h2, threshold):       # :457-490 with dist_kind = L1_tweaked
    b1 = convert(h1)
    b2 = convert(h2)
    thres = threshold * TWEAK[1]
    dist = 0
    for x, y in zip(b1, b2):
        dist += abs(TWEAK[x] - TWEAK[y])
    return dist < thres, dist, thres


def recyclable(h_old, h_new):         # :495-508
    for part in (0, 1):
        for o, n in zip(h_old[part], h_new[part]):
            if o == 0 and n > 0:
                return False
    return True


class Atom(object):
    __slots__ = ("kind", "plain", "distance", "length", "expanded")

    def __init__(self, kind, plain, distance, length, expanded):
        self.kind = kind              # 0 plain_byte, 1 distance_length
        self.plain = plain
        self.distance = distance
        self.length = length
        self.expanded = expanded


class Deflater(object):
    """State of one Zip.Compress.Deflate call behind the LZ77 front end."""

    def __init__(self, method, trace=None):
        self.method = method
        self.trace = trace            # optional list: ("block", first atom (global), atoms, format, bits) / ("cut", global atom, level)
        self.out = bytearray()
        self.bit_buffer = 0
        self.valid_bits = 0
        self.fixed = ((list(DEFAULT_LIT_LEN_BL), prepare_codes(DEFAULT_LIT_LEN_BL)), (list(DEFAULT_DIS_BL), prepare_codes(DEFAULT_DIS_BL)))
        self.curr = self.fixed        # curr_descr :722 -- ((bl, codes) lit_len, (bl, codes) dis)
        self.block_to_finish = False
        self.last_block_marked = False
        self.last_block_type = "reserved"
        self.lz_buffer = [None] * LZ_BUFFER_SIZE
        self.lz_buffer_index = 0
        self.past_lz_data = False
        self.flushed_atoms = 0        # atoms in front of the ring's current lap pair, for the trace only

    # ---- :101-160
    def put_byte(self, b):
        self.out.append(b & 0xFF)

    def flush_bit_buffer(self):
        while self.valid_bits > 0:
            self.put_byte(self.bit_buffer & 0xFF)
            self.bit_buffer >>= 8
            self.valid_bits = max(0, self.valid_bits - 8)
        self.bit_buffer = 0

    def put_bits(self, code, code_size):
        assert 1 <= code_size <= 15
        self.bit_buffer = (self.bit_buffer | (code << self.valid_bits)) & 0xFFFFFFFF      # Shift_Left on Unsigned_32 drops the high bits
        self.valid_bits += code_size
        if self.valid_bits > 32:
            self.put_byte(self.bit_buffer & 0xFF)
            self.put_byte((self.bit_buffer >> 8) & 0xFF)
            self.put_byte((self.bit_buffer >> 16) & 0xFF)
            self.put_byte((self.bit_buffer >> 24) & 0xFF)
            self.valid_bits -= 32
            self.bit_buffer = code >> (code_size - self.valid_bits)

    def put_huffman_code(self, table, sym):               # :523-532
        bl, codes = table
        self.put_bits(codes[sym], bl[sym])

    # ---- :725-728, :805-884
    def put_literal_byte(self, b):
        self.put_huffman_code(self.curr[0], b)

    def put_dl_code(self, distance, length):
        self.put_huffman_code(self.curr[0], CODE_FOR_LENGTH[length])
        off, eb = _len_extra(length)
        if eb > 0:
            self.put_bits((length - off) & ((1 << eb) - 1), eb)
        c, ev, eb = dist_code(distance)
        self.put_huffman_code(self.curr[1], c)
        if eb > 0:
            self.put_bits(ev, eb)

    # ---- :953-976 on the ring slice first .. last (an empty list of indices = Ada null slice)
    def get_statistics(self, idxs):
        sl = [0] * 288
        sl[END_OF_BLOCK] = 1
        sd = [0] * 32
        buf = self.lz_buffer
        for i in idxs:
            a = buf[i]
            if a.kind == 0:
                sl[a.plain] += 1
            else:
                sl[CODE_FOR_LENGTH[a.length]] += 1
                sd[dist_code(a.distance)[0]] += 1
        return sl, sd

    def put_lz_buffer(self, idxs):                        # :981-991
        buf = self.lz_buffer
        for i in idxs:
            a = buf[i]
            if a.kind == 0:
                self.put_literal_byte(a.plain)
            else:
                self.put_dl_code(a.distance, a.length)

    def mark_new_block(self, last_block_for_stream):      # :999-1007
        if self.block_to_finish and self.last_block_type in ("fixed", "dynamic"):
            self.put_huffman_code(self.curr[0], END_OF_BLOCK)
        self.block_to_finish = True
        self.put_bits(1 if last_block_for_stream else 0, 1)
        self.last_block_marked = last_block_for_stream

    def expand_lz_buffer(self, first, last, last_block):  # :1010-1062
        buf = self.lz_buffer
        to_be_sent = 0
        for i in range(first, last + 1):
            a = buf[i]
            to_be_sent += 1 if a.kind == 0 else a.length
        if to_be_sent > 0xFFFF:
            mid = (first + last) // 2
            self.expand_lz_buffer(first, mid, False)
            self.expand_lz_buffer(mid + 1, last, last_block)
            return
        b1 = to_be_sent % 256
        b2 = to_be_sent // 256
        self.mark_new_block(last_block)
        self.last_block_type = "stored"
        self.put_bits(0, 2)
        self.flush_bit_buffer()
        self.put_byte(b1)
        self.put_byte(b2)
        self.put_byte(b1 ^ 0xFF)
        self.put_byte(b2 ^ 0xFF)
        for i in range(first, last + 1):
            a = buf[i]
            if a.kind == 0:
                self.put_byte(a.plain)
            else:
                for j in range(a.length):
                    self.put_byte(a.expanded[j])

    # ---- :549-704
    def put_compression_structure(self, descr_bl, cost_analysis, codes=None):
        """descr_bl = (bl lit_len, bl dis).  cost_analysis: returns the bits; otherwise emits."""
        bl_ll, bl_d = descr_bl
        max_used_lln_code = 0
        for a in range(287, -1, -1):
            if bl_ll[a] > 0:
                max_used_lln_code = a
                break
        max_used_dis_code = 0
        for a in range(31, -1, -1):
            if bl_d[a] > 0:
                max_used_dis_code = a
                break
        cs_bl = [None]                                   # 1-based
        for a in range(0, max_used_lln_code + 1):
            cs_bl.append(bl_ll[a])
        for a in range(0, max_used_dis_code + 1):
            cs_bl.append(bl_d[a])
        last_cs_bl = len(cs_bl) - 1
        extra_bits_needed = [0] * 19
        extra_bits_needed[16] = 2
        extra_bits_needed[17] = 3
        extra_bits_needed[18] = 7
        truc_freq = [0] * 19
        truc = [None, None]                               # (bl, codes) of the local alphabet in "effective" mode

        def emit_structures(effective):
            def atom(x, extra_code=0):
                if not effective:
                    truc_freq[x] += 1
                else:
                    self.put_bits(truc[1][x], truc[0][x])
                    if extra_bits_needed[x] > 0:
                        self.put_bits(extra_code, extra_bits_needed[x])
            idx = 1
            while True:
                rep = 1
                for j in range(idx + 1, last_cs_bl + 1):
                    if cs_bl[j] != cs_bl[idx]:
                        break
                    rep += 1
                if idx > 1 and cs_bl[idx] == cs_bl[idx - 1] and rep >= 3 and not (cs_bl[idx] == 0 and rep > 6):
                    rep = min(rep, 6)
                    atom(16, rep - 3)
                    idx += rep
                elif cs_bl[idx] == 0 and rep >= 3:
                    if rep <= 10:
                        atom(17, rep - 3)
                    else:
                        rep = min(rep, 138)
                        atom(18, rep - 11)
                    idx += rep
                else:
                    atom(cs_bl[idx])
                    idx += 1
                if idx > last_cs_bl:
                    break

        alphabet_permutation = (16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15)
        emit_structures(False)
        truc_bl = length_limited_coding(truc_freq, 7)
        a_non_zero = 3
        for a in range(19):
            if a > a_non_zero and truc_bl[alphabet_permutation[a]] > 0:
                a_non_zero = a
        if cost_analysis:
            bits = 14 + (1 + a_non_zero) * 3
            for a in range(19):
                bits += truc_freq[a] * (truc_bl[a] + extra_bits_needed[a])
            return bits
        truc[0] = truc_bl
        truc[1] = prepare_codes(truc_bl, 15, True)
        self.put_bits(max_used_lln_code - 256, 5)
        self.put_bits(max_used_dis_code, 5)
        self.put_bits(a_non_zero - 3, 4)
        for a in range(0, a_non_zero + 1):
            # Put_Bits' code_size is 1 .. 15 and a 3-bit field is asked for: fine; the VALUE may be 0
            self.put_bits(truc_bl[alphabet_permutation[a]], 3)
        emit_structures(True)
        return 0

    # ---- :1105-1269
    def send_as_block(self, first, last, last_block):
        idxs = range(first, last + 1)
        buf = self.lz_buffer
        stats_lit_len, stats_dis = self.get_statistics(idxs)
        new_descr = build_descriptors_from_stats(stats_lit_len, stats_dis)
        stats_lit_len_2 = list(stats_lit_len)
        stats_dis_2 = list(stats_dis)
        tweak_for_better_rle(stats_lit_len_2)
        tweak_for_better_rle(stats_dis_2)
        new_descr_2 = build_descriptors_from_stats(stats_lit_len_2, stats_dis_2)
        stored_format_possible = all(stats_lit_len[c] == 0 for c in range(CODE_FOR_MAX_EXPAND + 1, 288))
        curr_bl = (self.curr[0][0], self.curr[1][0])
        recycling_possible = (self.last_block_type == "fixed"
                              or (self.last_block_type == "dynamic" and recyclable(curr_bl, new_descr)))
        stored_b = fixed_b = dyn_b = dyn2_b = rec_b = 0
        for i in range(0, 256):
            c = stats_lit_len[i]
            stored_b += 8 * c
            fixed_b += DEFAULT_LIT_LEN_BL[i] * c
            dyn_b += new_descr[0][i] * c
            dyn2_b += new_descr_2[0][i] * c
            rec_b += curr_bl[0][i] * c
        if stored_format_possible:
            for i in idxs:
                if buf[i].kind == 1:
                    stored_b += 8 * buf[i].length
        for i in range(257, 286):
            c = stats_lit_len[i]
            extra = EXTRA_LEN_CODE[i]
            fixed_b += (DEFAULT_LIT_LEN_BL[i] + extra) * c
            dyn_b += (new_descr[0][i] + extra) * c
            dyn2_b += (new_descr_2[0][i] + extra) * c
            rec_b += (curr_bl[0][i] + extra) * c
        for i in range(0, 30):
            c = stats_dis[i]
            extra = EXTRA_DIS_CODE[i]
            fixed_b += (DEFAULT_DIS_BL[i] + extra) * c
            dyn_b += (new_descr[1][i] + extra) * c
            dyn2_b += (new_descr_2[1][i] + extra) * c
            rec_b += (curr_bl[1][i] + extra) * c
        stored_b += (1 + (stored_b // 8) // 65535) * 5 * 8
        c = 1
        if self.block_to_finish and self.last_block_type in ("fixed", "dynamic"):
            c += curr_bl[0][END_OF_BLOCK]
        stored_b += c
        fixed_b += c + 2
        dyn_b += c + 2
        dyn2_b += c + 2
        dyn_b += self.put_compression_structure(new_descr, True)
        dyn2_b += self.put_compression_structure(new_descr_2, True)
        if not stored_format_possible:
            stored_b = BIG
        if not recycling_possible:
            rec_b = BIG
        optimal = min(min(stored_b, fixed_b), min(min(dyn_b, dyn2_b), rec_b))
        if fixed_b == optimal:
            fmt = "fixed"
            if self.last_block_type == "fixed":
                pass
            else:
                self.mark_new_block(last_block)
                self.curr = self.fixed
                self.put_bits(1, 2)
                self.last_block_type = "fixed"
            self.put_lz_buffer(idxs)
        elif dyn_b == optimal or dyn2_b == optimal:
            fmt = "dynamic" if dyn_b == optimal else "dynamic_rle"
            dyn = new_descr if dyn_b == optimal else new_descr_2
            self.mark_new_block(last_block)
            self.curr = ((dyn[0], prepare_codes(dyn[0])), (dyn[1], prepare_codes(dyn[1])))
            self.put_bits(2, 2)
            self.put_compression_structure(dyn, False)
            self.put_lz_buffer(idxs)
            self.last_block_type = "dynamic"
        elif rec_b == optimal:
            fmt = "recycled"
            self.put_lz_buffer(idxs)
        else:
            fmt = "stored"
            self.expand_lz_buffer(first, last, last_block)
        if self.trace is not None:
            self.trace.append(("block", self._global(first), last - first + 1, fmt, optimal))

    def _global(self, ring_index):
        return self.flushed_base + ring_index

    # ---- :1327-1408
    def scan_and_send_from_main_buffer(self, frm, to, last_flush):
        if ((to - frm) & RING) < SLIDER_MAX:
            self.send_as_block(frm, to, last_flush)
            return
        if self.past_lz_data:
            start = (frm - HALF_SLIDER) & RING
        else:
            start = frm
        if start > frm:
            copy = []
            copy_from = start
            for _ in range(SLIDER_MAX + 1):
                copy.append(copy_from)
                copy_from = (copy_from + 1) & RING
            initial_hd = build_descriptors_from_stats(*self.get_statistics(copy))
        else:
            initial_hd = build_descriptors_from_stats(*self.get_statistics(self._slice(start, (start + SLIDER_MAX) & RING)))
        send_from = frm
        slide_mid = (frm + MIN_STEP) & RING
        while slide_mid + HALF_SLIDER < to:
            sliding_hd = None
            for level in (1, 2, 3):
                if level > MAX_CHOICE[self.method]:
                    break
                step, threshold = STEP_CHOICE[level - 1]
                if ((slide_mid - frm) & RING) % step == 0:
                    if sliding_hd is None:
                        sliding_hd = build_descriptors_from_stats(
                            *self.get_statistics(self._slice((slide_mid - HALF_SLIDER) & RING, (slide_mid + HALF_SLIDER) & RING)))
                    sim, dist, thres = similar(initial_hd, sliding_hd, threshold)
                    if self.trace is not None:
                        self.trace.append(("similar", self._global(slide_mid), dist, thres, sim))
                    if not sim:
                        if self.trace is not None:
                            self.trace.append(("cut", self._global(slide_mid), level))
                        self.send_as_block(send_from, (slide_mid - 1) & RING, False)
                        send_from = slide_mid
                        initial_hd = sliding_hd
                        break
            if slide_mid + MIN_STEP + HALF_SLIDER >= to:
                break
            slide_mid = (slide_mid + MIN_STEP) & RING
        if send_from <= to:
            self.send_as_block(send_from, to, last_flush)

    @staticmethod
    def _slice(lo, hi):
        """Ada slice lo .. hi of an array indexed by a modular type: null when lo > hi."""
        return range(lo, hi + 1) if lo <= hi else range(0)

    def flush_half_buffer(self, last_flush):              # :1410-1422
        last_idx = (self.lz_buffer_index - 1) & RING
        n_div_2 = LZ_BUFFER_SIZE // 2
        # global index of ring slot 0 for the trace: completed laps of the ring
        self.flushed_base = (self.pushed - 1) // LZ_BUFFER_SIZE * LZ_BUFFER_SIZE
        if last_idx < n_div_2:
            self.scan_and_send_from_main_buffer(0, last_idx, last_flush)
        else:
            self.scan_and_send_from_main_buffer(n_div_2, last_idx, last_flush)
        self.past_lz_data = True

    pushed = 0
    flushed_base = 0

    def push(self, a):                                    # :1424-1432
        self.lz_buffer[self.lz_buffer_index] = a
        self.lz_buffer_index = (self.lz_buffer_index + 1) & RING
        self.pushed += 1
        if (self.lz_buffer_index * 2) & RING == 0:
            self.flush_half_buffer(False)

    # ---- :1434-1454
    def literal(self, b):
        if self.method == 6:
            self.put_literal_byte(b)
        else:
            self.push(Atom(0, b, 0, 0, None))

    def dl_code(self, distance, length, expand):
        if self.method == 6:
            self.put_dl_code(distance, length)
        else:
            self.push(Atom(1, 0, distance, length, expand))

    # ---- :1593-1606
    def begin(self):
        if self.method == 6:
            self.put_bits(1, 1)
            self.put_bits(1, 2)

    # ---- :1613-1635 and the body's :1665
    def finish(self):
        if self.method == 6:
            self.put_huffman_code(self.curr[0], END_OF_BLOCK)
        else:
            if (self.lz_buffer_index * 2) & RING == 0:
                if self.block_to_finish and self.last_block_type in ("fixed", "dynamic"):
                    self.put_huffman_code(self.curr[0], END_OF_BLOCK)
            else:
                self.flush_half_buffer(True)
                if self.last_block_type in ("fixed", "dynamic"):
                    self.put_huffman_code(self.curr[0], END_OF_BLOCK)
            if not self.last_block_marked:
                self.put_bits(1, 1)
                self.put_bits(1, 2)
                self.curr = self.fixed
                self.put_huffman_code(self.curr[0], END_OF_BLOCK)
        self.flush_bit_buffer()
        return bytes(self.out)


def deflate_from_tokens(data, tokens, method, trace=None):
    """data: the input bytes; tokens: the LZ77 stage's output as the repo's 32-bit words
    (bit 31 set: match, length in bits 16..24, distance in bits 0..15; otherwise a literal byte).
    The first 14 bytes of a match (lz_expanded, zip-compress-deflate.adb:1531-1539) are what the text
    buffer holds there, i.e. the input's own bytes at the match's position."""
    d = Deflater(method, trace)
    d.begin()
    pos = 0
    for t in tokens:
        t = int(t)
        if t & 0x80000000:
            length = (t >> 16) & 0x1FF
            distance = t & 0xFFFF
            d.dl_code(distance, length, data[pos:pos + min(length, MAX_EXPAND)])
            pos += length
        else:
            d.literal(t & 0xFF)
            pos += 1
    assert pos == len(data), (pos, len(data))
    return d.finish()
