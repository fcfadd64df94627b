// zada_bt4.h -- the BT4 match finder of LZMA Level_3 (zip_lib/lz77.adb:953-1827, LZ77_using_BT4) restated as a PRODUCER of match sets
// that runs ahead of the coder, one binary tree (= one hash-4 bucket) per lane.  Host + device inline functions: on the GPU they run
// inside the kernels of zada_bt4.hip; the same text is compiled for the host ONLY by tests/hostcheck (a test library) so that the
// reformulation can be checked against the oracle's sequential BT4 on a machine without a GPU.
//
// Why the match sets are a function of the input alone.  BT4_Algo.Skip (:1208-1232) and BT4_Algo.Read_One_and_Get_Matches
// (:1234-1361) update the hash heads and the tree in the same way: Skip's Skip_and_Update_Tree (:1154-1206) stops when a candidate
// shares niceLenLimit bytes; Read_One stops at `len >= niceLenLimit` (:1340), and there lenBest < niceLenLimit always holds (a hash
// match of nice length leaves through Skip_and_Update_Tree at :1297), so `len > lenBest` (:1335) is true whenever the stop test is met.
// Below that length both compute the same len, compare the same bytes and descend the same way.  Hence the tree after position p --
// and with it every set Read_One would return -- does not depend on which positions the coder reads and which it skips, and:
//   * a tree only ever links positions of ONE hash-4 bucket (the root is hash4Table [h4], children are candidates met on the way
//     down), so the buckets are independent of each other: one lane walks the positions of its bucket in increasing order;
//   * hash2Table / hash3Table [h] = the last inserted position with that hash = the predecessor in a stable sort by the hash.
// The reference's tree is a ring of cyclicSize nodes; a node is overwritten cyclicSize insertions later, when no walk can reach it
// any more (delta0 >= max_dist = cyclicSize - 275 ends a walk first, :1166, 1311), so the producer keeps one node per inserted
// position and no ring -- which is what makes the buckets independent in TIME as well.
//
// What does depend on the reader is WHEN the window is filled (Fill_Window :1389-1440 runs when Get_Available = 0, :1817-1826):
// positions closer than Nice_Length = 162 to the end of the filled window are not inserted when they are first visited (Move_Pos
// :1000-1017 with finishing = False, :959: "pending"), the next fill inserts them through Skip (processPendingBytes :1397-1406) -- if it
// adds enough bytes -- and the last ones of a stream never are.  The fills happen at fixed positions (the reader reaches the last
// filled byte before every fill), so the host replays them (bt4_schedule) into a table of RUNS of positions:
//   cls 0: read with `W - q - 1` bytes available, inserted at once;   cls 1: pending, inserted by the next fill (no match set);
//   cls 2: never inserted (they take no lzPos either: distances are counted in INSERTED positions, `ord`).
#pragma once
#include <stdint.h>
#if !defined(__HIPCC__)
#include <vector>
#endif

#if defined(__HIPCC__)
#define ZADA_BT_HD __host__ __device__ __forceinline__
#else
#define ZADA_BT_HD inline
#endif

namespace zada {

constexpr int BT4_LOOK = 273, BT4_NICE = 162, BT4_DEPTH = 48, BT4_OPTS = 4096;
constexpr int BT4_SET = 50;                         // most matches of one position: one per hash (2, 3 bytes) + one per tree step (Depth_Limit)
constexpr int32_t BT4_NONE = -1;
// Match sets in HBM: BT4_INLINE slots per position -- BT4_INLINE - 1 matches and, for a longer set, the number of its block in the overflow
// pool (in the last slot's distance), BT4_OVF more matches there.
constexpr int BT4_INLINE = 8, BT4_OVF = BT4_SET - (BT4_INLINE - 1);

struct Bt4Run { uint32_t start, end, W, gap, cls, pad; };   // positions [start, end) of the entry; ord = q - gap; available at the first visit = W - q - 1

// One entry (= one LZMA stream) as the producer sees it.
struct Bt4Job {
  uint64_t in_off;                                   // arena position of the entry's first byte (a multiple of 64)
  uint32_t n, sbs, hash4_mask, max_dist;
  uint32_t run_off, run_cnt;                         // its runs in the run table
  uint32_t sorted_off;                               // its inserted positions in the (entry, hash 4) order start here
  uint32_t small;                                    // 1: one window fill and at most BT4_LDS_N bytes -- its trees are built in LDS
};
constexpr uint32_t BT4_LDS_N = 16384;                // tree (2 x 2 bytes per inserted position; the last 162 never are) + text = 80 KB: two entries per CU

// String_buffer_size of Level_3 (lzma-encoding.adb:137-149) and BT4's hash-4 table size (lz77.adb:1019-1032)
ZADA_BT_HD uint32_t bt4_string_buffer_size(uint64_t dictionary_size) {
  uint64_t x = dictionary_size + 273 + 1 + 64, p = 1;
  while (p < 0x7FFFFFFFull / 2 && p < x) p *= 2;
  if (p < x) p = x;
  if (p > (1ull << 28)) p = 1ull << 28;
  if (p < 4096) p = 4096;
  return (uint32_t)p;
}
ZADA_BT_HD uint32_t bt4_hash4_size(uint32_t sbs) {
  uint32_t h = sbs - 1;
  h |= h >> 1; h |= h >> 2; h |= h >> 4; h |= h >> 8;
  h >>= 1;
  h |= 0xFFFF;
  if (h > (1u << 24)) h >>= 1;
  return h + 1;
}

ZADA_BT_HD uint32_t bt4_crc(uint32_t i) {          // Hash234.crcTable :1091-1101
  uint32_t r = i;
  for (int j = 0; j < 8; j++) r = (r & 1) ? (r >> 1) ^ 0xEDB88320u : r >> 1;
  return r;
}
// calcHashes :1061-1069 on the four bytes at b
ZADA_BT_HD void bt4_hashes(uint32_t c0, uint32_t b1, uint32_t b2, uint32_t c3, uint32_t mask, uint32_t &h2, uint32_t &h3, uint32_t &h4) {
  uint32_t t = c0 ^ b1;                              // c0 = crcTable [b0], c3 = crcTable [b3]
  h2 = t & 1023;
  t ^= b2 << 8;
  h3 = t & 65535;
  t ^= c3 << 5;
  h4 = t & mask;
}

// The run of the entry's position q (runs are sorted, contiguous from 0 to n).
ZADA_BT_HD const Bt4Run *bt4_run_of(const Bt4Run *runs, uint32_t cnt, uint32_t q) {
  uint32_t lo = 0, hi = cnt;
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (runs[mid].start <= q) lo = mid; else hi = mid; }
  return runs + lo;
}

// First index k in [len, limit) with in [a + k] /= in [b + k], or limit (the byte loops of :1183-1191, 1287-1291, 1331-1335).
ZADA_BT_HD int bt4_extend(const uint8_t *in, int64_t a, int64_t b, int len, int limit) {
  while (len < limit && in[a + len] == in[b + len]) len++;
  return len;
}

// The visit of ONE inserted position by its bucket's lane: Read_One_and_Get_Matches (record: cls 0) or Skip (cls 1) at the entry's
// position q = `in + q`, whose ordinal among the inserted positions is ordp.  root = ordinal of the bucket's previous position
// (hash4Table [h4]), o2 / o3 = ordinals of the previous inserted positions with the same hash2 / hash3; BT4_NONE where there is none.
// tree: the entry's nodes, two ints per ordinal (left / right child as ordinals).  A candidate of ordinal c is met at distance
// delta = ordp - c and its bytes are those at q - delta (the reference addresses buf [readPos - delta], :1285, 1318).
// In two parts, so that a lane can take its next position while its neighbours are still on their way down: bt4_begin (the two hash
// matches, :1263-1305) and bt4_step (one level of the tree, :1309-1360 / :1164-1205; true = the visit is over).  `put (i, len, dist)`
// receives match i (lengths strictly increasing); `extend (in, a, b, len, limit)` is bt4_extend or a faster form of it.
// The nodes of an entry's trees: two children per inserted position, as ordinals (BT4_NONE = no child).  In HBM they are ints; an
// entry small enough for LDS (zada_bt4.hip, k_bt4_walk_lds) keeps them as 16-bit values there.
struct Bt4TreeI32 {
  int32_t *t;
  ZADA_BT_HD int32_t get(uint32_t i) const { return t[i]; }
  ZADA_BT_HD void set(uint32_t i, int32_t v) const { t[i] = v; }
};
struct Bt4TreeU16 {
  uint16_t *t;
  ZADA_BT_HD int32_t get(uint32_t i) const { const uint16_t v = t[i]; return v == 0xFFFFu ? BT4_NONE : (int32_t)v; }
  ZADA_BT_HD void set(uint32_t i, int32_t v) const { t[i] = (uint16_t)v; }          // (BT4_NONE = -1 -> 0xFFFF)
};

struct Bt4Walk {
  const uint8_t *in; int64_t qq;
  int32_t ordp, max_dist, cur;
  uint32_t ptr0, ptr1;                               // (2 * ordinal + 1 needs all 32 bits for an entry of 1 GiB)
  int limit, depth, len0, len1, lenBest, count;
  bool record;
};
template <typename Ext, typename Put>
ZADA_BT_HD void bt4_begin(Bt4Walk &w, const uint8_t *in, uint32_t q, int32_t ordp, bool record, int matchLenLimit, int32_t max_dist, int32_t root, int32_t o2, int32_t o3,
                          Ext extend, Put put) {
  constexpr int32_t FAR = 0x7FFFFFFF;
  w.in = in; w.qq = (int64_t)q; w.ordp = ordp; w.max_dist = max_dist; w.cur = root; w.limit = matchLenLimit; w.record = record;
  w.count = 0; w.lenBest = 0;
  if (record) {
    const int32_t delta2 = o2 >= 0 ? ordp - o2 : FAR, delta3 = o3 >= 0 ? ordp - o3 : FAR;
    const bool m2 = delta2 < max_dist && in[w.qq - delta2] == in[w.qq];                                       // :1263-1270
    const bool m3 = delta2 != delta3 && delta3 < max_dist && in[w.qq - delta3] == in[w.qq];                   // :1275-1282
    if (m2 || m3) {
      const int32_t d = m3 ? delta3 : delta2;
      w.lenBest = extend(in, w.qq - d, w.qq, m3 ? 3 : 2, matchLenLimit);                                       // :1285-1292
      if (m2 && m3) put(0, 2, (uint32_t)delta2);
      put(m2 && m3 ? 1 : 0, w.lenBest, (uint32_t)d);
      w.count = m2 && m3 ? 2 : 1;
      if (w.lenBest >= BT4_NICE) w.record = false;   // :1294-1299: the tree is updated as Skip does it, no more matches
    }
    if (w.lenBest < 3) w.lenBest = 3;                // :1303-1305
  }
  w.depth = BT4_DEPTH; w.ptr0 = 2u * (uint32_t)ordp + 1u; w.ptr1 = 2u * (uint32_t)ordp; w.len0 = 0; w.len1 = 0;
}
template <typename Tree, typename Ext, typename Put>
ZADA_BT_HD bool bt4_step(Bt4Walk &w, const Tree tree, Ext extend, Put put) {
  constexpr int32_t FAR = 0x7FFFFFFF;
  const int nice = BT4_NICE;                         // niceLenLimit = min (Nice_Length, avail) = 162 for every inserted position
  const int32_t cur = w.cur, delta0 = cur >= 0 ? w.ordp - cur : FAR;
  if (w.depth == 0 || delta0 >= w.max_dist) { tree.set(w.ptr0, BT4_NONE); tree.set(w.ptr1, BT4_NONE); return true; }                          // :1166-1170, 1311-1315
  w.depth--;
  const uint32_t pair = 2u * (uint32_t)cur;
  // the candidate's two children now, next to its bytes: nothing below writes them (the stores go to slots of nodes met EARLIER on the
  // way down, or of the position itself), and loaded here they do not wait for the byte comparison
  const int32_t child0 = tree.get(pair), child1 = tree.get(pair + 1);
  const uint8_t *in = w.in;
  const int64_t qq = w.qq;
  int len = w.len0 < w.len1 ? w.len0 : w.len1;
  if (w.record) {
    if (in[qq + len - delta0] == in[qq + len]) {
      len = extend(in, qq - delta0, qq, len + 1, w.limit);
      if (len > w.lenBest) {
        w.lenBest = len;
        put(w.count, len, (uint32_t)delta0); w.count++;
        if (len >= nice) { tree.set(w.ptr1, child0); tree.set(w.ptr0, child1); return true; }                                                   // :1340-1345
      }
    }
  } else {
    len = extend(in, qq - delta0, qq, len, nice);
    if (len == nice) { tree.set(w.ptr1, child0); tree.set(w.ptr0, child1); return true; }                                                       // :1185-1189
  }
  if (in[qq + len - delta0] < in[qq + len]) { tree.set(w.ptr1, cur); w.ptr1 = pair + 1; w.cur = child1; w.len1 = len; }                     // :1195-1205, 1349-1359
  else { tree.set(w.ptr0, cur); w.ptr0 = pair; w.cur = child0; w.len0 = len; }
  return false;
}

// The fills of LZ77_using_BT4's main loop (:1798-1827) replayed for an entry of n bytes: Fill_Window (:1389-1440) with Move_Window
// (:1375-1386) and processPendingBytes (:1397-1406).  Writes the entry's runs (see the head of the file); returns false when a fill
// would re-insert positions that are in the tree already (pending bytes left over by an earlier fill AND taken up by a later one:
// cannot happen -- a fill that does not reach EOF adds min (sbs, room) > keepSizeAfter bytes unless sbs = 4096, which never catches up).
template <typename Vec> inline bool bt4_schedule(uint64_t n, uint32_t sbs, Vec &runs) {
  const int64_t keepBefore = BT4_OPTS + (int64_t)sbs, keepAfter = BT4_OPTS + BT4_LOOK;
  const int64_t r0 = (int64_t)sbs / 2 + 256 * 1024, rmax = 512ll << 20;
  const int64_t buf_len = keepBefore + keepAfter + (r0 < rmax ? r0 : rmax) + 1;
  int64_t readPos = -1, readLimit = -1, writePos = 0, pending = 0, moved = 0;
  uint64_t in_pos = 0;
  bool reproc = false;
  int64_t old_pending = 0;
  auto fill = [&]() -> int64_t {
    reproc = false;
    int64_t len = sbs;
    if (readPos >= buf_len - keepAfter) {
      const int64_t off = ((readPos + 1 - keepBefore) / 16) * 16;
      moved += off; readPos -= off; readLimit -= off; writePos -= off;
    }
    if (len > buf_len - writePos) len = buf_len - writePos;
    const int64_t actual = (uint64_t)len < n - in_pos ? len : (int64_t)(n - in_pos);
    writePos += actual; in_pos += (uint64_t)actual;
    if (writePos >= keepAfter) readLimit = writePos - keepAfter;
    if (pending > 0 && readPos < readLimit) { old_pending = pending; pending = 0; reproc = true; }
    return actual;
  };
  runs.clear();
  if (n == 0 || fill() == 0) return true;
  uint64_t Wprev = 0, gap = 0;
  for (;;) {
    const uint64_t W = (uint64_t)(writePos + moved);
    const uint64_t pstart = W >= (uint64_t)BT4_NICE && W - BT4_NICE > Wprev ? W - BT4_NICE : Wprev;     // avail = W - q - 1 < 162  <=>  q >= W - 162
    if (pstart > Wprev) runs.push_back(Bt4Run{(uint32_t)Wprev, (uint32_t)pstart, (uint32_t)W, (uint32_t)gap, 0u, 0u});
    pending += (int64_t)(W - pstart);
    readPos = writePos - 1;                          // the reader reaches the last filled byte, then Fill_Window runs (:1817-1826)
    const int64_t actual = fill();
    if (reproc) {
      if ((uint64_t)old_pending != W - pstart) return false;
      runs.push_back(Bt4Run{(uint32_t)pstart, (uint32_t)W, (uint32_t)(writePos + moved), (uint32_t)gap, 1u, 0u});
    } else {
      runs.push_back(Bt4Run{(uint32_t)pstart, (uint32_t)W, (uint32_t)W, (uint32_t)gap, 2u, 0u});
      gap += W - pstart;
    }
    Wprev = W;
    if (actual == 0) break;
  }
  return true;
}

// True when positions are read and inserted while lzPos lags behind readPos -- after pending bytes that no fill took up (a fill of at most
// keepSizeAfter - 1 = 4 368 bytes, i.e. the last one of a stream, or every one at String_buffer_size 4096; the reference notes the lag at
// lz77.adb:1010-1012).  From there on a distance taken from the hash tables or the tree into the positions before the gap is short by the gap:
// the reference compares the bytes at THAT distance, and its hash-2 / hash-3 matches compare only the first byte (:1262-1290, "the hashing
// algorithm guarantees ...": not across a gap) -- it can report, and code, matches that are none, and its stream then decodes to something
// else than the input.  With the dictionary Zip.Compress.LZMA_E asks for (the entry's size) the whole entry arrives in the first fill and
// nothing is read behind a gap; an entry beyond 256 MiB whose last fill brings 163 .. 4 368 bytes is (LZMA.Encoding's default of 32 KiB on
// longer data as well).  zada_lzma verifies the match sets of such an entry as it reads them and refuses it (ZADA_E_REFERENCE) on the first
// match that is none.
template <typename Vec> inline bool bt4_reads_behind_a_gap(const Vec &runs) {
  for (const Bt4Run &r : runs) if (r.cls == 0 && r.gap > 0 && r.end > r.start) return true;
  return false;
}

}  // namespace zada
