// zada_logic.h -- sequential building blocks of the encoder, written once as
// host+device inline functions.  On the GPU they run inside the kernels of zada_lz.hip /
// zada_huff.hip (one lane or one wave per instance, state in LDS); the same text is
// compiled for the host ONLY by tests/hostcheck (a test library), never by the product
// path, so that their arithmetic can be checked on a machine without a GPU.
//
// Reference semantics cited per function (paths relative to the reference's zip_lib/).
#pragma once
#include <type_traits>
#include <stdint.h>

#if defined(__HIPCC__)
#define ZADA_HD __host__ __device__ __forceinline__
#else
#define ZADA_HD inline
#endif

namespace zada {

// ----- token format (product-internal; equals the oracle's test format by construction) -----
constexpr uint32_t TOK_MATCH = 0x80000000u;
ZADA_HD uint32_t tok_match(uint32_t len, uint32_t dist) { return TOK_MATCH | (len << 16) | dist; }
ZADA_HD bool tok_is_match(uint32_t t) { return (t & TOK_MATCH) != 0; }
ZADA_HD uint32_t tok_len(uint32_t t) { return tok_is_match(t) ? ((t >> 16) & 0x1FF) : 1u; }
ZADA_HD uint32_t tok_dist(uint32_t t) { return t & 0xFFFF; }

// ----- LZ77 constants, lz77.adb:461-501 -----
constexpr int MIN_MATCH = 3, MAX_MATCH = 258, MAX_DIST = 32506, TOO_FAR = 4096;
constexpr int WSIZE = 32768;

struct LzConfig { int good, lazy, nice, chain; };   // lz77.adb:534-546
ZADA_HD LzConfig lz_config(int level) {
  switch (level) {
    case 4: return {4, 4, 16, 16};
    case 5: return {8, 16, 32, 32};
    case 6: return {8, 16, 128, 128};
    case 7: return {8, 32, 128, 256};
    case 8: return {32, 128, 258, 1024};
    case 9: return {32, 258, 258, 4096};
    default: return {34, 258, 258, 4096};   // 10
  }
}

ZADA_HD int clz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __clz((int)x);
#else
  return x ? __builtin_clz(x) : 32;
#endif
}

// ----- Deflate symbol tables, zip-compress-deflate.adb:757-803, 886-918, 1066-1091 -----
ZADA_HD int len_symbol(int len) {            // deflate_code_for_lz_length
  if (len == 258) return 285;
  int l = len - 3;
  if (l < 8) return 257 + l;
  int k = 31 - clz32((uint32_t)l);
  return 257 + 4 * (k - 1) + ((l >> (k - 2)) & 3);
}
ZADA_HD int len_extra_bits(int len) {        // extra_bits_for_lz_length
  if (len == 258) return 0;
  int l = len - 3;
  if (l < 8) return 0;
  return 31 - clz32((uint32_t)l) - 2;
}
ZADA_HD uint32_t len_extra_val(int len) { int e = len_extra_bits(len); return (uint32_t)(len - 3) & ((1u << e) - 1); }
ZADA_HD int dist_symbol(int dist) {          // Deflate_code_for_LZ_distance
  if (dist <= 4) return dist - 1;
  uint32_t d = (uint32_t)dist - 1;
  int k = 31 - clz32(d);
  return 2 * k + (int)((d >> (k - 1)) & 1);
}
ZADA_HD int dist_extra_bits(int dist) {
  if (dist <= 4) return 0;
  return 31 - clz32((uint32_t)dist - 1) - 1;
}
ZADA_HD uint32_t dist_extra_val(int dist) { int e = dist_extra_bits(dist); return ((uint32_t)dist - 1) & ((1u << e) - 1); }
ZADA_HD int litlen_sym_extra(int sym) { return (sym < 265 || sym == 285) ? 0 : (sym - 261) / 4; }   // :1066-1074
ZADA_HD int dist_sym_extra(int sym) { return sym < 4 ? 0 : (sym >> 1) - 1; }                          // :1076-1091
ZADA_HD int fixed_litlen_bl(int i) { return i <= 143 ? 8 : i <= 255 ? 9 : i <= 279 ? 7 : 8; }        // :709-715

// ----- lazy-evaluation parser, lz77.adb:827-933, over precomputed per-position match tables -----
// MF[p] / MQ[p] hold, for position p, the result of Longest_Match over the full / quartered
// hash chain: (len << 16) | dist with len >= 3, or 0 when there is no candidate.  A parser state
// at the top of the loop is (p, avail, mlen, mdist); "F" = fresh (avail = 0), "L" = literal
// pending without a match (avail = 1, mlen = 2).  Both F and L are history-free.
struct ParseState { uint32_t p; uint32_t avail; uint32_t mlen; uint32_t mdist; };

enum { SYNC_NONE = 0, SYNC_F = 1, SYNC_L = 2 };
ZADA_HD int sync_kind(const ParseState &s) { return !s.avail ? SYNC_F : (s.mlen == 2 ? SYNC_L : SYNC_NONE); }

// One loop iteration (lz77.adb:838-926).  Returns the token emitted (0xFFFFFFFF when none).
// `la` = n - p > 0.  mfq = MF[p] or MQ[p] as selected by the caller through need_quarter().
ZADA_HD bool parse_need_quarter(const ParseState &s, const LzConfig &c) { return (int)s.mlen >= c.good; }
ZADA_HD bool parse_searches(const ParseState &s, const LzConfig &c, uint64_t la) { return la >= 3 && (int)s.mlen < c.lazy; }

ZADA_HD uint32_t parse_step(ParseState &s, uint32_t mfq, bool searched, uint32_t byte_before) {
  const uint32_t L0 = s.mlen, pd = s.mdist;
  bool improved = false;
  uint32_t ml = 2, md = s.mdist;
  if (searched) {
    uint32_t len = mfq >> 16, dist = mfq & 0xFFFF;
    if (len > L0) { improved = true; ml = len; md = dist; if (ml == 3 && dist > (uint32_t)TOO_FAR) ml = 2; }
  }
  if (L0 >= 3 && !improved) {                      // :875-899  emit the previous match
    uint32_t t = tok_match(L0, pd);
    s.p += L0 - 1; s.avail = 0; s.mlen = 2; s.mdist = 0;
    return t;
  }
  s.mlen = ml; s.mdist = md; s.p += 1;
  if (s.avail) return byte_before;                  // :900-910  literal window(strstart-1)
  s.avail = 1;                                      // :911-915
  return 0xFFFFFFFFu;
}


// ----- chunked speculative parse + splice (run by one GPU lane per chunk) -----
// Longest_Match of one position over the full chain and its snapshot at the quarter-chain limit, (len << 16) | dist
// each; kept together so that the parser's scattered look-ups touch one 8-byte record instead of two arrays.
struct MatchPair { uint32_t full, quarter; };
// Flags in MatchPair::full (the value uses bits 0..24).  The match kernel first gives every position a bounded
// number of chain steps; a search cut short leaves its best-so-far as a GUESS.  A parse that lands on a guess uses
// it and marks it DEMANDed; demanded positions are then searched to the end, the parses that used a value which
// changed are redone, and so on until a parse has used exact values only (zada_lz.hip, lz_stage).
constexpr uint32_t M_GUESS = 0x80000000u, M_DEMAND = 0x40000000u, M_BYSPEC = 0x20000000u, M_HAVEQ = 0x10000000u, M_VALUE = 0x01FFFFFFu;   // M_BYSPEC: demanded by a speculative parse
// A guess record carries, in the seven bits above its quarter value, the smallest "length to beat" any speculative parse had when it landed on it
// (parse_step only asks whether the record's length exceeds the state's match length; 127 stands for "127 or more"; zada_lz.hip, DemandMarker / store_result).
constexpr uint32_t M_BEAT_SHIFT = 25, M_BEAT_MAX = 127;
struct NoGuess { ZADA_HD void operator()(uint32_t, uint32_t, uint32_t, uint32_t) const {} };
struct ParseIO { const uint8_t *in; uint64_t n; const MatchPair *M; LzConfig cfg; const uint32_t *segend; };   // segend: see Layout
// Where the entries of a batch lie in the LZ buffer (csrc/zada_lz.hip "Layout of the LZ buffer"); segend = nullptr: one stream [0, n)
struct Layout { const uint32_t *segend; uint64_t n; };
struct ExitState { uint32_t pos, kind; };

// Writes the bits of one chunk's words [first, last] exactly once (zeros where nothing is set).
struct BitmapWriter {
  uint32_t *bits; uint64_t word; uint32_t acc;
  ZADA_HD void init(uint32_t *b, uint64_t first) { bits = b; acc = 0; word = first; }
  ZADA_HD void set(uint64_t p) {
    uint64_t w = p >> 5;
    while (word < w) { bits[word] = acc; acc = 0; word++; }
    acc |= 1u << (p & 31);
  }
  ZADA_HD void finish(uint64_t last) {
    while (word < last) { bits[word] = acc; acc = 0; word++; }
    bits[word] = acc;
  }
};

// Runs the reference parser (lz77.adb:838-932) from state `s` until on_top(s) returns true at the
// top of the loop, or the input ends (then the trailing literal :930-932 is emitted).
// Fetch: how a match record is read (the GPU parser keeps the 64-byte line of its last look-up in LDS).
struct DirectFetch { const MatchPair *M; ZADA_HD MatchPair operator()(uint32_t p) const { return M[p]; } };
template <typename F> struct has_byte { template <typename U> static auto test(int) -> decltype(((U *)0)->byte(0u), char()); template <typename U> static long test(...); static const bool value = sizeof(test<F>(0)) == 1; };
template <typename F> ZADA_HD typename std::enable_if<has_byte<typename std::remove_reference<F>::type>::value, uint32_t>::type fetch_byte(F &f, const uint8_t *, uint32_t p) { return f.byte(p); }
template <typename F> ZADA_HD typename std::enable_if<!has_byte<typename std::remove_reference<F>::type>::value, uint32_t>::type fetch_byte(F &, const uint8_t *in, uint32_t p) { return in[p]; }
// Sink: where the tokens go (the GPU's speculative parse collects eight of them in LDS before it writes).
struct DirectSink { uint32_t *tok; uint32_t &ntok; ZADA_HD void push(uint32_t t) { tok[ntok++] = t; } };
// ... with room for `cap` tokens: what does not fit is counted, not written, and *overflow says so (the splice's tokens get a small slot per chunk --
// a splice meets the speculative parse after a few tokens -- and the whole slot only when one ever does not: csrc/zada_lz.hip, lz_shard)
struct CappedSink { uint32_t *tok; uint32_t &ntok; uint32_t cap; uint32_t *overflow; ZADA_HD void push(uint32_t t) { if (ntok < cap) tok[ntok] = t; else if (overflow) *overflow = 1u; ntok++; } };
template <typename Sink, typename OnTop, typename OnGuess, typename Fetch>
ZADA_HD void run_parser(ParseState &s, const ParseIO &io, Sink &&sink, OnTop &&on_top, OnGuess &&on_guess, Fetch &&fetch) {
  for (;;) {
    if ((uint64_t)s.p >= io.n) {
      if (s.avail) { sink.push(io.in[io.n - 1]); s.avail = 0; s.mlen = 2; }
      return;
    }
    if (on_top(s)) return;
    const uint64_t la = io.n - s.p;
    const bool srch = parse_searches(s, io.cfg, la);
    uint32_t m = 0;
    if (srch) {
      const MatchPair mm = fetch(s.p);
      if (mm.full & M_GUESS) on_guess(s.p, mm.full, mm.quarter, s.mlen);
      m = (parse_need_quarter(s, io.cfg) ? mm.quarter : mm.full) & M_VALUE;
    }
    const uint32_t bb = s.avail ? fetch_byte(fetch, io.in, s.p - 1) : 0;
    const uint32_t t = parse_step(s, m, srch, bb);
    if (t != 0xFFFFFFFFu) sink.push(t);
  }
}

// Speculative parse of chunk k, started in the fresh state at its first byte.  Records every
// history-free state inside the chunk (F / L bitmaps) and stops at the first one at or beyond the
// chunk's end (the chunk's exit); exit = (n, F) when the input ends first.
template <typename Sink, typename OnGuess, typename Fetch>
ZADA_HD void parse_spec_chunk_to(const ParseIO &io, uint32_t k, uint32_t chunk, Sink &&sink,
                                 uint32_t *Fbits, uint32_t *Lbits, ExitState &ex, OnGuess on_guess, Fetch &&fetch) {
  const uint64_t c0 = (uint64_t)k * chunk, c1 = (c0 + chunk < io.n) ? c0 + chunk : io.n;
  ParseState s{(uint32_t)c0, 0, 2, 0};
  BitmapWriter fw, lw;
  fw.init(Fbits, c0 >> 5); lw.init(Lbits, c0 >> 5);
  ExitState e; e.pos = (uint32_t)io.n; e.kind = SYNC_F;
  run_parser(s, io, sink, [&](const ParseState &st) {
    int kind = sync_kind(st);
    if (kind == SYNC_NONE) return false;
    if ((uint64_t)st.p >= c1) { e.pos = st.p; e.kind = (uint32_t)kind; return true; }
    if (kind == SYNC_F) fw.set(st.p); else lw.set(st.p);
    return false;
  }, on_guess, fetch);
  const uint64_t lastw = c1 > c0 ? (c1 - 1) >> 5 : c0 >> 5;     // (a chunk behind the end of its entry, in a batch: its own first word)
  fw.finish(lastw);
  lw.finish(lastw);
  ex = e;
}
template <typename OnGuess, typename Fetch>
ZADA_HD void parse_spec_chunk(const ParseIO &io, uint32_t k, uint32_t chunk, uint32_t *tok, uint32_t &ntok,
                              uint32_t *Fbits, uint32_t *Lbits, ExitState &ex, OnGuess on_guess, Fetch &&fetch) {
  parse_spec_chunk_to(io, k, chunk, DirectSink{tok, ntok}, Fbits, Lbits, ex, on_guess, fetch);
}
ZADA_HD void parse_spec_chunk(const ParseIO &io, uint32_t k, uint32_t chunk, uint32_t *tok, uint32_t &ntok,
                              uint32_t *Fbits, uint32_t *Lbits, ExitState &ex) {
  parse_spec_chunk_to(io, k, chunk, DirectSink{tok, ntok}, Fbits, Lbits, ex, NoGuess(), DirectFetch{io.M});
}

// True parse of chunk k from the true exit of chunk k-1 (`entry`) until it reaches a history-free
// state that the speculative parse of chunk k also went through (then the rest of the speculative
// tokens, from index `take`, are the true ones), or leaves the chunk unsynchronised.
template <typename OnGuess, typename Fetch>
ZADA_HD void parse_fix_chunk(const ParseIO &io, uint32_t k, uint32_t chunk, ExitState entry,
                             const uint32_t *spec_tok, uint32_t spec_cnt, const uint32_t *Fbits, const uint32_t *Lbits,
                             ExitState spec_exit, uint32_t *tok, uint32_t &ntok, uint32_t &take, uint32_t &u0, ExitState &new_exit,
                             OnGuess on_guess, Fetch &&fetch, uint32_t tok_cap = 0xFFFFFFFFu, uint32_t *tok_overflow = nullptr) {
  const uint64_t c0 = (uint64_t)k * chunk, c1 = (c0 + chunk < io.n) ? c0 + chunk : io.n;
  u0 = entry.kind == SYNC_L ? entry.pos - 1 : entry.pos;      // first byte not yet emitted at entry
  ntok = 0;
  if ((uint64_t)entry.pos >= c1) {
    // the previous chunk's parse ran past this whole chunk (only near the end of the input, where
    // no history-free state occurs any more): nothing of this chunk is emitted here
    new_exit = entry; take = spec_cnt;
    return;
  }
  ParseState s{entry.pos, entry.kind == SYNC_L ? 1u : 0u, 2, 0};
  bool synced = false;
  ExitState e; e.pos = (uint32_t)io.n; e.kind = SYNC_F;
  run_parser(s, io, CappedSink{tok, ntok, tok_cap, tok_overflow}, [&](const ParseState &st) {
    int kind = sync_kind(st);
    if (kind == SYNC_NONE) return false;
    if ((uint64_t)st.p >= c1) { e.pos = st.p; e.kind = (uint32_t)kind; return true; }
    uint32_t wbits = (kind == SYNC_F ? Fbits : Lbits)[st.p >> 5];
    if ((wbits >> (st.p & 31)) & 1) { synced = true; e.pos = st.p; e.kind = (uint32_t)kind; return true; }
    return false;
  }, on_guess, fetch);
  if (synced) {
    const uint32_t u = e.kind == SYNC_L ? e.pos - 1 : e.pos;
    uint32_t cur = (uint32_t)c0, j = 0;
    while (cur < u) { cur += tok_len(spec_tok[j]); j++; }   // speculative tokens that precede the sync state
    take = j;
    new_exit = spec_exit;
  } else {
    take = spec_cnt;
    new_exit = e;
  }
}
ZADA_HD void parse_fix_chunk(const ParseIO &io, uint32_t k, uint32_t chunk, ExitState entry,
                                    const uint32_t *spec_tok, uint32_t spec_cnt, const uint32_t *Fbits, const uint32_t *Lbits,
                                    ExitState spec_exit, uint32_t *tok, uint32_t &ntok, uint32_t &take, uint32_t &u0, ExitState &new_exit) {
  parse_fix_chunk(io, k, chunk, entry, spec_tok, spec_cnt, Fbits, Lbits, spec_exit, tok, ntok, take, u0, new_exit, NoGuess(), DirectFetch{io.M});
}

// ----- Patch_statistics_for_buggy_decoders, zip-compress-deflate.adb:340-365 -----
template <typename T>
ZADA_HD void patch_dist_stats(T *sd) {
  int used = 0;
  for (int i = 0; i < 32; i++) if (sd[i] != 0) used++;
  if (used == 0) { sd[0] = 1; sd[1] = 1; }
  else if (used == 1) { if (sd[0] == 0) sd[0] = 1; else sd[1] = 1; }
}

// ----- Tweak_for_better_RLE, zip-compress-deflate.adb:238-318 (in place; good[] scratch) -----
ZADA_HD void tweak_for_better_rle(uint32_t *counts, int counts_len, uint8_t *good) {
  int length = counts_len, stride;
  uint32_t symbol, sum, limit, new_count;
  for (int i = 0; i < counts_len; i++) good[i] = 0;
  for (;;) { if (length == 0) return; if (counts[length - 1] != 0) break; length--; }
  symbol = counts[0]; stride = 0;
  for (int i = 0; i <= length; i++) {
    if (i == length || counts[i] != symbol) {
      if ((symbol == 0 && stride >= 5) || (symbol != 0 && stride >= 7))
        for (int k = 0; k < stride; k++) good[i - k - 1] = 1;
      stride = 1;
      if (i != length) symbol = counts[i];
    } else stride++;
  }
  stride = 0; limit = counts[0]; sum = 0;
  for (int i = 0; i <= length; i++) {
    bool brk = (i == length) || good[i] || (i > 0 && good[i - 1]);
    if (!brk) { int64_t d = (int64_t)counts[i] - (int64_t)limit; if (d < 0) d = -d; brk = d >= 4; }
    if (brk) {
      if (stride >= 4 || (stride >= 3 && sum == 0)) {
        new_count = (sum + (uint32_t)stride / 2) / (uint32_t)stride;
        if (new_count < 1) new_count = 1;
        if (sum == 0) new_count = 0;
        for (int k = 0; k < stride; k++) counts[i - k - 1] = new_count;
      }
      stride = 0; sum = 0;
      if (i < length - 3) limit = (counts[i] + counts[i + 1] + counts[i + 2] + counts[i + 3] + 2) / 4;
      else if (i < length) limit = counts[i];
      else limit = 0;
    }
    stride++;
    if (i != length) sum += counts[i];
  }
}

// ----- canonical codes, huffman-encoding.adb:45-80 with invert_bit_order = True -----
// Emitted codes equal RFC 1951 3.2.2 codes bit-reversed.
ZADA_HD uint32_t bit_reverse(uint32_t code, int len) {
  uint32_t b = 0;
  for (int i = 0; i < len; i++) { b = (b << 1) | (code & 1); code >>= 1; }
  return b;
}
ZADA_HD void canonical_codes(const uint8_t *bl, int n, uint16_t *codes) {
  uint32_t bl_count[16], next_code[16];
  for (int i = 0; i < 16; i++) bl_count[i] = 0;
  for (int i = 0; i < n; i++) bl_count[bl[i]]++;
  bl_count[0] = 0;
  uint32_t code = 0; next_code[0] = 0;
  for (int bits = 1; bits <= 15; bits++) { code = (code + bl_count[bits - 1]) << 1; next_code[bits] = code; }
  for (int k = 0; k < n; k++) {
    int l = bl[k];
    codes[k] = l ? (uint16_t)bit_reverse(next_code[l]++, l) : 0;
  }
}

// ----- dynamic block header, Put_Compression_Structure, zip-compress-deflate.adb:549-704 -----
struct HeaderPlan {
  uint8_t cs_bl[320];       // concatenated bit lengths
  uint16_t last_cs_bl;      // their number
  uint8_t hlit_m257, hdist_m1, a_non_zero;   // HLIT = max_used_lln_code - 256, HDIST = max_used_dis_code
  uint8_t truc_bl[19];
  uint32_t truc_freq[19];
  uint32_t bits;            // exact header cost (:682-685) = 14 + (1+a_non_zero)*3 + sum
};

// RLE walk shared by "simulate" and "effective" (:597-650).  Sink gets (symbol, extra_value).
template <typename Sink>
ZADA_HD void header_rle_walk(const uint8_t *cs_bl, int last, Sink &&sink) {
  int idx = 0;   // 0-based here
  for (;;) {
    int rep = 1;
    for (int j = idx + 1; j < last; j++) { if (cs_bl[j] != cs_bl[idx]) break; rep++; }
    if (idx > 0 && cs_bl[idx] == cs_bl[idx - 1] && rep >= 3 && !(cs_bl[idx] == 0 && rep > 6)) {
      rep = rep < 6 ? rep : 6;
      sink(16, (uint32_t)(rep - 3));
      idx += rep;
    } else if (cs_bl[idx] == 0 && rep >= 3) {
      if (rep <= 10) sink(17, (uint32_t)(rep - 3));
      else { rep = rep < 138 ? rep : 138; sink(18, (uint32_t)(rep - 11)); }
      idx += rep;
    } else {
      sink((int)cs_bl[idx], 0u);
      idx++;
    }
    if (idx >= last) break;
  }
}

ZADA_HD int header_perm(int a) {                       // alphabet_permutation :657-658
  const uint8_t perm[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  return perm[a];
}
ZADA_HD int header_extra_bits(int x) { return x == 16 ? 2 : x == 17 ? 3 : x == 18 ? 7 : 0; }   // :592-593

// ----- similarity metric, zip-compress-deflate.adb:379-433, 457-490 (L1_tweaked only) -----
ZADA_HD int tweak_value(int bl) {   // tweak(), with bl = 0 meaning "unused" -> 16
  const int16_t t[17] = {1600, 100, 255, 379, 490, 594, 694, 791, 885, 978, 1069, 1159, 1249, 1338, 1426, 1513, 1600};
  return t[bl];
}

}  // namespace zada
