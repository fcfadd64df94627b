// hostcheck.cpp -- TEST library (not product code): compiles the host/device logic of
// zip-ada_amd/csrc/zada_logic.h with g++ so that the arithmetic the GPU kernels run per lane can
// be checked against the oracle on a machine without a GPU.  Nothing in the product loads this.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "hostcheck_logic.h"
#include "../../zip-ada_amd/csrc/zada_bt4.h"
#include <algorithm>
#include <numeric>
using namespace zada;

extern "C" {

void hc_llhc(const uint32_t *freq, int n, int max_bits, uint8_t *bl) {
  static LlhcScratch S;
  llhc_serial(freq, n, max_bits, bl, &S);
}
void hc_tweak(uint32_t *counts, int n) { uint8_t good[320]; tweak_for_better_rle(counts, n, good); }
void hc_patch_dist(uint32_t *sd) { patch_dist_stats(sd); }
uint32_t hc_header_bits(const uint8_t *ll, const uint8_t *dd, uint8_t *truc_bl20) {
  static LlhcScratch S; HeaderPlan hp;
  header_plan(ll, dd, &hp, &S);
  for (int i = 0; i < 19; i++) truc_bl20[i] = hp.truc_bl[i];
  truc_bl20[19] = hp.a_non_zero;
  return hp.bits;
}
void hc_canonical(const uint8_t *bl, int n, uint16_t *codes) { canonical_codes(bl, n, codes); }
int hc_len_symbol(int len) { return len_symbol(len); }
int hc_len_extra_bits(int len) { return len_extra_bits(len); }
uint32_t hc_len_extra_val(int len) { return len_extra_val(len); }
int hc_dist_symbol(int d) { return dist_symbol(d); }
int hc_dist_extra_bits(int d) { return dist_extra_bits(d); }
uint32_t hc_dist_extra_val(int d) { return dist_extra_val(d); }

// Match tables by a deliberately simple sequential model (independent of the GPU kernel):
// absolute positions, head/prev as in INSERT_STRING, bytes beyond n never match.
static void match_tables(const uint8_t *in, uint64_t n, const LzConfig &cfg, std::vector<uint32_t> &MF, std::vector<uint32_t> &MQ) {
  std::vector<int64_t> prev(n, -1), head(32768, -1);
  MF.assign(n, 0); MQ.assign(n, 0);
  for (uint64_t p = 0; p + 2 < n; p++) {
    uint32_t h = (((uint32_t)in[p] << 10) ^ ((uint32_t)in[p + 1] << 5) ^ in[p + 2]) & 0x7FFF;
    prev[p] = head[h]; head[h] = (int64_t)p;
    int la = (n - p) < 258 ? (int)(n - p) : 258;
    int nice = cfg.nice < la ? cfg.nice : la;
    int best = 2; uint32_t bd = 0; int steps = 0; bool haveq = false; uint32_t rq = 0;
    int64_t c = prev[p];
    while (c > 0) {
      uint64_t dist = p - (uint64_t)c;
      if (dist > (uint64_t)(steps == 0 ? MAX_DIST : MAX_DIST - 1)) break;
      steps++;
      int len = 0;
      while (len < la && in[c + len] == in[p + len]) len++;
      if (len > best) { best = len; bd = (uint32_t)dist; if (len >= nice) break; }
      if (steps == cfg.chain / 4) { haveq = true; rq = best >= 3 ? ((uint32_t)best << 16) | bd : 0; }
      if (steps == cfg.chain) break;
      c = prev[c];
    }
    MF[p] = best >= 3 ? ((uint32_t)best << 16) | bd : 0;
    MQ[p] = haveq ? rq : MF[p];
  }
}

// Emulates k_parse_spec / k_parse_fix (to the fixpoint) / k_tok_compact sequentially.
uint64_t hc_chunked_tokens(const uint8_t *in, uint64_t n, int level, uint32_t chunk, uint32_t *tokens, uint64_t cap, int *rounds_out) {
  if (n == 0) return 0;
  LzConfig cfg = lz_config(level);
  std::vector<uint32_t> MF, MQ;
  match_tables(in, n, cfg, MF, MQ);
  std::vector<MatchPair> M(n);
  for (uint64_t i = 0; i < n; i++) { M[i].full = MF[i]; M[i].quarter = MQ[i]; }
  ParseIO io{in, n, M.data(), cfg};
  const uint32_t nch = (uint32_t)((n + chunk - 1) / chunk), stride = chunk + 1024;
  std::vector<uint32_t> spec((size_t)nch * stride), fix((size_t)nch * stride), scnt(nch), fcnt(nch), take(nch), u0(nch);
  std::vector<uint32_t> Fb(n / 32 + 2, 0xDEADBEEF), Lb(n / 32 + 2, 0xDEADBEEF);
  std::vector<ExitState> sex(nch), tex(nch);
  for (uint32_t k = 0; k < nch; k++) { uint32_t nt = 0; parse_spec_chunk(io, k, chunk, &spec[(size_t)k * stride], nt, Fb.data(), Lb.data(), sex[k]); scnt[k] = nt; }
  tex = sex;
  std::vector<uint8_t> dirty(nch, 1), nd(nch, 0);
  int rounds = 0;
  for (;;) {
    std::fill(nd.begin(), nd.end(), 0);
    uint32_t changed = 0;
    std::vector<ExitState> snapshot = tex;      // a launch sees the previous round's exits (or newer)
    for (uint32_t k = 0; k < nch; k++) {
      if (!dirty[k]) continue;
      ExitState entry{0, SYNC_F};
      if (k > 0) entry = snapshot[k - 1];
      ExitState ne; uint32_t nt = 0, tk = 0, uu = 0;
      parse_fix_chunk(io, k, chunk, entry, &spec[(size_t)k * stride], scnt[k], Fb.data(), Lb.data(), sex[k], &fix[(size_t)k * stride], nt, tk, uu, ne);
      fcnt[k] = nt; take[k] = tk; u0[k] = uu;
      ExitState old = tex[k];
      tex[k] = ne;
      if (k + 1 < nch && (ne.pos != old.pos || ne.kind != old.kind)) { nd[k + 1] = 1; changed++; }
    }
    rounds++;
    if (!changed) break;
    dirty = nd;
  }
  if (rounds_out) *rounds_out = rounds;
  uint64_t T = 0;
  for (uint32_t k = 0; k < nch; k++) {
    for (uint32_t i = 0; i < fcnt[k]; i++) { if (T < cap) tokens[T] = fix[(size_t)k * stride + i]; T++; }
    for (uint32_t i = take[k]; i < scnt[k]; i++) { if (T < cap) tokens[T] = spec[(size_t)k * stride + i]; T++; }
  }
  return T;
}


// ---- Model of the demand loop (zada_lz.hip, lz_shard "parse / demand rounds") with the chunk logic of zada_logic.h: a first pass that gives every
// position `budget` chain steps and leaves its best-so-far as a GUESS where the chain goes on; parses that land on guesses use them and demand them;
// demanded positions become exact; the chunks whose speculative parse used a value that changed are parsed again -- from the second round on only if
// the new length exceeds the smallest "length to beat" a parse left with the guess (round 6) --; short lists are parsed with the exact value looked
// up inside the parse (k_parse_spec_exact); a splice that needs more token slots than the small stride holds starts the parse again with full slots.
// Sequential, one chunk after the other; the rules (who marks what, what a changed value flags, what a round re-parses) are restated here from the
// kernels' text.  Must give the oracle's tokens whatever the budget, the list threshold, the slots and with or without the lengths to beat.
// stats: [0] demand rounds, [1] chunks parsed again over all rounds, [2] changed values that flagged nothing thanks to their length to beat,
//        [3] restarts for splice slots, [4] guesses of the first pass, [5] chunks parsed with the exact look-up.
struct ModelMarker {
  MatchPair *M; uint8_t *demand; uint32_t *ndem; uint32_t by; bool track;
  void operator()(uint32_t p, uint32_t full, uint32_t quarter, uint32_t beat) const {
    const uint32_t b = beat < M_BEAT_MAX ? beat : M_BEAT_MAX;
    if (track && b < (quarter >> M_BEAT_SHIFT)) {
      const uint32_t v = (quarter & M_VALUE) | (b << M_BEAT_SHIFT);
      if (v < M[p].quarter) M[p].quarter = v;
    }
    const uint32_t want = M_DEMAND | by;
    if ((full & want) == want) return;
    M[p].full = full | want; demand[p] = 1; *ndem = 1;
  }
};
struct ExactFetch {                                   // the parse of a listed chunk that looks the exact value up itself: the record is exact when the parse reads it
  MatchPair *M; const uint32_t *XF, *XQ;
  MatchPair operator()(uint64_t p) const { if (M[p].full & M_GUESS) { M[p].full = XF[p]; M[p].quarter = XQ[p]; } return M[p]; }
};
uint64_t hc_demand_loop_tokens(const uint8_t *in, uint64_t n, int level, uint32_t chunk, int budget, int use_beat, uint32_t exact_max, uint32_t fix_cap,
                               uint32_t *tokens, uint64_t cap, uint64_t *stats) {
  for (int i = 0; i < 6; i++) stats[i] = 0;
  if (n == 0) return 0;
  LzConfig cfg = lz_config(level);
  // exact tables (the model of match_tables above; kept between calls on the same input and level: the test asks for several budgets) and the first
  // pass's records: where the walk takes more than `budget` steps, its state after `budget` steps
  static std::vector<uint8_t> c_in; static int c_level = -1;
  static std::vector<uint32_t> XF, XQ, nsteps; static std::vector<int64_t> prev;
  auto walk = [&](uint64_t p, int stop_after, uint32_t &f, uint32_t &q) -> int {      // stop_after < 0: to the end
    int la = (n - p) < 258 ? (int)(n - p) : 258;
    int nice = cfg.nice < la ? cfg.nice : la;
    int best = 2; uint32_t bd = 0; int steps = 0; bool haveq = false; uint32_t rq = 0;
    int64_t c = prev[p];
    while (c > 0) {
      uint64_t dist = p - (uint64_t)c;
      if (dist > (uint64_t)(steps == 0 ? MAX_DIST : MAX_DIST - 1)) break;
      if (steps == stop_after) break;
      steps++;
      int len = 0;
      while (len < la && in[c + len] == in[p + len]) len++;
      if (len > best) { best = len; bd = (uint32_t)dist; if (len >= nice) break; }
      if (steps == cfg.chain / 4) { haveq = true; rq = best >= 3 ? ((uint32_t)best << 16) | bd : 0; }
      if (steps == cfg.chain) break;
      c = prev[c];
    }
    f = best >= 3 ? ((uint32_t)best << 16) | bd : 0;
    q = haveq ? rq : f;
    return steps;
  };
  if (c_level != level || c_in.size() != n || memcmp(c_in.data(), in, n) != 0) {
    c_in.assign(in, in + n); c_level = level;
    XF.assign(n, 0); XQ.assign(n, 0); nsteps.assign(n, 0); prev.assign(n, -1);
    std::vector<int64_t> head(32768, -1);
    for (uint64_t p = 0; p + 2 < n; p++) {
      uint32_t h = (((uint32_t)in[p] << 10) ^ ((uint32_t)in[p + 1] << 5) ^ in[p + 2]) & 0x7FFF;
      prev[p] = head[h]; head[h] = (int64_t)p;
      nsteps[p] = (uint32_t)walk(p, -1, XF[p], XQ[p]);
    }
  }
  std::vector<MatchPair> M(n);
  for (uint64_t p = 0; p < n; p++) {
    M[p].full = XF[p]; M[p].quarter = XQ[p];
    if (p + 2 < n && nsteps[p] > (uint32_t)budget) {                               // the chain goes on behind the budget: a guess
      uint32_t gf, gq;
      walk(p, budget, gf, gq);
      M[p].full = gf | M_GUESS; M[p].quarter = gq | (M_BEAT_MAX << M_BEAT_SHIFT); stats[4]++;
    }
  }
  ParseIO io{in, n, M.data(), cfg};
  const uint32_t nch = (uint32_t)((n + chunk - 1) / chunk), stride = chunk + 1024;
  std::vector<uint32_t> spec((size_t)nch * stride), fix((size_t)nch * stride), scnt(nch), fcnt(nch), take(nch), u0(nch);
  std::vector<uint32_t> Fb(n / 32 + 2, 0xDEADBEEF), Lb(n / 32 + 2, 0xDEADBEEF);
  std::vector<ExitState> sex(nch), tex(nch);
  std::vector<uint8_t> demand(n, 0), chg(nch, 0);
  uint32_t ndem = 0;
  bool first = true;
  for (int guard = 0; guard < 100000; guard++) {
    const bool beat_valid = !first;
    if (first) {
      ModelMarker dm{M.data(), demand.data(), &ndem, M_BYSPEC, false};
      for (uint32_t k = 0; k < nch; k++) { uint32_t nt = 0; parse_spec_chunk(io, k, chunk, &spec[(size_t)k * stride], nt, Fb.data(), Lb.data(), sex[k], dm, DirectFetch{io.M}); scnt[k] = nt; }
    } else {
      uint32_t nl = 0;
      for (uint32_t k = 0; k < nch; k++) nl += chg[k];
      const bool exact = exact_max && nl <= exact_max;
      ModelMarker dm{M.data(), demand.data(), &ndem, M_BYSPEC, true};
      for (uint32_t k = 0; k < nch; k++) {
        if (!chg[k]) continue;
        uint32_t nt = 0; stats[1]++;
        if (exact) { stats[5]++; parse_spec_chunk(io, k, chunk, &spec[(size_t)k * stride], nt, Fb.data(), Lb.data(), sex[k], NoGuess(), ExactFetch{M.data(), XF.data(), XQ.data()}); }
        else parse_spec_chunk(io, k, chunk, &spec[(size_t)k * stride], nt, Fb.data(), Lb.data(), sex[k], dm, DirectFetch{io.M});
        scnt[k] = nt;
      }
    }
    // the splice, from scratch, to its fixpoint
    tex = sex;
    std::vector<uint8_t> dirty(nch, 1), nd(nch, 0);
    uint32_t overflow = 0;
    ModelMarker dmf{M.data(), demand.data(), &ndem, 0, false};
    for (;;) {
      std::fill(nd.begin(), nd.end(), 0);
      uint32_t changed = 0;
      std::vector<ExitState> snapshot = tex;
      for (uint32_t k = 0; k < nch; k++) {
        if (!dirty[k]) continue;
        ExitState entry{0, SYNC_F};
        if (k > 0) entry = snapshot[k - 1];
        ExitState ne; uint32_t nt = 0, tk = 0, uu = 0;
        parse_fix_chunk(io, k, chunk, entry, &spec[(size_t)k * stride], scnt[k], Fb.data(), Lb.data(), sex[k], &fix[(size_t)k * stride], nt, tk, uu, ne, dmf, DirectFetch{io.M},
                        fix_cap, &overflow);
        fcnt[k] = nt; take[k] = tk; u0[k] = uu;
        ExitState old = tex[k];
        tex[k] = ne;
        if (k + 1 < nch && (ne.pos != old.pos || ne.kind != old.kind)) { nd[k + 1] = 1; changed++; }
      }
      if (overflow || !changed) break;
      dirty = nd;
    }
    if (overflow) { fix_cap = 0xFFFFFFFFu; stats[3]++; first = true; continue; }      // full slots, and the parse starts again (what is exact stays exact)
    if (!ndem) break;
    // the demand pass: every marked position searched to the end
    stats[0]++;
    ndem = 0;
    std::fill(chg.begin(), chg.end(), 0);
    for (uint64_t p = 0; p < n; p++) {
      if (!demand[p]) continue;
      demand[p] = 0;
      const MatchPair og = M[p];
      M[p].full = XF[p]; M[p].quarter = XQ[p];
      if (!(og.full & M_BYSPEC)) continue;
      const uint32_t beat = (use_beat && beat_valid) ? og.quarter >> M_BEAT_SHIFT : 0u, ogq = og.quarter & M_VALUE;
      const bool df = XF[p] != (og.full & M_VALUE), dq = XQ[p] != ogq;
      if ((df && (XF[p] >> 16) > beat) || (dq && (XQ[p] >> 16) > beat)) {
        const uint64_t ch = p / chunk;
        chg[ch] = 1;
        if (ch > 0 && (uint64_t)sex[ch - 1].pos > p) chg[ch - 1] = 1;                     // the chunk before ran over into this position
      } else if (df || dq) stats[2]++;
    }
    first = false;
  }
  uint64_t T = 0;
  for (uint32_t k = 0; k < nch; k++) {
    for (uint32_t i = 0; i < fcnt[k]; i++) { if (T < cap) tokens[T] = fix[(size_t)k * stride + i]; T++; }
    for (uint32_t i = take[k]; i < scnt[k]; i++) { if (T < cap) tokens[T] = spec[(size_t)k * stride + i]; T++; }
  }
  return T;
}
}  // extern "C"

// ---- reference of the wave-parallel LLHC algorithm's MATH (classic package-merge with the
// "package before leaf on ties" rule + closed-form Hoare partition), to validate it against the
// oracle's boundary package-merge before the device version is trusted ----
#include <algorithm>
static void ref_partition_sort(std::vector<uint32_t> &w, std::vector<uint16_t> &s, int lo, int m) {
  if (m < 2) return;
  uint32_t p = w[lo + m / 2];
  std::vector<int> I, J;
  for (int i = 0; i < m; i++) if (w[lo + i] >= p) I.push_back(i);
  for (int i = m - 1; i >= 0; i--) if (w[lo + i] <= p) J.push_back(i);
  int K = 0;
  while (K < (int)I.size() && K < (int)J.size() && I[K] < J[K]) K++;
  for (int k = 0; k < K; k++) { std::swap(w[lo + I[k]], w[lo + J[k]]); std::swap(s[lo + I[k]], s[lo + J[k]]); }
  int i = 1 << 30;
  if (K < (int)I.size()) i = std::min(i, I[K]);
  if (K > 0) i = std::min(i, J[K - 1]);
  ref_partition_sort(w, s, lo, i);
  ref_partition_sort(w, s, lo + i, m - i);
}
extern "C" void hc_llhc_pm(const uint32_t *freq, int n, int max_bits, uint8_t *bl) {
  std::vector<uint32_t> w; std::vector<uint16_t> s;
  for (int a = 0; a < n; a++) { bl[a] = 0; if (freq[a]) { w.push_back(freq[a]); s.push_back((uint16_t)a); } }
  int ns = (int)w.size();
  if (ns == 0) return;
  if (ns == 1) { bl[s[0]] = 1; return; }
  ref_partition_sort(w, s, 0, ns);
  // lists: level 1 = leaves; level l = merge(leaves, packages(level l-1)), packages first on ties
  std::vector<std::vector<uint32_t>> lists(max_bits + 1);
  std::vector<std::vector<uint8_t>> isleaf(max_bits + 1);
  lists[1] = w; isleaf[1].assign(ns, 1);
  for (int l = 2; l <= max_bits; l++) {
    std::vector<uint32_t> P;
    for (size_t i = 0; i + 1 < lists[l - 1].size(); i += 2) P.push_back(lists[l - 1][i] + lists[l - 1][i + 1]);
    size_t a = 0, b = 0;
    while (a < w.size() || b < P.size()) {
      bool takeP = b < P.size() && (a >= w.size() || P[b] <= w[a]);
      if (takeP) { lists[l].push_back(P[b++]); isleaf[l].push_back(0); } else { lists[l].push_back(w[a++]); isleaf[l].push_back(1); }
    }
  }
  int x = 2 * ns - 2;
  std::vector<int> acnt(max_bits + 1, 0);
  for (int l = max_bits; l >= 1; l--) {
    int a = 0;
    for (int i = 0; i < x && i < (int)lists[l].size(); i++) a += isleaf[l][i];
    acnt[l] = a;
    x = 2 * (x - a);
  }
  for (int r = 0; r < ns; r++) { int len = 0; for (int l = 1; l <= max_bits; l++) if (acnt[l] > r) len++; bl[s[r]] = (uint8_t)len; }
}

// The producer form of BT4 (zada_bt4.h) on the CPU, the way the kernels of zada_bt4.hip run it: hashes of the inserted positions,
// three stable sorts (hash2 / hash3 predecessors, hash-4 buckets), then every bucket on its own -- in a SHUFFLED order of the buckets
// (seed), to show that the buckets do not depend on each other.  Same outputs as the oracle's zo_bt4_match_sets.  Returns 0, or -1 when
// the schedule is refused.
// seg_shift < 32: the stream in segments of 2 ** seg_shift positions, as one stream coded in launches takes it (bt4_walk_segment): the buckets
// of a segment in a shuffled order, the segments one after the other, a bucket's root handed on through a table indexed by the hash-4 key.
static int bt4_sets_impl(const uint8_t *in, uint64_t n, int64_t dict, uint8_t *cnt, uint16_t *len, uint32_t *dist, int stride, uint32_t seed, uint32_t seg_shift) {
  const uint32_t sbs = bt4_string_buffer_size((uint64_t)dict), mask = bt4_hash4_size(sbs) - 1;
  const int32_t max_dist = (int32_t)sbs - (BT4_LOOK + 2);
  std::vector<Bt4Run> runs;
  if (!bt4_schedule(n, sbs, runs)) return -1;
  memset(cnt, 0, n);
  std::vector<uint32_t> pos, h2, h3, h4;
  for (uint64_t q = 0; q < n; q++) {
    const Bt4Run *r = bt4_run_of(runs.data(), (uint32_t)runs.size(), (uint32_t)q);
    if (r->cls == 2) continue;
    uint32_t a, b, c;
    bt4_hashes(bt4_crc(in[q]), in[q + 1], in[q + 2], bt4_crc(in[q + 3]), mask, a, b, c);
    pos.push_back((uint32_t)q); h2.push_back(a); h3.push_back(b); h4.push_back(c);
  }
  const size_t m = pos.size();
  auto ord_of = [&](uint32_t q) { return (int32_t)(q - bt4_run_of(runs.data(), (uint32_t)runs.size(), q)->gap); };
  std::vector<int32_t> o2(n, BT4_NONE), o3(n, BT4_NONE);
  std::vector<uint32_t> idx(m);
  auto preds = [&](const std::vector<uint32_t> &h, std::vector<int32_t> &o) {
    std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return h[x] < h[y]; });
    for (size_t i = 1; i < m; i++) if (h[idx[i]] == h[idx[i - 1]]) o[pos[idx[i]]] = ord_of(pos[idx[i - 1]]);
  };
  preds(h2, o2); preds(h3, o3);
  std::iota(idx.begin(), idx.end(), 0u);
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return h4[x] < h4[y]; });
  auto seg_of = [&](uint32_t k) { return seg_shift < 32 ? pos[k] >> seg_shift : 0u; };
  if (seg_shift < 32) std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return seg_of(x) < seg_of(y); });
  std::vector<std::pair<size_t, size_t>> buckets;
  for (size_t i = 0; i < m;) { size_t j = i + 1; while (j < m && h4[idx[j]] == h4[idx[i]] && seg_of(idx[j]) == seg_of(idx[i])) j++; buckets.push_back({i, j}); i = j; }
  uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
  for (size_t s0 = 0; s0 < buckets.size();) {                          // shuffled within a segment
    size_t s1 = s0 + 1;
    while (s1 < buckets.size() && seg_of(idx[buckets[s1].first]) == seg_of(idx[buckets[s0].first])) s1++;
    for (size_t i = s1 - s0; i > 1; i--) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::swap(buckets[s0 + i - 1], buckets[s0 + x % i]); }
    s0 = s1;
  }
  std::vector<int32_t> htab(seg_shift < 32 ? (size_t)mask + 1 : 0, BT4_NONE);
  std::vector<int32_t> tree(2 * n + 2, 12345);       // (never read before written: a walk only reaches nodes of its own bucket)
  std::vector<uint16_t> tree16(2 * n + 2, 12345);    // the LDS form of the nodes (k_bt4_walk_lds) for entries it would take
  const bool small = n <= BT4_LDS_N && runs.size() <= 2;
  uint16_t ml[BT4_SET]; uint32_t md[BT4_SET];
  auto ext = [](const uint8_t *b, int64_t a, int64_t c, int l, int lim) { return bt4_extend(b, a, c, l, lim); };
  for (auto &bk : buckets) {
    int32_t root = htab.empty() ? BT4_NONE : htab[h4[idx[bk.first]]];
    for (size_t i = bk.first; i < bk.second; i++) {
      const uint32_t q = pos[idx[i]];
      const Bt4Run *r = bt4_run_of(runs.data(), (uint32_t)runs.size(), q);
      const int32_t ordp = (int32_t)(q - r->gap);
      const int avail = (int)(r->W - q - 1), limit = avail < BT4_LOOK ? avail : BT4_LOOK;
      auto put = [&](int i, int l, uint32_t dd) { ml[i] = (uint16_t)l; md[i] = dd; };
      Bt4Walk wk;
      bt4_begin(wk, in, q, ordp, r->cls == 0, limit, max_dist, root, o2[q], o3[q], ext, put);
      if (small) { while (!bt4_step(wk, Bt4TreeU16{tree16.data()}, ext, put)) {} }
      else { while (!bt4_step(wk, Bt4TreeI32{tree.data()}, ext, put)) {} }
      const int c = wk.count;
      if (c > stride) return -2;
      cnt[q] = (uint8_t)c;
      for (int k = 0; k < c; k++) { len[(uint64_t)q * stride + k] = ml[k]; dist[(uint64_t)q * stride + k] = md[k]; }
      root = ordp;
    }
    if (!htab.empty()) htab[h4[idx[bk.first]]] = root;
  }
  return 0;
}
extern "C" int hc_bt4_sets(const uint8_t *in, uint64_t n, int64_t dict, uint8_t *cnt, uint16_t *len, uint32_t *dist, int stride, uint32_t seed) {
  return bt4_sets_impl(in, n, dict, cnt, len, dist, stride, seed, 32);
}
// 1: an entry of n bytes coded with this dictionary has positions read behind pending bytes that no window fill took up (zada_bt4.h
// bt4_reads_behind_a_gap: zada_lzma then verifies the match sets it reads); 0: not; -1: the schedule is refused
extern "C" int hc_bt4_reads_behind_a_gap(uint64_t n, int64_t dict) {
  std::vector<Bt4Run> runs;
  if (!bt4_schedule(n, bt4_string_buffer_size((uint64_t)dict), runs)) return -1;
  return bt4_reads_behind_a_gap(runs) ? 1 : 0;
}
extern "C" int hc_bt4_sets_segments(const uint8_t *in, uint64_t n, int64_t dict, uint8_t *cnt, uint16_t *len, uint32_t *dist, int stride, uint32_t seed, uint32_t seg_shift) {
  return bt4_sets_impl(in, n, dict, cnt, len, dist, stride, seed, seg_shift);
}

// Analysis helper (not a test): steps of bt4_step per hash-4 bucket -- the longest sum is the producer's critical path.
extern "C" int hc_bt4_bucket_steps(const uint8_t *in, uint64_t n, int64_t dict, uint64_t *out /* [0] buckets, [1] positions, [2] total steps, [3] max steps of a bucket, [4] its positions, [5] max positions of a bucket */) {
  const uint32_t sbs = bt4_string_buffer_size((uint64_t)dict), mask = bt4_hash4_size(sbs) - 1;
  const int32_t max_dist = (int32_t)sbs - (BT4_LOOK + 2);
  std::vector<Bt4Run> runs;
  if (!bt4_schedule(n, sbs, runs)) return -1;
  std::vector<uint32_t> pos, h2, h3, h4;
  for (uint64_t q = 0; q < n; q++) {
    const Bt4Run *r = bt4_run_of(runs.data(), (uint32_t)runs.size(), (uint32_t)q);
    if (r->cls == 2) continue;
    uint32_t a, b, c;
    bt4_hashes(bt4_crc(in[q]), in[q + 1], in[q + 2], bt4_crc(in[q + 3]), mask, a, b, c);
    pos.push_back((uint32_t)q); h2.push_back(a); h3.push_back(b); h4.push_back(c);
  }
  const size_t m = pos.size();
  std::vector<int32_t> o2(n, BT4_NONE), o3(n, BT4_NONE);
  std::vector<uint32_t> idx(m);
  auto preds = [&](const std::vector<uint32_t> &h, std::vector<int32_t> &o) {
    std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return h[x] < h[y]; });
    for (size_t i = 1; i < m; i++) if (h[idx[i]] == h[idx[i - 1]]) o[pos[idx[i]]] = (int32_t)pos[idx[i - 1]];
  };
  preds(h2, o2); preds(h3, o3);
  std::iota(idx.begin(), idx.end(), 0u);
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return h4[x] < h4[y]; });
  std::vector<int32_t> tree(2 * n + 2, 12345);
  auto ext = [](const uint8_t *b, int64_t a, int64_t c, int l, int lim) { return bt4_extend(b, a, c, l, lim); };
  auto put = [&](int, int, uint32_t) {};
  memset(out, 0, 6 * sizeof(uint64_t));
  for (size_t i = 0; i < m;) {
    size_t j = i + 1; while (j < m && h4[idx[j]] == h4[idx[i]]) j++;
    int32_t root = BT4_NONE; uint64_t steps = 0;
    for (size_t k = i; k < j; k++) {
      const uint32_t q = pos[idx[k]];
      const Bt4Run *r = bt4_run_of(runs.data(), (uint32_t)runs.size(), q);
      const int avail = (int)(r->W - q - 1);
      Bt4Walk wk;
      bt4_begin(wk, in, q, (int32_t)q, r->cls == 0, avail < BT4_LOOK ? avail : BT4_LOOK, max_dist, root, o2[q], o3[q], ext, put);
      steps++;
      while (!bt4_step(wk, Bt4TreeI32{tree.data()}, ext, put)) steps++;
      root = (int32_t)q;
    }
    out[0]++; out[1] += j - i; out[2] += steps;
    if (steps > out[3]) { out[3] = steps; out[4] = j - i; }
    if (j - i > out[5]) out[5] = j - i;
    i = j;
  }
  return 0;
}
