// zada_lzma.hip -- LZMA encoding for gfx950: LZMA.Encoding.Encode (zip_lib/lzma-encoding.adb:59-1563) with the
// parameters of Zip.Compress.LZMA_E's methods LZMA_0 .. LZMA_3 (zip-compress-lzma_e.adb:121-126: lc 3, lp 0, pb 2, end marker).
//
// What is parallel and what is not.  The range coder (:964-1039) and the choice between the ways of writing a match (:349-946)
// both read the adaptive bit probabilities, which every coded bit updates: a stream is one chain of dependent steps, and the
// stream's bytes have to be the reference's.  So the unit of parallelism is the STREAM: one 64-lane workgroup per Zip entry, the
// entry's probability model (7 992 probabilities, 16 KB), its two match lists and the encoder's own state in LDS.  All lanes walk
// the chain in step (same data, same branches: the cost of one lane) and part where the reference compares INDEPENDENT
// simulations of what to write next -- a team of lanes per simulation, one lane per cut inside it (decide, scoring_top) -- to
// compare the results in the reference's order.  Level_1 / Level_2 take their LZ77 tokens from the Info-Zip matcher kernels of the
// Deflate path (IZ_6 / IZ_10, :118-122; zada_lz.hip) -- that part IS data parallel; Level_3's BT4 binary-tree matcher
// (lz77.adb:953-1827) runs AHEAD of the chain as well: its match sets are a function of the input alone, one tree per hash-4 bucket,
// and the producer of zada_bt4.hip leaves them in HBM before this kernel starts; what stays in the chain of LZ77_using_BT4 is the part
// that reads the coder's state (repeat distances, the look-ahead of one position scored by the estimates, :1605-1796).  The floating-point
// estimates are IEEE doubles multiplied in the reference's order, without contraction, so they are the values the Ada code computes.
//
// The sliding text buffer of the reference (Text_Buf, a ring of String_buffer_size bytes) only ever holds bytes of the input at
// their own positions -- also ahead of the encoder, where matches under test are expanded early (:1338-1353, 1488-1492) -- so the
// kernel reads the input itself: Text_Buf ((R - d) and mask) = in [total_pos - d].
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/zada.h"
#include "zada_internal.h"
#include "zada_bt4.h"

namespace zada {
namespace {

constexpr int LZ_LIT = 0x300 << 3;             // lc = 3, lp = 0
constexpr uint32_t LZ_PBM = 3;                 // pb = 2
constexpr int LZ_MAXM = BT4_SET + 2;           // matches of one position: BT4 finds at most 50 (zada_bt4.h), + 1 repeat match, (1 .. count)

struct LenProbs { uint16_t c1, c2, low[16][8], mid[16][8], high[256]; };
struct LzProbs {                               // lzma.ads:137-201
  uint16_t lit[LZ_LIT];
  uint16_t slot[4][64], align[16], pos[116];   // pos_coder (-1 .. 114) at +1
  LenProbs len, rep_len;
  uint16_t match[12][16], rep[12], g0[12], g1[12], g2[12], rep0_long[12][16];
};
struct Matches { int count; int dist[LZ_MAXM]; uint16_t len[LZ_MAXM]; };     // lz77.ads:70-75, (1 .. count); lengths are 2 .. 273

struct MS {                                    // Machine_State :212-219 without R (= total_pos mod the ring size)
  uint32_t state, pos_state, prev_byte;
  uint32_t rep[4];
  uint64_t pos;
  int tw;                                        // simulations: width of the team of lanes that runs this one in step (1: a lane alone) -- see decide
};

// The entry's probability model, the two match sets of BT4's look-ahead and what the lanes hand each other at a fork: file-scope LDS
// objects, so that every access is a ds_ instruction (through a pointer in a struct they were flat loads: 36 M of them per 64 KiB).
__shared__ LzProbs s_P;
__shared__ Matches s_MM[2];

struct Enc {
  const uint8_t *in; uint64_t n;
  int cv;                                      // compare_variants: 0 None, 1 Simple, 2 Splitting (:1539-1546)
  MS ES;
  uint32_t width; uint64_t low; uint32_t cache; uint64_t cache_size;      // Range_Encoder :952-957
  uint8_t *out; uint64_t cap, olen;
  uint32_t verify, defect;                     // Level_3: check the match sets against the text (LzmaJob::verify) / a match that is none was found
};
// The encoder's state as well: every lane holds the same values outside the forks, the simulations only read it.  (As a local of
// the kernel it was reached through a pointer into scratch: a flat load, 100+ cycles and both wait counters, per field read.)
__shared__ Enc s_E;

// A stream's workgroup is one wave -- or four ("helpers", one stream alone on the chip: k_lzma_encode): the chain is walked by wave 0 only,
// the other waves run shares of its forks.  lane_id: the lane within the wave; chain_sync: between the lanes of the wave that walks the chain.
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
// (HW: the kernel's template parameter -- the one-wave kernel of the batches holds none of the helpers' code)
template <bool HW> __device__ __forceinline__ void chain_sync() {
  if constexpr (!HW) __syncthreads();
  else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
}

#ifdef ZADA_LZ_PROF
__device__ unsigned long long g_lzprof[8];
#define PROF_T0 const unsigned long long prof_t0 = clock64()
#define PROF_ADD(i) g_lzprof[i] += clock64() - prof_t0
#else
#define PROF_T0
#define PROF_ADD(i)
#endif

// The state transitions of lzma.ads:86-89 (Update_State_Literal = 0 0 0 0 1 2 3 4 5 6 4 5, _Match = 7 x 7, 10 x 5, _Rep = 8 x 7, 11 x 5,
// _ShortRep = 9 x 7, 11 x 5) as arithmetic: as tables in device memory every simulated symbol waited for a load before the next
// symbol's probabilities could even be addressed.
__device__ __forceinline__ uint32_t t_lit(uint32_t s) { return s < 4 ? 0u : s < 10 ? s - 3u : s - 6u; }
__device__ __forceinline__ uint32_t t_match(uint32_t s) { return s < 7 ? 7u : 10u; }
__device__ __forceinline__ uint32_t t_rep(uint32_t s) { return s < 7 ? 8u : 11u; }
__device__ __forceinline__ uint32_t t_srep(uint32_t s) { return s < 7 ? 9u : 11u; }

__device__ inline uint32_t TB(int64_t p) { return p < 0 ? 0u : (uint32_t)s_E.in[p]; }

__device__ inline uint32_t dist_slot(uint32_t d) {                               // Get_dist_slot :73-102
  if (d <= 4) return d;
  const int i = 31 - __clz(d);
  return (uint32_t)(i * 2) + ((d >> (i - 1)) & 1);
}

// ---------------------------------------------------------------- Estimates :349-946

// Test_Bit_Encoding :359-370 is b + (1.0 - 2.0 * b) * (prob * 2**-11) with b = 0.0 or 1.0.  prob * 2**-11 is exact; for b = 0 the
// result is that value (0.0 + 1.0 * x), for b = 1 it is 1.0 - x = (2048 - prob) * 2**-11, representable, so the one rounding of the
// reference's sum returns exactly it.  Hence: pick the integer, convert, scale -- the same double with two operations instead of five.
__device__ inline double tbe(uint16_t p, uint32_t sym) {
  const uint32_t q = sym ? 2048u - (uint32_t)p : (uint32_t)p;
  return (double)q * (1.0 / 2048.0);
}
// The same without the scaling.  A product of such factors, multiplied in the reference's order and scaled ONCE at the end by 2 ** (-11 n), is
// the reference's product bit for bit: a scaling by a power of two commutes with every rounding as long as nothing leaves the normal range,
// and the unscaled partial products here stay between 31 ** 10 and 2048 ** 10 (probabilities are kept within 31 .. 2017 by their update
// rule; a function multiplies at most ten of them before it scales).  One multiplication per factor instead of two.
__device__ inline double tq(uint16_t p, uint32_t sym) {
  const uint32_t q = sym ? 2048u - (uint32_t)p : (uint32_t)p;
  return (double)q;
}
__device__ inline double scale11(double x, int factors) { return __builtin_ldexp(x, -11 * factors); }

__device__ double test_simple_literal(uint32_t b, uint32_t b_match, int idx, const MS &sim) {   // :372-419
  const uint16_t *prob = s_P.lit + idx;
  double pl = tq(s_P.match[sim.state][sim.pos_state], 0);
  uint32_t symb = b | 0x100;
  uint16_t pr[8];                                 // the eight probabilities first (their addresses do not depend on each other), then the products in order
  if (sim.state < 7) {
#pragma unroll
    for (int k = 0; k < 8; k++) pr[k] = prob[(symb << k) >> 8];
  } else {
    uint32_t offs = 0x100, match = b_match, sy = symb;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      match <<= 1;
      pr[k] = prob[offs + (match & offs) + (sy >> 8)];
      sy <<= 1;
      offs &= ~(match ^ sy);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) pl = pl * tq(pr[k], (symb >> (7 - k)) & 1);
  return scale11(pl, 9);
}

__device__ inline double test_short_rep(const MS &sim) {           // :421-428
  const LzProbs &P = s_P;
  return scale11(tq(P.match[sim.state][sim.pos_state], 1) * tq(P.rep[sim.state], 1) * tq(P.g0[sim.state], 0) * tq(P.rep0_long[sim.state][sim.pos_state], 0), 4);
}

__device__ inline int lit_idx(uint32_t prev_byte) { return 0x300 * (int)(prev_byte >> 5); }    // Idx_for_Literal_prob :193-201

// Simulate_Literal_Byte :431-458; b_match = the byte at the last distance, Text_Buf ((R - rep_dist (0) - 1) and mask)
__device__ __forceinline__ void sim_literal_bm(uint32_t b, uint32_t b_match, MS &sim, double &prob) {
  const int idx = lit_idx(sim.prev_byte);
  sim.pos_state = (uint32_t)sim.pos & LZ_PBM;
  const double ltr = test_simple_literal(b, b_match, idx, sim);
  bool srep = false;
  if (b == b_match && sim.pos > (uint64_t)(uint32_t)(sim.rep[0] + 1)) {
    const double srm = test_short_rep(sim);
    if (srm > ltr) { sim.state = t_srep(sim.state); prob = prob * srm; srep = true; }
  }
  if (!srep) { sim.state = t_lit(sim.state); prob = prob * ltr; }
  sim.pos += 1;
  sim.pos_state = (uint32_t)sim.pos & LZ_PBM;
  sim.prev_byte = b;
}
__device__ __forceinline__ void sim_literal(uint32_t b, MS &sim, double &prob) {
  sim_literal_bm(b, TB((int64_t)sim.pos - (int64_t)sim.rep[0] - 1), sim, prob);
}

__device__ inline double test_literal_byte(uint32_t b, const MS &sim) {          // :460-468
  MS v = sim; double prob = 1.0;
  sim_literal(b, v, prob);
  return prob;
}

// (the `_u` functions return UNSCALED products -- see tq -- and say how many factors they hold; their callers scale)
template <int NB> __device__ inline double sim_bit_tree_u(const uint16_t *prob, uint32_t symbol) {   // Simulate_Bit_Tree :470-481, NB factors
  uint16_t pr[NB];
  uint32_t m = 1;
#pragma unroll
  for (int k = 0; k < NB; k++) { pr[k] = prob[m]; m = 2 * m + ((symbol >> (NB - 1 - k)) & 1); }
  double res = 1.0;
#pragma unroll
  for (int k = 0; k < NB; k++) res = res * tq(pr[k], (symbol >> (NB - 1 - k)) & 1);
  return res;
}
__device__ inline double sim_bit_tree_rev_u(const uint16_t *prob, int num_bits, uint32_t symbol) {   // :548-563, num_bits factors (at most 5)
  double res = 1.0; uint32_t m = 1;
  for (int c = num_bits; c >= 1; c--) { const uint32_t bit = symbol & 1; res = res * tq(prob[m], bit); m = 2 * m + bit; symbol >>= 1; }
  return res;
}

__device__ double test_length_u(bool rep, uint32_t length, uint32_t ps, int &factors) {              // :483-509
  const LenProbs &pl = rep ? s_P.rep_len : s_P.len;
  uint32_t len = length - 2; double res;
  if (len < 8) { res = tq(pl.c1, 0) * sim_bit_tree_u<3>(pl.low[ps], len); factors = 4; }
  else {
    res = tq(pl.c1, 1); len -= 8;
    if (len < 8) { res = res * tq(pl.c2, 0) * sim_bit_tree_u<3>(pl.mid[ps], len); factors = 5; }
    else { res = res * tq(pl.c2, 1); len -= 8; res = res * sim_bit_tree_u<8>(pl.high, len); factors = 10; }
  }
  return res;
}

__device__ double test_repeat_match(int index_rm, uint32_t length, const MS &sim) {   // :511-538
  const LzProbs &P = s_P;
  // (the switches' and the length's partial products are scaled separately, as the reference forms them: two products, then their product)
  double res = tq(P.rep[sim.state], 1);
  int nsw;
  switch (index_rm) {
    case 0: res = res * tq(P.g0[sim.state], 0) * tq(P.rep0_long[sim.state][sim.pos_state], 1); nsw = 3; break;
    case 1: res = res * tq(P.g0[sim.state], 1) * tq(P.g1[sim.state], 0); nsw = 3; break;
    case 2: res = res * tq(P.g0[sim.state], 1) * tq(P.g1[sim.state], 1) * tq(P.g2[sim.state], 0); nsw = 4; break;
    default: res = res * tq(P.g0[sim.state], 1) * tq(P.g1[sim.state], 1) * tq(P.g2[sim.state], 1); nsw = 4; break;
  }
  int nlen;
  const double tl = test_length_u(true, length, sim.pos_state, nlen);
  return scale11(res * tl, nsw + nlen);
}

__device__ double test_simple_match(uint32_t distance, uint32_t length, const MS &sim) {   // :540-601
  const LzProbs &P = s_P;
  const uint32_t len_state = length - 2 < 3 ? length - 2 : 3, ds = dist_slot(distance);
  double td = sim_bit_tree_u<6>(P.slot[len_state], ds);
  int ntd = 6;
  if (ds >= 4) {
    const int footer = (int)(ds >> 1) - 1;
    const uint32_t base = (2 | (ds & 1)) << footer, red = distance - base;
    if (ds < 14) { td = td * sim_bit_tree_rev_u(P.pos + ((int)base - (int)ds - 1) + 1, footer, red); ntd += footer; }
    else {
      double h = 1.0;
      for (int i = 0; i < footer - 4; i++) h = h * 0.5;                          // 0.5 ** (footerBits - align_bits), exact
      td = td * h * sim_bit_tree_rev_u(P.align, 4, red & 15); ntd += 4;
    }
  }
  int nlen;
  const double tl = test_length_u(false, length, sim.pos_state, nlen);
  return scale11(tq(P.rep[sim.state], 0) * tl * td, 1 + nlen + ntd);
}

// Simulate_Strict_DL_Code :605-659 as the state it leaves and the two factors it multiplies the probability by (prob := prob * f1 * f2, in that
// order): Generic_any_DL_Code tests the strict code first (:661-677, from probability 1.0: 1.0 * f1 is f1) and, when nothing beats it, simulates
// it again on the same state (:826-830) -- the second time the factors are the first time's.
// (what is kept is small -- the two factors and WHICH code it was: the state it leaves is worked out again from that, a few integer operations and
// one byte of the text; kept whole, the state cost sim_any_impl<2> twenty spilled registers across its nested simulations)
struct StrictRes { double f1, f2; int found; };      // found: the repeat distance taken (0 .. 3), -1: a simple match
__device__ __forceinline__ StrictRes strict_factors(uint32_t distance, int length, const MS &sim) {
  StrictRes r;
  const uint32_t dist_ip = distance - 1;
  r.f1 = tbe(s_P.match[sim.state][sim.pos_state], 1);
  const double sma = test_simple_match(dist_ip, (uint32_t)length, sim);
  // (no run-time index into sim.rep: one would put the whole state in scratch memory)
  const int found = dist_ip == sim.rep[0] ? 0 : dist_ip == sim.rep[1] ? 1 : dist_ip == sim.rep[2] ? 2 : dist_ip == sim.rep[3] ? 3 : -1;
  r.f2 = sma; r.found = -1;
  if (found >= 0) {
    const double rma = test_repeat_match(found, (uint32_t)length, sim);
    if (rma >= sma * 0.55) { r.f2 = rma; r.found = found; }                        // Malus_simple_match_vs_rep :301
  }
  return r;
}
// the state behind the strict code whose factors are r
__device__ __forceinline__ void strict_apply(uint32_t distance, int length, const StrictRes &r, MS &sim) {
  const uint32_t dist_ip = distance - 1;
  if (r.found >= 0) {
    const uint32_t r0 = sim.rep[0], r1 = sim.rep[1], r2 = sim.rep[2];             // rep (found) to the front, the ones before it one down
    sim.rep[0] = dist_ip;
    if (r.found >= 1) sim.rep[1] = r0;
    if (r.found >= 2) sim.rep[2] = r1;
    if (r.found >= 3) sim.rep[3] = r2;
    sim.state = t_rep(sim.state);
  } else {
    sim.rep[3] = sim.rep[2]; sim.rep[2] = sim.rep[1]; sim.rep[1] = sim.rep[0]; sim.rep[0] = dist_ip;
    sim.state = t_match(sim.state);
  }
  sim.pos += (uint64_t)length;
  sim.pos_state = (uint32_t)sim.pos & LZ_PBM;
  sim.prev_byte = TB((int64_t)sim.pos - 1);
}
__device__ __forceinline__ void sim_strict(uint32_t distance, int length, MS &sim, double &prob) {
  const StrictRes r = strict_factors(distance, length, sim);
  prob = prob * r.f1 * r.f2;
  strict_apply(distance, length, r, sim);
}

__device__ double test_expanded(uint32_t distance, int length, double give_up, const MS &sim) {   // :680-726
  MS v = sim; double p = 1.0;
  const int64_t copy_start = (int64_t)sim.pos - (int64_t)distance;
  // The copied bytes and the bytes at the last distance (a literal does not change it) are two runs of consecutive bytes: eight of each
  // per load instead of two dependent byte loads per literal (where eight lie inside the entry; byte by byte at its edges).
  const int64_t match_start = (int64_t)sim.pos - (int64_t)sim.rep[0] - 1;
  const uint8_t *in = s_E.in;
  const int64_t n = (int64_t)s_E.n;
  unsigned long long wb = 0, wm = 0;
  for (int x = 1; x <= length; x++) {
    const int k = (x - 1) & 7;
    if (k == 0) {
      const int64_t pb = copy_start + (x - 1), pm = match_start + (x - 1);
      if (pm >= 0 && pb + 8 <= n && pm + 8 <= n) { __builtin_memcpy(&wb, in + pb, 8); __builtin_memcpy(&wm, in + pm, 8); }
      else {
        wb = 0; wm = 0;
        for (int j = 0; j < 8 && x + j <= length; j++) { wb |= (unsigned long long)TB(pb + j) << (8 * j); wm |= (unsigned long long)TB(pm + j) << (8 * j); }
      }
    }
    const uint32_t b = (uint32_t)(wb >> (8 * k)) & 255u;
    sim_literal_bm(b, (uint32_t)(wm >> (8 * k)) & 255u, v, p);
    if (p < give_up) break;
    v.prev_byte = b;
  }
  return p;
}

__device__ inline double fmax0(double x) { return x > 0.0 ? x : 0.0; }

// largest power of two t with n * t <= 64 (n >= 1): the lanes a task gets when n tasks share the wave
__device__ inline int team_width(int n) { return n <= 1 ? 64 : n <= 2 ? 32 : n <= 4 ? 16 : n <= 8 ? 8 : n <= 16 ? 4 : n <= 32 ? 2 : 1; }

// The cuts Test_Split_DL tries for a length (:899, 924-943): cut in 2 .. length - 2 with cut or length - cut in 4 .. 9 -- at most two
// runs of consecutive values (12 cuts), listed in increasing order; nothing is kept in an array (a lane-indexed one lives in scratch).
struct Cuts { int lo1, n1, lo2, n2; };
__device__ inline Cuts cuts_of(int length) {
  Cuts c{0, 0, 0, 0};
  if (length < 6) return c;
  const int a0 = 4, a1 = length - 2 < 9 ? length - 2 : 9;                      // cut in 4 .. 9
  const int b0 = length - 9 > 2 ? length - 9 : 2, b1 = length - 4;             // length - cut in 4 .. 9
  const bool a_first = a0 <= b0;
  const int x0 = a_first ? a0 : b0, x1 = a_first ? a1 : b1, y0 = a_first ? b0 : a0, y1 = a_first ? b1 : a1;
  if (y0 <= x1 + 1) { c.lo1 = x0; c.n1 = (x1 > y1 ? x1 : y1) - x0 + 1; }
  else { c.lo1 = x0; c.n1 = x1 - x0 + 1; c.lo2 = y0; c.n2 = y1 - y0 + 1; }
  return c;
}
__device__ inline int cut_at(const Cuts &c, int k) { return k < c.n1 ? c.lo1 + k : c.lo2 + (k - c.n1); }

enum { W_STRICT = 0, W_LIT_DL = 1, W_DL_LIT = 2, W_EXPAND = 3, W_SPLIT = 4 };

template <int R> __device__ void sim_any(uint32_t distance, int length, MS &sim, double &prob);   // Simulate_any_DL_Code, recursion_limit = R

// The body of Generic_any_DL_Code (:740-832) up to its choice; NEW = new_recursion_limit.  The simulations it asks for nest at
// most three deep (the limit goes down by one per level, :756-764), so the recursion of the reference unrolls into templates.
//
// PAR: the call comes from the chain itself, which all 64 lanes walk in step (same data, same branches: the cost of one lane).
// Where the reference compares INDEPENDENT simulations -- literal + code against code + literal (:783-805), the cuts of
// Test_Split_DL (:924-943) -- the lanes part: one simulation each, from their own copy of the state, nothing written but the
// result; the results are then compared by all lanes in the reference's order.  Same doubles, a shorter critical path.
__device__ int decide_with_helpers(uint32_t distance, int length, const MS &sim, int &best_cut);
template <int NEW, bool PAR, bool HW = false> __device__ __forceinline__ int decide(uint32_t distance, int length, const MS &sim, int &best_cut, StrictRes &strict) {
  if constexpr (PAR && HW) return decide_with_helpers(distance, length, sim, best_cut);
  double strict_dlc = 0.0, expanded_dlc = 0.0, soe = 0.0;
  [[maybe_unused]] const int lane = lane_id();
  if (s_E.cv >= 1) {
#ifdef ZADA_LZ_PROF
    const unsigned long long prof_a = clock64();
#endif
    strict = strict_factors(distance, length, sim);
    strict_dlc = strict.f1 * strict.f2;                                              // (Test_Strict_DL_Code starts from 1.0: 1.0 * f1 * f2)
    expanded_dlc = test_expanded(distance, length, strict_dlc, sim);
    soe = strict_dlc > expanded_dlc ? strict_dlc : expanded_dlc;
#ifdef ZADA_LZ_PROF
    if constexpr (PAR) { g_lzprof[7] += clock64() - prof_a; g_lzprof[0] += 1; }
#endif
    if (length > 2) {
      const uint32_t b_head = TB((int64_t)sim.pos - (int64_t)distance);
      const double head_lit = test_literal_byte(b_head, sim);
      if (head_lit >= 0.875) return W_LIT_DL;                                      // Lit_then_DL_threshold :306
      MS after = sim;
      after.state = t_lit(sim.state); after.pos = sim.pos + 1; after.pos_state = (uint32_t)after.pos & LZ_PBM; after.prev_byte = b_head;
      const double malus_dtl = fmax0(0.135 - (double)distance * 1.0e-8 - (double)length * 1.0e-4);     // DL_code_then_Literal :869-889
      double dal, dtl;
      if constexpr (PAR) {
        const int task = lane >> 5;                                                // two tasks, 32 lanes each
        MS v = task == 0 ? after : sim;
        v.tw = 32;
        double p = task == 0 ? 1.0 : malus_dtl;
        sim_any<NEW>(distance, length - 1, v, p);
        if (task == 1) sim_literal(TB((int64_t)v.pos - (int64_t)distance), v, p);
        dal = __shfl(p, 0); dtl = __shfl(p, 32);
      } else if (sim.tw >= 2) {
        // the team that runs this simulation parts in two for the same pair (the halves run sim_any<NEW> in step, as the wave's halves do above)
        const int tw = sim.tw, half = tw >> 1, task = (lane & (tw - 1)) >= half ? 1 : 0, tb = lane & ~(tw - 1);
        MS v = task == 0 ? after : sim;
        v.tw = half;
        double p = task == 0 ? 1.0 : malus_dtl;
        sim_any<NEW>(distance, length - 1, v, p);
        if (task == 1) sim_literal(TB((int64_t)v.pos - (int64_t)distance), v, p);
        dal = __shfl(p, tb); dtl = __shfl(p, tb + half);
      } else {
        dal = 1.0;
        sim_any<NEW>(distance, length - 1, after, dal);
        MS v = sim;
        dtl = malus_dtl;
        sim_any<NEW>(distance, length - 1, v, dtl);
        sim_literal(TB((int64_t)v.pos - (int64_t)distance), v, dtl);
      }
      if (head_lit * dal * fmax0(0.064 - (double)distance * 1.0e-9 - (double)length * 3.0e-5) > soe) return W_LIT_DL;
      if (dtl > soe) return W_DL_LIT;
    }
    if (expanded_dlc > strict_dlc) return W_EXPAND;
  }
  if (s_E.cv >= 2) {                                                                 // Test_Split_DL :901-944
    PROF_T0;
    constexpr int LOW = NEW - 1 > 0 ? NEW - 1 : 0;
    const double malus = fmax0(0.27 - (double)distance * 2.0e-6);
    double best_prob = 0.0;
    best_cut = 2;
    if (!(malus < soe)) {
      if constexpr (PAR) {
        const Cuts cuts = cuts_of(length);                                         // cut or length - cut in 4 .. 9 (:899, 925)
        const int nc = cuts.n1 + cuts.n2;
        const int tw = team_width(nc), task = lane / tw;                           // one cut per team of tw lanes
        double pm = 0.0, pf = 0.0;
        if (task < nc) {
          const int cut = cut_at(cuts, task);
          double p = malus;
          MS v = sim;
          v.tw = tw;
          sim_any<LOW>(distance, cut, v, p);
          pm = p; pf = p;
          if (!(p <= soe)) { sim_any<LOW>(distance, length - cut, v, p); pf = p; }
        }
        for (int k = 0; k < nc; k++) {
          const double pmk = __shfl(pm, k * tw), pfk = __shfl(pf, k * tw);
          if (!(pmk <= soe)) { if (pfk > best_prob) { best_prob = pfk; best_cut = cut_at(cuts, k); } }
        }
      } else if (sim.tw > 1) {
        // A team of sim.tw lanes runs this simulation in step (same state, same branches).  Its cuts are independent simulations
        // again: a lane each, tw at a time; the results go round the team by cross-lane reads and are taken in cut order.
        const int tw = sim.tw, tl = lane & (tw - 1), tb = lane & ~(tw - 1);
        const Cuts cuts = cuts_of(length);
        const int nc = cuts.n1 + cuts.n2;
        for (int r = 0; r * tw < nc; r++) {
          const int k = r * tw + tl;
          double pm = 0.0, pf = 0.0;
          if (k < nc) {
            const int cut = cut_at(cuts, k);
            double p = malus;
            MS v = sim;
            v.tw = 1;
            sim_any<LOW>(distance, cut, v, p);
            pm = p; pf = p;
            if (!(p <= soe)) { sim_any<LOW>(distance, length - cut, v, p); pf = p; }
          }
          for (int j = 0; j < tw && r * tw + j < nc; j++) {
            const double pmk = __shfl(pm, tb + j), pfk = __shfl(pf, tb + j);
            if (!(pmk <= soe)) { if (pfk > best_prob) { best_prob = pfk; best_cut = cut_at(cuts, r * tw + j); } }
          }
        }
      } else {
        for (int cut = 2; cut <= length - 2; cut++) {
          const int rest = length - cut;
          if ((cut >= 4 && cut <= 9) || (rest >= 4 && rest <= 9)) {
            double p = malus;
            MS v = sim;
            sim_any<LOW>(distance, cut, v, p);
            if (!(p <= soe)) {
              sim_any<LOW>(distance, rest, v, p);
              if (p > best_prob) { best_prob = p; best_cut = cut; }
            }
          }
        }
      }
    }
#ifdef ZADA_LZ_PROF
    if constexpr (PAR) g_lzprof[6] += clock64() - prof_t0;
#endif
    if (best_prob > soe) return W_SPLIT;
  }
  return W_STRICT;
}

// (the state and the probability go in and come back BY VALUE -- 14 registers -- instead of through references into scratch)
struct SimRes { MS sim; double prob; };
template <int R> __device__ __noinline__ SimRes sim_any_impl(uint32_t distance, int length, MS sim, double prob) {
  if constexpr (R - 1 < 0) {
    sim_strict(distance, length, sim, prob);
  } else {
    constexpr int NEW = R - 1;
    int cut = 2;
    StrictRes strict;
    const bool tested = s_E.cv >= 1;                                               // (Level_0 / Level_1 never come here: their codes are written strictly)
    switch (decide<NEW, false>(distance, length, sim, cut, strict)) {
      case W_LIT_DL:
        sim_literal(TB((int64_t)sim.pos - (int64_t)distance), sim, prob);
        sim_any<NEW>(distance, length - 1, sim, prob);
        break;
      case W_DL_LIT:
        sim_any<NEW>(distance, length - 1, sim, prob);
        sim_literal(TB((int64_t)sim.pos - (int64_t)distance), sim, prob);
        break;
      case W_EXPAND:
        for (int x = 1; x <= length; x++) sim_literal(TB((int64_t)sim.pos - (int64_t)distance), sim, prob);
        break;
      case W_SPLIT:
        sim_any<NEW>(distance, cut, sim, prob);
        sim_any<NEW>(distance, length - cut, sim, prob);
        break;
      default:
        if (tested) { prob = prob * strict.f1 * strict.f2; strict_apply(distance, length, strict, sim); }
        else sim_strict(distance, length, sim, prob);
    }
  }
  return SimRes{sim, prob};
}
template <int R> __device__ __forceinline__ void sim_any(uint32_t distance, int length, MS &sim, double &prob) {
  if constexpr (R - 1 < 0) sim_strict(distance, length, sim, prob);      // limit used up (:761-764): no call in between
  else { const SimRes r = sim_any_impl<R>(distance, length, sim, prob); sim = r.sim; prob = r.prob; }
}

// ---------------------------------------------------------------- range coder :964-1039

__device__ inline void put_byte(uint32_t b) { if (s_E.olen < s_E.cap) s_E.out[s_E.olen] = (uint8_t)b; s_E.olen++; }

__device__ __noinline__ void shift_low() {
  const uint64_t top = s_E.low >> 32;
  const uint32_t bottom = (uint32_t)s_E.low;
  if (bottom < 0xFF000000u || top != 0) {
    uint32_t temp = s_E.cache;
    const uint32_t carry = (uint32_t)top & 0xFF;
    do { put_byte((temp + carry) & 0xFF); temp = 0xFF; s_E.cache_size--; } while (s_E.cache_size != 0);
    s_E.cache = (bottom >> 24) & 0xFF;
  }
  s_E.cache_size++;
  s_E.low = (uint64_t)(uint32_t)(bottom << 8);
}

__device__ inline void normalize() { if (s_E.width < (1u << 24)) { s_E.width <<= 8; shift_low(); } }

__device__ inline void encode_bit(uint16_t &prob, uint32_t symbol) {
  const uint32_t cur = prob, bound = (s_E.width >> 11) * cur;
  if (symbol == 0) { s_E.width = bound; normalize(); prob = (uint16_t)(cur + ((2048 - cur) >> 5)); }
  else { s_E.low += bound; s_E.width -= bound; normalize(); prob = (uint16_t)(cur - (cur >> 5)); }
}

__device__ __forceinline__ void bit_tree_encode(uint16_t *prob, int num_bits, uint32_t symbol) {
  uint32_t m = 1;
  for (int i = num_bits - 1; i >= 0; i--) { const uint32_t bit = (symbol >> i) & 1; encode_bit(prob[m], bit); m = 2 * m + bit; }
}
__device__ __forceinline__ void bit_tree_rev_encode(uint16_t *prob, int num_bits, uint32_t symbol) {
  uint32_t m = 1;
  for (int c = num_bits; c >= 1; c--) { const uint32_t bit = symbol & 1; encode_bit(prob[m], bit); m = 2 * m + bit; symbol >>= 1; }
}

// ---------------------------------------------------------------- the machine :1045-1361

__device__ __noinline__ void emit_literal(uint32_t b) {                   // LZ77_emits_literal_byte :1097-1130
  PROF_T0;
  LzProbs &P = s_P;
  MS &S = s_E.ES;
  const int idx = lit_idx(S.prev_byte);
  const uint32_t b_match = TB((int64_t)S.pos - (int64_t)S.rep[0] - 1);
  if (b == b_match && S.pos > (uint64_t)(uint32_t)(S.rep[0] + 1) &&
      (s_E.cv == 0 || test_short_rep(S) > test_simple_literal(b, b_match, idx, S))) {
    encode_bit(P.match[S.state][S.pos_state], 1);
    encode_bit(P.rep[S.state], 1);
    encode_bit(P.g0[S.state], 0);
    encode_bit(P.rep0_long[S.state][S.pos_state], 0);
    S.state = t_srep(S.state);
  } else {
    encode_bit(P.match[S.state][S.pos_state], 0);
    uint16_t *prob = P.lit + idx;
    uint32_t symb = b | 0x100;
    if (S.state < 7) {
      do { encode_bit(prob[symb >> 8], (symb >> 7) & 1); symb <<= 1; } while (symb < 0x10000);
    } else {
      uint32_t offs = 0x100, match = b_match;
      do {
        match <<= 1;
        encode_bit(prob[offs + (match & offs) + (symb >> 8)], (symb >> 7) & 1);
        symb <<= 1;
        offs &= ~(match ^ symb);
      } while (symb < 0x10000);
    }
    S.state = t_lit(S.state);
  }
  S.pos += 1;
  S.pos_state = (uint32_t)S.pos & LZ_PBM;
  S.prev_byte = b;
  PROF_ADD(1);
}

__device__ void encode_length(bool rep, uint32_t length) {            // :1160-1181
  LenProbs &pl = rep ? s_P.rep_len : s_P.len;
  uint32_t len = length - 2;
  const uint32_t ps = s_E.ES.pos_state;
  if (len < 8) { encode_bit(pl.c1, 0); bit_tree_encode(pl.low[ps], 3, len); }
  else {
    encode_bit(pl.c1, 1); len -= 8;
    if (len < 8) { encode_bit(pl.c2, 0); bit_tree_encode(pl.mid[ps], 3, len); }
    else { encode_bit(pl.c2, 1); len -= 8; bit_tree_encode(pl.high, 8, len); }
  }
}

__device__ __noinline__ void write_simple_match(uint32_t dist_ip, uint32_t length) {   // :1183-1255
  LzProbs &P = s_P;
  MS &S = s_E.ES;
  encode_bit(P.rep[S.state], 0);
  S.state = t_match(S.state);
  encode_length(false, length);
  const uint32_t len_state = length - 2 < 3 ? length - 2 : 3, ds = dist_slot(dist_ip);
  bit_tree_encode(P.slot[len_state], 6, ds);
  if (ds >= 4) {
    const int footer = (int)(ds >> 1) - 1;
    const uint32_t base = (2 | (ds & 1)) << footer, red = dist_ip - base;
    if (ds < 14) bit_tree_rev_encode(P.pos + ((int)base - (int)ds - 1) + 1, footer, red);
    else {
      const uint32_t value = red >> 4;
      for (int i = footer - 4 - 1; i >= 0; i--) {                                  // Encode_Direct_Bits :1205-1215
        s_E.width >>= 1;
        s_E.low += (uint64_t)s_E.width & (0 - (uint64_t)((value >> i) & 1));
        normalize();
      }
      bit_tree_rev_encode(P.align, 4, red & 15);
    }
  }
  S.rep[3] = S.rep[2]; S.rep[2] = S.rep[1]; S.rep[1] = S.rep[0]; S.rep[0] = dist_ip;
}

__device__ __noinline__ void write_repeat_match(int index_rm, uint32_t length) {   // :1257-1286
  LzProbs &P = s_P;
  MS &S = s_E.ES;
  encode_bit(P.rep[S.state], 1);
  switch (index_rm) {
    case 0: encode_bit(P.g0[S.state], 0); encode_bit(P.rep0_long[S.state][S.pos_state], 1); break;
    case 1: encode_bit(P.g0[S.state], 1); encode_bit(P.g1[S.state], 0); break;
    case 2: encode_bit(P.g0[S.state], 1); encode_bit(P.g1[S.state], 1); encode_bit(P.g2[S.state], 0); break;
    default: encode_bit(P.g0[S.state], 1); encode_bit(P.g1[S.state], 1); encode_bit(P.g2[S.state], 1); break;
  }
  const uint32_t aux = S.rep[index_rm];
  for (int i = index_rm; i >= 1; i--) S.rep[i] = S.rep[i - 1];
  S.rep[0] = aux;
  encode_length(true, length);
  S.state = t_rep(S.state);
}

__device__ __noinline__ void write_strict(uint32_t distance, int length) {   // Write_Strict_DL_Code :1288-1328
  MS &S = s_E.ES;
  const uint32_t dist_ip = distance - 1;
  int found = -1;
  encode_bit(s_P.match[S.state][S.pos_state], 1);
  for (int i = 0; i < 4; i++) if (dist_ip == S.rep[i]) { found = i; break; }
  if (found >= 0 && (s_E.cv == 0 || test_repeat_match(found, (uint32_t)length, S) >= test_simple_match(dist_ip, (uint32_t)length, S) * 0.55))
    write_repeat_match(found, (uint32_t)length);
  else
    write_simple_match(dist_ip, (uint32_t)length);
  S.pos += (uint64_t)length;
  S.pos_state = (uint32_t)S.pos & LZ_PBM;
  S.prev_byte = TB((int64_t)S.pos - 1);
}

// LZ77_emits_DL_code :1355-1361 = Write_any_DL_code (..., ES, max_recursion): the writing instance of Generic_any_DL_Code does not
// lower its limit (:756-760), so its own recursion can go a match length deep; it is a work list here.  An item is a length
// still to be written at `distance`, or the literal that follows a shortened match (:797-805).
__shared__ uint16_t s_work[2 * 280];                                   // (every lane pushes and pops the same items: one copy in LDS, not 64 in scratch)
// (inlined at its one call site, the kernel's main loop: as a function of its own it saved and restored 42 registers per call)
template <bool HW> __device__ __forceinline__ void emit_dl(uint32_t distance, int length0) {
  PROF_T0;
  constexpr uint16_t POST_LIT = 0xFFFF;
  uint16_t *stack = s_work;
  int sp = 0;
  stack[sp++] = (uint16_t)length0;
  while (sp > 0) {
    const uint16_t it = stack[--sp];
    if (it == POST_LIT) { emit_literal(TB((int64_t)s_E.ES.pos - (int64_t)distance)); continue; }
    const int length = it;
    int cut = 2;
    StrictRes strict;
    switch (decide<2, true, HW>(distance, length, s_E.ES, cut, strict)) {
      case W_LIT_DL:
        emit_literal(TB((int64_t)s_E.ES.pos - (int64_t)distance));
        stack[sp++] = (uint16_t)(length - 1);
        break;
      case W_DL_LIT:
        stack[sp++] = POST_LIT;
        stack[sp++] = (uint16_t)(length - 1);
        break;
      case W_EXPAND:
        for (int x = 1; x <= length; x++) emit_literal(TB((int64_t)s_E.ES.pos - (int64_t)distance));
        break;
      case W_SPLIT:
        stack[sp++] = (uint16_t)(length - cut);
        stack[sp++] = (uint16_t)cut;
        break;
      default:
        write_strict(distance, length);
    }
  }
  PROF_ADD(2);
}

// ---------------------------------------------------------------- Estimate_DL_Codes_for_LZ77 :1363-1498

// What Scoring shares between its levels (:1370-1384): the same for every lane, so in LDS once (as a local passed by reference it
// lived in every lane's scratch memory).
struct ScoreCtx { int old_index, last_pos_any; MS sim_new; double head_lit_prob; };
__shared__ ScoreCtx s_S;

__device__ double scoring2(MS state, int start);

// One candidate of Scoring (:1404-1468): the probability of the message that starts with match i of set m.
template <int LEVEL> __device__ __forceinline__ double score_candidate(const MS &state, int start, int m, int i) {
  const Matches &M = s_MM[m];
  const int old_index = s_S.old_index;
  const int mlen = M.len[i], last_pos_i = mlen + (m != old_index ? 1 : 0);
  MS t; double p;
  if (m != old_index && start == 1) { t = s_S.sim_new; p = s_S.head_lit_prob; } else { t = state; p = 1.0; }
  t.tw = state.tw;
  int trunc;
  if (m == old_index) trunc = mlen - start + 1;
  else if (start == 1) trunc = mlen;
  else trunc = mlen - start + 2;
  if (trunc == 1) sim_literal(TB((int64_t)state.pos), t, p);
  else sim_any<1>((uint32_t)M.dist[i], trunc, t, p);
  if constexpr (LEVEL < 2) {
    if (last_pos_i < s_S.last_pos_any) p = p * scoring2(t, last_pos_i + 1);
  }
  return p;
}

// Scoring at recursion level 2 (:1385-1469; only the probability of its best candidate is used, :1459-1464).  It runs on the team of
// state.tw lanes that scores one candidate of level 1, in step.  Its own candidates -- the matches that reach last_pos_any (:1421-1424)
// -- are independent simulations again: the team parts into sub-teams, one candidate each, as many at a time as there are sub-teams,
// and the results are reduced over the team (a maximum: Scoring keeps the first strict maximum, and nothing but its value is used).
__device__ inline bool score2_takes(int m, int i, int start) {
  const int last_pos_i = s_MM[m].len[i] + (m != s_S.old_index ? 1 : 0);
  return last_pos_i >= start && last_pos_i >= s_S.last_pos_any;
}
__device__ __noinline__ double scoring2(MS state, int start) {
  const int tw = state.tw, lane = lane_id(), tl = lane & (tw - 1);
  const int c0 = s_MM[0].count, total = c0 + s_MM[1].count;
  int nq = 0;
  for (int k = 0; k < total; k++) nq += score2_takes(k < c0 ? 0 : 1, (k < c0 ? k : k - c0) + 1, start) ? 1 : 0;
  int stw = tw;                                                      // sub-team width: the largest power of two with nq * stw <= tw, at least 1
  while (stw > 1 && nq * stw > tw) stw >>= 1;
  const int groups = tw / stw, g = tl / stw;
  double best = 0.0;
  for (int r = 0; r * groups < nq; r++) {
    const int mine = r * groups + g;
    double p = 0.0;
    if (mine < nq) {
      int seen = 0, km = 0;
      for (int k = 0; k < total; k++)
        if (score2_takes(k < c0 ? 0 : 1, (k < c0 ? k : k - c0) + 1, start)) { if (seen == mine) km = k; seen++; }
      MS st = state;
      st.tw = stw;
      p = score_candidate<2>(st, start, km < c0 ? 0 : 1, (km < c0 ? km : km - c0) + 1);
    }
    for (int off = stw; off < tw; off <<= 1) { const double o = __shfl_xor(p, off); p = o > p ? o : p; }     // (the lanes of a sub-team hold the same p)
    if (p > best) best = p;
  }
  return best;
}


// ---------------------------------------------------------------- one stream on four waves ("helpers")
//
// A stream alone on the chip (zada_lzma: config 4's shape) leaves 255 CUs and three of its own CU's SIMDs idle while ONE wave walks the
// chain.  Teams of lanes shorten the chain only where the simulations they run take the same path -- a wave executes the union of its
// lanes' paths; DIFFERENT simulations side by side need different waves.  With blockDim.x = 256 wave 0 walks the chain as ever and posts
// its two forks -- the ways of writing a code (decide: the literal / shortened-code pair and the cuts, :783-805, 924-943) and Scoring's
// candidates (:1404-1468) -- in LDS; waves 1 .. 3 wait at a barrier, take their shares (a whole wave per simulation of the pair, the cuts
// and the candidates dealt over the waves: wider teams inside them as well), leave the results in LDS and wait again.  The model, the match
// sets and the coder's state are LDS objects of the workgroup already; the helpers only read them.  Same doubles, compared by wave 0 in the
// reference's order.
struct HelpArgs {
  int cmd;                                       // 0: the stream is over (or parked), 1: decide, 2: Scoring
  uint32_t distance; int length, pair, nc;
  Cuts cuts;
  MS sim, after;
  double malus_dtl, malus, soe;
  int total;
};
__shared__ HelpArgs s_H;
__shared__ double s_Hp[2 * LZ_MAXM + 2][2];      // decide: [0] / [1] the pair, [2 + k] cut k (after its first part, at its end); Scoring: [k] candidate k
#ifndef ZADA_HELP_WAVES
#define ZADA_HELP_WAVES 4
#endif
constexpr int HELP_WAVES = ZADA_HELP_WAVES;

// the cuts [k0, k1) of the posted decision, a team each
__device__ void help_cuts(int k0, int k1) {
  const int n = k1 - k0, lane = lane_id();
  if (n <= 0) return;
  const int tw = team_width(n), task = lane / tw;
  if (task < n) {
    const int cut = cut_at(s_H.cuts, k0 + task);
    double p = s_H.malus;
    MS v = s_H.sim;
    v.tw = tw;
    sim_any<1>(s_H.distance, cut, v, p);
    const double pm = p;
    // (the reference goes on only if pm > the strict / expanded probability, :933: that one is still being worked out by wave 0 -- both parts
    // here, the test when wave 0 collects)
    sim_any<1>(s_H.distance, s_H.length - cut, v, p);
    if ((lane & (tw - 1)) == 0) { s_Hp[2 + k0 + task][0] = pm; s_Hp[2 + k0 + task][1] = p; }
  }
}
// wave w's share of the posted decision: with a pair, waves 1 and 2 take its two simulations and waves 0 and 3 the cuts; without, all four the cuts
__device__ void help_decide(int w) {
  const int nc = s_H.nc;
  if (s_H.pair) {
    if (w == 1 || w == 2) {
      const int t = w - 1;
      MS v = t == 0 ? s_H.after : s_H.sim;
      v.tw = 64;
      double p = t == 0 ? 1.0 : s_H.malus_dtl;
      sim_any<2>(s_H.distance, s_H.length - 1, v, p);
      if (t == 1) sim_literal(TB((int64_t)v.pos - (int64_t)s_H.distance), v, p);
      if (lane_id() == 0) s_Hp[t][0] = p;
    } else {
      const int j = w == 0 ? 0 : w - 2, m = HELP_WAVES - 2;            // (the waves that are not at the pair)
      help_cuts(nc * j / m, nc * (j + 1) / m);
    }
  } else help_cuts(nc * w / HELP_WAVES, nc * (w + 1) / HELP_WAVES);
}
// wave w's share of Scoring's candidates: w, w + 4, ...
__device__ void help_score(int w) {
  const int lane = lane_id(), c0 = s_MM[0].count, total = s_H.total;
  const int mine = total > w ? (total - w + HELP_WAVES - 1) / HELP_WAVES : 0;
  if (mine == 0) return;
  const int tw = team_width(mine), per_round = 64 / tw;
  MS st = s_H.sim;
  st.tw = tw;
  for (int base = 0; base < mine; base += per_round) {
    const int idx = base + lane / tw;
    if (idx < mine) {
      const int k = w + HELP_WAVES * idx;
      const double p = score_candidate<1>(st, 1, k < c0 ? 0 : 1, (k < c0 ? k : k - c0) + 1);
      if ((lane & (tw - 1)) == 0) s_Hp[k][0] = p;
    }
  }
}
__device__ void helper_loop(int w) {
  for (;;) {
    __syncthreads();                                                   // (A) a fork is posted
    const int cmd = s_H.cmd;
    if (cmd == 0) return;
    if (cmd == 1) help_decide(w); else help_score(w);
    __syncthreads();                                                   // (B) the shares are done
  }
}
template <bool HW> __device__ __forceinline__ void helpers_release() {                  // wave 0, before it leaves the kernel
  if constexpr (HW) {
    if (lane_id() == 0) s_H.cmd = 0;
    __syncthreads();
  }
}

// decide <2, true> of the wave that walks the chain, its independent simulations on the helpers
__device__ int decide_with_helpers(uint32_t distance, int length, const MS &sim, int &best_cut) {
  const bool pair = length > 2;
  double head_lit = 0.0, malus_dtl = 0.0;
  MS after = sim;
  if (pair) {
    const uint32_t b_head = TB((int64_t)sim.pos - (int64_t)distance);
    head_lit = test_literal_byte(b_head, sim);
    if (head_lit >= 0.875) return W_LIT_DL;                                          // Lit_then_DL_threshold :306
    after.state = t_lit(sim.state); after.pos = sim.pos + 1; after.pos_state = (uint32_t)after.pos & LZ_PBM; after.prev_byte = b_head;
    malus_dtl = fmax0(0.135 - (double)distance * 1.0e-8 - (double)length * 1.0e-4);  // DL_code_then_Literal :869-889
  }
  const double malus = fmax0(0.27 - (double)distance * 2.0e-6);                      // Test_Split_DL :901-944
  const Cuts cuts = cuts_of(length);
  const int nc = s_E.cv >= 2 && malus > 0.0 ? cuts.n1 + cuts.n2 : 0;                 // (malus = 0: no cut can beat anything)
  const bool fork = pair || nc > 0;
  if (fork) {
    if (lane_id() == 0) {
      s_H.cmd = 1; s_H.distance = distance; s_H.length = length; s_H.pair = pair ? 1 : 0; s_H.nc = nc; s_H.cuts = cuts;
      s_H.sim = sim; s_H.after = after; s_H.malus_dtl = malus_dtl; s_H.malus = malus; s_H.soe = 0.0;
    }
    __syncthreads();                                                   // (A)
  }
  // the strict and the expanded code (:661-726) on this wave while the helpers are at the pair
  const StrictRes strict = strict_factors(distance, length, sim);
  const double strict_dlc = strict.f1 * strict.f2;
  const double expanded_dlc = test_expanded(distance, length, strict_dlc, sim);
  const double soe = strict_dlc > expanded_dlc ? strict_dlc : expanded_dlc;
  if (fork) {
    help_decide(0);
    __syncthreads();                                                   // (B)
  }
  const bool split = !(malus < soe);
  if (pair) {
    const double dal = s_Hp[0][0], dtl = s_Hp[1][0];
    if (head_lit * dal * fmax0(0.064 - (double)distance * 1.0e-9 - (double)length * 3.0e-5) > soe) return W_LIT_DL;
    if (dtl > soe) return W_DL_LIT;
  }
  if (expanded_dlc > strict_dlc) return W_EXPAND;
  double best_prob = 0.0;
  best_cut = 2;
  for (int k = 0; split && k < nc; k++) {
    const double pmk = s_Hp[2 + k][0], pfk = s_Hp[2 + k][1];
    if (!(pmk <= soe)) { if (pfk > best_prob) { best_prob = pfk; best_cut = cut_at(cuts, k); } }
  }
  if (best_prob > soe) return W_SPLIT;
  return W_STRICT;
}

// scoring_top of the wave that walks the chain, the candidates dealt over the four waves
__device__ void scoring_with_helpers(const MS &state, int &index, int &match_set) {
  const int c0 = s_MM[0].count, total = c0 + s_MM[1].count;
  if (lane_id() == 0) { s_H.cmd = 2; s_H.sim = state; s_H.total = total; }
  __syncthreads();                                                     // (A)
  help_score(0);
  __syncthreads();                                                     // (B)
  double prob = 0.0;
  for (int kk = 0; kk < total; kk++) {
    const double pj = s_Hp[kk][0];
    if (pj > prob) { prob = pj; index = (kk < c0 ? kk : kk - c0) + 1; match_set = kk < c0 ? 0 : 1; }
  }
}

// Scoring at level 1, start 1, called from the chain (all lanes in step): every match of both sets is a candidate (their
// last positions are >= 1), a team of lanes each; the best is then picked by all lanes in the reference's order (first strict maximum).
template <bool HW> __device__ __forceinline__ void scoring_top(const MS &state, int &index, int &match_set) {
  if constexpr (HW) { scoring_with_helpers(state, index, match_set); return; }
  const int lane = lane_id(), c0 = s_MM[0].count, total = c0 + s_MM[1].count;
  const int tw = team_width(total), per_round = 64 / tw;                           // a team of tw lanes per candidate
  MS st = state;
  st.tw = tw;
  double prob = 0.0;
  for (int base = 0; base < total; base += per_round) {
    const int k = base + lane / tw;
    double p = 0.0;
    if (k < total) p = score_candidate<1>(st, 1, k < c0 ? 0 : 1, (k < c0 ? k : k - c0) + 1);
    const int cnt = total - base < per_round ? total - base : per_round;
    for (int j = 0; j < cnt; j++) {
      const double pj = __shfl(p, j * tw);
      const int kk = base + j;
      if (pj > prob) { prob = pj; index = (kk < c0 ? kk : kk - c0) + 1; match_set = kk < c0 ? 0 : 1; }
    }
  }
}

template <bool HW> __device__ void estimate_dl_codes(int old_index, uint32_t prefix1, int &best_index, int &best_set) {
  PROF_T0;
  int last_pos_any = 0;
  for (int m = 0; m <= 1; m++)
    for (int i = 1; i <= s_MM[m].count; i++) {
      const int lp = s_MM[m].len[i] + (m != old_index ? 1 : 0);
      if (lp > last_pos_any) last_pos_any = lp;
    }
  MS sim_new = s_E.ES;
  double head_lit_prob = 1.0;
  sim_literal(prefix1, sim_new, head_lit_prob);
  s_S.old_index = old_index; s_S.last_pos_any = last_pos_any; s_S.sim_new = sim_new; s_S.head_lit_prob = head_lit_prob;
  chain_sync<HW>();
  best_index = 1; best_set = old_index;
  const MS sim_old = s_E.ES;
  scoring_top<HW>(sim_old, best_index, best_set);
  PROF_ADD(3);
}

// ---------------------------------------------------------------- LZ77_using_BT4 (lz77.adb:953-1827)

// What the chain keeps of BT4: the window bookkeeping (read / write positions, pending bytes: they decide how far matches and
// repeat matches may reach and when the window is filled) and the state of Get_Next_Symbol.  The hash tables and the trees are the
// producer's (zada_bt4.hip): Read_One_and_Get_Matches fetches the position's set from HBM, Skip only moves on.
struct BT4 {
  int sbs, readPos, readLimit, writePos, pendingSize;
  int keepSizeBefore, keepSizeAfter, buf_len;
  int64_t moved;                                   // sum of the window moves: buf (i) = in [i + moved]
  uint64_t in_pos;
  uint64_t base;                                   // arena position of the entry's first byte: the match sets are indexed by arena position
  Bt4Sets sets;
  int readAhead, rep_dist[4], len_rep[4], best_len_rep, best_rep_index;
  int cur;                                         // current_match_index
  uint32_t cur_literal;
};
__shared__ BT4 s_B;
constexpr int BT_LOOK = BT4_LOOK, BT_NICE = BT4_NICE, BT_MIN = 2, BT_OPTS = BT4_OPTS;
#define BUF(i) ((uint32_t)s_E.in[(int64_t)(i) + s_B.moved])

// First index k in [len, limit) at which buf [a + k] /= buf [b + k], or limit: the byte loop of lz77.adb:1456-1458, eight bytes at
// a time while eight remain below the limit (nothing beyond a + limit / b + limit is read).
__device__ inline int bt_extend(const uint8_t *buf, int64_t a, int64_t b, int len, int limit) {
  while (len + 8 <= limit) {
    unsigned long long x, y;
    __builtin_memcpy(&x, buf + a + len, 8);
    __builtin_memcpy(&y, buf + b + len, 8);
    x ^= y;
    if (x) return len + (__builtin_ctzll(x) >> 3);
    len += 8;
  }
  while (len < limit && buf[a + len] == buf[b + len]) len++;
  return len;
}

__device__ inline int bt_available() { return s_B.writePos - s_B.readPos - 1; }

// Move_Pos_in_BT4 :1127-1150 (finishing = False, :959) without lzPos / cyclicPos: the producer counts the inserted positions itself
__device__ inline int bt_move_pos() {
  s_B.readPos++;
  int avail = bt_available();
  if (avail < BT_NICE) { s_B.pendingSize++; avail = 0; }
  return avail;
}

__device__ void bt_skip(int len) {                           // BT4_Algo.Skip :1208-1232: the tree update is the producer's
  for (int count = len; count >= 1; count--) bt_move_pos();
}

// BT4_Algo.Read_One_and_Get_Matches :1234-1361: the set the producer found for this position (none for a pending position: the
// producer's schedule says so as well, cnt = 0), one match per lane
template <bool HW> __device__ __noinline__ void bt_get_matches(int set) {
  Matches &M = s_MM[set];
  M.count = 0;
  const int avail = bt_move_pos();
  if (avail == 0) return;
  const uint64_t p = s_B.base + (uint64_t)((int64_t)s_B.readPos + s_B.moved);
  // the count and the eight slots next to the position in ONE round trip (slots beyond the count hold nothing, and are not used); only a set of
  // more than seven matches takes a second one, to its overflow block
  const int i = lane_id();
  const Bt4Sets &S = s_B.sets;
  uint32_t l = 0, d = 0;
  if (i < BT4_INLINE) { l = S.sl[p * BT4_INLINE + i]; d = S.sd[p * BT4_INLINE + i]; }
  const int cnt = S.cnt[p];
  const uint32_t blk = __shfl(d, BT4_INLINE - 1);
  if (i < cnt) {
    if (i >= BT4_INLINE - 1) { const uint64_t o = (uint64_t)blk * BT4_OVF + (uint32_t)(i - (BT4_INLINE - 1)); l = S.ol[o]; d = S.od[o]; }
    M.len[i + 1] = (uint16_t)l; M.dist[i + 1] = (int)d;
  }
  M.count = cnt;
  if (s_E.verify) {
    // (a stream that reads positions behind a gap of never-inserted ones: the reference's distances into the text before the gap are short by
    // the gap and its hash-2 / hash-3 matches are not compared beyond their first byte -- a match that is none ends the stream, zada_bt4.h)
    bool bad = false;
    if (i < cnt) bad = bt_extend(s_E.in + s_B.moved, (int64_t)s_B.readPos - (int64_t)d, s_B.readPos, 0, (int)l) < (int)l;
    if (__any(bad)) s_E.defect = 1;
  }
  chain_sync<HW>();
}

__device__ int bt_fill_window(int len_initial) {             // Fill_Window :1389-1440, Move_Window :1375-1386
  int len = len_initial;
  if (s_B.readPos >= s_B.buf_len - s_B.keepSizeAfter) {
    const int moveOffset = ((s_B.readPos + 1 - s_B.keepSizeBefore) / 16) * 16;
    s_B.moved += moveOffset;
    s_B.readPos -= moveOffset; s_B.readLimit -= moveOffset; s_B.writePos -= moveOffset;
  }
  if (len > s_B.buf_len - s_B.writePos) len = s_B.buf_len - s_B.writePos;
  const uint64_t left = s_E.n - s_B.in_pos;
  const int actual = (uint64_t)len < left ? len : (int)left;
  s_B.writePos += actual; s_B.in_pos += (uint64_t)actual;
  if (s_B.writePos >= s_B.keepSizeAfter) s_B.readLimit = s_B.writePos - s_B.keepSizeAfter;
  if (s_B.pendingSize > 0 && s_B.readPos < s_B.readLimit) {                               // processPendingBytes :1397-1406
    const int old = s_B.pendingSize;
    s_B.readPos -= s_B.pendingSize;
    s_B.pendingSize = 0;
    bt_skip(old);
  }
  return actual;
}

__device__ inline int bt_match_len(int distance, int limit) {   // Compute_Match_Length :1442-1460
  if (distance < 2) return 0;
  return bt_extend(s_E.in + s_B.moved, (int64_t)s_B.readPos - distance, s_B.readPos, 0, limit);
}
__device__ inline bool much_smaller(int smallDist, int bigDist) { return (smallDist - 1) < (bigDist - 1) / 128; }   // :1469-1473

template <bool HW> __device__ void lz_read_one(int set) {                     // Read_One_and_Get_Matches :1477-1503
  PROF_T0;
  s_B.readAhead++;
  bt_get_matches<HW>(set);
  s_B.best_len_rep = 0;
  const int a = bt_available(), avail = a < BT_LOOK ? a : BT_LOOK;
  if (avail >= BT_MIN) {
    // (the four repeat distances on four lanes: their byte loads are in flight together instead of one after the other)
    const int mine = bt_match_len(s_B.rep_dist[threadIdx.x & 3], avail);
    for (int rep = 0; rep < 4; rep++) {
      const int len = __shfl(mine, rep);
      s_B.len_rep[rep] = len;
      if (len > s_B.best_len_rep) { s_B.best_rep_index = rep; s_B.best_len_rep = len; }
    }
  } else {
    for (int rep = 0; rep < 4; rep++) s_B.len_rep[rep] = 0;
  }
  PROF_ADD(4);
}

__device__ void lz_supplement(int set) {
  Matches &M = s_MM[set];                           // Get_supplemental_Matches_from_Repeat_Matches :1505-1566
  if (M.count == 0 && s_B.best_len_rep >= BT_MIN) { M.dist[1] = s_B.rep_dist[s_B.best_rep_index]; M.len[1] = (uint16_t)s_B.best_len_rep; M.count = 1; }
  for (int rep = 0; rep < 4; rep++) {
    const int len = s_B.len_rep[rep];
    if (len < BT_MIN) continue;
    int ins = 0;
    for (int i = M.count; i >= 1; i--) {
      if (len == M.len[i]) {
        if (s_B.rep_dist[rep] != M.dist[i]) { ins = much_smaller(M.dist[i], s_B.rep_dist[rep]) ? i : i + 1; break; }
      } else if (i < M.count) {
        if (len > M.len[i] && len < M.len[i + 1]) { ins = i + 1; break; }
      } else if (len > M.len[i]) { ins = i + 1; break; }
    }
    if (ins > 0) {
      for (int i = M.count; i >= ins; i--) { M.dist[i + 1] = M.dist[i]; M.len[i + 1] = M.len[i]; }
      M.dist[ins] = s_B.rep_dist[rep]; M.len[ins] = (uint16_t)len;
      M.count++;
      break;
    }
  }
}

__device__ inline void lz_reduce(int set) {
  Matches &m = s_MM[set];                                       // Reduce_consecutive_max_lengths :1575-1585
  while (m.count > 1 && m.len[m.count] == m.len[m.count - 1] + 1 && much_smaller(m.dist[m.count - 1], m.dist[m.count])) m.count--;
}

// What Get_Next_Symbol decided for the next symbol.  The emission itself (LZ77_emits_DL_code / LZ77_emits_literal_byte, the calls of
// Send_DL_code :1629 and Send_first_literal_of_match :1622) happens in the kernel's main loop, at ONE inlined site; nothing that
// Send_DL_code does after the emission reads the coder's state, so the matcher's bookkeeping is done here, before.
struct Symbol { int length; uint32_t distance; uint32_t literal; };  // length 0: a literal
__device__ __forceinline__ Symbol lz_send_dl(int distance, int length) {               // Send_DL_code :1627-1659
  s_B.readAhead -= length;
  int found = -1;
  for (int i = 0; i < 4; i++) if (distance == s_B.rep_dist[i]) { found = i; break; }
  if (found >= 0) {
    const int aux = s_B.rep_dist[found];
    for (int i = found; i >= 1; i--) s_B.rep_dist[i] = s_B.rep_dist[i - 1];
    s_B.rep_dist[0] = aux;
  } else {
    s_B.rep_dist[3] = s_B.rep_dist[2]; s_B.rep_dist[2] = s_B.rep_dist[1]; s_B.rep_dist[1] = s_B.rep_dist[0]; s_B.rep_dist[0] = distance;
  }
  return Symbol{length, (uint32_t)distance, 0u};
}
__device__ __forceinline__ Symbol lz_send_literal() { s_B.readAhead--; return Symbol{0, 0u, s_B.cur_literal}; }
__device__ inline void lz_skip(int len) { PROF_T0; s_B.readAhead += len; bt_skip(len); PROF_ADD(5); }

template <bool HW> __device__ __forceinline__ Symbol lz_next_symbol() {       // Get_Next_Symbol :1605-1796
  constexpr int hurdle = 40;
  if (s_B.readAhead == -1) lz_read_one<HW>(s_B.cur);
  s_B.cur_literal = BUF(s_B.readPos);
  const int a = bt_available(), avail = a < BT_LOOK ? a : BT_LOOK;
  if (avail < BT_MIN) return lz_send_literal();
  if (s_B.best_len_rep >= BT_NICE) {
    lz_skip(s_B.best_len_rep - 1);
    return lz_send_dl(s_B.rep_dist[s_B.best_rep_index], s_B.best_len_rep);
  }
  int main_len = 1, main_dist = 1;
  {
    Matches &C = s_MM[s_B.cur];
    if (C.count > 0) {
      main_len = C.len[C.count]; main_dist = C.dist[C.count];
      if (main_len >= BT_NICE) { lz_skip(main_len - 1); return lz_send_dl(main_dist, main_len); }
      lz_reduce(s_B.cur);
      lz_supplement(s_B.cur);
      main_len = C.len[C.count]; main_dist = C.dist[C.count];
      if (main_len == BT_MIN && main_dist > 128) main_len = 1;
    }
  }
  if (s_B.best_len_rep > BT_MIN &&
      (s_B.best_len_rep >= main_len || (s_B.best_len_rep >= main_len - 2 && main_dist > (1 << 9)) || (s_B.best_len_rep >= main_len - 3 && main_dist > (1 << 15)))) {
    lz_skip(s_B.best_len_rep - 1);
    return lz_send_dl(s_B.rep_dist[s_B.best_rep_index], s_B.best_len_rep);
  }
  if (main_len < BT_MIN || avail <= BT_MIN) return lz_send_literal();
  s_B.cur = 1 - s_B.cur;
  lz_read_one<HW>(s_B.cur);
  {
    Matches &C = s_MM[s_B.cur];
    if (C.count > 0) {
      const int nl = C.len[C.count], nd = C.dist[C.count];
      if ((nl >= main_len + hurdle && nd < main_dist) || (nl == main_len + hurdle + 1 && !much_smaller(main_dist, nd)) || nl > main_len + hurdle + 1 ||
          (nl >= main_len + hurdle - 1 && main_len >= BT_MIN + 1 && much_smaller(nd, main_dist))) {
        return lz_send_literal();
      }
      lz_reduce(s_B.cur);
      lz_supplement(s_B.cur);
      int idx = 1, set = 1 - s_B.cur;
      estimate_dl_codes<HW>(1 - s_B.cur, s_B.cur_literal, idx, set);
      if (set == 1 - s_B.cur) { main_len = s_MM[set].len[idx]; main_dist = s_MM[set].dist[idx]; }
      else return lz_send_literal();
    }
  }
  const int limit = main_len - 1 > BT_MIN ? main_len - 1 : BT_MIN;
  if (__any(bt_match_len(s_B.rep_dist[threadIdx.x & 3], limit) == limit)) return lz_send_literal();
  lz_skip(main_len - 2);
  return lz_send_dl(main_dist, main_len);
}

__device__ bool lz_bt4_begin(int sbs, uint64_t base, const Bt4Sets &sets) {        // LZ77_using_BT4 up to its main loop; False: nothing to code
  s_B.sbs = sbs; s_B.readPos = -1; s_B.readLimit = -1; s_B.writePos = 0; s_B.pendingSize = 0;
  s_B.keepSizeBefore = BT_OPTS + sbs;
  s_B.keepSizeAfter = BT_OPTS + BT_LOOK;
  const int64_t r = (int64_t)sbs / 2 + 256 * 1024, rmax = 512ll << 20;
  s_B.buf_len = s_B.keepSizeBefore + s_B.keepSizeAfter + (int)(r < rmax ? r : rmax) + 1;
  s_B.moved = 0; s_B.in_pos = 0;
  s_B.base = base; s_B.sets = sets;
  s_B.readAhead = -1;
  for (int i = 0; i < 4; i++) { s_B.rep_dist[i] = 1; s_B.len_rep[i] = 0; }
  s_B.best_len_rep = 0; s_B.best_rep_index = 0;
  s_B.cur = 0; s_B.cur_literal = 0;
  s_MM[0].count = 0; s_MM[1].count = 0;
  return bt_fill_window(sbs) > 0;
}
#undef BUF

// ---------------------------------------------------------------- one stream per workgroup

// A stream in several launches.  Zip.Compress.LZMA_E reports progress and can be aborted between any two bytes it reads
// (zip-compress-lzma_e.adb:78-92); one launch per stream would be minutes to hours without either.  Everything a stream carries from
// one step of its main loop to the next lives in the four LDS objects (model, match sets, coder, BT4; the hash tables and the tree are in
// HBM anyway), so a launch that has coded `budget` more positions writes them to the job's slot and the next launch goes on there.
struct LzSave {
  uint32_t phase;                              // 0: not started, 1: under way, 2: finished
  uint32_t running;                            // bit 0, Level_3: the main loop has not seen the end of the input yet; bits 8 .. 9: the level the state is of
  uint64_t iter;                               // Level_0: next byte; Level_1 / _2: next token
  LzProbs P; Matches MM[2]; Enc E; BT4 B;
};
constexpr uint64_t LZ_SAVE_STRIDE = (sizeof(LzSave) + 63) & ~63ull;
static_assert(sizeof(ScoreCtx) % 4 == 0 && sizeof(LzProbs) % 4 == 0 && sizeof(Matches) % 4 == 0 && sizeof(Enc) % 4 == 0 && sizeof(BT4) % 4 == 0, "word copies");

template <typename T> __device__ inline void words_out(T *dst, const T &src) {
  const uint32_t *s = (const uint32_t *)&src; uint32_t *d = (uint32_t *)dst;
  for (uint32_t i = threadIdx.x; i < sizeof(T) / 4; i += 64) d[i] = s[i];
}
template <typename T> __device__ inline void words_in(T &dst, const T *src) {
  const uint32_t *s = (const uint32_t *)src; uint32_t *d = (uint32_t *)&dst;
  for (uint32_t i = threadIdx.x; i < sizeof(T) / 4; i += 64) d[i] = s[i];
}

template <bool HW> __global__ void __launch_bounds__(HW ? 64 * HELP_WAVES : 64, 2) k_lzma_encode(const LzmaJob *jobs, const uint32_t *order, const uint8_t *in_base, const uint32_t *tok_base, uint8_t *out_base,
                                                    Bt4Sets sets, uint64_t *result, uint8_t *save_base, uint64_t budget, uint64_t pos_cap) {
  LzProbs &P = s_P;
  const uint32_t job = order ? order[blockIdx.x] : blockIdx.x;      // the longest entries first: workgroups start in index order
  const LzmaJob J = jobs[job];
  LzSave *S = save_base ? (LzSave *)(save_base + job * LZ_SAVE_STRIDE) : nullptr;
  const uint32_t phase = S ? S->phase : 0u;
  if (phase == 2) return;                                           // (coded by an earlier launch of this call)
  uint64_t iter = 0;
  bool running = false;
  if (phase == 1) {
    words_in(s_P, &S->P); words_in(s_MM[0], &S->MM[0]); words_in(s_MM[1], &S->MM[1]); words_in(s_E, &S->E); words_in(s_B, &S->B);
    iter = S->iter; running = (S->running & 1u) != 0;
    __syncthreads();
    // (the producer's buffers of THIS call -- and the entry's and the stream's addresses: a state that zada_lzma_export_state took out of another context,
    // or another process, goes on from where it stopped)
    if (threadIdx.x == 0) { s_B.sets = sets; s_E.in = in_base + J.in_off; s_E.out = out_base + J.out_off; s_E.cap = J.cap; }
    __syncthreads();
  } else {
    {
      uint16_t *p = (uint16_t *)&P;
      for (uint32_t i = threadIdx.x; i < sizeof(LzProbs) / 2; i += 64) p[i] = 1024;    // initial_probability
    }
    __syncthreads();
  }
  // (the kernel of a stream alone, 256 threads: the waves behind the first are its helpers from here on -- "one stream on four waves")
  if constexpr (HW) { if (threadIdx.x >= 64u) { helper_loop((int)(threadIdx.x >> 6)); return; } }
#ifdef ZADA_LZ_PROF
  const unsigned long long prof_k0 = clock64();
  for (int i = 0; i < 8; i++) g_lzprof[i] = 0;
#endif
  if (phase == 0) {
    s_E.in = in_base + J.in_off; s_E.n = J.n;
    s_E.cv = J.level <= 1 ? 0 : J.level == 2 ? 1 : 2;
    s_E.ES.state = 0; s_E.ES.pos_state = 0; s_E.ES.prev_byte = 0; s_E.ES.pos = 0; s_E.ES.tw = 1;
    s_E.ES.rep[0] = s_E.ES.rep[1] = s_E.ES.rep[2] = s_E.ES.rep[3] = 0;
    s_E.width = 0xFFFFFFFFu; s_E.low = 0; s_E.cache = 0; s_E.cache_size = 1;
    s_E.out = out_base + J.out_off; s_E.cap = J.cap; s_E.olen = 0;
    s_E.verify = J.level == 3 ? (uint32_t)J.verify : 0u; s_E.defect = 0;
    if (J.zip_prefix) { put_byte(16); put_byte(2); put_byte(5); put_byte(0); }   // zip-compress-lzma_e.adb:155-158
    put_byte(3 + 9 * 0 + 45 * 2);                                                   // Write_LZMA_header :1513-1536
    for (int i = 0; i < 4; i++) put_byte((J.sbs >> (8 * i)) & 255);
    if (J.level == 3) running = lz_bt4_begin((int)J.sbs, J.in_off, sets);
  }
  // (every step of the loops below codes at least one position; pos_cap: the match sets beyond are still being written, lzma_run "segments")
  const uint64_t stop = budget ? (s_E.ES.pos + budget < pos_cap ? s_E.ES.pos + budget : pos_cap) : ~0ull;
  bool done = true;
  const uint32_t *tok = tok_base + J.tok_off;
  // One loop for the three sources of symbols -- No_LZ77 (level 0: every byte a literal), the tokens of the Info-Zip matcher (levels 1, 2)
  // and Get_Next_Symbol of LZ77_using_BT4 (lz77.adb:1798-1827) -- so that the emission is inlined once.
  for (;;) {
    if (s_E.defect) {                                                 // (ZADA_E_REFERENCE: nothing more is coded)
      if (S && threadIdx.x == 0) S->phase = 2;
      if (threadIdx.x == 0) { result[2 * job] = 0; result[2 * job + 1] = s_E.ES.pos | (1ull << 62); }
      helpers_release<HW>();
      return;
    }
    if (s_E.ES.pos >= stop) { done = false; break; }
    Symbol sy;
    if (J.level == 0) {
      if (iter >= J.n) break;
      sy = Symbol{0, 0u, (uint32_t)s_E.in[iter]};
      iter++;
    } else if (J.level <= 2) {
      if (iter >= J.ntok) break;
      const uint32_t tk = tok[iter];
      sy = (tk & 0x80000000u) ? Symbol{(int)((tk >> 16) & 0x7FFF), tk & 0xFFFF, 0u} : Symbol{0, 0u, tk & 0xFF};
      iter++;
    } else {
      if (!running) break;
      sy = lz_next_symbol<HW>();
      if (bt_available() == 0 && bt_fill_window((int)J.sbs) == 0) running = false;    // (the window's bookkeeping: nothing the emission reads)
    }
    if (sy.length) emit_dl<HW>(sy.distance, sy.length); else emit_literal(sy.literal);
  }
  if (!done) {
    chain_sync<HW>();
    words_out(&S->P, s_P); words_out(&S->MM[0], s_MM[0]); words_out(&S->MM[1], s_MM[1]); words_out(&S->E, s_E); words_out(&S->B, s_B);
    if (threadIdx.x == 0) {
      S->phase = 1; S->running = (running ? 1u : 0u) | ((uint32_t)J.level << 8); S->iter = iter;      // (the level: lzma_save_fits)
      result[2 * job] = s_E.olen;
      result[2 * job + 1] = s_E.ES.pos | (1ull << 63);                             // bit 63: more to come
    }
    helpers_release<HW>();
    return;
  }
  encode_bit(P.match[s_E.ES.state][s_E.ES.pos_state], 1);                             // end marker :1549-1556
  write_simple_match(0xFFFFFFFFu, 2);
  for (int i = 0; i < 5; i++) shift_low();                                          // Flush_range_encoder
#ifdef ZADA_LZ_PROF
  if (blockIdx.x == 0 && threadIdx.x == 0) printf("LZPROF total %llu literal %llu emit_dl %llu (decisions %llu: strict + expanded %llu, cuts %llu) estimate %llu bt_get %llu bt_skip %llu\n", clock64() - prof_k0, g_lzprof[1], g_lzprof[2], g_lzprof[0], g_lzprof[7], g_lzprof[6], g_lzprof[3], g_lzprof[4], g_lzprof[5]);
#endif
  if (S && threadIdx.x == 0) S->phase = 2;
  result[2 * job] = s_E.olen;
  result[2 * job + 1] = s_E.ES.pos;
  helpers_release<HW>();
}

}  // namespace

// String_buffer_size of a level (:137-149) and the BT4 hash size (lz77.adb:1019-1032)
uint32_t lzma_string_buffer_size(int level, uint64_t dictionary_size) {
  if (level == 0) return 16;
  if (level <= 2) return 1u << 15;
  return bt4_string_buffer_size(dictionary_size);
}
uint32_t lzma_hash4_size(uint32_t sbs) { return bt4_hash4_size(sbs); }

// A batch whose tokens came from ONE pass of the LZ stage over all entries (zada_api.hip, lzma_batch_iz): entry e's tokens are
// those whose positions lie in its slot [ent_start[e], ent_start[e + 1]) of the packed buffer.
namespace {
__global__ void __launch_bounds__(256) k_lzma_token_ranges(uint32_t E, const uint32_t *__restrict__ apos, uint32_t T, const uint32_t *__restrict__ ent_start, LzmaJob *jobs) {
  const uint32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  uint32_t bound[2];
  for (int k = 0; k < 2; k++) {
    const uint32_t key = ent_start[e + k];
    uint32_t lo = 0, hi = T;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (apos[mid] < key) lo = mid + 1; else hi = mid; }
    bound[k] = lo;
  }
  jobs[e].tok_off = bound[0];
  jobs[e].ntok = bound[1] - bound[0];
}
}  // namespace
int lzma_token_ranges(Ctx *c, uint32_t E, const uint32_t *d_apos, uint32_t T, const uint32_t *d_ent_start, LzmaJob *d_jobs) {
  if (E == 0) return 0;
  hipLaunchKernelGGL(k_lzma_token_ranges, dim3((E + 255) / 256), dim3(256), 0, c->stream, E, d_apos, T, d_ent_start, d_jobs);
  return hip_check(c, hipGetLastError(), "k_lzma_token_ranges") ? ZADA_E_HIP : 0;
}

// jobs[0 .. count): device array; results: 2 x count uint64 (stream bytes, input bytes coded).  sets: the BT4 producer's match sets (Level_3).
// d_save / budget: a stream in several launches (count slots of lzma_save_stride() bytes, zero before the first launch; a launch codes
// `budget` more positions of every unfinished stream -- up to position pos_cap at most -- and flags bit 63 of its second result while there
// is more to come); nullptr / 0: one launch.
uint64_t lzma_save_stride() { return LZ_SAVE_STRIDE; }
// what a saved state (host copy of one slot) says about itself: 1 = a stream under way, with the positions it has coded and the stream bytes it has written
int lzma_save_info(const uint8_t *blob, uint64_t *pos, uint64_t *olen, uint64_t *n) {
  const LzSave *S = (const LzSave *)blob;
  if (S->phase != 1) return 0;
  *pos = S->E.ES.pos; *olen = S->E.olen; *n = S->E.n;
  return 1;
}
// does an imported state belong to THIS stream?  (length, dictionary, place in the arena; its counters are inside the stream and the window: the kernel
// takes them as they are)
int lzma_save_fits(const uint8_t *blob, const LzmaJob &J) {
  const LzSave *S = (const LzSave *)blob;
  if (S->phase != 1 || S->E.n != J.n || S->E.ES.pos > J.n) return 0;   // (olen beyond the caller's room is the coder's ordinary "counted, not written")
  if ((int32_t)(S->running >> 8) != J.level || (S->running & 0xFEu)) return 0;
  if (J.level < 3) return S->iter <= (J.level == 0 ? J.n : J.ntok);   // (next byte / next token)
  if ((uint64_t)(uint32_t)S->B.sbs != J.sbs || S->B.base != J.in_off || S->B.in_pos > J.n) return 0;
  if (S->B.buf_len < 0 || S->B.readPos < -1 || S->B.readPos > S->B.buf_len || S->B.writePos < 0 || S->B.writePos > S->B.buf_len || S->B.readLimit > S->B.buf_len) return 0;
  return 1;
}
int lzma_launch(Ctx *c, const LzmaJob *d_jobs, const uint32_t *d_order, uint32_t count, const uint8_t *d_in, const uint32_t *d_tok, uint8_t *d_out, const Bt4Sets &sets, uint64_t *d_result,
                uint8_t *d_save, uint64_t budget, uint64_t pos_cap, int waves) {
  if (count == 0) return 0;
  if (waves > 1)                                                     // (the chain's wave and HELP_WAVES - 1 helpers)
    hipLaunchKernelGGL(k_lzma_encode<true>, dim3(count), dim3(64 * HELP_WAVES), 0, c->stream, d_jobs, d_order, d_in, d_tok, d_out, sets, d_result, d_save, d_save ? budget : 0ull, pos_cap);
  else
  hipLaunchKernelGGL(k_lzma_encode<false>, dim3(count), dim3(64), 0, c->stream, d_jobs, d_order, d_in, d_tok, d_out, sets, d_result, d_save, d_save ? budget : 0ull, pos_cap);
  return hip_check(c, hipGetLastError(), "k_lzma_encode") ? ZADA_E_HIP : 0;
}

}  // namespace zada
