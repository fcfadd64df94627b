/* TEST INFRASTRUCTURE - CPU restatement of Zip-Ada's LZMA encoder.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this file;
 * the product (zip-ada_amd/) never does.
 *
 * PARITY UNPINNED: the reference tree holds no Zip-Ada-made LZMA stream and there is no GNAT in this image.
 * What is checked: every stream decodes with liblzma (Python's lzma module, raw LZMA1 filter) to the input,
 * and oracle/pin_with_gnat.sh compares `zipada -el1..3` payloads with tests/golden/lzma_digests.json
 * wherever a GNAT toolchain exists.  Valid-but-different choices would pass the first check.
 * "Every stream decodes": with the dictionary Zip.Compress.LZMA_E asks for (the entry's size).  With a dictionary SMALLER than the data the
 * reference's BT4 reads positions behind pending bytes that no window fill took up (lz77.adb:1000-1017, 1397-1406), lzPos lags behind readPos,
 * and its matcher reports matches that are none (:1262-1290): this file keeps that, it is the reference's behaviour -- such streams do NOT decode
 * to the input (tests/test_lzma_oracle.py::test_reference_defect_behind_pending_bytes_no_fill_took_up, tests/golden/lzma_defect.json, and
 * oracle/pin_lzma_defect.adb for a GNAT box); the product refuses those entries (ZADA_E_REFERENCE).
 *
 * Follows, function by function:
 *   zip_lib/lzma.ads:81-268            constants, probability model layout
 *   zip_lib/lzma-encoding.adb:59-1563  Encode (Estimates :349-946, range coder :952-1039, machine :1045-1361,
 *                                      Estimate_DL_Codes_for_LZ77 :1363-1498, header :1513-1536)
 *   zip_lib/lz77.adb:953-1827          LZ77_using_BT4 (Level_3); Level_1 / Level_2 take the Info-Zip matcher
 *                                      (IZ_6 / IZ_10, lzma-encoding.adb:118-122) restated in zada_oracle.c
 *   zip_lib/zip-compress-lzma_e.adb:121-172  method -> (lc, lp, pb, level), 4-byte Zip prefix, dictionary_size = input size
 *
 * Floating point: MProb is `digits 15` = IEEE double on every GNAT target of interest; every product below keeps the
 * reference's association (left to right).  Build without contraction (-ffp-contract=off) and without -ffast-math.
 */
#include "zada_oracle.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint16_t CProb;

enum {
  states_count = 12, max_pos_states_count = 16,
  probability_model_bits = 11, probability_model_count = 1 << 11, probability_change_bits = 5, initial_probability = 1 << 10,
  align_bits = 4, align_table_size = 16, align_mask = 15,
  len_to_pos_states = 4, Dist_slot_bits = 6, Start_dist_model_index = 4, End_dist_model_index = 14, Num_full_distances = 128,
  Len_low_bits = 3, Len_low_symbols = 8, Len_mid_bits = 3, Len_mid_symbols = 8, Len_high_bits = 8, Len_high_symbols = 256,
  Min_match_length = 2, Max_match_length = 273, Min_dictionary_size = 1 << 12
};
#define width_threshold (1u << 24)
#define end_of_stream_magic_distance 0xFFFFFFFFu

/* lzma.ads:86-89 */
static const uint8_t Update_State_Literal[12]  = {0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 4, 5};
static const uint8_t Update_State_Match[12]    = {7, 7, 7, 7, 7, 7, 7, 10, 10, 10, 10, 10};
static const uint8_t Update_State_Rep[12]      = {8, 8, 8, 8, 8, 8, 8, 11, 11, 11, 11, 11};
static const uint8_t Update_State_ShortRep[12] = {9, 9, 9, 9, 9, 9, 9, 11, 11, 11, 11, 11};

typedef struct {                                   /* lzma.ads:137-146 */
  CProb match[states_count][max_pos_states_count];
  CProb rep[states_count], rep_g0[states_count], rep_g1[states_count], rep_g2[states_count];
  CProb rep0_long[states_count][max_pos_states_count];
} Probs_for_Switches;

typedef struct {                                   /* lzma.ads:155-161 */
  CProb choice_1, choice_2;
  CProb low_coder[max_pos_states_count][8], mid_coder[max_pos_states_count][8];
  CProb high_coder[256];
} Probs_for_LZ_Lengths;

typedef struct {                                   /* lzma.ads:181-185; pos_coder index -1 .. 114 is stored at +1 */
  CProb slot_coder[len_to_pos_states][64];
  CProb align_coder[16];
  CProb pos_coder[Num_full_distances - End_dist_model_index + 2];
} Probs_for_LZ_Distances;

typedef struct {                                   /* lzma-encoding.adb:212-219 */
  unsigned state, pos_state;
  uint8_t prev_byte;
  uint32_t R;
  uint64_t total_pos;
  uint32_t rep_dist[4];
} Machine_State;

enum { CV_None = 0, CV_Simple = 1, CV_Splitting = 2 };
#define max_recursion 2

typedef struct { int distance, length; } DLP;
typedef struct { int count; DLP dl[513]; } Matches_Type;       /* lz77.ads:70-75, dl (1 .. 512) */

typedef struct {
  /* parameters */
  int level, lc, lp, pb, compare_variants;
  uint32_t pos_bits_mask, literal_pos_mask, Text_Buf_Mask;
  int String_buffer_size;
  /* probabilities */
  CProb *lit; int lit_count;
  Probs_for_LZ_Distances dist;
  Probs_for_LZ_Lengths len, rep_len;
  Probs_for_Switches sw;
  uint8_t *Text_Buf;
  Machine_State ES;
  /* range encoder (:952-957) */
  uint32_t width; uint64_t low; uint8_t cache; uint64_t cache_size;
  uint64_t encoded_uncompressed_bytes;
  /* sink */
  uint8_t *out; uint64_t cap, out_len;
  /* source */
  const uint8_t *in; uint64_t n, in_pos;
  /* decision counters for tests */
  uint64_t stat[8];
} Lz;

static void Write_Byte(Lz *L, uint8_t b) { if (L->out_len < L->cap) L->out[L->out_len] = b; L->out_len++; }

/* lzma-encoding.adb:73-102 */
static unsigned Get_dist_slot(uint32_t dist) {
  uint32_t n; int i;
  if (dist <= Start_dist_model_index) return dist;
  n = dist; i = 31;
  if ((n & 0xFFFF0000u) == 0) { n <<= 16; i = 15; }
  if ((n & 0xFF000000u) == 0) { n <<= 8; i -= 8; }
  if ((n & 0xF0000000u) == 0) { n <<= 4; i -= 4; }
  if ((n & 0xC0000000u) == 0) { n <<= 2; i -= 2; }
  if ((n & 0x80000000u) == 0) i -= 1;
  return (unsigned)(i * 2) + ((dist >> (i - 1)) & 1);
}

/* :105-112 */
static int64_t Ceiling_power_of_2(int64_t x) {
  int64_t p = 1;
  while (p < 0x7FFFFFFF / 2 && p < x) p *= 2;
  return p > x ? p : x;
}

/* :193-201 */
static int Idx_for_Literal_prob(const Lz *L, uint64_t position, uint8_t prev_byte) {
  return 0x300 * (int)((((uint32_t)position & L->literal_pos_mask) << L->lc) + ((uint32_t)prev_byte >> (8 - L->lc)));
}

/* ---------------------------------------------------------------- Estimates (:349-946) */

static inline double Test_Bit_Encoding(CProb prob_bit, unsigned symbol) {        /* :359-370 */
  double b = (double)symbol;
  return b + (1.0 - 2.0 * b) * ((double)prob_bit * (1.0 / 2048.0));
}

static double Test_Simple_Literal(const Lz *L, uint8_t b, uint8_t b_match, const CProb *prob, const Machine_State *sim) {  /* :372-419 */
  double prob_lit = Test_Bit_Encoding(L->sw.match[sim->state][sim->pos_state], 0);
  uint32_t symb = (uint32_t)b | 0x100;
  if (sim->state < 7) {
    do {
      prob_lit = prob_lit * Test_Bit_Encoding(prob[symb >> 8], (symb >> 7) & 1);
      symb <<= 1;
    } while (symb < 0x10000);
  } else {
    uint32_t offs = 0x100, match = b_match;
    do {
      match <<= 1;
      prob_lit = prob_lit * Test_Bit_Encoding(prob[offs + (match & offs) + (symb >> 8)], (symb >> 7) & 1);
      symb <<= 1;
      offs &= ~(match ^ symb);
    } while (symb < 0x10000);
  }
  return prob_lit;
}

static double Test_Short_Rep_Match(const Lz *L, const Machine_State *sim) {      /* :421-428 */
  return Test_Bit_Encoding(L->sw.match[sim->state][sim->pos_state], 1) *
         Test_Bit_Encoding(L->sw.rep[sim->state], 1) *
         Test_Bit_Encoding(L->sw.rep_g0[sim->state], 0) *
         Test_Bit_Encoding(L->sw.rep0_long[sim->state][sim->pos_state], 0);
}

static void Simulate_Literal_Byte(const Lz *L, uint8_t b, Machine_State *sim, double *prob) {   /* :431-458 */
  int probs_lit_idx = Idx_for_Literal_prob(L, sim->total_pos, sim->prev_byte);
  uint8_t b_match = L->Text_Buf[(sim->R - sim->rep_dist[0] - 1) & L->Text_Buf_Mask];
  double ltr, srm;
  sim->pos_state = (unsigned)((uint32_t)sim->total_pos & L->pos_bits_mask);
  ltr = Test_Simple_Literal(L, b, b_match, L->lit + probs_lit_idx, sim);
  if (b == b_match && sim->total_pos > (uint64_t)(uint32_t)(sim->rep_dist[0] + 1)) {
    srm = Test_Short_Rep_Match(L, sim);
    if (srm > ltr) {
      sim->state = Update_State_ShortRep[sim->state];
      *prob = *prob * srm;
      goto update;
    }
  }
  sim->state = Update_State_Literal[sim->state];
  *prob = *prob * ltr;
update:
  sim->R = (sim->R + 1) & L->Text_Buf_Mask;
  sim->total_pos += 1;
  sim->pos_state = (unsigned)((uint32_t)sim->total_pos & L->pos_bits_mask);
  sim->prev_byte = b;
}

static double Test_Literal_Byte(const Lz *L, uint8_t b, const Machine_State *sim) {   /* :460-468 */
  Machine_State sim_var = *sim;
  double prob = 1.0;
  Simulate_Literal_Byte(L, b, &sim_var, &prob);
  return prob;
}

static double Simulate_Bit_Tree(const CProb *prob, int num_bits, unsigned symbol) {   /* :470-481 */
  double res = 1.0;
  unsigned bit, m = 1;
  for (int i = num_bits - 1; i >= 0; i--) {
    bit = (symbol >> i) & 1;
    res = res * Test_Bit_Encoding(prob[m], bit);
    m = 2 * m + bit;
  }
  return res;
}

static double Test_Length(const Probs_for_LZ_Lengths *pl, unsigned length, unsigned sim_pos_state) {   /* :483-509 */
  unsigned len = length - Min_match_length;
  double res;
  if (len < Len_low_symbols) {
    res = Test_Bit_Encoding(pl->choice_1, 0) * Simulate_Bit_Tree(pl->low_coder[sim_pos_state], Len_low_bits, len);
  } else {
    res = Test_Bit_Encoding(pl->choice_1, 1);
    len -= Len_low_symbols;
    if (len < Len_mid_symbols) {
      res = res * Test_Bit_Encoding(pl->choice_2, 0) * Simulate_Bit_Tree(pl->mid_coder[sim_pos_state], Len_mid_bits, len);
    } else {
      res = res * Test_Bit_Encoding(pl->choice_2, 1);
      len -= Len_mid_symbols;
      res = res * Simulate_Bit_Tree(pl->high_coder, Len_high_bits, len);
    }
  }
  return res;
}

static double Test_Repeat_Match(const Lz *L, int index_rm, unsigned length, const Machine_State *sim) {   /* :511-538 */
  const Probs_for_Switches *s = &L->sw;
  double res = Test_Bit_Encoding(s->rep[sim->state], 1);
  switch (index_rm) {
    case 0: res = res * Test_Bit_Encoding(s->rep_g0[sim->state], 0) * Test_Bit_Encoding(s->rep0_long[sim->state][sim->pos_state], 1); break;
    case 1: res = res * Test_Bit_Encoding(s->rep_g0[sim->state], 1) * Test_Bit_Encoding(s->rep_g1[sim->state], 0); break;
    case 2: res = res * Test_Bit_Encoding(s->rep_g0[sim->state], 1) * Test_Bit_Encoding(s->rep_g1[sim->state], 1) * Test_Bit_Encoding(s->rep_g2[sim->state], 0); break;
    default: res = res * Test_Bit_Encoding(s->rep_g0[sim->state], 1) * Test_Bit_Encoding(s->rep_g1[sim->state], 1) * Test_Bit_Encoding(s->rep_g2[sim->state], 1); break;
  }
  return res * Test_Length(&L->rep_len, length, sim->pos_state);
}

static double Simulate_Bit_Tree_Reverse(const CProb *prob, int num_bits, uint32_t symbol) {   /* :548-563 */
  double res = 1.0;
  uint32_t symb = symbol;
  unsigned m = 1, bit;
  for (int c = num_bits; c >= 1; c--) {
    bit = symb & 1;
    res = res * Test_Bit_Encoding(prob[m], bit);
    m = 2 * m + bit;
    symb >>= 1;
  }
  return res;
}

static double half_power(int k) {                 /* 0.5 ** k, exact for every k met here (0 .. 26) */
  double r = 1.0;
  for (int i = 0; i < k; i++) r = r * 0.5;
  return r;
}

static double Test_Simple_Match(const Lz *L, uint32_t distance, unsigned length, const Machine_State *sim) {   /* :540-601 */
  /* Test_Distance :565-595 */
  unsigned len_state = length - 2 < len_to_pos_states - 1 ? length - 2 : len_to_pos_states - 1;
  unsigned dist_slot = Get_dist_slot(distance);
  double td = Simulate_Bit_Tree(L->dist.slot_coder[len_state], Dist_slot_bits, dist_slot);
  if (dist_slot >= Start_dist_model_index) {
    int footerBits = (int)(dist_slot >> 1) - 1;
    uint32_t base = (uint32_t)(2 | (dist_slot & 1)) << footerBits;
    uint32_t dist_reduced = distance - base;
    if (dist_slot < End_dist_model_index) {
      td = td * Simulate_Bit_Tree_Reverse(L->dist.pos_coder + ((int)base - (int)dist_slot - 1) + 1, footerBits, dist_reduced);
    } else {
      td = td * half_power(footerBits - align_bits) * Simulate_Bit_Tree_Reverse(L->dist.align_coder, align_bits, dist_reduced & align_mask);
    }
  }
  return Test_Bit_Encoding(L->sw.rep[sim->state], 0) * Test_Length(&L->len, length, sim->pos_state) * td;
}

#define Malus_simple_match_vs_rep 0.55
#define Lit_then_DL_threshold 0.875

static void Simulate_Strict_DL_Code(const Lz *L, uint32_t distance, int length, Machine_State *sim, double *prob) {   /* :605-659 */
  uint32_t dist_ip = distance - 1;
  int found_repeat = -1;
  double dlc = Test_Bit_Encoding(L->sw.match[sim->state][sim->pos_state], 1);
  double sma = Test_Simple_Match(L, dist_ip, (unsigned)length, sim);
  double rma;
  uint32_t aux;
  for (int i = 0; i < 4; i++) if (dist_ip == sim->rep_dist[i]) { found_repeat = i; break; }
  if (found_repeat >= 0) {
    rma = Test_Repeat_Match(L, found_repeat, (unsigned)length, sim);
    if (rma >= sma * Malus_simple_match_vs_rep) {
      *prob = *prob * dlc * rma;
      aux = sim->rep_dist[found_repeat];
      for (int i = found_repeat; i >= 1; i--) sim->rep_dist[i] = sim->rep_dist[i - 1];
      sim->rep_dist[0] = aux;
      sim->state = Update_State_Rep[sim->state];
      goto update;
    }
  }
  *prob = *prob * dlc * sma;
  for (int i = 3; i >= 1; i--) sim->rep_dist[i] = sim->rep_dist[i - 1];
  sim->rep_dist[0] = dist_ip;
  sim->state = Update_State_Match[sim->state];
update:
  sim->total_pos += (uint64_t)length;
  sim->pos_state = (unsigned)((uint32_t)sim->total_pos & L->pos_bits_mask);
  sim->R = (sim->R + (uint32_t)length) & L->Text_Buf_Mask;
  sim->prev_byte = L->Text_Buf[(sim->R - 1) & L->Text_Buf_Mask];
}

static double Test_Strict_DL_Code(const Lz *L, uint32_t distance, int length, const Machine_State *sim) {   /* :661-677 */
  Machine_State sim_var = *sim;
  double prob = 1.0;
  Simulate_Strict_DL_Code(L, distance, length, &sim_var, &prob);
  return prob;
}

static void Simulate_Expand_DL_code(const Lz *L, uint32_t distance, int length, double give_up, Machine_State *sim, double *prob) {   /* :680-707 */
  Machine_State sim_mem = *sim;
  double expanded_string_prob = 1.0;
  uint32_t Copy_start = (sim->R - distance) & L->Text_Buf_Mask;
  uint8_t b;
  for (int x = 1; x <= length; x++) {
    b = L->Text_Buf[(Copy_start + (uint32_t)(x - 1)) & L->Text_Buf_Mask];
    Simulate_Literal_Byte(L, b, sim, &expanded_string_prob);
    if (expanded_string_prob < give_up) { *sim = sim_mem; break; }
    sim->prev_byte = b;
  }
  *prob = *prob * expanded_string_prob;
}

static double Test_Expanded_DL_Code(const Lz *L, uint32_t distance, int length, double give_up, const Machine_State *sim) {   /* :709-726 */
  Machine_State sim_var = *sim;
  double prob = 1.0;
  Simulate_Expand_DL_code(L, distance, length, give_up, &sim_var, &prob);
  return prob;
}

static void Generic_any_DL_Code(Lz *L, uint32_t distance, int length, Machine_State *sim, double *prob, int recursion_limit, int I_am_a_simulation);
static void LZ77_emits_literal_byte(Lz *L, uint8_t b);
static void Write_Strict_DL_Code(Lz *L, uint32_t distance, int length);

static void Simulate_any_DL_Code(Lz *L, uint32_t distance, int length, Machine_State *sim, double *prob, int recursion_limit) {   /* :835-847 */
  Generic_any_DL_Code(L, distance, length, sim, prob, recursion_limit, 1);
}

static double Test_any_DL_Code(Lz *L, uint32_t distance, int length, const Machine_State *sim, int recursion_limit) {   /* :849-865 */
  Machine_State sim_var = *sim;
  double prob = 1.0;
  Simulate_any_DL_Code(L, distance, length, &sim_var, &prob, recursion_limit);
  return prob;
}

static double fmax0(double x) { return x > 0.0 ? x : 0.0; }      /* MProb'Max (0.0, x) */

static double DL_code_then_Literal(Lz *L, uint32_t distance, int length, const Machine_State *sim, int recursion_limit) {   /* :869-889 */
  Machine_State sim_var = *sim;
  double prob = fmax0(0.135 - (double)distance * 1.0e-8 - (double)length * 1.0e-4);
  Simulate_any_DL_Code(L, distance, length - 1, &sim_var, &prob, recursion_limit);
  Simulate_Literal_Byte(L, L->Text_Buf[(sim_var.R - distance) & L->Text_Buf_Mask], &sim_var, &prob);
  return prob;
}

static double Malus_lit_then_DL(uint32_t distance, int length) {   /* :891-895 */
  return fmax0(0.064 - (double)distance * 1.0e-9 - (double)length * 3.0e-5);
}

static int in_Splits_considered(int x) { return x >= 4 && x <= 9; }   /* :899 */

static void Test_Split_DL(Lz *L, uint32_t distance, int length, const Machine_State *sim, double hurdle, int recursion_limit,
                          double *best_prob, int *best_cut) {   /* :901-944 */
  Machine_State sim_var = *sim;
  double Malus = fmax0(0.27 - (double)distance * 2.0e-6);
  double prob;
  int lowered_recursion_limit = recursion_limit - 1 > 0 ? recursion_limit - 1 : 0;
  *best_prob = 0.0;
  *best_cut = Min_match_length;
  if (Malus < hurdle) return;
  for (int cut = 2; cut <= length - 2; cut++) {
    if (in_Splits_considered(cut) || in_Splits_considered(length - cut)) {
      prob = Malus;
      sim_var = *sim;
      Simulate_any_DL_Code(L, distance, cut, &sim_var, &prob, lowered_recursion_limit);
      if (prob <= hurdle) {
        /* give up this cut */
      } else {
        Simulate_any_DL_Code(L, distance, length - cut, &sim_var, &prob, lowered_recursion_limit);
        if (prob > *best_prob) { *best_prob = prob; *best_cut = cut; }
      }
    }
  }
}

/* :740-832.  In the instance that writes (I_am_a_simulation = 0) `sim` IS the encoder's state: the reference passes ES
 * by reference (a 48-byte record), and the writers update ES, so `sim` follows. */
static void Generic_any_DL_Code(Lz *L, uint32_t distance, int length, Machine_State *sim, double *prob, int recursion_limit, int I_am_a_simulation) {
  uint32_t Copy_start = (sim->R - distance) & L->Text_Buf_Mask;
  double strict_dlc, expanded_dlc, strict_or_expanded_dlc = 0.0, dlc_after_lit, head_lit;
  uint8_t b_head;
  double best_prob;
  int best_cut;
  int new_recursion_limit = I_am_a_simulation ? recursion_limit - 1 : recursion_limit;
#define EMIT_LITERAL(b)   do { if (I_am_a_simulation) Simulate_Literal_Byte(L, (b), sim, prob); else LZ77_emits_literal_byte(L, (b)); } while (0)
#define EMIT_STRICT(d, l) do { if (I_am_a_simulation) Simulate_Strict_DL_Code(L, (d), (l), sim, prob); else Write_Strict_DL_Code(L, (d), (l)); } while (0)
  if (new_recursion_limit < 0) { EMIT_STRICT(distance, length); return; }
  if (L->compare_variants >= CV_Simple) {
    strict_dlc = Test_Strict_DL_Code(L, distance, length, sim);
    expanded_dlc = Test_Expanded_DL_Code(L, distance, length, strict_dlc, sim);
    strict_or_expanded_dlc = strict_dlc > expanded_dlc ? strict_dlc : expanded_dlc;       /* MProb'Max :768 */
    if (length > Min_match_length) {
      b_head = L->Text_Buf[Copy_start & L->Text_Buf_Mask];
      head_lit = Test_Literal_Byte(L, b_head, sim);
      if (head_lit >= Lit_then_DL_threshold) {
        if (!I_am_a_simulation) L->stat[0]++;
        EMIT_LITERAL(b_head);
        Generic_any_DL_Code(L, distance, length - 1, sim, prob, new_recursion_limit, I_am_a_simulation);
        return;
      }
      {
        Machine_State after;
        after.state = Update_State_Literal[sim->state];
        after.pos_state = (unsigned)((uint32_t)(sim->total_pos + 1) & L->pos_bits_mask);
        after.prev_byte = b_head;
        after.R = (sim->R + 1) & L->Text_Buf_Mask;
        after.total_pos = sim->total_pos + 1;
        memcpy(after.rep_dist, sim->rep_dist, sizeof after.rep_dist);
        dlc_after_lit = Test_any_DL_Code(L, distance, length - 1, &after, new_recursion_limit);
      }
      if (head_lit * dlc_after_lit * Malus_lit_then_DL(distance, length) > strict_or_expanded_dlc) {
        if (!I_am_a_simulation) L->stat[1]++;
        EMIT_LITERAL(b_head);
        Generic_any_DL_Code(L, distance, length - 1, sim, prob, new_recursion_limit, I_am_a_simulation);
        return;
      }
      if (DL_code_then_Literal(L, distance, length, sim, new_recursion_limit) > strict_or_expanded_dlc) {
        if (!I_am_a_simulation) L->stat[2]++;
        Generic_any_DL_Code(L, distance, length - 1, sim, prob, new_recursion_limit, I_am_a_simulation);
        EMIT_LITERAL(L->Text_Buf[(sim->R - distance) & L->Text_Buf_Mask]);
        return;
      }
    }
    if (expanded_dlc > strict_dlc) {
      if (!I_am_a_simulation) L->stat[3]++;
      for (int x = 1; x <= length; x++) EMIT_LITERAL(L->Text_Buf[(Copy_start + (uint32_t)(x - 1)) & L->Text_Buf_Mask]);
      return;
    }
  }
  if (L->compare_variants >= CV_Splitting) {
    Test_Split_DL(L, distance, length, sim, strict_or_expanded_dlc, new_recursion_limit, &best_prob, &best_cut);
    if (best_prob > strict_or_expanded_dlc) {
      if (!I_am_a_simulation) L->stat[4]++;
      Generic_any_DL_Code(L, distance, best_cut, sim, prob, new_recursion_limit, I_am_a_simulation);
      Generic_any_DL_Code(L, distance, length - best_cut, sim, prob, new_recursion_limit, I_am_a_simulation);
      return;
    }
  }
  EMIT_STRICT(distance, length);
#undef EMIT_LITERAL
#undef EMIT_STRICT
}

/* ---------------------------------------------------------------- range encoder (:964-1039) */

static void Shift_low(Lz *L) {
  uint64_t lb_top32 = L->low >> 32;
  uint32_t lb_bottom32 = (uint32_t)(L->low & 0xFFFFFFFFu);
  uint8_t temp, lb_bits_33_40;
  if (lb_bottom32 < 0xFF000000u || lb_top32 != 0) {
    temp = L->cache;
    lb_bits_33_40 = (uint8_t)(lb_top32 & 0xFF);
    do {
      Write_Byte(L, (uint8_t)(temp + lb_bits_33_40));
      temp = 0xFF;
      L->cache_size--;
    } while (L->cache_size != 0);
    L->cache = (uint8_t)((lb_bottom32 >> 24) & 0xFF);
  }
  L->cache_size++;
  L->low = (uint64_t)(uint32_t)(lb_bottom32 << 8);
}

static void Flush_range_encoder(Lz *L) { for (int i = 1; i <= 5; i++) Shift_low(L); }

static inline void Normalize(Lz *L) {
  if (L->width < width_threshold) { L->width <<= 8; Shift_low(L); }
}

static void Encode_Bit(Lz *L, CProb *prob, unsigned symbol) {
  CProb cur_prob = *prob;
  uint32_t bound = (L->width >> probability_model_bits) * (uint32_t)cur_prob;
  if (symbol == 0) {
    L->width = bound;
    Normalize(L);
    *prob = (CProb)(cur_prob + ((probability_model_count - cur_prob) >> probability_change_bits));
  } else {
    L->low += (uint64_t)bound;
    L->width -= bound;
    Normalize(L);
    *prob = (CProb)(cur_prob - (cur_prob >> probability_change_bits));
  }
}

/* ---------------------------------------------------------------- literals (:1045-1139) */

static void Write_Literal(Lz *L, CProb *prob, uint32_t symbol) {
  uint32_t symb = symbol | 0x100;
  do {
    Encode_Bit(L, &prob[symb >> 8], (symb >> 7) & 1);
    symb <<= 1;
  } while (symb < 0x10000);
}

static void Write_Literal_Matched(Lz *L, CProb *prob, uint32_t symbol, uint32_t matched) {
  uint32_t symb = symbol | 0x100, offs = 0x100, match = matched;
  do {
    match <<= 1;
    Encode_Bit(L, &prob[offs + (match & offs) + (symb >> 8)], (symb >> 7) & 1);
    symb <<= 1;
    offs &= ~(match ^ symb);
  } while (symb < 0x10000);
}

static inline void Update_pos_state(Lz *L) { L->ES.pos_state = (unsigned)((uint32_t)L->ES.total_pos & L->pos_bits_mask); }

static void LZ77_emits_literal_byte(Lz *L, uint8_t b) {   /* :1097-1130 */
  Machine_State *ES = &L->ES;
  int pb_lit_idx = Idx_for_Literal_prob(L, ES->total_pos, ES->prev_byte);
  uint8_t b_match = L->Text_Buf[(ES->R - ES->rep_dist[0] - 1) & L->Text_Buf_Mask];
  if (b == b_match && ES->total_pos > (uint64_t)(uint32_t)(ES->rep_dist[0] + 1) &&
      (L->compare_variants == CV_None ||
       Test_Short_Rep_Match(L, ES) > Test_Simple_Literal(L, b, b_match, L->lit + pb_lit_idx, ES))) {
    Encode_Bit(L, &L->sw.match[ES->state][ES->pos_state], 1);
    Encode_Bit(L, &L->sw.rep[ES->state], 1);
    Encode_Bit(L, &L->sw.rep_g0[ES->state], 0);
    Encode_Bit(L, &L->sw.rep0_long[ES->state][ES->pos_state], 0);
    ES->state = Update_State_ShortRep[ES->state];
    L->stat[5]++;
  } else {
    Encode_Bit(L, &L->sw.match[ES->state][ES->pos_state], 0);
    if (ES->state < 7) Write_Literal(L, L->lit + pb_lit_idx, b);
    else Write_Literal_Matched(L, L->lit + pb_lit_idx, b, b_match);
    ES->state = Update_State_Literal[ES->state];
  }
  ES->total_pos += 1;
  Update_pos_state(L);
  ES->prev_byte = b;
  L->Text_Buf[ES->R] = b;
  ES->R = (ES->R + 1) & L->Text_Buf_Mask;
  L->encoded_uncompressed_bytes += 1;
}

/* ---------------------------------------------------------------- DL codes (:1145-1361) */

static void Bit_Tree_Encode(Lz *L, CProb *prob, int num_bits, unsigned symbol) {
  unsigned bit, m = 1;
  for (int i = num_bits - 1; i >= 0; i--) {
    bit = (symbol >> i) & 1;
    Encode_Bit(L, &prob[m], bit);
    m = 2 * m + bit;
  }
}

static void Encode_Length(Lz *L, Probs_for_LZ_Lengths *pl, unsigned length) {
  unsigned len = length - Min_match_length;
  if (len < Len_low_symbols) {
    Encode_Bit(L, &pl->choice_1, 0);
    Bit_Tree_Encode(L, pl->low_coder[L->ES.pos_state], Len_low_bits, len);
  } else {
    Encode_Bit(L, &pl->choice_1, 1);
    len -= Len_low_symbols;
    if (len < Len_mid_symbols) {
      Encode_Bit(L, &pl->choice_2, 0);
      Bit_Tree_Encode(L, pl->mid_coder[L->ES.pos_state], Len_mid_bits, len);
    } else {
      Encode_Bit(L, &pl->choice_2, 1);
      len -= Len_mid_symbols;
      Bit_Tree_Encode(L, pl->high_coder, Len_high_bits, len);
    }
  }
}

static void Bit_Tree_Reverse_Encode(Lz *L, CProb *prob, int num_bits, uint32_t symbol) {
  uint32_t symb = symbol;
  unsigned m = 1, bit;
  for (int c = num_bits; c >= 1; c--) {
    bit = symb & 1;
    Encode_Bit(L, &prob[m], bit);
    m = 2 * m + bit;
    symb >>= 1;
  }
}

static void Encode_Direct_Bits(Lz *L, uint32_t value, int num_bits) {
  for (int i = num_bits - 1; i >= 0; i--) {
    L->width >>= 1;
    L->low += (uint64_t)L->width & (0 - (uint64_t)((value >> i) & 1));
    Normalize(L);
  }
}

static void Write_Simple_Match(Lz *L, uint32_t dist_ip, unsigned length) {   /* :1183-1255 */
  Machine_State *ES = &L->ES;
  Encode_Bit(L, &L->sw.rep[ES->state], 0);
  ES->state = Update_State_Match[ES->state];
  Encode_Length(L, &L->len, length);
  {
    unsigned len_state = length - 2 < len_to_pos_states - 1 ? length - 2 : len_to_pos_states - 1;
    unsigned dist_slot = Get_dist_slot(dist_ip);
    Bit_Tree_Encode(L, L->dist.slot_coder[len_state], Dist_slot_bits, dist_slot);
    if (dist_slot >= Start_dist_model_index) {
      int footerBits = (int)(dist_slot >> 1) - 1;
      uint32_t base = (uint32_t)(2 | (dist_slot & 1)) << footerBits;
      uint32_t dist_reduced = dist_ip - base;
      if (dist_slot < End_dist_model_index) {
        Bit_Tree_Reverse_Encode(L, L->dist.pos_coder + ((int)base - (int)dist_slot - 1) + 1, footerBits, dist_reduced);
      } else {
        Encode_Direct_Bits(L, dist_reduced >> align_bits, footerBits - align_bits);
        Bit_Tree_Reverse_Encode(L, L->dist.align_coder, align_bits, dist_reduced & align_mask);
      }
    }
  }
  for (int i = 3; i >= 1; i--) ES->rep_dist[i] = ES->rep_dist[i - 1];
  ES->rep_dist[0] = dist_ip;
}

static void Write_Repeat_Match(Lz *L, int index_rm, unsigned length) {   /* :1257-1286 */
  Machine_State *ES = &L->ES;
  uint32_t aux;
  Encode_Bit(L, &L->sw.rep[ES->state], 1);
  switch (index_rm) {
    case 0:
      Encode_Bit(L, &L->sw.rep_g0[ES->state], 0);
      Encode_Bit(L, &L->sw.rep0_long[ES->state][ES->pos_state], 1);
      break;
    case 1:
      Encode_Bit(L, &L->sw.rep_g0[ES->state], 1);
      Encode_Bit(L, &L->sw.rep_g1[ES->state], 0);
      break;
    case 2:
      Encode_Bit(L, &L->sw.rep_g0[ES->state], 1);
      Encode_Bit(L, &L->sw.rep_g1[ES->state], 1);
      Encode_Bit(L, &L->sw.rep_g2[ES->state], 0);
      break;
    default:
      Encode_Bit(L, &L->sw.rep_g0[ES->state], 1);
      Encode_Bit(L, &L->sw.rep_g1[ES->state], 1);
      Encode_Bit(L, &L->sw.rep_g2[ES->state], 1);
      break;
  }
  aux = ES->rep_dist[index_rm];
  for (int i = index_rm; i >= 1; i--) ES->rep_dist[i] = ES->rep_dist[i - 1];
  ES->rep_dist[0] = aux;
  Encode_Length(L, &L->rep_len, length);
  ES->state = Update_State_Rep[ES->state];
}

static void Write_Strict_DL_Code(Lz *L, uint32_t distance, int length) {   /* :1288-1328 */
  Machine_State *ES = &L->ES;
  uint32_t dist_ip = distance - 1;
  int found_repeat = -1;
  Encode_Bit(L, &L->sw.match[ES->state][ES->pos_state], 1);
  for (int i = 0; i < 4; i++) if (dist_ip == ES->rep_dist[i]) { found_repeat = i; break; }
  if (found_repeat >= 0 &&
      (L->compare_variants == CV_None ||
       Test_Repeat_Match(L, found_repeat, (unsigned)length, ES) >= Test_Simple_Match(L, dist_ip, (unsigned)length, ES) * Malus_simple_match_vs_rep)) {
    Write_Repeat_Match(L, found_repeat, (unsigned)length);
    L->stat[6]++;
  } else {
    Write_Simple_Match(L, dist_ip, (unsigned)length);
    L->stat[7]++;
  }
  ES->total_pos += (uint64_t)length;
  Update_pos_state(L);
  ES->R = (ES->R + (uint32_t)length) & L->Text_Buf_Mask;
  ES->prev_byte = L->Text_Buf[(ES->R - 1) & L->Text_Buf_Mask];
}

static void Expand_DL_Code_to_Buffer(Lz *L, const Machine_State *sim, int distance, int length) {   /* :1338-1353 */
  uint32_t Rx = sim->R;
  uint32_t Copy_start = (sim->R - (uint32_t)distance) & L->Text_Buf_Mask;
  for (uint32_t K = 0; K <= (uint32_t)(length - 1); K++) {
    L->Text_Buf[Rx] = L->Text_Buf[(Copy_start + K) & L->Text_Buf_Mask];
    Rx = (Rx + 1) & L->Text_Buf_Mask;
  }
}

static void LZ77_emits_DL_code(Lz *L, int distance, int length) {   /* :1355-1361 */
  double dummy_prob = 0.0;
  Expand_DL_Code_to_Buffer(L, &L->ES, distance, length);
  L->encoded_uncompressed_bytes += (uint64_t)length;
  Generic_any_DL_Code(L, (uint32_t)distance, length, &L->ES, &dummy_prob, max_recursion, 0);
}

/* ---------------------------------------------------------------- Estimate_DL_Codes_for_LZ77 (:1363-1498) */

typedef struct {
  Lz *L;
  Matches_Type *matches;        /* (0 .. 1) */
  int old_match_index;
  int last_pos_any_DL;
  Machine_State sim_new;
  double head_lit_prob;
} Scoring_ctx;

static void Scoring(Scoring_ctx *S, const Machine_State *state, int start, int recursion_level, double *prob, int *index, int *match_set) {   /* :1385-1469 */
  Lz *L = S->L;
  double prob_i, tail_prob;
  Machine_State test_state;
  int length_trunc, some_index = 1, some_match_set = 0, last_pos_i;
  *prob = 0.0;
  for (int m = 0; m <= 1; m++) {
    for (int i = 1; i <= S->matches[m].count; i++) {
      last_pos_i = S->matches[m].dl[i].length + (m != S->old_match_index ? 1 : 0);
      if (last_pos_i >= start) {
        if (last_pos_i < S->last_pos_any_DL && recursion_level >= 2) {
          /* skipped :1408-1410 */
        } else {
          if (m != S->old_match_index && start == 1) { test_state = S->sim_new; prob_i = S->head_lit_prob; }
          else { test_state = *state; prob_i = 1.0; }
          if (m == S->old_match_index) length_trunc = S->matches[m].dl[i].length - start + 1;
          else if (start == 1) length_trunc = S->matches[m].dl[i].length;
          else length_trunc = S->matches[m].dl[i].length - start + 2;
          if (length_trunc == 1) Simulate_Literal_Byte(L, L->Text_Buf[state->R], &test_state, &prob_i);
          else Simulate_any_DL_Code(L, (uint32_t)S->matches[m].dl[i].distance, length_trunc, &test_state, &prob_i, 1);
          if (last_pos_i < S->last_pos_any_DL) {
            Scoring(S, &test_state, last_pos_i + 1, recursion_level + 1, &tail_prob, &some_index, &some_match_set);
            prob_i = prob_i * tail_prob;
          }
          if (prob_i > *prob) { *prob = prob_i; *index = i; *match_set = m; }
        }
      }
    }
  }
}

static void Estimate_DL_Codes_for_LZ77(Lz *L, Matches_Type *matches, int old_match_index, uint8_t prefix1, int *best_score_index, int *best_score_set) {
  Scoring_ctx S;
  int new_wins = 0, last_pos_single_DL;
  DLP match_for_max_last_pos = {1, 1};
  Machine_State sim_expand = L->ES, sim_old = L->ES;
  double best_prob;
  S.L = L; S.matches = matches; S.old_match_index = old_match_index; S.last_pos_any_DL = 0; S.sim_new = L->ES;
  for (int m = 0; m <= 1; m++)
    for (int i = 1; i <= matches[m].count; i++) {
      last_pos_single_DL = matches[m].dl[i].length + (m != old_match_index ? 1 : 0);
      if (last_pos_single_DL > S.last_pos_any_DL) {
        S.last_pos_any_DL = last_pos_single_DL;
        match_for_max_last_pos = matches[m].dl[i];
        new_wins = m != old_match_index;
      }
    }
  if (new_wins) {
    L->Text_Buf[sim_expand.R] = prefix1;
    sim_expand.R = (sim_expand.R + 1) & L->Text_Buf_Mask;
  }
  Expand_DL_Code_to_Buffer(L, &sim_expand, match_for_max_last_pos.distance, match_for_max_last_pos.length);
  S.head_lit_prob = 1.0;
  Simulate_Literal_Byte(L, prefix1, &S.sim_new, &S.head_lit_prob);
  *best_score_index = 1; *best_score_set = old_match_index;          /* `out` parameters the reference leaves unset if no score is > 0.0 */
  Scoring(&S, &sim_old, 1, 1, &best_prob, best_score_index, best_score_set);
}

/* ---------------------------------------------------------------- LZ77_using_BT4 (lz77.adb:953-1827) */

typedef struct {
  Lz *L;
  int String_buffer_size, Look_Ahead, Threshold, MATCH_LEN_MIN;
  int readPos, readLimit, writePos, pendingSize;
  uint8_t cur_literal;
  int keepSizeBefore, keepSizeAfter, reserveSize, getBufSize, buf_len;
  int hash_4_size; uint32_t hash_4_mask;
  int32_t *hash2Table, *hash3Table, *hash4Table;
  uint32_t hash2Value, hash3Value, hash4Value;
  uint32_t crcTable[256];
  int Nice_Length, cyclicSize, cyclicPos, lzPos, max_dist;
  uint8_t *buf;
  int32_t *tree;
  int readAhead;
  int rep_dist[4], len_rep_dist[4];
  int best_length_for_rep_dist, best_rep_dist_index;
  Matches_Type matches[2];
  int current_match_index;
} BT4;

enum { Depth_Limit = 48, HASH_2_SIZE = 1 << 10, HASH_3_SIZE = 1 << 16, OPTS = 4096, Null_position = -1 };

static int bt_More_Bytes(BT4 *B) { return B->L->in_pos < B->L->n; }
static uint8_t bt_Read_Byte(BT4 *B) { return B->L->in[B->L->in_pos++]; }

static inline int Get_Available(const BT4 *B) { return B->writePos - B->readPos - 1; }   /* :992-998 */

static int Move_Pos(BT4 *B, int requiredForFlushing) {   /* :1000-1017 with finishing = False */
  int avail;
  B->readPos++;
  avail = Get_Available(B);
  if (avail < requiredForFlushing) { B->pendingSize++; avail = 0; }
  return avail;
}

static int getHash4Size(int String_buffer_size) {   /* :1019-1032 */
  uint32_t h = (uint32_t)(String_buffer_size - 1);
  h |= h >> 1; h |= h >> 2; h |= h >> 4; h |= h >> 8;
  h >>= 1;
  h |= 0xFFFF;
  if (h > (1u << 24)) h >>= 1;
  return (int)(h + 1);
}

static void calcHashes(BT4 *B, int off) {   /* :1061-1069 */
  const uint8_t *buf = B->buf;
  uint32_t temp = B->crcTable[buf[off]] ^ (uint32_t)buf[off + 1];
  B->hash2Value = temp & (HASH_2_SIZE - 1);
  temp ^= (uint32_t)buf[off + 2] << 8;
  B->hash3Value = temp & (HASH_3_SIZE - 1);
  temp ^= B->crcTable[buf[off + 3]] << 5;
  B->hash4Value = temp & B->hash_4_mask;
}

static void updateTables(BT4 *B, int pos) {
  B->hash2Table[B->hash2Value] = pos; B->hash3Table[B->hash3Value] = pos; B->hash4Table[B->hash4Value] = pos;
}

static void bt_normalize(int32_t *positions, int64_t count, int normalizationOffset) {   /* :981-990 */
  for (int64_t i = 0; i < count; i++) positions[i] = positions[i] <= normalizationOffset ? 0 : positions[i] - normalizationOffset;
}

static int Move_Pos_in_BT4(BT4 *B) {   /* :1127-1150 */
  int avail = Move_Pos(B, B->Nice_Length);
  if (avail != 0) {
    B->lzPos++;
    if (B->lzPos == 0x7FFFFFFF) {
      int normalizationOffset = 0x7FFFFFFF - B->cyclicSize;
      bt_normalize(B->hash2Table, HASH_2_SIZE, normalizationOffset);
      bt_normalize(B->hash3Table, HASH_3_SIZE, normalizationOffset);
      bt_normalize(B->hash4Table, B->hash_4_size, normalizationOffset);
      bt_normalize(B->tree, (int64_t)B->cyclicSize * 2, normalizationOffset);
      B->lzPos -= normalizationOffset;
    }
    B->cyclicPos++;
    if (B->cyclicPos == B->cyclicSize) B->cyclicPos = 0;
  }
  return avail;
}

static void Skip_and_Update_Tree(BT4 *B, int niceLenLimit, int currentMatch) {   /* :1154-1206 */
  const uint8_t *buf = B->buf;
  int32_t *tree = B->tree;
  int delta0, depth = Depth_Limit, ptr0 = B->cyclicPos * 2 + 1, ptr1 = B->cyclicPos * 2, pair, len, len0 = 0, len1 = 0;
  int readPos = B->readPos;
  for (;;) {
    delta0 = B->lzPos - currentMatch;
    if (depth == 0 || delta0 >= B->max_dist) { tree[ptr0] = Null_position; tree[ptr1] = Null_position; return; }
    depth--;
    pair = B->cyclicPos - delta0 < 0 ? B->cyclicSize : 0;
    pair = (B->cyclicPos - delta0 + pair) * 2;
    len = len0 < len1 ? len0 : len1;
    if (buf[readPos + len - delta0] == buf[readPos + len]) {
      for (;;) {
        len++;
        if (len == niceLenLimit) { tree[ptr1] = tree[pair]; tree[ptr0] = tree[pair + 1]; return; }
        if (buf[readPos + len - delta0] != buf[readPos + len]) break;
      }
    }
    if (buf[readPos + len - delta0] < buf[readPos + len]) {
      tree[ptr1] = currentMatch; ptr1 = pair + 1; currentMatch = tree[ptr1]; len1 = len;
    } else {
      tree[ptr0] = currentMatch; ptr0 = pair; currentMatch = tree[ptr0]; len0 = len;
    }
  }
}

static void BT4_Skip(BT4 *B, int len) {   /* :1208-1232 */
  for (int count = len; count >= 1; count--) {
    int niceLenLimit = B->Nice_Length, avail = Move_Pos_in_BT4(B), currentMatch;
    if (avail < niceLenLimit) {
      if (avail == 0) continue;
      niceLenLimit = avail;
    }
    calcHashes(B, B->readPos);
    currentMatch = B->hash4Table[B->hash4Value];
    updateTables(B, B->lzPos);
    Skip_and_Update_Tree(B, niceLenLimit, currentMatch);
  }
}

static void BT4_Read_One_and_Get_Matches(BT4 *B, Matches_Type *matches) {   /* :1234-1361 */
  const uint8_t *buf = B->buf;
  int32_t *tree = B->tree;
  int matchLenLimit = B->Look_Ahead, niceLenLimit = B->Nice_Length, avail;
  int delta0, delta2, delta3, currentMatch, lenBest, depth, ptr0, ptr1, pair, len, len0, len1, readPos;
  matches->count = 0;
  avail = Move_Pos_in_BT4(B);
  if (avail < matchLenLimit) {
    if (avail == 0) return;
    matchLenLimit = avail;
    if (niceLenLimit > avail) niceLenLimit = avail;
  }
  readPos = B->readPos;
  calcHashes(B, readPos);
  delta2 = B->lzPos - B->hash2Table[B->hash2Value];
  delta3 = B->lzPos - B->hash3Table[B->hash3Value];
  currentMatch = B->hash4Table[B->hash4Value];
  updateTables(B, B->lzPos);
  lenBest = 0;
  if (delta2 < B->max_dist && buf[readPos - delta2] == buf[readPos]) {
    lenBest = 2;
    matches->count = 1;
    matches->dl[1].length = 2;
    matches->dl[1].distance = delta2;
  }
  if (delta2 != delta3 && delta3 < B->max_dist && buf[readPos - delta3] == buf[readPos]) {
    lenBest = 3;
    matches->count++;
    matches->dl[matches->count].distance = delta3;
    delta2 = delta3;
  }
  if (matches->count > 0) {
    while (lenBest < matchLenLimit && buf[readPos + lenBest - delta2] == buf[readPos + lenBest]) lenBest++;
    matches->dl[matches->count].length = lenBest;
    if (lenBest >= niceLenLimit) { Skip_and_Update_Tree(B, niceLenLimit, currentMatch); return; }
  }
  if (lenBest < 3) lenBest = 3;
  depth = Depth_Limit;
  ptr0 = B->cyclicPos * 2 + 1;
  ptr1 = B->cyclicPos * 2;
  len0 = 0; len1 = 0;
  for (;;) {
    delta0 = B->lzPos - currentMatch;
    if (depth == 0 || delta0 >= B->max_dist) { tree[ptr0] = Null_position; tree[ptr1] = Null_position; return; }
    depth--;
    pair = B->cyclicPos - delta0 < 0 ? B->cyclicSize : 0;
    pair = (B->cyclicPos - delta0 + pair) * 2;
    len = len0 < len1 ? len0 : len1;
    if (buf[readPos + len - delta0] == buf[readPos + len]) {
      do { len++; } while (!(len >= matchLenLimit || buf[readPos + len - delta0] != buf[readPos + len]));
      if (len > lenBest) {
        lenBest = len;
        matches->count++;
        matches->dl[matches->count].length = len;
        matches->dl[matches->count].distance = delta0;
        if (len >= niceLenLimit) { tree[ptr1] = tree[pair]; tree[ptr0] = tree[pair + 1]; return; }
      }
    }
    if (buf[readPos + len - delta0] < buf[readPos + len]) {
      tree[ptr1] = currentMatch; ptr1 = pair + 1; currentMatch = tree[ptr1]; len1 = len;
    } else {
      tree[ptr0] = currentMatch; ptr0 = pair; currentMatch = tree[ptr0]; len0 = len;
    }
  }
}

static void Move_Window(BT4 *B) {   /* :1375-1386 */
  int moveOffset = ((B->readPos + 1 - B->keepSizeBefore) / 16) * 16;
  int moveSize = B->writePos - moveOffset;
  memmove(B->buf, B->buf + moveOffset, (size_t)moveSize);
  B->readPos -= moveOffset; B->readLimit -= moveOffset; B->writePos -= moveOffset;
}

static int Fill_Window(BT4 *B, int len_initial) {   /* :1389-1440 */
  int len = len_initial, actual_len = 0;
  if (B->readPos >= B->buf_len - B->keepSizeAfter) Move_Window(B);
  if (len > B->buf_len - B->writePos) len = B->buf_len - B->writePos;
  while (len > 0 && bt_More_Bytes(B)) {
    B->buf[B->writePos++] = bt_Read_Byte(B);
    len--; actual_len++;
  }
  if (B->writePos >= B->keepSizeAfter) B->readLimit = B->writePos - B->keepSizeAfter;
  if (B->pendingSize > 0 && B->readPos < B->readLimit) {       /* processPendingBytes :1397-1406 */
    int oldPendingSize = B->pendingSize;
    B->readPos -= B->pendingSize;
    B->pendingSize = 0;
    BT4_Skip(B, oldPendingSize);
  }
  return actual_len;
}

static int Compute_Match_Length(const BT4 *B, int distance, int length_limit) {   /* :1442-1460 */
  int back_pos = B->readPos - distance, len = 0;
  if (distance < 2) return 0;
  while (len < length_limit && B->buf[B->readPos + len] == B->buf[back_pos + len]) len++;
  return len;
}

static inline int Has_much_smaller_Distance(int smallDist, int bigDist) { return (smallDist - 1) < (bigDist - 1) / 128; }   /* :1469-1473 */

static void Read_One_and_Get_Matches(BT4 *B, Matches_Type *matches) {   /* :1477-1503 (LZMA_friendly = True) */
  int avail, len;
  B->readAhead++;
  BT4_Read_One_and_Get_Matches(B, matches);
  B->best_length_for_rep_dist = 0;
  avail = Get_Available(B) < B->Look_Ahead ? Get_Available(B) : B->Look_Ahead;
  if (avail >= B->MATCH_LEN_MIN) {
    for (int rep = 0; rep < 4; rep++) {
      len = Compute_Match_Length(B, B->rep_dist[rep], avail);
      B->len_rep_dist[rep] = len;
      if (len > B->best_length_for_rep_dist) { B->best_rep_dist_index = rep; B->best_length_for_rep_dist = len; }
    }
  } else {
    for (int rep = 0; rep < 4; rep++) B->len_rep_dist[rep] = 0;
  }
}

static void Get_supplemental_Matches_from_Repeat_Matches(BT4 *B, Matches_Type *matches) {   /* :1505-1566 */
  int len, ins;
  if (matches->count == 0) {
    if (B->best_length_for_rep_dist >= B->MATCH_LEN_MIN) {
      matches->dl[1].distance = B->rep_dist[B->best_rep_dist_index];
      matches->dl[1].length = B->best_length_for_rep_dist;
      matches->count = 1;
    }
  }
  for (int rep = 0; rep < 4; rep++) {
    len = B->len_rep_dist[rep];
    if (len >= B->MATCH_LEN_MIN) {
      ins = 0;
      for (int i = matches->count; i >= 1; i--) {
        if (len == matches->dl[i].length) {
          if (B->rep_dist[rep] == matches->dl[i].distance) {
            /* identical match */
          } else {
            ins = Has_much_smaller_Distance(matches->dl[i].distance, B->rep_dist[rep]) ? i : i + 1;
            break;
          }
        } else if (i < matches->count) {
          if (len > matches->dl[i].length && len < matches->dl[i + 1].length) { ins = i + 1; break; }
        } else if (len > matches->dl[i].length) {
          ins = i + 1;
          break;
        }
      }
      if (ins > 0) {
        for (int i = matches->count; i >= ins; i--) matches->dl[i + 1] = matches->dl[i];
        matches->dl[ins].distance = B->rep_dist[rep];
        matches->dl[ins].length = len;
        matches->count++;
        break;
      }
    }
  }
}

static void LZ_Skip(BT4 *B, int len) { B->readAhead += len; BT4_Skip(B, len); }   /* :1568-1573 */

static void Reduce_consecutive_max_lengths(Matches_Type *m) {   /* :1575-1585 */
  while (m->count > 1 && m->dl[m->count].length == m->dl[m->count - 1].length + 1 &&
         Has_much_smaller_Distance(m->dl[m->count - 1].distance, m->dl[m->count].distance))
    m->count--;
}

static void Send_first_literal_of_match(BT4 *B) { LZ77_emits_literal_byte(B->L, B->cur_literal); B->readAhead--; }   /* :1621-1625 */

static void Send_DL_code(BT4 *B, int distance, int length) {   /* :1627-1659 */
  int found_repeat = -1, aux;
  LZ77_emits_DL_code(B->L, distance, length);
  B->readAhead -= length;
  for (int i = 0; i < 4; i++) if (distance == B->rep_dist[i]) { found_repeat = i; break; }
  if (found_repeat >= 0) {
    aux = B->rep_dist[found_repeat];
    for (int i = found_repeat; i >= 1; i--) B->rep_dist[i] = B->rep_dist[i - 1];
    B->rep_dist[0] = aux;
  } else {
    for (int i = 3; i >= 1; i--) B->rep_dist[i] = B->rep_dist[i - 1];
    B->rep_dist[0] = distance;
  }
}

static void Get_Next_Symbol(BT4 *B) {   /* :1605-1796 */
  DLP new_ld, main;
  int avail, limit, index_max_score, set_max_score;
  const int hurdle = 40;
  Matches_Type *cur;
  if (B->readAhead == -1) Read_One_and_Get_Matches(B, &B->matches[B->current_match_index]);
  B->cur_literal = B->buf[B->readPos];
  avail = Get_Available(B) < B->Look_Ahead ? Get_Available(B) : B->Look_Ahead;
  if (avail < B->MATCH_LEN_MIN) { Send_first_literal_of_match(B); return; }
  if (B->best_length_for_rep_dist >= B->Nice_Length) {
    LZ_Skip(B, B->best_length_for_rep_dist - 1);
    Send_DL_code(B, B->rep_dist[B->best_rep_dist_index], B->best_length_for_rep_dist);
    return;
  }
  main.length = 1; main.distance = 1;
  cur = &B->matches[B->current_match_index];
  if (cur->count > 0) {
    main = cur->dl[cur->count];
    if (main.length >= B->Nice_Length) {
      LZ_Skip(B, main.length - 1);
      Send_DL_code(B, main.distance, main.length);
      return;
    }
    Reduce_consecutive_max_lengths(cur);
    Get_supplemental_Matches_from_Repeat_Matches(B, cur);
    main = cur->dl[cur->count];
    if (main.length == B->MATCH_LEN_MIN && main.distance > 128) main.length = 1;
  }
  if (B->best_length_for_rep_dist > B->MATCH_LEN_MIN &&
      (B->best_length_for_rep_dist >= main.length ||
       (B->best_length_for_rep_dist >= main.length - 2 && main.distance > (1 << 9)) ||
       (B->best_length_for_rep_dist >= main.length - 3 && main.distance > (1 << 15)))) {
    LZ_Skip(B, B->best_length_for_rep_dist - 1);
    Send_DL_code(B, B->rep_dist[B->best_rep_dist_index], B->best_length_for_rep_dist);
    return;
  }
  if (main.length < B->MATCH_LEN_MIN || avail <= B->MATCH_LEN_MIN) { Send_first_literal_of_match(B); return; }
  B->current_match_index = 1 - B->current_match_index;
  Read_One_and_Get_Matches(B, &B->matches[B->current_match_index]);
  cur = &B->matches[B->current_match_index];
  if (cur->count > 0) {
    new_ld = cur->dl[cur->count];
    if ((new_ld.length >= main.length + hurdle && new_ld.distance < main.distance) ||
        (new_ld.length == main.length + hurdle + 1 && !Has_much_smaller_Distance(main.distance, new_ld.distance)) ||
        new_ld.length > main.length + hurdle + 1 ||
        (new_ld.length >= main.length + hurdle - 1 && main.length >= B->MATCH_LEN_MIN + 1 && Has_much_smaller_Distance(new_ld.distance, main.distance))) {
      Send_first_literal_of_match(B);
      return;
    }
    Reduce_consecutive_max_lengths(cur);
    Get_supplemental_Matches_from_Repeat_Matches(B, cur);
    Estimate_DL_Codes_for_LZ77(B->L, B->matches, 1 - B->current_match_index, B->cur_literal, &index_max_score, &set_max_score);
    if (set_max_score == 1 - B->current_match_index) main = B->matches[set_max_score].dl[index_max_score];
    else { Send_first_literal_of_match(B); return; }
  }
  limit = main.length - 1 > B->MATCH_LEN_MIN ? main.length - 1 : B->MATCH_LEN_MIN;
  for (int rep = 0; rep < 4; rep++)
    if (Compute_Match_Length(B, B->rep_dist[rep], limit) == limit) { Send_first_literal_of_match(B); return; }
  LZ_Skip(B, main.length - 2);
  Send_DL_code(B, main.distance, main.length);
}

/* The set-up of LZ77_using_BT4 (lz77.adb:953-1126, 1798-1810): window sizes, hash tables, tree. */
static BT4 *bt4_open(Lz *L, int String_buffer_size, int Look_Ahead, int Threshold) {
  BT4 *B = (BT4 *)calloc(1, sizeof(BT4));
  if (!B) return NULL;
  B->L = L; B->String_buffer_size = String_buffer_size; B->Look_Ahead = Look_Ahead; B->Threshold = Threshold;
  B->MATCH_LEN_MIN = Threshold + 1;
  B->readPos = -1; B->readLimit = -1; B->writePos = 0; B->pendingSize = 0;
  B->keepSizeBefore = OPTS + String_buffer_size;
  B->keepSizeAfter = OPTS + Look_Ahead;
  {
    int64_t a = (int64_t)String_buffer_size / 2 + 256 * 1024, b = 512LL << 20;
    B->reserveSize = (int)(a < b ? a : b);
  }
  B->getBufSize = B->keepSizeBefore + B->keepSizeAfter + B->reserveSize;
  B->buf_len = B->getBufSize + 1;
  B->hash_4_size = getHash4Size(String_buffer_size);
  B->hash_4_mask = (uint32_t)B->hash_4_size - 1;
  B->hash2Table = (int32_t *)calloc(HASH_2_SIZE, sizeof(int32_t));
  B->hash3Table = (int32_t *)calloc(HASH_3_SIZE, sizeof(int32_t));
  B->hash4Table = (int32_t *)calloc((size_t)B->hash_4_size, sizeof(int32_t));
  for (int i = 0; i < 256; i++) {
    uint32_t r = (uint32_t)i;
    for (int j = 0; j < 8; j++) r = (r & 1) ? (r >> 1) ^ 0xEDB88320u : r >> 1;
    B->crcTable[i] = r;
  }
  B->Nice_Length = 162 < Look_Ahead ? 162 : Look_Ahead;
  B->cyclicSize = String_buffer_size;
  B->cyclicPos = -1;
  B->lzPos = B->cyclicSize;
  B->max_dist = B->cyclicSize - (Look_Ahead + 2);
  B->buf = (uint8_t *)calloc((size_t)B->buf_len + 8, 1);
  B->tree = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)B->cyclicSize);
  if (!B->hash2Table || !B->hash3Table || !B->hash4Table || !B->buf || !B->tree) return B;   /* (the caller tests the tables) */
  for (int64_t i = 0; i < 2 * (int64_t)B->cyclicSize; i++) B->tree[i] = Null_position;
  B->readAhead = -1;
  for (int i = 0; i < 4; i++) { B->rep_dist[i] = 1; B->len_rep_dist[i] = 0; }
  B->current_match_index = 0;
  return B;
}
static int bt4_ready(const BT4 *B) { return B && B->hash2Table && B->hash3Table && B->hash4Table && B->buf && B->tree; }
static void bt4_close(BT4 *B) {
  if (!B) return;
  free(B->hash2Table); free(B->hash3Table); free(B->hash4Table); free(B->buf); free(B->tree); free(B);
}

static int LZ77_using_BT4(Lz *L, int String_buffer_size, int Look_Ahead, int Threshold) {
  BT4 *B = bt4_open(L, String_buffer_size, Look_Ahead, Threshold);
  int actual_written, rc = ZO_OK;
  if (!bt4_ready(B)) { rc = ZO_ENOMEM; goto done; }
  actual_written = Fill_Window(B, String_buffer_size);
  if (actual_written > 0) {
    for (;;) {
      Get_Next_Symbol(B);
      if (Get_Available(B) == 0) {
        actual_written = Fill_Window(B, String_buffer_size);
        if (actual_written == 0) break;
      }
    }
  }
done:
  bt4_close(B);
  return rc;
}

/* Stage export for the tests: the match set BT4_Algo.Read_One_and_Get_Matches (lz77.adb:1234-1361) returns at EVERY position of the
   input when every position is read (none skipped), with the window filled as LZ77_using_BT4's main loop fills it (:1798-1827) --
   Skip (:1208-1232) and Read_One share the tree update (:1154-1206), so the tree after position p, hence every set, does not depend
   on which positions the coder reads and which it skips.  cnt [p] = number of matches at p; len / dist [p * stride + i], i < cnt [p].
   sbs = the String_buffer_size of Level_3 for this dictionary_size (lzma-encoding.adb:137-149). */
int zo_bt4_match_sets(const uint8_t *in, uint64_t n, int64_t dictionary_size, uint8_t *cnt, uint16_t *len, uint32_t *dist, int stride) {
  Lz *L = (Lz *)calloc(1, sizeof(Lz));
  BT4 *B;
  int64_t sbs;
  uint64_t p = 0;
  int rc = ZO_OK;
  Matches_Type *m = (Matches_Type *)malloc(sizeof(Matches_Type));
  if (!L || !m) { free(L); free(m); return ZO_ENOMEM; }
  sbs = Ceiling_power_of_2(dictionary_size + 273 + 1 + 64);
  if (sbs > (1 << 28)) sbs = 1 << 28;
  if (sbs < Min_dictionary_size) sbs = Min_dictionary_size;
  L->in = in; L->n = n;
  B = bt4_open(L, (int)sbs, 273, 1);
  if (!bt4_ready(B)) { rc = ZO_ENOMEM; goto done; }
  memset(cnt, 0, (size_t)n);
  if (Fill_Window(B, (int)sbs) > 0) {
    for (;;) {
      BT4_Read_One_and_Get_Matches(B, m);
      if (p >= n || m->count > stride) { rc = ZO_EINVAL; goto done; }
      cnt[p] = (uint8_t)m->count;
      for (int i = 1; i <= m->count; i++) { len[p * (uint64_t)stride + (i - 1)] = (uint16_t)m->dl[i].length; dist[p * (uint64_t)stride + (i - 1)] = (uint32_t)m->dl[i].distance; }
      p++;
      if (Get_Available(B) == 0 && Fill_Window(B, (int)sbs) == 0) break;
    }
  }
  if (p != n) rc = ZO_EINVAL;
done:
  bt4_close(B); free(m); free(L);
  return rc;
}

/* ---------------------------------------------------------------- Encode (:59-170, :1513-1563) */

static void fill_probs(CProb *p, size_t count) { for (size_t i = 0; i < count; i++) p[i] = initial_probability; }

int zo_lzma_encode(const uint8_t *in, uint64_t n, int level, int lc, int lp, int pb, int end_marker, int64_t dictionary_size,
                   uint8_t *out, uint64_t cap, uint64_t *out_len, uint64_t *stats8) {
  Lz *L;
  int rc = ZO_OK;
  int64_t sbs;
  if (level < 0 || level > 3 || lc < 0 || lc > 8 || lp < 0 || lp > 4 || pb < 0 || pb > 4) return ZO_EINVAL;
  L = (Lz *)calloc(1, sizeof(Lz));
  if (!L) return ZO_ENOMEM;
  L->level = level; L->lc = lc; L->lp = lp; L->pb = pb;
  L->compare_variants = level <= 1 ? CV_None : level == 2 ? CV_Simple : CV_Splitting;       /* :1539-1546 */
  if (level == 0) sbs = 16;                                                                  /* :137-149 */
  else if (level <= 2) sbs = 1 << 15;
  else {
    sbs = Ceiling_power_of_2(dictionary_size + 273 + 1 + 64);
    if (sbs > (1 << 28)) sbs = 1 << 28;
    if (sbs < Min_dictionary_size) sbs = Min_dictionary_size;
  }
  L->String_buffer_size = (int)sbs;
  L->Text_Buf_Mask = (uint32_t)sbs - 1;
  L->pos_bits_mask = (1u << pb) - 1;
  L->literal_pos_mask = (1u << lp) - 1;
  L->lit_count = 0x300 << (lc + lp);
  L->lit = (CProb *)malloc(sizeof(CProb) * (size_t)L->lit_count);
  L->Text_Buf = (uint8_t *)calloc((size_t)sbs, 1);
  if (!L->lit || !L->Text_Buf) { rc = ZO_ENOMEM; goto done; }
  fill_probs(L->lit, (size_t)L->lit_count);
  fill_probs((CProb *)&L->dist, sizeof L->dist / sizeof(CProb));
  fill_probs((CProb *)&L->len, sizeof L->len / sizeof(CProb));
  fill_probs((CProb *)&L->rep_len, sizeof L->rep_len / sizeof(CProb));
  fill_probs((CProb *)&L->sw, sizeof L->sw / sizeof(CProb));
  L->width = 0xFFFFFFFFu; L->low = 0; L->cache = 0; L->cache_size = 1;
  L->out = out; L->cap = cap; L->in = in; L->n = n;
  /* Write_LZMA_header :1513-1536 (header_has_size = False in Zip entries) */
  Write_Byte(L, (uint8_t)(lc + 9 * lp + 9 * 5 * pb));
  { uint32_t dw = (uint32_t)sbs; for (int i = 0; i < 4; i++) { Write_Byte(L, (uint8_t)(dw & 255)); dw >>= 8; } }
  /* My_LZ77 :1500-1511 */
  if (level == 0) {
    for (uint64_t i = 0; i < n; i++) LZ77_emits_literal_byte(L, in[i]);                      /* No_LZ77, lz77.adb:2148-2179 */
  } else if (level <= 2) {
    uint64_t cap_t = n + 16, nt;
    uint32_t *tokens = (uint32_t *)malloc(sizeof(uint32_t) * cap_t);
    if (!tokens) { rc = ZO_ENOMEM; goto done; }
    nt = zo_lz77_tokens(in, n, level == 1 ? 6 : 10, tokens, cap_t);                          /* IZ_6 / IZ_10 :118-122 */
    for (uint64_t t = 0; t < nt; t++) {
      uint32_t tk = tokens[t];
      if (tk & 0x80000000u) LZ77_emits_DL_code(L, (int)(tk & 0xFFFF), (int)((tk >> 16) & 0x7FFF));
      else LZ77_emits_literal_byte(L, (uint8_t)tk);
    }
    free(tokens);
  } else {
    rc = LZ77_using_BT4(L, (int)sbs, 273, 1);
    if (rc != ZO_OK) goto done;
  }
  if (end_marker) {                                                                          /* :1549-1556 */
    Encode_Bit(L, &L->sw.match[L->ES.state][L->ES.pos_state], 1);
    Write_Simple_Match(L, end_of_stream_magic_distance, Min_match_length);
  }
  Flush_range_encoder(L);
  if (out_len) *out_len = L->out_len;
  if (stats8) memcpy(stats8, L->stat, sizeof L->stat);
  if (L->out_len > cap) rc = ZO_EINVAL;
  if (L->ES.total_pos != n) rc = ZO_EINVAL;       /* (encoded_uncompressed_bytes counts expanded DL codes twice, as in the reference) */
done:
  free(L->lit); free(L->Text_Buf); free(L);
  return rc;
}

/* Zip.Compress.LZMA_E (zip-compress-lzma_e.adb:121-172): methods LZMA_0 .. LZMA_3 = 15 .. 18. */
int zo_lzma(const uint8_t *in, uint64_t n, int method, uint8_t *out, uint64_t cap, uint64_t *out_len, uint32_t *crc_inout) {
  uint64_t len = 0;
  int rc;
  if (method < ZO_LZMA_0 || method > ZO_LZMA_3) return ZO_EINVAL;
  if (cap < 4) return ZO_EINVAL;
  out[0] = 16; out[1] = 2; out[2] = 5; out[3] = 0;                                           /* :155-158 */
  rc = zo_lzma_encode(in, n, method - ZO_LZMA_0, 3, 0, 2, 1, (int64_t)n, out + 4, cap - 4, &len, NULL);
  len += 4;
  if (out_len) *out_len = len;
  if (rc == ZO_EINVAL && len > cap && len >= n) rc = ZO_INEFFICIENT;
  if (rc < 0) return rc;
  if (crc_inout) *crc_inout = zo_crc32_update(*crc_inout, in, n);
  return len >= n ? ZO_INEFFICIENT : ZO_OK;
}
