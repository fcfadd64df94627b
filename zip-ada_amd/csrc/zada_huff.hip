// zada_huff.hip -- entropy stage on gfx950: the "Taillaule" block splitter, block-format
// chooser, Huffman code construction and bit emission of zip_lib/zip-compress-deflate.adb,
// re-expressed over ONE global array of LZ atoms (token + byte position):
//
//   k_window_descr   every descriptor the scanner can ask for (initial window per 65536-atom
//                    flush :1338-1360, sliding windows every 750 atoms :1372), one wave each:
//                    histogram (Get_statistics :953-976) + length-limited code lengths
//   k_cut_scan       Scan_and_send_from_main_buffer's similarity walk (:1363-1401), one wave
//                    per flush; emits the cut positions = candidate blocks
//   k_block_analyze  per candidate block: statistics, both descriptor variants (:1213-1219),
//                    header costs (:549-704) and data costs (:1147-1209)
//   k_choose         the one inherently sequential step: the format decision chain of
//                    Send_as_block (:1222-1268) + Mark_new_block (:999-1007) + stream epilogue
//                    (:1613-1635), a single wave walking the block table; assigns bit offsets
//   k_block_codes / k_tile_bits / k_tile_scan / k_emit_tiles / k_emit_headers / k_copy_pieces
//                    parallel emission: per-atom (code, length) -> prefix sums -> bit packing
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "zada_logic.h"
#include "zada_internal.h"
#include "zada_llhc_wave.h"

namespace zada {

// --------------------------------------------------------------------------------------------
// helpers
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void atom_symbols(uint32_t t, int &ls, int &ds) {
  if (tok_is_match(t)) { ls = len_symbol((int)((t >> 16) & 0x1FF)); ds = dist_symbol((int)(t & 0xFFFF)); }
  else { ls = (int)(t & 0xFF); ds = -1; }
}

__device__ __forceinline__ void put_bits_global(uint32_t *out32, uint64_t pos, uint32_t value, int nbits) {
  if (nbits <= 0) return;
  uint64_t w = pos >> 5; int sh = (int)(pos & 31);
  uint64_t v = (uint64_t)value << sh;
  atomicOr(&out32[w], (uint32_t)v);
  if (sh + nbits > 32) atomicOr(&out32[w + 1], (uint32_t)(v >> 32));
}

// the same for the writers that run BEFORE the host has seen the total size: a stream that outgrows the workspace
// (only possible when it is also larger than the input, i.e. Compression_inefficient) must not be written past its end
__device__ __forceinline__ void put_bits_lim(uint32_t *out32, uint64_t lim_bits, uint64_t pos, uint32_t value, int nbits) {
  if (pos + 64 <= lim_bits) put_bits_global(out32, pos, value, nbits);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// geometry of owned flush j: first atom F, last atom `to`, number in its stream gj; false: empty slot (batches only)
__device__ __forceinline__ bool flush_geom(const EntropyView &v, uint32_t j, uint32_t &F, uint32_t &to, uint64_t &gj, uint32_t &flags) {
  if (v.ftab) {
    const FlushGeom g = v.ftab[j];
    F = g.F; to = g.to; gj = g.gj; flags = g.flags;
    return !(g.flags & FG_EMPTY);
  }
  F = v.foff + j * FLUSH;
  to = (F + FLUSH - 1 < v.lvalid - 1) ? F + FLUSH - 1 : v.lvalid - 1;
  gj = v.j0 + j;
  // the stream's last flush, if it is not a full one, is sent with last_flush = True (:1613-1623)
  flags = ((j == v.nflush - 1) && v.stream_final && (v.lvalid - F < FLUSH)) ? (uint32_t)FG_LAST_PARTIAL : 0u;
  return true;
}

// --------------------------------------------------------------------------------------------
// k_window_descr : one wave per flush segment and GROUP of WD_GROUP neighbouring slots
// Round 6: neighbouring sliding windows (750 atoms apart, 4 097 atoms each) share 82 % of their atoms: a wave takes WD_GROUP slots one after the other and
// moves the histogram from one window to the next -- the 750 atoms that leave are taken off, the 750 that come are added (1 500 visits instead of
// 4 097) -- instead of counting every window from nothing.  The distance patch works in place: the raw distance counts wait in registers.  (Slot 0 -- the flush's initial window --, the empty windows of the null-slice quirk and the first window of a group are counted whole.)
// --------------------------------------------------------------------------------------------
#ifndef ZADA_WD_GROUP
#define ZADA_WD_GROUP 4
#endif
constexpr uint32_t WD_GROUP = ZADA_WD_GROUP, WD_NGROUPS = (SLOTS + WD_GROUP - 1) / WD_GROUP;
__global__ void __launch_bounds__(64) k_window_descr(EntropyView v, uint32_t kstep, uint8_t *__restrict__ descr) {
  __shared__ uint32_t hist[320];
  __shared__ uint8_t bl[320];
  __shared__ __attribute__((aligned(16))) uint8_t S[LLHC_WAVE_SCRATCH];
  const uint32_t *__restrict__ atoms = v.atoms;
  const uint32_t grp = blockIdx.x % WD_NGROUPS, j = blockIdx.x / WD_NGROUPS;   // j: owned flush (local number)
  uint64_t gj; uint32_t F, to, fl;                                          // its number in the stream, its first and last atom
  if (!flush_geom(v, j, F, to, gj, fl)) return;
  if (to - F < SLIDER - 1) return;                                  // :1333-1336 short flush: no scanning
  const int lane = threadIdx.x;
  // counts of the atoms [a, b] go to (add) or leave (!add) the histogram, eight loads in flight, then their counts
  auto count = [&](int64_t a, int64_t b, bool add) {
    for (int64_t a0 = a + lane; a0 <= b; a0 += 64 * 8) {
      uint32_t at[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int64_t x = a0 + 64 * u; at[u] = atoms[x <= b ? x : b]; }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (a0 + 64 * u <= b) {
          int ls, ds; atom_symbols(at[u], ls, ds);
          if (add) { atomicAdd(&hist[ls], 1u); if (ds >= 0) atomicAdd(&hist[288 + ds], 1u); }
          else { atomicSub(&hist[ls], 1u); if (ds >= 0) atomicSub(&hist[288 + ds], 1u); }
        }
      }
    }
  };
  bool have = false;                                                 // `hist` holds the counts of the window [plo, phi]
  int64_t plo = 0, phi = -1;
  for (uint32_t slot = grp * WD_GROUP; slot < (grp + 1) * WD_GROUP && slot < SLOTS; slot++) {
    int64_t lo, hi;
    if (slot == 0) { lo = (gj == 0) ? (int64_t)F : (int64_t)F - HALF_SLIDER; hi = lo + SLIDER - 1; }   // :1338-1360
    else {
      const uint32_t m = F + MIN_STEP * slot;
      if (!((uint64_t)m + HALF_SLIDER < to)) break;                   // :1364 (nor any later slot)
      if (slot % kstep) continue;
      lo = (int64_t)m - HALF_SLIDER; hi = (int64_t)m + HALF_SLIDER;
      // ring-index wrap => Ada null slice (:1372): first-half flushes only
      if ((gj & 1) == 0 && MIN_STEP * slot < HALF_SLIDER) { lo = 0; hi = -1; }
    }
    __syncthreads();
    const bool slide = have && slot != 0 && hi >= lo && lo > plo && lo <= phi && hi > phi;      // the window before overlaps this one
    if (slide) { count(plo, lo - 1, false); count(phi + 1, hi, true); }
    else {
      for (int i = lane; i < 320; i += 64) hist[i] = (i == 256) ? 1u : 0u;     // empty_lit_len_stat :946
      __syncthreads();
      count(lo, hi, true);
    }
    have = hi >= lo && slot != 0; plo = lo; phi = hi;
    __syncthreads();
    const uint32_t raw_dist = lane < 32 ? hist[288 + lane] : 0u;      // (the patch changes the distance counts in place: the raw ones go on to the next window)
    __syncthreads();
    if (lane == 0) patch_dist_stats(hist + 288);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 2; t++) llhc_wave<15>(hist + 288 * t, t ? 32 : 288, bl + 288 * t, S, lane);
    __syncthreads();
    if (lane < 32) hist[288 + lane] = raw_dist;
    uint8_t *dst = descr + ((uint64_t)j * SLOTS + slot) * 320;
    for (int i = lane; i < 320; i += 64) dst[i] = bl[i];
  }
}

// --------------------------------------------------------------------------------------------
// k_cut_scan : one wave per flush segment
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_cut_scan(EntropyView v, uint32_t kstep, const uint8_t *__restrict__ descr,
                                                 uint32_t *__restrict__ seg_nblk, uint32_t *__restrict__ seg_cut, uint32_t *__restrict__ trace) {
  const uint32_t j = blockIdx.x, lane = threadIdx.x;
  uint64_t gj; uint32_t F, to, fl;
  if (!flush_geom(v, j, F, to, gj, fl)) { if (lane == 0) seg_nblk[j] = 0; return; }
  uint32_t *cuts = seg_cut + (uint64_t)j * MAXBLK_PER_SEG;
  uint32_t nb = 0;
  if (lane == 0) cuts[0] = F;
  nb = 1;
  for (uint32_t k = lane; k < SLOTS; k += 64) { trace[((uint64_t)j * SLOTS + k) * 2] = 0xFFFFFFFFu; trace[((uint64_t)j * SLOTS + k) * 2 + 1] = 0; }   // no test here
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
  if (to - F >= SLIDER - 1) {
    const uint8_t *d0 = descr + (uint64_t)j * SLOTS * 320;
    int init[5];
    for (int r = 0; r < 5; r++) init[r] = tweak_value(d0[lane + 64 * r]);
    for (uint32_t k = 1; (uint64_t)F + MIN_STEP * k + HALF_SLIDER < to; k++) {
      if (k % kstep) continue;
      const int thr = (k % 8 == 0) ? 420 : (k % 4 == 0) ? 430 : 2050;          // step_choice :1304-1308
      const uint8_t *dk = d0 + (uint64_t)k * 320;
      int sl[5]; uint32_t dist = 0;
      for (int r = 0; r < 5; r++) { sl[r] = tweak_value(dk[lane + 64 * r]); int d = init[r] - sl[r]; dist += (uint32_t)(d < 0 ? -d : d); }
      dist = wave_sum_u32(dist);
      // the reference's trace (:480-488, 1384-1390): the distance at every test point, and which step level cut
      if (lane == 0) { uint32_t *t = trace + ((uint64_t)j * SLOTS + k) * 2; t[0] = dist; t[1] = dist < (uint32_t)thr * 100u ? 0u : ((k % 8 == 0) ? 1u : (k % 4 == 0) ? 2u : 3u); }
      if (!(dist < (uint32_t)thr * 100u)) {                                      // not Similar => cut (:1375-1395)
        if (lane == 0) cuts[nb] = F + MIN_STEP * k;
        nb++;
        for (int r = 0; r < 5; r++) init[r] = sl[r];
      }
    }
  }
  if (lane == 0) seg_nblk[j] = nb;
}

__global__ void k_fill_blocks(EntropyView v, const uint32_t *__restrict__ seg_nblk, const uint32_t *__restrict__ seg_cut,
                              const uint32_t *__restrict__ seg_off, BlockRange *__restrict__ blocks, uint32_t *__restrict__ blk_entry, int fixed_only) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= v.nflush) return;
  uint64_t gj; uint32_t F, to, fl;
  if (!flush_geom(v, j, F, to, gj, fl)) return;
  const uint32_t nb = seg_nblk[j], off = seg_off[j];
  const uint32_t *cuts = seg_cut + (uint64_t)j * MAXBLK_PER_SEG;
  for (uint32_t i = 0; i < nb; i++) {
    uint32_t first = cuts[i], end = (i + 1 < nb) ? cuts[i + 1] - 1 : to;
    BlockRange b; b.first = first; b.count = end - first + 1; b.pad = 0;
    b.last_flush = ((fl & FG_LAST_PARTIAL) && i + 1 == nb) ? (uint32_t)BR_LAST_FLUSH : 0u;
    // Deflate_Fixed (batches): ONE fixed block per entry, marked final where it is opened (:1600-1603)
    if (fixed_only) b.last_flush = ((fl & FG_ENTRY_FIRST) && i == 0) ? (uint32_t)BR_LAST_FLUSH : 0u;
    if (v.ftab) {
      if ((fl & FG_ENTRY_FIRST) && i == 0) b.last_flush |= BR_ENTRY_FIRST;
      if ((fl & FG_ENTRY_LAST) && i + 1 == nb) { b.last_flush |= BR_ENTRY_LAST; b.pad = v.ftab[j].end_byte; }
      blk_entry[off + i] = v.ftab[j].entry;
    }
    blocks[off + i] = b;
  }
}

// Deflate_Fixed: the range's own atoms in pieces of 65 536 (only to spread the cost analysis and the emission)
__global__ void k_fill_blocks_fixed(uint32_t first, uint32_t T, uint32_t nb, BlockRange *__restrict__ blocks) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nb) return;
  BlockRange b; b.first = first + j * FLUSH; b.count = (j + 1 == nb) ? T - j * FLUSH : FLUSH; b.last_flush = 0; b.pad = 0;
  blocks[j] = b;
}

// byte position behind the last atom of the local array (what apos[lvalid] would be): the blocks' byte counts are
// differences of positions
__global__ void k_apos_sentinel(const uint32_t *__restrict__ atoms, uint32_t *__restrict__ apos, uint32_t lvalid) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && lvalid > 0) apos[lvalid] = apos[lvalid - 1] + tok_len(atoms[lvalid - 1]);
}

// --------------------------------------------------------------------------------------------
// k_block_analyze : one workgroup (256 threads) per candidate block
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_block_analyze(const uint32_t *__restrict__ atoms, const uint32_t *__restrict__ apos,
                                                       const BlockRange *__restrict__ blocks, BlockInfo *__restrict__ binfo) {
  __shared__ uint32_t st1[320], st2[320], dtmp[2][32];
  __shared__ uint8_t bl1[320], bl2[320], good[320];
  __shared__ __attribute__((aligned(16))) uint8_t S[4][LLHC_WAVE_SCRATCH];
  __shared__ HeaderPlan hp[2];
  __shared__ uint64_t red[3][4];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const BlockRange br = blocks[blockIdx.x];
  for (int i = tid; i < 320; i += 256) st1[i] = (i == 256) ? 1u : 0u;
  __syncthreads();
  for (uint32_t a0 = tid; a0 < br.count; a0 += 256 * 8) {                 // eight loads in flight, then their counts
    uint32_t at[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const uint32_t a = a0 + 256u * u; at[u] = atoms[br.first + (a < br.count ? a : br.count - 1)]; }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (a0 + 256u * u < br.count) {
        int ls, ds; atom_symbols(at[u], ls, ds);
        atomicAdd(&st1[ls], 1u);
        if (ds >= 0) atomicAdd(&st1[288 + ds], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < 320; i += 256) st2[i] = st1[i];
  __syncthreads();
  if (tid == 0) tweak_for_better_rle(st2, 288, good);                       // :1217
  if (tid == 64) tweak_for_better_rle(st2 + 288, 32, good + 288);           // :1218
  __syncthreads();
  if (w >= 2 && lane == 0) {
    uint32_t *dt = dtmp[w - 2]; const uint32_t *src = (w == 2 ? st1 : st2) + 288;
    for (int i = 0; i < 32; i++) dt[i] = src[i];
    patch_dist_stats(dt);                                                     // :340-365, on a copy
  }
  wave_sync();
  llhc_wave<15>(w == 0 ? st1 : w == 1 ? st2 : dtmp[w - 2], w < 2 ? 288 : 32,      // one wave per code set
                ((w & 1) ? bl2 : bl1) + (w < 2 ? 0 : 288), S[w], lane);
  __syncthreads();
  if (w < 2) {                                                                // Put_Compression_Structure, cost analysis
    HeaderPlan *h = &hp[w];
    const uint8_t *ll = w == 0 ? bl1 : bl2;
    if (lane == 0) {
      int max_ll = 0, max_d = 0, idx = 0;
      for (int a = 287; a >= 0; a--) if (ll[a] > 0) { max_ll = a; break; }
      for (int a = 31; a >= 0; a--) if (ll[288 + a] > 0) { max_d = a; break; }
      for (int a = 0; a <= max_ll; a++) h->cs_bl[idx++] = ll[a];
      for (int a = 0; a <= max_d; a++) h->cs_bl[idx++] = ll[288 + a];
      h->last_cs_bl = (uint16_t)idx; h->hlit_m257 = (uint8_t)(max_ll - 256); h->hdist_m1 = (uint8_t)max_d;
      for (int a = 0; a < 19; a++) h->truc_freq[a] = 0;
      uint32_t *tf = h->truc_freq;
      header_rle_walk(h->cs_bl, idx, [tf](int x, uint32_t) { tf[x]++; });
    }
    wave_sync();
    llhc_wave<7>(h->truc_freq, 19, h->truc_bl, S[w], lane);
    if (lane == 0) {
      int anz = 3;
      for (int a = 0; a <= 18; a++) if (a > anz && h->truc_bl[header_perm(a)] > 0) anz = a;
      h->a_non_zero = (uint8_t)anz;
      uint32_t bits = 14 + (uint32_t)(1 + anz) * 3;
      for (int a = 0; a <= 18; a++) bits += h->truc_freq[a] * (uint32_t)(h->truc_bl[a] + header_extra_bits(a));
      h->bits = bits;
    }
  }
  // data costs, Compute_sizes_of_variants :1152-1190 (EOB, symbol 256, is never counted there)
  uint64_t cf = 0, c1 = 0, c2 = 0;
  for (int s = tid; s < 320; s += 256) {
    uint64_t c = st1[s];
    if (s < 288) {
      if (s != 256 && s <= 285) { int e = litlen_sym_extra(s); cf += c * (uint64_t)(fixed_litlen_bl(s) + e); c1 += c * (uint64_t)(bl1[s] + e); c2 += c * (uint64_t)(bl2[s] + e); }
    } else if (s - 288 <= 29) {
      int e = dist_sym_extra(s - 288); cf += c * (uint64_t)(5 + e); c1 += c * (uint64_t)(bl1[s] + e); c2 += c * (uint64_t)(bl2[s] + e);
    }
  }
  cf = wave_sum_u64(cf); c1 = wave_sum_u64(c1); c2 = wave_sum_u64(c2);
  if (lane == 0) { red[0][w] = cf; red[1][w] = c1; red[2][w] = c2; }
  __syncthreads();
  BlockInfo *bi = &binfo[blockIdx.x];
  for (int i = tid; i < 320; i += 256) { bi->stats[i] = st1[i]; bi->bl1[i] = bl1[i]; bi->bl2[i] = bl2[i]; }
  if (tid < 20) { bi->truc1[tid] = tid < 19 ? hp[0].truc_bl[tid] : hp[0].a_non_zero; bi->truc2[tid] = tid < 19 ? hp[1].truc_bl[tid] : hp[1].a_non_zero; }
  if (tid == 0) {
    bi->hdr1_bits = hp[0].bits; bi->hdr2_bits = hp[1].bits;
    bi->fixed_data = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    bi->dyn1_data = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    bi->dyn2_data = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    bi->bytes = (br.pad ? br.pad : apos[br.first + br.count]) - apos[br.first];     // (pad: the entry ends here, in a batch)
    uint32_t sp = 1;
    for (int i = 267; i <= 287; i++) if (st1[i] != 0) sp = 0;                // :1222 (Long_length_codes :1093-1095)
    bi->stored_possible = sp;
  }
}

// --------------------------------------------------------------------------------------------
// k_choose : single wave, sequential over the block table
// --------------------------------------------------------------------------------------------
// canonical (bit-reversed) code of symbol `sym` under lengths held 5-per-lane (symbol = lane + 64 r)
__device__ uint32_t wave_code_of(const int (&cur)[5], int sym, int lane, int nsyms_limit /*288: lit/len tree*/) {
  // length of sym
  int len = __shfl(cur[sym >> 6], sym & 63);
  if (len == 0) return 0;
  // next_code[len] = sum over shorter lengths: bl_count[l] << (len - l); plus rank among equal lengths below sym
  uint32_t part = 0;
  for (int r = 0; r < 5; r++) {
    int s = lane + 64 * r, l = cur[r];
    if (s < nsyms_limit && l != 0) {
      if (l < len) part += 1u << (len - l);
      else if (l == len && s < sym) part += 1u;
    }
  }
  part = wave_sum_u32(part);
  return bit_reverse(part, len);
}

// Per block, in parallel: everything the sequential chooser needs about "recycling" the codes that
// are most likely in force when the block is reached -- the fixed codes (c = 0) or one of the two
// descriptors of the PREVIOUS block (c = 1, 2): Recyclable (:495-508) and the recycled cost
// (:1158, 1180, 1189); plus the end-of-block code of the block's own two descriptors.
struct BlockRel { uint64_t bits[3]; uint32_t ok[3]; uint32_t eob[2]; uint32_t pad; };

// Everything the sequential chooser needs about one block, with all the state-independent work done: of the five
// costs of Compute_sizes_of_variants (:1147-1209) only "recycle" depends on what the previous blocks chose; the
// constant c (:1198-1201) is added to the other four alike, so their minimum and the format that attains it
// (tie order fixed, dynamic, dynamic-RLE: :1243-1250) are known in advance.
struct ChRec {
  uint64_t base3;                 // min(fixed, dyn1 + hdr1, dyn2 + hdr2) + 2, without c
  uint64_t stored;                // stored-block bits without c (:1193-1196), ~0 if a match is longer than 14
  uint64_t bits[3];               // LZ data under the fixed code / the previous block's plain / RLE-tweaked code
  uint64_t fixed_data, dyn1_data, dyn2_data;
  uint32_t hdr1, hdr2;
  uint32_t ok;                    // bit k: bits[k] is usable (Recyclable :495-508)
  uint32_t eob[2];                // this block's own end-of-block codes (len << 16 | code), plain / RLE-tweaked
  uint32_t fmt3;                  // FMT_FIXED / FMT_DYN1 / FMT_DYN2 attaining base3
  BlockRange br;
  uint32_t pad[6];
};
static_assert(sizeof(ChRec) == 128, "ChRec");

__global__ void __launch_bounds__(64) k_block_relate(uint32_t nblocks, const BlockInfo *__restrict__ binfo, const BlockRange *__restrict__ blocks,
                                                     ChRec *__restrict__ chrec) {
  const uint32_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const BlockInfo *bi = &binfo[i];
  uint32_t st[5]; int b1[5], b2[5];
  for (int r = 0; r < 5; r++) { st[r] = bi->stats[lane + 64 * r]; b1[r] = bi->bl1[lane + 64 * r]; b2[r] = bi->bl2[lane + 64 * r]; }
  BlockRel out;
  for (int c = 0; c < 3; c++) {
    int cur[5];
    for (int r = 0; r < 5; r++) {
      const int s = lane + 64 * r;
      if (c == 0 || i == 0) cur[r] = s < 288 ? fixed_litlen_bl(s) : 5;
      else cur[r] = c == 1 ? binfo[i - 1].bl1[s] : binfo[i - 1].bl2[s];
    }
    bool bad = false; uint64_t rc = 0;
    for (int r = 0; r < 5; r++) {
      const int s = lane + 64 * r;
      if (cur[r] == 0 && b1[r] > 0) bad = true;
      if (s < 288) { if (s != 256 && s <= 285) rc += (uint64_t)st[r] * (uint64_t)(cur[r] + litlen_sym_extra(s)); }
      else if (s - 288 <= 29) rc += (uint64_t)st[r] * (uint64_t)(cur[r] + dist_sym_extra(s - 288));
    }
    out.ok[c] = __any(bad) ? 0u : 1u;
    out.bits[c] = wave_sum_u64(rc);
  }
  out.eob[0] = ((uint32_t)__shfl(b1[4], 0) << 16) | wave_code_of(b1, 256, lane, 288);
  out.eob[1] = ((uint32_t)__shfl(b2[4], 0) << 16) | wave_code_of(b2, 256, lane, 288);
  out.pad = 0;
  if (lane == 0) {
    ChRec cr;
    const uint64_t fx = bi->fixed_data + 2, d1 = bi->dyn1_data + 2 + bi->hdr1_bits, d2 = bi->dyn2_data + 2 + bi->hdr2_bits;
    uint64_t m = fx; uint32_t f = FMT_FIXED;
    if (d1 < m) { m = d1; f = FMT_DYN1; }
    if (d2 < m) { m = d2; f = FMT_DYN2; }
    cr.base3 = m; cr.fmt3 = f;
    cr.stored = ~0ull;
    if (bi->stored_possible) { uint64_t sb = 8ull * bi->bytes; sb += (1 + (sb / 8) / 65535) * 40; cr.stored = sb; }   // :1193-1196
    for (int k = 0; k < 3; k++) cr.bits[k] = out.bits[k];
    cr.ok = (out.ok[0] ? 1u : 0u) | (out.ok[1] ? 2u : 0u) | (out.ok[2] ? 4u : 0u);
    cr.fixed_data = bi->fixed_data; cr.dyn1_data = bi->dyn1_data; cr.dyn2_data = bi->dyn2_data;
    cr.hdr1 = bi->hdr1_bits; cr.hdr2 = bi->hdr2_bits;
    cr.eob[0] = out.eob[0]; cr.eob[1] = out.eob[1];
    cr.br = blocks[i];
    for (int k = 0; k < 6; k++) cr.pad[k] = 0;
    chrec[i] = cr;
  }
}

struct ChooseState {
  int last_type, block_to_finish, last_marked;
  int code_block, code_variant;
  uint32_t cur_eob;                                  // (length << 16) | code of symbol 256 under curr_descr
  uint64_t pos;
};

// Deflate_Fixed: one fixed block for the whole stream (:1600-1603, :1615-1616).  The pseudo-blocks (one per 65 536
// atoms) only spread the cost analysis; their data positions are a running sum.  The range that starts the stream writes
// the block header, the one that ends it the end-of-block code.
__global__ void __launch_bounds__(64) k_choose_fixed(uint32_t nblocks, const BlockRange *__restrict__ blocks, const BlockInfo *__restrict__ binfo,
                                                     EmitRec *__restrict__ emit, uint32_t *__restrict__ tile_block, uint32_t cap_tiles,
                                                     uint32_t *__restrict__ out32, uint64_t lim_bits, ChooserOut *__restrict__ res,
                                                     const ChooserCarry *__restrict__ cin, ChooserCarry *__restrict__ cout, uint64_t base_bits,
                                                     int stream_first, int stream_last) {
  const int lane = threadIdx.x;
  uint32_t ntiles = 0, overflow = 0;
  uint64_t pos = cin->pos - base_bits;
  if (stream_first) {
    if (lane == 0) { put_bits_lim(out32, lim_bits, pos, 1, 1); put_bits_lim(out32, lim_bits, pos + 1, 1, 2); }
    pos += 3;
  }
  for (uint32_t i = 0; i < nblocks; i++) {
    const BlockRange br = blocks[i];
    const uint64_t d = binfo[i].fixed_data;
    const uint32_t nt = (br.count + TILE - 1) / TILE;
    if (lane == 0) {
      EmitRec e; e.hdr_bitpos = 0; e.data_bitpos = pos; e.cost_bits = d; e.fmt = FMT_FIXED; e.code_block = CODE_FIXED; e.code_variant = 0; e.tile_base = ntiles;
      e.pre_pos = 0; e.pre_eob = 0; e.pre_flags = 0;
      emit[i] = e;
    }
    if (ntiles + nt > cap_tiles) { overflow = 1; break; }
    for (uint32_t t = lane; t < nt; t += 64) tile_block[ntiles + t] = i;
    ntiles += nt;
    pos += d;
  }
  if (stream_last) pos += 7;                         // fixed EOB = 7 zero bits
  if (lane == 0) {
    res->total_bits = pos; res->n_tiles = ntiles; res->n_pieces = 0; res->n_blocks = nblocks; res->overflow = overflow;
    cout->pos = pos + base_bits; cout->last_type = BT_FIXED; cout->block_to_finish = 1; cout->last_marked = 1; cout->cur_eob = 7u << 16;
  }
}

// --------------------------------------------------------------------------------------------
// The block chooser: Send_as_block's decision (:1222-1268), Mark_new_block (:999-1007), Expand_LZ_buffer's split of
// oversized stored blocks (:1024-1038) and the stream epilogue (:1613-1635), WITHOUT the sequential walk over all blocks.
//
// What the reference carries from block to block is (last_block_type, curr_descr, block_to_finish, last_block_marked, bit
// position).  A block that is sent fixed, dynamic or stored leaves a state that depends on that block alone; only a
// RECYCLED block (no header, the codes in force go on) hands its predecessor's state through.  And of the five costs only
// "recycle" depends on the state at all (the constant c of :1198-1201 is added to the other four alike).  So:
//   k_ch_tentative   every block decides as if its predecessor had taken its own best non-recycled format (which is
//                    state-independent): right whenever the predecessor does not recycle.
//   k_ch_resolve     one wave visits only the blocks that recycle and those behind them (whose assumption was wrong),
//                    jumping from one tentative recycle to the next, 64 blocks per step.
//   k_ch_stored      the stored blocks count their pieces (:1024-1038 halves the ATOM range until a piece has < 64 KiB).
//   k_ch_layout      bit positions: a block maps the position p before it to p + a, a stored block to align8 (p + a) + b
//                    -- closed under composition, hence a parallel scan; tile and piece numbers; the state for the next
//                    range; the epilogue.
//   k_ch_stored_emit BFINAL / BTYPE / LEN / NLEN and the copy list of the stored pieces.
// --------------------------------------------------------------------------------------------
struct ChW {                      // what the chooser kernels pass on about a block
  uint64_t rdata;                 // LZ data bits under the codes in force (when the block recycles them)
  uint32_t eob_in;                // (length << 16) | code of symbol 256 under the codes in force before the block
  uint8_t dec, T_in, B_in, pad;   // decision FMT_*; last_block_type and block_to_finish before the block
  int32_t code_block;             // codes in force before the block (for a recycled block: the table it is coded with)
  uint32_t code_variant;
  uint32_t npieces;               // stored: pieces of < 64 KiB
  uint32_t bytes_pad;
};
static_assert(sizeof(ChW) == 32, "ChW");

__device__ __forceinline__ uint32_t ch_nonrec(const ChRec &r) { return r.base3 <= r.stored ? r.fmt3 : (uint32_t)FMT_STORED; }   // (a <= st <=> base3 <= stored: same c)

// the decision of :1243-1268 given the costs; tie order fixed / dynamic / dynamic-RLE (inside base3), recycled, stored
__device__ __forceinline__ uint32_t ch_decide(const ChRec &r, uint64_t c, uint64_t rb) {
  const uint64_t INF = ~0ull;
  const uint64_t a = r.base3 + c, st = r.stored == INF ? INF : r.stored + c;
  if (a <= rb && a <= st) return r.fmt3;
  return rb <= st ? (uint32_t)FMT_RECYCLE : (uint32_t)FMT_STORED;
}

__global__ void __launch_bounds__(256) k_ch_tentative(uint32_t nblocks, const ChRec *__restrict__ chrec, ChW *__restrict__ chw, int fixed_only) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblocks) return;
  ChW w; w.rdata = 0; w.eob_in = 7u << 16; w.dec = 0xFF; w.T_in = BT_RESERVED; w.B_in = 1; w.pad = 0; w.code_block = CODE_FIXED; w.code_variant = 0; w.npieces = 0; w.bytes_pad = 0;
  if (fixed_only) {
    // Deflate_Fixed in a batch: every block is sent fixed; an entry's first block opens the one Deflate block, the others go on in it
    w.dec = FMT_FIXED;
    if (chrec[i].br.last_flush & BR_ENTRY_FIRST) w.B_in = 0; else w.T_in = BT_FIXED;
  } else if (chrec[i].br.last_flush & BR_ENTRY_FIRST) {
    // a batch: the block starts its entry, i.e. a stream (last_block_type reserved, nothing to finish, nothing to recycle)
    w.B_in = 0; w.rdata = ~0ull;
    w.dec = (uint8_t)ch_decide(chrec[i], 1, ~0ull);
  } else if (i > 0) {
    const ChRec r = chrec[i], pr = chrec[i - 1];
    const uint32_t pn = ch_nonrec(pr);
    const uint64_t INF = ~0ull;
    uint64_t rb = INF, c = 1;
    if (pn == FMT_FIXED) { w.T_in = BT_FIXED; w.eob_in = 7u << 16; w.code_block = CODE_FIXED; rb = r.bits[0]; c += 7; }
    else if (pn == FMT_STORED) { w.T_in = BT_STORED; }
    else {
      const int v = pn == FMT_DYN1 ? 1 : 2;
      w.T_in = BT_DYNAMIC; w.eob_in = pr.eob[v - 1]; w.code_block = (int32_t)(i - 1); w.code_variant = (uint32_t)v;
      if ((r.ok >> v) & 1u) rb = r.bits[v];
      c += (uint64_t)(w.eob_in >> 16);
    }
    w.dec = (uint8_t)ch_decide(r, c, rb);
    w.rdata = rb;
  }
  chw[i] = w;                      // (block 0 is decided by k_ch_resolve, from the state the range before left behind)
}

__global__ void __launch_bounds__(64) k_ch_resolve(uint32_t nblocks, const ChRec *__restrict__ chrec, const BlockInfo *__restrict__ binfo,
                                                   ChW *__restrict__ chw, const ChooserCarry *__restrict__ cin, int batch) {
  const int lane = threadIdx.x;
  if (nblocks == 0) return;
  // actual state (uniform over the wave)
  int T = cin->last_type, B = cin->block_to_finish;
  uint32_t eob = cin->cur_eob;
  int code_block = T == BT_DYNAMIC ? CODE_CARRIED : CODE_FIXED, code_variant = 0;
  const uint64_t INF = ~0ull;
  auto out_state = [&](uint32_t k) {                 // the state block k leaves behind when it does not recycle
    const ChRec r = chrec[k];
    const uint32_t f = ch_nonrec(r);
    B = 1;
    if (f == FMT_FIXED) { T = BT_FIXED; eob = 7u << 16; code_block = CODE_FIXED; code_variant = 0; }
    else if (f == FMT_STORED) { T = BT_STORED; }
    else { T = BT_DYNAMIC; code_variant = f == FMT_DYN1 ? 1 : 2; eob = r.eob[code_variant - 1]; code_block = (int)k; }
  };
  uint32_t i = 0;
  bool need_eval = !batch;                           // (a batch: block 0 starts an entry and has decided for itself)
  while (i < nblocks) {
    uint32_t dec;
    if (need_eval && batch && (chrec[i].br.last_flush & BR_ENTRY_FIRST)) need_eval = false;   // a new entry: nothing carries over
    if (need_eval) {
      // Send_as_block's decision for block i with the actual state
      const ChRec r = chrec[i];
      uint64_t rb = INF;
      if (T == BT_FIXED) rb = r.bits[0];
      else if (T == BT_DYNAMIC) {
        if (code_block == (int)i - 1) { if ((r.ok >> code_variant) & 1u) rb = r.bits[code_variant]; }
        else {
          // the codes in force are older than the previous block (a chain of recycled blocks, or carried over): evaluate here
          const BlockInfo *bi = &binfo[i];
          const uint8_t *cl = code_block == CODE_CARRIED ? cin->bl : (code_variant == 1 ? binfo[code_block].bl1 : binfo[code_block].bl2);
          bool bad = false; uint64_t rc = 0;
          for (int q = 0; q < 5; q++) {
            const int s = lane + 64 * q;
            const int cu = cl[s]; const uint32_t stv = bi->stats[s];
            if (cu == 0 && bi->bl1[s] > 0) bad = true;                           // Recyclable :495-508
            if (s < 288) { if (s != 256 && s <= 285) rc += (uint64_t)stv * (uint64_t)(cu + litlen_sym_extra(s)); }
            else if (s - 288 <= 29) rc += (uint64_t)stv * (uint64_t)(cu + dist_sym_extra(s - 288));
          }
          rc = wave_sum_u64(rc);
          if (!__any(bad)) rb = rc;
        }
      }
      const bool finishing = B && (T == BT_FIXED || T == BT_DYNAMIC);
      const uint64_t c = 1 + (finishing ? (uint64_t)(eob >> 16) : 0);               // :1198-1201
      dec = ch_decide(r, c, rb);
      if (lane == 0) {
        ChW w; w.rdata = rb; w.eob_in = eob; w.dec = (uint8_t)dec; w.T_in = (uint8_t)T; w.B_in = (uint8_t)B; w.pad = 0;
        w.code_block = code_block; w.code_variant = (uint32_t)code_variant; w.npieces = 0; w.bytes_pad = 0;
        chw[i] = w;
      }
    } else dec = chw[i].dec;
    if (dec != FMT_RECYCLE) {
      // from here the tentative decisions hold up to and including the next block that recycles
      uint32_t j = nblocks;
      for (uint32_t b0 = i + 1; b0 < nblocks && j == nblocks; b0 += 64) {
        const uint32_t k = b0 + (uint32_t)lane;
        const bool rcy = k < nblocks && chw[k].dec == FMT_RECYCLE;
        const unsigned long long m = __ballot(rcy);
        if (m) j = b0 + (uint32_t)(__ffsll((long long)m) - 1);
      }
      if (j == nblocks) break;
      out_state(j - 1);                              // (j - 1 does not recycle: its own best format, as block j assumed)
      i = j; need_eval = false;
    } else {
      i++; need_eval = true;                         // the state goes through a recycled block unchanged
    }
  }
}

// pieces of a stored block: Expand_LZ_buffer halves the ATOM range while it holds more than 65 535 bytes (:1024-1038)
template <typename F>
__device__ __forceinline__ uint32_t ch_stored_walk(const uint32_t *__restrict__ apos, uint32_t first, uint32_t count, uint32_t end_byte, int last_block, F &&piece) {
  uint32_t stk_first[40], stk_last[40]; int stk_lb[40]; int sp = 0;
  uint32_t np = 0;
  stk_first[0] = first; stk_last[0] = first + count - 1; stk_lb[0] = last_block; sp = 1;
  while (sp > 0) {
    sp--;
    const uint32_t f = stk_first[sp], l = stk_last[sp]; const int lb = stk_lb[sp];
    // (end_byte: the block ends its entry of a batch, whose last atom has no successor in the position array)
    const uint32_t src = apos[f], nbytes = ((end_byte && l + 1 == first + count) ? end_byte : apos[l + 1]) - src;
    if (nbytes > 0xFFFF) {
      const uint32_t mid = (uint32_t)(((uint64_t)f + (uint64_t)l) / 2);
      stk_first[sp] = mid + 1; stk_last[sp] = l; stk_lb[sp] = lb; sp++;          // the second half comes after the first: pushed first
      stk_first[sp] = f; stk_last[sp] = mid; stk_lb[sp] = 0; sp++;
      continue;
    }
    piece(np, src, nbytes, lb);
    np++;
  }
  return np;
}

__global__ void __launch_bounds__(64) k_ch_stored(uint32_t nblocks, const ChRec *__restrict__ chrec, const uint32_t *__restrict__ apos, ChW *__restrict__ chw) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblocks || chw[i].dec != FMT_STORED) return;
  const BlockRange br = chrec[i].br;
  chw[i].npieces = ch_stored_walk(apos, br.first, br.count, br.pad, (int)(br.last_flush & BR_LAST_FLUSH), [](uint32_t, uint32_t, uint32_t, int) {});
}

// position transform of a block: kind 0: p -> p + a; kind 1: p -> align8 (p + a) + b; kind 2: p -> b (the first block of an
// entry of a batch: positions restart).  Closed under composition.
struct PosFn { uint64_t a, b; uint32_t kind; };
__device__ __forceinline__ uint64_t posfn_apply(const PosFn &f, uint64_t p) { return f.kind == 2 ? f.b : f.kind ? ((p + f.a + 7) & ~7ull) + f.b : p + f.a; }
__device__ __forceinline__ PosFn posfn_compose(const PosFn &f, const PosFn &g) {   // first f, then g
  PosFn h;
  if (g.kind == 2) return g;
  if (f.kind == 2) { h.kind = 2; h.a = 0; h.b = posfn_apply(g, f.b); return h; }
  if (g.kind == 0) { h.kind = f.kind; h.a = f.kind ? f.a : f.a + g.a; h.b = f.kind ? f.b + g.a : 0; }
  else if (f.kind == 0) { h.kind = 1; h.a = f.a + g.a; h.b = g.b; }
  else { h.kind = 1; h.a = f.a; h.b = ((f.b + g.a + 7) & ~7ull) + g.b; }
  return h;
}

__global__ void __launch_bounds__(1024) k_ch_layout(uint32_t nblocks, const ChRec *__restrict__ chrec, const BlockInfo *__restrict__ binfo, const ChW *__restrict__ chw,
                                                    EmitRec *__restrict__ emit, uint32_t *__restrict__ piece_base, uint32_t cap_tiles, uint32_t cap_pieces,
                                                    uint32_t *__restrict__ out32, uint64_t lim_bits, ChooserOut *__restrict__ res,
                                                    const ChooserCarry *__restrict__ cin, ChooserCarry *__restrict__ cout, uint64_t base_bits, int do_epilogue,
                                                    const uint32_t *__restrict__ blk_entry, EntOut *__restrict__ ent_out /* batches; else null */) {
  __shared__ PosFn wfn[16];
  __shared__ uint32_t wtiles[16], wpieces[16], wlast[16];
  __shared__ uint64_t s_pos; __shared__ uint32_t s_tiles, s_pieces, s_last;      // carried from tile to tile
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) { s_pos = cin->pos - base_bits; s_tiles = 0; s_pieces = 0; s_last = 0xFFFFFFFFu; }
  __syncthreads();
  for (uint32_t t0 = 0; t0 < nblocks; t0 += 1024) {
    const uint32_t i = t0 + (uint32_t)tid;
    const bool ex = i < nblocks;
    PosFn f; f.kind = 0; f.a = 0; f.b = 0;
    uint32_t nt = 0, np = 0, opener = 0xFFFFFFFFu;
    ChRec r; ChW cw;
    uint64_t pre_bits = 0, hdr = 0, data = 0, opt = 0;
    bool opens = false;
    if (ex) {
      r = chrec[i]; cw = chw[i];
      const bool fin = cw.B_in && (cw.T_in == BT_FIXED || cw.T_in == BT_DYNAMIC);
      const uint64_t eobl = fin ? (uint64_t)(cw.eob_in >> 16) : 0;
      const uint64_t c = 1 + eobl;
      if (cw.dec == FMT_STORED) {
        np = cw.npieces;
        // piece 1: [end-of-block code] BFINAL, BTYPE, pad, LEN, NLEN, bytes; every further piece starts byte aligned: 8 + 32 bits + bytes
        f.kind = 1; f.a = eobl + 3; f.b = 32ull * np + 8ull * (np - 1) + 8ull * binfo[i].bytes;
        opt = r.stored + c; opener = i;
      } else {
        nt = (r.br.count + TILE - 1) / TILE;
        if (cw.dec == FMT_RECYCLE) { data = cw.rdata; opt = cw.rdata; }
        else if (cw.dec == FMT_FIXED) { opens = cw.T_in != BT_FIXED; data = r.fixed_data; opt = r.base3 + c; }      // Send_fixed_block :1108-1121
        else { opens = true; hdr = cw.dec == FMT_DYN1 ? r.hdr1 : r.hdr2; data = cw.dec == FMT_DYN1 ? r.dyn1_data : r.dyn2_data; opt = r.base3 + c; }
        if (opens) { pre_bits = eobl + 3; opener = i; }
        f.a = pre_bits + hdr + data;
      }
    }
    // a batch: the first block of an entry starts at the entry's bit 0 whatever came before
    const bool efirst = ex && ent_out && (r.br.last_flush & BR_ENTRY_FIRST), elast = ex && ent_out && (r.br.last_flush & BR_ENTRY_LAST);
    const PosFn fown = f;
    if (efirst) { f.kind = 2; f.a = 0; f.b = posfn_apply(fown, 0); }
    // inclusive scans over the tile: position transforms (composition), tiles, pieces, last block that opened a Deflate block
    PosFn inc = f; uint32_t it = nt, ip = np, il = opener;
    for (int off = 1; off < 64; off <<= 1) {
      PosFn o; o.a = __shfl_up(inc.a, off); o.b = __shfl_up(inc.b, off); o.kind = __shfl_up(inc.kind, off);
      const uint32_t ot = __shfl_up(it, off), op = __shfl_up(ip, off), ol = __shfl_up(il, off);
      if (lane >= off) { inc = posfn_compose(o, inc); it += ot; ip += op; il = il == 0xFFFFFFFFu ? ol : il; }
    }
    if (lane == 63) { wfn[w] = inc; wtiles[w] = it; wpieces[w] = ip; }
    {
      // last opener of the wave: highest lane with one
      const unsigned long long m = __ballot(opener != 0xFFFFFFFFu);
      if (lane == 0) wlast[w] = m ? t0 + (uint32_t)(w * 64 + 63 - __builtin_clzll(m)) : 0xFFFFFFFFu;
    }
    __syncthreads();
    PosFn before; before.kind = 0; before.a = 0; before.b = 0;
    uint32_t tb = 0, pb = 0;
    for (int k = 0; k < w; k++) { before = posfn_compose(before, wfn[k]); tb += wtiles[k]; pb += wpieces[k]; }
    // exclusive within the wave: shift the inclusive values by one lane
    PosFn exl; exl.a = __shfl_up(inc.a, 1); exl.b = __shfl_up(inc.b, 1); exl.kind = __shfl_up(inc.kind, 1);
    uint32_t ext = __shfl_up(it, 1), exp_ = __shfl_up(ip, 1);
    if (lane == 0) { exl.kind = 0; exl.a = 0; exl.b = 0; ext = 0; exp_ = 0; }
    const PosFn upto = posfn_compose(before, exl);
    const uint64_t p0 = efirst ? 0 : posfn_apply(upto, s_pos);      // bit position before this block
    const uint32_t tile0 = s_tiles + tb + ext, piece0 = s_pieces + pb + exp_;
    if (ex) {
      EmitRec e; e.hdr_bitpos = 0; e.data_bitpos = 0; e.cost_bits = opt; e.fmt = cw.dec; e.code_block = CODE_FIXED; e.code_variant = 0; e.tile_base = tile0;
      e.pre_pos = 0; e.pre_eob = 0; e.pre_flags = 0;
      if (cw.dec == FMT_STORED) {
        e.pre_pos = p0;                                              // (k_ch_stored_emit starts here)
        e.pre_eob = (cw.B_in && (cw.T_in == BT_FIXED || cw.T_in == BT_DYNAMIC)) ? cw.eob_in : 0u;
        piece_base[i] = piece0;
      } else {
        if (opens) {
          e.pre_pos = p0;
          e.pre_eob = (cw.B_in && (cw.T_in == BT_FIXED || cw.T_in == BT_DYNAMIC)) ? cw.eob_in : 0u;
          e.pre_flags = 1u | ((r.br.last_flush & BR_LAST_FLUSH) << 1) | ((cw.dec == FMT_FIXED ? 1u : 2u) << 2);
        }
        e.hdr_bitpos = p0 + pre_bits;
        e.data_bitpos = p0 + pre_bits + hdr;
        if (cw.dec == FMT_RECYCLE || (cw.dec == FMT_FIXED && !opens)) { e.code_block = cw.code_block; e.code_variant = cw.code_variant; }
        else if (cw.dec == FMT_FIXED) { e.code_block = CODE_FIXED; e.code_variant = 0; }
        else { e.code_block = (int32_t)i; e.code_variant = cw.dec == FMT_DYN1 ? 1u : 2u; }
      }
      emit[i] = e;
      if (elast) {
        // the entry ends behind this block: what its epilogue (Encode :1613-1635) has to write follows from the last block
        // of the entry that opened a Deflate block (the entry's first block always does)
        uint32_t lo = il;
        if (lo == 0xFFFFFFFFu) { for (int k = w - 1; k >= 0; k--) if (wlast[k] != 0xFFFFFFFFu) { lo = wlast[k]; break; } }
        if (lo == 0xFFFFFFFFu) lo = s_last;
        const uint32_t dk = chw[lo].dec;
        const ChRec rk = chrec[lo];
        EntOut eo; eo.bits = posfn_apply(fown, p0);
        eo.eob = dk == FMT_STORED ? 0u : dk == FMT_FIXED ? (7u << 16) : rk.eob[dk == FMT_DYN1 ? 0 : 1];
        eo.fake = (rk.br.last_flush & BR_LAST_FLUSH) ? 0u : 1u;
        ent_out[blk_entry[i]] = eo;
      }
    }
    __syncthreads();
    if (tid == 1023) {                                               // totals of the tile -> carried state
      const PosFn all = posfn_compose(before, inc);
      s_pos = posfn_apply(all, s_pos); s_tiles += tb + it; s_pieces += pb + ip;
    }
    if (tid == 0) { for (int k = 15; k >= 0; k--) if (wlast[k] != 0xFFFFFFFFu) { s_last = wlast[k]; break; } }
    __syncthreads();
  }
  // the state after the range (uniform values: every lane computes them)
  uint64_t pos = s_pos;
  int T = cin->last_type, B = cin->block_to_finish, M = cin->last_marked;
  uint32_t eob = cin->cur_eob;
  const uint8_t *cl = T == BT_DYNAMIC ? cin->bl : nullptr;
  if (s_last != 0xFFFFFFFFu) {
    const uint32_t k = s_last;
    const uint32_t dec = chw[k].dec;
    const ChRec r = chrec[k];
    B = 1; M = (int)(r.br.last_flush & BR_LAST_FLUSH); cl = nullptr;
    if (dec == FMT_STORED) T = BT_STORED;
    else if (dec == FMT_FIXED) { T = BT_FIXED; eob = 7u << 16; }
    else { T = BT_DYNAMIC; eob = r.eob[dec == FMT_DYN1 ? 0 : 1]; cl = dec == FMT_DYN1 ? binfo[k].bl1 : binfo[k].bl2; }
    if (dec == FMT_STORED) eob = chw[k].eob_in;                     // (cur_eob is not touched by a stored block)
  }
  // stream epilogue, Encode :1613-1635 (by the range that owns the stream's last flush)
  if (do_epilogue) {
    if (B && (T == BT_FIXED || T == BT_DYNAMIC)) {
      if (tid == 0) put_bits_lim(out32, lim_bits, pos, eob & 0xFFFF, (int)(eob >> 16));
      pos += (uint64_t)(eob >> 16);
    }
    if (!M) {
      if (tid == 0) { put_bits_lim(out32, lim_bits, pos, 1, 1); put_bits_lim(out32, lim_bits, pos + 1, 1, 2); }
      pos += 3 + 7;                                                            // fake final fixed block: EOB = 0000000
    }
  }
  if (tid < 320) cout->bl[tid] = cl ? cl[tid] : (uint8_t)0;
  if (tid == 0) {
    res->total_bits = pos; res->n_tiles = s_tiles; res->n_pieces = s_pieces; res->n_blocks = nblocks;
    res->overflow = (s_tiles > cap_tiles || s_pieces > cap_pieces) ? 1u : 0u;
    cout->pos = pos + base_bits; cout->last_type = T; cout->block_to_finish = B; cout->last_marked = M; cout->cur_eob = eob; cout->pad[0] = cout->pad[1] = 0;
  }
}

// ---- batches of entries: what is per entry -------------------------------------------------------------------------
// Flush table of a batch: entry e has atoms [offsets[chunk0[e]], offsets[chunk0[e + 1]]) (exclusive scan of the parse chunks'
// token counts) and was given the flush slots [fl0[e], fl0[e + 1]) from its byte length.
__global__ void k_batch_geom(uint32_t E, const uint32_t *__restrict__ chunk0, const uint32_t *__restrict__ fl0, const uint32_t *__restrict__ ent_start,
                             const uint32_t *__restrict__ ent_len, const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ total_atoms,
                             FlushGeom *__restrict__ ftab, EntOut *__restrict__ ent_out) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const uint32_t a0 = offsets[chunk0[e]], a1 = e + 1 < E ? offsets[chunk0[e + 1]] : *total_atoms, T = a1 - a0;
  const uint32_t nfl = (T + FLUSH - 1) / FLUSH;
  for (uint32_t s = fl0[e], k = 0; s < fl0[e + 1]; s++, k++) {
    FlushGeom g; g.F = a0 + k * FLUSH; g.to = 0; g.gj = k; g.flags = FG_EMPTY; g.end_byte = ent_start[e] + ent_len[e]; g.entry = e; g.pad[0] = g.pad[1] = 0;
    if (k < nfl) {
      g.to = g.F + FLUSH - 1 < a1 - 1 ? g.F + FLUSH - 1 : a1 - 1;
      g.flags = (k == 0 ? (uint32_t)FG_ENTRY_FIRST : 0u) | (k + 1 == nfl ? (uint32_t)FG_ENTRY_LAST : 0u) |
                ((k + 1 == nfl && (T % FLUSH) != 0) ? (uint32_t)FG_LAST_PARTIAL : 0u);
    }
    ftab[s] = g;
  }
  EntOut eo; eo.bits = 0; eo.eob = 0; eo.fake = 1;     // an entry without atoms: the fake final block alone (empty input: 03 00)
  ent_out[e] = eo;
}

// bytes of every entry's stream (epilogue included)
__global__ void k_batch_sizes(uint32_t E, const EntOut *__restrict__ ent_out, uint32_t *__restrict__ ent_bytes) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const EntOut eo = ent_out[e];
  ent_bytes[e] = (uint32_t)((eo.bits + (eo.eob >> 16) + (eo.fake ? 10u : 0u) + 7) / 8);
}

// block positions were laid out from the entry's bit 0: move them to the entry's place in the output
__global__ void k_batch_rebase(uint32_t nblocks, const uint32_t *__restrict__ blk_entry, const uint32_t *__restrict__ ent_base, EmitRec *__restrict__ emit) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblocks) return;
  const uint64_t b = 8ull * ent_base[blk_entry[i]];
  emit[i].hdr_bitpos += b; emit[i].data_bitpos += b; emit[i].pre_pos += b;
}

// the epilogue of every entry (Encode :1613-1635): end-of-block code of the block being finished, fake final fixed block
__global__ void k_batch_finish(uint32_t E, const EntOut *__restrict__ ent_out, const uint32_t *__restrict__ ent_base, uint32_t *__restrict__ out32, uint64_t lim_bits) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const EntOut eo = ent_out[e];
  uint64_t pos = 8ull * ent_base[e] + eo.bits;
  if (eo.eob) { put_bits_lim(out32, lim_bits, pos, eo.eob & 0xFFFF, (int)(eo.eob >> 16)); pos += eo.eob >> 16; }
  if (eo.fake) { put_bits_lim(out32, lim_bits, pos, 1, 1); put_bits_lim(out32, lim_bits, pos + 1, 1, 2); }
}

// the stored blocks' headers and copy list, one lane per stored block (Mark_new_block + :1040-1051 for every piece)
__global__ void __launch_bounds__(64) k_ch_stored_emit(uint32_t nblocks, const ChRec *__restrict__ chrec, const ChW *__restrict__ chw, const EmitRec *__restrict__ emit,
                                                       const uint32_t *__restrict__ piece_base, const uint32_t *__restrict__ apos,
                                                       StoredPiece *__restrict__ pieces, uint32_t *__restrict__ out32, uint64_t lim_bits) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblocks || chw[i].dec != FMT_STORED) return;
  const BlockRange br = chrec[i].br;
  const EmitRec e = emit[i];
  uint64_t pos = e.pre_pos;
  const uint32_t pb = piece_base[i];
  bool first = true;
  ch_stored_walk(apos, br.first, br.count, br.pad, (int)(br.last_flush & BR_LAST_FLUSH), [&](uint32_t k, uint32_t src, uint32_t nbytes, int lb) {
    if (first && e.pre_eob) { put_bits_lim(out32, lim_bits, pos, e.pre_eob & 0xFFFF, (int)(e.pre_eob >> 16)); pos += e.pre_eob >> 16; }
    first = false;
    put_bits_lim(out32, lim_bits, pos, (uint32_t)lb, 1);                       // Mark_new_block: BFINAL; then BTYPE 00
    pos += 3;
    pos = (pos + 7) & ~7ull;                                                   // Flush_bit_buffer
    put_bits_lim(out32, lim_bits, pos, nbytes & 0xFFFF, 16);
    put_bits_lim(out32, lim_bits, pos + 16, (~nbytes) & 0xFFFF, 16);
    StoredPiece pc; pc.dst_byte = (pos >> 3) + 4; pc.src_byte = src; pc.nbytes = nbytes;
    pieces[pb + k] = pc;
    pos += 32 + 8ull * nbytes;
  });
}

// What k_choose_lean left out, one thread per block: the bits in front of a block that opens a new Deflate block (end-of-
// block code of the one before, BFINAL, BTYPE) and the block's entries of the tile table.
__global__ void __launch_bounds__(256) k_emit_prefix(uint32_t nblocks, const EmitRec *__restrict__ emit, const BlockRange *__restrict__ blocks,
                                                     uint32_t *__restrict__ tile_block, uint32_t *__restrict__ out32) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblocks) return;
  const EmitRec e = emit[i];
  if (e.pre_flags & 1u) {
    uint64_t p = e.pre_pos;
    if (e.pre_eob) { put_bits_global(out32, p, e.pre_eob & 0xFFFF, (int)(e.pre_eob >> 16)); p += e.pre_eob >> 16; }
    put_bits_global(out32, p, (e.pre_flags >> 1) & 1u, 1);
    put_bits_global(out32, p + 1, (e.pre_flags >> 2) & 3u, 2);
  }
  if (e.fmt != FMT_STORED) {
    const uint32_t nt = (blocks[i].count + TILE - 1) / TILE;
    for (uint32_t t = 0; t < nt; t++) tile_block[e.tile_base + t] = i;
  }
}

// --------------------------------------------------------------------------------------------
// code tables: codes[b*320 + s] = (len << 16) | bit-reversed canonical code
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_block_codes(uint32_t nblocks, const EmitRec *__restrict__ emit, const BlockInfo *__restrict__ binfo,
                                                    uint32_t *__restrict__ codes, const ChooserCarry *__restrict__ cin) {
  __shared__ uint8_t bl[320];
  __shared__ uint16_t cd[320];
  const uint32_t b = blockIdx.x;
  const int lane = threadIdx.x;
  if (b == nblocks) {                                                            // fixed table
    for (int i = lane; i < 320; i += 64) bl[i] = (uint8_t)(i < 288 ? fixed_litlen_bl(i) : 5);
  } else if (b == nblocks + 1) {                                                 // the table in force when the range began
    for (int i = lane; i < 320; i += 64) bl[i] = cin->bl[i];
  } else {
    const uint32_t f = emit[b].fmt;
    if (f != FMT_DYN1 && f != FMT_DYN2) return;
    const uint8_t *src = f == FMT_DYN1 ? binfo[b].bl1 : binfo[b].bl2;
    for (int i = lane; i < 320; i += 64) bl[i] = src[i];
  }
  __syncthreads();
  if (lane == 0) canonical_codes(bl, 288, cd);
  if (lane == 1) canonical_codes(bl + 288, 32, cd + 288);
  __syncthreads();
  for (int i = lane; i < 320; i += 64) codes[(uint64_t)b * 320 + i] = ((uint32_t)bl[i] << 16) | cd[i];
}

__device__ __forceinline__ uint32_t code_table_index(const EmitRec &e, uint32_t nblocks) {
  return e.code_block == CODE_FIXED ? nblocks : (e.code_block == CODE_CARRIED ? nblocks + 1 : (uint32_t)e.code_block);
}

// bits of one atom under a code table staged in LDS; value returned in v (LSB first)
__device__ __forceinline__ int atom_bits(uint32_t t, const uint32_t *tab, uint64_t &v) {
  if (!tok_is_match(t)) { uint32_t c = tab[t & 0xFF]; v = c & 0xFFFF; return (int)(c >> 16); }
  const int len = (int)((t >> 16) & 0x1FF), dist = (int)(t & 0xFFFF);
  uint32_t lc = tab[len_symbol(len)], dc = tab[288 + dist_symbol(dist)];
  int n = (int)(lc >> 16);
  v = lc & 0xFFFF;
  int le = len_extra_bits(len);
  v |= (uint64_t)len_extra_val(len) << n; n += le;
  v |= (uint64_t)(dc & 0xFFFF) << n; n += (int)(dc >> 16);
  int de = dist_extra_bits(dist);
  v |= (uint64_t)dist_extra_val(dist) << n; n += de;
  return n;
}

__global__ void __launch_bounds__(256) k_tile_bits(uint32_t nblocks, const uint32_t *__restrict__ atoms, const BlockRange *__restrict__ blocks,
                                                   const EmitRec *__restrict__ emit, const uint32_t *__restrict__ tile_block,
                                                   const uint32_t *__restrict__ codes, uint32_t *__restrict__ tile_bits) {
  __shared__ uint32_t tab[320];
  __shared__ uint32_t red[4];
  const uint32_t t = blockIdx.x, b = tile_block[t];
  const EmitRec e = emit[b];
  const BlockRange br = blocks[b];
  const uint32_t ti = t - e.tile_base;
  const uint32_t a0 = br.first + ti * TILE, a1 = (ti * TILE + TILE < br.count) ? a0 + TILE : br.first + br.count;
  const uint32_t *src = codes + (uint64_t)code_table_index(e, nblocks) * 320;
  const int tid = threadIdx.x;
  for (int i = tid; i < 320; i += 256) tab[i] = src[i];
  __syncthreads();
  uint32_t s = 0;
  for (uint32_t a = a0 + tid; a < a1; a += 256) { uint64_t v; s += (uint32_t)atom_bits(atoms[a], tab, v); }
  s = wave_sum_u32(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) tile_bits[t] = red[0] + red[1] + red[2] + red[3];
}

// one thread per block: running bit position of its tiles
__global__ void k_tile_scan(uint32_t nblocks, const BlockRange *__restrict__ blocks, const EmitRec *__restrict__ emit,
                            const uint32_t *__restrict__ tile_bits, uint64_t *__restrict__ tile_bitpos) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  const EmitRec e = emit[b];
  if (e.fmt == FMT_STORED) return;
  uint32_t nt = (blocks[b].count + TILE - 1) / TILE;
  uint64_t pos = e.data_bitpos;
  for (uint32_t t = 0; t < nt; t++) { tile_bitpos[e.tile_base + t] = pos; pos += tile_bits[e.tile_base + t]; }
}

// emit one tile: 256 threads x 8 consecutive atoms
__global__ void __launch_bounds__(256) k_emit_tiles(uint32_t nblocks, const uint32_t *__restrict__ atoms, const BlockRange *__restrict__ blocks,
                                                    const EmitRec *__restrict__ emit, const uint32_t *__restrict__ tile_block,
                                                    const uint32_t *__restrict__ codes, const uint64_t *__restrict__ tile_bitpos,
                                                    uint32_t *__restrict__ out32) {
  __shared__ uint32_t tab[320];
  __shared__ uint32_t buf[TILE * 48 / 32 + 4];
  __shared__ uint32_t wsum[4];
  const uint32_t t = blockIdx.x, b = tile_block[t];
  const EmitRec e = emit[b];
  const BlockRange br = blocks[b];
  const uint32_t ti = t - e.tile_base;
  const uint32_t a0 = br.first + ti * TILE, a1 = (ti * TILE + TILE < br.count) ? a0 + TILE : br.first + br.count;
  const uint32_t *src = codes + (uint64_t)code_table_index(e, nblocks) * 320;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 320; i += 256) tab[i] = src[i];
  for (int i = tid; i < (int)(TILE * 48 / 32 + 4); i += 256) buf[i] = 0;
  __syncthreads();
  uint64_t v[8]; int nb[8]; uint32_t tot = 0;
  for (int k = 0; k < 8; k++) {
    uint32_t a = a0 + tid * 8 + k;
    if (a < a1) { nb[k] = atom_bits(atoms[a], tab, v[k]); tot += (uint32_t)nb[k]; } else { nb[k] = 0; v[k] = 0; }
  }
  uint32_t incl = tot;
  for (int off = 1; off < 64; off <<= 1) { uint32_t x = __shfl_up(incl, off); if (lane >= off) incl += x; }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int k = 0; k < w; k++) base += wsum[k];
  const uint64_t b0 = tile_bitpos[t];
  uint32_t bp = (uint32_t)(b0 & 31) + base + incl - tot;
  for (int k = 0; k < 8; k++) {
    if (nb[k]) {
      uint32_t wd = bp >> 5; int sh = (int)(bp & 31);
      uint64_t lo = v[k] << sh;
      atomicOr(&buf[wd], (uint32_t)lo);
      if (sh + nb[k] > 32) atomicOr(&buf[wd + 1], (uint32_t)(lo >> 32));
      if (sh + nb[k] > 64) atomicOr(&buf[wd + 2], (uint32_t)(v[k] >> (64 - sh)));
      bp += (uint32_t)nb[k];
    }
  }
  __syncthreads();
  const uint32_t tile_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  const uint32_t nwords = ((uint32_t)(b0 & 31) + tile_total + 31) >> 5;
  uint32_t *dst = out32 + (b0 >> 5);
  for (uint32_t i = tid; i < nwords; i += 256) {
    uint32_t x = buf[i];
    if (i == 0 || i + 1 == nwords) { if (x) atomicOr(&dst[i], x); }
    else dst[i] = x;
  }
}

// dynamic block headers, Put_Compression_Structure :686-703 (effective mode); lane 0 of one wave per block
__global__ void __launch_bounds__(64) k_emit_headers(uint32_t nblocks, const EmitRec *__restrict__ emit, const BlockInfo *__restrict__ binfo,
                                                     uint32_t *__restrict__ out32) {
  __shared__ uint8_t bl[320], cs[320], tbl[20];
  __shared__ uint16_t tcode[19];
  const uint32_t b = blockIdx.x;
  const EmitRec e = emit[b];
  if (e.fmt != FMT_DYN1 && e.fmt != FMT_DYN2) return;
  const int lane = threadIdx.x;
  const BlockInfo *bi = &binfo[b];
  const uint8_t *src = e.fmt == FMT_DYN1 ? bi->bl1 : bi->bl2;
  const uint8_t *tsrc = e.fmt == FMT_DYN1 ? bi->truc1 : bi->truc2;
  for (int i = lane; i < 320; i += 64) bl[i] = src[i];
  if (lane < 20) tbl[lane] = tsrc[lane];
  __syncthreads();
  if (lane != 0) return;
  int max_ll = 0, max_d = 0, n = 0;
  for (int a = 287; a >= 0; a--) if (bl[a] > 0) { max_ll = a; break; }
  for (int a = 31; a >= 0; a--) if (bl[288 + a] > 0) { max_d = a; break; }
  for (int a = 0; a <= max_ll; a++) cs[n++] = bl[a];
  for (int a = 0; a <= max_d; a++) cs[n++] = bl[288 + a];
  canonical_codes(tbl, 19, tcode);
  const int anz = tbl[19];
  uint64_t pos = e.hdr_bitpos;
  put_bits_global(out32, pos, (uint32_t)(max_ll - 256), 5); pos += 5;
  put_bits_global(out32, pos, (uint32_t)max_d, 5); pos += 5;
  put_bits_global(out32, pos, (uint32_t)(anz - 3), 4); pos += 4;
  for (int a = 0; a <= anz; a++) { put_bits_global(out32, pos, tbl[header_perm(a)], 3); pos += 3; }
  header_rle_walk(cs, n, [&](int x, uint32_t extra) {
    put_bits_global(out32, pos, tcode[x], tbl[x]); pos += tbl[x];
    int eb = header_extra_bits(x);
    if (eb) { put_bits_global(out32, pos, extra, eb); pos += eb; }
  });
}

// raw bytes of stored blocks (runs AFTER every atomic bit writer: plain byte stores)
__global__ void __launch_bounds__(256) k_copy_pieces(uint32_t npieces, const StoredPiece *__restrict__ pieces,
                                                     const uint8_t *__restrict__ in, uint8_t *__restrict__ out) {
  const uint32_t p = blockIdx.x;
  if (p >= npieces) return;
  const StoredPiece pc = pieces[p];
  for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < pc.nbytes; i += gridDim.y * 256) out[pc.dst_byte + i] = in[(uint64_t)pc.src_byte + i];
}

__global__ void k_set_u32(uint32_t *p, uint32_t v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = v; }

// --------------------------------------------------------------------------------------------
// host side: the entropy stage of the range in flight (c->rg), in three steps so that ranges on different GPUs can
// exchange what they need in between (zada_range_*):
//   entropy_analyze   everything that does not depend on the blocks before the range: descriptors, cuts, block analysis
//   entropy_choose    the sequential walk, from the state the range before left behind (carry_in) to carry_out
//   entropy_emit      code tables, bit emission, stored bytes
// --------------------------------------------------------------------------------------------
static EntropyView range_view(Ctx *c) {
  const Range &R = c->rg;
  Workspace &W = c->ws;
  EntropyView v;
  v.ftab = R.batch ? W.ftab : nullptr;
  v.atoms = W.ea_atoms + (LB_CAP - R.n_lb); v.apos = W.ea_apos + (LB_CAP - R.n_lb);
  const uint64_t tv = R.T_view == ~0ull ? R.T : R.T_view;
  v.foff = R.foff; v.nflush = R.nflush; v.lvalid = (uint32_t)(R.n_lb + tv + R.n_la);
  v.stream_final = (R.G + tv + R.n_la == R.T_total) ? 1u : 0u;
  v.j0 = R.j0;
  return v;
}

int batch_geometry(Ctx *c, uint32_t E, const uint32_t *d_total_atoms) {
  Workspace &W = c->ws;
  hipLaunchKernelGGL(k_batch_geom, dim3((E + 255) / 256), dim3(256), 0, c->stream, E, W.ent_chunk0, W.ent_fl0, W.ent_start, W.ent_len, W.offsets,
                     d_total_atoms, W.ftab, W.ent_out);
  return hip_check(c, hipGetLastError(), "batch_geometry");
}

int entropy_analyze(Ctx *c) {
  hipStream_t st = c->stream;
  Workspace &W = c->ws;
  Range &R = c->rg;
  const bool fixed_only = (R.method == 6);
  const EntropyView v = range_view(c);
  uint32_t nblocks = 0;
  hipLaunchKernelGGL(k_apos_sentinel, dim3(1), dim3(1), 0, st, v.atoms, (uint32_t *)v.apos, v.lvalid);
  if (v.nflush > 0) {
    if (fixed_only && !R.batch) {
      nblocks = v.nflush;
      hipLaunchKernelGGL(k_fill_blocks_fixed, dim3((nblocks + 255) / 256), dim3(256), 0, st, v.foff, v.lvalid - v.foff, nblocks, W.blocks);
    } else {
      const uint32_t kstep = fixed_only ? (1u << 30) : R.method == 8 ? 8 : R.method == 9 ? 4 : 1;   // max_choice :1310-1311 (Deflate_Fixed in a batch: no scanning, one block per flush)
      if (!fixed_only) hipLaunchKernelGGL(k_window_descr, dim3(v.nflush * WD_NGROUPS), dim3(64), 0, st, v, kstep, W.descr);
      c->tmark("window_descr");
      hipLaunchKernelGGL(k_cut_scan, dim3(v.nflush), dim3(64), 0, st, v, kstep, W.descr, W.seg_nblk, W.seg_cut, W.cut_trace);
      exclusive_scan_u32(st, W.seg_nblk, W.seg_blk_off, W.scan2, W.total2, v.nflush);
      hipMemcpyAsync(&nblocks, W.total2, 4, hipMemcpyDeviceToHost, st);
      hipLaunchKernelGGL(k_fill_blocks, dim3((v.nflush + 255) / 256), dim3(256), 0, st, v, W.seg_nblk, W.seg_cut, W.seg_blk_off, W.blocks, W.blk_entry, fixed_only ? 1 : 0);
      if (hip_check(c, hipStreamSynchronize(st), "cut_scan")) return ZADA_E_HIP_;
      c->tmark("cut_scan");
    }
    if (nblocks > W.cap_blocks) { c->err = "block table overflow"; return -1; }
    if (nblocks > 0) hipLaunchKernelGGL(k_block_analyze, dim3(nblocks), dim3(256), 0, st, v.atoms, v.apos, W.blocks, W.binfo);
    c->tmark("block_analyze");
    if ((!fixed_only || R.batch) && nblocks > 0) hipLaunchKernelGGL(k_block_relate, dim3(nblocks), dim3(64), 0, st, nblocks, W.binfo, W.blocks, (ChRec *)W.chrec);
  }
  R.nblocks = nblocks;
  R.analyzed = true;
  return hip_check(c, hipGetLastError(), "entropy_analyze");
}

int entropy_choose(Ctx *c) {
  hipStream_t st = c->stream;
  Workspace &W = c->ws;
  Range &R = c->rg;
  const bool fixed_only = (R.method == 6);
  const EntropyView v = range_view(c);
  if (R.batch) { R.carry_in = ChooserCarry(); R.carry_in.last_type = BT_RESERVED; R.carry_in.cur_eob = 7u << 16; }
  R.base_bits = R.carry_in.pos & ~7ull;
  hipMemcpyAsync(W.carry, &R.carry_in, sizeof(ChooserCarry), hipMemcpyHostToDevice, st);
  // the chooser itself writes the few bits of stored-block headers and of the epilogue: the output must be zero before
  const uint64_t lim_bits = W.cap_out * 8;
  if (fixed_only && !R.batch)
    hipLaunchKernelGGL(k_choose_fixed, dim3(1), dim3(64), 0, st, R.nblocks, W.blocks, W.binfo, W.emit, W.tile_block, (uint32_t)W.cap_tiles,
                       (uint32_t *)W.out, lim_bits, W.chooser, W.carry, W.carry + 1, R.base_bits, R.G == 0 ? 1 : 0, v.stream_final ? 1 : 0);
  else {
    const int do_epilogue = !R.batch && v.stream_final && (v.nflush > 0 || R.T_total == 0);
    const uint32_t nb = R.nblocks;
    ChW *chw = (ChW *)W.chw;
    if (nb > 0) {
      hipLaunchKernelGGL(k_ch_tentative, dim3((nb + 255) / 256), dim3(256), 0, st, nb, (const ChRec *)W.chrec, chw, fixed_only ? 1 : 0);
      if (!fixed_only) hipLaunchKernelGGL(k_ch_resolve, dim3(1), dim3(64), 0, st, nb, (const ChRec *)W.chrec, W.binfo, chw, W.carry, R.batch ? 1 : 0);
      hipLaunchKernelGGL(k_ch_stored, dim3((nb + 63) / 64), dim3(64), 0, st, nb, (const ChRec *)W.chrec, v.apos, chw);
    }
    hipLaunchKernelGGL(k_ch_layout, dim3(1), dim3(1024), 0, st, nb, (const ChRec *)W.chrec, W.binfo, chw, W.emit, W.piece_base,
                       (uint32_t)W.cap_tiles, (uint32_t)W.cap_pieces, (uint32_t *)W.out, lim_bits, W.chooser, W.carry, W.carry + 1, R.base_bits, do_epilogue,
                       W.blk_entry, R.batch ? W.ent_out : nullptr);
    if (R.batch) {
      // every entry's stream at its own place in the output: sizes -> byte offsets -> positions and epilogues
      const uint32_t E = R.n_entries;
      hipLaunchKernelGGL(k_batch_sizes, dim3((E + 255) / 256), dim3(256), 0, st, E, W.ent_out, W.ent_bytes);
      exclusive_scan_u32(st, W.ent_bytes, W.ent_base, W.scan2, W.total2, E);
      if (nb > 0) hipLaunchKernelGGL(k_batch_rebase, dim3((nb + 255) / 256), dim3(256), 0, st, nb, W.blk_entry, W.ent_base, W.emit);
      hipLaunchKernelGGL(k_batch_finish, dim3((E + 255) / 256), dim3(256), 0, st, E, W.ent_out, W.ent_base, (uint32_t *)W.out, lim_bits);
    }
  }
  uint32_t batch_bytes = 0;
  if (R.batch) hipMemcpyAsync(&batch_bytes, W.total2, 4, hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(&R.co, W.chooser, sizeof(ChooserOut), hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(&R.carry_out, W.carry + 1, sizeof(ChooserCarry), hipMemcpyDeviceToHost, st);
  if (hip_check(c, hipStreamSynchronize(st), "choose")) return ZADA_E_HIP_;
  c->tmark("choose");
  if (R.co.overflow) { c->err = "emission table overflow"; return -1; }
  if (R.batch) R.co.total_bits = 8ull * batch_bytes;
  R.chosen = true;
  c->last_nblocks = R.nblocks;                        // block trace: zada_last_blocks reads emit / blocks from the workspace
  return 0;
}

// Emits the range's bits into W.out (zeroed before entropy_choose; the chooser has already written the stored-block
// headers there) and copies them to d_out if given.  Bit 0 of W.out is stream bit base_bits.
int entropy_emit(Ctx *c, uint8_t *d_out) {
  hipStream_t st = c->stream;
  Workspace &W = c->ws;
  Range &R = c->rg;
  const bool fixed_only = (R.method == 6);
  const uint32_t nblocks = R.nblocks;
  const ChooserOut &co = R.co;
  if ((co.total_bits + 7) / 8 + 8 > W.cap_out) { c->err = "output workspace overflow"; return -1; }
  if ((!fixed_only || R.batch) && nblocks > 0)
    hipLaunchKernelGGL(k_emit_prefix, dim3((nblocks + 255) / 256), dim3(256), 0, st, nblocks, W.emit, W.blocks, W.tile_block, (uint32_t *)W.out);
  hipLaunchKernelGGL(k_block_codes, dim3(nblocks + 2), dim3(64), 0, st, nblocks, W.emit, W.binfo, W.codes, W.carry);
  const EntropyView v = range_view(c);
  if (co.n_tiles > 0) {
    hipLaunchKernelGGL(k_tile_bits, dim3(co.n_tiles), dim3(256), 0, st, nblocks, v.atoms, W.blocks, W.emit, W.tile_block, W.codes, W.tile_bits);
    hipLaunchKernelGGL(k_tile_scan, dim3((nblocks + 255) / 256), dim3(256), 0, st, nblocks, W.blocks, W.emit, W.tile_bits, W.tile_bitpos);
    hipLaunchKernelGGL(k_emit_tiles, dim3(co.n_tiles), dim3(256), 0, st, nblocks, v.atoms, W.blocks, W.emit, W.tile_block, W.codes, W.tile_bitpos, (uint32_t *)W.out);
  }
  if (nblocks > 0 && !fixed_only) hipLaunchKernelGGL(k_emit_headers, dim3(nblocks), dim3(64), 0, st, nblocks, W.emit, W.binfo, (uint32_t *)W.out);
  if (co.n_pieces > 0) {
    hipLaunchKernelGGL(k_ch_stored_emit, dim3((nblocks + 63) / 64), dim3(64), 0, st, nblocks, (const ChRec *)W.chrec, (const ChW *)W.chw, W.emit, W.piece_base, v.apos,
                       W.pieces, (uint32_t *)W.out, W.cap_out * 8);
    hipLaunchKernelGGL(k_copy_pieces, dim3(co.n_pieces, 16), dim3(256), 0, st, co.n_pieces, W.pieces, R.rin, W.out);
  }
  if (d_out) hipMemcpyAsync(d_out, W.out, (co.total_bits + 7) / 8, hipMemcpyDeviceToDevice, st);
  c->tmark("emit");
  return hip_check(c, hipGetLastError(), "entropy_emit");
}

}  // namespace zada
