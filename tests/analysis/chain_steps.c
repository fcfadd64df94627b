// Analysis tool (not a test): share of the level-6 chain steps spent at positions the parser actually lands on.
// Build: gcc -O2 -o /tmp/chain_steps tests/analysis/chain_steps.c zip-ada_amd/csrc/silesia_mix.c -Loracle -lzada_oracle -lm -Wl,-rpath,$PWD/oracle ; run: /tmp/chain_steps 16
// Result for 16 MiB of silesia_mix_v1: profiles/r1/NOTES.md.
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include "../../oracle/zada_oracle.h"
void zada_silesia_mix(uint64_t seed, uint32_t class_mask, uint64_t offset, uint64_t len, uint8_t *dst);
int main(int argc, char **argv) {
  uint64_t n = (argc > 1 ? atoi(argv[1]) : 16) << 20;
  uint8_t *in = malloc(n + 512); memset(in + n, 0, 512);
  zada_silesia_mix(0x5A1E51A, 0x1F, 0, n, in);
  // true 6-gram chains through a hash table with verification (chains of the 16-bit hash have a few more collisions)
  uint32_t *head = malloc(sizeof(uint32_t) << 22); memset(head, 0xFF, sizeof(uint32_t) << 22);
  uint32_t *prev = malloc(n * 4);
  uint32_t *steps = calloc(n, 4);
  for (uint64_t p = 0; p + 8 < n; p++) {
    uint64_t v; memcpy(&v, in + p, 8); v &= 0xFFFFFFFFFFFFull;
    uint32_t h = (uint32_t)((v * 0x9E3779B97F4A7C15ull) >> 42);
    prev[p] = head[h]; head[h] = (uint32_t)p;
    // walk: candidates within MAX_DIST, stop when a 258 match is found or 4096 candidates
    uint32_t c = prev[p]; int best = 5, st = 0;
    while (c != 0xFFFFFFFFu && p - c <= 32505 && st < 4096) {
      st++;
      if (in[c + best] == in[p + best] && in[c + best - 1] == in[p + best - 1]) {
        int len = 0; while (len < 258 && in[c + len] == in[p + len]) len++;
        if (len > best) { best = len; if (len >= 258) break; }
      }
      c = prev[c];
    }
    steps[p] = st;
  }
  uint32_t *tok = malloc((n + 8) * 4); uint64_t nt = 0;
  nt = zo_lz77_tokens(in, n, 10, tok, n + 8);
  uint8_t *land = calloc(n + 2, 1);
  uint64_t pos = 0;
  for (uint64_t i = 0; i < nt; i++) {
    land[pos] = 1;
    if (tok[i] & 0x80000000u) { land[pos + 1] = 1; pos += (tok[i] >> 16) & 0x7FFF; } else pos += 1;
  }
  double s_all = 0, s_land = 0; uint64_t n_land = 0;
  uint64_t hist_all[6] = {0}, hist_land[6] = {0}; double sh_all[6] = {0}, sh_land[6] = {0};
  for (uint64_t p = 0; p < n; p++) {
    int b = steps[p] <= 8 ? 0 : steps[p] <= 32 ? 1 : steps[p] <= 128 ? 2 : steps[p] <= 512 ? 3 : steps[p] <= 2048 ? 4 : 5;
    s_all += steps[p]; hist_all[b]++; sh_all[b] += steps[p];
    if (land[p]) { s_land += steps[p]; n_land++; hist_land[b]++; sh_land[b] += steps[p]; }
  }
  printf("n=%llu tokens=%llu landing positions=%llu (%.1f%%)\n", (unsigned long long)n, (unsigned long long)nt, (unsigned long long)n_land, 100.0 * n_land / n);
  printf("steps/pos all=%.1f ; landing only=%.1f per landing pos; share of all steps spent at landings = %.1f%%\n", s_all / n, s_land / n_land, 100.0 * s_land / s_all);
  const char *nm[6] = {"<=8", "<=32", "<=128", "<=512", "<=2048", ">2048"};
  for (int b = 0; b < 6; b++) printf("  steps %-6s: %5.1f%% of positions, %5.1f%% of steps | landings: %5.1f%% of these positions, %5.1f%% of these steps\n", nm[b], 100.0 * hist_all[b] / n, 100.0 * sh_all[b] / s_all, 100.0 * hist_land[b] / (hist_all[b] + 1e-9), 100.0 * sh_land[b] / (sh_all[b] + 1e-9));
  return 0;
}
