// dev tool (CPU, round 6): WHY the k-NN fast path (grid_knn_med3) declines points of a cloud on the fused grid - the
// reasons counted one by one, and what a wider table / more rings would answer.
// g++ -O2 -std=c++17 -I. tools_dev/knn3_declines.cpp -o /tmp/sim/knn3_declines ; knn3_declines cloud.bin leaf [cells_per_point]
#include "../tests/emu/emu_pipeline.cpp"
#include <cstdio>

int main(int argc, char** argv) {
  const char* path = argv[1];
  const double leaf = argc > 2 ? atof(argv[2]) : 0.2;
  const int cpp = argc > 3 ? atoi(argv[3]) : 2;
  FILE* f = fopen(path, "rb");
  std::vector<float> xyz;
  float buf[4];
  const int rec = argc > 4 ? atoi(argv[4]) : 4;
  while (fread(buf, 4, rec, f) == (size_t)rec) xyz.insert(xyz.end(), buf, buf + 3);
  fclose(f);
  Fused F = build_fused(xyz.data(), (int)(xyz.size() / 3), 3, leaf, cpp);
  const GridParams& g = F.G.g;
  const uint32_t* cs = F.G.cell_start.data();
  const F4* pts = F.G.sorted.data();
  const int n = (int)F.G.sorted.size();
  printf("raw %zu filtered %d grid %d x %d x %d h %.3f m %d\n", xyz.size() / 3, n, g.dim[0], g.dim[1], g.dim[2], g.h, F.fz.m);
  long long r_build27 = 0, r_few = 0, r_beyond = 0, r_shell_tab = 0, r_tie = 0, ok = 0, need_shell = 0;
  long long seg27_hist[32] = {0}, segshell_hist[64] = {0};
  long long beyond_ring[16] = {0};
  for (int i = 0; i < n; ++i) {
    const F4& q = pts[i];
    uint32_t tab[256];
    uint32_t keys[21];
    for (int j = 0; j < 21; ++j) keys[j] = kKnn3Sentinel;
    int nseg; uint32_t total;
    if (!knn3_build27(g, cs, q.x, q.y, q.z, tab, 1, nseg, total)) { ++r_build27; continue; }
    seg27_hist[nseg]++;
    knn3_scan<21>(keys, tab, 1, 0, nseg, pts, q.x, q.y, q.z);
    float lim2 = 0.f;
    const float face = knn3_face(g, q.x, q.y, q.z);
    const int st = knn3_after27<21>(g, face, keys, lim2);
    if (st == 2) {
      if (keys[19] == kKnn3Sentinel) ++r_few; else ++r_beyond;
      // the true K-th distance in rings: exact search
      unsigned long long ref[20];
      grid_knn_sorted<20, true, true>(g, cs, pts, q.x, q.y, q.z, 20, ref);
      const float d2k = knn_key_d2(__builtin_bit_cast(double, ref[19]));
      int ring = (int)ceilf(sqrtf(d2k) / g.h - face);
      beyond_ring[std::min(std::max(ring, 0), 15)]++;
      continue;
    }
    if (st == 1) {
      ++need_shell;
      // count the shell's segments with an unbounded table
      int ns = nseg;
      // re-implementation of the push without the cap: count segments
      const int first = nseg;
      uint32_t big[256];
      std::memcpy(big, tab, sizeof(uint32_t) * nseg);
      // emulate knn3_build_shell but counting (use a huge local cap by temporarily copying code): we call it and detect failure
      int nseg2 = nseg; uint32_t tot2;
      const bool fit = knn3_build_shell(g, cs, q.x, q.y, q.z, lim2, tab, 1, nseg2, tot2);
      // count real segments needed: walk rows
      {
        const int ix = grid_coord(g, 0, q.x), iy = grid_coord(g, 1, q.y), iz = grid_coord(g, 2, q.z);
        const float eps = 2.0e-3f * g.h;
        for (int dz = -2; dz <= 2; ++dz) for (int dy = -2; dy <= 2; ++dy) {
          const int cz = iz + dz, cy = iy + dy;
          if (cz < 0 || cz >= g.dim[2] || cy < 0 || cy >= g.dim[1]) continue;
          const float zlo = g.origin[2] + (float)cz * g.h, ylo = g.origin[1] + (float)cy * g.h;
          const float fz2 = fmaxf(fmaxf(zlo - q.z, q.z - (zlo + g.h)) - eps, 0.f), fy2 = fmaxf(fmaxf(ylo - q.y, q.y - (ylo + g.h)) - eps, 0.f);
          const float rowd2 = fy2 * fy2 + fz2 * fz2;
          if (rowd2 > lim2) continue;
          const float rx = sqrtf(fmaxf(lim2 - rowd2, 0.f)) * 1.0001f + eps;
          const int xa = imax(imax(ix - 2, grid_coord(g, 0, q.x - rx)), 0), xb = imin(imin(ix + 2, grid_coord(g, 0, q.x + rx)), g.dim[0] - 1);
          const bool inner = dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
          const int rowbase = g.dim[0] * (cy + g.dim[1] * cz);
          if (!inner) { if (xa <= xb && cs[rowbase + xb + 1] > cs[rowbase + xa]) ++ns; }
          else {
            const int lb = imin(xb, ix - 2), ra = imax(xa, ix + 2);
            if (xa <= lb && cs[rowbase + lb + 1] > cs[rowbase + xa]) ++ns;
            if (ra <= xb && cs[rowbase + xb + 1] > cs[rowbase + ra]) ++ns;
          }
        }
      }
      segshell_hist[std::min(ns, 63)]++;
      if (!fit) { ++r_shell_tab; continue; }
      knn3_scan<21>(keys, tab, 1, first, nseg2, pts, q.x, q.y, q.z);
    }
    if (!knn3_unambiguous<21>(keys)) { ++r_tie; continue; }
    ++ok;
  }
  printf("answered %lld (%.1f %%), needed the shell %lld (%.1f %%)\n", ok, 100.0 * ok / n, need_shell, 100.0 * need_shell / n);
  printf("declined: 27-cell table / long range %lld (%.2f %%), fewer than K in 27 cells %lld (%.2f %%), K-th beyond the 5x5x5 proof %lld (%.2f %%), shell table full %lld (%.2f %%), tie at the K-th %lld (%.2f %%)\n",
         r_build27, 100.0 * r_build27 / n, r_few, 100.0 * r_few / n, r_beyond, 100.0 * r_beyond / n, r_shell_tab, 100.0 * r_shell_tab / n, r_tie, 100.0 * r_tie / n);
  printf("segments after the 27 cells:"); for (int i = 0; i < 16; ++i) printf(" %d:%lld", i, seg27_hist[i]); printf("\n");
  printf("segments 27 + shell (shell queries):"); for (int i = 0; i < 64; ++i) if (segshell_hist[i]) printf(" %d:%lld", i, segshell_hist[i]); printf("\n");
  printf("true K-th distance of the far declines, in rings (ceil(d/h - face)):"); for (int i = 0; i < 16; ++i) if (beyond_ring[i]) printf(" %d:%lld", i, beyond_ring[i]); printf("\n");
  return 0;
}
