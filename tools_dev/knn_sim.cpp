// dev tool (CPU): lock-step model of the k-NN pre-pass (K4) on one cloud - how many loop steps and insertion chains a
// 64-lane wave executes: round A = own row + face rows, round B = corner rows + pruned 5x5x5 shell as ONE flat round
// (the LDS segment-table design of DESIGN.md 6a (ix), measured slower and removed), ORDER=1|2 = candidates of round A in
// cell-class / exact distance order.  input: raw float32 xyz (e.g. slam3d_amd.synthetic.make_pair(100000, 0)[0].tofile).  g++ -O2 -std=c++17 -I. tools_dev/knn_sim.cpp -o /tmp/sim/knn_sim
#include "../tests/emu/emu_pipeline.cpp"
#include <cstdio>
#include <set>

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "/tmp/sim/a.bin";
  const double leaf = argc > 2 ? atof(argv[2]) : 0.2;
  const int K = 20;
  FILE* f = fopen(path, "rb");
  std::vector<float> xyz;
  float buf[3];
  while (fread(buf, 4, 3, f) == 3) xyz.insert(xyz.end(), buf, buf + 3);
  fclose(f);
  Cloud c = voxel(xyz.data(), (int)(xyz.size() / 3), 3, leaf);
  const float h0 = argc > 3 ? atof(argv[3]) : h0_for(leaf);
  Grid G = build_grid(c, h0, argc > 4 ? atoi(argv[4]) : 2);
  const GridParams& g = G.g;
  printf("points %zu filtered %zu  grid %d x %d x %d h %.3f  pts/nonempty cell: ", xyz.size() / 3, c.pts.size(), g.dim[0], g.dim[1], g.dim[2], g.h);
  { int ne = 0; for (int i = 0; i < g.ncells; ++i) ne += G.cell_start[i + 1] > G.cell_start[i]; printf("%.2f\n", (double)c.pts.size() / ne); }
  const int n = (int)c.pts.size();
  // per lane: the candidate sequence of phase 1 (5 rows) and phase 2 (4 corner rows, after pruning), lock-step per wave
  double steps1 = 0, steps2 = 0, chains = 0, cand_sum = 0, acc_sum = 0, acc_max_sum = 0, ring2 = 0, waves = 0, cand_max_sum = 0;
  double chains_b4 = 0, steps_b4 = 0;
  std::vector<int> perm(n);
  for (int i = 0; i < n; ++i) perm[i] = i;
  if (getenv("BLOCKSORT")) {   // experiment: the queries of a 256-thread block dealt to its waves by candidate count
    const int B = atoi(getenv("BLOCKSORT"));
    std::vector<uint32_t> tot(n);
    for (int i = 0; i < n; ++i) {
      const F4& q = G.sorted[i];
      const int ix = grid_coord(g, 0, q.x), iy = grid_coord(g, 1, q.y), iz = grid_coord(g, 2, q.z);
      const int xa = std::max(ix - 1, 0), xb = std::min(ix + 1, g.dim[0] - 1);
      uint32_t t = 0;
      for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) {
        const int cy = iy + dy, cz = iz + dz;
        if (cy < 0 || cy >= g.dim[1] || cz < 0 || cz >= g.dim[2]) continue;
        const int rb = g.dim[0] * (cy + g.dim[1] * cz);
        t += G.cell_start[rb + xb + 1] - G.cell_start[rb + xa];
      }
      tot[i] = t;
    }
    for (int b0 = 0; b0 < n; b0 += B)
      std::stable_sort(perm.begin() + b0, perm.begin() + std::min(n, b0 + B), [&](int a, int b) { return tot[a] < tot[b]; });
  }
  for (int w0 = 0; w0 < n; w0 += 64) {
    const int nl = std::min(64, n - w0);
    std::vector<std::vector<uint32_t>> seq1(nl), seq2(nl);
    std::vector<std::vector<double>> keys(nl, std::vector<double>(K, 1e300));
    std::vector<int> acc(nl, 0);
    struct LaneGeo { int ix, iy, iz; float qx, qy, qz; uint32_t rs[9], re[9]; };
    std::vector<LaneGeo> L(nl);
    for (int l = 0; l < nl; ++l) {
      const F4& q = G.sorted[perm[w0 + l]];
      LaneGeo& A = L[l];
      A.qx = q.x; A.qy = q.y; A.qz = q.z;
      A.ix = grid_coord(g, 0, q.x); A.iy = grid_coord(g, 1, q.y); A.iz = grid_coord(g, 2, q.z);
      const int xa = std::max(A.ix - 1, 0), xb = std::min(A.ix + 1, g.dim[0] - 1);
      for (int r = 0; r < 9; ++r) {
        const int rr = r == 0 ? 4 : (r <= 4 ? 2 * r - 1 : (r == 5 ? 0 : (r == 6 ? 2 : (r == 7 ? 6 : 8))));
        const int cy = A.iy + (rr % 3) - 1, cz = A.iz + (rr / 3) - 1;
        const bool ok = cy >= 0 && cy < g.dim[1] && cz >= 0 && cz < g.dim[2];
        const int rowbase = ok ? g.dim[0] * (cy + g.dim[1] * cz) : 0;
        A.rs[r] = G.cell_start[rowbase + (ok ? xa : 0)];
        A.re[r] = ok ? G.cell_start[rowbase + xb + 1] : A.rs[r];
      }
      for (int r = 0; r < 5; ++r) for (uint32_t t = A.rs[r]; t < A.re[r]; ++t) seq1[l].push_back(t);
    }
    auto keyof = [&](int l, uint32_t t) {
      const F4& p = G.sorted[t];
      const float d2 = dist2(L[l].qx, L[l].qy, L[l].qz, p.x, p.y, p.z);
      return (double)d2 * 4294967296.0 + (double)__builtin_bit_cast(uint32_t, p.w) * 1e-3;   // order only
    };
    auto run = [&](std::vector<std::vector<uint32_t>>& seq, double& steps) {
      size_t mx = 0;
      for (auto& s : seq) mx = std::max(mx, s.size());
      for (size_t t = 0; t < mx; t += 2) {
        steps += 1;
        for (int u = 0; u < 2; ++u) {
          bool any = false;
          for (int l = 0; l < nl; ++l) {
            if (t + u >= seq[l].size()) continue;
            const double c = keyof(l, seq[l][t + u]);
            if (c < keys[l][K - 1]) {
              any = true; ++acc[l];
              auto it = std::upper_bound(keys[l].begin(), keys[l].end(), c);
              keys[l].insert(it, c); keys[l].pop_back();
            }
          }
          chains += any;
        }
      }
      // batch-of-4 model: one merge per 4 candidates when any of them is accepted by any lane (same acceptance as above: upper bound)
      steps_b4 += (mx + 3) / 4;
    };
    if (getenv("ORDER")) {   // experiment: candidates of round A ordered by cell class (own cell, face cells, edge cells)
      const int mode = atoi(getenv("ORDER"));
      for (int l = 0; l < nl; ++l) {
        auto cls = [&](uint32_t t) {
          const F4& p = G.sorted[t];
          const int dx = abs(grid_coord(g, 0, p.x) - L[l].ix), dy = abs(grid_coord(g, 1, p.y) - L[l].iy), dz = abs(grid_coord(g, 2, p.z) - L[l].iz);
          if (mode == 1) return (double)(dx + dy + dz);
          return (double)dist2(L[l].qx, L[l].qy, L[l].qz, p.x, p.y, p.z);   // mode 2: exact distance order (the ideal)
        };
        std::stable_sort(seq1[l].begin(), seq1[l].end(), [&](uint32_t a, uint32_t b) { return cls(a) < cls(b); });
      }
    }
    run(seq1, steps1);
    static double chainsA = 0; chainsA = chains;
    if (w0 + 64 >= n) printf("chains after round A per wave: %.1f\n", chainsA / (waves + 1));
    // NEW design: round B = corner rows + ring-2 shell, all cut to the ball of the k-th distance after round A
    static double segs_sum = 0, segs_max_sum = 0, notfull = 0, over16 = 0, late = 0, potential = 0;
    std::vector<char> ring2_done(nl, 0);
    for (int l = 0; l < nl; ++l) {
      LaneGeo& A = L[l];
      const bool full = keys[l][K - 1] < 1e299;
      if (!full) notfull += 1;
      const float lim2 = full ? (float)(keys[l][K - 1] / 4294967296.0) : 3e38f;
      const float eps = 2.0e-3f * g.h;
      float ox = (A.qx - g.origin[0]) * g.inv_h - A.ix, oy = (A.qy - g.origin[1]) * g.inv_h - A.iy, oz = (A.qz - g.origin[2]) * g.inv_h - A.iz;
      float face = fminf(fminf(fminf(ox, 1.f - ox), fminf(oy, 1.f - oy)), fminf(oz, 1.f - oz));
      face = fmaxf(face - 2.0e-3f, 0.f);
      const float b2 = (2.0f + face) * g.h;
      const bool wide = full && lim2 <= b2 * b2;   // the ball lies inside the 5x5x5 cells
      int nseg = 0, ncorner = 0;
      std::vector<std::pair<uint32_t, uint32_t>> segs;
      for (int dz = -2; dz <= 2; ++dz) for (int dy = -2; dy <= 2; ++dy) {
        const bool inner = abs(dy) <= 1 && abs(dz) <= 1;
        const bool arow = inner && (dy == 0 || dz == 0);
        if (!inner && !wide) continue;
        const int cy = A.iy + dy, cz = A.iz + dz;
        if (cy < 0 || cy >= g.dim[1] || cz < 0 || cz >= g.dim[2]) continue;
        const float ylo = g.origin[1] + (float)cy * g.h, zlo = g.origin[2] + (float)cz * g.h;
        const float fy2 = fmaxf(fmaxf(ylo - A.qy, A.qy - (ylo + g.h)) - eps, 0.f), fz2 = fmaxf(fmaxf(zlo - A.qz, A.qz - (zlo + g.h)) - eps, 0.f);
        const float rowd2 = fy2 * fy2 + fz2 * fz2;
        if (rowd2 > lim2) continue;
        int xa, xb;
        if (wide) {
          const float rx = sqrtf(fmaxf(lim2 - rowd2, 0.f)) * 1.0001f + eps;
          xa = std::max(std::max(A.ix - 2, grid_coord(g, 0, A.qx - rx)), 0);
          xb = std::min(std::min(A.ix + 2, grid_coord(g, 0, A.qx + rx)), g.dim[0] - 1);
        } else { xa = std::max(A.ix - 1, 0); xb = std::min(A.ix + 1, g.dim[0] - 1); }
        const int rowbase = g.dim[0] * (cy + g.dim[1] * cz);
        for (int part = 0; part < 2; ++part) {
          int sa = xa, sb = xb;
          if (arow) { if (part == 0) sb = std::min(xb, A.ix - 2); else sa = std::max(xa, A.ix + 2); }
          else if (part) break;
          potential += 1;
          if (sa > sb) continue;
          const uint32_t s0 = G.cell_start[rowbase + sa], e0 = G.cell_start[rowbase + sb + 1];
          if (e0 > s0) segs.push_back({s0, e0});
        }
      }
      if (wide) ring2_done[l] = 1;
      if (segs.size() > 16) over16 += 1;
      segs_sum += segs.size();
      for (auto& sg : segs) for (uint32_t t = sg.first; t < sg.second; ++t) seq2[l].push_back(t);
    }
    { size_t m = 0; for (int l = 0; l < nl; ++l) m = std::max(m, seq2[l].size()); }
    run(seq2, steps2);
    for (int l = 0; l < nl; ++l) {
      const LaneGeo& A = L[l];
      float ox = (A.qx - g.origin[0]) * g.inv_h - A.ix, oy = (A.qy - g.origin[1]) * g.inv_h - A.iy, oz = (A.qz - g.origin[2]) * g.inv_h - A.iz;
      float face = fminf(fminf(fminf(ox, 1.f - ox), fminf(oy, 1.f - oy)), fminf(oz, 1.f - oz));
      face = fmaxf(face - 2.0e-3f, 0.f);
      const float bound = (1.0f + face) * g.h;
      const bool ok3 = keys[l][K - 1] < 1e299 && (float)(keys[l][K - 1] / 4294967296.0) <= bound * bound;
      if (!ring2_done[l] && !ok3) late += 1;
    }
    if (w0 + 64 >= n) printf("NEW: per query: potential segments %.1f stored %.2f ; lanes not full after A %.2f %%, >16 segments %.3f %%, needing the slow path after B %.3f %%\n",
                             potential / n, segs_sum / n, 100 * notfull / n, 100 * over16 / n, 100 * late / n);
    int amax = 0; size_t cmax = 0;
    for (int l = 0; l < nl; ++l) {
      cand_sum += seq1[l].size() + seq2[l].size(); acc_sum += acc[l]; amax = std::max(amax, acc[l]);
      cmax = std::max(cmax, seq1[l].size() + seq2[l].size());
      const LaneGeo& A = L[l];
      float ox = (A.qx - g.origin[0]) * g.inv_h - A.ix, oy = (A.qy - g.origin[1]) * g.inv_h - A.iy, oz = (A.qz - g.origin[2]) * g.inv_h - A.iz;
      float face = fminf(fminf(fminf(ox, 1.f - ox), fminf(oy, 1.f - oy)), fminf(oz, 1.f - oz));
      face = fmaxf(face - 2.0e-3f, 0.f);
      const float bound = (1.0f + face) * g.h;
      if (!(keys[l][K - 1] < 1e299 && (float)(keys[l][K - 1] / 4294967296.0) <= bound * bound)) ring2 += 1;
    }
    acc_max_sum += amax; cand_max_sum += cmax;
    waves += 1;
  }
  printf("waves %.0f  per wave: steps phase1 %.1f phase2 %.1f  chains %.1f (of %.1f slots)\n", waves, steps1 / waves, steps2 / waves, chains / waves,
         2 * (steps1 + steps2) / waves);
  printf("per query: candidates %.1f accepted %.1f ; per wave max candidates %.1f max accepted %.1f ; queries needing ring 2: %.2f %%\n", cand_sum / n, acc_sum / n,
         cand_max_sum / waves, acc_max_sum / waves, 100.0 * ring2 / n);
  printf("model: instr per wave ~ %.0f (steps x 65 + chains x 46)\n", 65 * (steps1 + steps2) / waves + 46 * chains / waves);
  return 0;
}
