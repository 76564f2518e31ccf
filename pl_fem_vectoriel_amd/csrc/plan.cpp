// Launch plan of a front tree (plan.h): per-level kernel forms, launch order, compact workgroup lists.  Host-only C++.
#include "plan.h"

#include <algorithm>
#include <cstdlib>
#include <functional>

#include "symbolic.h"

namespace plfem {

void build_launch_plan(const Symbolic& S, LaunchPlan& P, const std::function<void(const std::function<void()>&, const std::function<void()>&)>& run2) {
  const int nf = S.nfronts, L = S.L, dpn = S.dpn;
  // per-front DOF counts + level table
  P.fs2.resize(nf);
  P.fm.resize(nf);
  std::vector<int32_t>& fs2 = P.fs2;
  std::vector<int32_t>& fm = P.fm;
  for (int f = 0; f < nf; ++f) { fs2[f] = dpn * S.fs[f]; fm[f] = dpn * (S.fs[f] + S.fb[f]); }
  P.levels.assign(L + 1, LevelInfo());
  P.worst_m = 0;
  for (int lev = 0; lev <= L; ++lev) {
    LevelInfo& li = P.levels[lev];
    li.first = (1 << lev) - 1;
    li.count = 1 << lev;
    for (int f = li.first; f < li.first + li.count; ++f) {
      li.max_m = std::max(li.max_m, fm[f]);
      li.max_s2 = std::max(li.max_s2, fs2[f]);
      li.max_b2 = std::max(li.max_b2, fm[f] - fs2[f]);
      // one sweep over a front reads the s2 (s2 + 1) / 2 + s2 b2 entries of [L11^-1 ; Z] once, stages s2 (forward)
      // or m (backward) vector entries and writes m (forward) or s2 (backward)
      li.sweep_bytes += 8.0 * ((double)fs2[f] * fm[f] - 0.5 * (double)fs2[f] * fs2[f] + fm[f] + fs2[f]);
      li.sweep_vec_doubles += fm[f] + fs2[f];
    }
    P.worst_m = std::max(P.worst_m, li.max_m);
  }
  // Launch order of the fronts of a level: decreasing s2 (counting sort on s2 / 16, stable).  The fronts still
  // active at a block step of the factorisation are then a prefix, and in every batched launch the long fronts
  // start first.  blk: compact launch lists of the sweep kernels, (front, row block) per useful workgroup.
  P.forder.resize(nf);
  std::vector<int32_t>& forder = P.forder;
  P.forder_s2.assign(nf, 0);
  P.forder_maxm.assign(nf, 0);
  const int mix_big_s2 = getenv("PLFEM_MIX_BIG_S2") ? atoi(getenv("PLFEM_MIX_BIG_S2")) : MIX_BIG_S2;   // (tuning aid)
  {
    std::vector<int32_t> bucket;
    for (int lev = 0; lev <= L; ++lev) {
      LevelInfo& li = P.levels[lev];
      int32_t* o = forder.data() + li.first;
      const int nb = li.max_s2 / 16 + 2;
      bucket.assign(nb, 0);
      for (int f = li.first; f < li.first + li.count; ++f) bucket[nb - 2 - fs2[f] / 16 + 1]++;
      for (int b = 1; b < nb; ++b) bucket[b] += bucket[b - 1];
      for (int f = li.first; f < li.first + li.count; ++f) o[bucket[nb - 2 - fs2[f] / 16]++] = f;
      int mx = 0;
      for (int q = 0; q < li.count; ++q) {
        mx = std::max(mx, fm[o[q]]);
        P.forder_s2[li.first + q] = fs2[o[q]];
        P.forder_maxm[li.first + q] = mx;
      }
      li.fwd_rows = fwd_block_rows(li.count);
      li.bwd_rows = bwd_block_rows(li.count, lev == L);
    }
  }
  // the two families of lists are independent of one another (both follow the launch order above): side by side when the
  // caller offers a second thread
  auto sweep_lists = [&] {
    std::vector<Tile> blk;
    for (int lev = 0; lev <= L; ++lev) {
      LevelInfo& li = P.levels[lev];
      const int32_t* o = forder.data() + li.first;
      li.fwd_off = (int64_t)blk.size();
      for (int q = 0; q < li.count; ++q) {
        const int f = o[q];
        if (li.fwd_rows == 64 && fs2[f] > mix_big_s2) {      // long front of a tile-form level: row-form workgroups
          li.fwd_mixed = true;
          for (int t = 0; t * 16 < fm[f]; ++t) blk.push_back(Tile{f, t | SWEEP_ROW_JOB_FLAG});
        }
        else
          for (int t = 0; t * li.fwd_rows < fm[f]; ++t) blk.push_back(Tile{f, t});
      }
      li.fwd_n = (int)(blk.size() - li.fwd_off);
      li.bwd_off = (int64_t)blk.size();
      // (a non-leaf front without owned DOFs still gets one workgroup: it republishes its boundary values for its children)
      for (int q = 0; q < li.count; ++q) {
        const int f = o[q];
        const int rows = std::max(fs2[f], lev < L ? 1 : 0);
        for (int t = 0; t * li.bwd_rows < rows; ++t) blk.push_back(Tile{f, t});
      }
      li.bwd_n = (int)(blk.size() - li.bwd_off);
    }
  P.jobs.resize(blk.size());
  for (size_t q = 0; q < P.jobs.size(); ++q) {
    const int f = blk[q].x;
    P.jobs[q] = SweepJob{f, blk[q].y, fm[f], fs2[f], S.fnode_ptr[f], f > 0 ? S.fnode_ptr[(f - 1) >> 1] : 0, S.foff[f], 0};
  }
  P.frec.resize(nf);                                       // the fronts in launch order, with their parameters
  for (int q = 0; q < nf; ++q) {
    const int f = forder[q];
    P.frec[q] = FrontRec{f, fm[f], fs2[f], 0, S.foff[f], S.fnode_ptr[f]};
  }
  };
  // 64 x 64 tile lists of the factorisation kernels: only workgroups with work are launched (a dense
  // (tiles of the largest front)^2 x fronts grid is 85-90 % empty workgroups, which cost ~3 ns each)
  auto tile_lists = [&] {
    auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
    // pass 0 counts, pass 1 fills
    int64_t ntiles = 0;
    for (int pass = 0; pass < 2; ++pass) {
      Tile* out = nullptr;
      if (pass == 1) {
        P.tiles.resize((size_t)ntiles);
        out = P.tiles.data();
      }
      P.upd_off.clear();
      P.upd_n.clear();
      int64_t pos = 0;
      auto lower = [&](int f, int nt) {                  // blocks tx >= ty of an nt x nt square (symmetric trailing matrix)
        if (out)
          for (int ty = 0; ty < nt; ++ty)
            for (int tx = ty; tx < nt; ++tx) out[pos++] = Tile{f, tx | (ty << 16)};
        else
          pos += (int64_t)nt * (nt + 1) / 2;
      };
      auto z_blocks = [&](int f) {                       // (f, tb) for the 64-row blocks of Z = rows of F21 (k_form_z_mirror)
        for (int tb = 0; tb < cdiv(fm[f] - fs2[f], 64); ++tb) {
          if (fs2[f] == 0) break;
          if (out) out[pos] = Tile{f, tb};
          ++pos;
        }
      };
      auto block_rows = [&](int f) {                     // (f, kb) for the block rows kb >= 1 of F11, last (longest) first
        for (int kb = cdiv(fs2[f], NB) - 1; kb >= 1; --kb) {
          if (out) out[pos] = Tile{f, kb};
          ++pos;
        }
      };
      for (int lev = 0; lev <= L; ++lev) {
        LevelInfo& li = P.levels[lev];
        const int32_t* o = forder.data() + li.first;
        li.gather_off = pos;
        if (lev < L)
          for (int q = 0; q < li.count; ++q) lower(o[q], cdiv(fm[o[q]], 64));   // nothing reads the blocks above the diagonal
        li.gather_n = (int)(pos - li.gather_off);
        li.step0 = (int)P.upd_n.size();
        const int steps = (li.max_s2 + NB - 1) / NB;
        for (int kb = 0; kb < steps; ++kb) {
          const int k0 = kb * NB;
          P.upd_off.push_back(pos);
          for (int q = 0; q < li.count && fs2[o[q]] > k0; ++q) {     // active fronts: a prefix of the order
            const int f = o[q];
            const int t0 = k0 + std::min(NB, fs2[f] - k0);
            const int nt = cdiv(fm[f] - t0, 64);
            // even step of a front that has a next one: nothing (the columns of its next pivot block are the column
            // workgroups' job, the rest waits for the rank-64 pass of the odd step: k_ldl_update)
            if (!((kb & 1) == 0 && t0 < fs2[f])) lower(f, nt);
          }
          P.upd_n.push_back((int)(pos - P.upd_off.back()));
        }
        li.formz_off = pos;
        for (int q = 0; q < li.count; ++q) z_blocks(o[q]);
        li.formz_n = (int)(pos - li.formz_off);
        li.mirrorx_off = pos;
        for (int q = 0; q < li.count; ++q) block_rows(o[q]);
        li.mirrorx_n = (int)(pos - li.mirrorx_off);
      }
      // the same Z blocks once more as ONE list over all levels, root first: nothing in the factorisation reads Z, so
      // a complete run forms it for every front in a single launch at the end (the per-level lists above serve
      // plfem_debug_factor_until, which stops after a given level)
      P.formz_all_off = pos;
      for (int lev = 0; lev <= L; ++lev) {
        const LevelInfo& li = P.levels[lev];
        const int32_t* o = forder.data() + li.first;
        for (int q = 0; q < li.count; ++q) z_blocks(o[q]);
      }
      P.formz_all_n = (int)(pos - P.formz_all_off);
      // block rows >= 1 of every F11 (k_mirror_x), largest first within a level
      P.mirrorx_all_off = pos;
      for (int lev = 0; lev <= L; ++lev) {
        const LevelInfo& li = P.levels[lev];
        const int32_t* o = forder.data() + li.first;
        for (int q = 0; q < li.count; ++q) block_rows(o[q]);
      }
      P.mirrorx_all_n = (int)(pos - P.mirrorx_all_off);
      ntiles = pos;
    }
  };
  if (run2) run2(sweep_lists, tile_lists);
  else { sweep_lists(); tile_lists(); }
  // panel / block-row scratch of the factorisation: one tree level is in flight at a time, so these are sized by the
  // level with the most (padded) nodes and addressed relative to the level's first front (launch_factor)
  int64_t level_nodes = 0;
  for (int lev = 0; lev <= L; ++lev) {
    const int first = (1 << lev) - 1, last = std::min(nf, (1 << (lev + 1)) - 1);
    level_nodes = std::max(level_nodes, S.fnode_ptr[last] - S.fnode_ptr[first]);
  }
  P.level_nodes_max = level_nodes;
  P.built = true;
}

}  // namespace plfem
