// Launch plan of a front tree (plan.h): per-level kernel forms, launch order, compact workgroup lists.  Host-only C++.
#include "plan.h"

#include <algorithm>
#include <cstdlib>
#include <functional>

#include "symbolic.h"

namespace plfem {

void build_launch_plan(const Symbolic& S, LaunchPlan& P, const PlanTasks& par) {
  // tasks side by side when the caller lends threads (the analysis' team), else one after the other; the result does not
  // depend on it: every list is written at offsets fixed by a counting pass
  auto run = [&](int ntasks, const std::function<void(int)>& f) {
    if (par) par(ntasks, f);
    else for (int q = 0; q < ntasks; ++q) f(q);
  };
  const int nf = S.nfronts, L = S.L, dpn = S.dpn;
  // per-front DOF counts + level table
  P.fs2.resize(nf);
  P.fm.resize(nf);
  std::vector<int32_t>& fs2 = P.fs2;
  std::vector<int32_t>& fm = P.fm;
  P.levels.assign(L + 1, LevelInfo());
  P.forder.resize(nf);
  std::vector<int32_t>& forder = P.forder;
  P.forder_s2.assign(nf, 0);
  P.forder_maxm.assign(nf, 0);
  const int mix_big_s2 = getenv("PLFEM_MIX_BIG_S2") ? atoi(getenv("PLFEM_MIX_BIG_S2")) : MIX_BIG_S2;   // (tuning aid)
  auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
  // ---- the generators of the workgroup lists: out == nullptr counts, else fills; both passes run the same code ----------
  // sweep workgroups of one level and direction (launch order: decreasing s2)
  auto gen_jobs = [&](int lev, bool fwd, SweepJob* out) -> int64_t {
    const LevelInfo& li = P.levels[lev];
    const int32_t* o = forder.data() + li.first;
    int64_t n = 0;
    auto put = [&](int f, int rb) {
      if (out) out[n] = SweepJob{f, rb, fm[f], fs2[f], S.fnode_ptr[f], f > 0 ? S.fnode_ptr[(f - 1) >> 1] : 0, S.foff[f], 0};
      ++n;
    };
    for (int q = 0; q < li.count; ++q) {
      const int f = o[q];
      if (fwd) {
        if (li.fwd_rows == 64 && fs2[f] > mix_big_s2)          // long front of a tile-form level: row-form workgroups
          for (int t = 0; t * 16 < fm[f]; ++t) put(f, t | SWEEP_ROW_JOB_FLAG);
        else
          for (int t = 0; t * li.fwd_rows < fm[f]; ++t) put(f, t);
      } else {
        // (a non-leaf front without owned DOFs still gets one workgroup: it republishes its boundary values for its children)
        const int rows = std::max(fs2[f], lev < L ? 1 : 0);
        for (int t = 0; t * li.bwd_rows < rows; ++t) put(f, t);
      }
    }
    return n;
  };
  // 64 x 64 tile lists of the factorisation kernels: only workgroups with work are launched (a dense
  // (tiles of the largest front)^2 x fronts grid is 85-90 % empty workgroups, which cost ~3 ns each)
  auto lower = [](int f, int nt, Tile* out, int64_t& pos) {    // blocks tx >= ty of an nt x nt square (symmetric trailing matrix)
    if (out)
      for (int ty = 0; ty < nt; ++ty)
        for (int tx = ty; tx < nt; ++tx) out[pos++] = Tile{f, tx | (ty << 16)};
    else
      pos += (int64_t)nt * (nt + 1) / 2;
  };
  auto gen_gather = [&](int lev, Tile* out) -> int64_t {      // extend-add of the level's fronts (nothing reads the blocks above the diagonal)
    const LevelInfo& li = P.levels[lev];
    const int32_t* o = forder.data() + li.first;
    int64_t pos = 0;
    if (lev < L)
      for (int q = 0; q < li.count; ++q) lower(o[q], cdiv(fm[o[q]], 64), out, pos);
    return pos;
  };
  auto gen_update = [&](int lev, int kb, Tile* out) -> int64_t {   // trailing update of block step kb
    const LevelInfo& li = P.levels[lev];
    const int32_t* o = forder.data() + li.first;
    const int k0 = kb * NB;
    int64_t pos = 0;
    for (int q = 0; q < li.count && fs2[o[q]] > k0; ++q) {     // active fronts: a prefix of the order
      const int f = o[q];
      const int t0 = k0 + std::min(NB, fs2[f] - k0);
      // even step of a front that has a next one: nothing (the columns of its next pivot block are the column
      // workgroups' job, the rest waits for the rank-64 pass of the odd step: k_ldl_update)
      if (!((kb & 1) == 0 && t0 < fs2[f])) lower(f, cdiv(fm[f] - t0, 64), out, pos);
    }
    return pos;
  };
  auto gen_formz = [&](int lev, Tile* out) -> int64_t {       // (f, tb) for the 64-row blocks of Z = rows of F21 (k_form_z_mirror)
    const LevelInfo& li = P.levels[lev];
    const int32_t* o = forder.data() + li.first;
    int64_t pos = 0;
    for (int q = 0; q < li.count; ++q) {
      const int f = o[q];
      if (fs2[f] == 0) continue;
      for (int tb = 0; tb < cdiv(fm[f] - fs2[f], 64); ++tb) {
        if (out) out[pos] = Tile{f, tb};
        ++pos;
      }
    }
    return pos;
  };
  auto gen_mirrorx = [&](int lev, Tile* out) -> int64_t {     // (f, kb) for the block rows kb >= 1 of F11, last (longest) first
    const LevelInfo& li = P.levels[lev];
    const int32_t* o = forder.data() + li.first;
    int64_t pos = 0;
    for (int q = 0; q < li.count; ++q)
      for (int kb = cdiv(fs2[o[q]], NB) - 1; kb >= 1; --kb) {
        if (out) out[pos] = Tile{o[q], kb};
        ++pos;
      }
    return pos;
  };
  // ---- pass 1, a task per level: sizes, launch order, kernel forms, list lengths -------------------------------------------
  struct Counts { int64_t fwd = 0, bwd = 0, gather = 0, formz = 0, mirrorx = 0; std::vector<int64_t> upd; };
  std::vector<Counts> cnt(L + 1);
  run(L + 1, [&](int lev) {
    LevelInfo& li = P.levels[lev];
    li.first = (1 << lev) - 1;
    li.count = 1 << lev;
    for (int f = li.first; f < li.first + li.count; ++f) {
      fs2[f] = dpn * S.fs[f];
      fm[f] = dpn * (S.fs[f] + S.fb[f]);
      li.max_m = std::max(li.max_m, fm[f]);
      li.max_s2 = std::max(li.max_s2, fs2[f]);
      li.max_b2 = std::max(li.max_b2, fm[f] - fs2[f]);
      // one sweep over a front reads the s2 (s2 + 1) / 2 + s2 b2 entries of [L11^-1 ; Z] once, stages s2 (forward)
      // or m (backward) vector entries and writes m (forward) or s2 (backward)
      li.sweep_bytes += 8.0 * ((double)fs2[f] * fm[f] - 0.5 * (double)fs2[f] * fs2[f] + fm[f] + fs2[f]);
      li.sweep_vec_doubles += fm[f] + fs2[f];
    }
    // Launch order of the fronts of a level: decreasing s2 (counting sort on s2 / 16, stable).  The fronts still
    // active at a block step of the factorisation are then a prefix, and in every batched launch the long fronts
    // start first.
    int32_t* o = forder.data() + li.first;
    const int nb = li.max_s2 / 16 + 2;
    std::vector<int32_t> bucket(nb, 0);
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
    for (int q = 0; q < li.count; ++q)
      if (li.fwd_rows == 64 && fs2[o[q]] > mix_big_s2) li.fwd_mixed = true;
    Counts& c = cnt[lev];
    c.fwd = gen_jobs(lev, true, nullptr);
    c.bwd = gen_jobs(lev, false, nullptr);
    c.gather = gen_gather(lev, nullptr);
    const int steps = (li.max_s2 + NB - 1) / NB;
    c.upd.resize(steps);
    for (int kb = 0; kb < steps; ++kb) c.upd[kb] = gen_update(lev, kb, nullptr);
    c.formz = gen_formz(lev, nullptr);
    c.mirrorx = gen_mirrorx(lev, nullptr);
  });
  // ---- offsets (the order of the lists in d_blk / d_tiles is the one every earlier round had) ------------------------------
  P.worst_m = 0;
  struct Fill { int kind, lev, kb; int64_t off; };            // kind: 0 jobs fwd, 1 jobs bwd, 2 gather, 3 update, 4 formz, 5 mirrorx, 6 formz (all), 7 mirrorx (all), 8 front records
  std::vector<Fill> fills;
  int64_t njobs = 0, pos = 0;
  P.upd_off.clear();
  P.upd_n.clear();
  for (int lev = 0; lev <= L; ++lev) {
    LevelInfo& li = P.levels[lev];
    const Counts& c = cnt[lev];
    P.worst_m = std::max(P.worst_m, li.max_m);
    li.fwd_off = njobs; li.fwd_n = (int)c.fwd;
    fills.push_back({0, lev, 0, njobs});
    njobs += c.fwd;
    li.bwd_off = njobs; li.bwd_n = (int)c.bwd;
    fills.push_back({1, lev, 0, njobs});
    njobs += c.bwd;
    li.gather_off = pos; li.gather_n = (int)c.gather;
    if (c.gather) fills.push_back({2, lev, 0, pos});
    pos += c.gather;
    li.step0 = (int)P.upd_n.size();
    for (int kb = 0; kb < (int)c.upd.size(); ++kb) {
      P.upd_off.push_back(pos);
      P.upd_n.push_back((int)c.upd[kb]);
      if (c.upd[kb]) fills.push_back({3, lev, kb, pos});
      pos += c.upd[kb];
    }
    li.formz_off = pos; li.formz_n = (int)c.formz;
    if (c.formz) fills.push_back({4, lev, 0, pos});
    pos += c.formz;
    li.mirrorx_off = pos; li.mirrorx_n = (int)c.mirrorx;
    if (c.mirrorx) fills.push_back({5, lev, 0, pos});
    pos += c.mirrorx;
  }
  // the same Z blocks once more as ONE list over all levels, root first: nothing in the factorisation reads Z, so
  // a complete run forms it for every front in a single launch at the end (the per-level lists above serve
  // plfem_debug_factor_until, which stops after a given level)
  P.formz_all_off = pos;
  for (int lev = 0; lev <= L; ++lev) {
    if (cnt[lev].formz) fills.push_back({6, lev, 0, pos});
    pos += cnt[lev].formz;
  }
  P.formz_all_n = (int)(pos - P.formz_all_off);
  // block rows >= 1 of every F11 (mirror of X), largest first within a level
  P.mirrorx_all_off = pos;
  for (int lev = 0; lev <= L; ++lev) {
    if (cnt[lev].mirrorx) fills.push_back({7, lev, 0, pos});
    pos += cnt[lev].mirrorx;
  }
  P.mirrorx_all_n = (int)(pos - P.mirrorx_all_off);
  for (int lev = 0; lev <= L; ++lev) fills.push_back({8, lev, 0, 0});
  P.jobs.resize((size_t)njobs);
  P.tiles.resize((size_t)pos);
  P.frec.resize(nf);                                          // the fronts in launch order, with their parameters
  // ---- pass 2, a task per list: fill ---------------------------------------------------------------------------------------
  // (largest lists first: the leaf level's come last in the order above)
  std::stable_sort(fills.begin(), fills.end(), [&](const Fill& a, const Fill& b) { return a.lev > b.lev; });
  run((int)fills.size(), [&](int t) {
    const Fill& F = fills[t];
    switch (F.kind) {
      case 0: gen_jobs(F.lev, true, P.jobs.data() + F.off); break;
      case 1: gen_jobs(F.lev, false, P.jobs.data() + F.off); break;
      case 2: gen_gather(F.lev, P.tiles.data() + F.off); break;
      case 3: gen_update(F.lev, F.kb, P.tiles.data() + F.off); break;
      case 4: case 6: gen_formz(F.lev, P.tiles.data() + F.off); break;
      case 5: case 7: gen_mirrorx(F.lev, P.tiles.data() + F.off); break;
      default: {
        const LevelInfo& li = P.levels[F.lev];
        for (int q = li.first; q < li.first + li.count; ++q) {
          const int f = forder[q];
          P.frec[q] = FrontRec{f, fm[f], fs2[f], 0, S.foff[f], S.fnode_ptr[f]};
        }
      }
    }
  });
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
