// Launch plan of a front tree: everything the device kernels are launched from that depends on the MESH only -- the
// per-level kernel forms, the launch order of the fronts and the compact workgroup lists of the sweeps and of the
// factorisation.  Built once per analysis (plan.cpp, at the end of build_symbolic, beside the flattening of the front
// lists) and shared by every device context on that analysis; plfem_create only uploads it.  Host-only header (no HIP).
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <utility>
#include <vector>

namespace plfem {

// std::vector whose resize() leaves the new elements uninitialised: the big index arrays of the front
// tree are filled in parallel, and a value-initialising resize would first touch every page serially.
template <class T>
struct default_init_allocator : std::allocator<T> {
  template <class U>
  struct rebind { using other = default_init_allocator<U>; };
  using std::allocator<T>::allocator;
  template <class U>
  void construct(U* ptr) noexcept { ::new (static_cast<void*>(ptr)) U; }
  template <class U, class... Args>
  void construct(U* ptr, Args&&... args) { ::new (static_cast<void*>(ptr)) U(std::forward<Args>(args)...); }
};

constexpr int NB = 32;          // pivot-block width of the block LDL^T
constexpr int PANEL_CHUNK = 1024; // rows per partial sum of the tall-skinny panel products
// sweep kernel forms by level (launch_solve_p and the launch lists must agree)
constexpr int ROW_FORM_MAX_FRONTS = 32;   // levels with at most this many fronts use the row-form kernels in both sweeps
// 8 / 16: pure row form with that many rows per workgroup; 64: tile form (backward: leaf level only; forward: mixed
// launch -- tiles of 64 rows, row-form workgroups of 16 rows for the fronts with more than MIX_BIG_S2 owned DOFs)
constexpr int MIX_BIG_S2 = 192;
constexpr int SWEEP_ROW_JOB_FLAG = 1 << 30;
// One workgroup of a sweep kernel: the front, its row block and everything the workgroup would otherwise look up by
// front number (one dependent memory round trip less in front of every launch of a latency-bound level)
struct alignas(16) SweepJob {
  int32_t f, rb;          // front, row block (| SWEEP_ROW_JOB_FLAG in the mixed forward kernel)
  int32_t m, s2;          // order of the front, owned DOFs
  int64_t np, npp;        // fnode_ptr of the front and of its parent
  int64_t foff;           // offset of the front in d_front
  int64_t reserved;
};
static_assert(sizeof(SweepJob) == 48, "SweepJob layout");
// A front as the factorisation's chain workgroups (first panel, column workgroups) see it, in launch order (d_forder)
struct alignas(16) FrontRec {
  int32_t f, m, s2, reserved;
  int64_t foff, np;       // offset in d_front, fnode_ptr
};
static_assert(sizeof(FrontRec) == 32, "FrontRec layout");
inline int fwd_block_rows(int count) { return count <= 8 ? 8 : count <= ROW_FORM_MAX_FRONTS ? 16 : 64; }
// (round 3 re-measured the tile form at the two levels above the leaves: 28.1 / 25.5 us against 22.3 / 21.5 us in row form)
inline int bwd_block_rows(int count, bool leaf) { return leaf ? 64 : count <= ROW_FORM_MAX_FRONTS ? 8 : 16; }
#ifndef PLFEM_BLOCK_P
#define PLFEM_BLOCK_P 4             // (-DPLFEM_BLOCK_P=8: the experiment of DESIGN section 11; the product is built and tested with 4)
#endif
constexpr int BLOCK_P = PLFEM_BLOCK_P;      // right-hand sides per block solve / block Lanczos step
static_assert(BLOCK_P == 4 || BLOCK_P == 8, "a power of two: multi_reduce of the row-form sweeps");
constexpr int ELEM_FORMS = 8;   // Axx Axy Ayx Ayy Minv Dxx Dxy Dyy
constexpr int ELEM_STRIDE = ELEM_FORMS * 36;

struct LevelInfo {
  int first = 0;    // first front id of the level (heap order)
  int count = 0;
  int max_m = 0;    // DOFs
  int max_s2 = 0;
  int max_b2 = 0;
  // compact launch lists of the sweep kernels (d_blk): one entry per useful workgroup = (front, row block),
  // fronts in order of decreasing work so that the long ones start first
  // factorisation: 64 x 64 tile lists (d_tiles) of the extend-add, of Z, and (per block step: upd_off / upd_n of
  // the context, index step0 + kb) of the trailing updates
  int64_t gather_off = 0, formz_off = 0, mirrorx_off = 0;
  int gather_n = 0, formz_n = 0, mirrorx_n = 0, step0 = 0;
  int fwd_rows = 0, bwd_rows = 0;      // rows per workgroup of the forward / backward kernel of this level
  bool fwd_mixed = false;              // tile-form level with at least one long front (row-form workgroups in the same launch)
  int64_t fwd_off = 0, bwd_off = 0;    // first entry in d_blk
  int fwd_n = 0, bwd_n = 0;            // entries = workgroups
  double sweep_bytes = 0;   // algorithmic bytes one forward (or backward) sweep launch of this level moves (1 rhs)
  double sweep_vec_doubles = 0;   // vector doubles (staged + written) per rhs of that launch
};

struct Tile { int32_t x, y; };      // (front, tx | ty << 16): layout of HIP's int2
static_assert(sizeof(Tile) == 8, "Tile layout");

struct Symbolic;
struct LaunchPlan {
  bool built = false;
  std::vector<int32_t> fs2, fm;        // per front: owned DOFs, order of the front (DOFs)
  std::vector<int32_t> forder;         // per level: front ids in order of decreasing s2
  std::vector<int32_t> forder_s2, forder_maxm;   // s2 in that order, running maximum of m in that order
  std::vector<FrontRec, default_init_allocator<FrontRec>> frec;   // the fronts in launch order, with their parameters (this and the two lists below: filled in parallel into uninitialised storage)
  std::vector<LevelInfo> levels;
  std::vector<SweepJob, default_init_allocator<SweepJob>> jobs;   // one entry per sweep workgroup, level by level (forward then backward lists)
  std::vector<Tile, default_init_allocator<Tile>> tiles;   // 64 x 64 workgroup lists of the factorisation kernels
  std::vector<int64_t> upd_off;        // per (level, block step): first entry / entries of the trailing-update list
  std::vector<int32_t> upd_n;
  int64_t formz_all_off = 0, mirrorx_all_off = 0;
  int32_t formz_all_n = 0, mirrorx_all_n = 0;
  int64_t level_nodes_max = 0;         // sum of (padded) nodes over the fronts of one tree level, largest level
  int32_t worst_m = 0;                 // largest front order (LDS staging limit of the sweeps)
};
// par (may be empty): runs f(0) .. f(ntasks - 1), side by side if it can (the analysis lends part of its team); the plan is
// the same either way
using PlanTasks = std::function<void(int ntasks, const std::function<void(int)>& f)>;
void build_launch_plan(const Symbolic& S, LaunchPlan& P, const PlanTasks& par = {});

}  // namespace plfem
