// gfx950 kernels: the two solve sweeps of the multifrontal shift-invert operator (lu.solve of the reference's
// eigsh(..., sigma=...): scipy arpack.py:920-928 via reference solver_fem.py:197), P right-hand sides at a time
// (P = 1: plfem_solve and the single-vector Lanczos; P = 4: block Lanczos -- every entry of the factors is read
// once for P vectors).
//
//   forward : [t; u]  = [L11^-1 r ;  w_b - Z r],          r = rhs_own + children's updates
//   backward: x_own   = L11^-T ys - Z^T x_b,              ys = D^-1 t
// D is block diagonal with one 2 x 2 block per node pair (kernels_front.hip): ys_i = dd_i t_i + od_i t_(i^1).  The
// backward sweep applies it while it stages ys (a thread reads t_i, its partner t_(i^1) and the row's two entries of
// D^-1: one memory round trip, nothing crosses lanes), so the forward kernels store their sums as they are.
//
// One kernel per tree level and direction.  A workgroup of such a kernel is latency, not bandwidth (a few tens of
// KB of factor entries each), so everything is laid out for ONE memory round trip before the arithmetic starts:
//   * all vectors live in FRONT ORDER (entry (2 fnode_ptr[f] + i) * P + u for local DOF i of front f): the
//     right-hand side arrives permuted (k_permute_in, or straight from the Lanczos block scaling), so staging a
//     front's vector is a contiguous read, not node index -> value;
//   * a front PUSHES its update u into its parent's rows (u0: pushed by left children, u1: by right children; prow =
//     local node index in the parent, precomputed on the host), so the parent reads contiguous slots masked by its
//     child maps instead of chasing index -> child vector;
//   * the backward sweep keeps every front's complete local solution (xl); a child reads its boundary values from
//     the parent's xl through prow -- the index loads do not depend on data and are requested first;
//   * the epilogue operands (D, the row's own update slots, the push destination) and the first batch of matrix
//     entries are requested BEFORE the staged vector is written to LDS, so they travel together.
// Forward "tile" form (levels with many fronts): lane = output row, the waves split the columns, partial sums
// meet in LDS in a fixed order; "row" form (at most 32 fronts): a wave owns R rows of [L11^-1 ; Z] and runs along
// their columns in the mirrored upper storage.  Backward: row form (contiguous columns of the lower storage)
// except at the leaf level.  The cross-lane sums of the row forms use multi_reduce.  Deterministic throughout.
#include <algorithm>
#include <cstdlib>

#include "device.h"

#ifndef PLFEM_SWEEP_TB
#define PLFEM_SWEEP_TB 8
#endif
#ifndef PLFEM_SWEEP_TB_FWD_TILE
#define PLFEM_SWEEP_TB_FWD_TILE 8
#endif
#ifndef PLFEM_SWEEP_TB_BWD
#define PLFEM_SWEEP_TB_BWD 8
#endif
#ifndef PLFEM_SWEEP_ROWLOADS
#define PLFEM_SWEEP_ROWLOADS 8
#endif
#ifndef PLFEM_SWEEP_LD_TILE_DEFAULT
#define PLFEM_SWEEP_LD_TILE_DEFAULT 4
#endif
#ifndef PLFEM_SWEEP_LD_ROWS_DEFAULT
#define PLFEM_SWEEP_LD_ROWS_DEFAULT 4
#endif

namespace plfem {
namespace {

struct SweepArgs {
  int leaf_level;
  int dbg;                         // timing experiments only (plfem_debug_solve_block): 1 skip fronts with s2 > 128, 2 only those
  int ldv;                         // row forms: leading dimension of the staged vector in LDS (component-major [u][i])
  int sh;                          // unknowns per node - 1: node of a local DOF = i >> sh, component = i & sh
  const int32_t *cinv0, *cinv1, *prow;
  const double* front;
  const double2* dinv2;             // D^-1 of every front row: (diagonal, off-diagonal entry of the row's node pair)
  double *fr, *u0, *u1, *ys, *xl;
};

// Sums V (a power of two <= 64) per-lane values over the 64 lanes with V - 1 + log2(64 / V) shuffles instead of
// 6 V: each butterfly step halves the values a lane still carries.  On return a[0] of lane l is the complete sum
// of value multi_reduce_index<V>(l); lanes 0 .. V-1 cover every value once.  Fixed order, so deterministic.
template <int V>
__device__ __forceinline__ int multi_reduce_index(int lane) {
  int idx = 0;
#pragma unroll
  for (int h = V / 2, s = 0; h >= 1; h >>= 1, ++s) idx += ((lane >> s) & 1) ? h : 0;
  return idx;
}

template <int V>
__device__ __forceinline__ void multi_reduce(double (&a)[V], int lane) {
#pragma unroll
  for (int h = V / 2, bit = 1; h >= 1; h >>= 1, bit <<= 1) {
    const bool up = (lane & bit) != 0;
#pragma unroll
    for (int k = 0; k < h; ++k) {
      const double send = up ? a[k] : a[k + h];
      const double keep = up ? a[k + h] : a[k];
      a[k] = keep + __shfl_xor(send, bit);
    }
  }
#pragma unroll
  for (int off = V; off < 64; off <<= 1) a[0] += __shfl_xor(a[0], off);
}

// LDS layout of the staged vector: tile forms read one entry for all lanes (broadcast) and keep the P values of a DOF
// together ([i][P]: two 16-byte reads); row forms read a different DOF per lane, where [i][P] is a 32-byte lane stride
// (4-way bank conflicts: 60 % of the LDS cycles of those kernels) -- they use [u][i] (lane stride 8 bytes).
template <int P, bool SOA>
__device__ __forceinline__ int sidx(int i, int u, int ldv) { return SOA ? u * ldv + i : i * P + u; }

// ---- staging -------------------------------------------------------------------------------------------------
// forward: r_i = fr_i (owned rows) + the children's pushes (masked by the child maps)
template <int P>
struct FwdStage {
  double a[P], b0[P], b1[P];
  int c0, c1;
  __device__ __forceinline__ void request(const SweepArgs& A, int64_t np, int i, bool on) {
    c0 = c1 = -1;
#pragma unroll
    for (int u = 0; u < P; ++u) { a[u] = 0.0; b0[u] = 0.0; b1[u] = 0.0; }
    if (!on) return;
    const int64_t e = (2 * np + i) * P;
#pragma unroll
    for (int u = 0; u < P; ++u) a[u] = A.fr[e + u];
    if (!A.leaf_level) {
      c0 = A.cinv0[np + (i >> A.sh)];
      c1 = A.cinv1[np + (i >> A.sh)];
#pragma unroll
      for (int u = 0; u < P; ++u) { b0[u] = A.u0[e + u]; b1[u] = A.u1[e + u]; }   // unconditional: no dependent load
    }
  }
  __device__ __forceinline__ double value(int u) const {
    return a[u] + (c0 >= 0 ? b0[u] : 0.0) + (c1 >= 0 ? b1[u] : 0.0);
  }
};

template <int P, bool SOA>
__device__ __forceinline__ void stage_fwd(const SweepArgs& A, double* sv, int64_t np, int need, int T, int tid,
                                          FwdStage<P>& first) {
  for (int i = tid + T; i < need; i += T) {            // (fronts with more than T owned DOFs only)
    FwdStage<P> s;
    s.request(A, np, i, true);
#pragma unroll
    for (int u = 0; u < P; ++u) sv[sidx<P, SOA>(i, u, A.ldv)] = s.value(u);
  }
  first.request(A, np, tid, tid < need);
}

// epilogue of one forward row r (already summed: acc = [L11^-1 ; Z] r): owned rows -> t = acc, boundary rows ->
// u = w - acc pushed into the parent's slot.  The operands are requested by `request` before the sums.
template <int P>
struct FwdOut {
  double w0[P], w1[P];
  int c0, c1;
  int64_t dst;
  __device__ __forceinline__ void request(const SweepArgs& A, int64_t npp, int64_t np, int s2, int m, int r) {
    c0 = c1 = -1;
    dst = -1;
#pragma unroll
    for (int u = 0; u < P; ++u) { w0[u] = 0.0; w1[u] = 0.0; }
    if (r >= m || r < s2) return;
    const int pr = A.prow[np + (r >> A.sh)];
    if (!A.leaf_level) {
      c0 = A.cinv0[np + (r >> A.sh)];
      c1 = A.cinv1[np + (r >> A.sh)];
      const int64_t e = (2 * np + r) * P;
#pragma unroll
      for (int u = 0; u < P; ++u) { w0[u] = A.u0[e + u]; w1[u] = A.u1[e + u]; }
    }
    if (pr >= 0) dst = (2 * npp + (pr << A.sh) + (r & A.sh)) * P;
  }
  __device__ __forceinline__ double w(int u) const { return (c0 >= 0 ? w0[u] : 0.0) + (c1 >= 0 ? w1[u] : 0.0); }
};

// ---- forward, tile form ------------------------------------------------------------------------------------------
// (SweepJob.rb carries SWEEP_ROW_JOB_FLAG in the mixed kernels: row-form workgroup (16 rows) instead of a tile (64 rows))
// 16-BYTE LOADS (round 4): an 8-byte access per lane streams at 0.54-0.70 of the rate of a 16-byte one on this chip
// (MI355X_MICROARCH.md, cache-policy table), and the leaf-level kernels sat right at that ceiling (4.5 TB/s forward, 3.2
// backward).  A 16-byte access in the column-major factors is TWO CONSECUTIVE ROWS of one column, so a lane of the tile
// forms now owns a row pair (2 rp, 2 rp + 1), rp = lane & 31, and the two halves of a wave take two different columns of
// the same 64-row tile: one wave-level load = 2 columns x 64 rows = two 512-byte runs.  The halves' sums meet by one
// v_permlane32-style shuffle (lane ^ 32) before the waves' sums meet in LDS as before.
constexpr int TB = PLFEM_SWEEP_TB;      // matrix loads in flight per lane and trip (a long front is a chain of such trips)
template <int P, int NW, int TBF = TB>
__device__ __forceinline__ void fwd_tile_body(const SweepArgs& A, const SweepJob& J, double* __restrict__ sv, double* __restrict__ red) {
  const int f = J.f, m = J.m, s2 = J.s2;
  if (A.dbg && ((A.dbg == 1) == (s2 > 128))) return;
  const int r0 = (J.rb & ~SWEEP_ROW_JOB_FLAG) * 64;
  const int64_t np = J.np;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int need = (r0 + 64 <= s2) ? r0 + 64 : s2;
  const int r = r0 + lane;                                           // (epilogue: wave 0, lane = row)
  const bool valid = r < m;
  // this lane's row pair and column slot: columns c = 2 (wave + NW t) + h
  const int h = lane >> 5, ra = r0 + 2 * (lane & 31), rb = ra + 1;
  const bool pvalid = ra < m;                                        // (m is even: both rows or none)
  const int cea = pvalid ? ((ra < s2) ? ra + 1 : s2) : 0;           // rows of L11^-1 are lower triangular
  const int ceb = pvalid ? ((rb < s2) ? rb + 1 : s2) : 0;           // (ceb >= cea)
  const double* p = A.front + J.foff + ra;
  FwdOut<P> out;
  if (wave == 0) out.request(A, J.npp, np, s2, m, r);
  FwdStage<P> st;
  stage_fwd<P, false>(A, sv, np, need, NW * 64, tid, st);
  // first batch of this half-wave's columns, requested before the staged vector is complete
  double2 a0[TBF];
#pragma unroll
  for (int t = 0; t < TBF; ++t) {
    const int c = 2 * (wave + NW * t) + h;
    a0[t] = (c < ceb) ? *reinterpret_cast<const double2*>(p + (int64_t)c * m) : make_double2(0.0, 0.0);
  }
  if (tid < need) {
#pragma unroll
    for (int u = 0; u < P; ++u) sv[tid * P + u] = st.value(u);
  }
  __syncthreads();
  double acc[2][P];
#pragma unroll
  for (int u = 0; u < P; ++u) { acc[0][u] = 0.0; acc[1][u] = 0.0; }
#pragma unroll
  for (int t = 0; t < TBF; ++t) {
    const int c = 2 * (wave + NW * t) + h;
    if (c < ceb) {
      const double ax = (c < cea) ? a0[t].x : 0.0;                   // (c == rb: the entry above the diagonal is L11^-T's)
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const double v = sv[c * P + u];
        acc[0][u] += ax * v;
        acc[1][u] += a0[t].y * v;
      }
    }
  }
  for (int c0 = 2 * NW * TBF; c0 < ceb; c0 += 2 * NW * TBF) {         // (c0 + the lane's slot: uniform trip start)
    double2 a[TBF];
#pragma unroll
    for (int t = 0; t < TBF; ++t) {
      const int c = c0 + 2 * (wave + NW * t) + h;
      a[t] = (c < ceb) ? *reinterpret_cast<const double2*>(p + (int64_t)c * m) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int t = 0; t < TBF; ++t) {
      const int c = c0 + 2 * (wave + NW * t) + h;
      if (c < ceb) {
        const double ax = (c < cea) ? a[t].x : 0.0;
#pragma unroll
        for (int u = 0; u < P; ++u) {
          const double v = sv[c * P + u];
          acc[0][u] += ax * v;
          acc[1][u] += a[t].y * v;
        }
      }
    }
  }
  // the two halves of the wave hold the same row pairs: add them, then the waves' sums meet in LDS in a fixed order
#pragma unroll
  for (int u = 0; u < P; ++u) {
    acc[0][u] += __shfl_xor(acc[0][u], 32);
    acc[1][u] += __shfl_xor(acc[1][u], 32);
  }
  if (h == 0) {
#pragma unroll
    for (int u = 0; u < P; ++u)
      *reinterpret_cast<double2*>(&red[(wave * P + u) * 64 + 2 * lane]) = make_double2(acc[0][u], acc[1][u]);
  }
  __syncthreads();
  if (wave == 0 && valid) {
    double tot[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[(w * P + u) * 64 + lane];
      tot[u] = t;
    }
    if (r < s2) {
#pragma unroll
      for (int u = 0; u < P; ++u) A.ys[(2 * np + r) * P + u] = tot[u];
    } else if (out.dst >= 0) {
      double* d = ((f & 1) ? A.u0 : A.u1) + out.dst;
#pragma unroll
      for (int u = 0; u < P; ++u) d[u] = out.w(u) - tot[u];
    }
  }
}

// ---- forward, row form -------------------------------------------------------------------------------------------
// (16-byte loads: the lane index runs along the row's contiguous run, two entries per lane; the staged vector is read as
// double2 from its [u][i] planes, whose leading dimension ldv is even)
template <int P, int NW, int R, int LOADS = PLFEM_SWEEP_ROWLOADS>
__device__ __forceinline__ void fwd_rows_body(const SweepArgs& A, const SweepJob& J, double* __restrict__ sv) {
  constexpr int RB = NW * R, UNR = LOADS / R, V = R * P;
  const int f = J.f, m = J.m, s2 = J.s2;
  if (A.dbg && ((A.dbg == 1) == (s2 > 128))) return;
  const int j0 = (J.rb & ~SWEEP_ROW_JOB_FLAG) * RB;
  const int64_t np = J.np;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int need = min(s2, j0 + RB);                      // rows < s2 only read r[0 .. row]
  const double* F = A.front + J.foff;
  // row r of [L11^-1; Z] as a contiguous run: r < s2 -> column r of [F11; F21] (the upper mirror of L11^-1), r >= s2 ->
  // column r - s2 of Z^T (s2 x b2, stored behind [F11; F21])
  int ce[R], cmax = 0;
  const double* rowp[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int r = j0 + wave + NW * q;
    ce[q] = r < m ? ((r < s2) ? r + 1 : s2) : 0;
    cmax = max(cmax, ce[q]);
    rowp[q] = r < s2 ? F + (int64_t)r * m : F + (int64_t)m * s2 + (int64_t)(r - s2) * s2;
  }
  // the lane that will hold output (q, u) after the reduction requests that row's epilogue operands now
  const int oidx = multi_reduce_index<V>(lane & (V - 1));
  const int orow = j0 + wave + NW * (oidx / P);
  FwdOut<P> out;
  out.request(A, J.npp, np, s2, m, lane < V ? orow : m);
  FwdStage<P> st;
  stage_fwd<P, true>(A, sv, np, need, NW * 64, tid, st);
  double2 a0[UNR][R];
#pragma unroll
  for (int t = 0; t < UNR; ++t) {
    const int i = 128 * t + 2 * lane;
#pragma unroll
    for (int q = 0; q < R; ++q) a0[t][q] = (i < ce[q]) ? *reinterpret_cast<const double2*>(rowp[q] + i) : make_double2(0.0, 0.0);
  }
  if (tid < need) {
#pragma unroll
    for (int u = 0; u < P; ++u) sv[u * A.ldv + tid] = st.value(u);
  }
  __syncthreads();
  double acc[V];
#pragma unroll
  for (int v = 0; v < V; ++v) acc[v] = 0.0;
  // (an entry past the end of its row -- i + 1 == ce for the odd end of a triangular row -- is masked: what lies there is
  // the other triangle's; the staged vector is finite everywhere below ldv)
#define PLFEM_FWD_ROWS_CONSUME(a, c)                                                                   \
  _Pragma("unroll") for (int t = 0; t < UNR; ++t) {                                                    \
    const int i = (c) + 128 * t + 2 * lane;                                                            \
    if (i < cmax) {                                                                                    \
      _Pragma("unroll") for (int u = 0; u < P; ++u) {                                                  \
        const double2 vi = *reinterpret_cast<const double2*>(&sv[u * A.ldv + i]);                      \
        _Pragma("unroll") for (int q = 0; q < R; ++q) {                                                \
          const double ay = (i + 1 < ce[q]) ? a[t][q].y : 0.0;                                         \
          acc[q * P + u] = fma(ay, vi.y, fma(a[t][q].x, vi.x, acc[q * P + u]));                        \
        }                                                                                              \
      }                                                                                                \
    }                                                                                                  \
  }
  PLFEM_FWD_ROWS_CONSUME(a0, 0)
  for (int c = 128 * UNR; c < cmax; c += 128 * UNR) {
    double2 a[UNR][R];
#pragma unroll
    for (int t = 0; t < UNR; ++t) {
      const int i = c + 128 * t + 2 * lane;
#pragma unroll
      for (int q = 0; q < R; ++q) a[t][q] = (i < ce[q]) ? *reinterpret_cast<const double2*>(rowp[q] + i) : make_double2(0.0, 0.0);
    }
    PLFEM_FWD_ROWS_CONSUME(a, c)
  }
#undef PLFEM_FWD_ROWS_CONSUME
  multi_reduce<V>(acc, lane);
  if (lane < V && orow < m) {
    const int u = oidx % P;
    if (orow < s2) {
      A.ys[(2 * np + orow) * P + u] = acc[0];
    } else if (out.dst >= 0) {
      double wu = out.w(0);
#pragma unroll
      for (int q = 1; q < P; ++q) wu = (u == q) ? out.w(q) : wu;
      (((f & 1) ? A.u0 : A.u1) + out.dst)[u] = wu - acc[0];
    }
  }
}

// ---- backward ------------------------------------------------------------------------------------------------------
// stage v = [ys ; -x_b] of front f in LDS (entries >= lo only), [dof][P]; x_b comes from the parent's local solution
// through prow.  The workgroup with publish = true also writes the boundary part of this front's own local solution
// (its children will read it).  `first`: the entry of this thread in the first chunk, completed by finish().
template <int P>
struct BwdStage {
  double v[P];
  int pr, i;
  bool own, on;
  __device__ __forceinline__ void request_index(const SweepArgs& A, int64_t np, int s2, int i_, bool on_) {
    i = i_;
    on = on_;
    own = i < s2;
    pr = -1;
    if (on && !own) pr = A.prow[np + (i >> A.sh)];
    if (on && own) {                        // ys_i = dd_i t_i + od_i t_(i^1)  (s2 is even: the partner is an owned row too)
      const double2 d = A.dinv2[2 * np + i];
#pragma unroll
      for (int u = 0; u < P; ++u) v[u] = fma(d.x, A.ys[(2 * np + i) * P + u], d.y * A.ys[(2 * np + (i ^ 1)) * P + u]);
    }
  }
  __device__ __forceinline__ void request_value(const SweepArgs& A, int64_t npp) {
    if (on && !own) {
#pragma unroll
      for (int u = 0; u < P; ++u) v[u] = pr >= 0 ? -A.xl[(2 * npp + (pr << A.sh) + (i & A.sh)) * P + u] : 0.0;
    }
  }
  template <bool SOA>
  __device__ __forceinline__ void finish(const SweepArgs& A, double* sv, int64_t np, bool publish) const {
    if (!on) return;
#pragma unroll
    for (int u = 0; u < P; ++u) sv[sidx<P, SOA>(i, u, A.ldv)] = v[u];
    if (publish && !own) {
#pragma unroll
      for (int u = 0; u < P; ++u) A.xl[(2 * np + i) * P + u] = -v[u];
    }
  }
};

template <int P, bool SOA>
__device__ __forceinline__ void stage_bwd_rest(const SweepArgs& A, double* sv, int lo, int m, int s2, int64_t np,
                                               int64_t npp, int T, int tid, bool publish) {
  for (int i = lo + tid + T; i < m; i += T) {             // (fronts with more than T DOFs behind lo only)
    BwdStage<P> s;
    s.request_index(A, np, s2, i, true);
    s.request_value(A, npp);
    s.template finish<SOA>(A, sv, np, publish);
  }
}

template <int P, int NW, int TBB = PLFEM_SWEEP_TB_BWD>
__device__ __forceinline__ void bwd_tile_body(const SweepArgs& A, const SweepJob& J, double* __restrict__ sv, double* __restrict__ red) {
  const int rb = J.rb, m = J.m, s2 = J.s2;
  if (A.dbg && ((A.dbg == 1) == (s2 > 128))) return;
  const int r0 = rb * 64;
  const int64_t np = J.np;
  const int64_t npp = J.npp;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool publish = rb == 0 && !A.leaf_level;
  BwdStage<P> st;
  st.request_index(A, np, s2, r0 + tid, r0 + tid < m);
  const int r = r0 + lane;                                           // (epilogue: wave 0, lane = row)
  const bool valid = r < s2;
  // 16-byte loads: this lane's row pair (ra, ra + 1) and column slot: columns i = slot (mod 2 NW), slot = 2 wave + h.
  // Element (j, i) of [L11^-T | Z^T]: column i of [F11; F21] at row j for i < s2 (upper mirror), of Z^T (leading
  // dimension s2, behind [F11; F21]) for i >= s2; row j takes the columns i >= j.
  const int h = lane >> 5, ra = r0 + 2 * (lane & 31);
  const bool pvalid = ra < s2;                                       // (s2 is even: both rows or none)
  const double* p = A.front + J.foff + ra;
  const double* pz = A.front + J.foff + (int64_t)m * s2 + ra - (int64_t)s2 * s2;     // pz[i s2] for i >= s2
  auto entry = [&](int i) { return *reinterpret_cast<const double2*>(i < s2 ? p + (int64_t)i * m : pz + (int64_t)i * s2); };
  const int cb = pvalid ? ra : m, ce = m;
  const int slot = 2 * wave + h;
  const int cstart = cb + ((slot - cb) & (2 * NW - 1));
  double2 a0[TBB];
#pragma unroll
  for (int t = 0; t < TBB; ++t) a0[t] = (cstart + 2 * NW * t < ce) ? entry(cstart + 2 * NW * t) : make_double2(0.0, 0.0);
  st.request_value(A, npp);
  stage_bwd_rest<P, false>(A, sv, r0, m, s2, np, npp, NW * 64, tid, publish);
  st.template finish<false>(A, sv, np, publish);
  __syncthreads();
  double acc[2][P];
#pragma unroll
  for (int u = 0; u < P; ++u) { acc[0][u] = 0.0; acc[1][u] = 0.0; }
#pragma unroll
  for (int t = 0; t < TBB; ++t) {
    const int c = cstart + 2 * NW * t;
    if (c < ce) {
      const double ay = (c > ra) ? a0[t].y : 0.0;                    // (c == ra: below the diagonal lies L11^-1's entry)
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const double v = sv[c * P + u];
        acc[0][u] += a0[t].x * v;
        acc[1][u] += ay * v;
      }
    }
  }
  for (int c0 = cstart + 2 * NW * TBB; c0 < ce; c0 += 2 * NW * TBB) {
    double2 a[TBB];
#pragma unroll
    for (int t = 0; t < TBB; ++t) a[t] = (c0 + 2 * NW * t < ce) ? entry(c0 + 2 * NW * t) : make_double2(0.0, 0.0);
#pragma unroll
    for (int t = 0; t < TBB; ++t) {
      const int c = c0 + 2 * NW * t;
      if (c < ce) {
#pragma unroll
        for (int u = 0; u < P; ++u) {
          const double v = sv[c * P + u];
          acc[0][u] += a[t].x * v;
          acc[1][u] += a[t].y * v;                                   // (c > ra here: past the first batch)
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < P; ++u) {
    acc[0][u] += __shfl_xor(acc[0][u], 32);
    acc[1][u] += __shfl_xor(acc[1][u], 32);
  }
  if (h == 0) {
#pragma unroll
    for (int u = 0; u < P; ++u)
      *reinterpret_cast<double2*>(&red[(wave * P + u) * 64 + 2 * lane]) = make_double2(acc[0][u], acc[1][u]);
  }
  __syncthreads();
  if (wave == 0 && valid) {
#pragma unroll
    for (int u = 0; u < P; ++u) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[(w * P + u) * 64 + lane];
      A.xl[(2 * np + r) * P + u] = t;
    }
  }
}

// Backward sweep, row form: x_j = sum_{i >= j} [L11^-1 ; Z](i, j) v_i with v = [ys ; -x_b] staged in LDS.
// Column j of the lower storage is contiguous in i, so a wave reads 1-KB runs (two entries per lane, 16-byte loads); a wave
// owns R rows (j0 + wave + NW q) and keeps R x 8/R loads in flight; a block (NW waves) shares one staged vector for NW R rows.
template <int P, int NW, int R, int LOADS = PLFEM_SWEEP_ROWLOADS>
__device__ __forceinline__ void bwd_rows_body(const SweepArgs& A, const SweepJob& J, double* __restrict__ sv) {
  constexpr int RB = NW * R, UNR = LOADS / R, V = R * P;
  const int rb = J.rb, m = J.m, s2 = J.s2;
  if (A.dbg && ((A.dbg == 1) == (s2 > 128))) return;
  const int j0 = rb * RB;
  const int64_t np = J.np;
  const int64_t npp = J.npp;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool publish = rb == 0 && !A.leaf_level;
  const int lo = j0 & ~63;
  BwdStage<P> st;
  st.request_index(A, np, s2, lo + tid, lo + tid < m);
  const double* F = A.front + J.foff;
  int jq[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int j = j0 + wave + NW * q;
    jq[q] = j < s2 ? j : m;                                   // rows past the owned block: every term masked
  }
  const int c0 = (j0 + wave) & ~63;
  double2 a0[UNR][R];
#pragma unroll
  for (int t = 0; t < UNR; ++t) {
    const int i = c0 + 128 * t + 2 * lane;
#pragma unroll
    for (int q = 0; q < R; ++q) a0[t][q] = (i < m && i + 1 >= jq[q]) ? *reinterpret_cast<const double2*>(F + i + (int64_t)jq[q] * m) : make_double2(0.0, 0.0);
  }
  st.request_value(A, npp);
  stage_bwd_rest<P, true>(A, sv, lo, m, s2, np, npp, NW * 64, tid, publish);
  st.template finish<true>(A, sv, np, publish);
  __syncthreads();
  const int oidx = multi_reduce_index<V>(lane & (V - 1));
  const int oj = j0 + wave + NW * (oidx / P);
  double acc[V];
#pragma unroll
  for (int v = 0; v < V; ++v) acc[v] = 0.0;
  // (an entry above its column's start -- i == jq - 1 for an odd jq -- is masked: it belongs to the other triangle; the
  // staged vector is finite from lo on)
#define PLFEM_BWD_ROWS_CONSUME(a, c)                                                                   \
  _Pragma("unroll") for (int t = 0; t < UNR; ++t) {                                                    \
    const int i = (c) + 128 * t + 2 * lane;                                                            \
    if (i < m) {                                                                                       \
      _Pragma("unroll") for (int u = 0; u < P; ++u) {                                                  \
        const double2 vi = *reinterpret_cast<const double2*>(&sv[u * A.ldv + i]);                      \
        _Pragma("unroll") for (int q = 0; q < R; ++q) {                                                \
          const double ax = (i >= jq[q]) ? a[t][q].x : 0.0;                                            \
          acc[q * P + u] = fma(a[t][q].y, vi.y, fma(ax, vi.x, acc[q * P + u]));                        \
        }                                                                                              \
      }                                                                                                \
    }                                                                                                  \
  }
  PLFEM_BWD_ROWS_CONSUME(a0, c0)
  for (int c = c0 + 128 * UNR; c < m; c += 128 * UNR) {
    double2 a[UNR][R];
#pragma unroll
    for (int t = 0; t < UNR; ++t) {
      const int i = c + 128 * t + 2 * lane;
#pragma unroll
      for (int q = 0; q < R; ++q) a[t][q] = (i < m && i + 1 >= jq[q]) ? *reinterpret_cast<const double2*>(F + i + (int64_t)jq[q] * m) : make_double2(0.0, 0.0);
    }
    PLFEM_BWD_ROWS_CONSUME(a, c)
  }
#undef PLFEM_BWD_ROWS_CONSUME
  multi_reduce<V>(acc, lane);
  if (lane < V && oj < s2) A.xl[(2 * np + oj) * P + oidx % P] = acc[0];
}

// ---- kernels --------------------------------------------------------------------------------------------------------
// Levels with at most ROW_FORM_MAX_FRONTS fronts (all of them large): pure row form.  The other levels of the FORWARD
// sweep mix the two forms in ONE launch, per front: a front with more than MIX_BIG_S2 owned DOFs gets row-form
// workgroups (16 rows, 4 waves of 4 rows: four times the waves of the tile form, which is what a long front needs to
// keep enough loads in flight -- a level's few long fronts set its time: at C1 the forward level 7 took 25 us, 20 of
// them for 4 long fronts alone), every other front tile-form workgroups (64 rows, 4 waves splitting the columns).
// Measured and dropped: 8-wave tiles in the mixed launch (the leaf level twice as slow), the tile form for the short
// fronts of the backward mid levels (every level slower than the row form).
// The launch list comes as a kernel argument OF ITS OWN in front of the argument block: with kernel-argument preloading (Makefile:
// -amdgpu-kernarg-preload-count) it sits in scalar registers when a wave starts, and the load of the workgroup's job record --
// the head of every workgroup's chain of dependent memory round trips -- no longer waits for a load of the argument block.
// LD = 16-byte matrix loads in flight per lane and trip (8: twice the bytes of round 3's eight 8-byte loads, at the price of
// 16-28 more registers; 4: the same bytes in half the instructions); chosen per kernel form in sweeps() below
template <int P, int R, int LD>
__global__ __launch_bounds__(512) void k_fwd_rows(const SweepJob* __restrict__ blk, SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) double sv[];
  const SweepJob job = blk[blockIdx.x];
  fwd_rows_body<P, 8, R, (LD < R ? R : LD)>(A, job, sv);
}

template <int P, int R, int LD>
__global__ __launch_bounds__(512) void k_bwd_rows(const SweepJob* __restrict__ blk, SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) double sv[];
  const SweepJob job = blk[blockIdx.x];
  bwd_rows_body<P, 8, R, (LD < R ? R : LD)>(A, job, sv);
}

// forward, tile form only (levels without a long front: the mixed kernel's register budget costs occupancy there)
template <int P, int LD>
__global__ __launch_bounds__(256) void k_fwd(const SweepJob* __restrict__ blk, SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) double sv[];
  __shared__ __attribute__((aligned(16))) double red[4 * P * 64];
  const SweepJob job = blk[blockIdx.x];
  fwd_tile_body<P, 4, LD>(A, job, sv, red);
}

template <int P, int LD>
__global__ __launch_bounds__(256) void k_fwd_mix(const SweepJob* __restrict__ blk, SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) double sv[];
  __shared__ __attribute__((aligned(16))) double red[4 * P * 64];
  const SweepJob job = blk[blockIdx.x];
  if (job.rb & SWEEP_ROW_JOB_FLAG) fwd_rows_body<P, 4, 4, (LD < 4 ? 4 : LD)>(A, job, sv);
  else fwd_tile_body<P, 4, LD>(A, job, sv, red);
}

// backward, leaf level: tile form (leaf fronts have about as many owned rows as boundary columns)
template <int P, int LD>
__global__ __launch_bounds__(512) void k_bwd(const SweepJob* __restrict__ blk, SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) double sv[];
  __shared__ __attribute__((aligned(16))) double red[8 * P * 64];
  const SweepJob job = blk[blockIdx.x];
  bwd_tile_body<P, 8, LD>(A, job, sv, red);
}

// ---- global order <-> front order ---------------------------------------------------------------------------------
// fr[(npos[node] + c) P + u] = rhs[u ldx + c N + node]  (npos = front-order offset of the node's component 0;
// Dirichlet nodes: npos = -1, nothing to do)
template <int P>
__global__ __launch_bounds__(256) void k_permute_in(int64_t n2, int N, const int32_t* __restrict__ npos,
                                                    const double* __restrict__ rhs, int64_t ldx, double* __restrict__ fr) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= n2) return;
  const int c = g >= N, node = (int)(g - (int64_t)c * N);
  const int pos = npos[node];
  if (pos < 0) return;
#pragma unroll
  for (int u = 0; u < P; ++u) fr[((int64_t)pos + c) * P + u] = rhs[(int64_t)u * ldx + g];
}

// x[u ldx + g] = xl[(npos[node] + c) P + u], Dirichlet entries = 0 (the result needs no memset)
template <int P>
__global__ __launch_bounds__(256) void k_permute_out(int64_t n2, int N, const int32_t* __restrict__ npos,
                                                     const double* __restrict__ xl, double* __restrict__ x, int64_t ldx) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= n2) return;
  const int c = g >= N, node = (int)(g - (int64_t)c * N);
  const int pos = npos[node];
#pragma unroll
  for (int u = 0; u < P; ++u) x[(int64_t)u * ldx + g] = pos >= 0 ? xl[((int64_t)pos + c) * P + u] : 0.0;
}

template <int P>
void sweeps(plfem_ctx* c) {
  hipStream_t st = c->stream;
  SweepArgs A;
  A.sh = c->sh;
  A.dbg = c->debug_sweep_filter;
  A.cinv0 = c->d_cinv0; A.cinv1 = c->d_cinv1; A.prow = c->d_prow;
  A.front = c->d_front; A.dinv2 = reinterpret_cast<const double2*>(c->d_delta);
  A.fr = c->d_fvec; A.u0 = c->d_u0; A.u1 = c->d_u1; A.ys = c->d_fvec2; A.xl = c->d_xl;
  // 16-byte loads in flight per lane: tile forms / row forms (PLFEM_SWEEP_LD_TILE, PLFEM_SWEEP_LD_ROWS = 4 | 8: tuning aid)
  static const int ld_tile = getenv("PLFEM_SWEEP_LD_TILE") ? atoi(getenv("PLFEM_SWEEP_LD_TILE")) : PLFEM_SWEEP_LD_TILE_DEFAULT;
  static const int ld_rows = getenv("PLFEM_SWEEP_LD_ROWS") ? atoi(getenv("PLFEM_SWEEP_LD_ROWS")) : PLFEM_SWEEP_LD_ROWS_DEFAULT;
  double sweep_total = 0.0;                     // algorithmic bytes of one whole sweep (either direction)
  for (const LevelInfo& li : c->levels) sweep_total += li.sweep_bytes + 8.0 * (P - 1) * li.sweep_vec_doubles;
  // Kernel form by level: fwd_block_rows / bwd_block_rows (device.h); workgroups come from the compact launch
  // lists of the context (no empty workgroups, large fronts first).
  // live timing (plfem_profile_*): event records around single launches lengthen the sweep they sit in (~8 us each), so
  // a profiled step times whole sweeps and single launches in ALTERNATE block solves, never nested
  const bool time_launches = c->prof_on && (c->prof_toggle++ & 1);
  const int pid_fwd = time_launches ? -1 : prof_open(c, PLFEM_PROF_FWD_SWEEP, sweep_total);
  for (int lev = c->L; lev >= 0; --lev) {
    const LevelInfo& li = c->levels[lev];
    if (li.fwd_n == 0) continue;
    A.leaf_level = lev == c->L ? 1 : 0;
    const SweepJob* blk = c->d_blk + li.fwd_off;
    A.ldv = (li.max_s2 + 2) & ~1;                   // even: the row forms read the staged planes as double2
    const size_t lds = sizeof(double) * P * A.ldv;
    if (li.fwd_rows == 8) {
      if (ld_rows == 8) hipLaunchKernelGGL((k_fwd_rows<P, 1, 8>), dim3(li.fwd_n), dim3(512), lds, st, blk, A);
      else hipLaunchKernelGGL((k_fwd_rows<P, 1, 4>), dim3(li.fwd_n), dim3(512), lds, st, blk, A);
    } else if (li.fwd_rows == 16) {
      if (ld_rows == 8) hipLaunchKernelGGL((k_fwd_rows<P, 2, 8>), dim3(li.fwd_n), dim3(512), lds, st, blk, A);
      else hipLaunchKernelGGL((k_fwd_rows<P, 2, 4>), dim3(li.fwd_n), dim3(512), lds, st, blk, A);
    } else {
      // optional live timing of this kernel (bench.py roofline): HIP events on the launch stream
      const int pid = time_launches ? prof_open(c, PLFEM_PROF_KFWD, li.sweep_bytes + 8.0 * (P - 1) * li.sweep_vec_doubles) : -1;
      if (li.fwd_mixed) {
        if (ld_tile == 8) hipLaunchKernelGGL((k_fwd_mix<P, 8>), dim3(li.fwd_n), dim3(256), lds, st, blk, A);
        else hipLaunchKernelGGL((k_fwd_mix<P, 4>), dim3(li.fwd_n), dim3(256), lds, st, blk, A);
      } else {
        if (ld_tile == 8) hipLaunchKernelGGL((k_fwd<P, 8>), dim3(li.fwd_n), dim3(256), lds, st, blk, A);
        else hipLaunchKernelGGL((k_fwd<P, 4>), dim3(li.fwd_n), dim3(256), lds, st, blk, A);
      }
      prof_close(c, pid);
    }
  }
  prof_close(c, pid_fwd);
  const int pid_bwd = time_launches ? -1 : prof_open(c, PLFEM_PROF_BWD_SWEEP, sweep_total);
  for (int lev = 0; lev <= c->L; ++lev) {
    const LevelInfo& li = c->levels[lev];
    if (li.bwd_n == 0) continue;
    A.leaf_level = lev == c->L ? 1 : 0;
    const SweepJob* blk = c->d_blk + li.bwd_off;
    A.ldv = (li.max_m + 2) & ~1;
    const size_t lds = sizeof(double) * P * A.ldv;
    if (li.bwd_rows == 8) {        // few large fronts: one row per wave, most blocks
      if (ld_rows == 8) hipLaunchKernelGGL((k_bwd_rows<P, 1, 8>), dim3(li.bwd_n), dim3(512), lds, st, blk, A);
      else hipLaunchKernelGGL((k_bwd_rows<P, 1, 4>), dim3(li.bwd_n), dim3(512), lds, st, blk, A);
    } else if (li.bwd_rows == 16) {
      if (ld_rows == 8) hipLaunchKernelGGL((k_bwd_rows<P, 2, 8>), dim3(li.bwd_n), dim3(512), lds, st, blk, A);
      else hipLaunchKernelGGL((k_bwd_rows<P, 2, 4>), dim3(li.bwd_n), dim3(512), lds, st, blk, A);
    } else {                       // leaf level: tile form
      if (ld_tile == 8) hipLaunchKernelGGL((k_bwd<P, 8>), dim3(li.bwd_n), dim3(512), lds, st, blk, A);
      else hipLaunchKernelGGL((k_bwd<P, 4>), dim3(li.bwd_n), dim3(512), lds, st, blk, A);
    }
  }
  prof_close(c, pid_bwd);
}

}  // namespace

void launch_solve(plfem_ctx* c, const double* rhs, double* x) {
  const unsigned grid = (unsigned)((c->n2 + 255) / 256);
  hipLaunchKernelGGL(k_permute_in<1>, dim3(grid), dim3(256), 0, c->stream, c->n2, c->N, c->d_npos, rhs, c->n2, c->d_fvec);
  sweeps<1>(c);
  hipLaunchKernelGGL(k_permute_out<1>, dim3(grid), dim3(256), 0, c->stream, c->n2, c->N, c->d_npos, c->d_xl, x, c->n2);
}

// BLOCK_P right-hand sides given as columns (ldx apart); inside the sweeps the P values of a DOF are one 32-byte
// access.  rhs_in_front_order: the caller's previous kernel (k_block_scale) already left the right-hand side in the
// sweeps' layout (context buffer d_fvec)
void launch_solve_block(plfem_ctx* c, const double* rhs, double* x, int64_t ldx, bool rhs_in_front_order) {
  const unsigned grid = (unsigned)((c->n2 + 255) / 256);
  if (!rhs_in_front_order)
    hipLaunchKernelGGL(k_permute_in<BLOCK_P>, dim3(grid), dim3(256), 0, c->stream, c->n2, c->N, c->d_npos, rhs, ldx, c->d_fvec);
  sweeps<BLOCK_P>(c);
  if (x) hipLaunchKernelGGL(k_permute_out<BLOCK_P>, dim3(grid), dim3(256), 0, c->stream, c->n2, c->N, c->d_npos, c->d_xl, x, ldx);
}

}  // namespace plfem
