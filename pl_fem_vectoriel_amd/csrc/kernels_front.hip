// gfx950 kernels: multifrontal shift-invert operator on the nested-dissection front tree.
//
// Replaces splu((A - sigma B).tocsc()) and lu.solve() that the reference reaches through
// eigsh(..., sigma=...) (reference solver_fem.py:197 -> scipy arpack.py:915, 920-928).
//
// Every front F (order m, symmetric) is [F11 F12; F21 F22] with the first s2 DOFs fully summed.  Storage (symbolic.h):
// the columns of [F11; F21] (m x s2, leading dimension m) are kept, F22 -- the Schur complement, needed only until the
// parent has gathered it -- lives in one of two level arenas (b2 x b2, leading dimension b2), F12 is never formed
// (the trailing matrix is maintained in its lower triangle) and its place is taken, after the factorisation, by Z^T
// (s2 x b2, leading dimension s2, right behind [F11; F21]).  The factorisation is a right-looking block LDL^T of those s2 pivots, NB at a time:
//     F11 = L11 D L11^T,   L21 = F21 L11^-T D^-1,   S = F22 - L21 D L21^T   (S: gathered by the parent)
// and, interleaved with it, the explicit inverse of the unit lower triangular L11 (a triangular
// inverse is benign numerically, unlike F11^-1: the stiffness scale of near-degenerate elements
// stays in the diagonal D).  What is left in F for the solves:
//     lower(F11) = L11^-1, upper(F11) = L11^-T, F21 = Z = L21 L11^-1, F12 = Z^T, delta = D
// so both solve sweeps are batched dense column-times-vector products over contiguous columns
// (no triangular dependency chain inside a front).  A block Gauss-Jordan sweep (explicit F11^-1)
// was tried first and is unstable on meshes with sliver elements ("numerics").
#include <algorithm>

#include <type_traits>

#include "device.h"

namespace plfem {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

// the Schur complement block of front f: arena of its tree level (fronts are numbered in heap order)
__device__ __forceinline__ double* schur_of(int f, const int64_t* __restrict__ soff, double* __restrict__ schur, int64_t arena) {
  const int level = 31 - __clz(f + 1);
  return schur + (level & 1) * arena + soff[f];
}

// ------------------------------------------------------------------------------------------------
// front assembly
// ------------------------------------------------------------------------------------------------
// Leaf fronts: K_e = A_e - sigma B_e of the leaf's own elements, added element by element (fixed
// order, one block per front, 144 lanes per element) -- element-based multifrontal assembly.
template <int sh>
__global__ __launch_bounds__(256) void k_leaf_assemble(
    int first_front, int ne, double sigma, const int32_t* __restrict__ fs2, const int32_t* __restrict__ fm,
    const int64_t* __restrict__ foff, const int64_t* __restrict__ soff,
    const int64_t* __restrict__ fnode_ptr, const int32_t* __restrict__ fnodes,
    const int32_t* __restrict__ leaf_elem_ptr, const int32_t* __restrict__ leaf_elems,
    const int32_t* __restrict__ epos, const double* __restrict__ elem, double* __restrict__ front,
    double* __restrict__ schur, int64_t arena) {
  const int lf = blockIdx.x;
  const int f = first_front + lf;
  const int m = fm[f], s2 = fs2[f], b2 = m - s2;
  constexpr int dpn = sh + 1;                // unknowns per node: local DOF i of a front = (node i >> sh, component i & sh)
  double* F = front + foff[f];               // columns < s2
  double* S = schur_of(f, soff, schur, arena);   // columns >= s2, rows >= s2
  // entry (row i, column j) of the front; (i < s2 <= j is F12: not stored, the lower triangle is what everything reads)
  auto at = [&](int i, int j) -> int64_t { return j < s2 ? (int64_t)j * m + i : (i >= s2 ? -2 - ((int64_t)(j - s2) * b2 + (i - s2)) : -1); };
  const int tid = threadIdx.x;
  // (the leaf fronts are contiguous in memory, and so are their Schur complements in the arena: zeroed by two memsets
  // before this launch)
  const int32_t* fn = fnodes + fnode_ptr[f];
  for (int q = tid; q < (m >> sh); q += 256)
    if (fn[q] < 0) {   // padding node: unit pivot, no coupling
      for (int cc = 0; cc < dpn; ++cc) {
        const int d = dpn * q + cc;
        if (d < s2) F[(int64_t)d * m + d] = 1.0;
        else S[(int64_t)(d - s2) * b2 + (d - s2)] = 1.0;
      }
    }
  __syncthreads();
  constexpr int nd = 6 * dpn;                // DOFs of one element: 12 (Hx, Hy) or 6 (scalar)
  const int adof = tid / nd, bdof = tid % nd;
  const int a = adof >> sh, ca = adof & sh, b = bdof >> sh, cb = bdof & sh;
  const bool worker = tid < nd * nd;
  // scalar pencil: the AXX slot holds K - k0^2 M_eps and the MINV slot M (launch_element_matrices_scalar)
  const int blk_a = (ca == 0 && cb == 0) ? PLFEM_BLK_AXX : (ca == 0) ? PLFEM_BLK_AXY : (cb == 0) ? PLFEM_BLK_AYX : PLFEM_BLK_AYY;
  const bool diag_blk = ca == cb;
  // positions and values of EB elements are fetched together (they do not depend on F); only the additions into
  // F stay ordered element by element (two elements of a leaf may hit the same entry)
  constexpr int EB = 8;
  for (int q0 = leaf_elem_ptr[lf]; q0 < leaf_elem_ptr[lf + 1]; q0 += EB) {
    const int nb = min(EB, leaf_elem_ptr[lf + 1] - q0);
    int64_t dst[EB];
    double val[EB];
#pragma unroll
    for (int t = 0; t < EB; ++t) {
      dst[t] = -1;
      val[t] = 0.0;
      if (worker && t < nb) {
        const int e = leaf_elems[q0 + t];
        const int pa = epos[(size_t)(q0 + t) * 6 + a], pb = epos[(size_t)(q0 + t) * 6 + b];    // (leaf order: [ne][6])
        const double* em = elem + (size_t)e * ELEM_STRIDE + a * 6 + b;
        double v = em[blk_a * 36];
        if (diag_blk) v -= sigma * em[PLFEM_BLK_MINV * 36];
        if (pa >= 0 && pb >= 0) { dst[t] = at(dpn * pa + ca, dpn * pb + cb); val[t] = v; }
      }
    }
#pragma unroll
    for (int t = 0; t < EB; ++t) {
      if (t < nb) {
        if (dst[t] >= 0) F[dst[t]] += val[t];
        else if (dst[t] <= -2) S[-2 - dst[t]] += val[t];
        __syncthreads();
      }
    }
  }
}

// Internal fronts: gather formulation of the extend-add.  F[i,j] = S_child0[.,.] + S_child1[.,.]
// through the inverse index maps; every entry written exactly once (no zero fill, no atomics).
template <int sh>
__global__ __launch_bounds__(256) void k_front_gather(
    const int2* __restrict__ tiles, const int32_t* __restrict__ fs2, const int32_t* __restrict__ fm,
    const int64_t* __restrict__ foff, const int64_t* __restrict__ soff, const int64_t* __restrict__ fnode_ptr,
    const int32_t* __restrict__ fnodes, const int32_t* __restrict__ cinv0, const int32_t* __restrict__ cinv1,
    double* __restrict__ front, double* __restrict__ schur, int64_t arena) {
  const int2 job = tiles[blockIdx.x];                    // (front, tx | ty << 16): 64 x 64 entries
  const int f = job.x;
  const int m = fm[f], s2 = fs2[f], b2 = m - s2;
  constexpr int dpn = sh + 1;
  const int i = (job.y & 0xffff) * 64 + (threadIdx.x & 63);
  const int j0 = ((job.y >> 16) * 4 + (threadIdx.x >> 6)) * 16;
  if (i >= m || j0 >= m) return;
  if (j0 >= s2 && i < s2) return;                        // F12 is not stored (s2 and j0 are multiples of 16)
  const int64_t np = fnode_ptr[f];
  const int qi = i >> sh, ci = i & sh;
  const int c0i = cinv0[np + qi], c1i = cinv1[np + qi];
  const bool dummy_i = fnodes[np + qi] < 0;
  const int ch0 = 2 * f + 1, ch1 = 2 * f + 2;
  const int b0 = fm[ch0] - fs2[ch0], b1 = fm[ch1] - fs2[ch1];
  const double* S0 = schur_of(ch0, soff, schur, arena);   // the children's Schur complements (the other arena)
  const double* S1 = schur_of(ch1, soff, schur, arena);
  double* F = front + foff[f];
  double* S = schur_of(f, soff, schur, arena);
  // the column-node index pairs first (8 nodes at two DOFs per node, 16 at one), then all 32 child entries: two
  // memory round trips for 16 columns
  int c0j[16 >> sh], c1j[16 >> sh];
#pragma unroll
  for (int q = 0; q < (16 >> sh); ++q) {
    c0j[q] = cinv0[np + (j0 >> sh) + q];
    c1j[q] = cinv1[np + (j0 >> sh) + q];
  }
  double v[16];
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const int cj = jj & sh, q = jj >> sh;
    double a = 0.0, b = 0.0;
    // (row, column) = (larger, smaller) local index: only the lower triangle of a Schur complement is maintained
    if (c0i >= 0 && c0j[q] >= 0) {
      const int bi = dpn * c0i + ci, bj = dpn * c0j[q] + cj;
      a = S0[(int64_t)min(bi, bj) * b0 + max(bi, bj)];
    }
    if (c1i >= 0 && c1j[q] >= 0) {
      const int bi = dpn * c1i + ci, bj = dpn * c1j[q] + cj;
      b = S1[(int64_t)min(bi, bj) * b1 + max(bi, bj)];
    }
    v[jj] = a + b;
  }
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const int j = j0 + jj;
    const double val = (dummy_i && i == j) ? 1.0 : v[jj];
    if (j0 < s2) F[(int64_t)j * m + i] = val;
    else S[(int64_t)(j - s2) * b2 + (i - s2)] = val;
  }
}

// ------------------------------------------------------------------------------------------------
// block LDL^T of a level:  k_ldl_first_panel (pivot block + panel of step 0), then ONE launch per block step
// (k_ldl_update: update + inverse row + write-back, and the pivot block + panel of the next step)
// ------------------------------------------------------------------------------------------------
// The pivots are taken NODE PAIR by node pair: local DOFs (2q, 2q+1) of a front are the two field components of one P2
// node (scalar pencil: two neighbouring nodes), and a step of the LDL^T eliminates such a pair -- Bunch-Kaufman's choice
// between scalar and 2 x 2 pivots, restricted to the pair and without any permutation:
//     K = L D L^T,   L unit lower triangular,   D = blockdiag(D_q),   E_q = [[a, b], [b, c]] the pair's Schur complement,
//   * two scalar pivots in the static order (a, then c - b^2 / a): D_q diagonal, L[2q+1, 2q] = b / a;
//   * one genuine 2 x 2 pivot: D_q = E_q, L[2q+1, 2q] = 0, applied through the explicit inverse adj(E_q) / det.
// The pencil A - sigma B is indefinite (mid-spectrum shift), so a scalar Schur complement vanishes by chance now and then
// while its pair does not; a pair, on the other hand, can be nearly singular with healthy diagonal entries (a local
// resonance), and then only the scalar order keeps the error down.  Rule: whichever amplifies rounding errors less --
// (b / a)^2 for the scalar order against the condition number max|E|^2 / |det| of the explicit inverse.  L^-1 keeps its unit
// lower triangular form either way (what lower(F11) / upper(F11) store); D^-1 is kept as its diagonal plus ONE
// off-diagonal entry per DOF (the partner of local DOF i is i ^ 1; zero for scalar pivots).
// 1 / d by v_rcp_f64 + two Newton steps: full precision (~1 ulp, not correctly rounded), 5 instructions instead of
// the ~25 of the IEEE division sequence; for finite, normal d
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  return fma(fma(-d, r, 1.0), r, r);
}

// value of the same register in lane ^ 16 (v_permlane16_swap: rows 1 / 3 of the first operand trade places with rows
// 0 / 2 of the second)
__device__ __forceinline__ double lane_xor16(double v, bool odd_row) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(odd_row ? b[0] : b[1], odd_row ? a[0] : a[1]);
}

// One pivot pair E = [[a, b], [b, c]] (the same bits in every thread: all operands come from LDS broadcasts) and what
// this thread's row i does with it.  On entry y0, y1 = a[i][2q], a[i][2q+1] (zero for rows that are done), ra / rb = this
// thread's four entries of rows 2q / 2q+1 as staged.  The two pivot kinds are wave-uniform branches, not selects: a step of
// the pivot block costs what its waves issue (one wave per SIMD: ~5 clocks per instruction, 16 for v_rcp_f64) and a
// compare -> select pair on the path costs 56 clocks against 7 for a dependent v_fma_f64 (scripts/micro/f64_latency.hip).
// Vanishing pivots -- below thr = 1e-13 of the largest entry the pair's two rows had in the pivot block when the block
// step began -- are replaced statically by +-rep = 1e-8 of that entry and counted (plfem_solve_modes then repeats the
// eigen-solve with refinement inside the operator).  rep is the sqrt(eps)-sized replacement of static pivoting (Li &
// Demmel): the factor is that of a matrix 1e-8 away, the multipliers stay below 1e8, and ONE refinement pass brings K^-1 b
// back to full accuracy -- emulated on a pair made singular on purpose (scripts/singular_pair_emulation.py: 1.7e-10 before,
// 2.6e-13 after one pass; with rep = thr, as in round 3, 3.9e-6 and 1.9e-9).  The 2 x 2 form is guarded the same way:
// a = 0, |b| << thr, c ~ rmax passes the pivot rule (det = -b^2, p |det| > a^2 s^2) and used to be inverted with entries
// ~1 / b^2, up to inf, uncounted.
//   d0, d1, od: this pair's D^-1 (diagonal entries of rows 2q and 2q+1, off-diagonal entry), for the panel and the sweeps.
__device__ __forceinline__ void pair_step(double a, double b, double c, double thr, double rep, bool past_first, bool past_second,
                                          double y0, double y1, const double (&ra)[4], const double (&rb)[4],
                                          double (&v)[4], double& d0, double& d1, double& od, int& nper) {
  const double p = b * b;
  const double det = fma(a, c, -p) - fma(b, b, -p);          // Kahan: ac - b^2 to two roundings
  const double s = fmax(fmax(fabs(a), fabs(c)), fabs(b));
  if (p * fabs(det) <= (a * a) * (s * s)) {
    // scalar pivots a, then d2 = c - g b with g = b / a: two eliminations in the arithmetic of the sequential LDL^T
    // (row 2q+1 itself takes part in the first one: its x part becomes row 2q+1 of L^-1, entry 2q = -g)
    if (__builtin_expect(!(fabs(a) >= thr), 0)) {
      a = (a < 0.0) ? -rep : rep;
      nper += 1;
    }
    const double r1 = fast_rcp(a);
    const double g = b * r1;
    double d2 = fma(-g, b, c);
    if (__builtin_expect(!(fabs(d2) >= thr), 0)) {
      d2 = (d2 < 0.0) ? -rep : rep;
      nper += 1;
    }
    const double r2 = fast_rcp(d2);
    const double l0 = past_first ? y0 * r1 : 0.0;
    const double l1 = past_second ? fma(-l0, b, y1) * r2 : 0.0;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) v[cc] = fma(-l1, fma(-g, ra[cc], rb[cc]), fma(-l0, ra[cc], v[cc]));
    d0 = r1;
    d1 = r2;
    od = 0.0;
    asm volatile("" ::: "memory");                          // (keeps the arms apart: no if-conversion into selects)
  } else {
    // 2 x 2 pivot: |det| is a sizeable part of max|E|^2 here, the explicit inverse is benign -- unless the whole pair
    // vanishes against its rows (|det| < thr s: entries of the inverse beyond 1 / thr)
    double dt = det;
    if (__builtin_expect(!(fabs(dt) >= thr * s), 0)) {
      dt = (dt < 0.0) ? -rep * s : rep * s;
      nper += 1;
    }
    const double rd = fast_rcp(dt);
    const double x11 = c * rd, x12 = -b * rd, x22 = a * rd;
    const double l0 = past_second ? fma(y0, x11, y1 * x12) : 0.0;
    const double l1 = past_second ? fma(y0, x12, y1 * x22) : 0.0;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) v[cc] = fma(-l1, rb[cc], fma(-l0, ra[cc], v[cc]));
    d0 = x11;
    d1 = x22;
    od = x12;
    asm volatile("" ::: "memory");
  }
}

// The block LDL^T of the NB x NB pivot block by the 4 waves of a workgroup.  Thread (i, cg) keeps columns 4 cg .. 4 cg + 3
// of row i in registers -- entry c is a[i][c] while c belongs to a pair that is still to come and x[i][c] (X = L^-1, built
// by applying the row operations to the identity) afterwards.  A pair step publishes through LDS (double buffered, ONE
// workgroup barrier per step): rows 2q, 2q+1 (their x part, c <= 2q+1, is what gets read) and the RAW columns a[c][2q],
// a[c][2q+1] from thread row c -- behind the pair a step reads those, not row 2q's own copy a[2q][c] of them: the block is
// eliminated from its lower triangle alone, like the panel below it (the two copies of a 1e9-sized sliver entry differ by
// their rounding, and a mix of them costs two digits of K^-1).  Every thread then works out the pivot and the update of
// its row (pair_step, the same bits everywhere).
// Why four waves: what a step costs is its LDS instructions (a lone wave gets a fraction of the LDS rate; measured with
// scripts/micro/pivot_bench.hip: one wave holding 16 columns per lane spends 470 of a step's 1370 clocks issuing its 16
// ds_read_b128 and 460 on the then serial reciprocals); four waves read 4 columns each, in parallel.
// Inputs come straight from F; results go to LDS: tile[i][c] = X[i][c] (zero above the diagonal), sDd[i] / sDo[i] =
// diagonal / off-diagonal entry of D^-1 in row i (identity for the padding of a partial block).
struct PivotLds {
  double row[2][2][NB];       // x part of rows 2q, 2q+1 of the pair in flight (entries c <= 2q+1 are read)
  double col[2][2][NB];       // raw columns 2q, 2q+1: a[c][2q], a[c][2q+1] for every row c (entries c >= 2q are read)
  double rmax[NB];            // largest entry of every row pair of the block on arrival
};

// src: the block itself, src[i + c * ld] = a[i][c] -- the front in global memory (ld = m) or an LDS tile (column workgroups).
__device__ __forceinline__ void ldl_pivot_block(const double* src, int64_t ld, int nbk, int tid,
                                                double (*tile)[NB + 1], double* __restrict__ sDd, double* __restrict__ sDo,
                                                PivotLds& S, int32_t* __restrict__ counters) {
  static_assert(NB == 32, "thread map of ldl_pivot_block");
  const int i = tid & 31, cg = tid >> 5;
  double v[4];
  double rmax = 0.0;
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    const int c = 4 * cg + cc;
    v[cc] = (i < nbk && c < nbk) ? src[(int64_t)c * ld + i] : (i == c ? 1.0 : 0.0);
    rmax = fmax(rmax, fabs(v[cc]));
  }
  __syncthreads();                                             // (src may be the LDS tile that `tile` aliases: all loads first)
  // "vanishing" is judged against the pair's OWN two rows of the block as they arrive (not against the whole block: a
  // sliver element's 1e9-sized entries in the same block would declare a healthy pivot of 1e-4 a zero)
  tile[cg][i] = rmax;
  __syncthreads();
  if (tid < NB) {
    double r = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) r = fmax(r, fmax(tile[g][tid], tile[g][tid ^ 1]));
    S.rmax[tid] = r;
  }
  double dd = 1.0, od = 0.0;
  int nper = 0;
#pragma unroll
  for (int q = 0; q < NB / 2; ++q) {
    const int k = 2 * q, cgk = k >> 2, pc = k & 3, buf = q & 1;
    if (k >= nbk) break;                                       // (partial block: the rest is identity padding)
    if (cg == cgk) {                                           // columns k, k+1 leave a and become columns of x
      S.col[buf][0][i] = v[pc];
      S.col[buf][1][i] = v[pc + 1];
      v[pc] = (i == k) ? 1.0 : 0.0;
      v[pc + 1] = (i == k + 1) ? 1.0 : 0.0;
    }
    if ((i | 1) == k + 1) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) S.row[buf][i & 1][4 * cg + cc] = v[cc];
    }
    __syncthreads();
    const double a = S.col[buf][0][k], b = S.col[buf][0][k + 1], c = S.col[buf][1][k + 1];
    const double thr = fmax(1e-13 * S.rmax[k], 1e-300), rep = fmax(1e-8 * S.rmax[k], 1e-300);
    // this row's entries in the pair's columns (rows up to 2q are done; row 2q+1 still takes part in a scalar step)
    const double y0 = (i > k) ? S.col[buf][0][i] : 0.0, y1 = (i > k + 1) ? S.col[buf][1][i] : 0.0;
    double ra[4], rb[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int cidx = 4 * cg + cc;
      const bool behind = cidx > k + 1;
      ra[cc] = behind ? S.col[buf][0][cidx] : S.row[buf][0][cidx];
      rb[cc] = behind ? S.col[buf][1][cidx] : S.row[buf][1][cidx];
    }
    double d0, d1, o01;
    pair_step(a, b, c, thr, rep, i > k, i > k + 1, y0, y1, ra, rb, v, d0, d1, o01, nper);
    if ((i | 1) == k + 1) {
      dd = (i == k) ? d0 : d1;
      od = o01;
    }
  }
  __syncthreads();                                             // (tile was the scratch of the row maxima)
  // x[i][c] = v for c <= i, 0 above the diagonal
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    const int c = 4 * cg + cc;
    tile[i][c] = (c <= i) ? v[cc] : 0.0;
  }
  if (tid < NB) {
    sDd[i] = dd;
    sDo[i] = od;
  }
  if (tid == 0 && nper > 0 && counters) atomicAdd(&counters[0], nper);
}

// X = L^-1 of the pivot block -> dinv (D[r + c*NB] = X[r][c]), its D^-1 -> delta ((diagonal, off-diagonal) of every row)
__device__ __forceinline__ void store_pivot_results(int f, int k0, int nbk, int64_t np,
                                                    double* __restrict__ dinv, double* __restrict__ delta,
                                                    double (*tile)[NB + 1], const double* __restrict__ sDd,
                                                    const double* __restrict__ sDo) {
  double* D = dinv + (int64_t)f * NB * NB;
  for (int e = threadIdx.x; e < NB * NB; e += 256) D[e] = tile[e & (NB - 1)][e >> 5];
  if ((int)threadIdx.x < nbk)
    reinterpret_cast<double2*>(delta)[2 * np + k0 + threadIdx.x] = make_double2(sDd[threadIdx.x], sDo[threadIdx.x]);
}

// The panel of a block step, 16 rows of a wave: Y = R X^T (= R L^-T), W = Y D^-1 (= the L panel; D^-1 couples the two
// columns of a node pair: the partner column of an accumulator register sits in lane ^ 16).  W, Y are saved for the update
// kernel and W replaces R in F (a wave reads and writes its own rows only).
//   Y^T[c][i] = sum_j X[c][j] R[i][j] on v_mfma_f64_16x16x4_f64: A <- X (row c, k = j; xa0 / xa1 = rows lr / 16 + lr of the
//   LDS tile), B <- R^T (b[kk] = R[i][4 kk + lk]); the accumulator register r of lane l is Y[i][c = 16 tc + (l >> 4) + 4 r]:
//   128-B runs of W, Y and of the panel columns of F.
struct PanelOperands {
  double xa0[NB / 4], xa1[NB / 4], rdd[2][4], rdo[2][4];
  __device__ __forceinline__ void load(const double (*tile)[NB + 1], const double* sDd, const double* sDo, int nbk, int lr, int lk) {
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
      xa0[kk] = tile[lr][4 * kk + lk];
      xa1[kk] = tile[16 + lr][4 * kk + lk];
    }
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * tc + lk + 4 * r;
        rdd[tc][r] = (c < nbk) ? sDd[c] : 0.0;
        rdo[tc][r] = (c < nbk) ? sDo[c] : 0.0;
      }
  }
  // kcol: first column of the pivot block in the front; i: this lane's row
  __device__ __forceinline__ void rows(const double (&b)[NB / 4], int nbk, int kcol, int64_t m, int i, int lk,
                                       double* __restrict__ W, double* __restrict__ Y, double* __restrict__ F) const {
    v4d y0 = (v4d){0.0, 0.0, 0.0, 0.0}, y1 = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
      y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa0[kk], b[kk], y0, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa1[kk], b[kk], y1, 0, 0, 0);
    }
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * tc + lk + 4 * r;
        const double y = tc == 0 ? y0[r] : y1[r];
        const double w = fma(y, rdd[tc][r], lane_xor16(y, lk & 1) * rdo[tc][r]);   // (the pair's other column: lane ^ 16)
        W[(int64_t)c * m + i] = w;
        Y[(int64_t)c * m + i] = y;
        if (c < nbk) F[(int64_t)(kcol + c) * m + i] = w;
      }
  }
};

// First launch of a LEVEL: pivot block + panel of block step 0 (every later step's pivot block and panel come out of the
// update launch of the step before it, ldl_column_block).  Every panel workgroup (64 rows below the pivot block, 4 waves
// of 16 rows) factorises the pivot block itself -- the same arithmetic in every workgroup, so the same bits -- while its
// panel operands are in flight; workgroup 0 of the front stores X in dinv and D^-1 in delta.  Nobody writes the pivot block
// in this launch (it is written back into F by the next one).
// (no mirrored copy L^T in the rows of the pivot block: those entries are overwritten -- by the triangular-inverse
// update inside F11, by Z^T in F12 -- before anything reads them)
__global__ __launch_bounds__(256) void k_ldl_first_panel(const FrontRec* __restrict__ frec, double* __restrict__ front,
                                                         double* __restrict__ dinv, double* __restrict__ delta,
                                                         double* __restrict__ wbuf, double* __restrict__ rbuf,
                                                         int32_t* __restrict__ counters) {
  const FrontRec R = frec[blockIdx.x];                       // (one record instead of front number -> four lookups)
  const int f = R.f;
  const int s2 = R.s2;
  const int nbk = min(NB, s2);
  const int m = R.m;
  double* F = front + R.foff;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // panel workgroup bx of this front takes the 64-row chunks bx, bx + n_pan, ... below the pivot block (n_pan workgroups
  // per front: all chunks in parallel at the top of the tree, where the step is latency; at most two workgroups per
  // front where a level has hundreds of fronts and every extra workgroup is one more redundant pivot factorisation)
  const int bx = blockIdx.y;
  const int n_pan = gridDim.y;
  const int t0 = nbk;
  if (bx > 0 && t0 + bx * 64 >= m) return;
  __shared__ __attribute__((aligned(16))) PivotLds piv;
  __shared__ double tile[NB][NB + 1];
  __shared__ double sDd[NB], sDo[NB];
  // this wave's 16 panel rows of the first chunk, requested before the pivot work
  const int lr = lane & 15, lk = lane >> 4;
  double b[NB / 4];
  {
    const int ibase = t0 + bx * 64 + 16 * wave;
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
      const int jx = 4 * kk + lk;
      b[kk] = (ibase < m && jx < nbk) ? F[(int64_t)jx * m + ibase + lr] : 0.0;
    }
  }
  ldl_pivot_block(F, m, nbk, threadIdx.x, tile, sDd, sDo, piv, bx == 0 ? counters : nullptr);
  __syncthreads();
  if (bx == 0) store_pivot_results(f, 0, nbk, R.np, dinv, delta, tile, sDd, sDo);
  double* W = wbuf + 2 * R.np * NB;
  double* Y = rbuf + 2 * R.np * NB;
  PanelOperands po;
  po.load(tile, sDd, sDo, nbk, lr, lk);
  for (int ch = bx; t0 + ch * 64 < m; ch += n_pan) {
    const int ibase = t0 + ch * 64 + 16 * wave;
    if (ibase >= m) break;                                 // m is a multiple of 16: the wave's 16 rows are all valid
    const int i = ibase + lr;
    if (ch != bx) {
#pragma unroll
      for (int kk = 0; kk < NB / 4; ++kk) {
        const int jx = 4 * kk + lk;
        b[kk] = (jx < nbk) ? F[(int64_t)jx * m + i] : 0.0;
      }
    }
    po.rows(b, nbk, 0, m, i, lk, W, Y, F);
  }
}

// Triangular-inverse update: with X<k the inverse of the leading k0 x k0 block of L11,
//   X[k, <k] = -X[k,k] * ( L[k, <k] * X<k ).
// One workgroup per 16 columns c, both products on v_mfma_f64_16x16x4_f64:
//   T (32 x 16) = L[k, j>=c0] (A: rows k of the panel columns j in F, contiguous in q) * X<k[j, c] (B: upper mirror,
//   contiguous in c);
//   the accumulator register r of lane (lr, lk) holds T[lk + 4 r][lr], which is exactly the B operand
//   of k-step r of the second product  Xnew = -X[k,k] * T  -- no cross-lane movement.
// During the factorisation X lives in the UPPER triangle of F11 only (X^T: all that later block rows and k_form_z read):
// the lower triangle keeps L, which the other workgroups of this block row are still reading, and receives X at the very
// end (k_mirror_x) for the forward sweep.
// The 4 waves split the j range (late steps of a long front: k0 / 32 dependent memory round trips for one wave) and
// their partial T meet in LDS in a fixed order; wave 0 finishes.
__device__ __forceinline__ void ldl_invrow_block(int f, int bx, int kb, const int32_t* __restrict__ fs2,
                                                 const int32_t* __restrict__ fm, const int64_t* __restrict__ foff,
                                                 double* __restrict__ front, const double* __restrict__ dinv,
                                                 double* __restrict__ part /* LDS [3][8][64] */) {
  const int s2 = fs2[f];
  const int k0 = kb * NB;
  if (k0 >= s2 || k0 == 0) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = bx * 16;
  if (c0 >= k0) return;
  const int nbk = min(NB, s2 - k0);
  const int m = fm[f];
  double* F = front + foff[f];
  const double* D = dinv + (int64_t)f * NB * NB;       // D[r + q*NB] = X[k,k][r][q]
  const int lr = lane & 15, lk = lane >> 4;
  const int c = c0 + lr;
  v4d t0 = (v4d){0.0, 0.0, 0.0, 0.0}, t1 = (v4d){0.0, 0.0, 0.0, 0.0};
  // k0 - c0 is a multiple of 16.  G groups of four k-steps (12 G loads) are requested before their MFMAs: two
  // groups per trip while they last -- a long row (late steps of a large front) is a chain of memory round trips
  // (four groups per trip measured the same)
  auto groups = [&](int j0, auto G_) {
    constexpr int G = decltype(G_)::value;
    double a0[4 * G], a1[4 * G], b[4 * G];
#pragma unroll
    for (int t = 0; t < 4 * G; ++t) {
      const int j = j0 + 4 * t + lk;
      a0[t] = F[(int64_t)j * m + k0 + lr];                        // L[k0 + lr, j] (rows past nbk belong to F21)
      a1[t] = (nbk > 16) ? F[(int64_t)j * m + k0 + 16 + lr] : 0.0;
      b[t] = (j >= c) ? F[(int64_t)j * m + c] : 0.0;            // X<k[j, c] (unit diagonal stored)
    }
#pragma unroll
    for (int t = 0; t < 4 * G; ++t) {
      t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[t], b[t], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t], b[t], t1, 0, 0, 0);
    }
  };
  const int ngroups = (k0 - c0) / 16;                       // groups of 16 columns j; wave w takes [w G / 4, (w + 1) G / 4)
  int j0 = c0 + 16 * (wave * ngroups / 4);
  const int j_end = c0 + 16 * ((wave + 1) * ngroups / 4);
  for (; j0 + 16 < j_end; j0 += 32) groups(j0, std::integral_constant<int, 2>());
  if (j0 < j_end) groups(j0, std::integral_constant<int, 1>());
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      part[((wave - 1) * 8 + r) * 64 + lane] = t0[r];
      part[((wave - 1) * 8 + 4 + r) * 64 + lane] = t1[r];
    }
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      t0[r] += part[(w * 8 + r) * 64 + lane];
      t1[r] += part[(w * 8 + 4 + r) * 64 + lane];
    }
  v4d x0 = (v4d){0.0, 0.0, 0.0, 0.0}, x1 = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const int q = 4 * kk + lk;
    const double a0 = -D[(int64_t)q * NB + lr];          // -X[k,k][lr][q]
    const double a1 = -D[(int64_t)q * NB + 16 + lr];
    const double b = kk < 4 ? t0[kk & 3] : t1[kk & 3];   // T[q][c]
    x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, x0, 0, 0, 0);
    x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, x1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int r0 = lk + 4 * r, r1 = 16 + lk + 4 * r;
    if (r0 < nbk) F[(int64_t)(k0 + r0) * m + c] = x0[r];
    if (r1 < nbk) F[(int64_t)(k0 + r1) * m + c] = x1[r];
  }
}

// Trailing update: F[i,j] -= sum_c W[i,c] Y[j,c] for i, j >= k0 + nbk, one 32x32 tile per wave as 2x2
// v_mfma_f64_16x16x4_f64 tiles.  MFMA operand map (gfx950): A[row = l&15][k = l>>4],
// B[k = l>>4][col = l&15], D[row = (l>>4) + 4 r][col = l&15].  With A <- Y rows (j) and B <- W rows
// (i) the accumulator register r of lane l is F[i0 + (l&15), j0 + (l>>4) + 4 r]: 128-B runs.
// The read-modify-write of the trailing matrix is what bounds this kernel (2 flop/B at rank 32), so block
// steps are paired and the matrix is rewritten once per PAIR:
//   MODE 0, after an even step: a front that has a next step leaves its trailing matrix alone (the columns of its next
//           pivot block -- all the next step reads -- are the column workgroups' job, ldl_column_block); a front on its
//           last step updates everything.
//   MODE 1, after an odd step: rank-64 update of the trailing matrix with the panels of both steps
//           (W0/Y0: the even step's, still valid for every row below the odd pivot block), again without the columns
//           of the next pivot block.
// The panel code stores all NB columns of W and Y (zeros past nbk): fixed trip counts, and every operand
// of a rank-32 batch (32 loads) plus the 16 loads of the tile itself are requested before the first MFMA.
template <int MODE>
__device__ __forceinline__ void ldl_update_tile(const int2 job, int kb, const int32_t* __restrict__ fs2,
                                                const int32_t* __restrict__ fm, const int64_t* __restrict__ foff,
                                                const int64_t* __restrict__ soff, const int64_t* __restrict__ fnode_ptr,
                                                double* __restrict__ front, double* __restrict__ schur, int64_t arena,
                                                const double* __restrict__ wbuf, const double* __restrict__ rbuf,
                                                const double* __restrict__ wbuf_prev, const double* __restrict__ rbuf_prev) {
  // job = (front, tx | ty << 16): a 64 x 64 block of the trailing matrix
  const int f = job.x;
  const int s2 = fs2[f];
  const int k0 = kb * NB;
  if (k0 >= s2) return;
  const int m = fm[f];
  const int nbk = min(NB, s2 - k0);
  const int t0 = k0 + nbk;                 // first trailing index (multiple of 16)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i0 = t0 + ((job.y & 0xffff) * 2 + (wave & 1)) * 32;
  const int j0 = t0 + ((job.y >> 16) * 2 + (wave >> 1)) * 32;
  if (i0 >= m || j0 >= m) return;
  // The trailing matrix is symmetric and every later reader takes its lower triangle (rows >= columns; pivot
  // blocks are diagonal 32 x 32 blocks, panels lie below them, the extend-add swaps its indices), so the blocks
  // strictly above the diagonal are never updated
  if (i0 < j0) return;
  // A next step exists: the columns of its pivot block (32, or 16 if it is the front's last, partial block) belong to the
  // column workgroups of this launch (update + pivot + panel in one go).  MODE 0: everything to their right is updated
  // once, by the rank-64 pass after that step.
  const int ncol = (t0 < s2 && j0 == t0) ? min(NB, s2 - t0) : 0;
  if ((MODE == 0 && t0 < s2) || ncol == NB) return;
  double* F = front + foff[f];                          // columns < s2 (all rows)
  double* S = schur_of(f, soff, schur, arena);          // columns >= s2, rows >= s2
  const int b2 = m - s2;
  const int lr = lane & 15, lk = lane >> 4;
  const bool iv1 = i0 + 16 < m, jv1 = j0 + 16 < m;
  // a 16 x 16 quadrant lies on one side of s2 in either direction (s2, i0, j0 are multiples of 16); rows < s2 <= columns
  // is F12, which is not stored (it can only come up in the quadrant above the diagonal of a diagonal tile)
  auto quad_ok = [&](int tj, int ti) { return (tj == 0 || jv1) && (ti == 0 || iv1) && !(j0 + 16 * tj >= s2 && i0 + 16 * ti < s2); };
  auto addr = [&](int tj, int ti, int r) -> double* {
    const int j = j0 + 16 * tj + lk + 4 * r, i = i0 + 16 * ti + lr;
    return j0 + 16 * tj < s2 ? F + (int64_t)j * m + i : S + (int64_t)(j - s2) * b2 + (i - s2);
  };
  double fv[2][2][4];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) fv[tj][ti][r] = quad_ok(tj, ti) ? *addr(tj, ti, r) : 0.0;
  v4d acc[2][2];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) acc[tj][ti] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int pass = 0; pass <= MODE; ++pass) {
    const double* W = (pass == 0 ? wbuf : wbuf_prev) + 2 * fnode_ptr[f] * NB;
    const double* Y = (pass == 0 ? rbuf : rbuf_prev) + 2 * fnode_ptr[f] * NB;
    double a0[NB / 4], a1[NB / 4], b0[NB / 4], b1[NB / 4];
#pragma unroll
    for (int it = 0; it < NB / 4; ++it) {
      const int64_t col = (int64_t)(4 * it + lk) * m;
      a0[it] = Y[col + j0 + lr];
      a1[it] = jv1 ? Y[col + j0 + 16 + lr] : 0.0;
      b0[it] = W[col + i0 + lr];
      b1[it] = iv1 ? W[col + i0 + 16 + lr] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < NB / 4; ++it) {
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[it], b0[it], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[it], b1[it], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[it], b0[it], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[it], b1[it], acc[1][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      if (!quad_ok(tj, ti)) continue;
      if (16 * tj < ncol) continue;                      // (ncol = 16: the other 16 columns lie past s2, in S)
#pragma unroll
      for (int r = 0; r < 4; ++r) *addr(tj, ti, r) = fv[tj][ti][r] - acc[tj][ti][r];
    }
  }
}

// Column workgroups of an update launch: everything the front's NEXT block step needs, so that the next launch can be that
// step's update right away (one launch per block step; the first step of a level has k_ldl_first_panel).  With t0 the first
// index behind this step's pivot block and nbn the order of the next one:
//   1. the next pivot block -- rows and columns [t0, t0 + nbn) of the trailing matrix -- receives this step's update
//      (wave w = quadrant (w & 1, w >> 1); rank 32 after an even step, rank 64 with the panels of both steps after an odd
//      one, the arithmetic of ldl_update_tile), goes to LDS instead of back to F and is factorised on the spot.  Every
//      column workgroup of the front does this for itself (same arithmetic, same bits); workgroup 0 stores X -> dinv of the
//      next step's parity and D^-1 -> delta.  The chain of 16 dependent pair steps (7 us) runs beside the trailing update
//      of the other workgroups.
//   2. rows below it, 64 per workgroup and trip (chunks bx, bx + n_pan, ...), 16 per wave: columns [t0, t0 + nbn) get the
//      same update in registers -- the accumulator layout of the update IS the B operand layout of the panel product
//      (register r of quadrant tj = column 4 (r + 4 tj) + lk) -- and go through the panel code at once: W, Y into the
//      buffers of the next step, W into F.  The first chunk's update is computed before the pivot chain starts.
template <int MODE>
__device__ __forceinline__ void ldl_column_block(const FrontRec& R, int bx, int n_pan, int kb, double* __restrict__ front,
                                                 double* __restrict__ dinv_next, double* __restrict__ delta,
                                                 const double* __restrict__ wbuf, const double* __restrict__ rbuf,
                                                 const double* __restrict__ wbuf_prev, const double* __restrict__ rbuf_prev,
                                                 double* __restrict__ wbuf_next, double* __restrict__ rbuf_next,
                                                 int32_t* __restrict__ counters) {
  const int f = R.f, s2 = R.s2;
  const int k0 = kb * NB;
  const int t0 = k0 + NB;                      // (a front with a next step has a full block now)
  if (t0 >= s2) return;
  const int nbn = min(NB, s2 - t0);
  const int m = R.m;
  const int t1 = t0 + nbn;                     // first row below the next pivot block
  if (bx > 0 && t1 + bx * 64 >= m) return;
  double* F = front + R.foff;
  __shared__ __attribute__((aligned(16))) PivotLds piv;
  __shared__ double tile[NB][NB + 1];
  __shared__ double sDd[NB], sDo[NB];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const int64_t pbase = 2 * R.np * NB;
  const bool full = nbn > 16;
  // One pass over the panels of this step (MODE 1: and of the step before) for 16 rows of this wave -- columns
  // [t0, t0 + nbn) of rows ibase .. after the update, as B operand bq of the panel product -- and, with QUAD, for this wave's
  // quadrant of the next pivot block as well.  The A operands (the rows of Y that belong to the next pivot block's columns)
  // are asked for pass by pass and not kept: registers are what bounds the workgroups per CU of this kernel.
  const int qi = wave & 1, qj = wave >> 1;
  const bool quad_on = 16 * qi < nbn && 16 * qj < nbn;
  auto strip = [&](int ibase, double (&bq)[NB / 4], auto QUAD_, int m, int64_t pbase) {
    constexpr bool QUAD = decltype(QUAD_)::value;
    const bool rows_on = ibase < m;
    const int i = (rows_on ? ibase : t0) + lr;           // (no rows: a valid address, the result is dropped)
    double fv[2][4], qv[4];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) fv[tj][r] = (tj == 0 || full) ? F[(int64_t)(t0 + 16 * tj + lk + 4 * r) * m + i] : 0.0;
    if (QUAD) {
#pragma unroll
      for (int r = 0; r < 4; ++r) qv[r] = quad_on ? F[(int64_t)(t0 + 16 * qj + lk + 4 * r) * m + (t0 + 16 * qi + lr)] : 0.0;
    }
    v4d acc[2] = {(v4d){0.0, 0.0, 0.0, 0.0}, (v4d){0.0, 0.0, 0.0, 0.0}};
    v4d qacc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int pass = 0; pass <= MODE; ++pass) {
      const double* W = (pass == 0 ? wbuf : wbuf_prev) + pbase;
      const double* Y = (pass == 0 ? rbuf : rbuf_prev) + pbase;
      double ya[2][NB / 4], b[NB / 4], qb[NB / 4];
#pragma unroll
      for (int it = 0; it < NB / 4; ++it) {
        const int64_t col = (int64_t)(4 * it + lk) * m;
        ya[0][it] = Y[col + t0 + lr];
        ya[1][it] = full ? Y[col + t0 + 16 + lr] : 0.0;
        b[it] = W[col + i];
        if (QUAD) qb[it] = W[col + t0 + 16 * qi + lr];
      }
#pragma unroll
      for (int it = 0; it < NB / 4; ++it) {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[0][it], b[it], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[1][it], b[it], acc[1], 0, 0, 0);
        if (QUAD) qacc = __builtin_amdgcn_mfma_f64_16x16x4f64(qj == 0 ? ya[0][it] : ya[1][it], qb[it], qacc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) bq[r + 4 * tj] = (tj == 0 || full) ? fv[tj][r] - acc[tj][r] : 0.0;
    if (QUAD && quad_on) {
      // tile[c][i] = a[i][c]: column-major with leading dimension NB + 1, what ldl_pivot_block reads as src[i + c ld]
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[16 * qj + lk + 4 * r][16 * qi + lr] = qv[r] - qacc[r];
    }
  };
  double bq[NB / 4];
  const int ibase0 = t1 + bx * 64 + 16 * wave;
  // 1. the next pivot block, and the first chunk's rows on the way
  strip(ibase0, bq, std::true_type(), m, pbase);
  __syncthreads();
  ldl_pivot_block(&tile[0][0], NB + 1, nbn, threadIdx.x, tile, sDd, sDo, piv, bx == 0 ? counters : nullptr);
  __syncthreads();
  if (bx == 0) store_pivot_results(f, t0, nbn, R.np, dinv_next, delta, tile, sDd, sDo);
  // 2. the panel of the next step (X and D^-1 are read from LDS again for every further chunk instead of being carried
  // through its update: 64 registers)
  double* Wn = wbuf_next + pbase;
  double* Yn = rbuf_next + pbase;
  if (ibase0 >= m) return;                                 // m is a multiple of 16: a wave's 16 rows are all valid
  {
    PanelOperands po;
    po.load(tile, sDd, sDo, nbn, lr, lk);
    po.rows(bq, nbn, t0, m, ibase0 + lr, lk, Wn, Yn, F);
  }
  for (int ibase = ibase0 + n_pan * 64; ibase < m; ibase += n_pan * 64) {
    // (m and the panel offset pass through an empty asm in every trip: as loop invariants the 40-odd addresses derived
    // from them would be computed once and held in registers for the whole loop)
    int ml = m;
    int64_t pl = pbase;
    asm volatile("" : "+s"(ml), "+s"(pl) : : "memory");
    strip(ibase, bq, std::false_type(), ml, pl);
    asm volatile("" ::: "memory");
    PanelOperands po;
    po.load(tile, sDd, sDo, nbn, lr, lk);
    po.rows(bq, nbn, t0, ml, ibase + lr, lk, wbuf_next + pl, rbuf_next + pl, F);
  }
}

// The launch of a block step (after the level's first panel there is ONE per step).  Workgroups, in this order:
//   [0, n_look n_pan)  column workgroups of the fronts that have a next step (a prefix of the launch order): next pivot block
//                      + next panel (ldl_column_block) -- the longest chain of the launch, so they start first;
//   un                 trailing update, the step's tile list;
//   nact (n_inv + 1)   per active front the triangular-inverse update of the block row (n_inv workgroups) plus one workgroup
//                      that writes the pivot block of this step (lower X, upper X^T, from dinv) back into F
// -- all independent of one another: the update touches rows / columns behind the next pivot block's columns, the column
// workgroups those columns, the inverse update reads rows of this step's pivot block in earlier columns and writes their
// mirror image.  W / Y of three consecutive steps are in flight (read: this step's and, rank 64, the one before;
// written: the next step's), X of the pivot blocks of two.
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_ldl_update(
    int un, int n_inv, int n_look, int n_pan, const int2* __restrict__ tiles, const int32_t* __restrict__ forder,
    const FrontRec* __restrict__ frec, int kb,
    const int32_t* __restrict__ fs2, const int32_t* __restrict__ fm, const int64_t* __restrict__ foff,
    const int64_t* __restrict__ soff, const int64_t* __restrict__ fnode_ptr, double* __restrict__ front,
    double* __restrict__ schur, int64_t arena, const double* __restrict__ dinv, double* __restrict__ dinv_next,
    double* __restrict__ delta, const double* __restrict__ wbuf, const double* __restrict__ rbuf,
    const double* __restrict__ wbuf_prev, const double* __restrict__ rbuf_prev, double* __restrict__ wbuf_next,
    double* __restrict__ rbuf_next, int32_t* __restrict__ counters) {
  int bid = blockIdx.x;
  if (bid < n_look * n_pan) {
    const FrontRec R = frec[bid / n_pan];
    ldl_column_block<MODE>(R, bid % n_pan, n_pan, kb, front, dinv_next, delta, wbuf, rbuf, wbuf_prev, rbuf_prev, wbuf_next,
                           rbuf_next, counters);
    return;
  }
  bid -= n_look * n_pan;
  if (bid < un) {
    ldl_update_tile<MODE>(tiles[bid], kb, fs2, fm, foff, soff, fnode_ptr, front, schur, arena, wbuf, rbuf, wbuf_prev, rbuf_prev);
    return;
  }
  bid -= un;
  const int f = forder[bid / (n_inv + 1)];
  const int sub = bid % (n_inv + 1);
  if (sub < n_inv) {
    __shared__ double part[3 * 8 * 64];
    ldl_invrow_block(f, sub, kb, fs2, fm, foff, front, dinv, part);
    return;
  }
  const int s2 = fs2[f];
  const int k0 = kb * NB;
  if (k0 >= s2) return;
  const int nbk = min(NB, s2 - k0);
  const int m = fm[f];
  double* F = front + foff[f];
  const double* D = dinv + (int64_t)f * NB * NB;          // D[r + c*NB] = X[r][c], zero above the diagonal
  for (int q = threadIdx.x; q < NB * NB; q += 256) {
    const int i = q & (NB - 1), c = q >> 5;
    if (i < nbk && c < nbk) F[(int64_t)(k0 + c) * m + (k0 + i)] = (i >= c) ? D[i + c * NB] : D[c + i * NB];
  }
}

// ---- after the LDL^T of a front: what only the solve sweeps read --------------------------------------------------------
// ONE launch (round 4; rounds 2-3: k_form_z, k_mirror_z, k_mirror_x: 0.40 ms at C1, the fronts' 0.3 GB of Z written and read
// once more than necessary) with two kinds of workgroups:
//  * Z = L21 L11^-1 (with Z in place of L21 the forward sweep of a front is ONE product [L11^-1; Z] r and the backward sweep
//    ONE product [L11^-1; -Z]^T [D^-1 y; x_b]), written IN PLACE over L21 and, transposed (s2 x b2, leading dimension s2),
//    behind [F11; F21].  In place works because a workgroup owns a 64-row block of Z and takes its columns in ASCENDING
//    order: Z[:, c] = sum_{j >= c} L21[:, j] Linv[j, c] needs the columns j >= c of L21 only, so what has been overwritten
//    (columns < c) is never read again; inside a step of 64 columns the four waves (2 row halves x 2 column halves, a 32 x 32
//    tile each on v_mfma_f64_16x16x4_f64) finish their reads before a barrier and write behind it.  The tile goes through
//    LDS on its way into F21, so that both copies leave in runs of 128 B or more.
//  * lower(F11) = L11^-1 from its transpose in the upper triangle (the factorisation keeps X^T there): the block row k (rows
//    [k0, k0 + 32), columns < k0) of one front per workgroup, 32 x 32 tiles through LDS.  job = (front, kb).
// The Z workgroups read the upper triangle of F11, the others write the lower one: independent.
__device__ __forceinline__ void mirror_x_block(const int2 job, const int32_t* __restrict__ fs2, const int32_t* __restrict__ fm,
                                               const int64_t* __restrict__ foff, double* __restrict__ front,
                                               double (*tile)[33]) {
  const int f = job.x, k0 = job.y * NB;
  const int m = fm[f], s2 = fs2[f];
  const int nbk = min(NB, s2 - k0);
  double* F = front + foff[f];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int c0 = 0; c0 < k0; c0 += 32) {
    for (int yy = ty; yy < 32; yy += 8)                     // X^T: column k0 + yy, rows c0 + tx (contiguous)
      tile[yy][tx] = (yy < nbk) ? F[(int64_t)(k0 + yy) * m + c0 + tx] : 0.0;
    __syncthreads();
    for (int yy = ty; yy < 32; yy += 8)                     // X: column c0 + yy, rows k0 + tx (contiguous)
      if (tx < nbk) F[(int64_t)(c0 + yy) * m + k0 + tx] = tile[tx][yy];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_form_z_mirror(const int2* __restrict__ jobs, int nz, const int32_t* __restrict__ fs2,
                                                       const int32_t* __restrict__ fm, const int64_t* __restrict__ foff,
                                                       double* __restrict__ front) {
  __shared__ double lds[4][32][33];
  const int2 job = jobs[blockIdx.x];
  if ((int)blockIdx.x >= nz) {
    mirror_x_block(job, fs2, fm, foff, front, lds[0]);
    return;
  }
  const int f = job.x;                                   // (front, 64-row block of Z)
  const int m = fm[f], s2 = fs2[f];
  const int b2 = m - s2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b0 = (job.y * 2 + (wave & 1)) * 32;          // rows of Z (boundary DOFs) of this wave
  const bool rows_on = b0 < b2;
  double* F = front + foff[f];
  double* FZ = F + (int64_t)m * s2;
  const int lr = lane & 15, lk = lane >> 4;
  const bool bv1 = b0 + 16 < b2;
  double (*tile)[33] = lds[wave];
  for (int cstep = 0; cstep < s2; cstep += 64) {
    const int c0 = cstep + 32 * (wave >> 1);             // columns of Z (owned DOFs) of this wave in this step
    const bool on = rows_on && c0 < s2;
    const bool cv1 = c0 + 16 < s2;
    v4d acc[2][2];
    for (int tb = 0; tb < 2; ++tb)
      for (int tc = 0; tc < 2; ++tc) acc[tb][tc] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (on) {
      // Z[b, c] = sum_{j >= c} L21[b, j] Linv[j, c];  A <- L21 rows (contiguous in b), B <- Linv via the upper mirror
      // s2 - c0 is a multiple of 16: four k-steps (16 loads) are requested before their MFMAs
      const int cA = c0 + lr, cB = c0 + 16 + lr;
      for (int j0 = c0; j0 < s2; j0 += 16) {
        double a0[4], a1[4], x0[4], x1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int j = j0 + 4 * t + lk;
          a0[t] = F[(int64_t)j * m + s2 + b0 + lr];
          a1[t] = bv1 ? F[(int64_t)j * m + s2 + b0 + 16 + lr] : 0.0;
          x0[t] = (j >= cA) ? F[(int64_t)j * m + cA] : 0.0;
          x1[t] = (cv1 && j >= cB) ? F[(int64_t)j * m + cB] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[t], x0[t], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[t], x1[t], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t], x0[t], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t], x1[t], acc[1][1], 0, 0, 0);
        }
      }
      // D[row = b (lk + 4r)][col = c (lr)]: the tile into LDS as tile[c][b] for the in-place copy, and Z^T[c, b] at
      // FZ[c + b s2] straight from the registers (lanes lr contiguous)
      for (int tb = 0; tb < 2; ++tb)
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int bb = 16 * tb + lk + 4 * r, cc = 16 * tc + lr;
            tile[cc][bb] = acc[tb][tc][r];
            if ((tb == 0 || bv1) && (tc == 0 || cv1)) FZ[(int64_t)(b0 + bb) * s2 + (c0 + cc)] = acc[tb][tc][r];
          }
    }
    __syncthreads();          // every wave of the block has read the columns of L21 it needs: now they may be overwritten
    if (on) {
      // Z[b, c] -> F[c m + s2 + b]: 32 consecutive rows b per column c (lane = b, two columns per pass)
      const int bb = lane & 31;
#pragma unroll 4
      for (int cc = lane >> 5; cc < 32; cc += 2)
        if (b0 + bb < b2 && c0 + cc < s2) F[(int64_t)(c0 + cc) * m + s2 + b0 + bb] = tile[cc][bb];
    }
  }
}

}  // namespace

void launch_factor(plfem_ctx* c, double sigma, int stop_level, int stop_step, int stop_stage) {
  // stop_*: debugging aid (plfem_debug_factor_until); stop_level < 0 = run to completion.
  // stages: 0 assembled, 1 or 2 pivot block + panel of the step done (its own launch for step 0, the launch of the step before
  // otherwise), 3 or 4 the step's launch done (update + inverse row + pivot write-back; next pivot block + panel), 5 level done
  hipStream_t st = c->stream;
  (void)hipMemsetAsync(c->d_counters, 0, 4 * sizeof(int32_t), st);
  for (int lev = c->L; lev >= 0; --lev) {
    const LevelInfo& li = c->levels[lev];
    if (lev == c->L) {
      const int64_t lo = c->S->foff[li.first], hi = c->S->foff[li.first + li.count];
      (void)hipMemsetAsync(c->d_front + lo, 0, sizeof(double) * (size_t)(hi - lo), st);
      int64_t sch = 0;                                          // the leaf level's Schur complements in its arena
      for (int f = li.first; f < li.first + li.count; ++f) {
        const int64_t b2 = (int64_t)c->dpn * c->S->fb[f];
        sch = std::max(sch, c->S->soff[f] + b2 * b2);
      }
      (void)hipMemsetAsync(c->d_schur + (size_t)(lev & 1) * c->arena_doubles, 0, sizeof(double) * (size_t)sch, st);
      if (c->sh)
        hipLaunchKernelGGL(k_leaf_assemble<1>, dim3(li.count), dim3(256), 0, st, li.first, c->ne, sigma, c->d_fs2, c->d_fm, c->d_foff,
                           c->d_soff, c->d_fnode_ptr, c->d_fnodes, c->d_leaf_elem_ptr, c->d_leaf_elems, c->d_epos, c->d_elem,
                           c->d_front, c->d_schur, c->arena_doubles);
      else
        hipLaunchKernelGGL(k_leaf_assemble<0>, dim3(li.count), dim3(256), 0, st, li.first, c->ne, sigma, c->d_fs2, c->d_fm, c->d_foff,
                           c->d_soff, c->d_fnode_ptr, c->d_fnodes, c->d_leaf_elem_ptr, c->d_leaf_elems, c->d_epos, c->d_elem,
                           c->d_front, c->d_schur, c->arena_doubles);
    } else if (li.gather_n > 0) {
      if (c->sh)
        hipLaunchKernelGGL(k_front_gather<1>, dim3(li.gather_n), dim3(256), 0, st, c->d_tiles + li.gather_off, c->d_fs2, c->d_fm,
                           c->d_foff, c->d_soff, c->d_fnode_ptr, c->d_fnodes, c->d_cinv0, c->d_cinv1, c->d_front, c->d_schur,
                           c->arena_doubles);
      else
        hipLaunchKernelGGL(k_front_gather<0>, dim3(li.gather_n), dim3(256), 0, st, c->d_tiles + li.gather_off, c->d_fs2, c->d_fm,
                           c->d_foff, c->d_soff, c->d_fnode_ptr, c->d_fnodes, c->d_cinv0, c->d_cinv1, c->d_front, c->d_schur,
                           c->arena_doubles);
    }
    if (lev == stop_level && stop_stage == 0) return;
    // Fronts of the level in order of decreasing s2 (c->forder): the fronts still active at block step kb are a
    // prefix of that order, so every launch only covers them and is sized by the largest ACTIVE front.
    const int steps = (li.max_s2 + NB - 1) / NB;
    const int32_t* ford = c->d_forder + li.first;
    const FrontRec* frec = c->d_frec + li.first;
    const int* hs2 = c->forder_s2.data() + li.first;          // s2 in that order (descending)
    const int* hpm = c->forder_maxm.data() + li.first;        // running maximum of m in that order
    for (int kb = 0; kb < steps; ++kb) {
      const bool stop_here = (lev == stop_level && kb == stop_step);
      const int k0 = kb * NB;
      int nact = 0;                                            // fronts with s2 > k0
      {
        int lo = 0, hi = li.count;
        while (lo < hi) { int mid = (lo + hi) / 2; if (hs2[mid] > k0) lo = mid + 1; else hi = mid; }
        nact = lo;
      }
      if (nact == 0) break;
      // W / Y of three consecutive steps live in separate thirds of wbuf / rbuf, X of the pivot blocks of two steps in
      // halves of dinv (see k_ldl_update)
      // (the kernels address these buffers by 2 fnode_ptr[f] NB: the pointers handed to them are shifted back by the
      // offset of the level's first front, so that the level in flight starts at the beginning of the buffer)
      const size_t third = (size_t)2 * c->level_nodes_max * NB;
      const int64_t base = 2 * c->S->fnode_ptr[li.first] * NB;
      double* wb0 = c->d_wbuf - base;
      double* rb0 = c->d_rbuf - base;
      double* wb = wb0 + (kb % 3) * third;
      double* rb = rb0 + (kb % 3) * third;
      double* wb_prev = wb0 + ((kb + 2) % 3) * third;
      double* rb_prev = rb0 + ((kb + 2) % 3) * third;
      double* wb_next = wb0 + ((kb + 1) % 3) * third;
      double* rb_next = rb0 + ((kb + 1) % 3) * third;
      double* dinv_cur = c->d_dinv + (size_t)(kb & 1) * c->nfronts * NB * NB;
      double* dinv_nxt = c->d_dinv + (size_t)((kb + 1) & 1) * c->nfronts * NB * NB;
      // panel / column workgroups per front: all 64-row chunks in parallel where the level is a latency chain (few fronts);
      // where it is throughput every extra workgroup is one more redundant pivot factorisation: at most `cap`
      auto panel_wgs = [&](int rows_below, int cap) {
        int n = std::max(1, rows_below > 0 ? (rows_below + 63) / 64 : 0);
        return li.count > 64 ? std::min(n, cap) : n;
      };
      if (kb == 0) {
        // first launch of the level: pivot block + panel of step 0
        const int n_pan0 = panel_wgs(hpm[nact - 1] - 16, 2);
        hipLaunchKernelGGL(k_ldl_first_panel, dim3(nact, n_pan0), dim3(256), 0, st, frec, c->d_front, dinv_cur, c->d_delta, wb, rb,
                           c->d_counters);
      }
      if (stop_here && stop_stage >= 1 && stop_stage <= 2) return;
      // the step's launch: trailing update + triangular-inverse update + write-back of the pivot block, and for the
      // fronts with a next step its pivot block, its panel and the copy of its block row
      const int un = c->upd_n[li.step0 + kb];                  // 64 x 64 blocks of this step's trailing updates
      const int n_inv = kb > 0 ? (k0 + 15) / 16 : 0;        // one workgroup per 16 columns of the block row
      const int2* ut = c->d_tiles + c->upd_off[li.step0 + kb];
      int n_look = 0;                                          // fronts with a next block step: s2 > k0 + NB (a prefix)
      {
        int lo = 0, hi = nact;
        while (lo < hi) { int mid = (lo + hi) / 2; if (hs2[mid] > k0 + NB) lo = mid + 1; else hi = mid; }
        n_look = lo;
      }
      // (measured on C1, level of 2048 / 1024 / 512 / 256 / 128 fronts: 678 / 257 / 399 / 539 / 441 us with one column
      // workgroup per front, 749 / 256 / 332 / 383 / 321 us with up to four)
      static const int col_cap = getenv("PLFEM_COLUMN_WGS_CAP") ? std::max(1, atoi(getenv("PLFEM_COLUMN_WGS_CAP"))) : 0;
      const int cap = col_cap > 0 ? col_cap : (li.count >= 1024 ? 1 : 4);
      const int n_pan = n_look > 0 ? panel_wgs(hpm[n_look - 1] - (k0 + NB) - 16, cap) : 1;
      const unsigned gridB = (unsigned)(n_look * n_pan + un + nact * (n_inv + 1));
      if ((kb & 1) == 0)
        hipLaunchKernelGGL(k_ldl_update<0>, dim3(gridB), dim3(256), 0, st, un, n_inv, n_look, n_pan, ut, ford, frec, kb, c->d_fs2,
                           c->d_fm, c->d_foff, c->d_soff, c->d_fnode_ptr, c->d_front, c->d_schur, c->arena_doubles, dinv_cur,
                           dinv_nxt, c->d_delta, wb, rb, wb, rb, wb_next, rb_next, c->d_counters);
      else
        hipLaunchKernelGGL(k_ldl_update<1>, dim3(gridB), dim3(256), 0, st, un, n_inv, n_look, n_pan, ut, ford, frec, kb, c->d_fs2,
                           c->d_fm, c->d_foff, c->d_soff, c->d_fnode_ptr, c->d_front, c->d_schur, c->arena_doubles, dinv_cur,
                           dinv_nxt, c->d_delta, wb, rb, wb_prev, rb_prev, wb_next, rb_next, c->d_counters);
      if (stop_here && (stop_stage == 3 || stop_stage == 4)) return;
    }
    if (stop_level >= 0 && li.formz_n + li.mirrorx_n > 0)     // (debug run that stops after a level: its Z and lower(F11) now)
      hipLaunchKernelGGL(k_form_z_mirror, dim3(li.formz_n + li.mirrorx_n), dim3(256), 0, st, c->d_tiles + li.formz_off, li.formz_n,
                         c->d_fs2, c->d_fm, c->d_foff, c->d_front);
    if (lev == stop_level && stop_stage == 5) return;
  }
  // Z = L21 L11^-1 (in place and transposed) and lower(F11) = the mirror of upper(F11), for every front at once: only the
  // solve sweeps read them (the two job lists are adjacent in d_tiles: Z blocks first)
  if (stop_level < 0 && c->formz_all_n + c->mirrorx_all_n > 0)
    hipLaunchKernelGGL(k_form_z_mirror, dim3(c->formz_all_n + c->mirrorx_all_n), dim3(256), 0, st, c->d_tiles + c->formz_all_off,
                       c->formz_all_n, c->d_fs2, c->d_fm, c->d_foff, c->d_front);
}

}  // namespace plfem
