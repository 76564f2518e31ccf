// gfx950 kernels: tall-skinny panel products for the B-orthogonal Lanczos basis, Ritz rotation,
// per-mode post-processing.  All reductions are two-stage with a fixed summation order
// (deterministic; no float atomics).
//
// Replaces the re-orthogonalisation / Ritz-vector work ARPACK does inside dsaupd / dseupd
// (scipy arpack.py:542-602, reached from reference solver_fem.py:197) and the per-mode loop of
// reference solver_fem.py:200-225.
#include <algorithm>
#include <cmath>
#include <functional>

#include "device.h"

namespace plfem {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int CHUNK = PANEL_CHUNK;   // rows per partial sum

// partial[c * nchunks + chunk] = sum_{i in chunk} P[i, c] * w[i];  one wave per (chunk, column)
__global__ __launch_bounds__(256) void k_panel_dot(int64_t n, int ncols, int nchunks, const double* __restrict__ P,
                                                   const double* __restrict__ w, double* __restrict__ partial) {
  const int c = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (c >= ncols) return;
  const int lane = threadIdx.x & 63;
  const int64_t i0 = (int64_t)blockIdx.x * CHUNK;
  const int64_t i1 = min(n, i0 + CHUNK);
  const double* col = P + (int64_t)c * n;
  double acc = 0.0;
  for (int64_t i = i0 + lane; i < i1; i += 64) acc += col[i] * w[i];
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) partial[(int64_t)c * nchunks + blockIdx.x] = acc;
}

// h[c] = sum over chunks of partial[c][*]: one wave per column, fixed (lane-strided, then butterfly) order
__global__ __launch_bounds__(64) void k_panel_dot_finish(int ncols, int nchunks, const double* __restrict__ partial,
                                                         double* __restrict__ h) {
  const int c = blockIdx.x;
  if (c >= ncols) return;
  double acc = 0.0;
  for (int q = threadIdx.x; q < nchunks; q += 64) acc += partial[(int64_t)c * nchunks + q];
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (threadIdx.x == 0) h[c] = acc;
}

// w[i] -= sum_c P[i, c] h[c]
__global__ __launch_bounds__(256) void k_panel_axpy(int64_t n, int ncols, const double* __restrict__ P,
                                                    const double* __restrict__ h, double* __restrict__ w) {
  __shared__ double sh[PLFEM_MAX_NCV + 8];
  for (int c = threadIdx.x; c < ncols; c += 256) sh[c] = h[c];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (int c = 0; c < ncols; ++c) acc += P[(int64_t)c * n + i] * sh[c];
  w[i] -= acc;
}

__global__ void k_vec_add(int n, double* __restrict__ acc, const double* __restrict__ h) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) acc[i] += h[i];
}

// v = w / sqrt(beta2), bv = bw / sqrt(beta2); beta written to *beta_out
__global__ __launch_bounds__(256) void k_scale_store(int64_t n, const double* __restrict__ w,
                                                     const double* __restrict__ bw, const double* __restrict__ beta2,
                                                     double* __restrict__ v, double* __restrict__ bv,
                                                     double* __restrict__ beta_out) {
  const double b2 = *beta2;
  const double beta = sqrt(fmax(b2, 0.0));
  const double inv = beta > 0.0 ? 1.0 / beta : 0.0;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0 && beta_out) *beta_out = beta;
  if (i >= n) return;
  v[i] = w[i] * inv;
  bv[i] = bw[i] * inv;
}

__global__ __launch_bounds__(256) void k_axpby(int64_t n, double a, const double* __restrict__ x, double b,
                                               const double* __restrict__ y, double* __restrict__ z) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) z[i] = a * x[i] + b * y[i];
}

__global__ __launch_bounds__(256) void k_scale(int64_t n, double a, double* __restrict__ x) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] *= a;
}

// ---- block (P-vector) variants for block Lanczos --------------------------------------------------
// partial[(c*P + q) * nchunks + chunk] = sum_{i in chunk} Pm[i, c] * W[i, q].  One wave per (chunk, 4 columns):
// the P values of W are loaded once for four columns of the panel (a wave per column re-read W from L2 ncols
// times, which cost more than streaming the panel itself), two row groups per iteration = 16 loads in flight.
template <int P>
__global__ __launch_bounds__(256) void k_panel_dot_p(int64_t n, int ncols, int nchunks, const double* __restrict__ Pm,
                                                     const double* __restrict__ W, int64_t ldw,
                                                     double* __restrict__ partial) {
  constexpr int CW = 4;                                  // columns per wave
  const int c0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * CW;
  if (c0 >= ncols) return;
  const int lane = threadIdx.x & 63;
  const int64_t i0 = (int64_t)blockIdx.x * CHUNK;
  const int64_t i1 = min(n, i0 + CHUNK);
  const double* col[CW];
#pragma unroll
  for (int t = 0; t < CW; ++t) col[t] = Pm + (int64_t)min(c0 + t, ncols - 1) * n;   // clamped: result discarded
  double acc[CW][P];
#pragma unroll
  for (int t = 0; t < CW; ++t)
#pragma unroll
    for (int q = 0; q < P; ++q) acc[t][q] = 0.0;
  int64_t i = i0 + lane;
  for (; i + 64 < i1; i += 128) {
    double a[2][CW], w[2][P];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 0; t < CW; ++t) a[h][t] = col[t][i + 64 * h];
#pragma unroll
      for (int q = 0; q < P; ++q) w[h][q] = W[(int64_t)q * ldw + i + 64 * h];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < CW; ++t)
#pragma unroll
        for (int q = 0; q < P; ++q) acc[t][q] += a[h][t] * w[h][q];
  }
  for (; i < i1; i += 64) {
    double w[P];
#pragma unroll
    for (int q = 0; q < P; ++q) w[q] = W[(int64_t)q * ldw + i];
#pragma unroll
    for (int t = 0; t < CW; ++t) {
      const double a = col[t][i];
#pragma unroll
      for (int q = 0; q < P; ++q) acc[t][q] += a * w[q];
    }
  }
#pragma unroll
  for (int t = 0; t < CW; ++t) {
#pragma unroll
    for (int q = 0; q < P; ++q) {
      double v = acc[t][q];
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0 && c0 + t < ncols) partial[((int64_t)(c0 + t) * P + q) * nchunks + blockIdx.x] = v;
    }
  }
}

// h[c + q*ldh] = sum over chunks (one wave per (c, q), fixed order)
__global__ __launch_bounds__(64) void k_panel_dot_finish_p(int P, int nchunks, const double* __restrict__ partial,
                                                           double* __restrict__ h, int ldh, double* __restrict__ hacc,
                                                           int ldacc) {
  const int cq = blockIdx.x;          // c*P + q
  const int c = cq / P, q = cq % P;
  double acc = 0.0;
  for (int t = threadIdx.x; t < nchunks; t += 64) acc += partial[(int64_t)cq * nchunks + t];
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (threadIdx.x == 0) {
    h[c + (int64_t)q * ldh] = acc;
    if (hacc) hacc[c + (int64_t)q * ldacc] += acc;     // CGS2: the second-pass coefficients also go into T
  }
}

// W[i, q] -= sum_c Pm[i, c] H[c + q*ldh]
// wil (may be null): a second copy of the updated block with the P values of every DOF together,
// wil[(node dpn + component) P + q] -- what the block SpMV that follows gathers from (k_spmv_b_block_il)
template <int P>
__global__ __launch_bounds__(256) void k_panel_axpy_p(int64_t n, int ncols, const double* __restrict__ Pm,
                                                      const double* __restrict__ H, int ldh, double* __restrict__ W,
                                                      int64_t ldw, double* __restrict__ wil, int N, int dpn) {
  extern __shared__ double sh[];      // [c][P]
  for (int k = threadIdx.x; k < ncols * P; k += 256) sh[k] = H[(k / P) + (int64_t)(k % P) * ldh];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double acc[P];
#pragma unroll
  for (int q = 0; q < P; ++q) acc[q] = 0.0;
  for (int c = 0; c < ncols; ++c) {
    const double a = Pm[(int64_t)c * n + i];
#pragma unroll
    for (int q = 0; q < P; ++q) acc[q] += a * sh[c * P + q];
  }
  double w[P];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    w[q] = W[(int64_t)q * ldw + i] - acc[q];
    W[(int64_t)q * ldw + i] = w[q];
  }
  if (wil) {
    const int comp = (int)(i / N), node = (int)(i - (int64_t)comp * N);
    double* d = wil + ((int64_t)node * dpn + comp) * P;
#pragma unroll
    for (int q = 0; q < P; ++q) d[q] = w[q];
  }
}

// (Round 4 measured 16-byte variants of the two panel products -- two rows per lane, W staged in LDS, 8 columns per wave,
// double-buffered column batches: scripts/micro/panel_bench.hip.  At 72-96 columns k_panel_axpy_p already streams at
// 5.6-5.9 TB/s, what a plain streaming read of the same panel reaches on this chip (5.4-5.9), and k_panel_dot_p at 4.2;
// every variant was equal or slower, so the kernels above stay.  Below ~24 columns both are at their launch-latency floor.)

// ---- first Gram-Schmidt pass of a block step in TWO launches instead of four (round 4) ----------------------------
// The first pass runs over the last two blocks of the basis only (8 columns): its four launches -- the permutation of the
// sweeps' result into global order, the panel dot, the sum of its partials, the panel axpy -- are each at their 5-8 us
// latency floor.  k_permute_dot_first does the permutation and the dot in one pass over the rows (the block is read from
// the sweeps' front-order result where k_permute_out would read it, written out in global order, and multiplied with the
// up to 8 panel columns on the way); k_axpy_first sums the partials in its prologue (32 values x nseg partials, every
// workgroup for itself in the same fixed order: the same bits everywhere) before it applies the update.
constexpr int FIRST_ROWS = 1024;       // rows per workgroup of k_permute_dot_first = per partial sum (512 measured: 17.1 + 11.1 us against 15.8 + 9.5)
template <int P>
__global__ __launch_bounds__(256) void k_permute_dot_first(int64_t n2, int N, int nseg, int ncols, const int32_t* __restrict__ npos,
                                                           const double* __restrict__ xl, double* __restrict__ W, int64_t ldw,
                                                           const double* __restrict__ Pm, double* __restrict__ partial) {
  constexpr int CW = 8;
  __shared__ double red[4][CW * P];
  const int seg = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double acc[CW * P];
#pragma unroll
  for (int v = 0; v < CW * P; ++v) acc[v] = 0.0;
  const double* col[CW];
#pragma unroll
  for (int t = 0; t < CW; ++t) col[t] = Pm + (int64_t)min(t, ncols - 1) * n2;      // (clamped: the result is dropped)
  // the thread's four rows: every load that does not depend on another is requested first (the index of a row's
  // front-order slot and its 8 panel entries), then the 16 gathers; a row-by-row loop was a chain of 8 memory round trips
  constexpr int RPT = FIRST_ROWS / 256;
  int64_t g[RPT];
  int pos[RPT];
  double a[RPT][CW], w[RPT][P];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    g[j] = (int64_t)seg * FIRST_ROWS + j * 256 + threadIdx.x;
    const bool on = g[j] < n2;
    const int c = on && g[j] >= N;
    pos[j] = on ? npos[(int)(g[j] - (int64_t)c * N)] : -1;
#pragma unroll
    for (int t = 0; t < CW; ++t) a[j][t] = on ? col[t][g[j]] : 0.0;
  }
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int c = g[j] >= N;
#pragma unroll
    for (int q = 0; q < P; ++q) w[j][q] = pos[j] >= 0 ? xl[((int64_t)pos[j] + c) * P + q] : 0.0;
  }
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    if (g[j] < n2) {
#pragma unroll
      for (int q = 0; q < P; ++q) W[(int64_t)q * ldw + g[j]] = w[j][q];
    }
#pragma unroll
    for (int t = 0; t < CW; ++t)
#pragma unroll
      for (int q = 0; q < P; ++q) acc[t * P + q] = fma(a[j][t], w[j][q], acc[t * P + q]);
  }
#pragma unroll
  for (int v = 0; v < CW * P; ++v) {
    double x = acc[v];
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    if (lane == 0) red[wave][v] = x;
  }
  __syncthreads();
  if ((int)threadIdx.x < CW * P) {
    const int v = threadIdx.x, t = v / P;
    if (t < ncols) partial[(int64_t)v * nseg + seg] = (red[0][v] + red[1][v]) + (red[2][v] + red[3][v]);
  }
}

// W[i, q] -= sum_c Pm[i, c] h[c][q] with h[c][q] = sum over the nseg partials of k_permute_dot_first (prologue); workgroup 0
// also stores h into the projected matrix (Hout[c + q ldh])
template <int P>
__global__ __launch_bounds__(256) void k_axpy_first(int64_t n, int ncols, int nseg, const double* __restrict__ Pm,
                                                    const double* __restrict__ partial, double* __restrict__ Hout, int ldh,
                                                    double* __restrict__ W, int64_t ldw) {
  constexpr int CW = 8;
  __shared__ double sh[CW * P];
  {
    // 8 threads per value: thread part p adds partials p, p + 8, ... in order, then a fixed butterfly over the 8 parts
    const int v = threadIdx.x >> 3, part = threadIdx.x & 7;
    double x = 0.0;
    if (v / P < ncols) {
      const double* pp = partial + (int64_t)v * nseg;
      int k = part;
      for (; k + 7 * 8 < nseg; k += 8 * 8) {               // eight loads in flight (a load per add was a chain of ~22 round trips)
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = pp[k + 8 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) x += t[u];
      }
      for (; k < nseg; k += 8) x += pp[k];
    }
    x += __shfl_xor(x, 1, 8);
    x += __shfl_xor(x, 2, 8);
    x += __shfl_xor(x, 4, 8);
    if (part == 0) {
      sh[v] = x;
      if (blockIdx.x == 0 && v / P < ncols) Hout[(v / P) + (int64_t)(v % P) * ldh] = x;
    }
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double a[CW];
#pragma unroll
  for (int t = 0; t < CW; ++t) a[t] = Pm[(int64_t)min(t, ncols - 1) * n + i];
  double acc[P];
#pragma unroll
  for (int q = 0; q < P; ++q) acc[q] = 0.0;
#pragma unroll
  for (int t = 0; t < CW; ++t)
    if (t < ncols) {
#pragma unroll
      for (int q = 0; q < P; ++q) acc[q] = fma(a[t], sh[t * P + q], acc[q]);
    }
#pragma unroll
  for (int q = 0; q < P; ++q) W[(int64_t)q * ldw + i] -= acc[q];
}

// Cholesky G = R^T R of the P x P Gram matrix of the new block (one lane), R into the projected matrix
// (rows nc.., columns c0..), R^-1 for the block scaling.  A non-positive pivot (rank-deficient block:
// the Krylov space is exhausted) raises counters[2].
// G comes either ready-made (G != nullptr) or as the chunk partials of k_panel_dot_p (partial[((c P + q) nchunks +
// chunk)], summed here in the same fixed order as k_panel_dot_finish_p: one launch less per block step)
template <int P>
__global__ __launch_bounds__(64 * P * P > 1024 ? 1024 : 64 * P * P) void k_chol_small(const double* __restrict__ G, int ldg, const double* __restrict__ partial,
                                                           int nchunks, double* __restrict__ Tblk, int ldT,
                                                           double* __restrict__ Rinv, int32_t* __restrict__ counters) {
  __shared__ double sG[P * P];
  if (G) {
    if (threadIdx.x < P * P) sG[threadIdx.x] = G[(threadIdx.x % P) + (int64_t)(threadIdx.x / P) * ldg];
  } else {
    const int lane = threadIdx.x & 63;
    for (int cq = threadIdx.x >> 6; cq < P * P; cq += blockDim.x >> 6) {   // one wave per entry; cq = c P + q  ->  G[c + q ldg]
      // (up to ~3000 partials per entry since the fused B product delivers them: eight loads in flight per lane, fixed order)
      const double* pp = partial + (int64_t)cq * nchunks;
      double acc = 0.0;
      int t = lane;
      for (; t + 7 * 64 < nchunks; t += 8 * 64) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pp[t + 64 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; t < nchunks; t += 64) acc += pp[t];
      for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
      if (lane == 0) sG[(cq / P) + (cq % P) * P] = acc;
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  ldg = P;
  G = sG;
  double R[P][P], X[P][P];
  for (int i = 0; i < P; ++i)
    for (int j = 0; j < P; ++j) { R[i][j] = 0.0; X[i][j] = 0.0; }
  for (int j = 0; j < P; ++j) {
    for (int i = 0; i <= j; ++i) {
      double v = 0.5 * (G[i + (int64_t)j * ldg] + G[j + (int64_t)i * ldg]);
      for (int k = 0; k < i; ++k) v -= R[k][i] * R[k][j];
      if (i == j) {
        if (!(v > 0.0)) { atomicAdd(&counters[2], 1); v = 1.0; }
        R[j][j] = sqrt(v);
      } else {
        R[i][j] = v / R[i][i];
      }
    }
  }
  for (int j = 0; j < P; ++j) {           // X = R^-1 (upper), column by column
    X[j][j] = 1.0 / R[j][j];
    for (int i = j - 1; i >= 0; --i) {
      double v = 0.0;
      for (int k = i + 1; k <= j; ++k) v -= R[i][k] * X[k][j];
      X[i][j] = v / R[i][i];
    }
  }
  for (int i = 0; i < P; ++i)
    for (int j = 0; j < P; ++j) {
      Tblk[i + (int64_t)j * ldT] = R[i][j];
      Rinv[i + j * P] = X[i][j];
    }
}

// Vn = W R^-1, BVn = BW R^-1 (R^-1 upper triangular)
// exp_*: workgroup 0 also stores the step's new projected-matrix columns and the counters into the pinned slot of
// the host (a store over the host link from the last kernel of the step costs nothing; two blit copies behind it
// cost ~25 us of stream time per block step)
template <int P>
__global__ __launch_bounds__(256) void k_block_scale(int64_t n, const double* __restrict__ W, const double* __restrict__ BW,
                                                     int64_t ldw, const double* __restrict__ Rinv,
                                                     double* __restrict__ Vn, double* __restrict__ BVn, int64_t ldv,
                                                     const double* __restrict__ exp_src, int exp_n, double* __restrict__ exp_dst,
                                                     const int32_t* __restrict__ cnt_src, int32_t* __restrict__ cnt_dst,
                                                     double* __restrict__ bv_front, const int32_t* __restrict__ npos, int N) {
  __shared__ double X[P * P];
  if (threadIdx.x < P * P) X[threadIdx.x] = Rinv[threadIdx.x];
  if (blockIdx.x == 0 && exp_dst) {
    for (int t = threadIdx.x; t < exp_n; t += 256) exp_dst[t] = exp_src[t];
    if (threadIdx.x < 4) cnt_dst[threadIdx.x] = cnt_src[threadIdx.x];
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double w[P], bw[P];
#pragma unroll
  for (int q = 0; q < P; ++q) { w[q] = W[(int64_t)q * ldw + i]; bw[q] = BW[(int64_t)q * ldw + i]; }
  // the next solve's right-hand side also goes out in the sweeps' layout (front order, P values per DOF together)
  int64_t fo = -1;
  if (bv_front) {
    const int c = i >= N;
    const int pos = npos[(int)(i - (int64_t)c * N)];
    if (pos >= 0) fo = ((int64_t)pos + c) * P;
  }
#pragma unroll
  for (int q = 0; q < P; ++q) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k <= q; ++k) { a += w[k] * X[k + q * P]; b += bw[k] * X[k + q * P]; }
    Vn[(int64_t)q * ldv + i] = a;
    BVn[(int64_t)q * ldv + i] = b;
    if (fo >= 0) bv_front[fo + q] = b;
  }
}

// Ritz rotation out[:, 0:p] = V[:, 0:m] S[0:m, 0:p] on the matrix cores: one wave per 16 rows,
// v_mfma_f64_16x16x4_f64 tiles (A <- V rows, B <- S), S staged in LDS.  m, p <= 144.
__global__ __launch_bounds__(256) void k_rotate(int64_t n, int m, int p, const double* __restrict__ V,
                                                const double* __restrict__ Smat, int ldS, double* __restrict__ out) {
  extern __shared__ double sS[];   // [mpad][ppad], mpad multiple of 4, ppad multiple of 16
  const int mpad = (m + 3) & ~3, ppad = (p + 15) & ~15;
  for (int k = threadIdx.x; k < mpad * ppad; k += 256) {
    int r = k / ppad, c = k % ppad;
    sS[k] = (r < m && c < p) ? Smat[(int64_t)c * ldS + r] : 0.0;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
  if (i0 >= n) return;
  const int lr = lane & 15, lk = lane >> 4;
  const int64_t row = i0 + lr;
  const bool rv = row < n;
  // out^T tile: A <- S^T (MFMA row = output column), B <- V^T (MFMA col = vector row), so that the
  // accumulator register r of lane l is out[i0 + (l&15), c0 + (l>>4) + 4 r]: 128-B runs per column.
  for (int c0 = 0; c0 < ppad; c0 += 16) {
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int kk = 0; kk < mpad; kk += 4) {
      int k = kk + lk;
      double a = sS[k * ppad + c0 + lr];
      double b = (rv && k < m) ? V[(int64_t)k * n + row] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int oc = c0 + lk + 4 * r;
      if (rv && oc < p) out[(int64_t)oc * n + row] = acc[r];
    }
  }
}

// ---- post-processing ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_core_mask(int N, const double* __restrict__ doflocs,
                                                   const double* __restrict__ cores, int ncore,
                                                   const uint8_t* __restrict__ bmask, uint8_t* __restrict__ mask,
                                                   int32_t* __restrict__ counters) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool in = false;
  if (i < N) {
    double X = doflocs[i], Y = doflocs[N + i];
    for (int c = 0; c < ncore; ++c) {
      double dx = X - cores[3 * c], dy = Y - cores[3 * c + 1], r = cores[3 * c + 2];
      in |= (dx * dx + dy * dy <= r * r);
    }
    mask[i] = in ? 1 : 0;
  }
  // integer count of interior DOF nodes inside a core (order independent => deterministic)
  unsigned long long b = __ballot(i < N && in && !bmask[i]);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&counters[1], (int)__popcll(b));
}

// per (row chunk, mode) partial sums: [0] sum vx^2, [1] sum vy^2, [2] core vx^2, [3] core vy^2,
// [4] vx.Dxx vx + 2 vx.Dxy vy + vy.Dyy vy.   8 lanes per row.
constexpr int POST_ROWS = 256;   // rows per block
// DPN = 1 (scalar solver, solver_fem.py:268-271): [0] sum v^2, [2] core v^2, [4] v.M v with M in the dxx argument
// MB modes per workgroup (blockIdx.y = group of MB): the matrix entries and column indices of a row are read once for
// MB vectors instead of once per vector (the 22 eigenvectors of C1 read the three D blocks 22 times: 167 us); each mode's
// sums are formed in the order of the one-mode kernel, so the results are the same bits.
constexpr int POST_MB = 4;
template <int DPN>
__global__ __launch_bounds__(256) void k_post_sums(int N, int k, int nblocks, const int32_t* __restrict__ rowptr,
                                                   const int32_t* __restrict__ colind, const double* __restrict__ dxx,
                                                   const double* __restrict__ dxy, const double* __restrict__ dyy,
                                                   const uint8_t* __restrict__ mask, const double* __restrict__ evecs,
                                                   double* __restrict__ partial) {
  constexpr int MB = POST_MB;
  const int mode0 = blockIdx.y * MB;
  const int nm = min(MB, k - mode0);
  const double* vx[MB];
#pragma unroll
  for (int j = 0; j < MB; ++j) vx[j] = evecs + (int64_t)min(mode0 + j, k - 1) * DPN * N;     // (clamped: result dropped)
  const int64_t yoff = DPN == 2 ? N : 0;
  __shared__ double red[4][MB][5];
  double acc[MB][5];
#pragma unroll
  for (int j = 0; j < MB; ++j)
#pragma unroll
    for (int t = 0; t < 5; ++t) acc[j][t] = 0.0;
  const int sub = threadIdx.x & 7;
  for (int rr = threadIdx.x >> 3; rr < POST_ROWS; rr += 32) {
    int row = blockIdx.x * POST_ROWS + rr;
    if (row >= N) break;
    double px[MB], py[MB];   // (Dxx vx + Dxy vy)_row, (Dyy vy)_row
#pragma unroll
    for (int j = 0; j < MB; ++j) { px[j] = 0.0; py[j] = 0.0; }
    int q1 = rowptr[row + 1];
    for (int q = rowptr[row] + sub; q < q1; q += 8) {
      int c = colind[q];
      const double a = dxx[q];
      if (DPN == 1) {
#pragma unroll
        for (int j = 0; j < MB; ++j) px[j] += a * vx[j][c];
        continue;
      }
      const double b = dxy[q], d = dyy[q];
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        double ux = vx[j][c], uy = vx[j][yoff + c];
        px[j] += a * ux + 2.0 * b * uy;
        py[j] += d * uy;
      }
    }
    const bool msk = mask[row] != 0;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      double x = vx[j][row], y = DPN == 2 ? vx[j][yoff + row] : 0.0;
      acc[j][4] += x * px[j] + y * py[j];
      if (sub == 0) {
        acc[j][0] += x * x;
        acc[j][1] += y * y;
        if (msk) { acc[j][2] += x * x; acc[j][3] += y * y; }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MB; ++j)
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      double v = acc[j][t];
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j][t] = v;
    }
  __syncthreads();
  if ((int)threadIdx.x < 5 * nm) {
    const int j = threadIdx.x / 5, t = threadIdx.x % 5;
    partial[((int64_t)(mode0 + j) * nblocks + blockIdx.x) * 5 + t] = red[0][j][t] + red[1][j][t] + red[2][j][t] + red[3][j][t];
  }
}

// one wave per (mode, quantity): lane l adds blocks l, l + 64, ... in order, then a fixed-order lane reduction
__global__ __launch_bounds__(64) void k_post_finish(int k, int nblocks, const double* __restrict__ partial,
                                                    double* __restrict__ out) {
  const int t = blockIdx.x;             // mode * 5 + q
  const int mode = t / 5, q = t % 5;
  double acc = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) acc += partial[((int64_t)mode * nblocks + b) * 5 + q];
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (threadIdx.x == 0) out[t] = acc;
}

// normalise each mode in place (solver_fem.py:213)
// dpn = 1: M-normalisation v / (sqrt(v.M v) + 1e-30) as solver_fem.py:268
__global__ __launch_bounds__(256) void k_post_scale(int N, int dpn, const double* __restrict__ sums, double* __restrict__ evecs) {
  const int mode = blockIdx.y;
  const double n2 = dpn == 2 ? sums[mode * 5 + 0] + sums[mode * 5 + 1] : sums[mode * 5 + 4];
  const double nrm = sqrt(n2) + 1e-30;
  const double inv = 1.0 / nrm;
  double* v = evecs + (int64_t)mode * dpn * N;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < dpn * (int64_t)N) v[i] *= inv;
}

__global__ __launch_bounds__(256) void k_gather_interior(int N, int nsolve, int dpn, const int32_t* __restrict__ interior,
                                                         const double* __restrict__ evecs,
                                                         double* __restrict__ modes_int) {
  const int mode = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= dpn * (int64_t)nsolve) return;
  int comp = i >= nsolve ? 1 : 0;
  int q = (int)(i - (int64_t)comp * nsolve);
  modes_int[(int64_t)mode * dpn * nsolve + i] = evecs[(int64_t)mode * dpn * N + (int64_t)comp * N + interior[q]];
}

// ---- a-posteriori residuals: per (row chunk, vector) partial sums [0] sum |A v - lambda B v|^2, [1] sum |A v|^2
// over the interior rows; 8 lanes per scalar row, both field components in one pass over the shared pattern.
template <int DPN>
__global__ __launch_bounds__(256) void k_resid_sums(int N, int k, int nblocks, const int32_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ colind, const uint8_t* __restrict__ bmask,
                                                    const double* __restrict__ vxx, const double* __restrict__ vxy,
                                                    const double* __restrict__ vyx, const double* __restrict__ vyy,
                                                    const double* __restrict__ vm, const double* __restrict__ lam,
                                                    const double* __restrict__ evecs, double* __restrict__ partial) {
  constexpr int MB = POST_MB;                           // modes per workgroup, as in k_post_sums
  const int mode0 = blockIdx.y * MB;
  const int nm = min(MB, k - mode0);
  const double* vx[MB];
  double l[MB];
#pragma unroll
  for (int j = 0; j < MB; ++j) {
    const int mode = min(mode0 + j, k - 1);
    vx[j] = evecs + (int64_t)mode * DPN * N;
    l[j] = lam[mode];
  }
  const int64_t yoff = DPN == 2 ? N : 0;
  __shared__ double red[4][MB][2];
  double r2[MB], a2[MB];
#pragma unroll
  for (int j = 0; j < MB; ++j) { r2[j] = 0.0; a2[j] = 0.0; }
  const int sub = threadIdx.x & 7;
  for (int rr = threadIdx.x >> 3; rr < POST_ROWS; rr += 32) {
    const int row = blockIdx.x * POST_ROWS + rr;
    if (row >= N) break;
    if (bmask[row]) continue;
    double ax[MB], ay[MB], bx[MB], by[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) { ax[j] = 0; ay[j] = 0; bx[j] = 0; by[j] = 0; }
    const int q1 = rowptr[row + 1];
    for (int q = rowptr[row] + sub; q < q1; q += 8) {
      const int c = colind[q];
      const double axx = vxx[q], m = vm[q];
      if (DPN == 1) {
#pragma unroll
        for (int j = 0; j < MB; ++j) { const double ux = vx[j][c]; ax[j] += axx * ux; bx[j] += m * ux; }
        continue;
      }
      const double axy = vxy[q], ayx = vyx[q], ayy = vyy[q];
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        const double ux = vx[j][c], uy = vx[j][yoff + c];
        ax[j] += axx * ux + axy * uy;
        ay[j] += ayx * ux + ayy * uy;
        bx[j] += m * ux;
        by[j] += m * uy;
      }
    }
#pragma unroll
    for (int j = 0; j < MB; ++j) {
#pragma unroll
      for (int off = 4; off >= 1; off >>= 1) {
        ax[j] += __shfl_xor(ax[j], off, 8); ay[j] += __shfl_xor(ay[j], off, 8);
        bx[j] += __shfl_xor(bx[j], off, 8); by[j] += __shfl_xor(by[j], off, 8);
      }
      if (sub == 0) {
        const double rx = ax[j] - l[j] * bx[j], ry = ay[j] - l[j] * by[j];
        r2[j] += rx * rx + ry * ry;
        a2[j] += ax[j] * ax[j] + ay[j] * ay[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MB; ++j) {
    for (int off = 32; off >= 1; off >>= 1) { r2[j] += __shfl_xor(r2[j], off); a2[j] += __shfl_xor(a2[j], off); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][j][0] = r2[j]; red[threadIdx.x >> 6][j][1] = a2[j]; }
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * nm) {
    const int j = threadIdx.x >> 1, t = threadIdx.x & 1;
    partial[((int64_t)(mode0 + j) * nblocks + blockIdx.x) * 2 + t] = red[0][j][t] + red[1][j][t] + red[2][j][t] + red[3][j][t];
  }
}

__global__ __launch_bounds__(64) void k_resid_finish(int nblocks, const double* __restrict__ partial, double* __restrict__ out) {
  const int t = blockIdx.x;             // mode * 2 + q
  const int mode = t >> 1, q = t & 1;
  double acc = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) acc += partial[((int64_t)mode * nblocks + b) * 2 + q];
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (threadIdx.x == 0) out[t] = acc;
}

}  // namespace

// pinned scratch of the two result read-backs (doubles from h_pinned): post-processing sums [0, 5 k), residual check
// lambda up [2048, 2048 + k) and sums down [2048 + k, 2048 + 3 k) -- disjoint, so that both can be in flight together
constexpr int RESID_HS = 2048;

// enqueue only: the k residual sums end up in pinned memory once the stream has drained (resid_finish)
void resid_enqueue(plfem_ctx* c, int k, const double* lam_host, const double* evecs) {
  hipStream_t st = c->stream;
  const int N = c->N;
  const int nblocks = (N + POST_ROWS - 1) / POST_ROWS;
  double* hs = c->h_pinned + RESID_HS;
  for (int i = 0; i < k; ++i) hs[i] = lam_host[i];
  (void)hipMemcpyAsync(c->d_hacc, hs, sizeof(double) * k, hipMemcpyHostToDevice, st);
  double* partial = c->d_post + c->post_doubles;     // second half of d_post: [k][nblocks][2], then the sums
  double* sums = partial + (int64_t)k * nblocks * 2;
  if (c->dpn == 1)
    hipLaunchKernelGGL(k_resid_sums<1>, dim3(nblocks, (k + POST_MB - 1) / POST_MB), dim3(256), 0, st, N, k, nblocks, c->d_rowptr, c->d_colind, c->d_bmask,
                       c->d_vals[PLFEM_BLK_AXX], c->d_vals[PLFEM_BLK_AXY], c->d_vals[PLFEM_BLK_AYX], c->d_vals[PLFEM_BLK_AYY],
                       c->d_vals[PLFEM_BLK_MINV], c->d_hacc, evecs, partial);
  else
    hipLaunchKernelGGL(k_resid_sums<2>, dim3(nblocks, (k + POST_MB - 1) / POST_MB), dim3(256), 0, st, N, k, nblocks, c->d_rowptr, c->d_colind, c->d_bmask,
                       c->d_vals[PLFEM_BLK_AXX], c->d_vals[PLFEM_BLK_AXY], c->d_vals[PLFEM_BLK_AYX], c->d_vals[PLFEM_BLK_AYY],
                       c->d_vals[PLFEM_BLK_MINV], c->d_hacc, evecs, partial);
  hipLaunchKernelGGL(k_resid_finish, dim3(2 * k), dim3(64), 0, st, nblocks, partial, sums);
  (void)hipMemcpyAsync(hs + k, sums, sizeof(double) * 2 * k, hipMemcpyDeviceToHost, st);
}

// after the stream has been synchronised
void resid_finish(plfem_ctx* c, int k, double* out_host) {
  const double* hs = c->h_pinned + RESID_HS;
  for (int i = 0; i < k; ++i) {
    const double r2 = hs[k + 2 * i], a2 = hs[k + 2 * i + 1];
    out_host[i] = a2 > 0.0 ? std::sqrt(r2 / a2) : (r2 > 0.0 ? INFINITY : 0.0);
  }
}

void launch_residuals(plfem_ctx* c, int k, const double* lam_host, const double* evecs, double* out_host) {
  resid_enqueue(c, k, lam_host, evecs);
  (void)hipStreamSynchronize(c->stream);
  resid_finish(c, k, out_host);
}

void launch_axpby_n(plfem_ctx* c, int64_t n, double a, const double* x, double b, const double* y, double* z) {
  hipLaunchKernelGGL(k_axpby, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, a, x, b, y, z);
}

void launch_scale(plfem_ctx* c, int64_t n, double a, double* x) {
  hipLaunchKernelGGL(k_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, a, x);
}

void launch_panel_dot(plfem_ctx* c, const double* P, int ncols, const double* w, double* h) {
  const int nchunks = c->npartial;
  hipLaunchKernelGGL(k_panel_dot, dim3(nchunks, (ncols + 3) / 4), dim3(256), 0, c->stream, c->n2, ncols, nchunks, P,
                     w, c->d_partial);
  hipLaunchKernelGGL(k_panel_dot_finish, dim3(ncols), dim3(64), 0, c->stream, ncols, nchunks, c->d_partial, h);
}

void launch_panel_axpy(plfem_ctx* c, const double* P, int ncols, const double* h, double* w) {
  hipLaunchKernelGGL(k_panel_axpy, dim3((unsigned)((c->n2 + 255) / 256)), dim3(256), 0, c->stream, c->n2, ncols, P, h,
                     w);
}

void launch_dot(plfem_ctx* c, const double* a, const double* b, double* out) { launch_panel_dot(c, a, 1, b, out); }

void launch_vec_add(plfem_ctx* c, double* acc, const double* h, int n) {
  hipLaunchKernelGGL(k_vec_add, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, acc, h);
}

void launch_scale_store(plfem_ctx* c, const double* w, const double* bw, const double* beta2, double* v, double* bv,
                        double* beta_out) {
  hipLaunchKernelGGL(k_scale_store, dim3((unsigned)((c->n2 + 255) / 256)), dim3(256), 0, c->stream, c->n2, w, bw,
                     beta2, v, bv, beta_out);
}

void launch_axpby(plfem_ctx* c, double a, const double* x, double b, const double* y, double* z) {
  hipLaunchKernelGGL(k_axpby, dim3((unsigned)((c->n2 + 255) / 256)), dim3(256), 0, c->stream, c->n2, a, x, b, y, z);
}

void launch_panel_dot_block(plfem_ctx* c, const double* Pm, int ncols, const double* W, int64_t ldw, double* h, int ldh,
                            double* hacc, int ldacc) {
  constexpr int P = BLOCK_P;
  const int nchunks = c->npartial;
  hipLaunchKernelGGL(k_panel_dot_p<P>, dim3(nchunks, (ncols + 15) / 16), dim3(256), 0, c->stream, c->n2, ncols, nchunks,
                     Pm, W, ldw, c->d_partial);
  hipLaunchKernelGGL(k_panel_dot_finish_p, dim3(ncols * P), dim3(64), 0, c->stream, P, nchunks, c->d_partial, h, ldh,
                     hacc, ldacc);
}

void launch_panel_axpy_block(plfem_ctx* c, const double* Pm, int ncols, const double* H, int ldh, double* W, int64_t ldw,
                             double* w_interleaved) {
  constexpr int P = BLOCK_P;
  hipLaunchKernelGGL(k_panel_axpy_p<P>, dim3((unsigned)((c->n2 + 255) / 256)), dim3(256), sizeof(double) * ncols * P,
                     c->stream, c->n2, ncols, Pm, H, ldh, W, ldw, w_interleaved, c->N, c->dpn);
}

// first Gram-Schmidt pass over ncols <= 8 columns, fused with the permutation of the sweeps' result (d_xl, front order)
// into W (global order): h -> Hout, W -= Vm h
void launch_first_pass_block(plfem_ctx* c, const double* BVm, const double* Vm, int ncols, double* W, int64_t ldw, double* Hout,
                             int ldh) {
  constexpr int P = BLOCK_P;
  if constexpr (P == 4) {            // (8 columns x P sums per thread: the two kernels are written for P = 4; see lanczos_block)
    const int nseg = (int)((c->n2 + FIRST_ROWS - 1) / FIRST_ROWS);      // (= npartial: PANEL_CHUNK rows per partial sum)
    hipLaunchKernelGGL(k_permute_dot_first<P>, dim3(nseg), dim3(256), 0, c->stream, c->n2, c->N, nseg, ncols, c->d_npos, c->d_xl, W, ldw,
                       BVm, c->d_partial);
    hipLaunchKernelGGL(k_axpy_first<P>, dim3((unsigned)((c->n2 + 255) / 256)), dim3(256), 0, c->stream, c->n2, ncols, nseg, Vm,
                       c->d_partial, Hout, ldh, W, ldw);
  }
}

void launch_chol_block(plfem_ctx* c, const double* G, int ldg, double* Tblk, int ldT, double* Rinv) {
  hipLaunchKernelGGL(k_chol_small<BLOCK_P>, dim3(1), dim3(64), 0, c->stream, G, ldg, (const double*)nullptr, 0, Tblk, ldT,
                     Rinv, c->d_counters);
}

void launch_chol_from_partials(plfem_ctx* c, int nchunks, double* Tblk, int ldT, double* Rinv) {
  constexpr int P = BLOCK_P;
  hipLaunchKernelGGL(k_chol_small<P>, dim3(1), dim3(std::min(1024, 64 * P * P)), 0, c->stream, (const double*)nullptr, 0, c->d_partial, nchunks,
                     Tblk, ldT, Rinv, c->d_counters);
}

void launch_block_scale(plfem_ctx* c, const double* W, const double* BW, int64_t ldw, const double* Rinv, double* Vn,
                        double* BVn, int64_t ldv, const double* exp_src, int exp_n, double* exp_dst, int32_t* cnt_dst,
                        double* bv_front) {
  hipLaunchKernelGGL(k_block_scale<BLOCK_P>, dim3((unsigned)((c->n2 + 255) / 256)), dim3(256), 0, c->stream, c->n2, W,
                     BW, ldw, Rinv, Vn, BVn, ldv, exp_src, exp_n, exp_dst, c->d_counters, cnt_dst, bv_front, c->d_npos, c->N);
}

// Start block of the Lanczos drivers: the fixed pseudo-random interior field (a 64-bit LCG stream, element e of the
// stream = state after e + 1 updates, values in [-1, 1)), written straight into the padded vectors.  Every thread
// jumps the generator to its own element (a^k and the matching increment by binary powering), so the field is the
// same as a sequential host loop over (vector, component, interior DOF) and costs no host time or upload.
__global__ __launch_bounds__(256) void k_start_field(int nsolve, int N, int dpn, int64_t ld, int nvec,
                                                     const int32_t* __restrict__ interior, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)nvec * dpn * nsolve) return;
  uint64_t k = (uint64_t)e + 1, am = 1, ap = 0, cm = 6364136223846793005ull, cp = 1442695040888963407ull;
  while (k) {
    if (k & 1) { am *= cm; ap = ap * cm + cp; }
    cp = (cm + 1) * cp;
    cm *= cm;
    k >>= 1;
  }
  const uint64_t s = am * 0x9E3779B97F4A7C15ull + ap;
  const int q = (int)(e / (dpn * (int64_t)nsolve));
  const int r = (int)(e - (int64_t)q * dpn * nsolve);
  const int comp = r >= nsolve, i = r - comp * nsolve;
  out[(int64_t)q * ld + (int64_t)comp * N + interior[i]] = ((double)(s >> 11) / 9007199254740992.0) * 2.0 - 1.0;
}

void launch_start_field(plfem_ctx* c, int nvec, double* out) {
  (void)hipMemsetAsync(out, 0, sizeof(double) * c->n2 * nvec, c->stream);
  const int64_t total = (int64_t)nvec * c->dpn * c->nsolve;
  hipLaunchKernelGGL(k_start_field, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->nsolve, c->N, c->dpn,
                     c->n2, nvec, c->d_interior, out);
}

void launch_rotate(plfem_ctx* c, const double* V, int m, const double* Smat, int ldS, int p, double* out) {
  // S is staged in LDS (8 mpad ppad bytes): output columns go in chunks that keep it under ~64 KB, so that a long
  // basis (m up to PLFEM_MAX_NCV + BLOCK_P) never asks for more LDS than a workgroup may have
  const int mpad = (m + 3) & ~3;
  const int chunk = std::max(16, ((64 * 1024) / (8 * mpad)) & ~15);
  unsigned grid = (unsigned)((c->n2 + 63) / 64);
  for (int p0 = 0; p0 < p; p0 += chunk) {
    const int pc = std::min(chunk, p - p0);
    const size_t lds = sizeof(double) * mpad * ((pc + 15) & ~15);
    hipLaunchKernelGGL(k_rotate, dim3(grid), dim3(256), lds, c->stream, c->n2, m, pc, V, Smat + (size_t)p0 * ldS, ldS,
                       out + (size_t)p0 * c->n2);
  }
}

// group_done (may be null): called after the kernels of every group of POST_GROUP modes [g0, g0 + kg) have been enqueued
// -- their interior copies in modes_int are then complete in stream order -- so that the caller can start sending them to
// the host while the next group is still being processed (plfem_solve_modes)
constexpr int POST_GROUP = 8;
void post_enqueue(plfem_ctx* c, int k, double* evecs, int ncore, double* modes_int, const std::function<void(int, int)>* group_done) {
  hipStream_t st = c->stream;
  const int N = c->N;
  (void)hipMemsetAsync(c->d_counters + 1, 0, sizeof(int32_t), st);
  hipLaunchKernelGGL(k_core_mask, dim3((N + 255) / 256), dim3(256), 0, st, N, c->d_doflocs, c->d_cores, ncore,
                     c->d_bmask, c->d_coremask, c->d_counters);
  const int nblocks = (N + POST_ROWS - 1) / POST_ROWS;
  static const int post_group = getenv("PLFEM_POST_GROUP") ? std::max(1, atoi(getenv("PLFEM_POST_GROUP"))) : POST_GROUP;   // (tuning aid)
  const int group = group_done ? post_group : k;
  for (int g0 = 0; g0 < k; g0 += group) {
    const int kg = std::min(group, k - g0);
    double* ev = evecs + (int64_t)g0 * c->n2;
    double* partial = c->d_post + (int64_t)g0 * nblocks * 5;          // [k][nblocks][5]
    double* sums = c->d_post + (int64_t)k * nblocks * 5 + (int64_t)g0 * 5;   // [k][5]
    if (c->dpn == 1)      // scalar solver: v.M v with M = the MINV slot
      hipLaunchKernelGGL(k_post_sums<1>, dim3(nblocks, (kg + POST_MB - 1) / POST_MB), dim3(256), 0, st, N, kg, nblocks, c->d_rowptr, c->d_colind,
                         c->d_vals[PLFEM_BLK_MINV], c->d_vals[PLFEM_BLK_DXY], c->d_vals[PLFEM_BLK_DYY], c->d_coremask, ev,
                         partial);
    else
      hipLaunchKernelGGL(k_post_sums<2>, dim3(nblocks, (kg + POST_MB - 1) / POST_MB), dim3(256), 0, st, N, kg, nblocks, c->d_rowptr, c->d_colind,
                         c->d_vals[PLFEM_BLK_DXX], c->d_vals[PLFEM_BLK_DXY], c->d_vals[PLFEM_BLK_DYY], c->d_coremask, ev,
                         partial);
    hipLaunchKernelGGL(k_post_finish, dim3(kg * 5), dim3(64), 0, st, kg, nblocks, partial, sums);
    hipLaunchKernelGGL(k_post_scale, dim3((unsigned)((c->n2 + 255) / 256), kg), dim3(256), 0, st, N, c->dpn, sums, ev);
    if (modes_int)
      hipLaunchKernelGGL(k_gather_interior, dim3((unsigned)((c->dpn * (int64_t)c->nsolve + 255) / 256), kg), dim3(256), 0, st,
                         N, c->nsolve, c->dpn, c->d_interior, ev, modes_int + (int64_t)g0 * c->dpn * c->nsolve);
    if (group_done) (*group_done)(g0, kg);
  }
  // results to the host
  double* hs = c->h_pinned;
  (void)hipMemcpyAsync(hs, c->d_post + (int64_t)k * nblocks * 5, sizeof(double) * k * 5, hipMemcpyDeviceToHost, st);
  int32_t* hc = reinterpret_cast<int32_t*>(c->h_pinned + 4096);
  (void)hipMemcpyAsync(hc, c->d_counters, sizeof(int32_t) * 4, hipMemcpyDeviceToHost, st);
}

// after the stream has been synchronised
void post_finish(plfem_ctx* c, int k, double* out_host, double* frac_core) {
  const double* hs = c->h_pinned;
  const int32_t* hc = reinterpret_cast<const int32_t*>(c->h_pinned + 4096);
  for (int mode = 0; mode < k; ++mode) {
    const double* s = hs + mode * 5;
    double nrm2 = c->dpn == 2 ? s[0] + s[1] : s[4];      // scalar solver: M-norm (solver_fem.py:268)
    double nrm = std::sqrt(nrm2) + 1e-30;
    double inv2 = 1.0 / (nrm * nrm);
    double* o = out_host + (size_t)mode * PLFEM_POST_COUNT;
    o[PLFEM_POST_NORM] = nrm;
    o[PLFEM_POST_DIV_ENERGY] = s[4] * inv2;
    o[PLFEM_POST_CORE_X] = s[2] * inv2;
    o[PLFEM_POST_CORE_Y] = s[3] * inv2;
    o[PLFEM_POST_ALL_X] = s[0] * inv2;
    o[PLFEM_POST_ALL_Y] = s[1] * inv2;
  }
  if (frac_core) *frac_core = (double)hc[1] / (double)c->nsolve;
}

void launch_post(plfem_ctx* c, int k, double* evecs, int ncore, double* out_host, double* frac_core,
                 double* modes_int) {
  post_enqueue(c, k, evecs, ncore, modes_int, nullptr);
  (void)hipStreamSynchronize(c->stream);
  post_finish(c, k, out_host, frac_core);
}

}  // namespace plfem
