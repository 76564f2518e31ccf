// gfx950 kernels: multifrontal shift-invert operator on the nested-dissection front tree.
//
// Replaces splu((A - sigma B).tocsc()) and lu.solve() that the reference reaches through
// eigsh(..., sigma=...) (reference solver_fem.py:197 -> scipy arpack.py:915, 920-928).
//
// Every front F (order m, column major, symmetric) holds [F11 F12; F21 F22] with the first s2 DOFs
// fully summed.  The factorisation applies the symmetric sweep operator to the first s2 indices,
// NB columns at a time (block Gauss-Jordan with partial pivoting inside each NB x NB pivot block):
//     F  ->  [ -F11^-1      F11^-1 F12 ;  F21 F11^-1     F22 - F21 F11^-1 F12 ]
// i.e. the explicit inverse of the pivot block, the coupling panel X and the Schur complement that
// the parent front gathers.  With explicit inverses both solve sweeps are batched dense
// column-times-vector products (no triangular dependency chains inside a front).
#include "device.h"

namespace plfem {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// front assembly
// ------------------------------------------------------------------------------------------------
// Leaf fronts: K_e = A_e - sigma B_e of the leaf's own elements, added element by element (fixed
// order, one block per front, 144 lanes per element) -- element-based multifrontal assembly.
__global__ __launch_bounds__(256) void k_leaf_assemble(
    int first_front, int ne, double sigma, const int32_t* __restrict__ fm, const int64_t* __restrict__ foff,
    const int64_t* __restrict__ fnode_ptr, const int32_t* __restrict__ fnodes,
    const int32_t* __restrict__ leaf_elem_ptr, const int32_t* __restrict__ leaf_elems,
    const int32_t* __restrict__ epos, const double* __restrict__ elem, double* __restrict__ front) {
  const int lf = blockIdx.x;
  const int f = first_front + lf;
  const int m = fm[f];
  double* F = front + foff[f];
  const int tid = threadIdx.x;
  const int64_t mm = (int64_t)m * m;
  for (int64_t k = tid; k < mm; k += 256) F[k] = 0.0;
  __syncthreads();
  const int32_t* fn = fnodes + fnode_ptr[f];
  for (int q = tid; q < m / 2; q += 256)
    if (fn[q] < 0) {   // padding node: unit pivot, no coupling
      F[(int64_t)(2 * q) * m + 2 * q] = 1.0;
      F[(int64_t)(2 * q + 1) * m + 2 * q + 1] = 1.0;
    }
  __syncthreads();
  const int adof = tid / 12, bdof = tid % 12;
  const int a = adof >> 1, ca = adof & 1, b = bdof >> 1, cb = bdof & 1;
  for (int q = leaf_elem_ptr[lf]; q < leaf_elem_ptr[lf + 1]; ++q) {
    const int e = leaf_elems[q];
    if (tid < 144) {
      int pa = epos[(size_t)a * ne + e], pb = epos[(size_t)b * ne + e];
      if (pa >= 0 && pb >= 0) {
        const double* em = elem + (size_t)e * ELEM_STRIDE + a * 6 + b;
        double v;
        if (ca == 0 && cb == 0) v = em[PLFEM_BLK_AXX * 36] - sigma * em[PLFEM_BLK_MINV * 36];
        else if (ca == 0 && cb == 1) v = em[PLFEM_BLK_AXY * 36];
        else if (ca == 1 && cb == 0) v = em[PLFEM_BLK_AYX * 36];
        else v = em[PLFEM_BLK_AYY * 36] - sigma * em[PLFEM_BLK_MINV * 36];
        F[(int64_t)(2 * pb + cb) * m + (2 * pa + ca)] += v;
      }
    }
    __syncthreads();
  }
}

// Internal fronts: gather formulation of the extend-add.  F[i,j] = S_child0[.,.] + S_child1[.,.]
// through the inverse index maps; every entry written exactly once (no zero fill, no atomics).
__global__ __launch_bounds__(256) void k_front_gather(
    int first_front, const int32_t* __restrict__ fs2, const int32_t* __restrict__ fm,
    const int64_t* __restrict__ foff, const int64_t* __restrict__ fnode_ptr, const int32_t* __restrict__ fnodes,
    const int32_t* __restrict__ cinv0, const int32_t* __restrict__ cinv1, double* __restrict__ front) {
  const int f = first_front + blockIdx.z;
  const int m = fm[f];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 16;
  if (i >= m || j0 >= m) return;
  const int64_t np = fnode_ptr[f];
  const int qi = i >> 1, ci = i & 1;
  const int c0i = cinv0[np + qi], c1i = cinv1[np + qi];
  const bool dummy_i = fnodes[np + qi] < 0;
  const int ch0 = 2 * f + 1, ch1 = 2 * f + 2;
  const int m0 = fm[ch0], s0 = fs2[ch0], m1 = fm[ch1], s1 = fs2[ch1];
  const double* F0 = front + foff[ch0];
  const double* F1 = front + foff[ch1];
  double* F = front + foff[f];
  for (int jj = 0; jj < 16; ++jj) {
    int j = j0 + jj;
    int qj = j >> 1, cj = j & 1;
    int c0j = cinv0[np + qj], c1j = cinv1[np + qj];
    double v = 0.0;
    if (c0i >= 0 && c0j >= 0) v += F0[(int64_t)(s0 + 2 * c0j + cj) * m0 + (s0 + 2 * c0i + ci)];
    if (c1i >= 0 && c1j >= 0) v += F1[(int64_t)(s1 + 2 * c1j + cj) * m1 + (s1 + 2 * c1i + ci)];
    if (dummy_i && i == j) v = 1.0;
    F[(int64_t)j * m + i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// block Gauss-Jordan sweep, step kb of a level:  diag -> panel -> update
// ------------------------------------------------------------------------------------------------
// Explicit inverse of the NB x NB pivot block (in-place Gauss-Jordan, partial pivoting restricted
// to the block, static perturbation of vanishing pivots).
__global__ __launch_bounds__(256) void k_sweep_diag(int first_front, int kb, const int32_t* __restrict__ fs2,
                                                    const int32_t* __restrict__ fm, const int64_t* __restrict__ foff,
                                                    const double* __restrict__ front, double* __restrict__ dinv,
                                                    int32_t* __restrict__ counters) {
  const int f = first_front + blockIdx.x;
  const int s2 = fs2[f];
  const int k0 = kb * NB;
  if (k0 >= s2) return;
  const int nbk = min(NB, s2 - k0);
  const int m = fm[f];
  const double* F = front + foff[f];
  __shared__ double a[NB][NB + 1];
  __shared__ int piv[NB];
  __shared__ double s_red[4];
  const int tid = threadIdx.x;
  double amax = 0.0;
  for (int k = tid; k < NB * NB; k += 256) {
    int r = k % NB, c = k / NB;
    double v = (r < nbk && c < nbk) ? F[(int64_t)(k0 + c) * m + (k0 + r)] : (r == c ? 1.0 : 0.0);
    a[r][c] = v;
    amax = fmax(amax, fabs(v));
  }
  for (int off = 32; off >= 1; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off));
  if ((tid & 63) == 0) s_red[tid >> 6] = amax;
  __syncthreads();
  amax = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  const double thr = 1e-13 * amax;
  const int i = tid >> 3, cg = tid & 7;   // elimination: row i, columns cg*4 .. cg*4+3
  for (int k = 0; k < NB; ++k) {
    if (tid < 64) {
      double v = (tid < NB && tid >= k) ? fabs(a[tid][k]) : -1.0;
      int idx = tid;
      for (int off = 32; off >= 1; off >>= 1) {
        double ov = __shfl_xor(v, off);
        int oi = __shfl_xor(idx, off);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
      }
      if (tid == 0) piv[k] = idx;
    }
    __syncthreads();
    const int p = piv[k];
    if (p != k && tid < NB) {
      double t = a[k][tid];
      a[k][tid] = a[p][tid];
      a[p][tid] = t;
    }
    __syncthreads();
    double pv = a[k][k];
    if (fabs(pv) < thr || pv == 0.0) {
      pv = (pv < 0.0) ? -fmax(thr, 1e-300) : fmax(thr, 1e-300);
      if (tid == 0) atomicAdd(&counters[0], 1);
    }
    const double f_ik = a[i][k];
    __syncthreads();
    if (tid < NB) a[k][tid] = ((tid == k) ? 1.0 : a[k][tid]) / pv;
    __syncthreads();
    if (i != k) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        int c = cg * 4 + cc;
        double v = (c == k) ? 0.0 : a[i][c];
        a[i][c] = v - f_ik * a[k][c];
      }
    }
    __syncthreads();
  }
  for (int k = NB - 1; k >= 0; --k) {
    const int p = piv[k];
    if (p != k && tid < NB) {
      double t = a[tid][k];
      a[tid][k] = a[tid][p];
      a[tid][p] = t;
    }
    __syncthreads();
  }
  double* D = dinv + (int64_t)f * NB * NB;
  for (int k = tid; k < NB * NB; k += 256) D[k] = a[k % NB][k / NB];
}

// Panel: W = R * Dinv for all rows outside the pivot block, R = F[:, pivot columns].  Saves R and W
// for the update kernel and writes the swept pivot columns / rows back into F.
__global__ __launch_bounds__(256) void k_sweep_panel(int first_front, int kb, const int32_t* __restrict__ fs2,
                                                     const int32_t* __restrict__ fm, const int64_t* __restrict__ foff,
                                                     const int64_t* __restrict__ fnode_ptr, double* __restrict__ front,
                                                     const double* __restrict__ dinv, double* __restrict__ wbuf,
                                                     double* __restrict__ rbuf) {
  const int f = first_front + blockIdx.y;
  const int s2 = fs2[f];
  const int k0 = kb * NB;
  if (k0 >= s2) return;
  const int m = fm[f];
  const int i0 = blockIdx.x * 64;
  if (i0 >= m) return;
  const int nbk = min(NB, s2 - k0);
  double* F = front + foff[f];
  const double* D = dinv + (int64_t)f * NB * NB;
  double* W = wbuf + 2 * fnode_ptr[f] * NB;
  double* R = rbuf + 2 * fnode_ptr[f] * NB;
  __shared__ double sD[NB][NB + 1];
  __shared__ double sR[64][NB + 1];
  const int tid = threadIdx.x;
  for (int k = tid; k < NB * NB; k += 256) sD[k % NB][k / NB] = D[k];
  const int r = tid & 63, cq = tid >> 6;   // row r, column group cq (8 columns)
  const int i = i0 + r;
  const bool valid = i < m;
  const bool inpiv = valid && i >= k0 && i < k0 + nbk;
  for (int cc = 0; cc < 8; ++cc) {
    int c = cq * 8 + cc;
    sR[r][c] = (valid && !inpiv && c < nbk) ? F[(int64_t)(k0 + c) * m + i] : 0.0;
  }
  __syncthreads();
  if (!valid) return;
  for (int cc = 0; cc < 8; ++cc) {
    int c = cq * 8 + cc;
    if (c >= nbk) {
      W[(int64_t)c * m + i] = 0.0;
      R[(int64_t)c * m + i] = 0.0;
      continue;
    }
    double w = 0.0;
    if (!inpiv) {
#pragma unroll 8
      for (int k = 0; k < NB; ++k) w += sR[r][k] * sD[k][c];
    }
    W[(int64_t)c * m + i] = w;
    R[(int64_t)c * m + i] = sR[r][c];
    if (inpiv) {
      F[(int64_t)(k0 + c) * m + i] = -sD[i - k0][c];
    } else {
      F[(int64_t)(k0 + c) * m + i] = w;
      F[(int64_t)i * m + (k0 + c)] = w;
    }
  }
}

// Update: F[i,j] -= sum_k W[i,k] R[j,k] for all i, j outside the pivot block, one 32x32 tile per
// wave as 2x2 v_mfma_f64_16x16x4_f64 tiles.  MFMA operand map (gfx950): A[row = l&15][k = l>>4],
// B[k = l>>4][col = l&15], D[row = (l>>4) + 4 r][col = l&15].  With A <- R rows (j) and B <- W rows
// (i) the accumulator register r of lane l is F[i0 + (l&15), j0 + (l>>4) + 4 r]: 128-B runs.
__global__ __launch_bounds__(256) void k_sweep_update(int first_front, int kb, const int32_t* __restrict__ fs2,
                                                      const int32_t* __restrict__ fm, const int64_t* __restrict__ foff,
                                                      const int64_t* __restrict__ fnode_ptr, double* __restrict__ front,
                                                      const double* __restrict__ wbuf, const double* __restrict__ rbuf) {
  const int f = first_front + blockIdx.z;
  const int s2 = fs2[f];
  const int k0 = kb * NB;
  if (k0 >= s2) return;
  const int m = fm[f];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i0 = (blockIdx.x * 2 + (wave & 1)) * 32;
  const int j0 = (blockIdx.y * 2 + (wave >> 1)) * 32;
  if (i0 >= m || j0 >= m) return;
  const int nbk = min(NB, s2 - k0);
  double* F = front + foff[f];
  const double* W = wbuf + 2 * fnode_ptr[f] * NB;
  const double* R = rbuf + 2 * fnode_ptr[f] * NB;
  const int lr = lane & 15, lk = lane >> 4;
  // 16-wide sub-tiles that are entirely inside the pivot block, or beyond m, are skipped
  bool iv[2], jv[2];
  for (int t = 0; t < 2; ++t) {
    int ii = i0 + 16 * t, jj = j0 + 16 * t;
    iv[t] = ii < m && !(ii >= k0 && ii < k0 + nbk);
    jv[t] = jj < m && !(jj >= k0 && jj < k0 + nbk);
  }
  if (!((iv[0] || iv[1]) && (jv[0] || jv[1]))) return;
  v4d acc[2][2];
  for (int tj = 0; tj < 2; ++tj)
    for (int ti = 0; ti < 2; ++ti) acc[tj][ti] = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < nbk; kk += 4) {
    const int64_t col = (int64_t)(kk + lk) * m;
    double a0 = jv[0] ? R[col + j0 + lr] : 0.0;
    double a1 = jv[1] ? R[col + j0 + 16 + lr] : 0.0;
    double b0 = iv[0] ? W[col + i0 + lr] : 0.0;
    double b1 = iv[1] ? W[col + i0 + 16 + lr] : 0.0;
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
  for (int tj = 0; tj < 2; ++tj) {
    if (!jv[tj]) continue;
    for (int ti = 0; ti < 2; ++ti) {
      if (!iv[ti]) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int64_t idx = (int64_t)(j0 + 16 * tj + lk + 4 * r) * m + (i0 + 16 * ti + lr);
        F[idx] -= acc[tj][ti][r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// solve sweeps
// ------------------------------------------------------------------------------------------------
// forward, step 1: local right-hand side = global rhs at owned DOFs + children's updates
__global__ __launch_bounds__(256) void k_fwd_gather(int first_front, int N, int leaf_level,
                                                    const int32_t* __restrict__ fs2, const int32_t* __restrict__ fm,
                                                    const int64_t* __restrict__ fnode_ptr,
                                                    const int32_t* __restrict__ fnodes, const int32_t* __restrict__ cinv0,
                                                    const int32_t* __restrict__ cinv1, const double* __restrict__ rhs,
                                                    double* __restrict__ fvec) {
  const int f = first_front + blockIdx.y;
  const int m = fm[f];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const int64_t np = fnode_ptr[f];
  const int q = i >> 1, c = i & 1;
  const int node = fnodes[np + q];
  double v = (i < fs2[f] && node >= 0) ? rhs[(int64_t)c * N + node] : 0.0;
  if (!leaf_level) {
    int c0 = cinv0[np + q], c1 = cinv1[np + q];
    if (c0 >= 0) { int ch = 2 * f + 1; v += fvec[2 * fnode_ptr[ch] + fs2[ch] + 2 * c0 + c]; }
    if (c1 >= 0) { int ch = 2 * f + 2; v += fvec[2 * fnode_ptr[ch] + fs2[ch] + 2 * c1 + c]; }
  }
  fvec[2 * np + i] = v;
}

// forward, step 2: u = w_b - X^T z, one wave per boundary DOF (column of X)
__global__ __launch_bounds__(256) void k_fwd_gemv(int first_front, const int32_t* __restrict__ fs2,
                                                  const int32_t* __restrict__ fm, const int64_t* __restrict__ foff,
                                                  const int64_t* __restrict__ fnode_ptr, const double* __restrict__ front,
                                                  double* __restrict__ fvec) {
  const int f = first_front + blockIdx.y;
  const int m = fm[f], s2 = fs2[f];
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s2 + j >= m) return;
  const int lane = threadIdx.x & 63;
  const double* col = front + foff[f] + (int64_t)(s2 + j) * m;
  double* w = fvec + 2 * fnode_ptr[f];
  double acc = 0.0;
  for (int i = lane; i < s2; i += 64) acc += col[i] * w[i];
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) w[s2 + j] -= acc;
}

// backward: x_own = -[F11' ; X^T]^T [z ; x_b] = F11^-1 z - X x_b, one wave per owned DOF
__global__ __launch_bounds__(256) void k_bwd_gemv(int first_front, int N, const int32_t* __restrict__ fs2,
                                                  const int32_t* __restrict__ fm, const int64_t* __restrict__ foff,
                                                  const int64_t* __restrict__ fnode_ptr,
                                                  const int32_t* __restrict__ fnodes, const double* __restrict__ front,
                                                  const double* __restrict__ fvec, double* __restrict__ x) {
  const int f = first_front + blockIdx.y;
  const int m = fm[f], s2 = fs2[f];
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= s2) return;
  const int64_t np = fnode_ptr[f];
  const int node_i = fnodes[np + (i >> 1)];
  if (node_i < 0) return;
  const int lane = threadIdx.x & 63;
  const double* col = front + foff[f] + (int64_t)i * m;
  const double* w = fvec + 2 * np;
  double acc = 0.0;
  for (int j = lane; j < m; j += 64) {
    double v;
    if (j < s2) v = w[j];
    else {
      int node = fnodes[np + (j >> 1)];
      v = node >= 0 ? x[(int64_t)(j & 1) * N + node] : 0.0;
    }
    acc += col[j] * v;
  }
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) x[(int64_t)(i & 1) * N + node_i] = -acc;
}

}  // namespace

void launch_factor(plfem_ctx* c, double sigma) {
  hipStream_t st = c->stream;
  hipMemsetAsync(c->d_counters, 0, 4 * sizeof(int32_t), st);
  for (int lev = c->L; lev >= 0; --lev) {
    const LevelInfo& li = c->levels[lev];
    if (lev == c->L) {
      hipLaunchKernelGGL(k_leaf_assemble, dim3(li.count), dim3(256), 0, st, li.first, c->ne, sigma, c->d_fm, c->d_foff,
                         c->d_fnode_ptr, c->d_fnodes, c->d_leaf_elem_ptr, c->d_leaf_elems, c->d_epos, c->d_elem,
                         c->d_front);
    } else {
      dim3 grid((li.max_m + 63) / 64, (li.max_m + 63) / 64, li.count);
      hipLaunchKernelGGL(k_front_gather, grid, dim3(256), 0, st, li.first, c->d_fs2, c->d_fm, c->d_foff,
                         c->d_fnode_ptr, c->d_fnodes, c->d_cinv0, c->d_cinv1, c->d_front);
    }
    const int steps = (li.max_s2 + NB - 1) / NB;
    for (int kb = 0; kb < steps; ++kb) {
      hipLaunchKernelGGL(k_sweep_diag, dim3(li.count), dim3(256), 0, st, li.first, kb, c->d_fs2, c->d_fm, c->d_foff,
                         c->d_front, c->d_dinv, c->d_counters);
      hipLaunchKernelGGL(k_sweep_panel, dim3((li.max_m + 63) / 64, li.count), dim3(256), 0, st, li.first, kb,
                         c->d_fs2, c->d_fm, c->d_foff, c->d_fnode_ptr, c->d_front, c->d_dinv, c->d_wbuf, c->d_rbuf);
      dim3 ug((li.max_m + 63) / 64, (li.max_m + 63) / 64, li.count);
      hipLaunchKernelGGL(k_sweep_update, ug, dim3(256), 0, st, li.first, kb, c->d_fs2, c->d_fm, c->d_foff,
                         c->d_fnode_ptr, c->d_front, c->d_wbuf, c->d_rbuf);
    }
  }
}

void launch_solve(plfem_ctx* c, const double* rhs, double* x) {
  hipStream_t st = c->stream;
  hipMemsetAsync(x, 0, sizeof(double) * c->n2, st);
  for (int lev = c->L; lev >= 0; --lev) {
    const LevelInfo& li = c->levels[lev];
    hipLaunchKernelGGL(k_fwd_gather, dim3((li.max_m + 255) / 256, li.count), dim3(256), 0, st, li.first, c->N,
                       lev == c->L ? 1 : 0, c->d_fs2, c->d_fm, c->d_fnode_ptr, c->d_fnodes, c->d_cinv0, c->d_cinv1,
                       rhs, c->d_fvec);
    if (li.max_b2 > 0)
      hipLaunchKernelGGL(k_fwd_gemv, dim3((li.max_b2 + 3) / 4, li.count), dim3(256), 0, st, li.first, c->d_fs2,
                         c->d_fm, c->d_foff, c->d_fnode_ptr, c->d_front, c->d_fvec);
  }
  for (int lev = 0; lev <= c->L; ++lev) {
    const LevelInfo& li = c->levels[lev];
    if (li.max_s2 > 0)
      hipLaunchKernelGGL(k_bwd_gemv, dim3((li.max_s2 + 3) / 4, li.count), dim3(256), 0, st, li.first, c->N, c->d_fs2,
                         c->d_fm, c->d_foff, c->d_fnode_ptr, c->d_fnodes, c->d_front, c->d_fvec, x);
  }
}

}  // namespace plfem
