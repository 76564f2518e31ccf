// gfx950 kernels: P2 element-batch bilinear forms, ordered COO->CSR gather, block CSR SpMV.
//
// Replaces the nine scikit-fem BilinearForm closures + asm() calls of the reference
// (solver_fem.py:131-156), geometry.epsilon at the quadrature points (geometry_unified.py:325-336,
// real part only) and the block build solver_fem.py:158-167.
#include "device.h"

namespace plfem {
namespace {

// 6-point degree-4 rule on the reference triangle (weights sum to 1/2) — scikit-fem's default
// intorder = 2*maxdeg = 4 for ElementTriP2.
__constant__ double c_qx[6] = {0.445948490915965, 0.10810301816807, 0.445948490915965,
                               0.091576213509771, 0.816847572980458, 0.091576213509771};
__constant__ double c_qy[6] = {0.445948490915965, 0.445948490915965, 0.10810301816807,
                               0.091576213509771, 0.091576213509771, 0.816847572980458};
__constant__ double c_qw[6] = {0.1116907948390055, 0.1116907948390055, 0.1116907948390055,
                               0.054975871827661, 0.054975871827661, 0.054975871827661};

constexpr int EPB = 7;   // elements per 256-thread block: 7*36 = 252 (i,j) pairs

// One block = EPB elements.  Phase 1 (element, quadrature point) lanes stage the physical
// gradients of the six P2 basis functions and the two quadrature weights (plain and 1/eps) in LDS;
// phase 2 (element, local pair) lanes reduce the six quadrature points into the eight element
// matrices the pencil needs and store them as contiguous 288-B runs.
__global__ __launch_bounds__(256) void k_element_matrices(
    int ne, int N, const int32_t* __restrict__ tsorted, const double* __restrict__ doflocs,
    const double* __restrict__ cores, int ncore, double inv_eps_core, double inv_eps_clad, double k0sq,
    double alpha_p, int scalar, double* __restrict__ elem) {
  __shared__ double s_phi[6][6];            // [basis][qp]
  __shared__ double s_gx[EPB][6][6];        // [el][basis][qp]
  __shared__ double s_gy[EPB][6][6];
  __shared__ double s_w1[EPB][6];
  __shared__ double s_we[EPB][6];
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * EPB;
  if (tid < 36) {
    int i = tid / 6, q = tid % 6;
    double x = c_qx[q], y = c_qy[q], v;
    switch (i) {
      case 0: v = 1 - 3 * x - 3 * y + 2 * x * x + 4 * x * y + 2 * y * y; break;
      case 1: v = 2 * x * x - x; break;
      case 2: v = 2 * y * y - y; break;
      case 3: v = 4 * x - 4 * x * x - 4 * x * y; break;
      case 4: v = 4 * x * y; break;
      default: v = 4 * y - 4 * x * y - 4 * y * y; break;
    }
    s_phi[i][q] = v;
  }
  if (tid < EPB * 6) {
    int el = tid / 6, q = tid % 6;
    int e = e0 + el;
    if (e < ne) {
      int v0 = tsorted[e], v1 = tsorted[ne + e], v2 = tsorted[2 * (size_t)ne + e];
      double x0 = doflocs[v0], y0 = doflocs[N + v0];
      double j00 = doflocs[v1] - x0, j10 = doflocs[N + v1] - y0;      // J = [p1-p0, p2-p0]
      double j01 = doflocs[v2] - x0, j11 = doflocs[N + v2] - y0;
      // det J of a sliver element cancels to ~1e-8 of its terms: keep the two products individually
      // rounded (as the reference's NumPy arithmetic does); a fused multiply-add here changes 1e9-sized
      // element entries at the 1e-8 relative level.  hipcc's default -ffp-contract=fast ignores the
      // contract pragma, hence the opaque multiplies.
      double t1, t2;
      asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t1) : "v"(j00), "v"(j11));
      asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t2) : "v"(j01), "v"(j10));
      double det = t1 - t2;
      double idet = 1.0 / det;
      // J^-1 = 1/det [[j11, -j01], [-j10, j00]];  grad = J^-T grad_hat
      double i00 = j11 * idet, i01 = -j01 * idet, i10 = -j10 * idet, i11 = j00 * idet;
      double xi = c_qx[q], eta = c_qy[q];
      double X = x0 + j00 * xi + j01 * eta, Y = y0 + j10 * xi + j11 * eta;
      bool in_core = false;
      for (int c = 0; c < ncore; ++c) {
        double dx = X - cores[3 * c], dy = Y - cores[3 * c + 1], r = cores[3 * c + 2];
        in_core |= (dx * dx + dy * dy <= r * r);
      }
      double w1 = fabs(det) * c_qw[q];
      s_w1[el][q] = w1;
      s_we[el][q] = w1 * (in_core ? inv_eps_core : inv_eps_clad);
      double dxh[6] = {-3 + 4 * xi + 4 * eta, 4 * xi - 1, 0.0, 4 - 8 * xi - 4 * eta, 4 * eta, -4 * eta};
      double dyh[6] = {-3 + 4 * xi + 4 * eta, 0.0, 4 * eta - 1, -4 * xi, 4 * xi, 4 - 4 * xi - 8 * eta};
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        s_gx[el][i][q] = i00 * dxh[i] + i10 * dyh[i];
        s_gy[el][i][q] = i01 * dxh[i] + i11 * dyh[i];
      }
    }
  }
  __syncthreads();
  if (tid < EPB * 36) {
    int el = tid / 36, ab = tid % 36;
    int e = e0 + el;
    if (e < ne) {
      int a = ab / 6, b = ab % 6;   // a = test / row i, b = trial / column j
      double m1 = 0, me = 0, xx1 = 0, yy1 = 0, xy1 = 0, yx1 = 0, xxe = 0, yye = 0, xye = 0, yxe = 0;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        double w1 = s_w1[el][q], we = s_we[el][q];
        double pa = s_phi[a][q], pb = s_phi[b][q];
        double gxa = s_gx[el][a][q], gya = s_gy[el][a][q];
        double gxb = s_gx[el][b][q], gyb = s_gy[el][b][q];
        double pp = pa * pb;
        m1 += pp * w1;
        me += pp * we;
        double txx = gxb * gxa, tyy = gyb * gya, txy = gxb * gya, tyx = gyb * gxa;   // d?phi_b(trial) d?phi_a(test)
        xx1 += txx * w1; yy1 += tyy * w1; xy1 += txy * w1; yx1 += tyx * w1;
        xxe += txx * we; yye += tyy * we; xye += txy * we; yxe += tyx * we;
      }
      double* o = elem + (size_t)e * ELEM_STRIDE + ab;
      if (scalar) {
        // scalar Helmholtz pencil of the reference's ScalarHelmholtzSolver (solver_fem.py:252-259): the second weight is
        // eps (not 1/eps) here, so me = eps_m;  AXX slot <- stiff - k0^2 eps_m,  MINV slot <- mass
        o[PLFEM_BLK_AXX * 36] = (xx1 + yy1) - k0sq * me;
        o[PLFEM_BLK_AXY * 36] = 0.0;
        o[PLFEM_BLK_AYX * 36] = 0.0;
        o[PLFEM_BLK_AYY * 36] = 0.0;
        o[PLFEM_BLK_MINV * 36] = scalar == 2 ? me : m1;        // 2: only the weighted mass is wanted (CMT coupling)
        o[PLFEM_BLK_DXX * 36] = xx1;
        o[PLFEM_BLK_DXY * 36] = xy1;
        o[PLFEM_BLK_DYY * 36] = yy1;
        return;
      }
      // kxx = e^-1 u_y v_y, kyy = e^-1 u_x v_x, kxy = -e^-1 u_y v_x, kyx = -e^-1 u_x v_y  (solver_fem.py:132-138)
      // div_xx = u_x v_x, div_yy = u_y v_y, div_xy = u_x v_y                                (solver_fem.py:141-145)
      o[PLFEM_BLK_AXX * 36] = yye + alpha_p * xx1 - k0sq * m1;     // Kxx + a Dxx - k0^2 M
      o[PLFEM_BLK_AXY * 36] = -yxe + alpha_p * xy1;                // Kxy + a Dxy
      o[PLFEM_BLK_AYX * 36] = -xye + alpha_p * yx1;                // Kyx + a Dxy^T
      o[PLFEM_BLK_AYY * 36] = xxe + alpha_p * yy1 - k0sq * m1;     // Kyy + a Dyy - k0^2 M
      o[PLFEM_BLK_MINV * 36] = me;
      o[PLFEM_BLK_DXX * 36] = xx1;
      o[PLFEM_BLK_DXY * 36] = xy1;
      o[PLFEM_BLK_DYY * 36] = yy1;
    }
  }
}

// Number of quadrature points inside a core (same point and same closed-disc test as k_element_matrices): the
// unweighted mean permittivity over all quadrature points that the reference's CMT form subtracts (config.py:297-300)
// follows from it in closed form.  Integer atomics: order independent.
__global__ __launch_bounds__(256) void k_count_core_qp(int ne, int N, const int32_t* __restrict__ tsorted,
                                                       const double* __restrict__ doflocs, const double* __restrict__ cores,
                                                       int ncore, unsigned long long* __restrict__ count) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool in_core = false;
  if (t < (int64_t)ne * 6) {
    const int e = (int)(t / 6), q = (int)(t % 6);
    int v0 = tsorted[e], v1 = tsorted[ne + e], v2 = tsorted[2 * (size_t)ne + e];
    double x0 = doflocs[v0], y0 = doflocs[N + v0];
    double j00 = doflocs[v1] - x0, j10 = doflocs[N + v1] - y0;
    double j01 = doflocs[v2] - x0, j11 = doflocs[N + v2] - y0;
    double xi = c_qx[q], eta = c_qy[q];
    double X = x0 + j00 * xi + j01 * eta, Y = y0 + j10 * xi + j11 * eta;
    for (int c = 0; c < ncore; ++c) {
      double dx = X - cores[3 * c], dy = Y - cores[3 * c + 1], r = cores[3 * c + 2];
      in_core |= (dx * dx + dy * dy <= r * r);
    }
  }
  const unsigned long long b = __ballot(in_core);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (unsigned long long)__popcll(b));
}

// CSR column lists of the scalar pattern: row i = ascending union of the DOFs of the elements adjacent to node i
// (what coo_matrix(...).tocsr() leaves of asm()'s triplets, reference solver_fem.py:153-156).  The row length is
// known (closed form on the host).  Ordinary rows (at most 10 adjacent elements): one lane per row reads its 6 d
// candidates once into an LDS row (stride 65 words: conflict-free) and emits the next larger candidate rowlen
// times.  High-degree rows (the centre vertex of a core's point rings has 16 rho neighbours) are done by the whole
// block: every candidate finds the number of distinct smaller ones and, if it is the first of its value, its slot.
// The slots of the 128 rows of a block are contiguous, so they are collected in LDS and written back coalesced.
constexpr int PATTERN_CAP = 64, PATTERN_OUT = 4096, PATTERN_BIG = 1536;
__global__ __launch_bounds__(128) void k_pattern_fill(int N, int ne, const int32_t* __restrict__ nptr,
                                                      const int32_t* __restrict__ nadj, const int32_t* __restrict__ edof,
                                                      const int32_t* __restrict__ rowptr, int32_t* __restrict__ colind,
                                                      int32_t* __restrict__ slot_row) {
  __shared__ int32_t cand[128][PATTERN_CAP + 1];
  __shared__ int32_t outbuf[PATTERN_OUT];
  __shared__ uint8_t outrow[PATTERN_OUT];
  __shared__ int32_t big[PATTERN_BIG];
  __shared__ uint8_t bigfirst[PATTERN_BIG];
  __shared__ int32_t biglist[128];
  __shared__ int nbig;
  const int i0 = blockIdx.x * 128;
  const int i = i0 + threadIdx.x;
  const int base = rowptr[i0], end = rowptr[min(i0 + 128, N)];
  const bool buffered = end - base <= PATTERN_OUT;
  if (threadIdx.x == 0) nbig = 0;
  __syncthreads();
  auto put = [&](int k, int col, int local_row) {
    if (buffered) { outbuf[k - base] = col; outrow[k - base] = (uint8_t)local_row; }
    else { colind[k] = col; slot_row[k] = i0 + local_row; }
  };
  if (i < N) {
    const int q0 = nptr[i], q1 = nptr[i + 1];
    const int nc = 6 * (q1 - q0);
    int32_t* mine = cand[threadIdx.x];
    if (nc > PATTERN_CAP && nc <= PATTERN_BIG) {
      biglist[atomicAdd(&nbig, 1)] = threadIdx.x;           // order irrelevant: every slot has one writer
    } else {
      const bool staged = nc <= PATTERN_CAP;
      if (staged) {
        for (int q = q0; q < q1; ++q) {
          const int e = nadj[q];
#pragma unroll
          for (int a = 0; a < 6; ++a) mine[6 * (q - q0) + a] = edof[(size_t)a * ne + e];
        }
      }
      int prev = -1;
      for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        int best = 0x7fffffff;
        if (staged) {
          for (int j = 0; j < nc; ++j) {
            const int v = mine[j];
            best = (v > prev && v < best) ? v : best;
          }
        } else {                                            // more than PATTERN_BIG candidates: slow but correct
          for (int q = q0; q < q1; ++q) {
            const int e = nadj[q];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
              const int v = edof[(size_t)a * ne + e];
              best = (v > prev && v < best) ? v : best;
            }
          }
        }
        put(k, best, threadIdx.x);
        prev = best;
      }
    }
  }
  __syncthreads();
  const int nb = nbig;
  for (int b = 0; b < nb; ++b) {
    const int lr = biglist[b], r = i0 + lr;
    const int q0 = nptr[r], nc = 6 * (nptr[r + 1] - q0);
    for (int j = threadIdx.x; j < nc; j += 128) big[j] = edof[(size_t)(j % 6) * ne + nadj[q0 + j / 6]];
    __syncthreads();
    for (int j = threadIdx.x; j < nc; j += 128) {
      const int v = big[j];
      bool first = true;
      for (int m = 0; m < j; ++m) first = first && big[m] != v;
      bigfirst[j] = first;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nc; j += 128) {
      if (!bigfirst[j]) continue;
      const int v = big[j];
      int rank = 0;
      for (int m = 0; m < nc; ++m) rank += (bigfirst[m] && big[m] < v) ? 1 : 0;
      const int k = rowptr[r] + rank;
      if (k < rowptr[r + 1]) put(k, v, lr);
    }
    __syncthreads();
  }
  if (buffered) {
    for (int k = base + threadIdx.x; k < end; k += 128) {
      colind[k] = outbuf[k - base];
      slot_row[k] = i0 + outrow[k - base];
    }
  }
}

// Ordered gather of the element-matrix entries into the shared scalar CSR pattern: one lane per CSR
// slot (i, j) walks the elements adjacent to node i in ascending order and adds the entry of every
// element that also contains node j -- deterministic, no float atomics, every value written once.
// Replaces coo_matrix(...).tocsr() duplicate summation.
__global__ __launch_bounds__(256) void k_csr_gather(int nnz, int ne, const int32_t* __restrict__ slot_row,
                                                    const int32_t* __restrict__ colind, const int32_t* __restrict__ nptr,
                                                    const int32_t* __restrict__ nadj, const uint8_t* __restrict__ nloc,
                                                    const int32_t* __restrict__ edof, const double* __restrict__ elem,
                                                    double* __restrict__ v0, double* __restrict__ v1,
                                                    double* __restrict__ v2, double* __restrict__ v3,
                                                    double* __restrict__ v4, double* __restrict__ v5,
                                                    double* __restrict__ v6, double* __restrict__ v7) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  const int i = slot_row[k], col = colind[k];
  double acc[ELEM_FORMS] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int q1 = nptr[i + 1];
  for (int q = nptr[i]; q < q1; ++q) {
    const int e = nadj[q];
    int b = -1;
#pragma unroll
    for (int bb = 0; bb < 6; ++bb)
      if (edof[(size_t)bb * ne + e] == col) b = bb;
    if (b < 0) continue;
    const double* p = elem + (size_t)e * ELEM_STRIDE + (int)nloc[q] * 6 + b;
#pragma unroll
    for (int f = 0; f < ELEM_FORMS; ++f) acc[f] += p[f * 36];
  }
  v0[k] = acc[0]; v1[k] = acc[1]; v2[k] = acc[2]; v3[k] = acc[3];
  v4[k] = acc[4]; v5[k] = acc[5]; v6[k] = acc[6]; v7[k] = acc[7];
}

// y = A_int x (which = 0) or y = B_int x (which = 1) on 2N-vectors; Dirichlet rows forced to zero,
// Dirichlet columns are zero in x by construction.  8 lanes per scalar row (avg 11.5 nnz / row),
// both field components of the row computed in the same pass over the shared pattern.
template <int WHICH, int DPN>
__global__ __launch_bounds__(256) void k_spmv(int N, const int32_t* __restrict__ rowptr,
                                              const int32_t* __restrict__ colind, const uint8_t* __restrict__ bmask,
                                              const double* __restrict__ vxx, const double* __restrict__ vxy,
                                              const double* __restrict__ vyx, const double* __restrict__ vyy,
                                              const double* __restrict__ x, double* __restrict__ y) {
  int gt = blockIdx.x * blockDim.x + threadIdx.x;
  int row = gt >> 3, sub = gt & 7;
  double sx = 0, sy = 0;
  if (row < N && !bmask[row]) {
    int q0 = rowptr[row], q1 = rowptr[row + 1];
    for (int q = q0 + sub; q < q1; q += 8) {
      int c = colind[q];
      if (DPN == 1) {                       // scalar pencil: one block (A: the AXX slot, B: the MINV slot)
        sx += vxx[q] * x[c];
        continue;
      }
      double xx = x[c], xy = x[N + c];
      if (WHICH == 0) {
        sx += vxx[q] * xx + vxy[q] * xy;
        sy += vyx[q] * xx + vyy[q] * xy;
      } else {
        double mv = vxx[q];
        sx += mv * xx;
        sy += mv * xy;
      }
    }
  }
#pragma unroll
  for (int off = 4; off >= 1; off >>= 1) {
    sx += __shfl_xor(sx, off, 8);
    sy += __shfl_xor(sy, off, 8);
  }
  if (row < N && sub == 0) {
    y[row] = sx;
    if (DPN == 2) y[N + row] = sy;
  }
}

// y_q = B_int x_q for the P vectors of a block (column q at offset q*ld): the pattern and the Minv
// values are read once for all P vectors.
template <int P, int DPN>
__global__ __launch_bounds__(256) void k_spmv_b_block(int N, int64_t ld, const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ colind,
                                                      const uint8_t* __restrict__ bmask, const double* __restrict__ vm,
                                                      const double* __restrict__ x, double* __restrict__ y) {
  int gt = blockIdx.x * blockDim.x + threadIdx.x;
  int row = gt >> 3, sub = gt & 7;
  double sx[P], sy[P];
#pragma unroll
  for (int q = 0; q < P; ++q) { sx[q] = 0.0; sy[q] = 0.0; }
  if (row < N && !bmask[row]) {
    int q0 = rowptr[row], q1 = rowptr[row + 1];
    for (int k = q0 + sub; k < q1; k += 8) {
      const int c = colind[k];
      const double mv = vm[k];
#pragma unroll
      for (int q = 0; q < P; ++q) {
        sx[q] += mv * x[(int64_t)q * ld + c];
        if (DPN == 2) sy[q] += mv * x[(int64_t)q * ld + N + c];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < P; ++q) {
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1) {
      sx[q] += __shfl_xor(sx[q], off, 8);
      sy[q] += __shfl_xor(sy[q], off, 8);
    }
  }
  if (row < N && sub == 0) {
#pragma unroll
    for (int q = 0; q < P; ++q) {
      y[(int64_t)q * ld + row] = sx[q];
      if (DPN == 2) y[(int64_t)q * ld + N + row] = sy[q];
    }
  }
}

// The same product with the input block INTERLEAVED, xi[(node DPN + component) P + q]: the gather of a matrix entry is
// one 32 P-byte run (64 B for the vectorial pencil) instead of DPN P doubles in DPN P different cache lines.  The
// Lanczos step gets this layout for free from the orthogonalisation kernel that writes the block (k_panel_axpy_p).
// Measured at C1: 23.3 -> 19.5 us per launch (HIP events); requesting a lane's three entries together: no faster.
// GRAM (round 4): the same launch also leaves the chunk partials of the Gram matrix X^T (B X) of the block -- what the
// CholQR of a block Lanczos step needs next and used to get from a panel-dot launch of its own (8 us at its latency
// floor): gram[(p P + q) gridDim.x + blockIdx.x] = sum over the workgroup's 32 rows and the components of x[row, p] (B x)[row, q],
// summed in a fixed order (k_chol_small adds the workgroups' partials).
template <int P, int DPN, bool GRAM>
__global__ __launch_bounds__(256) void k_spmv_b_block_il(int N, int64_t ld, const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ colind,
                                                         const uint8_t* __restrict__ bmask, const double* __restrict__ vm,
                                                         const double* __restrict__ xi, double* __restrict__ y,
                                                         double* __restrict__ gram) {
  int gt = blockIdx.x * blockDim.x + threadIdx.x;
  int row = gt >> 3, sub = gt & 7;
  double s[DPN * P];
#pragma unroll
  for (int q = 0; q < DPN * P; ++q) s[q] = 0.0;
  if (row < N && !bmask[row]) {
    int q0 = rowptr[row], q1 = rowptr[row + 1];
    for (int k = q0 + sub; k < q1; k += 8) {
      const int c = colind[k];
      const double mv = vm[k];
      const double* xc = xi + (int64_t)c * (DPN * P);
#pragma unroll
      for (int q = 0; q < DPN * P; ++q) s[q] += mv * xc[q];
    }
  }
#pragma unroll
  for (int q = 0; q < DPN * P; ++q) {
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1) s[q] += __shfl_xor(s[q], off, 8);
  }
  if (row < N && sub == 0) {
#pragma unroll
    for (int comp = 0; comp < DPN; ++comp)
#pragma unroll
      for (int q = 0; q < P; ++q) y[(int64_t)q * ld + (int64_t)comp * N + row] = s[comp * P + q];
  }
  if (GRAM) {
    constexpr int GE = P * P / 8;                        // Gram entries per sub-lane (P = 4: two)
    static_assert(!GRAM || GE * 8 == P * P, "the 8 lanes of a row share the P x P entries");
    __shared__ double red[4][P * P];
    // every one of the row's 8 lanes holds (B x)[row] now: lane `sub` takes the entries GE sub .. GE sub + GE - 1 of the P x P matrix
    double g[GE];
#pragma unroll
    for (int t = 0; t < GE; ++t) g[t] = 0.0;
    if (row < N) {
      const double* xr = xi + (int64_t)row * (DPN * P);
#pragma unroll
      for (int t = 0; t < GE; ++t) {
        const int idx = GE * sub + t, p = idx / P, q = idx % P;
#pragma unroll
        for (int comp = 0; comp < DPN; ++comp) {
          double sq = s[comp * P];
#pragma unroll
          for (int qq = 1; qq < P; ++qq) sq = (q == qq) ? s[comp * P + qq] : sq;
          g[t] = fma(xr[comp * P + p], sq, g[t]);
        }
      }
    }
    // the 8 rows of a wave, then its 4 waves: fixed order
#pragma unroll
    for (int t = 0; t < GE; ++t) {
      g[t] += __shfl_xor(g[t], 8);
      g[t] += __shfl_xor(g[t], 16);
      g[t] += __shfl_xor(g[t], 32);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 8) {
#pragma unroll
      for (int t = 0; t < GE; ++t) red[wave][GE * lane + t] = g[t];
    }
    __syncthreads();
    if ((int)threadIdx.x < P * P)
      gram[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

// y_q = A_int x_q for the P vectors of a block (iterative refinement of the block Lanczos operator)
template <int P>
__global__ __launch_bounds__(256) void k_spmv_a_block(int N, int64_t ld, const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ colind, const uint8_t* __restrict__ bmask,
                                                      const double* __restrict__ vxx, const double* __restrict__ vxy,
                                                      const double* __restrict__ vyx, const double* __restrict__ vyy,
                                                      const double* __restrict__ x, double* __restrict__ y) {
  int gt = blockIdx.x * blockDim.x + threadIdx.x;
  int row = gt >> 3, sub = gt & 7;
  double sx[P], sy[P];
#pragma unroll
  for (int q = 0; q < P; ++q) { sx[q] = 0.0; sy[q] = 0.0; }
  if (row < N && !bmask[row]) {
    int q0 = rowptr[row], q1 = rowptr[row + 1];
    for (int k = q0 + sub; k < q1; k += 8) {
      const int c = colind[k];
      const double axx = vxx[k], axy = vxy[k], ayx = vyx[k], ayy = vyy[k];
#pragma unroll
      for (int q = 0; q < P; ++q) {
        const double xx = x[(int64_t)q * ld + c], xy = x[(int64_t)q * ld + N + c];
        sx[q] += axx * xx + axy * xy;
        sy[q] += ayx * xx + ayy * xy;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < P; ++q) {
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1) {
      sx[q] += __shfl_xor(sx[q], off, 8);
      sy[q] += __shfl_xor(sy[q], off, 8);
    }
  }
  if (row < N && sub == 0) {
#pragma unroll
    for (int q = 0; q < P; ++q) {
      y[(int64_t)q * ld + row] = sx[q];
      y[(int64_t)q * ld + N + row] = sy[q];
    }
  }
}

}  // namespace

void launch_spmv_a_block(plfem_ctx* c, const double* x, double* y, int64_t ld) {
  int64_t threads = (int64_t)c->N * 8;
  int grid = (int)((threads + 255) / 256);
  if (c->dpn == 1) {       // scalar pencil: A is one block (the AXX slot)
    hipLaunchKernelGGL((k_spmv_b_block<BLOCK_P, 1>), dim3(grid), dim3(256), 0, c->stream, c->N, ld, c->d_rowptr, c->d_colind,
                       c->d_bmask, c->d_vals[PLFEM_BLK_AXX], x, y);
    return;
  }
  hipLaunchKernelGGL(k_spmv_a_block<BLOCK_P>, dim3(grid), dim3(256), 0, c->stream, c->N, ld, c->d_rowptr, c->d_colind,
                     c->d_bmask, c->d_vals[PLFEM_BLK_AXX], c->d_vals[PLFEM_BLK_AXY], c->d_vals[PLFEM_BLK_AYX],
                     c->d_vals[PLFEM_BLK_AYY], x, y);
}

void launch_element_matrices(plfem_ctx* c, int ncore, double eps_core, double eps_clad, double k0, double alpha_p) {
  int grid = (c->ne + EPB - 1) / EPB;
  hipLaunchKernelGGL(k_element_matrices, dim3(grid), dim3(256), 0, c->stream, c->ne, c->N, c->d_tsorted,
                     c->d_doflocs, c->d_cores, ncore, 1.0 / eps_core, 1.0 / eps_clad, k0 * k0, alpha_p, 0, c->d_elem);
}

void launch_element_matrices_scalar(plfem_ctx* c, int ncore, double eps_core, double eps_clad, double k0) {
  int grid = (c->ne + EPB - 1) / EPB;
  hipLaunchKernelGGL(k_element_matrices, dim3(grid), dim3(256), 0, c->stream, c->ne, c->N, c->d_tsorted,
                     c->d_doflocs, c->d_cores, ncore, eps_core, eps_clad, k0 * k0, 0.0, 1, c->d_elem);
}

// M_deps = asm((eps - mean(eps)) u v) into the MINV slot (CMT coupling, config.py:296-303); returns mean(eps)
double launch_delta_eps_mass(plfem_ctx* c, int ncore, double eps_core, double eps_clad) {
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(c->d_counters + 2);   // 8-byte aligned pair of counters
  (void)hipMemsetAsync(cnt, 0, sizeof(unsigned long long), c->stream);
  const int64_t nq = (int64_t)c->ne * 6;
  hipLaunchKernelGGL(k_count_core_qp, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, c->stream, c->ne, c->N, c->d_tsorted,
                     c->d_doflocs, c->d_cores, ncore, cnt);
  unsigned long long* h = reinterpret_cast<unsigned long long*>(c->h_pinned + 4000);
  (void)hipMemcpyAsync(h, cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
  (void)hipStreamSynchronize(c->stream);
  const double frac = (double)*h / (double)nq;
  const double mean = eps_clad + (eps_core - eps_clad) * frac;
  int grid = (c->ne + EPB - 1) / EPB;
  hipLaunchKernelGGL(k_element_matrices, dim3(grid), dim3(256), 0, c->stream, c->ne, c->N, c->d_tsorted,
                     c->d_doflocs, c->d_cores, ncore, eps_core - mean, eps_clad - mean, 0.0, 0.0, 2, c->d_elem);
  launch_csr_gather(c);
  (void)hipMemsetAsync(cnt, 0, sizeof(unsigned long long), c->stream);     // counters[2] is the Lanczos rank flag
  return mean;
}

void launch_pattern_fill(plfem_ctx* c) {
  hipLaunchKernelGGL(k_pattern_fill, dim3((c->N + 127) / 128), dim3(128), 0, c->stream, c->N, c->ne, c->d_nptr, c->d_nadj,
                     c->d_edof, c->d_rowptr, c->d_colind, c->d_slot_row);
}

void launch_csr_gather(plfem_ctx* c) {
  int grid = (c->nnz + 255) / 256;
  hipLaunchKernelGGL(k_csr_gather, dim3(grid), dim3(256), 0, c->stream, c->nnz, c->ne, c->d_slot_row, c->d_colind,
                     c->d_nptr, c->d_nadj, c->d_nloc, c->d_edof, c->d_elem,
                     c->d_vals[0], c->d_vals[1], c->d_vals[2], c->d_vals[3], c->d_vals[4], c->d_vals[5],
                     c->d_vals[6], c->d_vals[7]);
}

void launch_spmv_b_block(plfem_ctx* c, const double* x, double* y, int64_t ld) {
  int64_t threads = (int64_t)c->N * 8;
  int grid = (int)((threads + 255) / 256);
  if (c->dpn == 1)
    hipLaunchKernelGGL((k_spmv_b_block<BLOCK_P, 1>), dim3(grid), dim3(256), 0, c->stream, c->N, ld, c->d_rowptr, c->d_colind,
                       c->d_bmask, c->d_vals[PLFEM_BLK_MINV], x, y);
  else
    hipLaunchKernelGGL((k_spmv_b_block<BLOCK_P, 2>), dim3(grid), dim3(256), 0, c->stream, c->N, ld, c->d_rowptr, c->d_colind,
                       c->d_bmask, c->d_vals[PLFEM_BLK_MINV], x, y);
}

// gram != nullptr: also the Gram partials of the block, gram[(p P + q) nblocks + block]; returns the number of workgroups
// (= partials per entry)
int launch_spmv_b_block_il(plfem_ctx* c, const double* xi, double* y, int64_t ld, double* gram) {
  int64_t threads = (int64_t)c->N * 8;
  int grid = (int)((threads + 255) / 256);
  if (c->dpn == 1) {
    if (gram)
      hipLaunchKernelGGL((k_spmv_b_block_il<BLOCK_P, 1, true>), dim3(grid), dim3(256), 0, c->stream, c->N, ld, c->d_rowptr, c->d_colind,
                         c->d_bmask, c->d_vals[PLFEM_BLK_MINV], xi, y, gram);
    else
      hipLaunchKernelGGL((k_spmv_b_block_il<BLOCK_P, 1, false>), dim3(grid), dim3(256), 0, c->stream, c->N, ld, c->d_rowptr, c->d_colind,
                         c->d_bmask, c->d_vals[PLFEM_BLK_MINV], xi, y, gram);
  } else {
    if (gram)
      hipLaunchKernelGGL((k_spmv_b_block_il<BLOCK_P, 2, true>), dim3(grid), dim3(256), 0, c->stream, c->N, ld, c->d_rowptr, c->d_colind,
                         c->d_bmask, c->d_vals[PLFEM_BLK_MINV], xi, y, gram);
    else
      hipLaunchKernelGGL((k_spmv_b_block_il<BLOCK_P, 2, false>), dim3(grid), dim3(256), 0, c->stream, c->N, ld, c->d_rowptr, c->d_colind,
                         c->d_bmask, c->d_vals[PLFEM_BLK_MINV], xi, y, gram);
  }
  return grid;
}

void launch_spmv(plfem_ctx* c, int which, const double* x, double* y) {
  int64_t threads = (int64_t)c->N * 8;
  int grid = (int)((threads + 255) / 256);
  if (c->dpn == 1) {
    hipLaunchKernelGGL((k_spmv<1, 1>), dim3(grid), dim3(256), 0, c->stream, c->N, c->d_rowptr, c->d_colind, c->d_bmask,
                       c->d_vals[which == 0 ? PLFEM_BLK_AXX : PLFEM_BLK_MINV], nullptr, nullptr, nullptr, x, y);
    return;
  }
  if (which == 0)
    hipLaunchKernelGGL((k_spmv<0, 2>), dim3(grid), dim3(256), 0, c->stream, c->N, c->d_rowptr, c->d_colind, c->d_bmask,
                       c->d_vals[PLFEM_BLK_AXX], c->d_vals[PLFEM_BLK_AXY], c->d_vals[PLFEM_BLK_AYX],
                       c->d_vals[PLFEM_BLK_AYY], x, y);
  else
    hipLaunchKernelGGL((k_spmv<1, 2>), dim3(grid), dim3(256), 0, c->stream, c->N, c->d_rowptr, c->d_colind, c->d_bmask,
                       c->d_vals[PLFEM_BLK_MINV], nullptr, nullptr, nullptr, x, y);
}

}  // namespace plfem
