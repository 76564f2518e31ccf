// Times the tall-skinny panel products of the block Lanczos step in isolation (C1 sizes: n = 181 278, P = 4): the library's
// kernels and candidate variants, plus a plain streaming read as the yardstick.  Prints us and GB/s of panel bytes.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/panel_bench.hip -o scripts/micro/build/panel_bench
#include "../../pl_fem_vectoriel_amd/csrc/kernels_lanczos.hip"

#include <cstdio>
#include <vector>

namespace plfem {
namespace {

// ---- 16-byte variants measured in round 4 and NOT adopted by the library (see kernels_lanczos.hip) ----
// ---- the same two panel products with 16-byte accesses (n even: every column starts 16-byte aligned) -------------
// An 8-byte access per lane streams at 0.54-0.70 of the rate of a 16-byte one on this chip (MI355X_MICROARCH.md, cache
// policy table) -- the round-3 kernels above sat at 3.1 TB/s for exactly that reason -- so a lane takes TWO consecutive
// rows per load.
// Sums V (a power of two <= 64) per-lane values over the 64 lanes with V - 1 + log2(64 / V) shuffles instead of 6 V (the
// butterfly of the solve sweeps); on return a[0] of lane l is the complete sum of value multi_reduce_index<V>(l).
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

// partial[(c P + q) nseg + seg] = sum over the rows of segment seg of Pm[i, c] W[i, q].  One WAVE per (row segment of
// DOT_SEG rows, group of 4 columns), the waves of a workgroup on consecutive column groups of one segment (they share
// the segment's rows of W in L1 / L2).  Two trips of (4 + P) 16-byte loads per lane in flight = 16 KB per wave.
constexpr int DOT_SEG = 1024;          // rows per partial sum (8 trips of 128 rows)
template <int P>
__global__ __launch_bounds__(256) void k_panel_dot_p16(int64_t n, int ncols, int ncg, int nseg, const double* __restrict__ Pm,
                                                       const double* __restrict__ W, int64_t ldw, double* __restrict__ partial) {
  constexpr int CW = 4;
  static_assert(CW * P == 16, "epilogue reduces 16 values per wave");
  const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (job >= nseg * ncg) return;
  const int seg = job / ncg, c0 = (job - seg * ncg) * CW;
  const int lane = threadIdx.x & 63;
  const int64_t i0 = (int64_t)seg * DOT_SEG, i1 = min(n, i0 + DOT_SEG);
  const double2* col[CW];
#pragma unroll
  for (int t = 0; t < CW; ++t) col[t] = reinterpret_cast<const double2*>(Pm + (int64_t)min(c0 + t, ncols - 1) * n);   // clamped: result discarded
  const double2* wq[P];
#pragma unroll
  for (int q = 0; q < P; ++q) wq[q] = reinterpret_cast<const double2*>(W + (int64_t)q * ldw);
  double acc[CW * P];
#pragma unroll
  for (int v = 0; v < CW * P; ++v) acc[v] = 0.0;
  // pair index p covers rows 2p, 2p + 1 (n even: no pair straddles the end)
  const int64_t p1 = i1 >> 1;
  int64_t p = (i0 >> 1) + lane;
  for (; p + 64 < p1; p += 128) {
    double2 a[2][CW], w[2][P];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 0; t < CW; ++t) a[h][t] = col[t][p + 64 * h];
#pragma unroll
      for (int q = 0; q < P; ++q) w[h][q] = wq[q][p + 64 * h];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < CW; ++t)
#pragma unroll
        for (int q = 0; q < P; ++q) acc[t * P + q] = fma(a[h][t].y, w[h][q].y, fma(a[h][t].x, w[h][q].x, acc[t * P + q]));
  }
  for (; p < p1; p += 64) {
    double2 w[P];
#pragma unroll
    for (int q = 0; q < P; ++q) w[q] = wq[q][p];
#pragma unroll
    for (int t = 0; t < CW; ++t) {
      const double2 a = col[t][p];
#pragma unroll
      for (int q = 0; q < P; ++q) acc[t * P + q] = fma(a.y, w[q].y, fma(a.x, w[q].x, acc[t * P + q]));
    }
  }
  multi_reduce<CW * P>(acc, lane);
  if (lane < CW * P) {
    const int v = multi_reduce_index<CW * P>(lane), t = v / P, q = v % P;
    if (c0 + t < ncols) partial[((int64_t)(c0 + t) * P + q) * nseg + seg] = acc[0];
  }
}

// W[i, q] -= sum_c Pm[i, c] H[c + q ldh], two rows per lane, UNR columns (16-byte loads) in flight per lane
template <int P, int UNR>
__global__ __launch_bounds__(256) void k_panel_axpy_p16(int64_t n, int ncols, const double* __restrict__ Pm,
                                                        const double* __restrict__ H, int ldh, double* __restrict__ W,
                                                        int64_t ldw, double* __restrict__ wil, int N, int dpn) {
  extern __shared__ double sh[];      // [c][P]
  for (int k = threadIdx.x; k < ncols * P; k += 256) sh[k] = H[(k / P) + (int64_t)(k % P) * ldh];
  __syncthreads();
  const int64_t pr = (int64_t)blockIdx.x * 256 + threadIdx.x;       // row pair
  if (2 * pr >= n) return;
  const double2* base = reinterpret_cast<const double2*>(Pm) + pr;
  const int64_t cs = n >> 1;                                          // column stride in pairs
  double2 acc[P];
#pragma unroll
  for (int q = 0; q < P; ++q) acc[q] = make_double2(0.0, 0.0);
  int c = 0;
  for (; c + UNR <= ncols; c += UNR) {
    double2 a[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) a[u] = base[(int64_t)(c + u) * cs];
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int q = 0; q < P; ++q) {
        const double hq = sh[(c + u) * P + q];
        acc[q].x = fma(a[u].x, hq, acc[q].x);
        acc[q].y = fma(a[u].y, hq, acc[q].y);
      }
  }
  for (; c < ncols; ++c) {
    const double2 a = base[(int64_t)c * cs];
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const double hq = sh[c * P + q];
      acc[q].x = fma(a.x, hq, acc[q].x);
      acc[q].y = fma(a.y, hq, acc[q].y);
    }
  }
  double2 w[P];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    double2* wp = reinterpret_cast<double2*>(W + (int64_t)q * ldw) + pr;
    w[q] = *wp;
    w[q].x -= acc[q].x;
    w[q].y -= acc[q].y;
    *wp = w[q];
  }
  if (wil) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t i = 2 * pr + h;
      const int comp = (int)(i / N), node = (int)(i - (int64_t)comp * N);
      double* d = wil + ((int64_t)node * dpn + comp) * P;
#pragma unroll
      for (int q = 0; q < P; ++q) d[q] = h == 0 ? w[q].x : w[q].y;
    }
  }
}


// yardstick: sum of a panel, 8-byte and 16-byte loads, grid-stride, 8 loads in flight per lane
template <int W>
__global__ __launch_bounds__(256) void k_stream_sum(int64_t n8, const double* __restrict__ x, double* __restrict__ out) {
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (W == 1) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += 8 * stride) {
      double a[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) a[t] = (i + t * stride < n8) ? x[i + t * stride] : 0.0;
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += a[t];
    }
  } else {
    const int64_t n16 = n8 / 2;
    const double2* x2 = reinterpret_cast<const double2*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += 8 * stride) {
      double2 a[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) a[t] = (i + t * stride < n16) ? x2[i + t * stride] : make_double2(0.0, 0.0);
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += a[t].x + a[t].y;
    }
  }
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}

// dot with the W rows of the block's segment staged in LDS: one workgroup = (segment of SEG rows, NWV waves x CW columns);
// WIDE: 16-byte panel loads (two rows per lane)
template <int P, int CW, int NWV, int SEG, bool WIDE>
__global__ __launch_bounds__(64 * NWV) void k_dot_lds(int64_t n, int ncols, int nseg, const double* __restrict__ Pm,
                                                      const double* __restrict__ W, int64_t ldw, double* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) double sw[P][SEG];
  const int seg = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = (blockIdx.y * NWV + wave) * CW;
  const int64_t i0 = (int64_t)seg * SEG, i1 = min(n, i0 + SEG);
  const int rows = (int)(i1 - i0);
  for (int k = threadIdx.x; k < P * SEG; k += 64 * NWV) {
    const int q = k / SEG, r = k - q * SEG;
    sw[q][r] = r < rows ? W[(int64_t)q * ldw + i0 + r] : 0.0;
  }
  __syncthreads();
  if (c0 >= ncols) return;
  double acc[CW * P];
#pragma unroll
  for (int v = 0; v < CW * P; ++v) acc[v] = 0.0;
  if (WIDE) {
    const double2* col[CW];
#pragma unroll
    for (int t = 0; t < CW; ++t) col[t] = reinterpret_cast<const double2*>(Pm + (int64_t)min(c0 + t, ncols - 1) * n + i0);
    const int np = rows >> 1;
    for (int p = lane; p < np; p += 128) {
      double2 a[2][CW];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < CW; ++t) a[h][t] = (p + 64 * h < np) ? col[t][p + 64 * h] : make_double2(0.0, 0.0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pp = min(p + 64 * h, SEG / 2 - 1);
#pragma unroll
        for (int q = 0; q < P; ++q) {
          const double2 w = *reinterpret_cast<const double2*>(&sw[q][2 * pp]);
#pragma unroll
          for (int t = 0; t < CW; ++t) acc[t * P + q] = fma(a[h][t].y, w.y, fma(a[h][t].x, w.x, acc[t * P + q]));
        }
      }
    }
  } else {
    const double* col[CW];
#pragma unroll
    for (int t = 0; t < CW; ++t) col[t] = Pm + (int64_t)min(c0 + t, ncols - 1) * n + i0;
    for (int r = lane; r < rows; r += 128) {
      double a[2][CW];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < CW; ++t) a[h][t] = (r + 64 * h < rows) ? col[t][r + 64 * h] : 0.0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rr = min(r + 64 * h, SEG - 1);
#pragma unroll
        for (int q = 0; q < P; ++q) {
          const double w = sw[q][rr];
#pragma unroll
          for (int t = 0; t < CW; ++t) acc[t * P + q] = fma(a[h][t], w, acc[t * P + q]);
        }
      }
    }
  }
  multi_reduce<CW * P>(acc, lane);
  if (lane < CW * P) {
    const int v = multi_reduce_index<CW * P>(lane), t = v / P, q = v % P;
    if (c0 + t < ncols) partial[((int64_t)(c0 + t) * P + q) * nseg + seg] = acc[0];
  }
}

// axpy, 8-byte loads, one row per lane, UNR columns in flight and the NEXT batch requested before the current one is used
template <int P, int UNR>
__global__ __launch_bounds__(256) void k_axpy_db(int64_t n, int ncols, const double* __restrict__ Pm, const double* __restrict__ H,
                                                 int ldh, double* __restrict__ W, int64_t ldw) {
  extern __shared__ double sh[];
  for (int k = threadIdx.x; k < ncols * P; k += 256) sh[k] = H[(k / P) + (int64_t)(k % P) * ldh];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double acc[P];
#pragma unroll
  for (int q = 0; q < P; ++q) acc[q] = 0.0;
  double a[UNR], b[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) a[u] = (u < ncols) ? Pm[(int64_t)u * n + i] : 0.0;
  for (int c = 0; c < ncols; c += UNR) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) b[u] = (c + UNR + u < ncols) ? Pm[(int64_t)(c + UNR + u) * n + i] : 0.0;
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      if (c + u < ncols) {
#pragma unroll
        for (int q = 0; q < P; ++q) acc[q] = fma(a[u], sh[(c + u) * P + q], acc[q]);
      }
#pragma unroll
    for (int u = 0; u < UNR; ++u) a[u] = b[u];
  }
#pragma unroll
  for (int q = 0; q < P; ++q) W[(int64_t)q * ldw + i] -= acc[q];
}

}  // namespace
}  // namespace plfem

using namespace plfem;

template <class F>
static double time_us(F&& launch, int reps = 20) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) launch();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return 1e3 * ms / reps;
}

int main() {
  constexpr int P = 4;
  const int64_t n = 181278;
  const int maxc = 96;
  double *V, *BV, *W, *H, *partial, *out;
  hipMalloc(&V, sizeof(double) * n * (maxc + 8));
  hipMalloc(&BV, sizeof(double) * n * (maxc + 8));
  hipMalloc(&W, sizeof(double) * n * P);
  hipMalloc(&H, sizeof(double) * (maxc + 8) * P);
  hipMalloc(&partial, sizeof(double) * 4096 * (maxc + 8) * P);
  hipMalloc(&out, sizeof(double) * 65536);
  hipMemset(V, 0, sizeof(double) * n * (maxc + 8));
  hipMemset(BV, 0, sizeof(double) * n * (maxc + 8));
  hipMemset(W, 0, sizeof(double) * n * P);
  hipMemset(H, 0, sizeof(double) * (maxc + 8) * P);
  const int nchunks = (int)((n + 1023) / 1024);
  for (int ncols : {8, 24, 48, 72, 96}) {
    const double mb = 8.0 * n * ncols / 1e6;
    printf("== ncols %d: panel %.1f MB\n", ncols, mb);
    auto rep = [&](const char* name, double us) { printf("  %-44s %7.1f us  %6.0f GB/s\n", name, us, mb / us * 1e3); };
    rep("stream sum, 8-byte loads, 2048 WGs", time_us([&] { hipLaunchKernelGGL(k_stream_sum<1>, dim3(2048), dim3(256), 0, 0, n * ncols, BV, out); }));
    rep("stream sum, 16-byte loads, 2048 WGs", time_us([&] { hipLaunchKernelGGL(k_stream_sum<2>, dim3(2048), dim3(256), 0, 0, n * ncols, BV, out); }));
    rep("stream sum, 16-byte loads, 1024 WGs", time_us([&] { hipLaunchKernelGGL(k_stream_sum<2>, dim3(1024), dim3(256), 0, 0, n * ncols, BV, out); }));
    rep("dot: k_panel_dot_p<4> (round 3, 8 B)", time_us([&] { hipLaunchKernelGGL(k_panel_dot_p<P>, dim3(nchunks, (ncols + 15) / 16), dim3(256), 0, 0, n, ncols, nchunks, BV, W, n, partial); }));
    {
      const int nseg = nchunks, ncg = (ncols + 3) / 4;
      rep("dot: k_panel_dot_p16<4> (16 B, CW 4)", time_us([&] { hipLaunchKernelGGL(k_panel_dot_p16<P>, dim3((nseg * ncg + 3) / 4), dim3(256), 0, 0, n, ncols, ncg, nseg, BV, W, n, partial); }));
    }
    {
      constexpr int SEG = 1024;
      const int nseg = (int)((n + SEG - 1) / SEG);
      rep("dot: W in LDS, 8 B, CW 4, 4 waves, seg 1024", time_us([&] { hipLaunchKernelGGL((k_dot_lds<P, 4, 4, SEG, false>), dim3(nseg, (ncols + 15) / 16), dim3(256), 0, 0, n, ncols, nseg, BV, W, n, partial); }));
      rep("dot: W in LDS, 16 B, CW 4, 4 waves, seg 1024", time_us([&] { hipLaunchKernelGGL((k_dot_lds<P, 4, 4, SEG, true>), dim3(nseg, (ncols + 15) / 16), dim3(256), 0, 0, n, ncols, nseg, BV, W, n, partial); }));
      rep("dot: W in LDS, 8 B, CW 4, 8 waves, seg 1024", time_us([&] { hipLaunchKernelGGL((k_dot_lds<P, 4, 8, SEG, false>), dim3(nseg, (ncols + 31) / 32), dim3(512), 0, 0, n, ncols, nseg, BV, W, n, partial); }));
      rep("dot: W in LDS, 16 B, CW 4, 8 waves, seg 1024", time_us([&] { hipLaunchKernelGGL((k_dot_lds<P, 4, 8, SEG, true>), dim3(nseg, (ncols + 31) / 32), dim3(512), 0, 0, n, ncols, nseg, BV, W, n, partial); }));
      rep("dot: W in LDS, 8 B, CW 8, 4 waves, seg 1024", time_us([&] { hipLaunchKernelGGL((k_dot_lds<P, 8, 4, SEG, false>), dim3(nseg, (ncols + 31) / 32), dim3(256), 0, 0, n, ncols, nseg, BV, W, n, partial); }));
    }
    {
      constexpr int SEG = 512;
      const int nseg = (int)((n + SEG - 1) / SEG);
      rep("dot: W in LDS, 8 B, CW 4, 8 waves, seg 512", time_us([&] { hipLaunchKernelGGL((k_dot_lds<P, 4, 8, SEG, false>), dim3(nseg, (ncols + 31) / 32), dim3(512), 0, 0, n, ncols, nseg, BV, W, n, partial); }));
      rep("dot: W in LDS, 16 B, CW 4, 8 waves, seg 512", time_us([&] { hipLaunchKernelGGL((k_dot_lds<P, 4, 8, SEG, true>), dim3(nseg, (ncols + 31) / 32), dim3(512), 0, 0, n, ncols, nseg, BV, W, n, partial); }));
    }
    rep("axpy: k_panel_axpy_p<4> (round 3, 8 B)", time_us([&] { hipLaunchKernelGGL(k_panel_axpy_p<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), sizeof(double) * ncols * P, 0, n, ncols, V, H, maxc + 8, W, n, (double*)nullptr, (int)(n / 2), 2); }));
    rep("axpy: k_panel_axpy_p16<4, 16>", time_us([&] { hipLaunchKernelGGL((k_panel_axpy_p16<P, 16>), dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), sizeof(double) * ncols * P, 0, n, ncols, V, H, maxc + 8, W, n, (double*)nullptr, (int)(n / 2), 2); }));
    rep("axpy: k_panel_axpy_p16<4, 8>", time_us([&] { hipLaunchKernelGGL((k_panel_axpy_p16<P, 8>), dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), sizeof(double) * ncols * P, 0, n, ncols, V, H, maxc + 8, W, n, (double*)nullptr, (int)(n / 2), 2); }));
    rep("axpy: 8 B, double-buffered, 8 in flight", time_us([&] { hipLaunchKernelGGL((k_axpy_db<P, 8>), dim3((unsigned)((n + 255) / 256)), dim3(256), sizeof(double) * ncols * P, 0, n, ncols, V, H, maxc + 8, W, n); }));
    rep("axpy: 8 B, double-buffered, 16 in flight", time_us([&] { hipLaunchKernelGGL((k_axpy_db<P, 16>), dim3((unsigned)((n + 255) / 256)), dim3(256), sizeof(double) * ncols * P, 0, n, ncols, V, H, maxc + 8, W, n); }));
  }
  hipError_t e = hipDeviceSynchronize();
  printf("status: %s\n", hipGetErrorString(e));
  return e == hipSuccess ? 0 : 1;
}
