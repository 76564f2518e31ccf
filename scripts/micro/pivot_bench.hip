// Times ldl_pivot_block (the 32 x 32 pivot block of the block LDL^T, one workgroup) in isolation: R back-to-back
// factorisations of the same L2-resident block, wall-clock (100 MHz) and shader cycles per block.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/pivot_bench.hip -o scripts/micro/build/pivot_bench
#include "../../pl_fem_vectoriel_amd/csrc/kernels_front.hip"

#include <cstdio>
#include <vector>

namespace plfem {
namespace {
__global__ __launch_bounds__(256) void k_bench(const double* F, int m, int reps, double* out, long long* ticks) {
  __shared__ __attribute__((aligned(16))) PivotLds piv;
  __shared__ double tile[NB][NB + 1];
  __shared__ double sDd[NB], sDo[NB];
  const int lane = threadIdx.x;
  const long long w0 = wall_clock64();
  const long long c0 = clock64();
  for (int r = 0; r < reps; ++r) {
    ldl_pivot_block(F, m, NB, lane, tile, sDd, sDo, piv, nullptr);
    __syncthreads();
  }
  const long long c1 = clock64();
  const long long w1 = wall_clock64();
  if (lane == 0) { ticks[0] = w1 - w0; ticks[1] = c1 - c0; }
  if (lane < NB) { out[lane] = sDd[lane]; out[NB + lane] = tile[lane][0]; }
}
}  // namespace
}  // namespace plfem

int main() {
  const int m = 64, reps = 200;
  std::vector<double> h((size_t)m * m, 0.0);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) h[(size_t)j * m + i] = (i == j ? 8.0 + (i % 3) : 1.0 / (1.0 + abs(i - j))) * ((i + j) % 5 == 0 ? -1.0 : 1.0);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < i; ++j) h[(size_t)i * m + j] = h[(size_t)j * m + i];
  double *dF, *dout;
  long long* dt;
  hipMalloc(&dF, sizeof(double) * m * m);
  hipMalloc(&dout, sizeof(double) * 64);
  hipMalloc(&dt, 16);
  hipMemcpy(dF, h.data(), sizeof(double) * m * m, hipMemcpyHostToDevice);
  for (int pass = 0; pass < 3; ++pass) {
    hipLaunchKernelGGL(plfem::k_bench, dim3(1), dim3(256), 0, 0, dF, m, reps, dout, dt);
    long long t[2];
    hipMemcpy(t, dt, 16, hipMemcpyDeviceToHost);
    double o[64];
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("pivot block: %.1f ns (%.0f shader clocks) per 32 x 32 block, %.0f clocks per pair step; d[0] = %g x[31][0] = %g\n",
           10.0 * t[0] / reps, (double)t[1] / reps, (double)t[1] / reps / 16, o[0], o[32 + 31]);
  }
  return 0;
}
