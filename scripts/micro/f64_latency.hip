// Dependent-chain latencies on gfx950 that the pivot-block design rests on: v_fma_f64, v_mul_f64, v_rcp_f64, v_max_f64,
// a v_cmp + v_cndmask pair, an LDS write -> read round trip, s_barrier with 4 waves.  One workgroup, shader clocks per op.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAIN 256
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, long long* t, double x0, double y0) {
  __shared__ double lds[512];
  double x = x0 + threadIdx.x * 1e-9, y = y0;
  lds[threadIdx.x] = x;
  __syncthreads();
  const long long c0 = clock64();
#pragma unroll 16
  for (int r = 0; r < CHAIN; ++r) {
    if (OP == 0) x = fma(x, y, y);
    if (OP == 1) x = x * y;
    if (OP == 2) x = __builtin_amdgcn_rcp(x);
    if (OP == 3) x = fmax(x, y) + 0.0 * r;
    if (OP == 4) x = (x > y) ? y : x * 1.0000001;
    if (OP == 5) { lds[threadIdx.x ^ 1] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); x = lds[threadIdx.x] + 1.0; }
    if (OP == 6) { __syncthreads(); }
    if (OP == 7) { lds[(threadIdx.x + 64) & 255] = x; __syncthreads(); x = lds[threadIdx.x] + 1.0; }
    if (OP == 8) x = fma(x, y, y) + (double)__builtin_amdgcn_readlane((int)r, 3);
    if (OP == 9) { x = fma(x, y, y); x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 5), __builtin_amdgcn_readlane(__double2loint(x), 5)); }
  }
  const long long c1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) t[OP] = c1 - c0;
}
int main() {
  double* d; long long* t;
  hipMalloc(&d, 8 * 256); hipMalloc(&t, 8 * 16);
  const char* names[] = {"v_fma_f64 dependent", "v_mul_f64 dependent", "v_rcp_f64 dependent", "v_max_f64 + v_add_f64 dependent", "v_cmp + v_cndmask(2) + v_mul dependent",
                         "LDS write -> wave barrier -> read + add (one workgroup of 4 waves)", "s_barrier alone (4 waves)", "LDS write -> s_barrier -> read + add", "fma + readlane(int) + cvt + add", "fma -> readlane x2 -> next fma"};
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, d, t, 1.0, 0.999);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, d, t, 1.0, 0.999);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, d, t, 1.3, 0.999);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), 0, 0, d, t, 1.0, 0.999);
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(256), 0, 0, d, t, 1.0, 1e9);
    hipLaunchKernelGGL(k<5>, dim3(1), dim3(256), 0, 0, d, t, 1.0, 0.999);
    hipLaunchKernelGGL(k<6>, dim3(1), dim3(256), 0, 0, d, t, 1.0, 0.999);
    hipLaunchKernelGGL(k<7>, dim3(1), dim3(256), 0, 0, d, t, 1.0, 0.999);
    hipLaunchKernelGGL(k<8>, dim3(1), dim3(256), 0, 0, d, t, 1.0, 0.999);
    hipLaunchKernelGGL(k<9>, dim3(1), dim3(256), 0, 0, d, t, 1.0, 0.999);
    long long h[16];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    if (pass) for (int o = 0; o < 10; ++o) printf("%-70s %7.1f clocks\n", names[o], (double)h[o] / CHAIN);
  }
  return 0;
}
