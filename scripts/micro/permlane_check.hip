// Checks the lane maps kernels_front.hip assumes for v_permlane16_swap / v_permlane32_swap on gfx950:
// lane_xor16(v) = v of lane ^ 16, half_bcast<H>(v) = v of lane (l & 31) + 32 H.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ double lane_xor16(double v, bool odd_row) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(odd_row ? b[0] : b[1], odd_row ? a[0] : a[1]);
}
template <int H>
__device__ double half_bcast(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[H], a[H]);
}
__global__ void k(double* out) {
  const int l = threadIdx.x;
  const double v = 1000.0 + l;
  out[l] = lane_xor16(v, (l >> 4) & 1);
  out[64 + l] = half_bcast<0>(v);
  out[128 + l] = half_bcast<1>(v);
}
int main() {
  double* d;
  double h[192];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    bad += h[l] != 1000.0 + (l ^ 16);
    bad += h[64 + l] != 1000.0 + (l & 31);
    bad += h[128 + l] != 1000.0 + (l & 31) + 32;
  }
  printf("permlane check: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
  return bad ? 1 : 0;
}
