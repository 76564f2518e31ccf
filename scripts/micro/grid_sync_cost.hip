// Micro-benchmark: cost of cooperative_groups grid.sync() on this GPU for several grid sizes
// (hipcc --offload-arch=gfx950 -O3 grid_sync_cost.hip -o grid_sync_cost).
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(512) void k_syncs(int nsync, double* out) {
  cg::grid_group grid = cg::this_grid();
  double v = threadIdx.x;
  for (int s = 0; s < nsync; ++s) {
    v = v * 1.0000001 + 1.0;
    if (threadIdx.x == 0) out[blockIdx.x] = v;       // something to make visible
    grid.sync();
    v += out[(blockIdx.x + 1) % gridDim.x];
  }
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}

int main() {
  int dev = 0, coop = 0, cus = 0;
  hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int perCU = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_syncs, 512, 36 * 1024);
  printf("cooperative launch %d, CUs %d, co-resident blocks per CU (512 thr, 36 KB LDS) %d\n", coop, cus, perCU);
  double* out; hipMalloc(&out, sizeof(double) * 8192);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 512, 1024}) {
    if (blocks > perCU * cus) continue;
    for (int nsync : {0, 12, 112}) {
      void* args[] = {&nsync, &out};
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipError_t rc = hipLaunchCooperativeKernel((void*)k_syncs, dim3(blocks), dim3(512), args, 36 * 1024, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        if (rc != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(rc)); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
      }
      printf("blocks %4d syncs %3d: %.1f us\n", blocks, nsync, best * 1e3);
    }
  }
  return 0;
}
