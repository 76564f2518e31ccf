// Test hooks of the library (include/plfem.h, section guarded by PLFEM_TEST_HOOKS): libplfem_testhooks.so.
// NOT part of libplfem_hip.so -- the product library exports none of these symbols and has no switch that alters a
// result.  This add-on links against the product library (it calls its internal launch_* functions and works on
// contexts the product library created), so the tests that need a hook still run the product's own kernels.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#ifndef PLFEM_TEST_HOOKS
#define PLFEM_TEST_HOOKS 1
#endif
#include "device.h"

#define HIP_TRY(ctx, call)                                                                   \
  do {                                                                                       \
    hipError_t e__ = (call);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                       \
      return PLFEM_EHIP;                                                                     \
    }                                                                                        \
  } while (0)

namespace {
int check_launch(plfem_ctx* c, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    c->err = std::string(what) + ": " + hipGetErrorString(e);
    return PLFEM_EHIP;
  }
  return PLFEM_OK;
}

// a slightly wrong factor: D^-1 of the root front scaled by 1 + test_perturb after every factorisation
void perturb_root_pivots(plfem_ctx* c) {
  if (c->test_perturb != 0.0)
    plfem::launch_scale(c, (int64_t)2 * c->dpn * c->S->fs[0], 1.0 + c->test_perturb, c->d_delta);
}
}  // namespace

extern "C" int plfem_debug_set_perturb(plfem_ctx* c, double value) {
  if (!c) return PLFEM_EINVAL;
  c->test_perturb = value;
  c->test_post_factor = value != 0.0 ? perturb_root_pivots : nullptr;
  c->factored = false;              // takes effect at the next plfem_factor
  return PLFEM_OK;
}

extern "C" int plfem_debug_factor_until(plfem_ctx* c, double sigma, int32_t level, int32_t step, int32_t stage) {
  if (!c) return PLFEM_EINVAL;
  if (!c->assembled) { c->err = "debug factor before assemble"; return PLFEM_ESTATE; }
  if (c->upload_pending) {               // (as plfem_factor: the front-level index arrays travel on the copy stream)
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_upload, 0));
    c->upload_pending = false;
  }
  plfem::launch_factor(c, sigma, level, step, stage);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return check_launch(c, "debug factor");
}

// timing aid: reps block solves (BLOCK_P right-hand sides out of the Lanczos work buffers) with an optional front
// filter (0 all fronts, 1 skip fronts with more than 128 owned DOFs, 2 only those: wrong results, kernel times only)
extern "C" int plfem_debug_solve_block(plfem_ctx* c, int32_t reps, int32_t filter) {
  if (!c || reps < 1) return PLFEM_EINVAL;
  if (!c->factored) { c->err = "debug solve before factor"; return PLFEM_ESTATE; }
  if (c->max_block_p < plfem::BLOCK_P) { c->err = "plfem_debug_solve_block: the LDS budget of this context allows one right-hand side per sweep only"; return PLFEM_EINVAL; }
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemsetAsync(c->d_bw, 0, sizeof(double) * c->n2 * plfem::BLOCK_P, c->stream));
  c->debug_sweep_filter = filter;
  for (int r = 0; r < reps; ++r) plfem::launch_solve_block(c, c->d_bw, c->d_w, c->n2, false);
  c->debug_sweep_filter = 0;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return check_launch(c, "debug solve block");
}

extern "C" int plfem_debug_copy(plfem_ctx* c, const char* name, int64_t offset, int64_t count, double* out_host) {
  if (!c || !name || !out_host || offset < 0 || count < 0) return PLFEM_EINVAL;
  std::string n(name);
  const double* src = nullptr;
  if (n == "front") src = c->d_front;
  else if (n == "schur") src = c->d_schur;            // (arena of level l starts at (l & 1) * arena_doubles)
  else if (n == "fvec") src = c->d_fvec;
  else if (n == "wbuf") src = c->d_wbuf;
  else if (n == "rbuf") src = c->d_rbuf;
  else if (n == "dinv") src = c->d_dinv;
  else if (n == "delta") src = c->d_delta;
  else if (n == "fvec2") src = c->d_fvec2;
  else if (n == "xl") src = c->d_xl;
  else if (n == "elem") src = c->d_elem;
  else if (n == "colind" || n == "slot_row") {            // int32 index arrays, delivered as doubles
    if (offset + count > c->nnz) return PLFEM_EINVAL;
    std::vector<int32_t> tmp((size_t)count);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(tmp.data(), (n == "colind" ? c->d_colind : c->d_slot_row) + offset, sizeof(int32_t) * count, hipMemcpyDeviceToHost));
    for (int64_t q = 0; q < count; ++q) out_host[q] = (double)tmp[q];
    return PLFEM_OK;
  }
  else return PLFEM_EINVAL;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(out_host, src + offset, sizeof(double) * count, hipMemcpyDeviceToHost));
  return PLFEM_OK;
}
