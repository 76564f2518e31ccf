// Host-only test hooks (libplfem_testhooks.so, include/plfem.h under PLFEM_TEST_HOOKS): the small dense eigensolver of the
// Lanczos drivers, exposed for the CPU test-suite and the sanitizer harness.  No HIP here.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

#define PLFEM_TEST_HOOKS 1
#include "../../include/plfem.h"
#include "host_eig.h"

// host eigensolver of the Lanczos drivers, exposed for the CPU test-suite
extern "C" int plfem_debug_symeig(int32_t n, const double* a_host, int32_t last_rows, double* w_out, double* v_out) {
  if (n < 1 || n > 4096 || !a_host || !w_out || !v_out || last_rows > n) return PLFEM_EINVAL;
  std::vector<double> A(a_host, a_host + (size_t)n * n), V, w;
  const bool ok = last_rows < 0 ? plfem::sym_eig(n, A, V, w) : plfem::sym_eig_last_rows(n, last_rows, A, V, w);
  std::copy(w.begin(), w.end(), w_out);
  std::copy(V.begin(), V.end(), v_out);
  return ok ? PLFEM_OK : PLFEM_ENOCONV;
}

// band path of the host eigensolver (see host_eig.h), exposed for the CPU test-suite
extern "C" int plfem_debug_symeig_band(int32_t n, int32_t b, const double* a_host, int32_t nsel, double* w_out, double* v_out) {
  if (n < 1 || n > 4096 || b < 0 || !a_host || !w_out || !v_out || nsel < 0 || nsel > n) return PLFEM_EINVAL;
  std::vector<double> w;
  bool ok = plfem::sym_band_eigenvalues(n, b, a_host, n, w);
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int x, int y) { return std::fabs(w[x]) > std::fabs(w[y]); });
  order.resize(nsel);
  std::fill(v_out, v_out + (size_t)n * n, 0.0);
  ok = plfem::sym_band_eigenvectors(n, b, a_host, n, w, order, v_out, n) && ok;
  std::copy(w.begin(), w.end(), w_out);
  return ok ? PLFEM_OK : PLFEM_ENOCONV;
}
