#include "plfem.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
int main(int argc, char** argv) {
  for (int a = 1; a < argc; ++a) {
    FILE* f = fopen(argv[a], "rb"); if (!f) return 2;
    int32_t hdr[2]; if (fread(hdr, 4, 2, f) != 2) return 2;
    int nv = hdr[0], ne = hdr[1];
    std::vector<double> p(2 * (size_t)nv); std::vector<int32_t> t(3 * (size_t)ne);
    if (fread(p.data(), 8, p.size(), f) != p.size() || fread(t.data(), 4, t.size(), f) != t.size()) return 2;
    fclose(f);
    for (int nth : {1, 3, 8}) {
      plfem_symbolic* s = nullptr; char err[256];
      int rc = plfem_symbolic_create(nv, ne, p.data(), t.data(), 0, nth, &s, err, 256);
      if (rc) { printf("create failed %d %s\n", rc, err); return 1; }
      int64_t info[32]; plfem_symbolic_info(s, info);
      for (const char* name : {"colind", "slot_row", "fnodes", "cinv0", "epos", "rowptr", "owner"}) {
        int64_t nb = plfem_symbolic_array_bytes(s, name);
        std::vector<char> buf(nb > 0 ? nb : 1);
        if (nb < 0 || plfem_symbolic_get(s, name, buf.data(), nb)) { printf("get %s failed\n", name); return 1; }
      }
      printf("%s threads %d: N %lld nnz %lld fronts %lld\n", argv[a], nth, (long long)info[3], (long long)info[5], (long long)info[7]);
      plfem_symbolic_destroy(s);
    }
    int32_t nedges = 0; char err[256];
    if (plfem_mesh_edge_count(nv, ne, p.data(), t.data(), &nedges, err, 256)) return 1;
    std::vector<double> p2(2 * (size_t)(nv + nedges)); std::vector<int32_t> t2(12 * (size_t)ne);
    if (plfem_mesh_refine(nv, ne, p.data(), t.data(), p2.data(), t2.data(), err, 256)) return 1;
    printf("refine ok: %d edges\n", nedges);
  }
  // host eigensolver
  for (int n : {1, 5, 44, 104}) {
    std::vector<double> A((size_t)n * n), w(n), V((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = (double)((i * 131 + j * 71) % 97) / 97.0 - 0.5; A[(size_t)i * n + j] = v; A[(size_t)j * n + i] = v; }
    if (plfem_debug_symeig(n, A.data(), -1, w.data(), V.data())) return 1;
    if (plfem_debug_symeig(n, A.data(), n < 4 ? n : 4, w.data(), V.data())) return 1;
    // band path (half bandwidth 4): eigenvalues + the vectors of the (up to) 22 largest |w|
    const int rc = plfem_debug_symeig_band(n, 4, A.data(), n < 22 ? n : 22, w.data(), V.data());
    if (rc != 0 && rc != PLFEM_ENOCONV) return 1;
  }
  printf("symeig ok\n");
  return 0;
}
