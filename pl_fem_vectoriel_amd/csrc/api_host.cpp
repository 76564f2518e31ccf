// C-ABI of the host-only symbolic phase (include/plfem.h, "Symbolic phase").  No HIP here.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <new>
#include <string>
#include <utility>

#include "../../include/plfem.h"
#include "internal.h"
#include "host_eig.h"

using plfem::Symbolic;

static void set_err(char* err, int32_t errlen, const std::string& msg) {
  if (err && errlen > 0) {
    std::snprintf(err, (size_t)errlen, "%s", msg.c_str());
  }
}

extern "C" int plfem_symbolic_create(int32_t nv, int32_t ne, const double* p_host, const int32_t* t_host,
                                     int32_t leaf_elems, int32_t nthreads, plfem_symbolic** out,
                                     char* err, int32_t errlen) {
  return plfem_symbolic_create_ex(nv, ne, p_host, t_host, leaf_elems, nthreads, 2, 1, out, err, errlen);
}

extern "C" int plfem_symbolic_create_ex(int32_t nv, int32_t ne, const double* p_host, const int32_t* t_host,
                                        int32_t leaf_elems, int32_t nthreads, int32_t dofs_per_node, int32_t dirichlet,
                                        plfem_symbolic** out, char* err, int32_t errlen) {
  if (!out) return PLFEM_EINVAL;
  *out = nullptr;
  if (!p_host || !t_host || nv < 3 || ne < 1) {
    set_err(err, errlen, "plfem_symbolic_create: empty mesh or null pointer");
    return PLFEM_EINVAL;
  }
  if (dofs_per_node != 1 && dofs_per_node != 2) {
    set_err(err, errlen, "plfem_symbolic_create_ex: dofs_per_node must be 1 (scalar) or 2 (vectorial)");
    return PLFEM_EINVAL;
  }
  plfem_symbolic* h = new (std::nothrow) plfem_symbolic();
  if (!h) { set_err(err, errlen, "out of memory"); return PLFEM_EINVAL; }
  std::string msg;
  try {
    msg = plfem::build_symbolic(nv, ne, p_host, t_host, leaf_elems <= 0 ? 24 : leaf_elems,
                                nthreads <= 0 ? 1 : nthreads, h->S, dofs_per_node, dirichlet != 0);
  } catch (const std::exception& e) {
    msg = std::string("exception in symbolic analysis: ") + e.what();
  }
  if (!msg.empty()) {
    set_err(err, errlen, msg);
    delete h;
    return PLFEM_EMESH;
  }
  *out = h;
  return PLFEM_OK;
}

extern "C" void plfem_symbolic_destroy(plfem_symbolic* sym) { delete sym; }

extern "C" int plfem_symbolic_info(const plfem_symbolic* sym, int64_t* info) {
  if (!sym || !info) return PLFEM_EINVAL;
  const Symbolic& S = sym->S;
  info[PLFEM_INFO_NV] = S.nv;
  info[PLFEM_INFO_NE] = S.ne;
  info[PLFEM_INFO_NEDGES] = S.nedges;
  info[PLFEM_INFO_N] = S.N;
  info[PLFEM_INFO_NSOLVE] = S.nsolve;
  info[PLFEM_INFO_NNZ] = S.rowptr.empty() ? 0 : (int64_t)S.rowptr[S.N];
  info[PLFEM_INFO_LEVELS] = S.L;
  info[PLFEM_INFO_NFRONTS] = S.nfronts;
  info[PLFEM_INFO_FRONT_DOUBLES] = S.foff.empty() ? 0 : S.foff.back();
  info[PLFEM_INFO_MAX_FRONT] = S.max_m;
  info[PLFEM_INFO_SOLVE_ENTRIES] = S.solve_entries;
  info[PLFEM_INFO_FACTOR_FLOPS] = (int64_t)S.factor_flops;
  info[PLFEM_INFO_T_NUMBERING_US] = (int64_t)(S.t_numbering * 1e6);
  info[PLFEM_INFO_T_PATTERN_US] = (int64_t)(S.t_pattern * 1e6);
  info[PLFEM_INFO_T_TREE_US] = (int64_t)(S.t_tree * 1e6);
  info[PLFEM_INFO_T_FRONTS_US] = (int64_t)(S.t_fronts * 1e6);
  info[PLFEM_INFO_ARENA_DOUBLES] = S.arena_doubles;
  info[PLFEM_INFO_DOFS_PER_NODE] = S.dpn;
  return PLFEM_OK;
}

namespace {
struct ArrayRef { const void* ptr; int64_t bytes; };
template <class T>
ArrayRef aref(const std::vector<T>& v) { return {v.data(), (int64_t)(v.size() * sizeof(T))}; }
ArrayRef aref(const plfem::rawvec_i32& v) { return {v.data(), (int64_t)(v.size() * sizeof(int32_t))}; }

bool lookup(const Symbolic& S, const char* name, ArrayRef& r) {
  std::string n(name ? name : "");
  if (n == "edof") r = aref(S.edof);
  else if (n == "tsorted") r = aref(S.tsorted);
  else if (n == "edges") r = aref(S.edges);
  else if (n == "doflocs") r = aref(S.doflocs);
  else if (n == "bmask") r = aref(S.bmask);
  else if (n == "interior") r = aref(S.interior);
  else if (n == "int_index") r = aref(S.int_index);
  else if (n == "rowptr") r = aref(S.rowptr);
  else if (n == "colind") { plfem::ensure_pattern(S); r = aref(S.colind); }
  else if (n == "slot_row") { plfem::ensure_pattern(S); r = aref(S.slot_row); }
  else if (n == "nptr") r = aref(S.nptr);
  else if (n == "nadj") r = aref(S.nadj);
  else if (n == "nloc") r = aref(S.nloc);
  else if (n == "leaf_of_elem") r = aref(S.leaf_of_elem);
  else if (n == "leaf_elem_ptr") r = aref(S.leaf_elem_ptr);
  else if (n == "leaf_elems") r = aref(S.leaf_elems);
  else if (n == "epos") {                       // by element id, [6][ne] (the analysis keeps it in leaf order)
    if (S.epos_by_elem.size() != S.epos.size()) {
      S.epos_by_elem.resize(S.epos.size());
      for (int q = 0; q < S.ne; ++q)
        for (int a = 0; a < 6; ++a) S.epos_by_elem[(size_t)a * S.ne + S.leaf_elems[q]] = S.epos[(size_t)q * 6 + a];
    }
    r = aref(S.epos_by_elem);
  }
  else if (n == "epos_leaf") r = aref(S.epos);
  else if (n == "owner") r = aref(S.owner);
  else if (n == "fs") r = aref(S.fs);
  else if (n == "fb") r = aref(S.fb);
  else if (n == "fs_true") r = aref(S.fs_true);
  else if (n == "fb_true") r = aref(S.fb_true);
  else if (n == "fnode_ptr") r = aref(S.fnode_ptr);
  else if (n == "fnodes") r = aref(S.fnodes);
  else if (n == "cinv0") r = aref(S.cinv0);
  else if (n == "cinv1") r = aref(S.cinv1);
  else if (n == "foff") r = aref(S.foff);
  else if (n == "soff") r = aref(S.soff);
  else if (n == "prow") r = aref(S.prow);
  else if (n == "npos") r = aref(S.npos);
  else return false;
  return true;
}
}  // namespace

extern "C" int64_t plfem_symbolic_array_bytes(const plfem_symbolic* sym, const char* name) {
  if (!sym) return PLFEM_EINVAL;
  ArrayRef r;
  if (!lookup(sym->S, name, r)) return PLFEM_EINVAL;
  return r.bytes;
}

extern "C" int plfem_symbolic_get(const plfem_symbolic* sym, const char* name, void* out_host, int64_t nbytes) {
  if (!sym || !out_host) return PLFEM_EINVAL;
  ArrayRef r;
  if (!lookup(sym->S, name, r)) return PLFEM_EINVAL;
  if (r.bytes != nbytes) return PLFEM_EINVAL;
  std::memcpy(out_host, r.ptr, (size_t)nbytes);
  return PLFEM_OK;
}

// ---------------------------------------------------------------------------------------------
// Uniform red refinement (mesh producer).
// ---------------------------------------------------------------------------------------------
extern "C" int plfem_mesh_edge_count(int32_t nv, int32_t ne, const double* p_host, const int32_t* t_host,
                                     int32_t* nedges, char* err, int32_t errlen) {
  if (!p_host || !t_host || !nedges) return PLFEM_EINVAL;
  plfem::Symbolic S;
  std::string msg = plfem::numbering_only(nv, ne, p_host, t_host, S);
  if (!msg.empty()) { set_err(err, errlen, msg); return PLFEM_EMESH; }
  *nedges = S.nedges;
  return PLFEM_OK;
}

extern "C" int plfem_mesh_refine(int32_t nv, int32_t ne, const double* p_host, const int32_t* t_host,
                                 double* p_out, int32_t* t_out, char* err, int32_t errlen) {
  if (!p_host || !t_host || !p_out || !t_out) return PLFEM_EINVAL;
  plfem::Symbolic S;
  std::string msg = plfem::numbering_only(nv, ne, p_host, t_host, S);
  if (!msg.empty()) { set_err(err, errlen, msg); return PLFEM_EMESH; }
  const int N = S.N;
  std::memcpy(p_out, S.doflocs.data(), sizeof(double) * 2 * (size_t)N);   // vertices, then edge midpoints
  const int32_t* d = S.edof.data();
  const size_t n4 = (size_t)4 * ne;
  for (int e = 0; e < ne; ++e) {
    const int32_t v0 = d[e], v1 = d[(size_t)ne + e], v2 = d[(size_t)2 * ne + e];
    const int32_t e0 = d[(size_t)3 * ne + e], e1 = d[(size_t)4 * ne + e], e2 = d[(size_t)5 * ne + e];
    // children hstacked as (t0,e0,e2), (t1,e0,e1), (t2,e2,e1), (e0,e1,e2), every column sorted ascending
    const int32_t ch[4][3] = {{v0, e0, e2}, {v1, e0, e1}, {v2, e2, e1}, {e0, e1, e2}};
    for (int c = 0; c < 4; ++c) {
      int32_t a = ch[c][0], b = ch[c][1], cc = ch[c][2];
      if (a > b) std::swap(a, b);
      if (b > cc) std::swap(b, cc);
      if (a > b) std::swap(a, b);
      const size_t col = (size_t)c * ne + e;
      t_out[col] = a;
      t_out[n4 + col] = b;
      t_out[2 * n4 + col] = cc;
    }
  }
  return PLFEM_OK;
}
