// Host-side symbolic analysis (see symbolic.h).  Pure C++17.
#include "symbolic.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <numeric>
#include <atomic>
#include <thread>

namespace plfem {
namespace {

using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) {
  return std::chrono::duration<double>(b - a).count();
}

// Small spin-waiting worker pool, alive for the duration of one build_symbolic() call: the parallel
// regions of the analysis are ~0.1-3 ms each, far too short to pay a thread creation per region.
struct Pool {
  int nt;
  std::vector<std::thread> th;
  std::atomic<int> gen{0}, done{0};
  std::atomic<bool> stop{false};
  std::function<void(int)> job;
  explicit Pool(int n) : nt(n) {
    for (int t = 1; t < nt; ++t)
      th.emplace_back([this, t] {
        int seen = 0;
        while (true) {
          while (gen.load(std::memory_order_acquire) == seen) {
            if (stop.load(std::memory_order_acquire)) return;
            std::this_thread::yield();
          }
          seen = gen.load(std::memory_order_acquire);
          job(t);
          done.fetch_add(1, std::memory_order_release);
        }
      });
  }
  void run(const std::function<void(int)>& f) {
    job = f;
    done.store(0, std::memory_order_release);
    gen.fetch_add(1, std::memory_order_release);
    f(0);
    while (done.load(std::memory_order_acquire) < nt - 1) std::this_thread::yield();
  }
  ~Pool() {
    stop.store(true, std::memory_order_release);
    for (auto& x : th) x.join();
  }
};
thread_local Pool* g_pool = nullptr;

// run f(begin, end, tid) over [0, n) in static chunks on the pool of the current build (if any)
template <class F>
void parallel_for(int64_t n, int nthreads, F f, int64_t min_parallel = 4096) {
  Pool* pool = g_pool;
  if (nthreads <= 1 || n < min_parallel || !pool) {
    f((int64_t)0, n, 0);
    return;
  }
  const int nt = pool->nt;
  const int64_t chunk = (n + nt - 1) / nt;
  pool->run([&](int tdx) {
    int64_t b = tdx * chunk, e = std::min(n, b + chunk);
    if (b < e) f(b, e, tdx);
  });
}

inline int pad8(int x) { return (x + 7) & ~7; }

// ------------------------------------------------------------------------------------------------
// P2 numbering, scikit-fem compatible (MeshTri sort_t + build_entities + ElementTriP2 dof layout)
// ------------------------------------------------------------------------------------------------
std::string p2_numbering(int nv, int ne, const double* p, const int32_t* t, Symbolic& S) {
  S.nv = nv;
  S.ne = ne;
  S.tsorted.resize((size_t)3 * ne);
  int32_t* t0 = S.tsorted.data();
  int32_t* t1 = t0 + ne;
  int32_t* t2 = t1 + ne;
  for (int e = 0; e < ne; ++e) {
    int32_t a = t[e], b = t[ne + e], c = t[2 * (size_t)ne + e];
    if (a < 0 || b < 0 || c < 0 || a >= nv || b >= nv || c >= nv) return "mesh.t refers to a vertex outside mesh.p";
    if (a > b) std::swap(a, b);
    if (b > c) std::swap(b, c);
    if (a > b) std::swap(a, b);
    if (a == b || b == c) return "degenerate element (repeated vertex)";
    t0[e] = a; t1[e] = b; t2[e] = c;
  }
  // edges bucketed by their smaller vertex: local edges (0,1),(1,2),(0,2) -> (t0,t1),(t1,t2),(t0,t2)
  std::vector<int32_t> cnt((size_t)nv + 1, 0);
  for (int e = 0; e < ne; ++e) { cnt[t0[e] + 1] += 2; cnt[t1[e] + 1] += 1; }
  for (int v = 0; v < nv; ++v) cnt[v + 1] += cnt[v];
  std::vector<int32_t> nb((size_t)3 * ne);
  {
    std::vector<int32_t> fill(cnt.begin(), cnt.end() - 1);
    for (int e = 0; e < ne; ++e) {
      nb[fill[t0[e]]++] = t1[e];
      nb[fill[t1[e]]++] = t2[e];
      nb[fill[t0[e]]++] = t2[e];
    }
  }
  // sort + unique each bucket; edge id = running count => lexicographic (min, max) rank
  // (two passes so that the per-vertex work runs on the worker pool: sort + count, prefix, write)
  std::vector<int32_t> eoff((size_t)nv + 1, 0);
  std::atomic<int> bad{0};
  const int nth = g_pool ? g_pool->nt : 1;
  parallel_for(nv, nth, [&](int64_t vb, int64_t ve, int) {
    for (int64_t v = vb; v < ve; ++v) {
      int32_t* b = nb.data() + cnt[v];
      int32_t* e = nb.data() + cnt[v + 1];
      std::sort(b, e);
      int32_t nu = 0;
      for (int32_t* q = b; q < e;) {
        int32_t* r = q;
        while (r < e && *r == *q) ++r;
        if (r - q > 2) bad.store(1);
        ++nu;
        q = r;
      }
      eoff[v + 1] = nu;
    }
  });
  if (bad.load()) return "non-manifold mesh: an edge is shared by more than two triangles";
  for (int v = 0; v < nv; ++v) eoff[v + 1] += eoff[v];
  const int nedges_total = eoff[nv];
  std::vector<uint8_t> mult((size_t)nedges_total);
  std::vector<int32_t> ea((size_t)nedges_total), eb((size_t)nedges_total);
  parallel_for(nv, nth, [&](int64_t vb, int64_t ve, int) {
    for (int64_t v = vb; v < ve; ++v) {
      const int32_t* b = nb.data() + cnt[v];
      const int32_t* e = nb.data() + cnt[v + 1];
      int32_t k = eoff[v];
      for (const int32_t* q = b; q < e;) {
        const int32_t* r = q;
        while (r < e && *r == *q) ++r;
        ea[k] = (int32_t)v;
        eb[k] = *q;
        mult[k] = (uint8_t)(r - q);
        ++k;
        q = r;
      }
    }
  });
  S.nedges = nedges_total;
  S.edges.resize((size_t)2 * S.nedges);
  std::copy(ea.begin(), ea.end(), S.edges.begin());
  std::copy(eb.begin(), eb.end(), S.edges.begin() + S.nedges);
  S.N = nv + S.nedges;
  const int N = S.N;
  auto edge_id = [&](int32_t a, int32_t b) -> int32_t {
    const int32_t* lo = eb.data() + eoff[a];
    const int32_t* hi = eb.data() + eoff[a + 1];
    return (int32_t)(std::lower_bound(lo, hi, b) - eb.data());
  };
  S.edof.resize((size_t)6 * ne);
  int32_t* d = S.edof.data();
  parallel_for(ne, nth, [&](int64_t e0, int64_t e1, int) {
    for (int64_t e = e0; e < e1; ++e) {
      d[e] = t0[e];
      d[(size_t)ne + e] = t1[e];
      d[(size_t)2 * ne + e] = t2[e];
      d[(size_t)3 * ne + e] = nv + edge_id(t0[e], t1[e]);
      d[(size_t)4 * ne + e] = nv + edge_id(t1[e], t2[e]);
      d[(size_t)5 * ne + e] = nv + edge_id(t0[e], t2[e]);
    }
  });
  S.doflocs.resize((size_t)2 * N);
  std::memcpy(S.doflocs.data(), p, sizeof(double) * nv);
  std::memcpy(S.doflocs.data() + N, p + nv, sizeof(double) * nv);
  parallel_for(S.nedges, nth, [&](int64_t k0, int64_t k1, int) {
    for (int64_t k = k0; k < k1; ++k) {
      S.doflocs[nv + k] = 0.5 * (p[ea[k]] + p[eb[k]]);
      S.doflocs[(size_t)N + nv + k] = 0.5 * (p[nv + ea[k]] + p[nv + eb[k]]);
    }
  });
  S.bmask.assign(N, 0);
  for (int k = 0; k < S.nedges; ++k)
    if (mult[k] == 1) { S.bmask[ea[k]] = 1; S.bmask[eb[k]] = 1; S.bmask[nv + k] = 1; }
  S.int_index.assign(N, -1);
  S.interior.clear();
  S.interior.reserve(N);
  for (int i = 0; i < N; ++i)
    if (!S.bmask[i]) { S.int_index[i] = (int32_t)S.interior.size(); S.interior.push_back(i); }
  S.nsolve = (int)S.interior.size();
  return "";
}

// node -> adjacent elements (CSR), elements ascending within each node
void node_to_elem(Symbolic& S) {
  const int N = S.N, ne = S.ne;
  std::vector<int32_t>& ptr = S.nptr;
  std::vector<int32_t>& adj = S.nadj;
  std::vector<uint8_t>& loc = S.nloc;
  ptr.assign((size_t)N + 1, 0);
  for (int a = 0; a < 6; ++a)
    for (int e = 0; e < ne; ++e) ptr[S.edof[(size_t)a * ne + e] + 1]++;
  for (int i = 0; i < N; ++i) ptr[i + 1] += ptr[i];
  adj.resize((size_t)6 * ne);
  loc.resize((size_t)6 * ne);
  std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
  for (int e = 0; e < ne; ++e)
    for (int a = 0; a < 6; ++a) {
      int32_t i = S.edof[(size_t)a * ne + e];
      adj[fill[i]] = e;
      loc[fill[i]] = (uint8_t)a;
      fill[i]++;
    }
}

// ------------------------------------------------------------------------------------------------
// scalar CSR pattern: row i = sorted union of the DOFs of the elements adjacent to node i
// ------------------------------------------------------------------------------------------------
void csr_pattern(Symbolic& S, int nthreads) {
  const int N = S.N, ne = S.ne;
  const std::vector<int32_t>& nptr = S.nptr;
  const std::vector<int32_t>& nadj = S.nadj;
  std::vector<int64_t> soff((size_t)N + 1);
  soff[0] = 0;
  for (int i = 0; i < N; ++i) soff[i + 1] = soff[i] + 6 * (int64_t)(nptr[i + 1] - nptr[i]);
  std::vector<int32_t> scratch((size_t)soff[N]);
  S.rowptr.assign((size_t)N + 1, 0);
  parallel_for(N, nthreads, [&](int64_t b, int64_t e_, int) {
    for (int64_t i = b; i < e_; ++i) {
      int32_t* s = scratch.data() + soff[i];
      int n = 0;
      for (int32_t q = nptr[i]; q < nptr[i + 1]; ++q) {
        int32_t e = nadj[q];
        for (int a = 0; a < 6; ++a) s[n++] = S.edof[(size_t)a * ne + e];
      }
      std::sort(s, s + n);
      S.rowptr[i + 1] = (int32_t)(std::unique(s, s + n) - s);
    }
  });
  for (int i = 0; i < N; ++i) S.rowptr[i + 1] += S.rowptr[i];
  const int64_t nnz = S.rowptr[N];
  S.colind.resize(nnz);
  S.slot_row.resize(nnz);
  parallel_for(N, nthreads, [&](int64_t b, int64_t e_, int) {
    for (int64_t i = b; i < e_; ++i) {
      const int32_t* s = scratch.data() + soff[i];
      const int32_t r0 = S.rowptr[i], len = S.rowptr[i + 1] - r0;
      std::copy(s, s + len, S.colind.data() + r0);
      std::fill(S.slot_row.data() + r0, S.slot_row.data() + r0 + len, (int32_t)i);
    }
  });
}

// ------------------------------------------------------------------------------------------------
// element-based geometric nested dissection: complete binary tree of depth L over the elements
// ------------------------------------------------------------------------------------------------
void nd_tree(Symbolic& S, int leaf_elems, int nthreads) {
  const int ne = S.ne, N = S.N;
  int L = 0;
  while (L < 24 && (((int64_t)ne + ((int64_t)1 << L) - 1) >> L) > leaf_elems) ++L;
  while (L > 0 && ((int64_t)1 << L) > ne) --L;
  S.L = L;
  S.nfronts = (1 << (L + 1)) - 1;
  std::vector<double> cx(ne), cy(ne);
  const double* X = S.doflocs.data();
  const double* Y = X + N;
  for (int e = 0; e < ne; ++e) {
    int32_t a = S.tsorted[e], b = S.tsorted[(size_t)ne + e], c = S.tsorted[(size_t)2 * ne + e];
    cx[e] = (X[a] + X[b] + X[c]) / 3.0;
    cy[e] = (Y[a] + Y[b] + Y[c]) / 3.0;
  }
  // element extents along both axes (for counting elements a cut line would straddle)
  std::vector<double> exlo(ne), exhi(ne), eylo(ne), eyhi(ne);
  for (int e = 0; e < ne; ++e) {
    int32_t a = S.tsorted[e], b = S.tsorted[(size_t)ne + e], c = S.tsorted[(size_t)2 * ne + e];
    exlo[e] = std::min(X[a], std::min(X[b], X[c])); exhi[e] = std::max(X[a], std::max(X[b], X[c]));
    eylo[e] = std::min(Y[a], std::min(Y[b], Y[c])); eyhi[e] = std::max(Y[a], std::max(Y[b], Y[c]));
  }
  std::vector<int32_t> perm(ne);
  std::iota(perm.begin(), perm.end(), 0);
  S.leaf_of_elem.resize(ne);
  S.leaf_elem_ptr.assign((size_t)(1 << L) + 1, 0);
  constexpr int NBIN = 512;
  // Recursive bisection.  Large subdomains: the cut is an axis-parallel line chosen among NBIN-1
  // candidates per axis to minimise the number of straddled elements (~ separator size) subject to
  // a balance window; small subdomains: plain median split along the longer extent.
  std::function<void(int, int, int, int, int)> split = [&](int lo, int hi, int level, int idx, int depth_par) {
    if (level == L) {
      S.leaf_elem_ptr[idx] = lo;
      for (int q = lo; q < hi; ++q) S.leaf_of_elem[perm[q]] = idx;
      return;
    }
    const int n = hi - lo;
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
    for (int q = lo; q < hi; ++q) {
      int e = perm[q];
      x0 = std::min(x0, cx[e]); x1 = std::max(x1, cx[e]);
      y0 = std::min(y0, cy[e]); y1 = std::max(y1, cy[e]);
    }
    int mid = -1;
    const int remaining = L - level;            // every leaf below must stay non-empty
    const int min_side = std::max(1 << (remaining - 1), 1);
    if (n >= 192 && n >= 4 * min_side) {
      // both children must stay within a factor RHO of the ideal size ne / 2^(level+1): bounds the
      // leaf-size spread by RHO overall (no compounding), so batched front kernels stay balanced
      constexpr double RHO = 2.2;
      const double ideal = (double)ne / (double)((int64_t)2 << level);
      const double clo = ideal / RHO, chi = ideal * RHO;
      double best_cost = 1e300, best_thr = 0;
      int best_axis = -1;
      for (int axis = 0; axis < 2; ++axis) {
        const double a0 = axis ? y0 : x0, a1 = axis ? y1 : x1;
        if (!(a1 > a0)) continue;
        const std::vector<double>& c = axis ? cy : cx;
        const std::vector<double>& elo = axis ? eylo : exlo;
        const std::vector<double>& ehi = axis ? eyhi : exhi;
        const double scale = NBIN / (a1 - a0);
        int32_t cnt[NBIN + 1] = {0};
        int32_t diff[NBIN + 2] = {0};
        for (int q = lo; q < hi; ++q) {
          int e = perm[q];
          int bc = std::min(NBIN - 1, std::max(0, (int)((c[e] - a0) * scale)));
          cnt[bc]++;
          // thresholds t_j = a0 + j/scale, j = 1..NBIN-1; the element touches the cut line iff
          // elo <= t_j <= ehi (closed: a line running along mesh edges still costs its nodes)
          int j0 = (int)std::ceil((elo[e] - a0) * scale - 1e-9);
          int j1 = (int)std::floor((ehi[e] - a0) * scale + 1e-9);
          j0 = std::max(j0, 1); j1 = std::min(j1, NBIN - 1);
          if (j0 <= j1) { diff[j0]++; diff[j1 + 1]--; }
        }
        int64_t below = 0, strad = 0;
        for (int j = 1; j < NBIN; ++j) {
          below += cnt[j - 1];
          strad += diff[j];
          double f = (double)below / n;
          if (below < clo || below > chi || n - below < clo || n - below > chi) continue;
          if (below < min_side || n - below < min_side) continue;
          double cost = (double)strad * (1.0 + 0.5 * std::fabs(f - 0.5));
          if (cost < best_cost) { best_cost = cost; best_thr = a0 + j / scale; best_axis = axis; }
        }
      }
      if (best_axis >= 0) {
        const std::vector<double>& c = best_axis ? cy : cx;
        auto it = std::partition(perm.begin() + lo, perm.begin() + hi, [&](int32_t e) { return c[e] < best_thr; });
        mid = (int)(it - perm.begin());
        if (mid - lo < min_side || hi - mid < min_side) mid = -1;
      }
    }
    if (mid < 0) {
      const std::vector<double>& key = (x1 - x0 >= y1 - y0) ? cx : cy;
      mid = lo + n / 2;
      std::nth_element(perm.begin() + lo, perm.begin() + mid, perm.begin() + hi,
                       [&](int32_t a, int32_t b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });
    }
    if (depth_par > 0 && n > 8192) {
      std::thread th([&] { split(lo, mid, level + 1, 2 * idx, depth_par - 1); });
      split(mid, hi, level + 1, 2 * idx + 1, depth_par - 1);
      th.join();
    } else {
      split(lo, mid, level + 1, 2 * idx, 0);
      split(mid, hi, level + 1, 2 * idx + 1, 0);
    }
  };
  int par_depth = 0;
  while ((1 << par_depth) < nthreads) ++par_depth;
  split(0, ne, 0, 0, nthreads > 1 ? par_depth : 0);
  S.leaf_elem_ptr[(size_t)1 << L] = ne;
  S.leaf_elems = perm;
  // keep element ids ascending inside every leaf (deterministic assembly order)
  for (int lf = 0; lf < (1 << L); ++lf)
    std::sort(S.leaf_elems.begin() + S.leaf_elem_ptr[lf], S.leaf_elems.begin() + S.leaf_elem_ptr[lf + 1]);
}

inline int bitlen(uint32_t x) {
  int n = 0;
  while (x) { ++n; x >>= 1; }
  return n;
}

std::string build_fronts(Symbolic& S, int nthreads) {
  const std::vector<int32_t>& nptr = S.nptr;
  const std::vector<int32_t>& nadj = S.nadj;
  const int N = S.N, ne = S.ne, L = S.L, nf = S.nfronts;
  // owner front of every non-Dirichlet node = deepest tree node containing all its elements
  S.owner.assign(N, -1);
  bool orphan = false;
  parallel_for(N, nthreads, [&](int64_t b, int64_t e_, int) {
    for (int64_t i = b; i < e_; ++i) {
      if (nptr[i] == nptr[i + 1]) { orphan = true; continue; }
      if (S.bmask[i]) continue;
      uint32_t lo = 0xffffffffu, hi = 0;
      for (int32_t q = nptr[i]; q < nptr[i + 1]; ++q) {
        uint32_t lf = (uint32_t)S.leaf_of_elem[nadj[q]];
        lo = std::min(lo, lf);
        hi = std::max(hi, lf);
      }
      int level = L - bitlen(lo ^ hi);
      S.owner[i] = (1 << level) - 1 + (int)(lo >> (L - level));
    }
  }, 8192);
  if (orphan) return "mesh has a vertex that belongs to no element";
  // per-front node lists, bottom-up, level by level (fronts of one level are independent).
  // lists[f] = own (ascending ids) ++ boundary (ascending ids)
  std::vector<std::vector<int32_t>> own(nf), bnd(nf);
  std::vector<std::vector<int32_t>> inv0(nf), inv1(nf);   // parent local (unpadded own++bnd) -> child bnd index
  const int leaf0 = (1 << L) - 1;
  parallel_for((int64_t)1 << L, nthreads, [&](int64_t b, int64_t e_, int) {
    std::vector<int32_t> tmp;
    for (int64_t lf = b; lf < e_; ++lf) {
      int f = leaf0 + (int)lf;
      tmp.clear();
      for (int32_t q = S.leaf_elem_ptr[lf]; q < S.leaf_elem_ptr[lf + 1]; ++q) {
        int32_t e = S.leaf_elems[q];
        for (int a = 0; a < 6; ++a) {
          int32_t i = S.edof[(size_t)a * ne + e];
          if (!S.bmask[i]) tmp.push_back(i);
        }
      }
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      own[f].reserve(tmp.size());
      bnd[f].reserve(tmp.size());
      for (int32_t i : tmp) (S.owner[i] == f ? own[f] : bnd[f]).push_back(i);
    }
  }, 64);
  for (int lev = L - 1; lev >= 0; --lev) {
    const int first = (1 << lev) - 1;
    parallel_for((int64_t)1 << lev, nthreads, [&](int64_t b, int64_t e_, int) {
      for (int64_t q = b; q < e_; ++q) {
        const int f = first + (int)q;
        const auto& b0 = bnd[2 * f + 1];
        const auto& b1 = bnd[2 * f + 2];
        const size_t cap = b0.size() + b1.size();
        std::vector<int32_t> bn, o0, o1, c0, c1;
        bn.reserve(cap); c0.reserve(cap); c1.reserve(cap);
        size_t i0 = 0, i1 = 0;
        while (i0 < b0.size() || i1 < b1.size()) {     // merge the two ascending boundary lists
          int32_t v, p0 = -1, p1 = -1;
          if (i1 >= b1.size() || (i0 < b0.size() && b0[i0] < b1[i1])) { v = b0[i0]; p0 = (int32_t)i0++; }
          else if (i0 >= b0.size() || b1[i1] < b0[i0]) { v = b1[i1]; p1 = (int32_t)i1++; }
          else { v = b0[i0]; p0 = (int32_t)i0++; p1 = (int32_t)i1++; }
          if (S.owner[v] == f) { own[f].push_back(v); o0.push_back(p0); o1.push_back(p1); }
          else { bn.push_back(v); c0.push_back(p0); c1.push_back(p1); }
        }
        bnd[f] = std::move(bn);
        o0.insert(o0.end(), c0.begin(), c0.end());
        o1.insert(o1.end(), c1.begin(), c1.end());
        inv0[f] = std::move(o0);
        inv1[f] = std::move(o1);
      }
    }, 2);
  }
  if (!bnd[0].empty()) return "internal error: root front has boundary nodes";
  {
    int64_t tot = 0;
    for (int f = 0; f < nf; ++f) tot += (int64_t)own[f].size();
    if (tot != S.nsolve) return "internal error: owned nodes do not partition the interior DOFs";
  }
  // flatten with padding to multiples of 8 nodes (16 DOFs)
  S.fs.resize(nf); S.fb.resize(nf); S.fs_true.resize(nf); S.fb_true.resize(nf);
  S.fnode_ptr.assign((size_t)nf + 1, 0);
  S.foff.assign((size_t)nf + 1, 0);
  S.factor_flops = 0; S.solve_entries = 0; S.max_m = 0;
  for (int f = 0; f < nf; ++f) {
    S.fs_true[f] = (int32_t)own[f].size();
    S.fb_true[f] = (int32_t)bnd[f].size();
    S.fs[f] = pad8(S.fs_true[f]);
    S.fb[f] = pad8(S.fb_true[f]);
    int64_t mn = S.fs[f] + S.fb[f];
    S.fnode_ptr[f + 1] = S.fnode_ptr[f] + mn;
    int64_t m = 2 * mn, s2 = 2 * (int64_t)S.fs[f];
    S.foff[f + 1] = S.foff[f] + m * m;
    S.factor_flops += (double)s2 * (double)m * (double)m;     // ~ block LDL^T + triangular inverse
    S.solve_entries += s2 * (m + (m - s2));
    S.max_m = std::max<int>(S.max_m, (int)m);
  }
  const int64_t tot = S.fnode_ptr[nf];
  S.fnodes.assign(tot, -1);
  S.cinv0.assign(tot, -1);
  S.cinv1.assign(tot, -1);
  S.epos.assign((size_t)6 * ne, -1);
  parallel_for(nf, nthreads, [&](int64_t b, int64_t e_, int) {
    for (int64_t f = b; f < e_; ++f) {
      int32_t* fn = S.fnodes.data() + S.fnode_ptr[f];
      std::copy(own[f].begin(), own[f].end(), fn);
      std::copy(bnd[f].begin(), bnd[f].end(), fn + S.fs[f]);
      if (f < leaf0) {
        int32_t* c0 = S.cinv0.data() + S.fnode_ptr[f];
        int32_t* c1 = S.cinv1.data() + S.fnode_ptr[f];
        const int so = S.fs_true[f];
        for (int q = 0; q < so; ++q) { c0[q] = inv0[f][q]; c1[q] = inv1[f][q]; }
        for (int q = 0; q < S.fb_true[f]; ++q) { c0[S.fs[f] + q] = inv0[f][so + q]; c1[S.fs[f] + q] = inv1[f][so + q]; }
      } else {
        // element node positions inside their leaf front (binary search in the two ascending lists)
        const int lf = (int)f - leaf0;
        for (int32_t q = S.leaf_elem_ptr[lf]; q < S.leaf_elem_ptr[lf + 1]; ++q) {
          int32_t e = S.leaf_elems[q];
          for (int a = 0; a < 6; ++a) {
            int32_t i = S.edof[(size_t)a * ne + e];
            if (S.bmask[i]) continue;
            int32_t pos;
            if (S.owner[i] == (int)f) pos = (int32_t)(std::lower_bound(own[f].begin(), own[f].end(), i) - own[f].begin());
            else pos = S.fs[f] + (int32_t)(std::lower_bound(bnd[f].begin(), bnd[f].end(), i) - bnd[f].begin());
            S.epos[(size_t)a * ne + e] = pos;
          }
        }
      }
    }
  }, 64);
  return "";
}

}  // namespace

std::string numbering_only(int nv, int ne, const double* p, const int32_t* t, Symbolic& S) {
  if (nv < 3 || ne < 1) return "empty mesh";
  return p2_numbering(nv, ne, p, t, S);
}

std::string build_symbolic(int nv, int ne, const double* p, const int32_t* t, int leaf_elems,
                           int nthreads, Symbolic& S) {
  if (nv < 3 || ne < 1) return "empty mesh";
  if (leaf_elems < 1) leaf_elems = 16;
  if (nthreads < 1) nthreads = 1;
  std::unique_ptr<Pool> pool;
  if (nthreads > 1) pool.reset(new Pool(nthreads));
  g_pool = pool.get();
  struct Reset { ~Reset() { g_pool = nullptr; } } reset_guard;
  auto t0 = clk::now();
  std::string err = p2_numbering(nv, ne, p, t, S);
  if (!err.empty()) return err;
  if (S.nsolve < 1) return "mesh has no interior DOF";
  auto t1 = clk::now();
  node_to_elem(S);
  csr_pattern(S, nthreads);
  auto t2 = clk::now();
  nd_tree(S, leaf_elems, nthreads);
  auto t3 = clk::now();
  err = build_fronts(S, nthreads);
  auto t4 = clk::now();
  S.t_numbering = secs(t0, t1);
  S.t_pattern = secs(t1, t2);
  S.t_tree = secs(t2, t3);
  S.t_fronts = secs(t3, t4);
  return err;
}

}  // namespace plfem
