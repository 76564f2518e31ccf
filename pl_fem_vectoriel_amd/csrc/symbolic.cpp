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
#include <condition_variable>
#include <mutex>
#include <thread>
#include <pthread.h>

namespace plfem {
namespace {

using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) {
  return std::chrono::duration<double>(b - a).count();
}

// The pool's waits are short (the regions are 10-500 us apart): spin on the cache line with a pause instruction
// first -- a yield costs a system call and several microseconds of wake-up latency per region, ~100 regions per
// analysis -- and only fall back to yielding when the wait drags on (oversubscribed host).
inline void cpu_relax(int spins) {
  if (spins < 4096) __builtin_ia32_pause();
  else std::this_thread::yield();
}

// Type-erased reference to a callable void(int rank) that outlives the call (fork-join: the caller waits).
struct FnRef {
  void (*call)(void*, int) = nullptr;
  void* obj = nullptr;
};
template <class F>
FnRef fn_ref(F& f) {
  return FnRef{[](void* o, int r) { (*static_cast<F*>(o))(r); }, &f};
}

// Worker pool of one analysis.  The parallel regions of the analysis are 0.05-1 ms each, so
//   * the workers live across analyses (a process-wide cache of idle pools: creating and joining 15 threads cost
//     0.3-0.5 ms per analysis), parked on a condition variable between analyses and spin-waiting inside one;
//   * every worker has its own mailbox, so DISJOINT thread ranges ("teams") work independently: a team's leader
//     (its first thread) hands a job to the members [first + 1, first + count) and joins them, or splits the team in
//     two (fork2: the second half's first thread becomes a leader of its own).  That is what lets the numbering chain
//     run on a few threads beside the bisection tree, and sibling subdomains of the top tree levels be bisected at
//     the same time, each by its share of the threads, instead of one after the other by all of them.
// Thread 0 of a pool is the caller of build_symbolic.
struct Pool {
  struct alignas(128) Slot {
    std::atomic<uint32_t> seq{0};       // bumped by the dispatcher when a job is posted
    std::atomic<uint32_t> taken{0};     // = seq once somebody has claimed the job: the worker, or the leader in its stead
    std::atomic<uint32_t> ack{0};       // = seq once the job is done
    FnRef fn;
    int rank = 0, team_first = 0, team_count = 1;
    bool stealable = false;             // (a plain share of a parallel loop: whoever runs it, the result is the same)
  };
  const int nt;
  std::unique_ptr<Slot[]> slots;
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<bool> active{false}, stop{false}, failed{false};
  const double linger_ms = getenv("PLFEM_POOL_LINGER_MS") ? atof(getenv("PLFEM_POOL_LINGER_MS")) : 0.0;
  std::exception_ptr error;              // first exception thrown inside a worker's job (rethrown by the leader's join)
  explicit Pool(int n);
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop.store(true, std::memory_order_release);
    }
    cv.notify_all();
    for (auto& x : th) x.join();
  }
  void begin() {                         // wake the workers (they spin from here on)
    {
      std::lock_guard<std::mutex> lk(mu);
      active.store(true, std::memory_order_release);
    }
    cv.notify_all();
  }
  void end() { active.store(false, std::memory_order_release); }   // the workers park after a few hundred idle spins
  void rethrow() {                       // leader side, after a join
    if (!failed.load(std::memory_order_acquire)) return;
    std::exception_ptr e;
    {
      std::lock_guard<std::mutex> lk(mu);
      e = error;
      error = nullptr;
      failed.store(false, std::memory_order_release);
    }
    if (e) std::rethrow_exception(e);
  }
  void post(int thread, FnRef fn, int rank, int team_first, int team_count, bool stealable) {
    Slot& s = slots[thread];
    s.fn = fn; s.rank = rank; s.team_first = team_first; s.team_count = team_count; s.stealable = stealable;
    s.seq.store(s.seq.load(std::memory_order_relaxed) + 1, std::memory_order_release);
  }
  // Leader side.  A member that has not STARTED its share yet -- still waking from its condition variable after 17 ms of
  // GPU work, or pushed off its core by another tenant of the shared host -- does not hold the team up: the leader claims
  // the share and runs it itself (the share is a function of the rank alone, so the result is the same); only a member
  // that is in the middle of its share has to be waited for.
  void join(int thread) {
    Slot& s = slots[thread];
    const uint32_t want = s.seq.load(std::memory_order_relaxed);
    if (s.ack.load(std::memory_order_acquire) == want) return;
    if (s.stealable) {
      uint32_t expect = want - 1;
      if (s.taken.compare_exchange_strong(expect, want, std::memory_order_acq_rel)) {
        try {
          s.fn.call(s.fn.obj, s.rank);
        } catch (...) {                  // (join runs inside a destructor: recorded like a worker's, rethrown by rethrow())
          std::lock_guard<std::mutex> lk(mu);
          if (!error) error = std::current_exception();
          failed.store(true, std::memory_order_release);
        }
        s.ack.store(want, std::memory_order_release);
        return;
      }
    }
    for (int spins = 0; s.ack.load(std::memory_order_acquire) != want; ++spins) cpu_relax(spins);
  }
};

// the team the current thread leads (count = 1: no members, everything it starts runs serially)
struct TeamCtx {
  Pool* pool = nullptr;
  int first = 0, count = 1;
};
thread_local TeamCtx g_team;
inline int team_size() { return g_team.pool ? g_team.count : 1; }
struct TeamScope {                       // sets the calling thread's team for a scope
  TeamCtx saved;
  TeamScope(Pool* p, int first, int count) : saved(g_team) { g_team = TeamCtx{p, first, count}; }
  ~TeamScope() { g_team = saved; }
};

Pool::Pool(int n) : nt(n), slots(new Slot[n]) {
  for (int t = 1; t < nt; ++t)
    th.emplace_back([this, t] {
      Slot& s = slots[t];
      uint32_t seen = 0;
      while (true) {
        // Wait for the next job.  Inside an analysis (active): spin.  Between analyses the workers park on the condition
        // variable -- after lingering (still spinning, as OpenMP runtimes do after a parallel region) for PLFEM_POOL_LINGER_MS
        // if that is set.  Workers woken from a condition variable come up on idle cores (deep C-state, cold caches): measured
        // on the MI355X host, the analysis of C1 takes 3.7 ms that way against 2.85 ms back to back.  But lingering is OFF by
        // default: the GPU boxes give a process a CPU QUOTA (16 cores per GPU), and 15 + 7 workers spinning through the 17 ms
        // of GPU work between two cold solves exhaust it -- the kernel then throttles the whole process for the rest of the
        // accounting period: every third step of the bench took 50 ms instead of 21 (measured with a linger of 30 ms).
        clk::time_point idle_since{};
        bool idle = false;
        for (int spins = 0; s.seq.load(std::memory_order_acquire) == seen; ++spins) {
          if (stop.load(std::memory_order_acquire)) return;
          if (active.load(std::memory_order_acquire)) {
            if (idle) { idle = false; spins = 0; }
            cpu_relax(spins);
            continue;
          }
          if (spins <= 256) { __builtin_ia32_pause(); continue; }     // (the owner may be about to post the last jobs)
          if (!idle) { idle = true; idle_since = clk::now(); }
          if (linger_ms > 0 && ((spins & 1023) != 0 || secs(idle_since, clk::now()) * 1e3 < linger_ms)) {
            __builtin_ia32_pause();
            continue;
          }
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return active.load(std::memory_order_acquire) || stop.load(std::memory_order_acquire); });
          if (stop.load(std::memory_order_acquire)) return;
          idle = false;
          spins = 0;
        }
        seen = s.seq.load(std::memory_order_acquire);
        {
          uint32_t expect = seen - 1;
          if (!s.taken.compare_exchange_strong(expect, seen, std::memory_order_acq_rel)) continue;   // the leader ran it
        }
        try {
          TeamScope scope(this, s.team_first, s.team_count);
          s.fn.call(s.fn.obj, s.rank);
        } catch (...) {                  // (an exception must not leave a thread body)
          std::lock_guard<std::mutex> lk(mu);
          if (!error) error = std::current_exception();
          failed.store(true, std::memory_order_release);
        }
        s.ack.store(seen, std::memory_order_release);
      }
    });
}

// Process-wide cache of idle pools (a sweep prepares several analyses at once, each on its own pool).  The pools are
// destroyed by an atexit handler; a forked child starts with an empty cache (its parent's worker threads do not exist
// in it).
struct PoolCache {
  std::mutex mu;
  std::vector<Pool*> idle;
  bool registered = false;
};
PoolCache& pool_cache() { static PoolCache* c = new PoolCache(); return *c; }
void pool_cache_shutdown() {
  PoolCache& c = pool_cache();
  std::vector<Pool*> all;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    all.swap(c.idle);
  }
  for (Pool* p : all) delete p;
}
void pool_cache_forked() {                // (child side of fork: no locks held by design: only forget the pools)
  PoolCache& c = pool_cache();
  new (&c.mu) std::mutex();
  for (Pool*& p : c.idle) p = nullptr;    // leaked on purpose: their threads are gone
  c.idle.clear();
}
Pool* pool_acquire(int nthreads) {
  PoolCache& c = pool_cache();
  {
    std::lock_guard<std::mutex> lk(c.mu);
    if (!c.registered) {
      c.registered = true;
      std::atexit(pool_cache_shutdown);
      pthread_atfork(nullptr, nullptr, pool_cache_forked);
    }
    for (size_t q = 0; q < c.idle.size(); ++q)
      if (c.idle[q]->nt == nthreads) {
        Pool* p = c.idle[q];
        c.idle.erase(c.idle.begin() + q);
        return p;
      }
  }
  return new Pool(nthreads);
}
void pool_release(Pool* p) {
  PoolCache& c = pool_cache();
  {
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.idle.size() < 8) { c.idle.push_back(p); return; }
  }
  delete p;
}

// f(rank) on every thread of the current team, the caller as rank 0.  Inside the job a thread leads a team of one.
template <class F>
void team_run(F&& f) {
  const TeamCtx T = g_team;
  if (!T.pool || T.count <= 1) { f(0); return; }
  auto body = [&](int rank) { f(rank); };
  FnRef ref = fn_ref(body);
  for (int r = 1; r < T.count; ++r) T.pool->post(T.first + r, ref, r, T.first + r, 1, true);
  {
    TeamScope alone(T.pool, T.first, 1);   // (also while the leader runs shares it claimed from members that had not started)
    struct Join {                          // the members reference this frame: join them on every way out
      const TeamCtx& T;
      ~Join() { for (int r = 1; r < T.count; ++r) T.pool->join(T.first + r); }
    } join{T};
    f(0);
  }
  T.pool->rethrow();
}

// fa() on the first count_a threads of the current team, fb() on the rest, side by side; each half is a team of its own
template <class FA, class FB>
void team_fork2(int count_a, FA&& fa, FB&& fb) {
  const TeamCtx T = g_team;
  if (!T.pool || T.count <= 1 || count_a <= 0 || count_a >= T.count) { fa(); fb(); return; }
  auto body = [&](int) { fb(); };
  FnRef ref = fn_ref(body);
  const int mid = T.first + count_a;
  T.pool->post(mid, ref, 0, mid, T.count - count_a, false);   // (the other half needs its own leader: never claimed back)
  struct Join {
    Pool* pool; int mid;
    ~Join() { pool->join(mid); }
  };
  {
    Join join{T.pool, mid};
    TeamScope half(T.pool, T.first, count_a);
    fa();
  }
  T.pool->rethrow();
}

// run f(begin, end, rank) over [0, n) in static chunks on the current team
template <class F>
void parallel_for(int64_t n, int nthreads, F f, int64_t min_parallel = 4096) {
  const int nt = team_size();
  if (nthreads <= 1 || n < min_parallel || nt <= 1) {
    f((int64_t)0, n, 0);
    return;
  }
  const int64_t chunk = (n + nt - 1) / nt;
  team_run([&](int rank) {
    int64_t b = rank * chunk, e = std::min(n, b + chunk);
    if (b < e) f(b, e, rank);
  });
}

// run f(task) for task in [0, ntasks), tasks handed out one at a time (uneven task sizes)
template <class F>
void parallel_tasks(int ntasks, int nthreads, F f) {
  if (nthreads <= 1 || ntasks <= 1 || team_size() <= 1) {
    for (int q = 0; q < ntasks; ++q) f(q);
    return;
  }
  std::atomic<int> next{0};
  team_run([&](int) {
    for (int q = next.fetch_add(1, std::memory_order_relaxed); q < ntasks; q = next.fetch_add(1, std::memory_order_relaxed)) f(q);
  });
}

// fronts are padded to multiples of 16 DOFs (the MFMA tile): 8 nodes at two DOFs per node, 16 at one
inline int pad_nodes(int x, int dpn) { const int q = 16 / dpn; return (x + q - 1) / q * q; }

// PLFEM_SYM_TRACE=1: sub-phase wall times of the analysis on stderr (tuning aid)
struct Trace {
  bool on = getenv("PLFEM_SYM_TRACE") != nullptr;
  clk::time_point t = clk::now();
  void lap(const char* what) {
    if (!on) return;
    auto n = clk::now();
    fprintf(stderr, "[sym] %-22s %8.3f ms\n", what, secs(t, n) * 1e3);
    t = n;
  }
};

// ------------------------------------------------------------------------------------------------
// P2 numbering, scikit-fem compatible (MeshTri sort_t + build_entities + ElementTriP2 dof layout)
// ------------------------------------------------------------------------------------------------
// Part 1: validated, column-sorted element table (all the bisection tree needs besides the vertex coordinates).
std::string p2_sort_columns(int nv, int ne, const int32_t* t, Symbolic& S) {
  Trace tr;
  S.nv = nv;
  S.ne = ne;
  S.tsorted.resize((size_t)3 * ne);
  int32_t* t0 = S.tsorted.data();
  int32_t* t1 = t0 + ne;
  int32_t* t2 = t1 + ne;
  std::atomic<int> bad_t{0};
  parallel_for(ne, team_size(), [&](int64_t eb_, int64_t ee_, int) {
    for (int64_t e = eb_; e < ee_; ++e) {
      int32_t a = t[e], b = t[ne + e], c = t[2 * (size_t)ne + e];
      if (a < 0 || b < 0 || c < 0 || a >= nv || b >= nv || c >= nv) { bad_t.store(1); continue; }
      if (a > b) std::swap(a, b);
      if (b > c) std::swap(b, c);
      if (a > b) std::swap(a, b);
      if (a == b || b == c) { bad_t.store(2); continue; }
      t0[e] = a; t1[e] = b; t2[e] = c;
    }
  });
  if (bad_t.load() == 1) return "mesh.t refers to a vertex outside mesh.p";
  if (bad_t.load() == 2) return "degenerate element (repeated vertex)";
  tr.lap("num: sort columns");
  return "";
}

// Part 2: edges, element DOFs, DOF locations, boundary mask, interior list (reads tsorted, writes nothing the tree reads)
std::string p2_edges_and_dofs(int nv, int ne, const double* p, Symbolic& S) {
  Trace tr;
  const int32_t* t0 = S.tsorted.data();
  const int32_t* t1 = t0 + ne;
  const int32_t* t2 = t1 + ne;
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
  tr.lap("num: bucket edges");
  // sort + unique each bucket; edge id = running count => lexicographic (min, max) rank
  // (two passes so that the per-vertex work runs on the worker pool: sort + count, prefix, write)
  std::vector<int32_t> eoff((size_t)nv + 1, 0);
  std::atomic<int> bad{0};
  const int nth = team_size();
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
  tr.lap("num: unique edges");
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
  tr.lap("num: edof");
  S.doflocs.resize((size_t)2 * N);
  std::memcpy(S.doflocs.data(), p, sizeof(double) * nv);
  std::memcpy(S.doflocs.data() + N, p + nv, sizeof(double) * nv);
  parallel_for(S.nedges, nth, [&](int64_t k0, int64_t k1, int) {
    for (int64_t k = k0; k < k1; ++k) {
      S.doflocs[nv + k] = 0.5 * (p[ea[k]] + p[eb[k]]);
      S.doflocs[(size_t)N + nv + k] = 0.5 * (p[nv + ea[k]] + p[nv + eb[k]]);
    }
  });
  tr.lap("num: doflocs");
  // boundary = vertices and mid-edge node of every edge with a single adjacent element; every byte of bmask has
  // one writer value (1), so the edge loop can run on the pool
  S.bmask.assign(N, 0);
  uint8_t* bm = S.bmask.data();
  if (S.dirichlet) parallel_for(S.nedges, nth, [&](int64_t k0, int64_t k1, int) {
    for (int64_t k = k0; k < k1; ++k)
      if (mult[k] == 1) {   // several edges may mark the same vertex: relaxed atomic stores of the same value
        __atomic_store_n(&bm[ea[k]], (uint8_t)1, __ATOMIC_RELAXED);
        __atomic_store_n(&bm[eb[k]], (uint8_t)1, __ATOMIC_RELAXED);
        bm[nv + k] = 1;
      }
  });
  // interior list / inverse map: per-chunk counts, prefix, fill
  S.int_index.resize(N);
  {
    const int nt = nth > 1 && N >= 4096 ? team_size() : 1;
    const int64_t chunk = ((int64_t)N + nt - 1) / nt;
    std::vector<int32_t> cnt((size_t)nt + 1, 0);
    parallel_for(N, nt, [&](int64_t b, int64_t e_, int tid) {
      int32_t c = 0;
      for (int64_t i = b; i < e_; ++i) c += bm[i] == 0;
      cnt[tid + 1] = c;
    }, 1);
    for (int t2 = 0; t2 < nt; ++t2) cnt[t2 + 1] += cnt[t2];
    S.interior.resize(cnt[nt]);
    parallel_for(N, nt, [&](int64_t b, int64_t e_, int tid) {
      int32_t pos = cnt[tid];
      for (int64_t i = b; i < e_; ++i) {
        if (bm[i]) S.int_index[i] = -1;
        else { S.int_index[i] = pos; S.interior[pos++] = (int32_t)i; }
      }
    }, 1);
    (void)chunk;
  }
  S.nsolve = (int)S.interior.size();
  tr.lap("num: boundary + interior");
  return "";
}

// node -> adjacent elements (CSR), elements ascending within each node.  No two threads ever write the same node's list
// (adjacent lists share cache lines: element-range partitions, tried in round 4 with private counters and with atomic
// cursors, bounce those lines between cores and get SLOWER with every thread added on the 256-CPU host: 0.9 ms on 5
// threads, 4.6 ms on 8, against 0.4 ms serial).  Instead the NODES are partitioned: vertex nodes only occur in rows 0-2 of
// the element table, edge nodes only in rows 3-5, and each class is cut into contiguous node ranges; a thread scans the
// three rows of its class in element order and files the nodes of its range -- half the table per thread (round 3: all of
// it), ascending element ids without sorting, the same result for every team size.
void node_to_elem(Symbolic& S, int nthreads) {
  const int N = S.N, ne = S.ne, nv = S.nv;
  std::vector<int32_t>& ptr = S.nptr;
  std::vector<int32_t>& adj = S.nadj;
  std::vector<uint8_t>& loc = S.nloc;
  ptr.assign((size_t)N + 1, 0);
  adj.resize((size_t)6 * ne);
  loc.resize((size_t)6 * ne);
  const int32_t* d = S.edof.data();
  const int nt = (nthreads > 1 && ne >= 4096) ? team_size() : 1;
  // parts: (class, range).  With one thread: one part per class, run in sequence.
  const int per_class = std::max(1, nt / 2);
  struct Part { int a0, lo, hi; };
  std::vector<Part> parts;
  for (int cls = 0; cls < 2; ++cls) {
    const int base = cls == 0 ? 0 : nv, count = cls == 0 ? nv : N - nv;
    for (int r = 0; r < per_class; ++r)
      parts.push_back(Part{3 * cls, base + (int)((int64_t)count * r / per_class), base + (int)((int64_t)count * (r + 1) / per_class)});
  }
  auto for_parts = [&](auto&& body) {
    if (nt <= 1) { for (const Part& p : parts) body(p); return; }
    team_run([&](int rank) { for (size_t q = rank; q < parts.size(); q += nt) body(parts[q]); });
  };
  for_parts([&](const Part& p) {
    for (int a = p.a0; a < p.a0 + 3; ++a) {
      const int32_t* row = d + (size_t)a * ne;
      for (int e = 0; e < ne; ++e) {
        const int32_t i = row[e];
        if (i >= p.lo && i < p.hi) ptr[i + 1]++;
      }
    }
  });
  for (int i = 0; i < N; ++i) ptr[i + 1] += ptr[i];
  for_parts([&](const Part& p) {
    std::vector<int32_t> fill(ptr.begin() + p.lo, ptr.begin() + p.hi);
    const int32_t* r0 = d + (size_t)p.a0 * ne;
    const int32_t* r1 = r0 + ne;
    const int32_t* r2 = r1 + ne;
    for (int e = 0; e < ne; ++e) {
      const int32_t ii[3] = {r0[e], r1[e], r2[e]};
      for (int k = 0; k < 3; ++k) {
        const int32_t i = ii[k];
        if (i >= p.lo && i < p.hi) {
          int32_t& f = fill[i - p.lo];
          adj[f] = e;
          loc[f] = (uint8_t)(p.a0 + k);
          ++f;
        }
      }
    }
  });
}

// ------------------------------------------------------------------------------------------------
// scalar CSR pattern: row i = sorted union of the DOFs of the elements adjacent to node i.
// Row lengths have a closed form that needs no manifold assumption:
//   vertex v with d adjacent elements and a incident edges: itself + a neighbour vertices + a incident edges
//     + d opposite edges (one per element, all distinct)            = 1 + 2a + d
//   edge node with m adjacent elements (1 or 2): 2 + m vertices, itself + 2m other edges  = 3 + 3m
// so rowptr (and nnz) cost O(N); the column lists themselves are built by the device (k_pattern_fill) or,
// on demand, by ensure_pattern below.
// ------------------------------------------------------------------------------------------------
void csr_rowptr(Symbolic& S, int nthreads) {
  const int N = S.N, nv = S.nv, nedges = S.nedges;
  // edges incident to each vertex, by the same private-count scheme (the counters are only nv ints per thread)
  const int nt = (nthreads > 1 && nedges >= 8192) ? std::min(team_size(), 8) : 1;
  std::vector<int32_t> part((size_t)nt * nv, 0);
  const int32_t* ea = S.edges.data();
  const int64_t chunk = ((int64_t)2 * nedges + nt - 1) / nt;
  team_run([&](int rank) {
    if (rank >= nt) return;
    int32_t* c = part.data() + (size_t)rank * nv;
    const int64_t q0 = rank * chunk, q1 = std::min<int64_t>((int64_t)2 * nedges, q0 + chunk);
    for (int64_t q = q0; q < q1; ++q) c[ea[q]]++;
  });
  S.rowptr.assign((size_t)N + 1, 0);
  for (int i = 0; i < N; ++i) {
    const int d = S.nptr[i + 1] - S.nptr[i];
    int inc = 0;
    if (i < nv)
      for (int t = 0; t < nt; ++t) inc += part[(size_t)t * nv + i];
    S.rowptr[i + 1] = S.rowptr[i] + (i < nv ? 1 + 2 * inc + d : 3 + 3 * d);
  }
  S.colind.clear();
  S.slot_row.clear();
}

}  // namespace

void ensure_pattern(const Symbolic& S) {
  const int N = S.N, ne = S.ne;
  const int64_t nnz = S.rowptr.empty() ? 0 : S.rowptr[N];
  if ((int64_t)S.colind.size() == nnz && (int64_t)S.slot_row.size() == nnz) return;
  S.colind.resize(nnz);
  S.slot_row.resize(nnz);
  std::vector<int32_t> s;
  for (int i = 0; i < N; ++i) {
    s.clear();
    for (int32_t q = S.nptr[i]; q < S.nptr[i + 1]; ++q) {
      const int32_t e = S.nadj[q];
      for (int a = 0; a < 6; ++a) s.push_back(S.edof[(size_t)a * ne + e]);
    }
    std::sort(s.begin(), s.end());
    const int32_t len = (int32_t)(std::unique(s.begin(), s.end()) - s.begin());
    const int32_t r0 = S.rowptr[i];
    // the closed-form length is exact for every mesh p2_numbering accepts; keep the arrays in bounds regardless
    const int32_t n = std::min(len, S.rowptr[i + 1] - r0);
    std::copy(s.begin(), s.begin() + n, S.colind.begin() + r0);
    std::fill(S.colind.begin() + r0 + n, S.colind.begin() + S.rowptr[i + 1], -1);
    std::fill(S.slot_row.begin() + r0, S.slot_row.begin() + S.rowptr[i + 1], (int32_t)i);
  }
}

namespace {

// ------------------------------------------------------------------------------------------------
// element-based geometric nested dissection: complete binary tree of depth L over the elements
// ------------------------------------------------------------------------------------------------
// One record per element, moved around by the bisection (sort the data, not an index: every pass over a
// subdomain is then a sequential stream; at 64 B the records of C1 fit the L2 cache).
struct alignas(64) ElemGeo {
  double c[2];        // centroid x, y
  double lo[2];       // extent along x, y
  double hi[2];
  int32_t id;
};

void nd_tree(Symbolic& S, const double* p, int leaf_elems, int nthreads) {
  const int ne = S.ne;
  int L = 0;
  while (L < 24 && (((int64_t)ne + ((int64_t)1 << L) - 1) >> L) > leaf_elems) ++L;
  while (L > 0 && ((int64_t)1 << L) > ne) --L;
  S.L = L;
  S.nfronts = (1 << (L + 1)) - 1;
  Trace tr;
  // Two element-record buffers: every bisection SCATTERS its subdomain from one into the other (stable, left part
  // first), so a level costs no copy back and the children's bounding boxes come out of the same pass.
  // (default-initialised storage: a value-initialising vector would zero 2 x 64 B x ne on one thread first)
  std::vector<ElemGeo, default_init_allocator<ElemGeo>> bufA(ne), bufB(ne);
  ElemGeo* const buf[2] = {bufA.data(), bufB.data()};
  struct Box {
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;      // of the element CENTROIDS of a subdomain
    void add(const ElemGeo& g) {
      x0 = std::min(x0, g.c[0]); x1 = std::max(x1, g.c[0]);
      y0 = std::min(y0, g.c[1]); y1 = std::max(y1, g.c[1]);
    }
    void merge(const Box& o) {
      x0 = std::min(x0, o.x0); x1 = std::max(x1, o.x1);
      y0 = std::min(y0, o.y0); y1 = std::max(y1, o.y1);
    }
  };
  const double* X = p;                 // vertex coordinates of the caller's mesh (the DOF table is built concurrently)
  const double* Y = p + S.nv;
  Box root_box;
  {
    const int nt0 = nthreads > 1 ? team_size() : 1;
    std::vector<Box> tb(nt0);
    parallel_for(ne, nthreads, [&](int64_t b_, int64_t e_, int tid) {
      Box bx;
      for (int64_t e = b_; e < e_; ++e) {
        int32_t a = S.tsorted[e], b = S.tsorted[(size_t)ne + e], c = S.tsorted[(size_t)2 * ne + e];
        ElemGeo& g = buf[0][e];
        g.c[0] = (X[a] + X[b] + X[c]) / 3.0;
        g.c[1] = (Y[a] + Y[b] + Y[c]) / 3.0;
        g.lo[0] = std::min(X[a], std::min(X[b], X[c])); g.hi[0] = std::max(X[a], std::max(X[b], X[c]));
        g.lo[1] = std::min(Y[a], std::min(Y[b], Y[c])); g.hi[1] = std::max(Y[a], std::max(Y[b], Y[c]));
        g.id = (int32_t)e;
        bx.add(g);
      }
      tb[tid].merge(bx);
    });
    for (const Box& b : tb) root_box.merge(b);       // (min / max: the same for any thread count)
  }
  tr.lap("tree: centroids");
  S.leaf_of_elem.resize(ne);
  S.leaf_elem_ptr.assign((size_t)(1 << L) + 1, 0);
  S.leaf_elems.resize(ne);
  constexpr int NBIN = 512;
  struct Hist {
    int32_t cnt[3][NBIN + 1];      // "axis" 2 = distance from the subdomain's median point (circular cuts)
    int32_t diff[3][NBIN + 2];
    Box left, right;
  };
  struct Cut {
    int mid;
    Box left, right;
  };
  // Bisection of src[lo, hi) into dst[lo, mid) + dst[mid, hi).  Large subdomains: the cut is chosen among NBIN-1
  // candidates of three families -- lines parallel to either axis, circles around the median point -- to minimise
  // the number of straddled elements (~ separator size) subject to a balance window; small subdomains: plain median
  // split along the longer extent.  A candidate is a BIN boundary of the centroid histogram and an element goes left
  // iff its bin lies below it: the sizes of both parts (per thread chunk, for the stable parallel scatter) then come
  // out of the histograms, no counting pass.  par: the passes over the elements run on the pool (top of the tree,
  // where there are fewer nodes than threads).  The result does not depend on par or on the thread count.
  auto bisect = [&](const ElemGeo* src, ElemGeo* dst, int lo, int hi, int level, bool par, const Box& box) -> Cut {
    const int n = hi - lo;
    const int nt = par ? team_size() : 1;
    std::vector<Hist> hs(nt);
    const ElemGeo* Gl = src + lo;
    ElemGeo* Dl = dst + lo;
    const double x0 = box.x0, x1 = box.x1, y0 = box.y0, y1 = box.y1;
    Cut cut;
    cut.mid = -1;
    const int remaining = L - level;            // every leaf below must stay non-empty
    const int min_side = std::max(1 << (remaining - 1), 1);
    auto chunk_ran = [&](int t) { return t == 0 || (int64_t)t * ((n + nt - 1) / nt) < n; };   // else its arrays are stale
    if (n >= 192 && n >= 4 * min_side) {
      // both children must stay within a factor RHO of the ideal size ne / 2^(level+1): bounds the
      // leaf-size spread by RHO overall (no compounding), so batched front kernels stay balanced
      static const double RHO = getenv("PLFEM_RHO") ? atof(getenv("PLFEM_RHO")) : 2.2;
      const double ideal = (double)ne / (double)((int64_t)2 << level);
      const double clo = ideal / RHO, chi = ideal * RHO;
      const double a0s[2] = {x0, y0};
      const double scales[2] = {x1 > x0 ? NBIN / (x1 - x0) : 0.0, y1 > y0 ? NBIN / (y1 - y0) : 0.0};
      auto axis_bin = [&](const ElemGeo& g, int axis) {
        return std::min(NBIN - 1, std::max(0, (int)((g.c[axis] - a0s[axis]) * scales[axis])));
      };
      auto hist = [&](int64_t b, int64_t e_, int tid) {
        Hist& h = hs[tid];
        std::memset(h.cnt, 0, sizeof(h.cnt));
        std::memset(h.diff, 0, sizeof(h.diff));
        for (int64_t q = b; q < e_; ++q) {
          const ElemGeo& g = Gl[q];
          for (int axis = 0; axis < 2; ++axis) {
            const double a0 = a0s[axis], scale = scales[axis];
            if (!(scale > 0.0)) continue;
            h.cnt[axis][axis_bin(g, axis)]++;
            // thresholds t_j = a0 + j/scale, j = 1..NBIN-1; the element touches the cut line iff
            // elo <= t_j <= ehi (closed: a line running along mesh edges still costs its nodes)
            // ceil / floor by truncation + correction (the baseline x86-64 target has no rounding instruction,
            // std::ceil / std::floor would be libm calls in the innermost loop of the analysis)
            const double v0 = (g.lo[axis] - a0) * scale - 1e-9, v1 = (g.hi[axis] - a0) * scale + 1e-9;
            const int t0 = (int)v0, t1 = (int)v1;
            int j0 = t0 + (v0 > (double)t0);
            int j1 = t1 - (v1 < (double)t1);
            j0 = std::max(j0, 1); j1 = std::min(j1, NBIN - 1);
            if (j0 <= j1) { h.diff[axis][j0]++; h.diff[axis][j1 + 1]--; }
          }
        }
      };
      if (nt > 1) parallel_for(n, nt, hist, 1); else hist(0, n, 0);
      // Third family of cuts: circles around the median point O of the element centroids (from the two histograms,
      // integer counts: independent of the thread count).  The lantern meshes are polar inside every core -- a
      // straight cut through a core crosses every ring twice, a circle between two rings crosses one ring's worth
      // of elements -- and a subdomain dominated by one core has its median point near that core's centre.
      double O[2] = {0.0, 0.0}, rscale = 0.0;
      {
        for (int axis = 0; axis < 2; ++axis) {
          int64_t acc = 0;
          int j = 0;
          for (; j < NBIN; ++j) {
            for (int t = 0; t < nt; ++t) if (chunk_ran(t)) acc += hs[t].cnt[axis][j];
            if (2 * acc >= n) break;
          }
          O[axis] = scales[axis] > 0.0 ? a0s[axis] + (j + 0.5) / scales[axis] : a0s[axis];
        }
        const double ex = std::max(O[0] - x0, x1 - O[0]), ey = std::max(O[1] - y0, y1 - O[1]);
        const double rmax = std::sqrt(ex * ex + ey * ey);
        rscale = rmax > 0.0 ? NBIN / rmax : 0.0;
      }
      auto radial_bin = [&](const ElemGeo& g) {
        const double cx = g.c[0] - O[0], cy = g.c[1] - O[1];
        return std::min(NBIN - 1, std::max(0, (int)(std::sqrt(cx * cx + cy * cy) * rscale)));
      };
      static const int RADIAL_MIN = getenv("PLFEM_RADIAL_MIN") ? atoi(getenv("PLFEM_RADIAL_MIN")) : 2000;   // circular cuts pay in subdomains that still hold a whole core (measured: C1 flops 25.1 -> 24.7 G, level steps 125 -> 116)
      if (n < RADIAL_MIN) rscale = 0.0;
      if (rscale > 0.0) {
        auto rhist = [&](int64_t b, int64_t e_, int tid) {
          Hist& h = hs[tid];
          for (int64_t q = b; q < e_; ++q) {
            const ElemGeo& g = Gl[q];
            h.cnt[2][radial_bin(g)]++;
            // radial extent of the element's bounding box (a superset of the element: cost estimate only)
            const double nx = std::max(std::max(g.lo[0] - O[0], O[0] - g.hi[0]), 0.0);
            const double ny = std::max(std::max(g.lo[1] - O[1], O[1] - g.hi[1]), 0.0);
            const double fx = std::max(std::fabs(g.lo[0] - O[0]), std::fabs(g.hi[0] - O[0]));
            const double fy = std::max(std::fabs(g.lo[1] - O[1]), std::fabs(g.hi[1] - O[1]));
            const double v0 = std::sqrt(nx * nx + ny * ny) * rscale - 1e-9, v1 = std::sqrt(fx * fx + fy * fy) * rscale + 1e-9;
            const int t0 = (int)v0, t1 = (int)v1;
            int j0 = t0 + (v0 > (double)t0);
            int j1 = t1 - (v1 < (double)t1);
            j0 = std::max(j0, 1); j1 = std::min(j1, NBIN - 1);
            if (j0 <= j1) { h.diff[2][j0]++; h.diff[2][j1 + 1]--; }
          }
        };
        if (nt > 1) parallel_for(n, nt, rhist, 1); else rhist(0, n, 0);
      }
      const double scales3[3] = {scales[0], scales[1], rscale};
      double best_cost = 1e300;
      int best_axis = -1, best_j = 0;
      int64_t best_below = 0;
      for (int axis = 0; axis < 3; ++axis) {
        if (!(scales3[axis] > 0.0)) continue;
        int64_t below = 0, strad = 0;
        for (int j = 1; j < NBIN; ++j) {
          for (int t = 0; t < nt; ++t) {
            if (!chunk_ran(t)) break;
            below += hs[t].cnt[axis][j - 1];
            strad += hs[t].diff[axis][j];
          }
          double f = (double)below / n;
          if (below < clo || below > chi || n - below < clo || n - below > chi) continue;
          if (below < min_side || n - below < min_side) continue;
          double cost = (double)strad * (1.0 + 0.5 * std::fabs(f - 0.5));
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_j = j; best_below = below; }
        }
      }
      if (best_axis >= 0) {
        const int ax = best_axis;
        auto left = [&](const ElemGeo& g) { return (ax < 2 ? axis_bin(g, ax) : radial_bin(g)) < best_j; };
        // stable scatter: chunk t writes its left elements behind those of the chunks before it, its right elements
        // likewise behind the left part; the offsets are prefix sums of the chunks' histograms
        std::vector<int64_t> nl(nt + 1, 0);
        for (int t = 0; t < nt; ++t) {
          int64_t k = 0;
          if (chunk_ran(t))
            for (int j = 0; j < best_j; ++j) k += hs[t].cnt[ax][j];
          nl[t + 1] = nl[t] + k;
        }
        const int64_t nleft = nl[nt];
        auto scatter = [&](int64_t b, int64_t e_, int tid) {
          int64_t wl = nl[tid], wr = nleft + (b - nl[tid]);
          Box bl, br;
          for (int64_t q = b; q < e_; ++q) {
            const ElemGeo& g = Gl[q];
            if (left(g)) { Dl[wl++] = g; bl.add(g); } else { Dl[wr++] = g; br.add(g); }
          }
          hs[tid].left = bl;
          hs[tid].right = br;
        };
        if (nt > 1) parallel_for(n, nt, scatter, 1); else scatter(0, n, 0);
        for (int t = 0; t < nt; ++t)
          if (chunk_ran(t)) { cut.left.merge(hs[t].left); cut.right.merge(hs[t].right); }
        cut.mid = lo + (int)nleft;
        (void)best_below;
      }
    }
    if (cut.mid < 0) {
      // median split along the longer extent (small subdomains, or no admissible cut): order a copy in dst
      const int ax = (x1 - x0 >= y1 - y0) ? 0 : 1;
      std::memcpy((void*)Dl, (const void*)Gl, sizeof(ElemGeo) * (size_t)n);
      cut.mid = lo + n / 2;
      std::nth_element(Dl, Dl + n / 2, Dl + n, [&](const ElemGeo& a, const ElemGeo& b) {
        return a.c[ax] < b.c[ax] || (a.c[ax] == b.c[ax] && a.id < b.id);
      });
      for (int q = 0; q < n / 2; ++q) cut.left.add(Dl[q]);
      for (int q = n / 2; q < n; ++q) cut.right.add(Dl[q]);
    }
    return cut;
  };
  // src = index of the buffer that holds [lo, hi)
  std::function<void(int, int, int, int, int, const Box&)> subtree = [&](int lo, int hi, int level, int idx, int src, const Box& box) {
    if (level == L) {
      S.leaf_elem_ptr[idx] = lo;
      // element ids ascending inside every leaf (deterministic assembly order)
      for (int q = lo; q < hi; ++q) S.leaf_elems[q] = buf[src][q].id;
      std::sort(S.leaf_elems.begin() + lo, S.leaf_elems.begin() + hi);
      for (int q = lo; q < hi; ++q) S.leaf_of_elem[S.leaf_elems[q]] = idx;
      return;
    }
    const Cut c = bisect(buf[src], buf[src ^ 1], lo, hi, level, false, box);
    subtree(lo, c.mid, level + 1, 2 * idx, src ^ 1, c.left);
    subtree(c.mid, hi, level + 1, 2 * idx + 1, src ^ 1, c.right);
  };
  // Top of the tree: the passes over a subdomain run on the team that holds it, and after every bisection the team
  // splits with it -- sibling subdomains are bisected SIDE BY SIDE, each by its share of the threads (one after the
  // other by all threads, a top level of 2^l subdomains cost 2^l rounds of short synchronised passes: 1.2-1.5 ms of the
  // analysis of C1 went into the four top levels).  Below `top` levels: one task per subtree, handed out dynamically
  // over the whole team.  The result does not depend on the team sizes.
  struct Node { int lo, hi, idx; Box box; };
  int top = 0;
  const int nteam = nthreads > 1 ? team_size() : 1;
  if (nteam > 1) {
    // ~2 subtrees per thread (uneven subtrees: dynamic hand-out evens them out)
    const int want = getenv("PLFEM_TREE_SUBTREES") ? atoi(getenv("PLFEM_TREE_SUBTREES")) : 2 * nteam;
    while ((1 << top) < want) ++top;
    top = std::min(top, L);
  }
  std::vector<Node> cur((size_t)1 << top);
  std::function<void(const Node&, int)> descend = [&](const Node& nd, int lev) {
    if (lev == top) { cur[nd.idx] = nd; return; }
    const Cut c = bisect(buf[lev & 1], buf[(lev & 1) ^ 1], nd.lo, nd.hi, lev, team_size() > 1 && nd.hi - nd.lo >= 4096, nd.box);
    const Node l{nd.lo, c.mid, 2 * nd.idx, c.left}, r{c.mid, nd.hi, 2 * nd.idx + 1, c.right};
    const int nt = team_size();
    // threads in proportion to the halves' sizes (at least one each)
    const int na = nt > 1 ? std::min(nt - 1, std::max(1, (int)(((int64_t)nt * (c.mid - nd.lo) + (nd.hi - nd.lo) / 2) / std::max(1, nd.hi - nd.lo)))) : 0;
    team_fork2(na, [&] { descend(l, lev + 1); }, [&] { descend(r, lev + 1); });
  };
  descend(Node{0, ne, 0, root_box}, 0);
  const int level = top;
  tr.lap("tree: top levels");
  const int src0 = level & 1;
  parallel_tasks((int)cur.size(), nthreads, [&](int q) { subtree(cur[q].lo, cur[q].hi, level, cur[q].idx, src0, cur[q].box); });
  tr.lap("tree: subtrees");
  S.leaf_elem_ptr[(size_t)1 << L] = ne;
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
  Trace tr;
  S.owner.assign(N, -1);
  std::atomic<bool> orphan{false};
  parallel_for(N, nthreads, [&](int64_t b, int64_t e_, int) {
    for (int64_t i = b; i < e_; ++i) {
      if (nptr[i] == nptr[i + 1]) { orphan.store(true, std::memory_order_relaxed); continue; }
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
  if (orphan.load()) return "mesh has a vertex that belongs to no element";
  tr.lap("fronts: owner");
  // Per-front node lists, bottom-up, level by level (fronts of one level are independent):
  // own (ascending ids) and boundary (ascending ids).  One flat buffer per level and list, front q of the
  // level owning the slot [off[q], off[q+1]) sized by an upper bound (6 nodes per leaf element; the two
  // children's boundary lists for an internal front), so no per-front heap allocation.
  struct LevelBuf {
    std::vector<int64_t> off;
    rawvec_i32 own, bnd, o0, o1, c0, c1;     // o*/c*: index in child 0/1's boundary list of each own / boundary node
  };
  std::vector<LevelBuf> lv(L + 1);
  S.fs.resize(nf); S.fb.resize(nf); S.fs_true.resize(nf); S.fb_true.resize(nf);
  const int leaf0 = (1 << L) - 1, nleaf = 1 << L;
  S.epos.resize((size_t)6 * ne);                 // uninitialised: every (element, local node) entry is written below
  {
    // Leaf fronts from the ELEMENT side, one leaf per task: the nodes of a leaf's elements (6 per element, ~130 in all),
    // sorted and made unique, split into owned and boundary nodes -- ascending lists by construction -- and the position of
    // every (element, local node) pair in them (epos) through a per-thread node -> position table.  The leaves are
    // independent: no counters shared between threads, one pass.  (Round 3 walked the NODES twice with per-thread
    // cursors inside every leaf list: 0.55-0.85 ms of the analysis of C1 against ~0.15 ms this way.)
    LevelBuf& lb = lv[L];
    lb.off.assign((size_t)nleaf + 1, 0);
    for (int lf = 0; lf < nleaf; ++lf) lb.off[lf + 1] = lb.off[lf] + 6 * (int64_t)(S.leaf_elem_ptr[lf + 1] - S.leaf_elem_ptr[lf]);
    lb.own.resize(lb.off[nleaf]);
    lb.bnd.resize(lb.off[nleaf]);
    // element-major copy of the element DOF table: a leaf's elements are neighbours in space, not in id, and the [6][ne]
    // table costs six cache lines per element where this one costs half a line
    rawvec_i32 ed6((size_t)6 * ne);
    {
      const int32_t* ed = S.edof.data();
      parallel_for(ne, nthreads, [&](int64_t b, int64_t e_, int) {
        for (int64_t e = b; e < e_; ++e)
          for (int a = 0; a < 6; ++a) ed6[(size_t)e * 6 + a] = ed[(size_t)a * ne + e];      // six read streams, one write stream
      });
    }
    tr.lap("fronts: leaves (transpose)");
    constexpr int LB = 8;                          // leaves per task
    parallel_tasks((nleaf + LB - 1) / LB, nthreads, [&](int task) {
      std::vector<int32_t> nodes;
      nodes.reserve(1024);
      // node -> position in this leaf's lists; only entries written for the current leaf are ever read (no clearing)
      static thread_local rawvec_i32 pos_of;
      // node -> stamp of the leaf that listed it last (table and counter belong to the thread and outlive the analysis): a
      // node goes into the list once, so the sort below runs over the ~70 distinct nodes of a leaf, not its ~130 references
      static thread_local std::vector<uint32_t> seen;
      static thread_local uint32_t last_stamp = 0;
      if ((int)pos_of.size() < N) pos_of.resize(N);
      if ((int)seen.size() < N) { seen.assign(N, 0u); last_stamp = 0; }
      for (int lf = task * LB; lf < std::min(nleaf, (task + 1) * LB); ++lf) {
        const int e0 = S.leaf_elem_ptr[lf], e1 = S.leaf_elem_ptr[lf + 1];
        if (++last_stamp == 0) {                   // (wrapped after 4e9 leaves on this thread: start over with a clean table)
          std::fill(seen.begin(), seen.end(), 0u);
          last_stamp = 1;
        }
        const uint32_t stamp = last_stamp;
        nodes.clear();
        for (int q = e0; q < e1; ++q) {
          const int32_t e = S.leaf_elems[q];
          for (int a = 0; a < 6; ++a) {
            const int32_t i = ed6[(size_t)e * 6 + a];
            if (!S.bmask[i] && seen[i] != stamp) { seen[i] = stamp; nodes.push_back(i); }
          }
        }
        std::sort(nodes.begin(), nodes.end());
        int32_t* own = lb.own.data() + lb.off[lf];
        int32_t* bnd = lb.bnd.data() + lb.off[lf];
        int no = 0, nb = 0;
        for (int32_t i : nodes) {
          if (S.owner[i] == leaf0 + lf) { pos_of[i] = no; own[no++] = i; }
          else { pos_of[i] = -2 - nb; bnd[nb++] = i; }       // (boundary positions are offset by the padded owned count below)
        }
        S.fs_true[leaf0 + lf] = no;
        S.fb_true[leaf0 + lf] = nb;
        const int pad = pad_nodes(no, S.dpn);
        for (int q = e0; q < e1; ++q) {
          const int32_t e = S.leaf_elems[q];
          for (int a = 0; a < 6; ++a) {
            const int32_t i = ed6[(size_t)e * 6 + a];
            int32_t pos = -1;                      // Dirichlet node
            if (!S.bmask[i]) {
              const int32_t pp = pos_of[i];
              pos = pp >= 0 ? pp : pad + (-2 - pp);
            }
            S.epos[(size_t)q * 6 + a] = pos;        // (contiguous per leaf: by element id these were 132 scattered lines)
          }
        }
      }
    });
  }
  tr.lap("fronts: leaves");
  for (int lev = L - 1; lev >= 0; --lev) {
    const int first = (1 << lev) - 1, n = 1 << lev;
    LevelBuf& pb = lv[lev];
    const LevelBuf& cb = lv[lev + 1];
    pb.off.assign((size_t)n + 1, 0);
    for (int q = 0; q < n; ++q) {
      const int f = first + q;
      pb.off[q + 1] = pb.off[q] + S.fb_true[2 * f + 1] + S.fb_true[2 * f + 2];
    }
    const int64_t cap = pb.off[n];
    pb.own.resize(cap); pb.bnd.resize(cap); pb.o0.resize(cap); pb.o1.resize(cap); pb.c0.resize(cap); pb.c1.resize(cap);
    parallel_for(n, nthreads, [&](int64_t b, int64_t e_, int) {
      for (int64_t q = b; q < e_; ++q) {
        const int f = first + (int)q;
        const int32_t* b0 = cb.bnd.data() + cb.off[2 * q];
        const int32_t* b1 = cb.bnd.data() + cb.off[2 * q + 1];
        const int n0 = S.fb_true[2 * f + 1], n1 = S.fb_true[2 * f + 2];
        const int64_t o = pb.off[q];
        int32_t *ow = pb.own.data() + o, *bd = pb.bnd.data() + o;
        int32_t *o0 = pb.o0.data() + o, *o1 = pb.o1.data() + o, *c0 = pb.c0.data() + o, *c1 = pb.c1.data() + o;
        int no = 0, nb = 0, i0 = 0, i1 = 0;
        while (i0 < n0 || i1 < n1) {     // merge the two ascending boundary lists
          int32_t v, p0 = -1, p1 = -1;
          if (i1 >= n1 || (i0 < n0 && b0[i0] < b1[i1])) { v = b0[i0]; p0 = i0++; }
          else if (i0 >= n0 || b1[i1] < b0[i0]) { v = b1[i1]; p1 = i1++; }
          else { v = b0[i0]; p0 = i0++; p1 = i1++; }
          if (S.owner[v] == f) { ow[no] = v; o0[no] = p0; o1[no] = p1; ++no; }
          else { bd[nb] = v; c0[nb] = p0; c1[nb] = p1; ++nb; }
        }
        S.fs_true[f] = no;
        S.fb_true[f] = nb;
      }
    }, 2);
  }
  tr.lap("fronts: levels");
  if (S.fb_true[0] != 0) return "internal error: root front has boundary nodes";
  {
    int64_t tot = 0;
    for (int f = 0; f < nf; ++f) tot += S.fs_true[f];
    if (tot != S.nsolve) return "internal error: owned nodes do not partition the interior DOFs";
  }
  // flatten with padding to multiples of 16 DOFs
  S.fnode_ptr.assign((size_t)nf + 1, 0);
  S.foff.assign((size_t)nf + 1, 0);
  S.factor_flops = 0; S.solve_entries = 0; S.max_m = 0;
  for (int f = 0; f < nf; ++f) {
    S.fs[f] = pad_nodes(S.fs_true[f], S.dpn);
    S.fb[f] = pad_nodes(S.fb_true[f], S.dpn);
    int64_t mn = S.fs[f] + S.fb[f];
    S.fnode_ptr[f + 1] = S.fnode_ptr[f] + mn;
    int64_t m = S.dpn * mn, s2 = S.dpn * (int64_t)S.fs[f];
    S.foff[f + 1] = S.foff[f] + s2 * (m + (m - s2));     // [F11; F21] and Z^T; the Schur complement lives in a level arena
    S.factor_flops += (double)s2 * (double)m * (double)m;     // ~ block LDL^T + triangular inverse
    S.solve_entries += s2 * (m + (m - s2));
    S.max_m = std::max<int>(S.max_m, (int)m);
  }
  S.soff.assign((size_t)nf, 0);
  S.arena_doubles = 0;
  for (int lev = 0; (1 << lev) - 1 < nf; ++lev) {
    int64_t off = 0;
    for (int f = (1 << lev) - 1; f < std::min(nf, (1 << (lev + 1)) - 1); ++f) {
      const int64_t b2 = S.dpn * (int64_t)S.fb[f];
      S.soff[f] = off;
      off += b2 * b2;
    }
    S.arena_doubles = std::max(S.arena_doubles, off);
  }
  const int64_t tot = S.fnode_ptr[nf];
  S.fnodes.resize(tot);                  // uninitialised: every entry (padding included) is written below
  S.cinv0.resize(tot);
  S.cinv1.resize(tot);
  S.prow.resize(tot);
  S.npos.assign((size_t)N, -1);          // Dirichlet nodes stay -1
  auto put = [](int32_t* dst, const int32_t* src, int count, int padded) {
    if (src) std::copy(src, src + count, dst);
    else std::fill(dst, dst + count, -1);
    std::fill(dst + count, dst + padded, -1);
  };
  constexpr int FB = 32;                 // fronts per task
  // beside the flattening, on one thread of the team: the launch plan of the device kernels (plan.cpp) -- it needs the
  // front sizes and offsets computed above, not the lists being written here
  auto flatten = [&] {
  parallel_tasks((nf + FB - 1) / FB, nthreads, [&](int task) {
    for (int f = task * FB; f < std::min(nf, (task + 1) * FB); ++f) {
      const int level = bitlen((uint32_t)f + 1) - 1;
      const int64_t q = f - ((1 << level) - 1);
      const LevelBuf& b = lv[level];
      const int64_t o = b.off[q];
      const int no = S.fs_true[f], nb = S.fb_true[f], fs = S.fs[f], fb = S.fb[f];
      int32_t* fn = S.fnodes.data() + S.fnode_ptr[f];
      int32_t* c0 = S.cinv0.data() + S.fnode_ptr[f];
      int32_t* c1 = S.cinv1.data() + S.fnode_ptr[f];
      put(fn, b.own.data() + o, no, fs);
      put(fn + fs, b.bnd.data() + o, nb, fb);
      const bool leaf = f >= leaf0;
      put(c0, leaf ? nullptr : b.o0.data() + o, no, fs);
      put(c0 + fs, leaf ? nullptr : b.c0.data() + o, nb, fb);
      put(c1, leaf ? nullptr : b.o1.data() + o, no, fs);
      put(c1 + fs, leaf ? nullptr : b.c1.data() + o, nb, fb);
      // front order of the solve vectors: offset of the owned node's component 0, and (the inverse of the child maps)
      // the local node index of every boundary node of a child in THIS front -- each child slot has one writer
      const int64_t np = S.fnode_ptr[f];
      for (int q = 0; q < no; ++q) S.npos[fn[q]] = (int32_t)(2 * np + S.dpn * q);
      int32_t* pr = S.prow.data() + np;
      std::fill(pr, pr + fs, -1);                            // owned nodes have no slot in the parent
      if (f == 0) std::fill(pr + fs, pr + fs + fb, -1);      // (the root has no parent; its boundary list is empty)
      if (!leaf) {
        int32_t* p0 = S.prow.data() + S.fnode_ptr[2 * f + 1] + S.fs[2 * f + 1];
        int32_t* p1 = S.prow.data() + S.fnode_ptr[2 * f + 2] + S.fs[2 * f + 2];
        std::fill(p0, p0 + S.fb[2 * f + 1], -1);
        std::fill(p1, p1 + S.fb[2 * f + 2], -1);
        for (int q = 0; q < fs + fb; ++q) {
          if (c0[q] >= 0) p0[c0[q]] = q;
          if (c1[q] >= 0) p1[c1[q]] = q;
        }
      }
    }
  });
  };
  // (the plan is ~200 list-filling tasks of up to 20 us, 0.25 ms on two threads: a quarter of the team)
  const int n_plan = team_size() >= 4 ? std::max(2, team_size() / 4) : 1;
  double t_flat = 0.0, t_plan = 0.0;                     // (trace only: which side of the fork the lap below is made of)
  team_fork2(team_size() - n_plan, [&] { const auto t0 = clk::now(); flatten(); t_flat = secs(t0, clk::now()); }, [&] {
    const auto t0 = clk::now();
    build_launch_plan(S, S.plan, [&](int ntasks, const std::function<void(int)>& f) { parallel_tasks(ntasks, nthreads, f); });
    t_plan = secs(t0, clk::now());
  });
  if (tr.on) fprintf(stderr, "[sym]   (flatten %.3f ms | plan %.3f ms)\n", t_flat * 1e3, t_plan * 1e3);
  tr.lap("fronts: flatten + plan");
  return "";
}

}  // namespace

void host_parallel(int nthreads, const std::function<void(int, int)>& f) {
  if (nthreads <= 1) { f(0, 1); return; }
  struct Lease {
    Pool* pool;
    explicit Lease(int n) : pool(pool_acquire(n)) { pool->begin(); }
    ~Lease() { pool->end(); pool_release(pool); }
  } lease(nthreads);
  TeamScope whole(lease.pool, 0, nthreads);
  team_run([&](int rank) { f(rank, nthreads); });
}

std::string numbering_only(int nv, int ne, const double* p, const int32_t* t, Symbolic& S) {
  if (nv < 3 || ne < 1) return "empty mesh";
  std::string err = p2_sort_columns(nv, ne, t, S);
  if (!err.empty()) return err;
  return p2_edges_and_dofs(nv, ne, p, S);
}

std::string build_symbolic(int nv, int ne, const double* p, const int32_t* t, int leaf_elems,
                           int nthreads, Symbolic& S, int dofs_per_node, bool dirichlet) {
  if (nv < 3 || ne < 1) return "empty mesh";
  if ((int64_t)ne >= ((int64_t)1 << 28)) return "mesh too large: at most 2^28 - 1 elements (32-bit index structures)";
  if (dofs_per_node != 1 && dofs_per_node != 2) return "dofs_per_node must be 1 or 2";
  S.dpn = dofs_per_node;
  S.dirichlet = dirichlet;
  if (leaf_elems < 1) leaf_elems = 16;
  if (nthreads < 1) nthreads = 1;
  // the worker pool of this analysis (from the process-wide cache of idle pools) and the caller as leader of all of it
  struct PoolLease {
    Pool* pool = nullptr;
    explicit PoolLease(int n) { if (n > 1) { pool = pool_acquire(n); pool->begin(); } }
    ~PoolLease() { if (pool) { pool->end(); pool_release(pool); } }
  } lease(nthreads);
  TeamScope whole(lease.pool, 0, lease.pool ? nthreads : 1);
  auto t0 = clk::now();
  std::string err = p2_sort_columns(nv, ne, t, S);
  if (!err.empty()) return err;
  auto t1 = clk::now();
  // Two chains that touch disjoint data run side by side, each on its own team of the pool: the bisection tree (reads the
  // sorted element table and the caller's vertex coordinates) on three quarters of the threads, and the P2 numbering
  // (edges, element DOFs, DOF locations, boundary) followed by the node -> element adjacency and the CSR row pointers
  // on the rest.  (Round 3 ran the second chain serially on one extra thread: 3.3 ms beside a 2.1-ms tree on the MI355X
  // host -- it was the critical path.)  The fronts need both.
  double t_num = 0.0, t_side = 0.0;
  std::string err_side;
  auto side = [&] {
    try {                                        // (an exception must not leave a thread body)
      auto a = clk::now();
      err_side = p2_edges_and_dofs(nv, ne, p, S);
      auto b = clk::now();
      t_num = secs(a, b);
      if (!err_side.empty()) return;
      node_to_elem(S, team_size());
      csr_rowptr(S, team_size());
      t_side = secs(b, clk::now());
    } catch (const std::exception& e) {
      err_side = std::string("exception in the numbering chain: ") + e.what();
    }
  };
  static const bool chains_in_sequence = getenv("PLFEM_SYM_SEQUENTIAL") != nullptr;   // (A/B timing aid)
  static const int side_share = getenv("PLFEM_SIDE_THREADS") ? atoi(getenv("PLFEM_SIDE_THREADS")) : 0;
  if (nthreads > 1 && !chains_in_sequence) {
    const int n_side = std::min(nthreads - 1, std::max(1, side_share > 0 ? side_share : (nthreads + 1) / 3));
    team_fork2(nthreads - n_side, [&] { nd_tree(S, p, leaf_elems, nthreads); }, side);
  } else {
    side();
    if (err_side.empty()) nd_tree(S, p, leaf_elems, nthreads);
  }
  if (!err_side.empty()) return err_side;
  if (S.nsolve < 1) return "mesh has no interior DOF";
  auto t3 = clk::now();
  err = build_fronts(S, nthreads);
  auto t4 = clk::now();
  S.t_numbering = secs(t0, t1) + t_num;        // (with a pool: t_num and t_pattern overlap the tree)
  S.t_pattern = t_side;
  S.t_tree = secs(t1, t3) - (nthreads > 1 && !chains_in_sequence ? 0.0 : t_num + t_side);
  S.t_fronts = secs(t3, t4);
  return err;
}

}  // namespace plfem
