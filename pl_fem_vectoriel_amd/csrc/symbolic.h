// Host-side symbolic analysis for the P2 H-field eigenmode path (mesh-only; reusable across
// wavelengths / core indices of a sweep).  Pure C++17, no HIP: unit-testable without a GPU.
//
// Replaces, for this path, what the reference obtains from scikit-fem / SciPy:
//   * Basis(mesh, ElementTriP2())  DOF numbering, doflocs          (reference solver_fem.py:126)
//   * basis.get_dofs().all()       Dirichlet set                   (reference solver_fem.py:179)
//   * coo_matrix(...).tocsr()      sparsity pattern of asm()       (reference solver_fem.py:153-156)
//   * splu's ordering + symbolic factorisation of (A - sigma B)    (scipy arpack.py:915, via
//     reference solver_fem.py:197) -- here: element-based geometric nested dissection producing a
//     complete binary tree of dense frontal matrices.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "plan.h"

namespace plfem {

using rawvec_i32 = std::vector<int32_t, default_init_allocator<int32_t>>;

struct Symbolic {
  // ---- mesh / P2 numbering (scikit-fem compatible) ------------------------------------------
  int nv = 0, ne = 0, nedges = 0, N = 0, nsolve = 0;
  int dpn = 2;                     // unknowns per P2 node: 2 = vectorial H-field (Hx, Hy), 1 = scalar Helmholtz
  bool dirichlet = true;           // eliminate the outer-boundary nodes (H = 0); false = natural boundary, all nodes kept
  std::vector<int32_t> tsorted;    // [3][ne]  vertex ids, each column ascending
  std::vector<int32_t> edof;       // [6][ne]  element_dofs: rows 0-2 vertices, 3-5 edges (0,1),(1,2),(0,2)
  std::vector<int32_t> edges;      // [2][nedges] sorted vertex pairs in lexicographic order
  std::vector<double> doflocs;     // [2][N]
  std::vector<uint8_t> bmask;      // [N] 1 = on the outer boundary (Dirichlet H = 0)
  std::vector<int32_t> interior;   // [nsolve] ascending DOF ids
  std::vector<int32_t> int_index;  // [N] DOF -> interior index or -1
  // ---- scalar CSR pattern (full N x N, shared by every block of A and B) ----------------------
  // rowptr comes from closed-form row lengths; the column lists are built on the DEVICE (k_pattern_fill) for
  // the hot path and on the host only on demand (ensure_pattern: plfem_symbolic_get("colind" / "slot_row")).
  std::vector<int32_t> rowptr;     // [N+1]
  mutable rawvec_i32 colind;       // [nnz]  (lazy on the host)
  mutable rawvec_i32 slot_row;     // [nnz] row of every CSR slot (lazy on the host)
  // node -> adjacent elements (ascending element ids): the contributions to row i come from these
  std::vector<int32_t> nptr;       // [N+1]
  std::vector<int32_t> nadj;       // [6 ne] element id
  std::vector<uint8_t> nloc;       // [6 ne] local index of the node in that element
  // ---- nested-dissection front tree ----------------------------------------------------------
  int L = 0;                       // leaves at level L; fronts in heap order, nfronts = 2^(L+1)-1
  int nfronts = 0;
  std::vector<int32_t> leaf_of_elem;   // [ne]
  std::vector<int32_t> leaf_elem_ptr;  // [2^L + 1]
  std::vector<int32_t> leaf_elems;     // [ne] element ids grouped by leaf
  rawvec_i32 epos;                     // [ne][6] in the order of leaf_elems: local node index of node a of element leaf_elems[q] in its
                                       // leaf front at epos[6 q + a], -1 = Dirichlet
  mutable rawvec_i32 epos_by_elem;     // [6][ne] the same by element id (built on demand: plfem_symbolic_get("epos"))
  std::vector<int32_t> fs, fb;         // [nfronts] padded (to 16 / dpn nodes = 16 DOFs) counts of owned / boundary nodes
  std::vector<int32_t> fs_true, fb_true;
  std::vector<int64_t> fnode_ptr;      // [nfronts+1] offsets into fnodes/cinv*
  rawvec_i32 fnodes;                   // node (scalar DOF) id per local node, -1 = padding
  rawvec_i32 cinv0, cinv1;             // per local node of an internal front: index in child's boundary list or -1
  rawvec_i32 prow;                     // per local node: local node index in the PARENT front (boundary nodes), -1 = none
  std::vector<int32_t> npos;           // [N] node -> front-order offset of its component 0: 2 fnode_ptr[owner] + dpn q, -1 = Dirichlet
  // Storage of a front (m = dpn (fs + fb) DOFs, s2 = dpn fs owned, b2 = m - s2 boundary), what the factorisation KEEPS:
  //   foff[f]           : [F11; F21], m x s2 column major (leading dimension m); after the factorisation lower(F11) = L11^-1,
  //                       upper(F11) = L11^-T, F21 = Z
  //   foff[f] + m s2    : Z^T, s2 x b2 column major (leading dimension s2)
  // and what it only needs until the parent has gathered it, the Schur complement F22 (b2 x b2, leading dimension b2):
  //   soff[f]           : offset inside the arena of the front's tree level; two arenas (even / odd levels) of
  //                       arena_doubles each are alive at a time
  std::vector<int64_t> foff;           // [nfronts+1]
  std::vector<int64_t> soff;           // [nfronts]
  int64_t arena_doubles = 0;           // max over the levels of sum b2^2
  std::vector<int32_t> owner;          // [N] front that eliminates the node, -1 for Dirichlet nodes
  // statistics
  double factor_flops = 0.0;           // sum over fronts of 2 * s2 * m^2  (block Gauss-Jordan sweep)
  int64_t solve_entries = 0;           // sum over fronts of s2 * (m + (m - s2)) matrix entries read per solve
  int max_m = 0;
  double t_numbering = 0, t_pattern = 0, t_tree = 0, t_fronts = 0;  // seconds
  // ---- launch plan of the device kernels (plan.h): mesh-only, shared by every context on this analysis
  LaunchPlan plan;
};

// p: [2][nv] (x row then y row), t: [3][ne].  leaf_elems: target elements per leaf front.
// Returns empty string on success, error message otherwise.
std::string build_symbolic(int nv, int ne, const double* p, const int32_t* t, int leaf_elems,
                           int nthreads, Symbolic& S, int dofs_per_node = 2, bool dirichlet = true);

// Host copy of the CSR column lists (sorted union of the DOFs of the elements adjacent to each node); no-op
// if already built.  Not thread-safe against concurrent first calls on the same Symbolic.
void ensure_pattern(const Symbolic& S);

// f(rank, nthreads) on nthreads threads (the caller is rank 0) of a worker pool from the process-wide cache of idle pools
// the analysis uses -- for short host-side bursts outside the analysis (the staging copy of plfem_create), which would
// otherwise pay a thread creation per helper.
void host_parallel(int nthreads, const std::function<void(int, int)>& f);

// P2 numbering only (fills nv..int_index of S): what uniform red refinement needs, since the refined
// mesh's vertices are exactly the P2 nodes of the coarse mesh (new vertex id = nv + edge id).
std::string numbering_only(int nv, int ne, const double* p, const int32_t* t, Symbolic& S);

}  // namespace plfem
