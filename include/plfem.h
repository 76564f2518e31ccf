/* plfem.h — C-ABI of libplfem_hip.so: the MI355X (gfx950) implementation of the vectorial H-field
 * P2 FEM eigenmode path of KhaoulaAguech/pl-fem-vectoriel.
 *
 * The reference has no native boundary: the whole path is Python calling scikit-fem and SciPy
 * (reference solver_fem.py:113-239).  Each entry point below names the reference statement(s) it
 * replaces; the Python host (pl_fem_vectoriel_amd/solver_fem.py) binds them with ctypes and keeps
 * the reference's class / method surface.  INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *  - every function returns 0 on success and a negative PLFEM_E* code on failure; the message is
 *    available from plfem_last_error(ctx) (or the err buffer for the context-free symbolic calls);
 *    nothing throws across the boundary;
 *  - all array arguments are caller-owned; "host" / "dev" in a parameter name says where the
 *    pointer must live.  The library never frees caller memory;
 *  - DOF order of all full-length vectors is the reference's block order (solver_fem.py:166):
 *    x[0:N] = Hx DOFs, x[N:2N] = Hy DOFs, N = number of P2 DOFs including boundary DOFs, whose
 *    entries are kept at zero (Dirichlet H = 0, solver_fem.py:179-182); a context of the scalar solver
 *    (plfem_symbolic_create_ex with one unknown per node) has vectors of length N and, with dirichlet = 0, no
 *    eliminated DOFs -- wherever a size below says 2N / 2 nsolve it is dofs_per_node x N / nsolve;
 *  - all floating point data is IEEE double; indices are int32 (nnz < 2^31), offsets int64;
 *  - a plfem_ctx owns one HIP stream's worth of state; contexts are independent: the only process-wide
 *    state is two mutex-protected recycling pools (pinned staging blocks, timing events), so different
 *    contexts may be used from different threads; one context must not be used from two threads at once.
 */
#ifndef PLFEM_H
#define PLFEM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLFEM_OK 0
#define PLFEM_EINVAL (-1)   /* bad argument / inconsistent sizes            -> ValueError   */
#define PLFEM_EMESH (-2)    /* malformed mesh                               -> ValueError   */
#define PLFEM_EHIP (-3)     /* HIP runtime error (message has the call)     -> RuntimeError */
#define PLFEM_ENOCONV (-4)  /* Lanczos did not converge within maxiter      -> scipy ArpackNoConvergence */
#define PLFEM_ESTATE (-5)   /* call order violated (e.g. solve before factor) -> RuntimeError */
#define PLFEM_ESINGULAR (-6)/* factorisation broke down (sigma is an eigenvalue) -> RuntimeError */
#define PLFEM_ERESIDUAL (-7)/* plfem_solve_modes: eigenpairs fail the a-posteriori check even after the refined pass -> RuntimeError */

typedef struct plfem_symbolic plfem_symbolic; /* host-only, mesh-only analysis               */
typedef struct plfem_ctx plfem_ctx;           /* device + stream + workspaces for one symbolic */

/* ---------------------------------------------------------------------------------------------
 * Symbolic phase (host, no GPU needed).
 * Replaces: Basis(mesh, ElementTriP2())            reference solver_fem.py:126
 *           basis.get_dofs().all() / setdiff1d     reference solver_fem.py:179-180
 *           sparsity work inside asm()/tocsr()     reference solver_fem.py:153-156
 *           splu ordering + symbolic factorisation scipy arpack.py:915 via solver_fem.py:197
 * p_host: [2][nv] doubles (x row, y row) = mesh.p; t_host: [3][ne] int32 = mesh.t.
 * leaf_elems: target triangles per leaf front of the nested-dissection tree (<=0: default).
 * ------------------------------------------------------------------------------------------- */
int plfem_symbolic_create(int32_t nv, int32_t ne, const double* p_host, const int32_t* t_host,
                          int32_t leaf_elems, int32_t nthreads, plfem_symbolic** out,
                          char* err, int32_t errlen);
/* The same analysis for the scalar solver of the reference (ScalarHelmholtzSolver.solve, solver_fem.py:245-276,
 * SURVEY.md row f3): dofs_per_node = 1 (one unknown per P2 node; 2 = the vectorial H-field path) and
 * dirichlet = 0 (natural boundary: every node is kept, solver_fem.py:259 passes the full matrices to eigsh). */
int plfem_symbolic_create_ex(int32_t nv, int32_t ne, const double* p_host, const int32_t* t_host,
                             int32_t leaf_elems, int32_t nthreads, int32_t dofs_per_node, int32_t dirichlet,
                             plfem_symbolic** out, char* err, int32_t errlen);
void plfem_symbolic_destroy(plfem_symbolic* sym);

/* info[] indices */
enum {
  PLFEM_INFO_NV = 0, PLFEM_INFO_NE, PLFEM_INFO_NEDGES, PLFEM_INFO_N, PLFEM_INFO_NSOLVE,
  PLFEM_INFO_NNZ, PLFEM_INFO_LEVELS, PLFEM_INFO_NFRONTS, PLFEM_INFO_FRONT_DOUBLES,
  PLFEM_INFO_MAX_FRONT, PLFEM_INFO_SOLVE_ENTRIES, PLFEM_INFO_FACTOR_FLOPS,
  PLFEM_INFO_T_NUMBERING_US, PLFEM_INFO_T_PATTERN_US, PLFEM_INFO_T_TREE_US, PLFEM_INFO_T_FRONTS_US,
  PLFEM_INFO_DOFS_PER_NODE, PLFEM_INFO_ARENA_DOUBLES, PLFEM_INFO_COUNT
};
int plfem_symbolic_info(const plfem_symbolic* sym, int64_t* info /* [PLFEM_INFO_COUNT] */);

/* Copy a named host array out of the analysis (for the Python compatibility surface and tests).
 * names: "edof"[6][ne] i32, "doflocs"[2][N] f64, "bmask"[N] u8, "interior"[nsolve] i32,
 * "rowptr"[N+1] i32, "colind"[nnz] i32, "slot_row"[nnz] i32, "nptr"[N+1] / "nadj"[6 ne] i32 / "nloc"[6 ne] u8
 * (node -> adjacent elements), "edges"[2][nedges] i32,
 * "leaf_of_elem"[ne] i32, "owner"[N] i32, "fs","fb"[nfronts] i32, "fnode_ptr","foff"[nfronts+1] i64, "soff"[nfronts] i64
 * (foff: kept part of a front = [F11; F21] m x s2 then Z^T s2 x b2; soff: its Schur complement inside the level's arena),
 * "fnodes","cinv0","cinv1"[fnode_ptr[nfronts]] i32, "epos"[6][ne] i32 (by element id), "epos_leaf"[ne][6] i32 (in the order of
 * "leaf_elems": what the device reads).
 * plfem_symbolic_array_bytes returns the size in bytes or a negative error. */
int64_t plfem_symbolic_array_bytes(const plfem_symbolic* sym, const char* name);
int plfem_symbolic_get(const plfem_symbolic* sym, const char* name, void* out_host, int64_t nbytes);

/* ---------------------------------------------------------------------------------------------
 * Mesh producer helper (the step before the path).
 * Replaces: MeshTri.refined() in MeshGenerator._generate_mesh   reference mesh.py:317-327
 * Uniform red refinement: new vertex id = nv + edge id (the P2 edge numbering), children
 * (t0,e0,e2), (t1,e0,e1), (t2,e2,e1), (e0,e1,e2), columns sorted.  p_out: [2][nv + nedges],
 * t_out: [3][4 ne]; nedges from plfem_mesh_edge_count.
 * ------------------------------------------------------------------------------------------- */
int plfem_mesh_edge_count(int32_t nv, int32_t ne, const double* p_host, const int32_t* t_host,
                          int32_t* nedges, char* err, int32_t errlen);
int plfem_mesh_refine(int32_t nv, int32_t ne, const double* p_host, const int32_t* t_host,
                      double* p_out, int32_t* t_out, char* err, int32_t errlen);

/* ---------------------------------------------------------------------------------------------
 * Context: binds a symbolic analysis to a device and stream, uploads the index structures and
 * allocates every workspace (nothing is allocated later, so calls are graph-capturable).
 * hip_stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = the default (null) stream.
 * max_ncv: largest Lanczos basis the context must hold, 3 <= max_ncv <= PLFEM_MAX_NCV.
 * Fails with PLFEM_EINVAL (message names the front order and the limit) when the largest front of the
 * analysis does not fit the solve sweeps' LDS staging even for one right-hand side (8 (m + 1) bytes of the
 * device's LDS per workgroup: ~20 000 DOFs on MI355X); between that and the limit for BLOCK_P = 4
 * right-hand sides (~5 000 DOFs) the eigen-solve silently uses the single-vector recurrence.
 * workspace_dev / workspace_bytes: optional caller-owned device memory (256-byte aligned, at least
 * plfem_workspace_bytes(sym, max_ncv) bytes, e.g. a torch tensor so that torch's caching allocator
 * recycles it between contexts) out of which EVERY device buffer of the context is carved; NULL / 0 =
 * the library hipMalloc's one slab itself and frees it in plfem_destroy.
 * plfem_create returns without synchronising: the one staged upload of the index structures (through a
 * pinned block from a small process-wide cache) and the pattern kernel are still in flight on the stream,
 * and every later call on the context is ordered behind them.  The symbolic handle must outlive the context.
 * ------------------------------------------------------------------------------------------- */
#define PLFEM_MAX_NCV 320
int plfem_workspace_bytes(const plfem_symbolic* sym, int32_t max_ncv, int64_t* bytes);
int plfem_create(const plfem_symbolic* sym, int32_t device, void* hip_stream, int32_t max_ncv,
                 void* workspace_dev, int64_t workspace_bytes, plfem_ctx** out, char* err, int32_t errlen);
void plfem_destroy(plfem_ctx* ctx);
const char* plfem_last_error(const plfem_ctx* ctx);
int plfem_synchronize(plfem_ctx* ctx);

/* ---------------------------------------------------------------------------------------------
 * Numeric assembly.
 * Replaces: the nine @BilinearForm closures + 9 x asm()   reference solver_fem.py:131-156
 *           geometry.epsilon at quadrature points          reference geometry_unified.py:325-336
 *           block build A_xx..A_yy, B                      reference solver_fem.py:158-167
 * cores_host: [ncore][3] = cx, cy, r (closed discs, later cores overwrite earlier ones — with two
 * permittivities that is a union).  eps_core / eps_clad are n_core^2, n_clad^2 (the PML factor has
 * no real part contribution: solver_fem.py:132 takes np.real).  alpha_p is the divergence penalty
 * (solver_fem.py:158).  Fills the device CSR value arrays of the blocks
 * Axx, Axy, Ayx, Ayy, Minv (= B_xx = B_yy), Dxx, Dxy, Dyy on the shared scalar pattern.
 * ------------------------------------------------------------------------------------------- */
int plfem_assemble_hfield(plfem_ctx* ctx, const double* cores_host, int32_t ncore, double eps_core,
                          double eps_clad, double k0, double alpha_p);

/* Scalar Helmholtz pencil of the reference's ScalarHelmholtzSolver.solve, for a context whose
 * analysis has one unknown per node:
 * Replaces: stiff / mass_s / eps_m forms + 3 x asm() and K - k0^2 Me   reference solver_fem.py:251-259
 * Fills the AXX slot with K - k0^2 M_eps (the "A" of plfem_spmv / plfem_factor / plfem_lanczos_shift_invert /
 * plfem_residuals), the MINV slot with the plain mass matrix M (their "B"); all vectors then have length N. */
int plfem_assemble_scalar(plfem_ctx* ctx, const double* cores_host, int32_t ncore, double eps_core,
                          double eps_clad, double k0);

/* Coupled-mode coupling integrals, scalar context only.
 * Replaces: the epsilon_product form + asm() and the E_i^H M_eps E_j loop of
 *           CoupledModeTheory._compute_rigorous_coupling                      reference config.py:296-320
 * M_deps = asm((Re eps - mean) u v), mean = plain mean of Re eps over ALL quadrature points (config.py:297-300; the
 * imaginary part is discarded by scikit-fem's float64 assembly); n real fields of length N each:
 * raw_host[i + j n] = E_i^T M_deps F_j with E = fields_i_dev[n][N], F = fields_j_dev[n][N]; pi/pj_host[i] = E_i.E_i, F_i.F_i.
 * The omega / 4 factor, the normalisation and the beta diagonal stay on the Python host as in the reference.
 * Overwrites the MINV slot: plfem_assemble_scalar must be called again before the next eigen-solve (PLFEM_ESTATE otherwise). */
int plfem_cmt_coupling(plfem_ctx* ctx, int32_t n, const double* fields_i_dev, const double* fields_j_dev,
                       const double* cores_host, int32_t ncore, double eps_core, double eps_clad,
                       double* raw_host, double* pi_host, double* pj_host, double* eps_mean_host);

enum { PLFEM_BLK_AXX = 0, PLFEM_BLK_AXY, PLFEM_BLK_AYX, PLFEM_BLK_AYY, PLFEM_BLK_MINV,
       PLFEM_BLK_DXX, PLFEM_BLK_DXY, PLFEM_BLK_DYY, PLFEM_BLK_COUNT };
/* device pointer of a block's CSR values (length nnz), valid until plfem_destroy */
int plfem_block_values_dev(plfem_ctx* ctx, int32_t block, const double** values_dev);
/* copy a block's CSR values to the host (synchronises the stream) */
int plfem_block_values_host(plfem_ctx* ctx, int32_t block, double* values_host);

/* ---------------------------------------------------------------------------------------------
 * CSR SpMV on the interior-restricted pencil, y = A_int x or y = B_int x embedded in 2N-vectors.
 * Replaces: B @ x inside ARPACK's reverse communication (scipy arpack.py:568-569) and the
 *           restriction A[idx,:][:,idx] (reference solver_fem.py:181-182) — boundary rows/cols masked.
 * which: 0 = A, 1 = B.
 * ------------------------------------------------------------------------------------------- */
int plfem_spmv(plfem_ctx* ctx, int32_t which, const double* x_dev, double* y_dev);

/* ---------------------------------------------------------------------------------------------
 * Shift-invert operator.
 * Replaces: splu((A - sigma B).tocsc())   scipy arpack.py:915 (via reference solver_fem.py:197)
 *           lu.solve(rhs)                 scipy arpack.py:920-928
 * Multifrontal block LDL^T factorisation (static pivoting: vanishing pivots are perturbed and counted,
 * plfem_timings()[5]) of the symmetric indefinite K = A_int - sigma B_int on the nested-dissection front
 * tree, all fronts dense in HBM, the unit-triangular pivot blocks inverted explicitly; solve = two sweeps
 * of batched dense panel products over the tree levels.  refine_steps extra iterative-refinement passes.
 * ------------------------------------------------------------------------------------------- */
int plfem_factor(plfem_ctx* ctx, double sigma);
int plfem_solve(plfem_ctx* ctx, const double* rhs_dev, double* x_dev, int32_t refine_steps);

/* ---------------------------------------------------------------------------------------------
 * Eigen-solve.
 * Replaces: eigsh(A_int, k, M=B_int, sigma=sigma, which='LM', tol, maxiter)
 *           reference solver_fem.py:196-197 -> scipy arpack.py:1359-1700 (mode 3, bmat='G').
 * Thick-restart Lanczos in the B inner product on OP = (A - sigma B)^-1 B with full (CGS2)
 * re-orthogonalisation; returns the k eigenvalues nearest sigma (largest |1/(lambda - sigma)|),
 * ascending, and B-orthonormal eigenvectors as 2N-vectors on the device.
 * Requires plfem_assemble_hfield + plfem_factor(sigma) before the call.
 * tol: a pair counts as converged when its Ritz residual ||r|| <= tol |theta|, theta = 1 / (lambda - sigma), the
 * criterion of ARPACK's dsaupd -- but tested after EVERY block step, and the iteration stops at the first step where
 * all k pairs meet it, where ARPACK tests at its restarts and usually ends orders of magnitude below its tolerance:
 * pass 1e-8 where the reference passes 1e-7 (measured agreement of the fields with eigsh: 3e-9 at 1e-8, 1e-7 at 1e-7).
 * evals_host[k]; evecs_dev[k][2N] (row c = vector c); stats_host[8] (may be NULL):
 *   [0] converged pairs, [1] OP applications, [2] restarts, [3] max relative Ritz residual.
 * stats_host[4] = passes over the factors (block solves; 0 = single-vector recurrence).
 * Returns PLFEM_ENOCONV if fewer than k pairs converged after maxiter RESTARTS of the basis (ARPACK
 * counts implicit restarts too: maxiter -> iparam[2] = mxiter, scipy arpack.py:358; outputs still hold the current Ritz pairs,
 * like ArpackNoConvergence.eigenvalues).  With the option "refine_steps" > 0 (plfem_set_option) every
 * OP application is followed by that many iterative-refinement passes r = Bx - K y, y += K^-1 r.
 * ------------------------------------------------------------------------------------------- */
int plfem_lanczos_shift_invert(plfem_ctx* ctx, int32_t k, int32_t ncv, double tol, int32_t maxiter,
                               double sigma, double* evals_host, double* evecs_dev, double* stats_host);

/* ---------------------------------------------------------------------------------------------
 * Per-mode post-processing.
 * Replaces: the per-mode loop of reference solver_fem.py:200-225 and _polarization_from_interp
 *           (solver_fem.py:68-107): Euclidean normalisation, divergence energy with the interior
 *           Dxx/Dxy/Dyy, core-mask sums.
 * evecs_dev[k][2N] as returned by plfem_lanczos_shift_invert (normalised IN PLACE to
 * sum(vx^2)+sum(vy^2) = 1 as solver_fem.py:213).  cores_host as in plfem_assemble_hfield.
 * out_host[k][PLFEM_POST_COUNT]; frac_core_host = (#interior DOF nodes inside a core)/N_solve.
 * modes_int_dev (may be NULL): [k][2*nsolve] interior-only copies (vx then vy) in the reference's
 * 'Ex_dofs'/'Ey_dofs' layout.
 * ------------------------------------------------------------------------------------------- */
enum { PLFEM_POST_NORM = 0, PLFEM_POST_DIV_ENERGY, PLFEM_POST_CORE_X, PLFEM_POST_CORE_Y,
       PLFEM_POST_ALL_X, PLFEM_POST_ALL_Y, PLFEM_POST_COUNT };
int plfem_postprocess(plfem_ctx* ctx, int32_t k, double* evecs_dev, const double* cores_host,
                      int32_t ncore, double* out_host, double* frac_core_host, double* modes_int_dev);

/* ---------------------------------------------------------------------------------------------
 * A-posteriori check of eigenpairs against the ASSEMBLED pencil (independent of the factorisation):
 * out_host[i] = || A v_i - lambda_i B v_i ||_2 / || A v_i ||_2 for the k vectors evecs_dev[k][2N].
 * No reference counterpart (eigsh trusts SuperLU's pivoting); here the LDL^T pivoting is static, so the
 * Python host checks every solve with this and re-runs with refinement when the check fails.
 * ------------------------------------------------------------------------------------------- */
int plfem_residuals(plfem_ctx* ctx, int32_t k, const double* evals_host, const double* evecs_dev, double* out_host);

/* ---------------------------------------------------------------------------------------------
 * The whole numeric solve in ONE call.
 * Replaces: everything TrueVectorialMaxwellSolver.solve_vectorial_modes does between the mesh analysis and its mode
 *           list -- assemble_hfield_system, the Dirichlet restriction, eigsh(..., sigma=...), the per-mode loop
 *           (reference solver_fem.py:176-225); on a scalar context ScalarHelmholtzSolver.solve (solver_fem.py:251-271).
 * = plfem_assemble_hfield (plfem_assemble_scalar on a context with one unknown per node) + plfem_factor(sigma) +
 *   plfem_lanczos_shift_invert(k, ncv, tol, maxiter) + plfem_postprocess + plfem_residuals, enqueued back to back on the
 *   context's stream, with the policy of the a-posteriori guard inside: if the largest residual exceeds residual_tol, or a
 *   vanishing pivot was perturbed, the eigen-solve is repeated with one more refinement pass inside the operator and the
 *   Ritz tolerance min(tol, tol_refined); if the check fails again: PLFEM_ERESIDUAL (message: plfem_last_error).
 * Behind the Lanczos iteration (which waits on its own step events) the host waits for the device ONCE; the copy of the
 * mode vectors to the host runs on a side stream beside the residual check.  Six C-ABI calls with four synchronisations
 * and the host work between them become one call: 0.3-0.5 ms of a 23-ms solve at C1.
 * Outputs (all caller-owned): evals_host[k] ascending; post_host[k][PLFEM_POST_COUNT] and *frac_core_host as
 * plfem_postprocess; resid_host[k] as plfem_residuals; modes_int_host (may be NULL): [k][dofs_per_node nsolve] interior
 * parts of the normalised vectors in the reference's 'Ex_dofs' / 'Ey_dofs' layout -- HOST memory, pinned for an
 * asynchronous copy (pageable memory works, synchronously); stats_host[PLFEM_SOLVE_STATS] (may be NULL), see the enum.
 * The full-length normalised vectors stay on the device in the context's own workspace: plfem_modes_dev.
 * PLFEM_ENOCONV: as plfem_lanczos_shift_invert (evals_host and plfem_modes_dev hold the current Ritz pairs).
 * ------------------------------------------------------------------------------------------- */
enum { PLFEM_SOLVE_NCONV = 0, PLFEM_SOLVE_NOPINV, PLFEM_SOLVE_RESTARTS, PLFEM_SOLVE_MAX_REL_RES, PLFEM_SOLVE_BLOCK_SOLVES,
       PLFEM_SOLVE_RESIDUAL_FIRST, PLFEM_SOLVE_RESIDUAL, PLFEM_SOLVE_REFINED, PLFEM_SOLVE_PERTURBED,
       PLFEM_SOLVE_T_ASSEMBLE_US, PLFEM_SOLVE_T_FACTOR_US, PLFEM_SOLVE_T_LANCZOS_US, PLFEM_SOLVE_T_POST_US,
       PLFEM_SOLVE_T_UPLOAD_US, PLFEM_SOLVE_T_RESIDUAL_US, PLFEM_SOLVE_T_CALL_US, PLFEM_SOLVE_STATS };
int plfem_solve_modes(plfem_ctx* ctx, const double* cores_host, int32_t ncore, double eps_core, double eps_clad,
                      double k0, double alpha_p, double sigma, int32_t k, int32_t ncv, double tol, int32_t maxiter,
                      double residual_tol, double tol_refined, double* evals_host, double* post_host,
                      double* frac_core_host, double* resid_host, double* modes_int_host, double* stats_host);
/* device pointer of the k full-length vectors ([k][dofs_per_node N], row c = vector c) the last plfem_solve_modes (or
 * plfem_lanczos_shift_invert) of the context produced; valid until the next plfem_factor / eigen-solve on the context */
int plfem_modes_dev(plfem_ctx* ctx, const double** evecs_dev, int32_t* k);

/* Options by name: "refine_steps" (iterative-refinement passes inside every OP application of
 * plfem_lanczos_shift_invert, default 0).  PLFEM_EINVAL for unknown names.  (No option alters a result in any other
 * way: the fault-injection hook of the test-suite is not in this library, see PLFEM_TEST_HOOKS below.) */
int plfem_set_option(plfem_ctx* ctx, const char* name, double value);

/* timings of the last calls in microseconds (HIP events on the context's stream):
 * [0] assemble, [1] factor, [2] lanczos, [3] postprocess, [4] upload; plus counters
 * [5] pivot perturbations in the last factorisation; [6] residual check of the last plfem_solve_modes. */
int plfem_timings(plfem_ctx* ctx, double* out_host /* [8] */);

/* ---------------------------------------------------------------------------------------------
 * Live timing for bench.py's "roofline" object (no reference counterpart): between begin and end
 * the ranges below are bracketed by HIP events on the context's stream.
 * out_host[PLFEM_PROF_COUNT][3] = { ranges timed, total microseconds, total algorithmic bytes } per slot:
 *   KFWD       every launch of the tile-form forward-sweep kernel (k_fwd, the largest single consumer
 *              of GPU time in a solve); algorithmic bytes of one launch = 8 B x sum over the level's fronts
 *              of (s2 m - s2^2/2 + P (m + s2)): the entries of [L11^-1 ; Z] read once + P staged / written vectors
 *   FWD_SWEEP  one whole forward sweep (all levels), same formula summed over all fronts
 *   BWD_SWEEP  one whole backward sweep
 *   SPMV_B     the block product B X of a Lanczos step: 12 B x nnz (Minv values + column indices of the
 *              shared pattern) + 4 B x (N + 1) row pointers + 2 x 8 B x P x 2N vector entries (read, written)
 * ------------------------------------------------------------------------------------------- */
enum { PLFEM_PROF_KFWD = 0, PLFEM_PROF_FWD_SWEEP, PLFEM_PROF_BWD_SWEEP, PLFEM_PROF_SPMV_B, PLFEM_PROF_COUNT };
int plfem_profile_begin(plfem_ctx* ctx, int32_t max_ranges);
int plfem_profile_end(plfem_ctx* ctx, double* out_host /* [PLFEM_PROF_COUNT][3] */);

#ifdef PLFEM_TEST_HOOKS
/* ---------------------------------------------------------------------------------------------
 * TEST HOOKS -- NOT exported by libplfem_hip.so.  They live in the add-on libplfem_testhooks.so (csrc/api_debug.hip,
 * built next to the product library and linked against it), which only tests/ and scripts/ load; define
 * PLFEM_TEST_HOOKS before including this header to see the declarations.  They take contexts created by the product
 * library and run the product library's own kernels.
 * Debugging aids for the test-suite (no reference counterpart): run the factorisation only up to
 * a given (tree level, block step, stage: 0 assembled, 1 or 2 pivot block + panel of the step done (its own launch for
 * step 0 of a level, the launch of the step before otherwise), 3 or 4 the step's launch done: trailing update + inverse
 * row + pivot write-back, next pivot block + panel; 5 level done), and copy a slice of a named device workspace
 * ("front","schur","fvec","fvec2","xl","wbuf","rbuf","dinv","delta","elem"; "colind","slot_row": the device-built CSR
 * index arrays, converted to double).  "schur", "wbuf" and "rbuf" -- scratch of the factorisation -- share their part of the
 * workspace with the Lanczos bases: copy them before the next plfem_lanczos_shift_invert / plfem_solve of the context.
 * plfem_debug_symeig: the host eigensolver of the Lanczos drivers (projected matrices of order
 * <= ~200; needs no GPU).  a_host: n x n symmetric.  last_rows < 0: v_out[i*n + k] = component k of
 * eigenvector i; last_rows = p >= 0: v_out[i*p + a] = component n-p+a of eigenvector i only (the
 * cheap form used by the per-step convergence test).  w_out[n]: eigenvalues, same (arbitrary) order.
 * plfem_debug_symeig_band: the band path used for the final Ritz vectors of a run without restart
 * (half bandwidth b; entries further from the diagonal are ignored): w_out[n] ascending, v_out[i*n + k] =
 * component k of the eigenvector of w_out[i] for the nsel eigenvalues of largest magnitude, zero rows elsewhere.
 * ------------------------------------------------------------------------------------------- */
int plfem_debug_factor_until(plfem_ctx* ctx, double sigma, int32_t level, int32_t step, int32_t stage);
int plfem_debug_copy(plfem_ctx* ctx, const char* name, int64_t offset, int64_t count, double* out_host);
/* timing aid of scripts/: reps block solves of a zero right-hand side; filter != 0 leaves out a class of fronts (results
 * are then wrong, only the kernel times mean something): 1 = skip fronts with more than 128 owned DOFs, 2 = only those */
int plfem_debug_solve_block(plfem_ctx* ctx, int32_t reps, int32_t filter);
int plfem_debug_symeig(int32_t n, const double* a_host, int32_t last_rows, double* w_out, double* v_out);
int plfem_debug_symeig_band(int32_t n, int32_t b, const double* a_host, int32_t nsel, double* w_out, double* v_out);
/* fault injection for the a-posteriori guard: from the next plfem_factor on, D^-1 of the root front is scaled by
 * 1 + value after every factorisation (0 = off) */
int plfem_debug_set_perturb(plfem_ctx* ctx, double value);
#endif /* PLFEM_TEST_HOOKS */

#ifdef __cplusplus
}
#endif
#endif /* PLFEM_H */
