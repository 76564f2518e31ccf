// Device-side context of libplfem_hip.so (gfx950).  Internal header.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/plfem.h"
#include "internal.h"
#include "plan.h"

struct plfem_ctx {
  const plfem::Symbolic* S = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  // sizes
  int nv = 0, ne = 0, N = 0, nnz = 0, nsolve = 0, L = 0, nfronts = 0, max_ncv = 0;
  int dpn = 2, sh = 1;            // unknowns per node (2: Hx, Hy; 1: scalar Helmholtz) and sh = dpn - 1: node = dof >> sh, component = dof & sh
  int64_t fnodes_total = 0;       // sum over fronts of (padded) nodes = fnode_ptr[nfronts]
  int64_t level_nodes_max = 0;    // the same sum over the fronts of one tree level, largest level
  int64_t n2 = 0;   // dpn N: length of every global vector (component-major blocks of N)
  std::vector<plfem::LevelInfo> levels;
  // ---- index structures on the device
  int32_t* d_forder = nullptr;    // [nfronts] per level: front ids in order of decreasing s2 (factorisation launches)
  plfem::FrontRec* d_frec = nullptr;   // the same order, with the front's parameters
  std::vector<int> forder_s2, forder_maxm;   // host: s2 in that order, running max of m in that order
  int2* d_tiles = nullptr;        // (front, tx | ty << 16) of every useful 64 x 64 workgroup of the factorisation
  std::vector<int64_t> upd_off;   // per (level, block step): first entry / entries of the trailing-update list
  std::vector<int> upd_n;
  int64_t formz_all_off = 0;      // d_tiles: the Z blocks of every front in one list (root first)
  int formz_all_n = 0;
  int64_t mirrorx_all_off = 0;    // d_tiles: (front, block row >= 1) of every front, for k_mirror_x
  int mirrorx_all_n = 0;
  plfem::SweepJob* d_blk = nullptr;   // one entry per sweep workgroup, level by level
  int32_t *d_tsorted = nullptr, *d_edof = nullptr, *d_rowptr = nullptr, *d_colind = nullptr;
  int32_t *d_slot_row = nullptr, *d_nptr = nullptr, *d_nadj = nullptr, *d_interior = nullptr;
  uint8_t* d_nloc = nullptr;
  uint8_t* d_bmask = nullptr;
  double* d_doflocs = nullptr;
  int32_t *d_fs2 = nullptr, *d_fm = nullptr;          // per front: owned DOFs (2 fs), front order m = 2 (fs + fb)
  int64_t *d_fnode_ptr = nullptr, *d_foff = nullptr;
  int32_t *d_fnodes = nullptr, *d_cinv0 = nullptr, *d_cinv1 = nullptr;
  int32_t *d_epos = nullptr, *d_leaf_elem_ptr = nullptr, *d_leaf_elems = nullptr;
  // ---- numeric data
  double* d_cores = nullptr;      // [64][3]
  double* d_elem = nullptr;       // [ne][8][36]
  double* d_vals[PLFEM_BLK_COUNT] = {nullptr};
  double* d_front = nullptr;      // what a front keeps: [F11; F21] (m x s2) and Z^T (s2 x b2), see symbolic.h
  double* d_schur = nullptr;      // two arenas (even / odd tree levels) of arena_doubles for the Schur complements in flight
  int64_t arena_doubles = 0;
  int64_t* d_soff = nullptr;      // per front: offset of its Schur complement inside its level's arena
  double* d_fvec = nullptr;       // per-front solve vectors in front order, offset 2*fnode_ptr[f]: right-hand side (owned rows)
  double *d_u0 = nullptr, *d_u1 = nullptr;   // updates pushed into a front's rows by its left / right child (forward sweep)
  double* d_xl = nullptr;         // complete local solution of every front (backward sweep)
  int32_t* d_npos = nullptr;      // [N] node -> front-order offset of its component 0: 2 fnode_ptr[owner] + dpn * local index, -1 = Dirichlet
  int32_t* d_prow = nullptr;      // per local node of a front: local node index in the PARENT front, -1 = none / padding
  double *d_wbuf = nullptr, *d_rbuf = nullptr;   // per-front panels m x NB of the level in flight, offset 2*(fnode_ptr[f] - fnode_ptr[level first])*NB;
                                                 // three thirds (block steps mod 3).  d_schur, d_wbuf, d_rbuf (alive during a factorisation
                                                 // only) share their part of the workspace with d_V, d_BV, d_V2, d_BV2 (alive during a Lanczos run only)
  double* d_dinv = nullptr;       // 2 x per-front NB x NB (inverse of the unit-lower pivot block of even / odd block steps)
  double* d_delta = nullptr;      // per-front D^-1 of the block LDL^T: (diagonal, off-diagonal of the node pair) per row, offset 2 * (2*fnode_ptr[f])
  double* d_fvec2 = nullptr;      // forward-sweep results of the owned rows (t = L11^-1 r; the backward sweep applies D^-1), front order
  int32_t* d_counters = nullptr;  // [0] pivot perturbations
  // ---- Lanczos workspace
  double *d_V = nullptr, *d_BV = nullptr, *d_V2 = nullptr, *d_BV2 = nullptr;   // n2 x (max_ncv+1), column major
  double *d_w = nullptr, *d_bw = nullptr, *d_t1 = nullptr, *d_t2 = nullptr;    // n2 (d_w, d_bw: n2 x BLOCK_P)
  double *d_hblk = nullptr, *d_G = nullptr, *d_Rinv = nullptr;                 // block Lanczos small matrices
  double *d_h = nullptr, *d_hacc = nullptr, *d_partial = nullptr, *d_scal = nullptr, *d_S = nullptr;
  double* d_Hcols = nullptr;      // (max_ncv+1) x (max_ncv+1) projected matrix columns
  uint8_t* d_coremask = nullptr;  // [N]
  double* d_post = nullptr;       // partial sums: post-processing in [0, post_doubles), residual check behind it
  size_t post_doubles = 0;
  int npartial = 0;
  double* h_pinned = nullptr;     // pinned staging
  size_t h_pinned_bytes = 0;      // size of that block (it returns to a process-wide cache)
  double* h_staging = nullptr;    // pinned staging block of the one upload of the host arrays (same cache)
  size_t h_staging_bytes = 0;
  double* h_slots = nullptr;      // pinned: new projected-matrix columns of the two block steps in flight
  char* slab = nullptr;           // the one device allocation every buffer above is carved from
  size_t slab_off = 0, slab_bytes = 0;
  bool own_slab = false;
  int64_t workspace_need = 0;
  // state
  bool assembled = false, factored = false;
  hipEvent_t ev_step[2] = {nullptr, nullptr};   // block Lanczos: completion of the two block steps in flight
  bool defer_sync = false;        // plfem_solve_modes: the Lanczos drivers leave their final stream synchronisation to it
  hipStream_t copy_stream = nullptr;   // plfem_solve_modes: side stream of the device-to-host copy of the mode vectors
  hipEvent_t ev_copy = nullptr;        // "mode vectors ready" (main stream -> copy stream)
  hipEvent_t ev_upload = nullptr;      // "front-level index arrays uploaded" (copy stream -> main stream)
  bool upload_pending = false;
  // live kernel timing (plfem_profile_*): event pairs around every tile-form forward-sweep launch
  bool prof_on = false;
  unsigned prof_toggle = 0;       // block solves alternate between timing whole sweeps and timing single launches
  int prof_n = 0, prof_max = 0;
  double prof_bytes = 0;
  std::vector<hipEvent_t> prof_ev;   // taken from the process-wide pool at profile_begin, handed back at profile_end
  std::vector<int> prof_slot;        // PLFEM_PROF_* of every timed range
  std::vector<double> prof_rbytes;   // algorithmic bytes of every timed range
  // options (plfem_set_option)
  int refine_steps = 0;           // iterative-refinement passes inside every OP application of the Lanczos drivers
  // Test hooks: nothing in libplfem_hip.so sets these (no option, no export).  The add-on libplfem_testhooks.so
  // (api_debug.hip, plfem_debug_*) installs them on a context the tests hand it.
  void (*test_post_factor)(plfem_ctx*) = nullptr;   // called at the end of every plfem_factor
  double test_perturb = 0.0;      // plfem_debug_set_perturb: relative perturbation of the root front's D^-1
  int debug_sweep_filter = 0;     // plfem_debug_solve_block: timing experiments (results are wrong when set)
  int max_block_p = plfem::BLOCK_P;   // right-hand sides per sweep the LDS budget allows (BLOCK_P or 1)
  int lds_limit = 0;              // bytes of LDS one workgroup may use on this device
  double sigma = 0.0, k0 = 0.0;
  hipEvent_t ev[6][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  bool ev_used[6] = {false, false, false, false, false, false};   // assemble, factor, lanczos, post, upload, residual check
  double* modes_dev = nullptr;    // where the last eigen-solve left its vectors (caller's buffer or the context's own)
  int modes_k = 0;
  double timings[8] = {0};
};

namespace plfem {

// live timing (plfem_profile_*): a timed range = two HIP events on the context's stream around one or more launches
inline int prof_open(plfem_ctx* c, int slot, double bytes) {
  if (!c->prof_on || c->prof_n >= c->prof_max) return -1;
  while ((int)c->prof_ev.size() < 2 * (c->prof_n + 1)) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return -1;
    c->prof_ev.push_back(e);
  }
  const int id = c->prof_n++;
  c->prof_slot.push_back(slot);
  c->prof_rbytes.push_back(bytes);
  (void)hipEventRecord(c->prof_ev[2 * id], c->stream);
  return id;
}
inline void prof_close(plfem_ctx* c, int id) {
  if (id >= 0) (void)hipEventRecord(c->prof_ev[2 * id + 1], c->stream);
}

// kernels_assembly.hip
void launch_element_matrices(plfem_ctx* c, int ncore, double eps_core, double eps_clad, double k0, double alpha_p);
void launch_element_matrices_scalar(plfem_ctx* c, int ncore, double eps_core, double eps_clad, double k0);
double launch_delta_eps_mass(plfem_ctx* c, int ncore, double eps_core, double eps_clad);   // MINV slot <- asm((eps - mean eps) u v)
void launch_csr_gather(plfem_ctx* c);
void launch_pattern_fill(plfem_ctx* c);   // colind / slot_row from the node -> element adjacency (once per context)
void launch_spmv(plfem_ctx* c, int which, const double* x, double* y);
void launch_spmv_b_block(plfem_ctx* c, const double* x, double* y, int64_t ld);   // y_q = B x_q, BLOCK_P vectors
// same, x as [node][component][q]; gram != nullptr: also the chunk partials of the Gram matrix x^T (B x), gram[(p P + q) nb + b]
// for workgroup b of nb (returned)
int launch_spmv_b_block_il(plfem_ctx* c, const double* x_interleaved, double* y, int64_t ld, double* gram = nullptr);
void launch_spmv_a_block(plfem_ctx* c, const double* x, double* y, int64_t ld);   // y_q = A x_q, BLOCK_P vectors
// out_host[i] = ||A v_i - lambda_i B v_i|| / ||A v_i||  (k vectors, row i of evecs; synchronises)
void launch_residuals(plfem_ctx* c, int k, const double* lam_host, const double* evecs, double* out_host);
// kernels_front.hip (factorisation), kernels_sweep.hip (solve sweeps)
void launch_factor(plfem_ctx* c, double sigma, int stop_level = -1, int stop_step = 0, int stop_stage = 0);
void launch_solve(plfem_ctx* c, const double* rhs, double* x);
// BLOCK_P right-hand sides; x == nullptr: the result stays in front order in d_xl (the caller permutes it itself)
void launch_solve_block(plfem_ctx* c, const double* rhs, double* x, int64_t ldx, bool rhs_in_front_order = false);
// kernels_lanczos.hip
void launch_panel_dot(plfem_ctx* c, const double* P, int ncols, const double* w, double* h);   // h = P^T w
void launch_panel_axpy(plfem_ctx* c, const double* P, int ncols, const double* h, double* w);  // w -= P h
void launch_dot(plfem_ctx* c, const double* a, const double* b, double* out);                  // *out = a.b
void launch_vec_add(plfem_ctx* c, double* acc, const double* h, int n);                        // acc += h
void launch_scale_store(plfem_ctx* c, const double* w, const double* bw, const double* beta2, double* v, double* bv,
                        double* beta_out);  // v = w/sqrt(beta2), bv = bw/sqrt(beta2)
void launch_axpby(plfem_ctx* c, double a, const double* x, double b, const double* y, double* z);  // z = a x + b y
void launch_axpby_n(plfem_ctx* c, int64_t n, double a, const double* x, double b, const double* y, double* z);
void launch_scale(plfem_ctx* c, int64_t n, double a, double* x);   // x *= a
void launch_rotate(plfem_ctx* c, const double* V, int m, const double* Smat, int ldS, int p, double* out);  // out = V[:, :m] S
// block (BLOCK_P vectors) variants; H matrices are column major with leading dimension ldh
// h = Pm^T W (ncols x BLOCK_P); hacc (optional) += the same coefficients
void launch_panel_dot_block(plfem_ctx* c, const double* Pm, int ncols, const double* W, int64_t ldw, double* h, int ldh,
                            double* hacc = nullptr, int ldacc = 0);
void launch_panel_axpy_block(plfem_ctx* c, const double* Pm, int ncols, const double* H, int ldh, double* W, int64_t ldw,
                             double* w_interleaved = nullptr);   // w_interleaved: see k_spmv_b_block_il
// first Gram-Schmidt pass of a block step over ncols <= 8 columns in two launches: reads the sweeps' result d_xl (front
// order), writes W in global order (what k_permute_out would have done), h = BVm^T W -> Hout, W -= Vm h
void launch_first_pass_block(plfem_ctx* c, const double* BVm, const double* Vm, int ncols, double* W, int64_t ldw, double* Hout, int ldh);
// the Cholesky half alone, from nchunks Gram partials per entry already in d_partial (launch_spmv_b_block_il with gram)
void launch_chol_from_partials(plfem_ctx* c, int nchunks, double* Tblk, int ldT, double* Rinv);
void launch_chol_block(plfem_ctx* c, const double* G, int ldg, double* Tblk, int ldT, double* Rinv);
void launch_block_scale(plfem_ctx* c, const double* W, const double* BW, int64_t ldw, const double* Rinv, double* Vn,
                        double* BVn, int64_t ldv, const double* exp_src = nullptr, int exp_n = 0, double* exp_dst = nullptr,
                        int32_t* cnt_dst = nullptr, double* bv_front = nullptr);
void launch_start_field(plfem_ctx* c, int nvec, double* out);
void launch_post(plfem_ctx* c, int k, double* evecs, int ncore, double* out_host, double* frac_core, double* modes_int);
// the same in two halves (plfem_solve_modes: one stream synchronisation for everything behind the Lanczos run)
void post_enqueue(plfem_ctx* c, int k, double* evecs, int ncore, double* modes_int, const std::function<void(int, int)>* group_done);
void post_finish(plfem_ctx* c, int k, double* out_host, double* frac_core);
void resid_enqueue(plfem_ctx* c, int k, const double* lam_host, const double* evecs);
void resid_finish(plfem_ctx* c, int k, double* out_host);

}  // namespace plfem
