#include "host_eig.h"

#include <cmath>
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <numeric>
#include <utility>

namespace plfem {

namespace {

// QL with implicit shifts on the tridiagonal (w, e) (e[i] couples i and i + 1 on entry at e[i + 1], shifted
// here); the rotations are applied to the rows of Z (n rows of np entries; np = 0: eigenvalues only)
bool ql_implicit(int n, double* w, double* e, double* Z, int np) {
  auto zrow = [&](int i) -> double* { return Z + (size_t)i * np; };
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  bool ok = true;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        double dd = std::fabs(w[m]) + std::fabs(w[m + 1]);
        if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
      }
      if (m != l) {
        if (iter++ == 120) { ok = false; break; }   // no convergence: leave the current approximation
        double gq = (w[l + 1] - w[l]) / (2.0 * e[l]);
        double r = std::hypot(gq, 1.0);
        gq = w[m] - w[l] + e[l] / (gq + (gq >= 0.0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, pp = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i], b = c * e[i];
          e[i + 1] = (r = std::sqrt(f * f + gq * gq));   // entries are O(|theta|), far from over/underflow: no hypot
          if (r == 0.0) {
            w[i + 1] -= pp;
            e[m] = 0.0;
            break;
          }
          s = f / r;
          c = gq / r;
          gq = w[i + 1] - pp;
          r = (w[i] - gq) * s + 2.0 * c * b;
          w[i + 1] = gq + (pp = s * r);
          gq = c * r - b;
          double* z0 = zrow(i);
          double* z1 = zrow(i + 1);
          for (int k = 0; k < np; ++k) {
            const double f1 = z1[k], f0 = z0[k];
            z1[k] = s * f0 + c * f1;
            z0[k] = c * f0 - s * f1;
          }
        }
        if (r == 0.0 && i >= l) continue;
        w[l] -= pp;
        e[l] = gq;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  return ok;
}

// last < 0: full eigenvectors into V (n x n); last = p >= 0: last p components only into V (n x p)
bool sym_eig_impl(int n, int last, std::vector<double>& A, std::vector<double>& V, std::vector<double>& w) {
  std::vector<double> e(n, 0.0), p(n, 0.0), g(n, 0.0), Yt;
  w.assign(n, 0.0);
  if (last < 0) V.swap(A);           // work in place; only the lower triangle (row i, columns <= i) is read
  double* Z = last < 0 ? V.data() : A.data();
  auto row = [&](int i) -> double* { return Z + (size_t)i * n; };
  // ---- Householder reduction to tridiagonal form; reflector i is kept in row i (columns 0 .. i-1)
  for (int i = n - 1; i > 0; --i) {
    const int l = i - 1;
    double* zi = row(i);
    double h = 0.0, scale = 0.0;
    if (l > 0) {
      for (int k = 0; k <= l; ++k) scale += std::fabs(zi[k]);
      if (scale == 0.0) {
        e[i] = zi[l];
      } else {
        for (int k = 0; k <= l; ++k) { zi[k] /= scale; h += zi[k] * zi[k]; }
        double f = zi[l];
        double gg = (f >= 0.0) ? -std::sqrt(h) : std::sqrt(h);
        e[i] = scale * gg;
        h -= f * gg;
        zi[l] = f - gg;
        // p = A u / h from the lower triangle only: a row dot product and an axpy per row
        for (int j = 0; j <= l; ++j) p[j] = 0.0;
        for (int j = 0; j <= l; ++j) {
          const double* zj = row(j);
          const double uj = zi[j];
          double s = 0.0;
          for (int k = 0; k < j; ++k) { s += zj[k] * zi[k]; p[k] += zj[k] * uj; }
          p[j] += s + zj[j] * uj;
        }
        f = 0.0;
        for (int j = 0; j <= l; ++j) {
          row(j)[i] = zi[j] / h;       // column i above the diagonal keeps u / h for the accumulation
          e[j] = p[j] / h;
          f += e[j] * zi[j];
        }
        const double hh = f / (h + h);
        for (int j = 0; j <= l; ++j) e[j] -= hh * zi[j];
        for (int j = 0; j <= l; ++j) {
          double* zj = row(j);
          const double fj = zi[j], gj = e[j];
          for (int k = 0; k <= j; ++k) zj[k] -= fj * e[k] + gj * zi[k];
        }
      }
    } else {
      e[i] = zi[l];
    }
    w[i] = h;
  }
  w[0] = 0.0;
  e[0] = 0.0;
  const int np = last < 0 ? n : last;          // components carried through the QL rotations
  if (last >= 0) {
    // rows n-p .. n-1 of Q = H_{n-1} ... H_1: y <- y H_i in that order, H_i = I - (u/h) u^T on [0, i)
    Yt.assign((size_t)n * np, 0.0);           // Yt[i*np + a] = Q[n-p+a][i]
    std::vector<double> y(n);
    for (int a = 0; a < np; ++a) {
      const int r = n - np + a;
      for (int k = 0; k < n; ++k) y[k] = 0.0;
      y[r] = 1.0;
      for (int i = n - 1; i > 0; --i) {
        if (w[i] == 0.0) continue;
        const double* zi = row(i);
        double dot = 0.0;
        for (int k = 0; k < i; ++k) dot += y[k] * row(k)[i];
        if (dot != 0.0)
          for (int k = 0; k < i; ++k) y[k] -= dot * zi[k];
      }
      for (int i = 0; i < n; ++i) Yt[(size_t)i * np + a] = y[i];
    }
    for (int i = 0; i < n; ++i) w[i] = row(i)[i];
  }
  // ---- accumulate the transformation (row-oriented: g[j] = sum_k u_k Q[k][j], Q[k][j] -= g[j] (u/h)_k)
  for (int i = 0; i < n && last < 0; ++i) {
    const int l = i - 1;
    double* zi = row(i);
    if (w[i] != 0.0) {
      for (int j = 0; j <= l; ++j) g[j] = 0.0;
      for (int k = 0; k <= l; ++k) {
        const double* zk = row(k);
        const double uk = zi[k];
        for (int j = 0; j <= l; ++j) g[j] += uk * zk[j];
      }
      for (int k = 0; k <= l; ++k) {
        double* zk = row(k);
        const double vk = zk[i];       // (u / h)_k
        for (int j = 0; j <= l; ++j) zk[j] -= g[j] * vk;
      }
    }
    w[i] = zi[i];
    zi[i] = 1.0;
    for (int j = 0; j <= l; ++j) { row(j)[i] = 0.0; zi[j] = 0.0; }
  }
  // ---- QL with implicit shifts on (w, e); the rotations act on ROWS of Z^T, so transpose once
  if (last < 0) {
    for (int i = 0; i < n; ++i)
      for (int k = i + 1; k < n; ++k) std::swap(Z[(size_t)i * n + k], Z[(size_t)k * n + i]);
  } else {
    Z = Yt.data();
  }
  const bool ok = ql_implicit(n, w.data(), e.data(), Z, np);
  if (last >= 0) V.swap(Yt);
  return ok;
}

}  // namespace

bool sym_eig(int n, std::vector<double>& A, std::vector<double>& V, std::vector<double>& w) {
  return sym_eig_impl(n, -1, A, V, w);
}

bool sym_eig_last_rows(int n, int p, std::vector<double>& A, std::vector<double>& Y, std::vector<double>& w) {
  if (p > n) p = n;
  return sym_eig_impl(n, p, A, Y, w);
}


// ------------------------------------------------------------------------------------------------
// Band path: the projected matrix of a block Lanczos run without restart is block tridiagonal (half bandwidth
// = block size), and only the few Ritz vectors of largest |theta| are ever used.  Eigenvalues: band -> tridiagonal
// by Givens rotations (annihilate the outer diagonals column by column, chase every bulge off the end; O(n^2 b),
// no transformation accumulated) + QL without vectors.  Selected eigenvectors: inverse iteration on the band
// matrix itself (band LU with partial pivoting), re-orthogonalised inside clusters as LAPACK's dstein does.
// ------------------------------------------------------------------------------------------------
bool sym_band_eigenvalues(int n, int b, const double* A, int lda, std::vector<double>& w) {
  std::vector<double> M((size_t)n * n, 0.0), e(n, 0.0);
  auto at = [&](int i, int j) -> double& { return M[(size_t)i * n + j]; };
  for (int i = 0; i < n; ++i)
    for (int j = i; j <= std::min(n - 1, i + b); ++j) at(i, j) = at(j, i) = A[(size_t)i * lda + j];
  // similarity rotation in the plane (p, p + 1), restricted to the index window [lo, hi] that holds their non-zeros
  auto rotate = [&](int p, double c, double s_, int lo, int hi) {
    const int q = p + 1;
    double* rp = &at(p, 0);
    double* rq = &at(q, 0);
    // the 2 x 2 diagonal block takes the rotation from both sides, the rest of rows p, q from the left only
    const double app = rp[p], apq = rp[q], aqq = rq[q];
    for (int k = lo; k <= hi; ++k) {
      const double x = rp[k], y = rq[k];
      rp[k] = c * x + s_ * y;
      rq[k] = c * y - s_ * x;
    }
    rp[p] = c * c * app + 2.0 * c * s_ * apq + s_ * s_ * aqq;
    rq[q] = s_ * s_ * app - 2.0 * c * s_ * apq + c * c * aqq;
    rp[q] = rq[p] = c * s_ * (aqq - app) + (c * c - s_ * s_) * apq;
    for (int k = lo; k <= hi; ++k) {                // mirror (the matrix is kept with both triangles)
      at(k, p) = rp[k];
      at(k, q) = rq[k];
    }
  };
  auto annihilate = [&](int r, int col) {           // zero (r, col) against (r - 1, col); returns false if already zero
    const double x = at(r - 1, col), y = at(r, col);
    if (y == 0.0) return false;
    const double h = std::sqrt(x * x + y * y);      // entries are O(|theta|): no scaling needed
    rotate(r - 1, x / h, y / h, col, std::min(n - 1, r + b));
    at(r, col) = 0.0;
    at(col, r) = 0.0;
    return true;
  };
  for (int j = 0; j + 2 < n; ++j)
    for (int d = std::min(b, n - 1 - j); d >= 2; --d) {
      int r = j + d;
      if (!annihilate(r, j)) continue;
      while (r + b < n) {                           // the rotation filled (r + b, r - 1): chase it
        const int col = r - 1;
        r += b;
        if (!annihilate(r, col)) break;
      }
    }
  w.resize(n);
  for (int i = 0; i < n; ++i) {
    w[i] = at(i, i);
    e[i] = i > 0 ? at(i, i - 1) : 0.0;
  }
  const bool ok = ql_implicit(n, w.data(), e.data(), nullptr, 0);
  std::sort(w.begin(), w.end());
  return ok;
}

bool sym_band_eigenvectors(int n, int b, const double* A, int lda, const std::vector<double>& w, const std::vector<int>& ids,
                           double* S, int lds) {
  const double eps = 2.220446049250313e-16;
  double anorm = 0.0;
  for (int i = 0; i < n; ++i) {
    double rs = 0.0;
    for (int j = std::max(0, i - b); j <= std::min(n - 1, i + b); ++j) rs += std::fabs(A[(size_t)std::min(i, j) * lda + std::max(i, j)]);
    anorm = std::max(anorm, rs);
  }
  if (anorm == 0.0) anorm = 1.0;
  const double sep = 10.0 * eps * anorm, ortol = 1e-3 * anorm, tiny = eps * anorm;
  std::vector<int> seq(ids);
  std::sort(seq.begin(), seq.end(), [&](int x, int y) { return w[x] < w[y] || (w[x] == w[y] && x < y); });
  std::vector<double> M((size_t)n * n, 0.0), x(n), y(n);
  std::vector<int> piv(n);
  auto at = [&](int i, int j) -> double& { return M[(size_t)i * n + j]; };
  bool ok = true;
  double lam_prev = 0.0, w_prev = 0.0;
  size_t cluster0 = 0;                                // first member (position in seq) of the current cluster
  uint64_t rng = 0x2545F4914F6CDD1Dull;
  for (size_t t = 0; t < seq.size(); ++t) {
    const int id = seq[t];
    double lam = w[id];
    if (t > 0) {
      if (w[id] - w_prev > ortol) cluster0 = t;
      if (lam - lam_prev < sep) lam = lam_prev + sep;   // keep the shifted systems of a cluster distinct
    }
    lam_prev = lam;
    w_prev = w[id];
    // band LU of A - lam I with partial pivoting (multipliers in place, not permuted; U has bandwidth 2 b)
    for (int i = 0; i < n; ++i) {
      for (int j = std::max(0, i - b); j <= std::min(n - 1, i + 2 * b); ++j) at(i, j) = 0.0;
      for (int j = std::max(0, i - b); j <= std::min(n - 1, i + b); ++j) at(i, j) = A[(size_t)std::min(i, j) * lda + std::max(i, j)];
      at(i, i) -= lam;
    }
    for (int c = 0; c < n; ++c) {
      const int rmax = std::min(n - 1, c + b), cmax = std::min(n - 1, c + 2 * b);
      int pr = c;
      for (int r = c + 1; r <= rmax; ++r)
        if (std::fabs(at(r, c)) > std::fabs(at(pr, c))) pr = r;
      piv[c] = pr;
      if (pr != c)
        for (int k = c; k <= cmax; ++k) std::swap(at(c, k), at(pr, k));
      if (std::fabs(at(c, c)) < tiny) at(c, c) = at(c, c) < 0.0 ? -tiny : tiny;
      for (int r = c + 1; r <= rmax; ++r) {
        const double l = at(r, c) / at(c, c);
        at(r, c) = l;
        if (l != 0.0)
          for (int k = c + 1; k <= cmax; ++k) at(r, k) -= l * at(c, k);
      }
    }
    for (int i = 0; i < n; ++i) {
      rng = rng * 6364136223846793005ull + 1442695040888963407ull;
      x[i] = ((double)(rng >> 11) / 9007199254740992.0) * 2.0 - 1.0;
    }
    const double want_growth = 1.0 / (100.0 * std::sqrt((double)n) * eps * anorm);
    int extra = -1;                                    // iterations left after the growth test passed
    for (int it = 0; it < 8 && extra != 0; ++it) {
      double nx = 0.0;
      for (int i = 0; i < n; ++i) nx += x[i] * x[i];
      nx = std::sqrt(nx);
      for (int i = 0; i < n; ++i) y[i] = x[i] / nx;
      for (int c = 0; c < n; ++c) {
        if (piv[c] != c) std::swap(y[c], y[piv[c]]);
        const double yc = y[c];
        for (int r = c + 1; r <= std::min(n - 1, c + b); ++r) y[r] -= at(r, c) * yc;
      }
      for (int c = n - 1; c >= 0; --c) {
        double v = y[c];
        for (int k = c + 1; k <= std::min(n - 1, c + 2 * b); ++k) v -= at(c, k) * y[k];
        y[c] = v / at(c, c);
      }
      double growth = 0.0;
      for (int i = 0; i < n; ++i) growth += y[i] * y[i];
      growth = std::sqrt(growth);
      for (size_t u = cluster0; u < t; ++u) {          // modified Gram-Schmidt against the cluster's earlier vectors
        const double* v = S + (size_t)seq[u] * lds;
        double dot = 0.0;
        for (int i = 0; i < n; ++i) dot += v[i] * y[i];
        for (int i = 0; i < n; ++i) y[i] -= dot * v[i];
      }
      x.swap(y);
      if (extra > 0) --extra;
      else if (extra < 0 && growth >= want_growth) extra = 2;
    }
    if (extra < 0) ok = false;                          // never reached the expected growth: vector kept as is
    double nx = 0.0;
    for (int i = 0; i < n; ++i) nx += x[i] * x[i];
    nx = 1.0 / std::sqrt(nx);
    double* out = S + (size_t)id * lds;
    for (int i = 0; i < n; ++i) out[i] = x[i] * nx;
  }
  return ok;
}

}  // namespace plfem
