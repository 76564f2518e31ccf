#include "host_eig.h"

#include <cmath>
#include <cstddef>
#include <utility>

namespace plfem {

namespace {

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
          e[i + 1] = (r = std::hypot(f, gq));
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

}  // namespace plfem
