// Small dense symmetric eigensolver for the projected matrices of the Lanczos drivers (order <= ~200).
// Host-only C++ (no HIP): compiled into the library and into the CPU unit test harness.
#pragma once
#include <vector>

namespace plfem {

// Dense symmetric eigen-decomposition: Householder tridiagonalisation + implicit-shift QL (the
// classical tred2 / tql2 pair) with every inner loop running along contiguous memory.
// A: n x n symmetric (either storage order; destroyed), V: eigenvector i in V[i*n .. i*n+n), w: eigenvalues
// (unordered).  Returns false if the QL iteration did not converge for some eigenvalue.
bool sym_eig(int n, std::vector<double>& A, std::vector<double>& V, std::vector<double>& w);

// Same eigenvalues, but only the LAST p components of every eigenvector: Y[i*p + a] = component n-p+a of
// eigenvector i.  That is all a Lanczos convergence test needs (residual of Ritz pair i = || R_m s_i[last block] ||)
// and it skips both cubic-cost stages that carry full eigenvectors: O(2/3 n^3) instead of O(~5 n^3).
bool sym_eig_last_rows(int n, int p, std::vector<double>& A, std::vector<double>& Y, std::vector<double>& w);

// Band path (block tridiagonal projected matrices, only a few eigenvectors wanted).  A: n x n, upper triangle read
// (A[i*lda + j], j >= i), entries more than b off the diagonal are ignored.
// sym_band_eigenvalues: all eigenvalues, ascending.  sym_band_eigenvectors: unit eigenvectors of w[id] for the
// listed ids, written to S[id*lds .. id*lds + n) (rows of other ids untouched); orthonormal to rounding, also
// inside clusters.  Both return false if an iteration did not converge as expected (result still usable).
bool sym_band_eigenvalues(int n, int b, const double* A, int lda, std::vector<double>& w);
bool sym_band_eigenvectors(int n, int b, const double* A, int lda, const std::vector<double>& w, const std::vector<int>& ids,
                           double* S, int lds);

}  // namespace plfem
