// Test harness for bifurcationkit.jl_amd/csrc/dense.h (host-only): reads "mode n" + n*n row-major doubles,
// prints eigenvalues and eigenvectors so tests/test_dense_host.py can compare with NumPy.
#include <cstdio>
#include <cstdlib>
#include "../../bifurcationkit.jl_amd/csrc/dense.h"
using namespace bk::dense;
int main() {
    int mode, n;
    if (scanf("%d %d", &mode, &n) != 2) return 2;
    Mat A(n, n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { double x; if (scanf("%lf", &x) != 1) return 2; A(i, j) = x; }
    if (mode == 0) {
        std::vector<double> w; Mat Z;
        int s = jacobi_eigh(A, w, Z);
        printf("%d\n", s);
        for (int i = 0; i < n; ++i) printf("%.17g\n", w[i]);
        for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) printf("%.17g ", Z(i, j)); printf("\n"); }
    } else {
        std::vector<cplx> w; CMat Y;
        int s = eig_general(A, w, Y);
        printf("%d\n", s);
        for (int i = 0; i < n; ++i) printf("%.17g %.17g\n", w[i].real(), w[i].imag());
        for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) printf("%.17g %.17g ", Y(i, j).real(), Y(i, j).imag()); printf("\n"); }
    }
    return 0;
}
