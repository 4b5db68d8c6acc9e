#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
extern "C" int gemhip_sym_eig_top(int32_t n, double *A, int32_t m, double *w, double *Z);
extern "C" int gemhip_sym_eig_builtin(int32_t n, double *A, double *w);
extern "C" int gemhip_set_host_threads(int32_t t, int32_t *eff);
int main()
{
    uint64_t s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)((s >> 11) & 0xFFFFFFFFFFFFFull) / 4503599627370496.0 - 0.5; };
    for (int kind = 0; kind < 3; ++kind)
        for (int n : {192, 257, 320}) {
            std::vector<double> A((size_t)n * n, 0.0);
            if (kind == 0) { for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) A[(size_t)i * n + j] = A[(size_t)j * n + i] = rnd(); }
            else if (kind == 1) { for (int i = 0; i < n; ++i) A[(size_t)i * n + i] = 1.0 + i; }
            else { int k = n / 2; for (int i = 0; i < k; ++i) for (int j = 0; j <= i; ++j) A[(size_t)i * n + j] = A[(size_t)j * n + i] = rnd(); }
            for (int T : {2, 3, 4}) {
                int32_t eff = 0; gemhip_set_host_threads(T, &eff);
                std::vector<double> B(A), w(40), Z((size_t)40 * n);
                int rc = gemhip_sym_eig_top(n, B.data(), 40, w.data(), Z.data());
                std::vector<double> C(A), wf(n);
                int rc2 = gemhip_sym_eig_builtin(n, C.data(), wf.data());
                printf("kind %d n %d T %d rc %d %d  top %.6f  full-top %.6f\n", kind, n, eff, rc, rc2, w[0], wf[n - 1]);
                if (std::fabs(w[0] - wf[n - 1]) > 1e-9 * (1 + std::fabs(w[0]))) return 1;
            }
        }
    return 0;
}
