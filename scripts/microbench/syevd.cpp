// How long does rocSOLVER take for the projected eigenproblem of the HOPE solver (symmetric fp64, n <= 512)?  (decision input for DESIGN 3.4)
//   hipcc -O2 -o scripts/microbench/syevd scripts/microbench/syevd.cpp -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <random>
int main()
{
    rocblas_handle h; rocblas_create_handle(&h);
    for (int n : {192, 320, 448, 512}) {
        std::vector<double> A((size_t)n * n);
        std::mt19937 g(1); std::normal_distribution<double> nd;
        for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = nd(g); A[(size_t)i * n + j] = v; A[(size_t)j * n + i] = v; }
        double *dA, *dD, *dE; rocblas_int *info;
        hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dD, sizeof(double) * n); hipMalloc(&dE, sizeof(double) * n); hipMalloc(&info, 4);
        for (int rep = 0; rep < 4; ++rep) {
            hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, n, dA, n, dD, dE, info);
            hipDeviceSynchronize();
            auto t1 = std::chrono::steady_clock::now();
            if (rep) printf("{\"n\": %d, \"rocsolver_dsyevd_ms\": %.3f}\n", n, std::chrono::duration<double, std::milli>(t1 - t0).count());
        }
        hipFree(dA); hipFree(dD); hipFree(dE); hipFree(info);
    }
    return 0;
}
