// accuracy of v_rsq_f64 and of 1 / 2 Newton refinements (developer probe)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* d, double* y0, double* y1, double* y2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = d[i];
    double y = __builtin_amdgcn_rsq(x);
    y0[i] = y;
    const double h = 0.5 * x;
    y = y * __builtin_fma(-h * y, y, 1.5);
    y1[i] = y;
    y = y * __builtin_fma(-h * y, y, 1.5);
    y2[i] = y;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> h(n), a(n), b(n), c(n);
    for (int i = 0; i < n; ++i) h[i] = std::exp((i / (double)n) * 60.0 - 20.0) * (1.0 + 1e-3 * (i % 977));
    double *d, *y0, *y1, *y2;
    hipMalloc(&d, n * 8); hipMalloc(&y0, n * 8); hipMalloc(&y1, n * 8); hipMalloc(&y2, n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, y0, y1, y2, n);
    hipMemcpy(a.data(), y0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), y1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), y2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / sqrtl((long double)h[i]);
        e0 = fmax(e0, fabs((double)((a[i] - t) / t)));
        e1 = fmax(e1, fabs((double)((b[i] - t) / t)));
        e2 = fmax(e2, fabs((double)((c[i] - t) / t)));
    }
    printf("v_rsq_f64 max rel err: seed %.3e, 1 Newton %.3e, 2 Newton %.3e\n", e0, e1, e2);
    return 0;
}
