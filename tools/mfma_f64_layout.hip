// Probe of the v_mfma_f64_16x16x4f64 operand/result layout (run once on gfx950; result documented in kernels.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(double* out)
{
    const int l = threadIdx.x;
    // hypothesis: A[i][k]: lane holds i = l % 16, k = l / 16 ; B[k][j]: lane holds j = l % 16, k = l / 16
    const int i = l % 16, k = l / 16;
    const double a = 1.0 + i + 100.0 * k;   // A[i][k]
    const double b = (i == 3 ? 1.0 : 0.0) * (k == 2 ? 1.0 : 0.0);  // B[k][j] = delta(k,2) delta(j,3)  -> D[i][3] = A[i][2]
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main()
{
    double* d; hipMalloc(&d, 256 * sizeof(double));
    probe<<<1, 64>>>(d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // expected non-zeros: D[i][3] = A[i][2] = 201 + i
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (h[l * 4 + r] != 0.0) printf("lane %d reg %d = %g\n", l, r, h[l * 4 + r]);
    return 0;
}
