// development probe: per-instruction latency of dependent fp64 chains on one wave (clock64 deltas / chain length).   hipcc --offload-arch=gfx950 -O3 lat_probe.hip -o lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
#define T(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 0" :: "v"(x), "v"(idx), "v"(f)); t = clock64(); asm volatile("s_nop 0" : "+v"(x), "+v"(idx), "+v"(f)); __builtin_amdgcn_sched_barrier(0); } while (0)
__global__ void probe(double* out, long long* t, double a, double b)
{
    double x = a + threadIdx.x * 1e-3, y = b, z = a * 0.5, w = b * 0.25;
    __shared__ double lds[256];
    lds[threadIdx.x] = x;
    __syncthreads();
    int idx = threadIdx.x; float f = (float)a, g = (float)b;
    long long t0, t1, t2, t3, t4, t5, t6, t7, t8; T(t0);
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, b);              // dependent FMA chain
    T(t1);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) { x = __builtin_fma(x, y, b); z = __builtin_fma(z, y, b); w = __builtin_fma(w, y, b); y = __builtin_fma(y, a, b); }   // 4 independent chains
    T(t2);
#pragma unroll
    for (int i = 0; i < N / 8; ++i) { double r = __builtin_amdgcn_rsq(x); x = __builtin_fma(r, y, b); }   // rsq + fma dependent
    T(t3);
#pragma unroll
    for (int i = 0; i < N / 8; ++i) { double v = lds[idx & 255]; idx = (int)v + idx; }   // dependent LDS loads
    T(t4);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {   // dependent DPP quad broadcast of a double
        int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x55, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x55, 0xF, 0xF, true);
        x = __hiloint2double(hi, lo) + b;
    }
    T(t5);
    f += (float)x;
#pragma unroll
    for (int i = 0; i < N; ++i) f = __builtin_fmaf(f, g, 1.0f);   // dependent fp32 FMA chain
    T(t6);
#pragma unroll
    for (int i = 0; i < N; ++i) x = x * y;   // dependent fp64 mul
    T(t7);
#pragma unroll
    for (int i = 0; i < N; ++i) x = x + y;   // dependent fp64 add
    T(t8);
    out[threadIdx.x + blockIdx.x * blockDim.x] = x + z + w + y + idx + f;
    if (threadIdx.x == 0) { long long* tt = t + blockIdx.x * 8; tt[0] = t1 - t0; tt[1] = t2 - t1; tt[2] = t3 - t2; tt[3] = t4 - t3; tt[4] = t5 - t4; tt[5] = t6 - t5; tt[6] = t7 - t6; tt[7] = t8 - t7; }
}
int main()
{
    double* out; long long* t;
    hipMalloc(&out, 1 << 20); hipMalloc(&t, 4096);
    for (int waves = 1; waves <= 2; ++waves) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64 * waves * 4), 0, 0, out, t, 1.0000001, 0.9999999);   // waves per SIMD = waves (4 SIMDs)
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        printf("waves/SIMD %d: dep fma64 %.1f | 4-chain fma64 %.1f per instr | rsq+fma pair %.1f | dep LDS load %.1f | dpp-bcast+add %.1f | dep fma32 %.1f | dep mul64 %.1f | dep add64 %.1f  (cycles)\n", waves,
               h[0] / 256.0, h[1] / 256.0, h[2] / 32.0, h[3] / 32.0, h[4] / 64.0, h[5] / 256.0, h[6] / 256.0, h[7] / 256.0);
    }
    return 0;
}
