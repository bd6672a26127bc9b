// hwid_probe.hip -- where do the waves of a 1024 x 256-thread launch with 39 KB of LDS per workgroup land?  (development aid)
// Prints, per CU, the workgroups resident on it and the SIMD of each of their 4 waves.
//   hipcc --offload-arch=gfx950 -O2 tools/hwid_probe.hip -o tools/hwid_probe && tools/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>
#include <algorithm>

__global__ __launch_bounds__(256) void probe(unsigned* out, long long spin)
{
    extern __shared__ double sm[];
    sm[threadIdx.x] = threadIdx.x;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = clock64();
    while (clock64() - t0 < spin) { __builtin_amdgcn_s_sleep(8); }
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2]     = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (sm[(threadIdx.x + 1) & 255] < 0) out[0] = 0;
}

int main()
{
    const int B = 1024;
    unsigned* d;
    hipMalloc(&d, B * 4 * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(B), dim3(256), 39800, 0, d, 200000LL);
    hipDeviceSynchronize();
    std::vector<unsigned> h(B * 8);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
    std::map<std::tuple<int, int, int, int>, std::vector<std::pair<int, std::vector<int>>>> cus;
    for (int b = 0; b < B; ++b) {
        std::vector<int> simds;
        unsigned hw0 = h[(b * 4) * 2], x0 = h[(b * 4) * 2 + 1] & 0xF;
        for (int w = 0; w < 4; ++w) simds.push_back((h[(b * 4 + w) * 2] >> 4) & 3);
        cus[{(int)x0, (int)((hw0 >> 13) & 7), (int)((hw0 >> 12) & 1), (int)((hw0 >> 8) & 15)}].push_back({b, simds});
    }
    printf("CUs used: %zu\n", cus.size());
    int shown = 0;
    std::map<int, int> per_cu_hist, simd0_hist;
    for (auto& kv : cus) {
        per_cu_hist[(int)kv.second.size()]++;
        int s0[4] = {0, 0, 0, 0};
        for (auto& wg : kv.second) s0[wg.second[0]]++;
        simd0_hist[*std::max_element(s0, s0 + 4)]++;
        if (shown++ < 12) {
            printf("xcc %d se %d sh %d cu %2d:", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first));
            for (auto& wg : kv.second) printf("  wg %4d simds %d%d%d%d", wg.first, wg.second[0], wg.second[1], wg.second[2], wg.second[3]);
            printf("\n");
        }
    }
    for (auto& kv : per_cu_hist) printf("CUs with %d workgroups: %d\n", kv.first, kv.second);
    for (auto& kv : simd0_hist) printf("CUs whose most loaded SIMD hosts wave 0 of %d workgroups: %d\n", kv.first, kv.second);
    return 0;
}
