// FETCH_SIZE / WRITE_SIZE calibration on known byte counts (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide
// coalesced streaming read (16 B per lane) ... other access widths and WRITE_SIZE are uncalibrated").  One launch per access pattern over a buffer four times the
// Infinity Cache, every launch moves exactly `bytes`:
//   read16 / read8 / read4     coalesced streaming reads, 16 / 8 / 4 bytes per lane
//   read_rec                   the chain kernels' pattern (big_chain_kernel & co.: one wave per instance walks stage records of 616 doubles and reads 12 x 12 blocks in
//                              row form): per record one contiguous run of 144 doubles (lanes l, l + 64, l + 128 < 144), records 616 doubles apart
//   read_row12                 the stage kernel's local Jacobian rows: 12 doubles per lane group of 12, rows 28 doubles apart (partial 128-byte lines)
//   write16 / write8           coalesced streaming writes
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calibration.hip -o tools/fetch_calibration; run under rocprofv3 --pmc FETCH_SIZE resp. WRITE_SIZE:
// tools/fetch_calibration.sh prints counter / known bytes per pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class T>
__global__ __launch_bounds__(256) void read_coalesced(const T* __restrict__ src, size_t n, double* sink)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = src[i];
        if constexpr (sizeof(T) == 16) acc += v.x + v.y;
        else acc += (double)v;
    }
    if (acc == 12345.678) sink[0] = acc;   // (never true for the zero-filled buffer: keeps the loads alive)
}

// one wave per record chain: wave w reads records w, w + W, ...; 144 of each record's 616 doubles
__global__ __launch_bounds__(64) void read_rec(const double* __restrict__ src, size_t records, double* sink)
{
    const int lane = threadIdx.x;
    double acc = 0.0;
    for (size_t r = blockIdx.x; r < records; r += gridDim.x) {
        const double* rec = src + r * 616;
        acc += rec[lane] + rec[lane + 64];
        if (lane < 16) acc += rec[lane + 128];
    }
    if (acc == 12345.678) sink[0] = acc;
}

// rows of 12 doubles, 28 doubles apart: lane l of a wave reads row (l / 12), column (l % 12) (60 of 64 lanes: five rows per wave instruction)
__global__ __launch_bounds__(256) void read_row12(const double* __restrict__ src, size_t rows, double* sink)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t waves = (size_t)gridDim.x * blockDim.x / 64;
    double acc = 0.0;
    if (lane < 60)
        for (size_t r0 = (size_t)wave * 5; r0 + 5 <= rows; r0 += waves * 5) acc += src[(r0 + lane / 12) * 28 + lane % 12];
    if (acc == 12345.678) sink[0] = acc;
}

template <class T>
__global__ __launch_bounds__(256) void write_coalesced(T* __restrict__ dst, size_t n, T v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;   // 1 GiB per launch: four times the 256 MiB Infinity Cache
    void* buf = nullptr;
    double* sink = nullptr;
    CHECK(hipMalloc(&buf, bytes + 4096));
    CHECK(hipMalloc((void**)&sink, 64));
    CHECK(hipMemset(buf, 0, bytes + 4096));
    CHECK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_coalesced<double2>, dim3(grid), dim3(256), 0, 0, (const double2*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(read_coalesced<double>, dim3(grid), dim3(256), 0, 0, (const double*)buf, bytes / 8, sink);
        hipLaunchKernelGGL(read_coalesced<float>, dim3(grid), dim3(256), 0, 0, (const float*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(read_rec, dim3(grid * 4), dim3(64), 0, 0, (const double*)buf, bytes / (616 * 8), sink);
        hipLaunchKernelGGL(read_row12, dim3(grid), dim3(256), 0, 0, (const double*)buf, bytes / (28 * 8), sink);
        hipLaunchKernelGGL(write_coalesced<double2>, dim3(grid), dim3(256), 0, 0, (double2*)buf, bytes / 16, double2{0.0, 0.0});
        hipLaunchKernelGGL(write_coalesced<double>, dim3(grid), dim3(256), 0, 0, (double*)buf, bytes / 8, 0.0);
        CHECK(hipDeviceSynchronize());
    }
    // known bytes per launch (what the lanes ask for)
    const size_t recs = bytes / (616 * 8), rows = bytes / (28 * 8) / 5 * 5;
    printf("KNOWN read_coalesced<HIP_vector_type<double, 2u>> %zu\n", bytes);
    printf("KNOWN read_coalesced<double> %zu\n", bytes);
    printf("KNOWN read_coalesced<float> %zu\n", bytes);
    printf("KNOWN read_rec %zu\n", recs * 144 * 8);
    printf("KNOWN read_row12 %zu\n", rows * 12 * 8);
    printf("KNOWN write_coalesced<HIP_vector_type<double, 2u>> %zu\n", bytes);
    printf("KNOWN write_coalesced<double> %zu\n", bytes);
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
