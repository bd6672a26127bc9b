// multi_gpu_batch.cpp -- the batch-sharded solve from a C++ host, without Python / torch.
//
// The batch is the sharding unit: independent OCP instances, NO collective on the data path.  One host thread per slice drives its own
// handle (a handle is not thread-safe, distinct handles are independent):
//     corbo_hip_shard_bounds(global, world, rank) -> [first, first + count)     slice of the global batch
//     corbo_hip_create(desc, count, rank % devices)                             handle on the slice's device
//     set_instance_data / solve                                                 the whole LM loop, device-resident
// and the results of all slices are collected straight from device memory: corbo_hip_device_views gives each handle's iterate array
// [count][row_stride] and its stream, and ONE ncclAllGather per rank (RCCL over xGMI when the slices sit on different GPUs) leaves the
// global result on every device -- what a downstream consumer on the GPUs (a batched evaluation, the next planning layer) reads.
// When several slices share a device (world > devices; the 1-GPU test box) RCCL cannot form that communicator (one rank per device):
// the gather of such a configuration is a device-to-device copy into the gathered array, the solve path is the same.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include examples/multi_gpu_batch.cpp -o examples/multi_gpu_batch \
//         -L control_box_rst_amd/csrc -lcorbo_hip -lrccl -Wl,-rpath,'$ORIGIN/../control_box_rst_amd/csrc'
//   examples/multi_gpu_batch [world = devices] [global_batch = 1024 * world] [N = 100]
// Prints one JSON line; exit code 0 iff the gathered result is bit-identical to ONE handle solving the whole batch on device 0.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "corbo_hip.h"

#define CHECK(call)                                                                                       \
    do {                                                                                                  \
        if ((call) != CORBO_HIP_OK) { fprintf(stderr, "%s: %s\n", #call, corbo_hip_last_error()); std::exit(2); } \
    } while (0)
#define HIPCHECK(call)                                                                                    \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); std::exit(2); } \
    } while (0)
#define NCCLCHECK(call)                                                                                   \
    do {                                                                                                  \
        ncclResult_t r_ = (call);                                                                         \
        if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #call, ncclGetErrorString(r_)); std::exit(2); } \
    } while (0)

// cfg 3 / 4 of BASELINE.json: unicycle point-to-point, FiniteDifferencesGrid, Crank-Nicolson (control_box_rst_amd/problems.py unicycle_desc)
static corbo_hip_problem_desc unicycle_desc(int N)
{
    corbo_hip_problem_desc d;
    std::memset(&d, 0, sizeof(d));
    d.grid = CORBO_HIP_GRID_FD; d.defect = CORBO_HIP_DEFECT_CRANK_NICOLSON; d.dynamics = CORBO_HIP_DYN_UNICYCLE;
    d.stage_cost = CORBO_HIP_COST_QUADRATIC_LSQ; d.final_cost = 1; d.nx = 3; d.nu = 2; d.N = N; d.dt_ref = 0.1; d.dt_ub = CORBO_HIP_INF;
    const double q[3] = {1.0, 1.0, 0.1}, r[2] = {0.1, 0.05};
    for (int i = 0; i < CORBO_HIP_MAX_NX; ++i) { d.x_lb[i] = -CORBO_HIP_INF; d.x_ub[i] = CORBO_HIP_INF; }
    for (int i = 0; i < CORBO_HIP_MAX_NU; ++i) { d.u_lb[i] = -CORBO_HIP_INF; d.u_ub[i] = CORBO_HIP_INF; }
    for (int i = 0; i < 3; ++i) { d.q_diag[i] = q[i]; d.qf_diag[i] = 10.0 * q[i]; d.x_lb[i] = -10.0; d.x_ub[i] = 10.0; }
    for (int i = 0; i < 2; ++i) { d.r_diag[i] = r[i]; d.u_lb[i] = -1.0; d.u_ub[i] = 1.0; }
    return d;
}

// instance b of the global batch: start and goal state from a counter-based generator (the same instance whatever slice it lands in)
static double unit(unsigned long long b, unsigned k)
{
    unsigned long long z = (b + 1) * 0x9E3779B97F4A7C15ull + k * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    return (double)(z >> 11) / 9007199254740992.0;   // [0, 1)
}
static void instances(int first, int count, std::vector<double>& x0, std::vector<double>& xf)
{
    x0.resize((size_t)count * 3); xf.resize((size_t)count * 3);
    for (int i = 0; i < count; ++i) {
        const unsigned long long b = (unsigned long long)(first + i);
        x0[i * 3 + 0] = 2 * unit(b, 0) - 1; x0[i * 3 + 1] = 2 * unit(b, 1) - 1; x0[i * 3 + 2] = (2 * unit(b, 2) - 1) * M_PI / 4;
        xf[i * 3 + 0] = 2.0 + unit(b, 3) - 0.5; xf[i * 3 + 1] = 1.0 + unit(b, 4) - 0.5; xf[i * 3 + 2] = 0.5 + unit(b, 5) - 0.5;
    }
}

struct Slice {
    int rank = 0, first = 0, count = 0, device = 0;
    corbo_hip_handle h = nullptr;
    double* x_dev = nullptr;     // [count][stride] (the handle's own HBM buffer)
    void* stream = nullptr;
    int32_t stride = 0;
    double solve_ms = 0;
};

static void solve_slice(Slice& s, const corbo_hip_problem_desc& d, const corbo_hip_dims& dims)
{
    std::vector<double> x0, xf, X((size_t)s.count * dims.nv);
    instances(s.first, s.count, x0, xf);
    CHECK(corbo_hip_init_trajectory(&d, s.count, x0.data(), xf.data(), X.data()));
    CHECK(corbo_hip_create(&d, s.count, s.device, &s.h));
    CHECK(corbo_hip_set_instance_data(s.h, X.data(), nullptr, nullptr, xf.data()));
    corbo_hip_lm_opts o;
    corbo_hip_default_lm_opts(&o);
    o.weight_eq = o.weight_ineq = o.weight_bounds = 10.0;
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(corbo_hip_solve(s.h, &o, 1));
    CHECK(corbo_hip_synchronize(s.h));
    s.solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    CHECK(corbo_hip_device_views(s.h, &s.x_dev, nullptr, &s.stream));
    CHECK(corbo_hip_device_row_stride(s.h, &s.stride));
}

int main(int argc, char** argv)
{
    int devices = 0;
    CHECK(corbo_hip_device_count(&devices));
    if (devices < 1) { fprintf(stderr, "no GPU (the product has no CPU fallback)\n"); return 2; }
    const int world  = argc > 1 ? std::atoi(argv[1]) : devices;
    const int global = argc > 2 ? std::atoi(argv[2]) : 1024 * world;
    const int N      = argc > 3 ? std::atoi(argv[3]) : 100;
    const corbo_hip_problem_desc d = unicycle_desc(N);
    corbo_hip_dims dims;
    CHECK(corbo_hip_get_dims(&d, &dims));

    // ---- one host thread per slice, one handle per thread
    std::vector<Slice> slices(world);
    for (int r = 0; r < world; ++r) {
        slices[r].rank = r; slices[r].device = r % devices;
        CHECK(corbo_hip_shard_bounds(global, world, r, &slices[r].first, &slices[r].count));
    }
    const auto t0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> th;
        for (int r = 0; r < world; ++r) th.emplace_back([&, r] { solve_slice(slices[r], d, dims); });
        for (auto& t : th) t.join();
    }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const int32_t stride = slices[0].stride;

    // ---- gather: every device ends up with the global [global][stride] iterate array
    const bool one_rank_per_device = (world <= devices);
    std::vector<double*> gathered(world, nullptr);
    const char* how = "";
    int base = global / world;
    if (one_rank_per_device && global % world == 0) {
        how = "ncclAllGather (RCCL) on the handles' HBM buffers and streams";
        std::vector<int> devs(world);
        for (int r = 0; r < world; ++r) devs[r] = slices[r].device;
        std::vector<ncclComm_t> comms(world);
        NCCLCHECK(ncclCommInitAll(comms.data(), world, devs.data()));
        for (int r = 0; r < world; ++r) { HIPCHECK(hipSetDevice(devs[r])); HIPCHECK(hipMalloc((void**)&gathered[r], (size_t)global * stride * sizeof(double))); }
        NCCLCHECK(ncclGroupStart());
        for (int r = 0; r < world; ++r)
            NCCLCHECK(ncclAllGather(slices[r].x_dev, gathered[r], (size_t)base * stride, ncclDouble, comms[r], (hipStream_t)slices[r].stream));
        NCCLCHECK(ncclGroupEnd());
        for (int r = 0; r < world; ++r) { HIPCHECK(hipSetDevice(devs[r])); HIPCHECK(hipStreamSynchronize((hipStream_t)slices[r].stream)); }
        for (int r = 0; r < world; ++r) ncclCommDestroy(comms[r]);
    }
    else {
        how = one_rank_per_device ? "device-to-device copies (uneven slices: all-gather needs equal counts)" : "device-to-device copies (several slices share a device: no RCCL communicator with one rank per device)";
        for (int r = 0; r < world; ++r) {
            HIPCHECK(hipSetDevice(slices[r].device));
            HIPCHECK(hipMalloc((void**)&gathered[r], (size_t)global * stride * sizeof(double)));
            for (int q = 0; q < world; ++q)
                HIPCHECK(hipMemcpy(gathered[r] + (size_t)slices[q].first * stride, slices[q].x_dev, (size_t)slices[q].count * stride * sizeof(double), hipMemcpyDeviceToDevice));
        }
    }

    // ---- check: ONE handle solving the whole batch on device 0 gives the same bits (the instances are independent)
    Slice whole;
    whole.first = 0; whole.count = global; whole.device = 0;
    solve_slice(whole, d, dims);
    std::vector<double> a((size_t)global * stride), b((size_t)global * stride);
    HIPCHECK(hipSetDevice(0));
    HIPCHECK(hipMemcpy(a.data(), whole.x_dev, a.size() * sizeof(double), hipMemcpyDeviceToHost));
    bool identical = true;
    for (int r = 0; r < world && identical; ++r) {
        HIPCHECK(hipSetDevice(slices[r].device));
        HIPCHECK(hipMemcpy(b.data(), gathered[r], b.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < global && identical; ++i)
            identical = std::memcmp(&a[(size_t)i * stride], &b[(size_t)i * stride], dims.nv * sizeof(double)) == 0;
    }
    double slowest = 0;
    for (auto& s : slices) slowest = std::max(slowest, s.solve_ms);
    printf("{\"world\": %d, \"devices\": %d, \"global_batch\": %d, \"N\": %d, \"gather\": \"%s\", \"identical_to_single_handle\": %s, "
           "\"slowest_slice_solve_ms\": %.4f, \"wall_ms_incl_create_upload\": %.3f, \"lm_iterations\": %d}\n",
           world, devices, global, N, how, identical ? "true" : "false", slowest, wall_ms, global * 10);
    for (int r = 0; r < world; ++r) { HIPCHECK(hipSetDevice(slices[r].device)); (void)hipFree(gathered[r]); corbo_hip_destroy(slices[r].h); }
    corbo_hip_destroy(whole.h);
    return identical ? 0 : 1;
}
