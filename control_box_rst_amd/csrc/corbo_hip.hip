// corbo_hip.hip -- implementation of the C-ABI declared in include/corbo_hip.h.
//
// Host side of the MI355X-native NLP inner loop: owns the device buffers of a batch of OCP instances, uploads the static
// task tables (structure.hpp), and drives the pass loop  [factor_kernel -> sweep_kernel(LM trial)]  until every instance
// has run its outer iterations (LevenbergMarquardtSparse::solve, levenberg_marquardt_sparse.cpp:129-217).  All per-instance
// control flow (accept / reject / damping) lives on the device; the host only counts unfinished instances per pass.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <climits>
#include <map>
#include <utility>
#include <vector>

#include "../../include/corbo_hip.h"
#include "kernels.hpp"
#include "structure.hpp"

using namespace corbo_hip;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(CORBO_HIP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

template <class T>
int upload(const std::vector<T>& v, T** dptr)
{
    *dptr = nullptr;
    if (v.empty()) return 0;
    HIP_TRY(hipMalloc((void**)dptr, v.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

constexpr int MAX_PASSES = 4096;

// Every entry point runs on the handle's device and leaves the caller's current device as it found it (the library shares the
// process with the caller's own HIP / PyTorch code).
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        err = hipSetDevice(device);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
// Pending corbo_hip_solve_async launches are drained before any entry point touches iterates, flags, pinned result views or the shared
// h_counter slots (a mutator between solve_async and fetch_solution must not leave the sink "valid" with stale rows).

// A mutator drains and then DOES ITS OWN WORK: a failure of the drained solve ("pass limit reached") is remembered in the handle and reported by the
// next call that hands out results or starts a synchronous solve (corbo_hip_solve / synchronize / fetch_solution / get_*), not by the mutator (ADVICE r5:
// a caller that ignored the mutator's return code carried on with stale data).  Device errors of the drain itself are returned at once.
#define DRAIN_ASYNC(h)                                   \
    do {                                                 \
        const int rc_drain_ = finish_async(h, false);    \
        if (rc_drain_) return rc_drain_;                 \
    } while (0)
#define ON_DEVICE_OF(h)                      \
    DeviceGuard device_guard_((h)->device);  \
    HIP_TRY(device_guard_.err)

// No C++ exception crosses the C boundary (host-side staging buffers are std::vector).
#define ABI_CATCH                                                                                       \
    catch (const std::bad_alloc&) { return fail(CORBO_HIP_ERR_DEVICE, "host allocation failed"); }      \
    catch (const std::exception& e) { return fail(CORBO_HIP_ERR_DEVICE, std::string("exception: ") + e.what()); } \
    catch (...) { return fail(CORBO_HIP_ERR_DEVICE, "unknown exception"); }

// corbo_hip_eval_stage_function: the user stage functions (model.hpp) on the host, by vertex dimension
template <int DIM>
void eval_stage_fn(int id, int kind, int n, const double* v, const double* prm, double* out)
{
    for (int p = 0; p < n; ++p)
        out[p] = (kind == 1) ? stage_ineq_control<DIM>(id, v + (size_t)p * DIM, prm) : stage_ineq_state<DIM>(id, v + (size_t)p * DIM, prm);
}

// per-phase events of a profiled solve: destroyed on every exit path
struct EventList {
    std::vector<hipEvent_t> v;
    ~EventList() { for (hipEvent_t e : v) (void)hipEventDestroy(e); }
};

}  // namespace

static int finish_async(corbo_hip_handle h, bool report = true);   // (DRAIN_ASYNC: report = false)

static constexpr int PTL_LEN = 150 + 18 * 64;   // pass timeline buffer (diagnostics): stamps + per-pass phase log
struct corbo_hip_solver {
    Structure S;
    int batch  = 0;    // capacity: instances the device buffers hold
    int active = 0;    // instances [0, active) take part in solve / warm start / statistics (corbo_hip_set_active; default = batch)
    int device = 0;
    hipStream_t stream = nullptr;
    // the LM passes of a solve run as `nsub` independent sub-batches on their own streams: their kernels interleave on the chip at
    // different phases (latency-bound factor phase of one beside the throughput-bound sweep phase of the other)
    static constexpr int MAX_SUB = 4;
    int nsub = 1;
    hipStream_t sub_stream[MAX_SUB] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t sub_done[MAX_SUB]    = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t sub_chk[MAX_SUB][2]  = {};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_chk[2] = {nullptr, nullptr};
    // static tables
    StageCols* d_stage_cols  = nullptr;
    CompInfo* d_comp         = nullptr;
    int32_t* d_ineq_cols     = nullptr;
    int32_t* d_ineq_rows     = nullptr;
    // per-instance data (HBM resident)
    double *d_x0 = nullptr;  // shadow of the uploaded x (corbo_hip_restore_instance_data)
    double* h_xnew = nullptr;  // pinned, device-visible [batch][MAX_NX]: measured states of corbo_hip_warm_start (read by the kernel)
    // pinned staging of [batch][nvs] doubles for the host-buffer side of the boundary (set_instance_data / get_solution).  Pageable
    // copies of this size make the runtime pin and unpin the caller's (or a temporary's) pages on the fly; releasing those pages
    // afterwards stalls the queue for ~20 ms (measured: first solve after an upload 25 ms instead of 0.76 ms).
    double* h_stage      = nullptr;
    double* h_stage_b[2] = {nullptr, nullptr};   // pinned staging of per-instance lower / upper bounds (allocated on first use)
    double* d_bound_rows = nullptr;  // [2][nvs] the descriptor's bound pattern of one instance (lower row, upper row)
    std::vector<double> bound_rows;  // host copy of the same
    double* d_lin    = nullptr;      // [A | B] of a LinearStateSpaceModel (row-major), else null
    double* d_wdense = nullptr;      // [q_sqrt | r_sqrt | qf_sqrt] (16 doubles each) of a descriptor with non-diagonal weights, else null
    double* d_xplant = nullptr;      // [batch][MAX_NX] plant states of the closed loop (corbo_hip_plant_*)
    double* h_dist   = nullptr;      // pinned, device-visible [batch][MAX_NX]: state disturbance of corbo_hip_plant_step
    bool have_plant  = false;
    // grow-only scratch of corbo_hip_closed_loop: pinned staging (disturbances up, logs down) and the device-side logs
    double* h_loop = nullptr; size_t h_loop_doubles = 0;
    double* d_loop = nullptr; size_t d_loop_doubles = 0;
    LmState* h_state = nullptr;      // pinned [batch] read-back of the per-instance LM state (get_solution / get_stats)
    double* h_chi2   = nullptr;      // pinned [batch] chi2 + [batch] status (int32) of corbo_hip_fetch_solution
    double solve_ms_sum = 0.0;       // HIP-event time of every corbo_hip_solve since the last reset (corbo_hip_get_timing)
    int64_t solve_count = 0;
    double *d_x = nullptr, *d_xt = nullptr, *d_lb = nullptr, *d_ub = nullptr, *d_xref = nullptr;
    double* d_refvec = nullptr;   // per-component references [batch][nvs] (corbo_hip_set_references), allocated on first use
    bool refvec_on   = false;
    double* d_plant_prm = nullptr;   // per-instance plant model parameters [batch][8] (corbo_hip_plant_set_params) or null
    double* d_dyn_inst  = nullptr;   // per-instance parameters of the controller's dynamics [batch][8] (corbo_hip_set_instance_params) or null
    double* d_reftraj = nullptr;  // resident reference trajectory [batch][ref_T][nx] (corbo_hip_set_reference_trajectory)
    int ref_T = 0, ref_step = 0;
    double *d_values0 = nullptr, *d_values1 = nullptr, *d_jac = nullptr;
    LmState* d_state      = nullptr;
    double* d_chi2        = nullptr;  // [batch]
    double* d_work        = nullptr;  // factor workspace (big-block kernel only)
    double* d_xe0         = nullptr;  // big-block family: [2][batch][N][nx] end states of the unperturbed Runge-Kutta steps (SweepParams::xe0)
    size_t work_stride    = 0;
    int num_cus = 0;                  // compute units of the handle's device
    int32_t* d_queue      = nullptr;  // ticket counter of the run-to-completion kernel's instance queue (batches beyond 4 workgroups per CU)
    int32_t* d_counters   = nullptr;  // MAX_PASSES
    int32_t* h_counter    = nullptr;  // pinned
    // corbo_hip_solve_async: solves enqueued and not yet waited for (their timing event pairs; the pass-limit flag alternates between two pinned slots)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> async_events, event_pool;
    int async_pending = 0;
    int m_pad = 0, nnz_pad = 0;
    int jlean_lo2 = 0, jlean_hi2 = 0;   // kernels.hpp SweepParams::jlean_*
    // Hessian-path operators: the structure of a (handle, lower) pair is built once, scratch buffers only grow (an interior-point
    // loop calls these once per iteration)
    struct GrowBuf {
        void* p = nullptr; size_t cap = 0; bool host = false;
        hipError_t need(size_t bytes) {
            if (bytes <= cap && p) return hipSuccess;
            if (p) { if (host) (void)hipHostFree(p); else (void)hipFree(p); p = nullptr; cap = 0; }
            hipError_t e = host ? hipHostMalloc(&p, bytes ? bytes : 8) : hipMalloc(&p, bytes ? bytes : 8);
            if (e == hipSuccess) cap = bytes ? bytes : 8;
            return e;
        }
        void release() { if (p) { if (host) (void)hipHostFree(p); else (void)hipFree(p); } p = nullptr; cap = 0; }
        double* d() const { return static_cast<double*>(p); }
    };
    struct HessCache { bool valid = false; HessianStructure H; GrowBuf d_so, d_lo; } hess_cache[2];
    GrowBuf hb_vals[3], hb_me, hb_mi, hb_lin, hb_lb, hb_ub, hb_grad, hb_obj;   // device
    GrowBuf hb_pin{nullptr, 0, true};                                          // pinned host staging (multipliers up, results down)
    std::vector<int32_t> jmap;        // public Jacobian value index -> device-internal index (see corbo_hip_create)
    int nnz_int = 0;                  // values of the device-internal layout (pads included)
    int32_t fin_joff_dev[CORBO_HIP_MAX_NX];
    bool have_data = false;
    double w_eq = 2, w_ineq = 2, w_b = 2;  // current penalty weights (levenberg_marquardt_sparse.h:126-128)
    corbo_hip_stats stats{};
    bool profile = false;
    bool force_split = false;   // descriptor family without a fused pass kernel
    // integral-form constraint edges / control-deviation edges (Structure::xedges): the sweep's edge table, the plug-in functions' parameters, the
    // previously applied control per instance, and the tables of the band factorisation (kernels.hpp BandParams)
    XEdge* d_xedges = nullptr;
    double *d_xparams = nullptr, *d_uprev = nullptr, *d_band_work = nullptr;
    int32_t *d_band_target = nullptr, *d_band_ptr = nullptr, *d_band_pairs = nullptr, *d_band_rptr = nullptr, *d_band_rent = nullptr, *d_band_voff = nullptr;
    BandParams band{};
    bool split_passes = false;  // profiling: factor and sweep phases of a pass as two launches
    bool result_sink = false;   // corbo_hip_set_result_sink: the run-to-completion kernel writes results into pinned host memory itself
    bool sink_valid  = false;   // ... and the last solve did so
    bool sink_invalidated = false;   // an enqueued re-arm (corbo_hip_restore_instance_data) came after the last enqueued solve: draining must not re-validate the views
    // handles whose passes are launched from the host (big-block family, band route, ...): the results are delivered by a copy instead -- a
    // device-to-device snapshot behind the last pass (10 us), then snapshot -> pinned host memory on a stream of its own, off the critical path of
    // the next solve (13 MB over PCIe for cfg 5: 0.24 ms); corbo_hip_synchronize / corbo_hip_fetch_solution / corbo_hip_get_* wait for it
    double* d_snap = nullptr;            // [batch][nvs] iterates | [batch] LmState
    double* h_sink = nullptr;            // the same layout in pinned host memory (its own buffer: h_stage / h_state stay the staging areas of the other calls)
    bool sink_delivered = false;         // the valid results are in h_sink (else: h_stage / h_state, written by the run-to-completion kernel)
    hipStream_t sink_stream = nullptr;
    hipEvent_t ev_snap = nullptr, ev_sink = nullptr;
    bool sink_pending = false;
    // reject-streak speculation of the big-block family (kernels.hpp SpecParams): spare instance rows behind the batch in the per-instance arrays
    static constexpr int SPEC_GROUPS = 8, SPEC_SLOTS = 4;
    int spare = 0;              // SPEC_GROUPS * SPEC_SLOTS for a big-block handle on the stage / chain path, else 0
    int reject_speculation = 1; // corbo_hip_set_option("reject_speculation"): 0 = every rejected step is a pass of its own (A/B, tests)
    double* d_stage_cache = nullptr;   // big-block family: the stage waves' local Jacobians between the two passes of a solve's first factorisation (FactorParams::stage_cache)
    size_t stage_cache_stride = 0;
    int32_t* d_xtasks = nullptr;   // the extra edges' Jacobian columns, one per sweep lane (kernels.hpp SweepParams::xtasks)
    int n_xtasks = 0;
    uint32_t *d_bt_pairs = nullptr, *d_bt_target = nullptr;   // block-tridiagonal route (structure.hpp BtTables): the small-block families with extra edges, run to completion
    int32_t* d_bt_off = nullptr;
    int bt_rounds = 0;
    double* d_bt_snap = nullptr;       // FactorParams::bt_snap
    int bt_snap_stride = 0;
    int bt_waves = 0;                  // option "bt_waves": 0 = by batch size, 2 / 3 = that instantiation of lm_bt_kernel (A/B)
    bool async_error_deferred = false;   // an enqueued solve hit the pass limit and a mutator drained it: reported by the next result / solve call
    int32_t *d_spec_parent = nullptr, *d_spec_seen = nullptr, *d_spec_slotrej = nullptr, *d_spec_prev = nullptr, *d_spec_adopted = nullptr;
    int hess_split = -1;        // corbo_hip_set_option("hess_split"): -1 = automatic, 0 / 1 / 2 (HessParams::split; tests, A/B)
    int band_wide = 0;          // corbo_hip_set_option("band_wide"): FactorParams::band_wide
    int chain_variant = 0;      // corbo_hip_set_option("chain_variant"): big-block family, see FactorParams::chain_variant
    int pass_limit = 0;         // corbo_hip_set_option("pass_limit"): > 0 lowers the run-to-completion kernel's limit of 4096 LM passes
    int pass_timeline_inst = -1;   // corbo_hip_set_option("pass_timeline"): >= 0 prints that instance's per-pass shader-clock stamps
    bool sweep_timeline = false;   // corbo_hip_set_option("sweep_timeline")
    int stagger = 0;               // corbo_hip_set_option("stagger")
    int pass_threads = 0;          // corbo_hip_set_option("pass_threads"): 0 = the default workgroup size of the run-to-completion kernel
    int lag_priority = 1;          // corbo_hip_set_option("lag_priority")
    bool phase_cycles = false;     // corbo_hip_set_option("phase_cycles"): per-instance phase totals (FactorParams::phase_cycles)
    long long* d_phase = nullptr;  // [batch][8]
    bool raw_stamps = false;       // corbo_hip_set_option("raw_stamps"): the per-pass stamp log as raw offsets (development builds)
    int solve_timing = 1;          // corbo_hip_set_option("solve_timing"): 0 = no HIP events around the launches of a solve (stats.solve_ms stays 0): two
                                   // event records and an event wait cost a batch-1 solve 10 - 13 us, a plain stream synchronisation the rest
    int ff_converged = 1;          // corbo_hip_set_option("ff_converged"): 0 = compute the outer iterations that follow a converged step (A/B, tests)
    bool loop_mode = true;      // run-to-completion pass kernel: one launch per solve (CORBO_HIP_LOOP=0: one launch per LM pass)

    // the 8 model parameters as the kernels see them: the descriptor's, except for the linear state-space model, whose first slot
    // carries the device address of its [A | B] table as a bit pattern (model.hpp, LinearDynamics)
    void fill_dyn(double (&dyn)[8]) const
    {
        std::memcpy(dyn, S.desc.dyn_params, sizeof(dyn));
        dyn[7] = (double)S.desc.shooting_integrator;   // slot 7: the shooting grids' integrator (model.hpp, rk4_end_state)
        if (S.desc.dynamics == CORBO_HIP_DYN_LINEAR_STATE_SPACE) {
            const long long bits = (long long)reinterpret_cast<uintptr_t>(d_lin);
            std::memcpy(&dyn[0], &bits, sizeof(double));
        }
    }
    SweepParams sweep_params(int mode, int iterations, double weq, double wineq, double wb, int32_t* counter) const
    {
        SweepParams p{};
        p.batch = active; p.nvs = S.nvs; p.m = S.dims.m; p.nnz = S.dims.nnz; p.N = S.N; p.s = S.s; p.nx = S.nx; p.off_dt = S.off_dt; p.dt_free = S.dt_free;
        p.batch_total = batch + spare; p.xe0 = d_xe0; p.skip_jac = (d_xe0 && mode >= 2 && band.n == 0) ? 1 : 0;   // (band route: the factorisation reads the stored Jacobian)
        p.eq_row0 = S.eq_row0; p.ineq_row0 = S.ineq_row0;
        p.ineq_stride = 1 + (S.desc.stage_ineq_control ? 1 : 0) + (S.desc.ctrl_dev ? S.nu : 0);   // (creation order per interval: state term, control term, control-deviation term)
        p.stage_cols = d_stage_cols; p.comp = d_comp; p.ineq_cols = d_ineq_cols;
        fill_dyn(p.mp.dyn);
        std::memcpy(p.mp.ineq, S.desc.ineq_params, sizeof(p.mp.ineq));
        p.mp.ineq_id = S.desc.stage_ineq; p.mp.ineq_ctrl_id = S.desc.stage_ineq_control;
        std::memcpy(p.mp.sq, S.sq, sizeof(p.mp.sq));
        std::memcpy(p.mp.sr, S.sr, sizeof(p.mp.sr));
        std::memcpy(p.mp.sqf, S.sqf, sizeof(p.mp.sqf));
        p.mp.dt_weight = S.dt_weight;
        std::memcpy(p.mp.fin, S.desc.final_ineq_params, sizeof(p.mp.fin));
        p.fin_eq_row0 = S.fin_eq_row0; p.fin_eq_dim = S.fin_eq_dim;
        p.xedges = d_xedges; p.n_xedges = (int32_t)S.xedges.size(); p.eq_stride = S.eq_stride; p.eq_defect_off = S.eq_defect_off;
        p.xtasks = reinterpret_cast<const int4*>(d_xtasks); p.n_xtasks = n_xtasks;
        p.xparams = d_xparams; p.uprev = d_uprev;
        p.mp.wdense = d_wdense; p.mp.wdense_mask = d_wdense ? S.desc.weights_dense : 0;
        p.mp.fin_eq_mask = S.desc.final_eq ? (int32_t)S.desc.final_eq_mask : 0;
        p.fin_row = S.fin_row;
        for (int i = 0; i < CORBO_HIP_MAX_NX; ++i) p.fin_joff[i] = fin_joff_dev[i];
        p.dt_fixed = S.desc.dt_ref;
        p.ff_converged = ff_converged;
        p.mode = mode; p.iterations = iterations; p.w_eq = weq; p.w_ineq = wineq; p.w_b = wb;
        p.x = d_x; p.xt = d_xt; p.lb = d_lb; p.ub = d_ub; p.xref = d_xref;
        p.refvec = refvec_on ? d_refvec : nullptr;
        p.dyn_inst = d_dyn_inst;
        p.values0 = d_values0; p.values1 = d_values1; p.jac = d_jac; p.m_pad = m_pad; p.nnz_pad = nnz_pad; p.jlean_lo2 = jlean_lo2; p.jlean_hi2 = jlean_hi2;
        p.st = d_state; p.active_count = counter; p.chi2 = d_chi2;
        return p;
    }
    FactorParams factor_params() const
    {
        FactorParams p{};
        p.batch = active; p.nvs = S.nvs; p.m = S.dims.m; p.N = S.N; p.nx = S.nx; p.nu = S.nu; p.s = S.s; p.off_dt = S.off_dt; p.dt_free = S.dt_free;
        p.eq_row0 = S.eq_row0;
        p.stage_cols = d_stage_cols; p.comp = d_comp; p.ineq_cols = d_ineq_cols; p.ineq_rows = d_ineq_rows;
        p.fin_row = S.fin_row;
        for (int i = 0; i < CORBO_HIP_MAX_NX; ++i) p.fin_joff[i] = fin_joff_dev[i];
        p.x = d_x; p.xt = d_xt; p.values0 = d_values0; p.values1 = d_values1; p.jac = d_jac; p.m_pad = m_pad; p.nnz_pad = nnz_pad; p.jlean_lo2 = jlean_lo2; p.jlean_hi2 = jlean_hi2;
        p.st = d_state; p.delta_out = nullptr;
        p.work = d_work; p.work_stride = (int64_t)work_stride;
        // big-block family, automatic choice of the partitioned chain: four segments (8 waves per instance: one workgroup fills a CU) while every instance of the
        // handle gets a CU of its own in ONE round; two segments (4 waves, two workgroups per CU) for larger batches -- 512 instances are then one round
        // instead of two rounds of single-instance latency (cfg 5: 8.33 -> 8.06 ms per solve; a single OCP: 1.76 against 2.03 ms, hence not always).
        // By the HANDLE's batch, not the launch's active count: every pass of a handle factorises the same way (the segment count changes the
        // elimination order, i.e. the iterates at rounding level -- tests/test_gpu_chain_variants.py).
        p.band_wide = band_wide;
        p.chain_variant = (chain_variant == 0 && big_family_dims(S.nx, S.nu) && S.N >= 64 && num_cus > 0 && batch > num_cus) ? 4 : chain_variant;
        p.stage_cache = d_stage_cache; p.stage_cache_stride = (int64_t)stage_cache_stride;
        p.defect = S.desc.defect;
        p.wdense_mask = d_wdense ? S.desc.weights_dense : 0;
        p.bt_pairs = d_bt_pairs; p.bt_off = d_bt_off; p.bt_target = d_bt_target; p.bt_rounds = bt_rounds; p.bt_snap = d_bt_snap; p.bt_snap_stride = bt_snap_stride; p.bt_waves = bt_waves; p.num_cus = num_cus;
        return p;
    }
};


extern "C" {

void corbo_hip_default_lm_opts(corbo_hip_lm_opts* o)
{
    if (!o) return;
    o->iterations = 10;
    o->weight_eq = o->weight_ineq = o->weight_bounds = 2;
    o->adapt_factor_eq = o->adapt_factor_ineq = o->adapt_factor_bounds = 1;
    o->adapt_max_eq = o->adapt_max_ineq = o->adapt_max_bounds = 500;
}

int corbo_hip_get_dims(const corbo_hip_problem_desc* desc, corbo_hip_dims* dims)
try {
    if (!desc || !dims) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    Structure S;
    std::string err = build_structure(*desc, S);
    if (!err.empty()) return fail(CORBO_HIP_ERR_INVALID, err);
    *dims = S.dims;
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_get_structure(const corbo_hip_problem_desc* desc, int32_t* rows, int32_t* cols)
try {
    if (!desc || !rows || !cols) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    Structure S;
    std::string err = build_structure(*desc, S);
    if (!err.empty()) return fail(CORBO_HIP_ERR_INVALID, err);
    std::memcpy(rows, S.jac_rows.data(), S.jac_rows.size() * sizeof(int32_t));
    std::memcpy(cols, S.jac_cols.data(), S.jac_cols.size() * sizeof(int32_t));
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_init_trajectory(const corbo_hip_problem_desc* desc, int batch, const double* x0, const double* xf, double* x_out)
try {
    if (!desc || !x0 || !xf || !x_out || batch < 0) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    std::string err = validate_desc(*desc);
    if (!err.empty()) return fail(CORBO_HIP_ERR_INVALID, err);
    init_trajectory(*desc, batch, x0, xf, x_out);
    return CORBO_HIP_OK;
}
ABI_CATCH

static int create_impl(const corbo_hip_problem_desc* desc, int batch, int device, uint32_t route, corbo_hip_handle* out);

int corbo_hip_create(const corbo_hip_problem_desc* desc, int batch, int device, corbo_hip_handle* out)
try {
    if (!desc || !out || batch < 1) return fail(CORBO_HIP_ERR_INVALID, "null argument or batch < 1");
    *out = nullptr;
    return create_impl(desc, batch, device, 0u, out);
}
ABI_CATCH

int corbo_hip_create_routed(const corbo_hip_problem_desc* desc, int batch, int device, uint32_t route, corbo_hip_handle* out)
try {
    if (!desc || !out || batch < 1) return fail(CORBO_HIP_ERR_INVALID, "null argument or batch < 1");
    *out = nullptr;
    return create_impl(desc, batch, device, route, out);
}
ABI_CATCH

static int create_impl(const corbo_hip_problem_desc* desc, int batch, int device, uint32_t route, corbo_hip_handle* out)
{
    // owns the half-built handle until it is handed to the caller (exceptions and early returns free everything created so far)
    struct Owner { corbo_hip_solver* p; ~Owner() { if (p) corbo_hip_destroy(p); } } owner{new corbo_hip_solver()};
    corbo_hip_solver* h = owner.p;
    std::string err = build_structure(*desc, h->S);
    if (!err.empty()) return fail(CORBO_HIP_ERR_INVALID, err);
    const Structure& S = h->S;
    {   // device kernels exist for this descriptor?
        FactorParams fp{};
        fp.N = S.N;
        const bool big = big_family_dims(desc->nx, desc->nu);   // big-block family (workspace + stage kernels)
        // small-block families: N <= 256 LDS-resident, 256 < N <= 1024 the long-horizon kernels (factor workspace in HBM; kernels.hip, factor_body GWS)
        if (!device_kernels_exist(*desc) || factor_lds_bytes(*desc, fp) == 0 || (!big && S.N > 256 && factor_work_doubles(*desc) == 0)) {
            return fail(CORBO_HIP_ERR_UNSUPPORTED, "no device kernel for this (nx, nu, N, dynamics) yet");
        }
    }
    h->batch  = batch;
    h->active = batch;
    h->device = device;
    int ndev  = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(CORBO_HIP_ERR_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(CORBO_HIP_ERR_INVALID, "device index out of range");
#define CREATE_TRY(expr)                                                                                                  \
    do {                                                                                                                  \
        hipError_t e_ = (expr);                                                                                           \
        if (e_ != hipSuccess) return fail(CORBO_HIP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)
    DeviceGuard device_guard(device);
    CREATE_TRY(device_guard.err);
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreate(&h->ev0));
    CREATE_TRY(hipEventCreate(&h->ev1));
    {
        int want = (batch >= 512 ? 2 : 1);
        if (want < 1) want = 1;
        if (want > corbo_hip_solver::MAX_SUB) want = corbo_hip_solver::MAX_SUB;
        if (want > batch) want = batch;
        h->nsub = want;
        for (int i = 0; i < h->nsub; ++i) {
            CREATE_TRY(hipStreamCreateWithFlags(&h->sub_stream[i], hipStreamNonBlocking));
            CREATE_TRY(hipEventCreateWithFlags(&h->sub_done[i], hipEventDisableTiming));
            CREATE_TRY(hipEventCreateWithFlags(&h->sub_chk[i][0], hipEventDisableTiming));
            CREATE_TRY(hipEventCreateWithFlags(&h->sub_chk[i][1], hipEventDisableTiming));
        }
    }
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_chk[0], hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_chk[1], hipEventDisableTiming));
    // Device-internal Jacobian value layout (small-block families): the public value order (corbo_hip_get_structure) with ONE pad double in
    // front of every defect edge's block.  The blocks of consecutive stages are then 25 / 19 / ... doubles apart instead of 24 / 18 / ...: a
    // stride of 24 doubles maps the 32 lanes of an LDS access (one lane per stage) onto 4 bank groups -- 8-way conflicts on every gather of
    // the factor phase and every scatter of the sweep phase (SQ_LDS_BANK_CONFLICT: 31 % of the solve kernel's LDS cycles) -- an odd stride
    // spreads them over all banks.  Only the offset tables change; corbo_hip_eval maps the values back to the public order.
    {
        const bool pad_layout = jacobian_staged_in_lds(S.nx, S.N) && !S.has_extra();   // (extra edges: rows per interval are not nx -- public order kept)   // the same predicate as STAGE in sweep_body: only the LDS staging area has bank conflicts to avoid
        const int nnz = S.dims.nnz;
        h->jmap.resize(nnz);
        for (int i = 0; i < nnz; ++i) {
            const int row = S.jac_rows[i];
            int pad = 0;
            if (pad_layout && row >= S.eq_row0) { pad = (row - S.eq_row0) / S.nx + 1; if (pad > S.N) pad = S.N; }
            h->jmap[i] = i + pad;
        }
        auto mp = [&](int o) { return o < 0 ? o : h->jmap[o]; };
        std::vector<StageCols> sc = S.stage_cols;
        for (auto& c : sc) for (int& o : c.col) o = mp(o);
        std::vector<CompInfo> ci = S.comp;
        for (auto& c : ci) { c.cost_joff = mp(c.cost_joff); c.bnd_joff = mp(c.bnd_joff); c.cost2_joff = mp(c.cost2_joff); }
        std::vector<int32_t> ic = S.ineq_cols;
        for (int32_t& o : ic) o = mp(o);
        for (int i = 0; i < CORBO_HIP_MAX_NX; ++i) h->fin_joff_dev[i] = (i < S.nx) ? mp(S.fin_joff[i]) : -1;
        h->nnz_int = nnz ? h->jmap[nnz - 1] + 1 : 0;
        {   // the part of the Jacobian the factor phase reads from the staging area / HBM in the two-wave run-to-completion shape (kernels.hpp, jlean_*):
            // defect blocks (nx rows per column), inequality entries, the final-stage rows and whatever the dt component carries -- not the cost blocks
            // and bound rows of the stage components
            int lo = INT32_MAX, hi = -1;
            auto take = [&](int o, int len) { if (o >= 0) { lo = std::min(lo, o); hi = std::max(hi, o + len); } };
            for (const auto& c : sc) for (int o : c.col) take(o, S.nx);
            for (int32_t o : ic) take(o, 1);
            for (int i = 0; i < S.nx; ++i) take(h->fin_joff_dev[i], 1);
            for (int i = 0; i < S.nx; ++i) take(ci[(size_t)(S.N - 1) * S.s + i].cost2_joff, S.nx);   // (terminal equality: a column of up to nx rows)
            { const CompInfo& d = ci[S.off_dt]; take(d.cost_joff, 1); take(d.cost2_joff, 1); take(d.bnd_joff, 1); }
            if (hi > lo) { h->jlean_lo2 = lo / 2; h->jlean_hi2 = (hi + 1) / 2; }
        }
        if (upload(sc, &h->d_stage_cols) || upload(ci, &h->d_comp) || upload(ic, &h->d_ineq_cols) || upload(S.ineq_rows, &h->d_ineq_rows))
            return CORBO_HIP_ERR_DEVICE;  // message set by upload()
    }
    h->m_pad   = (S.dims.m + 1) & ~1;
    h->nnz_pad = (h->nnz_int + 1) & ~1;
    const size_t B = (size_t)batch;
    // big-block family on the stage / chain path: spare rows behind the batch for the candidates of the reject-streak speculation
    // (a free dt on the stage / chain route -- even block sizes -- carries its border through the candidates' rows like any parameter; the band route has no candidates)
    const bool free_dt_band_route = S.dt_free && big_family_dims(S.nx, S.nu) && (S.nx % 2 != 0 || (route & CORBO_HIP_ROUTE_FREE_DT_BAND));
    h->spare = (big_family_dims(S.nx, S.nu) && !free_dt_band_route && !S.has_extra()) ? corbo_hip_solver::SPEC_GROUPS * corbo_hip_solver::SPEC_SLOTS : 0;
    const size_t BT = B + (size_t)h->spare;
    CREATE_TRY(hipMalloc((void**)&h->d_x, BT * S.nvs * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->d_xt, BT * S.nvs * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->d_x0, B * S.nvs * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->d_lb, BT * S.nvs * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->d_ub, BT * S.nvs * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->d_xref, BT * CORBO_HIP_MAX_NX * sizeof(double)));
    CREATE_TRY(hipHostMalloc((void**)&h->h_xnew, B * CORBO_HIP_MAX_NX * sizeof(double)));
    CREATE_TRY(hipHostMalloc((void**)&h->h_stage, B * (size_t)(S.nvs > CORBO_HIP_MAX_NX ? S.nvs : CORBO_HIP_MAX_NX) * sizeof(double)));
    CREATE_TRY(hipHostMalloc((void**)&h->h_state, B * sizeof(LmState)));
    CREATE_TRY(hipHostMalloc((void**)&h->h_chi2, B * (sizeof(double) + sizeof(int32_t))));
    CREATE_TRY(hipHostMalloc((void**)&h->h_dist, B * CORBO_HIP_MAX_NX * sizeof(double)));
    if (S.desc.dynamics == CORBO_HIP_DYN_LINEAR_STATE_SPACE) {
        std::vector<double> ab((size_t)S.nx * S.nx + (size_t)S.nx * S.nu);
        for (int i = 0; i < S.nx * S.nx; ++i) ab[i] = S.desc.lin_a[i];
        for (int i = 0; i < S.nx * S.nu; ++i) ab[(size_t)S.nx * S.nx + i] = S.desc.lin_b[i];
        CREATE_TRY(hipMalloc((void**)&h->d_lin, ab.size() * sizeof(double)));
        CREATE_TRY(hipMemcpy(h->d_lin, ab.data(), ab.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (S.desc.weights_dense) {
        double wd[48];
        std::memcpy(wd, S.desc.q_sqrt, 16 * sizeof(double)); std::memcpy(wd + 16, S.desc.r_sqrt, 16 * sizeof(double)); std::memcpy(wd + 32, S.desc.qf_sqrt, 16 * sizeof(double));
        CREATE_TRY(hipMalloc((void**)&h->d_wdense, sizeof(wd)));
        CREATE_TRY(hipMemcpy(h->d_wdense, wd, sizeof(wd), hipMemcpyHostToDevice));
    }
    CREATE_TRY(hipMalloc((void**)&h->d_xplant, B * CORBO_HIP_MAX_NX * sizeof(double)));
    CREATE_TRY(hipMemset(h->d_xplant, 0, B * CORBO_HIP_MAX_NX * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->d_bound_rows, 2 * (size_t)S.nvs * sizeof(double)));
    {   // bound pattern of one instance (descriptor boxes along the horizon, x_f, free dt)
        std::vector<double>& rows = h->bound_rows;
        rows.assign(2 * (size_t)S.nvs, 0.0);
        double *dlb = rows.data(), *dub = rows.data() + S.nvs;
        for (int i = 0; i < S.nvs; ++i) { dlb[i] = -CORBO_HIP_INF; dub[i] = CORBO_HIP_INF; }
        for (int k = 0; k < S.N - 1; ++k) {
            for (int i = 0; i < S.nx; ++i) { dlb[k * S.s + i] = S.desc.x_lb[i]; dub[k * S.s + i] = S.desc.x_ub[i]; }
            for (int i = 0; i < S.nu; ++i) { dlb[k * S.s + S.nx + i] = S.desc.u_lb[i]; dub[k * S.s + S.nx + i] = S.desc.u_ub[i]; }
        }
        for (int i = 0; i < S.nx; ++i) { dlb[S.off_xf + i] = S.desc.x_lb[i]; dub[S.off_xf + i] = S.desc.x_ub[i]; }
        if (S.dt_free) { dlb[S.off_dt] = S.desc.dt_lb; dub[S.off_dt] = S.desc.dt_ub; }
        CREATE_TRY(hipMemcpy(h->d_bound_rows, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    CREATE_TRY(hipMalloc((void**)&h->d_values0, BT * h->m_pad * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->d_values1, BT * h->m_pad * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->d_jac, B * h->nnz_pad * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&h->d_state, BT * sizeof(LmState)));
    CREATE_TRY(hipMalloc((void**)&h->d_chi2, BT * sizeof(double)));
    // (the band route -- decided below -- reads the sweep's stored Jacobian: it needs neither the stage / chain workspace nor the first factorisation's cache)
    // A free dt around a big-block model: even state-block sizes (6, 8, 10, 12 rows) carry it through the partitioned chain as a second right-hand side (round 5);
    // odd block sizes -- and corbo_hip_create_routed(.., CORBO_HIP_ROUTE_FREE_DT_BAND), the A/B switch of bench.py's band leg -- take the band route.
    const bool free_dt_band = free_dt_band_route;
    const bool band_route_early = S.has_extra() || free_dt_band;
    h->work_stride = factor_work_doubles(*desc);
    if (h->work_stride) {
        if (!band_route_early || !big_family_dims(S.nx, S.nu)) CREATE_TRY(hipMalloc((void**)&h->d_work, BT * h->work_stride * sizeof(double)));
        if (big_family_dims(S.nx, S.nu)) {
            CREATE_TRY(hipMalloc((void**)&h->d_xe0, 2 * BT * (size_t)S.N * S.nx * sizeof(double)));
            CREATE_TRY(hipMemset(h->d_xe0, 0, 2 * BT * (size_t)S.N * S.nx * sizeof(double)));
        }
        if (big_family_dims(S.nx, S.nu) && !band_route_early) {
            h->stage_cache_stride = big_stage_cache_doubles(*desc, S.N);
            const size_t bytes = BT * h->stage_cache_stride * sizeof(double);
            if (bytes > ((size_t)2 << 30)) h->stage_cache_stride = 0;   // (beyond 2 GB the first factorisation integrates twice, as before)
            else CREATE_TRY(hipMalloc((void**)&h->d_stage_cache, bytes));
        }
        if (h->spare) {
            CREATE_TRY(hipMalloc((void**)&h->d_spec_parent, corbo_hip_solver::SPEC_GROUPS * sizeof(int32_t)));
            CREATE_TRY(hipMalloc((void**)&h->d_spec_seen, corbo_hip_solver::SPEC_GROUPS * sizeof(int32_t)));
            CREATE_TRY(hipMalloc((void**)&h->d_spec_slotrej, (size_t)h->spare * sizeof(int32_t)));
            CREATE_TRY(hipMalloc((void**)&h->d_spec_prev, B * sizeof(int32_t)));
            CREATE_TRY(hipMalloc((void**)&h->d_spec_adopted, sizeof(int32_t)));
            CREATE_TRY(hipMemset(h->d_spec_adopted, 0, sizeof(int32_t)));
        }
        h->force_split = true;  // no fused pass kernel for the big-block family / the long horizons: factor and sweep are separate launches
    }
    // The band factorisation (band_factor_kernel: H = J^T J from the stored Jacobian through static product lists, natural parameter order, a free dt
    // as a border) takes the structures the stage-parallel kernels do not cover: integral-form constraint edges / control-deviation edges, and a FREE dt
    // with state blocks of 5, 7, 9 or 11 rows (MultipleShootingVariableGrid / FiniteDifferencesVariableGrid around a big-block model whose chain
    // kernel is not the partitioned one) -- the general, slower path: one workgroup per instance, n sequential pivots.
    const bool band_route = band_route_early;
    if (S.has_extra()) {
        // ---- the sweep's extra-edge table, the plug-in parameters, the previous control (zeros, dt_ref: structured_optimal_control_problem.cpp:67-71)
        std::vector<XEdge> xe = S.xedges;   // (Jacobian offsets: the device-internal layout is the public order for these handles)
        CREATE_TRY(hipMalloc((void**)&h->d_xedges, xe.size() * sizeof(XEdge)));
        CREATE_TRY(hipMemcpy(h->d_xedges, xe.data(), xe.size() * sizeof(XEdge), hipMemcpyHostToDevice));
        {   // one sweep lane per Jacobian column of these edges
            std::vector<int32_t> tasks;
            for (size_t e = 0; e < xe.size(); ++e)
                for (int vi = 0; vi < xe[e].nverts; ++vi) {
                    if (xe[e].joff[vi] < 0) continue;
                    int col = 0;
                    for (int c = 0; c < xe[e].vdim[vi]; ++c) {
                        if ((xe[e].fixed[vi] >> c) & 1u) continue;
                        tasks.push_back((int32_t)e); tasks.push_back(vi); tasks.push_back(c); tasks.push_back(xe[e].joff[vi] + col * xe[e].edim + xe[e].rie);
                        ++col;
                    }
                }
            h->n_xtasks = (int32_t)(tasks.size() / 4);
            if (upload(tasks, &h->d_xtasks)) return CORBO_HIP_ERR_DEVICE;
        }
        std::vector<double> xp(CORBO_HIP_MAX_NX + 2 * CORBO_HIP_MAX_NU + 2 + 8, 0.0);   // [stage_eq: a, b, c | ctrl_dev: r_max | control inequality: 8 parameters]
        for (int i = 0; i < S.nx + S.nu + 1; ++i) xp[i] = S.desc.stage_eq_params[i];
        for (int i = 0; i < S.nu; ++i) xp[S.nx + S.nu + 1 + i] = S.desc.ctrl_dev_params[i];
        for (int i = 0; i < 8; ++i) xp[S.nx + 2 * S.nu + 1 + i] = S.desc.ineq_control_params[i];
        CREATE_TRY(hipMalloc((void**)&h->d_xparams, xp.size() * sizeof(double)));
        CREATE_TRY(hipMemcpy(h->d_xparams, xp.data(), xp.size() * sizeof(double), hipMemcpyHostToDevice));
        std::vector<double> up(B * (CORBO_HIP_MAX_NU + 1), 0.0);
        for (size_t b = 0; b < B; ++b) up[b * (CORBO_HIP_MAX_NU + 1) + CORBO_HIP_MAX_NU] = S.desc.dt_ref;
        CREATE_TRY(hipMalloc((void**)&h->d_uprev, up.size() * sizeof(double)));
        CREATE_TRY(hipMemcpy(h->d_uprev, up.data(), up.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    // Small-block families with extra edges: H is block tridiagonal in the stage blocks (x_k, u_k) whatever the extra edges are (control deviation: u_k with
    // u_{k-1}; integral forms: (x_k, u_k) with x_{k+1}) -- the run-to-completion kernel of the block-tridiagonal route (kernels.hip lm_bt_kernel)
    // solves them in ONE launch.  The band tables are built all the same: the per-pass modes (option run_to_completion = 0, profiling, corbo_hip_time_factor)
    // and corbo_hip_create_routed(.., CORBO_HIP_ROUTE_XE_BAND) -- the A/B switch -- run the band kernels.
    bool bt_route = false;
    if (S.has_extra() && !big_family_dims(S.nx, S.nu) && !(route & CORBO_HIP_ROUTE_XE_BAND) && S.desc.shooting_integrator < 5 && !S.desc.weights_dense) {
        const int max_rounds = bt_route_max_rounds(S.nx, S.nu, S.dt_free, S.N, (int)h->nnz_pad, (int)h->m_pad, S.nvs);
        BtTables bt;
        std::string why;
        if (max_rounds > 0 && build_bt_tables(S, h->jmap, (int)h->nnz_pad, (int)h->m_pad, BT_THREADS, bt, &why) && bt.rounds <= max_rounds) {
            if (upload(bt.pairs, &h->d_bt_pairs) || upload(bt.off, &h->d_bt_off) || upload(bt.target, &h->d_bt_target)) return CORBO_HIP_ERR_DEVICE;
            h->bt_rounds = bt.rounds;
            bt_route = true;
            {   // the assembled blocks of an instance's last factorisation (re-used after a rejected step): N blocks of 2 s^2 + s (+ s: free dt) doubles, odd stride, + 3
                const int s_ = S.nx + S.nu, szp = (2 * s_ * s_ + s_ + (S.dt_free ? s_ : 0)) | 1;
                h->bt_snap_stride = (S.N * szp + 3 + 1) / 2 * 2;
                CREATE_TRY(hipMalloc((void**)&h->d_bt_snap, B * (size_t)h->bt_snap_stride * sizeof(double)));
            }
        }
    }
    if (band_route) {
        if (!bt_route) h->force_split = true;   // separate launches: sweep_kernel (residual + Jacobian in HBM) + band_factor_kernel
        // ---- band factorisation tables: H = J^T J entry by entry as sums of products of Jacobian values (natural parameter order; a free dt
        //      -- the last parameter -- as a border)
        const int n = S.dims.n, nb = S.dt_free ? n - 1 : n, m = S.dims.m, nnz = S.dims.nnz;
        std::vector<std::vector<std::pair<int, int>>> rows(m);   // per residual row: (column, Jacobian value index)
        for (int i = 0; i < nnz; ++i) rows[S.jac_rows[i]].push_back({S.jac_cols[i], h->jmap[i]});
        int bw = 0;
        std::map<std::pair<int, int>, std::vector<std::pair<int, int>>> ent;   // (r, c <= r) -> products
        std::vector<std::vector<std::pair<int, int>>> rl(n);                   // per column: (Jacobian value index, residual row)
        for (int r = 0; r < m; ++r)
            for (const auto& a : rows[r]) {
                rl[a.first].push_back({a.second, r});
                for (const auto& b : rows[r]) {
                    if (b.first > a.first) continue;
                    ent[{a.first, b.first}].push_back({a.second, b.second});
                    if (a.first < nb && a.first - b.first > bw) bw = a.first - b.first;
                }
            }
        // every position of the band gets a list (an empty one for a structural zero inside the band): the assembly kernel writes all of it, no zero fill
        for (int r = 0; r < nb; ++r)
            for (int c = (r - bw > 0 ? r - bw : 0); c <= r; ++c) ent[{r, c}];
        std::vector<int32_t> tgt, ptr{0}, pairs, rptr{0}, rent;
        for (const auto& kv : ent) {
            const int r = kv.first.first, c = kv.first.second;
            tgt.push_back(r < nb ? r * (bw + 1) + bw - (r - c) : (c < nb ? -1 - c : INT32_MIN));
            for (const auto& pr : kv.second) { pairs.push_back(pr.first); pairs.push_back(pr.second); }
            ptr.push_back((int32_t)(pairs.size() / 2));
        }
        for (int c = 0; c < n; ++c) {
            for (const auto& pr : rl[c]) { rent.push_back(pr.first); rent.push_back(pr.second); }
            rptr.push_back((int32_t)(rent.size() / 2));
        }
        if (upload(tgt, &h->d_band_target) || upload(ptr, &h->d_band_ptr) || upload(pairs, &h->d_band_pairs) || upload(rptr, &h->d_band_rptr) ||
            upload(rent, &h->d_band_rent) || upload(S.param_voff, &h->d_band_voff))
            return CORBO_HIP_ERR_DEVICE;
        BandParams& bp = h->band;
        if (S.desc.shooting_integrator >= 5 && big_family_dims(S.nx, S.nu)) {   // (only reachable through CORBO_HIP_ROUTE_FREE_DT_BAND: structure.cpp refuses the others)
            g_last_error = "band factorisation around a big-block model: shooting integrators up to Runge-Kutta 4";
            return CORBO_HIP_ERR_UNSUPPORTED;
        }
        if (!band_route_supported(nb, bw)) {   // refused here, not at the first solve (include/corbo_hip.h: unsupported descriptors fail at create)
            g_last_error = "band factorisation: half-bandwidth " + std::to_string(bw) + " beyond 63, or window + vectors of " + std::to_string(nb) + " parameters beyond 160 KB of LDS";
            return CORBO_HIP_ERR_UNSUPPORTED;
        }
        bp.n = n; bp.nb = nb; bp.bw = bw; bp.n_ent = (int32_t)tgt.size();
        bp.ent_target = h->d_band_target; bp.ent_ptr = h->d_band_ptr; bp.ent_pairs = h->d_band_pairs; bp.rhs_ptr = h->d_band_rptr; bp.rhs_ent = h->d_band_rent;
        bp.param_voff = h->d_band_voff;
        bp.work_stride = (int64_t)band_work_doubles(nb, bw);
        bp.use_lds = 0;   // (the band lives in HBM, its sliding window and the vectors in LDS: band_factor_kernel)
        CREATE_TRY(hipMalloc((void**)&h->d_band_work, B * (size_t)bp.work_stride * sizeof(double)));
        CREATE_TRY(hipMemset(h->d_band_work, 0, B * (size_t)bp.work_stride * sizeof(double)));
        bp.work = h->d_band_work;
    }
    if (S.desc.shooting_integrator >= 5) h->force_split = true;   // Runge-Kutta 5 - 7: a defect formula of the stand-alone kernels only (model.hpp, DEFECT_SHOOTING_HIGH)
    if (S.desc.weights_dense && S.N > 256) h->force_split = true;   // non-diagonal weights beyond 256 grid points: the DENSE instantiations of the long-horizon (stand-alone) kernels; up to 256: lm_pass_kernel<.., DENSE> (round 6)
    CREATE_TRY(hipMemset(h->d_chi2, 0, B * sizeof(double)));
    {
        hipDeviceProp_t prop;
        CREATE_TRY(hipGetDeviceProperties(&prop, device));
        h->num_cus = prop.multiProcessorCount;
    }
    CREATE_TRY(hipMalloc((void**)&h->d_queue, (size_t)(16 + 2048 * 16) * sizeof(int32_t)));   // [0] ticket counter, [16..] per-CU progress table
    CREATE_TRY(hipMemset(h->d_queue, 0, (size_t)(16 + 2048 * 16) * sizeof(int32_t)));
    CREATE_TRY(hipMalloc((void**)&h->d_counters, (size_t)corbo_hip_solver::MAX_SUB * MAX_PASSES * sizeof(int32_t)));
    CREATE_TRY(hipHostMalloc((void**)&h->h_counter, 2 * corbo_hip_solver::MAX_SUB * sizeof(int32_t)));
    CREATE_TRY(hipMemset(h->d_state, 0, (B + (size_t)h->spare) * sizeof(LmState)));
    CREATE_TRY(hipMemset(h->d_values0, 0, B * h->m_pad * sizeof(double)));
    CREATE_TRY(hipMemset(h->d_values1, 0, B * h->m_pad * sizeof(double)));
    CREATE_TRY(hipMemset(h->d_jac, 0, B * h->nnz_pad * sizeof(double)));
    // nothing the kernels may read depends on what the allocator handed out
    CREATE_TRY(hipMemset(h->d_x, 0, B * S.nvs * sizeof(double)));
    CREATE_TRY(hipMemset(h->d_xt, 0, B * S.nvs * sizeof(double)));
    CREATE_TRY(hipMemset(h->d_x0, 0, B * S.nvs * sizeof(double)));
    CREATE_TRY(hipMemset(h->d_lb, 0, B * S.nvs * sizeof(double)));
    CREATE_TRY(hipMemset(h->d_ub, 0, B * S.nvs * sizeof(double)));
    CREATE_TRY(hipMemset(h->d_xref, 0, B * CORBO_HIP_MAX_NX * sizeof(double)));
    if (h->d_work) CREATE_TRY(hipMemset(h->d_work, 0, B * h->work_stride * sizeof(double)));
    // hipMemset of device memory is asynchronous with respect to the host and runs on the NULL stream; the handle's streams are
    // non-blocking, i.e. NOT ordered behind it.  Without this wait a memset that is still queued (several processes sharing the GPU)
    // wiped data uploaded by the first corbo_hip_set_instance_data (found by tools/stress_eval.py: four concurrent processes).
    CREATE_TRY(hipDeviceSynchronize());
#undef CREATE_TRY

    owner.p = nullptr;
    *out    = h;
    return CORBO_HIP_OK;
}

void corbo_hip_destroy(corbo_hip_handle h)
{
    if (!h) return;
    DeviceGuard device_guard(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* ptrs[] = {h->d_stage_cols, h->d_comp, h->d_ineq_cols, h->d_ineq_rows,
                    h->d_x0, h->d_x, h->d_xt, h->d_lb, h->d_ub, h->d_xref, h->d_values0, h->d_values1, h->d_jac, h->d_state, h->d_chi2, h->d_work, h->d_xe0, h->d_counters, h->d_queue, h->d_bound_rows, h->d_xplant, h->d_loop, h->d_lin, h->d_wdense, h->d_refvec, h->d_reftraj, h->d_plant_prm, h->d_dyn_inst,
                    h->d_xedges, h->d_xparams, h->d_uprev, h->d_band_work, h->d_band_target, h->d_band_ptr, h->d_band_pairs, h->d_band_rptr, h->d_band_rent, h->d_band_voff,
                    h->d_spec_parent, h->d_spec_seen, h->d_spec_slotrej, h->d_spec_prev, h->d_spec_adopted, h->d_stage_cache, h->d_phase, h->d_bt_pairs, h->d_bt_target, h->d_bt_off, h->d_xtasks, h->d_bt_snap};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (auto& c : h->hess_cache) { c.d_so.release(); c.d_lo.release(); }
    for (auto& b : h->hb_vals) b.release();
    h->hb_me.release(); h->hb_mi.release(); h->hb_lin.release(); h->hb_lb.release(); h->hb_ub.release(); h->hb_grad.release(); h->hb_obj.release(); h->hb_pin.release();
    for (auto* list : {&h->async_events, &h->event_pool})
        for (auto& pr : *list) { if (pr.first) (void)hipEventDestroy(pr.first); if (pr.second) (void)hipEventDestroy(pr.second); }
    if (h->h_counter) (void)hipHostFree(h->h_counter);
    if (h->h_xnew) (void)hipHostFree(h->h_xnew);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    for (int w = 0; w < 2; ++w) if (h->h_stage_b[w]) (void)hipHostFree(h->h_stage_b[w]);
    if (h->h_state) (void)hipHostFree(h->h_state);
    if (h->h_chi2) (void)hipHostFree(h->h_chi2);
    if (h->h_dist) (void)hipHostFree(h->h_dist);
    if (h->h_loop) (void)hipHostFree(h->h_loop);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (hipEvent_t e : h->ev_chk) if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < corbo_hip_solver::MAX_SUB; ++i) {
        if (h->sub_done[i]) (void)hipEventDestroy(h->sub_done[i]);
        for (hipEvent_t e : h->sub_chk[i]) if (e) (void)hipEventDestroy(e);
        if (h->sub_stream[i]) { (void)hipStreamSynchronize(h->sub_stream[i]); (void)hipStreamDestroy(h->sub_stream[i]); }
    }
    if (h->sink_stream) { (void)hipStreamSynchronize(h->sink_stream); (void)hipStreamDestroy(h->sink_stream); }
    if (h->ev_snap) (void)hipEventDestroy(h->ev_snap);
    if (h->ev_sink) (void)hipEventDestroy(h->ev_sink);
    if (h->d_snap) (void)hipFree(h->d_snap);
    if (h->h_sink) (void)hipHostFree(h->h_sink);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int corbo_hip_set_instance_data(corbo_hip_handle h, const double* x, const double* lb, const double* ub, const double* xref)
try {
    if (!h || !x) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    h->sink_valid = false;   // the pinned result views are stale from here on
    const Structure& S = h->S;
    const int nv = S.dims.nv, nvs = S.nvs, B = h->batch;
    // Repack the public vertex layout (nv per row) into the device vertex storage (nvs per row: + fixed dt + padding) through the
    // pinned staging buffer, one array at a time on the handle's stream; the shadow copies are made on the device.
    const size_t row_bytes = (size_t)nvs * sizeof(double), all_bytes = (size_t)B * row_bytes;
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int b = 0; b < B; ++b) {
        double* o = h->h_stage + (size_t)b * nvs;
        std::memcpy(o, x + (size_t)b * nv, nv * sizeof(double));
        for (int i = nv; i < nvs; ++i) o[i] = 0.0;
        if (!S.dt_free) o[S.off_dt] = S.desc.dt_ref;   // a fixed dt lives in the vertex storage too
    }
    // bounds: the descriptor's pattern for every instance, overwritten by the caller's per-instance arrays where given
    const double* b_src[2] = {nullptr, nullptr};
    if (lb || ub) {
        // entries beyond nv (fixed dt, padding) keep the pattern's "unbounded": stage the pattern row, then the caller's values
        const std::vector<double>& rows = h->bound_rows;
        // the bound ROWS are static (one per unfixed component with a finite descriptor bound): a per-instance bound must not turn an
        // unbounded component into a bounded one -- it would silently not be enforced -- nor the other way round
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < nv; ++i) {
                if (S.comp[i].fixed) continue;
                const double l = lb ? lb[(size_t)b * nv + i] : rows[i], u = ub ? ub[(size_t)b * nv + i] : rows[(size_t)nvs + i];
                const bool finite = (l > -CORBO_HIP_INF) || (u < CORBO_HIP_INF);   // vector_vertex.h:174-184
                if (finite != (S.comp[i].bnd_row >= 0))
                    return fail(CORBO_HIP_ERR_INVALID, "per-instance bounds change the descriptor's finiteness pattern (instance " + std::to_string(b) +
                                                           ", component " + std::to_string(i) + "): bound rows are static");
            }
        // each array through its own pinned staging buffer: no copy engine (a hipMemcpyAsync wakes one: 0.1 - 0.3 ms before the next kernel
        // may start) and no synchronisation in between -- this is the per-solve path of the drop-in adapter (one OCP, bounds re-read from the
        // vertices on every solve)
        for (int which = 0; which < 2; ++which) {
            const double* src = which == 0 ? lb : ub;
            if (!src) continue;
            if (!h->h_stage_b[which]) HIP_TRY(hipHostMalloc((void**)&h->h_stage_b[which], all_bytes));
            double* st = h->h_stage_b[which];
            for (int b = 0; b < B; ++b) {
                double* o = st + (size_t)b * nvs;
                std::memcpy(o, rows.data() + (size_t)which * nvs, row_bytes);
                std::memcpy(o, src + (size_t)b * nv, nv * sizeof(double));
            }
            b_src[which] = st;
        }
    }
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < CORBO_HIP_MAX_NX; ++i)
            h->h_xnew[(size_t)b * CORBO_HIP_MAX_NX + i] = (xref && i < S.nx) ? xref[(size_t)b * S.nx + i] : 0.0;
    // ONE kernel reads the pinned staging buffers (device-visible host memory): iterate -> accepted / trial / re-arm copies, the bounds (staged
    // arrays or the pattern rows), the state references
    UploadParams up{};
    up.x = reinterpret_cast<const double2*>(h->h_stage); up.dx = reinterpret_cast<double2*>(h->d_x); up.dxt = reinterpret_cast<double2*>(h->d_xt);
    up.dx0 = reinterpret_cast<double2*>(h->d_x0); up.n2 = all_bytes / sizeof(double2);
    up.lb_src = reinterpret_cast<const double2*>(b_src[0]); up.ub_src = reinterpret_cast<const double2*>(b_src[1]);
    up.row_lb = reinterpret_cast<const double2*>(h->d_bound_rows); up.row_ub = reinterpret_cast<const double2*>(h->d_bound_rows + nvs);
    up.dlb = reinterpret_cast<double2*>(h->d_lb); up.dub = reinterpret_cast<double2*>(h->d_ub); up.nvs2 = (size_t)nvs / 2;
    up.xref = reinterpret_cast<const double2*>(h->h_xnew); up.dxref = reinterpret_cast<double2*>(h->d_xref); up.nref2 = (size_t)B * CORBO_HIP_MAX_NX / 2;
    launch_upload_instance(up, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));   // the staging buffers are the caller's again (deferring this wait to the solve that follows was
                                                // measured: no gain -- waiting on a stream that is about to drain costs next to nothing)
    h->have_data = true;
    return CORBO_HIP_OK;
}
ABI_CATCH

static int launch_sweep_checked(corbo_hip_handle h, const SweepParams& p)
{
    if (!launch_sweep(h->S.desc, p, h->stream)) return fail(CORBO_HIP_ERR_UNSUPPORTED, "no sweep kernel for this dynamics/defect");
    HIP_TRY(hipGetLastError());
    return 0;
}
static int launch_factor_checked(corbo_hip_handle h, const FactorParams& p)
{
    const SweepParams sp = h->sweep_params(3, 0, h->w_eq, h->w_ineq, h->w_b, nullptr);
    if (h->band.n > 0) { if (!launch_band_factor(p, h->band, h->stream)) return fail(CORBO_HIP_ERR_UNSUPPORTED, "band factorisation: half-bandwidth beyond 63 or more than 160 KB of LDS"); }
    else if (!launch_factor(h->S.desc, p, h->stream, &sp)) return fail(CORBO_HIP_ERR_UNSUPPORTED, "no factor kernel for this nx/nu/N");
    HIP_TRY(hipGetLastError());
    return 0;
}

// penalty weights: resetWeights / adaptWeights (levenberg_marquardt_sparse.cpp:83-86, 270-287)
static void update_penalty_weights(corbo_hip_handle h, const corbo_hip_lm_opts* o, int new_run)
{
    if (new_run) { h->w_eq = o->weight_eq; h->w_ineq = o->weight_ineq; h->w_b = o->weight_bounds; return; }
    h->w_eq *= o->adapt_factor_eq;       if (h->w_eq > o->adapt_max_eq) h->w_eq = o->adapt_max_eq;
    h->w_ineq *= o->adapt_factor_ineq;   if (h->w_ineq > o->adapt_max_ineq) h->w_ineq = o->adapt_max_ineq;
    h->w_b *= o->adapt_factor_bounds;    if (h->w_b > o->adapt_max_bounds) h->w_b = o->adapt_max_bounds;
}

int corbo_hip_set_previous_control(corbo_hip_handle h, const double* u_prev, const double* dt_prev)
try {
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    if (!h->d_uprev) return CORBO_HIP_OK;   // no edge of this handle looks at the previous control
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    const Structure& S = h->S;
    std::vector<double> up((size_t)h->batch * (CORBO_HIP_MAX_NU + 1), 0.0);
    for (int b = 0; b < h->batch; ++b) {
        for (int i = 0; i < S.nu; ++i) up[(size_t)b * (CORBO_HIP_MAX_NU + 1) + i] = (u_prev && b < h->active) ? u_prev[(size_t)b * S.nu + i] : 0.0;
        const double dtp = (dt_prev && b < h->active) ? dt_prev[b] : S.desc.dt_ref;
        if (!(dtp > 0)) return fail(CORBO_HIP_ERR_INVALID, "corbo_hip_set_previous_control: dt_prev must be > 0");
        up[(size_t)b * (CORBO_HIP_MAX_NU + 1) + CORBO_HIP_MAX_NU] = dtp;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(h->d_uprev, up.data(), up.size() * sizeof(double), hipMemcpyHostToDevice));
    return CORBO_HIP_OK;
}
ABI_CATCH

// what a synchronous solve does behind its last launch, for the solves of corbo_hip_solve_async that are still in flight: wait, add their HIP-event times
// to the handle's timing, check the pass-limit flags
// the delivery of a host-driven handle's results (snapshot -> pinned host memory on sink_stream) has landed
static int wait_sink(corbo_hip_handle h)
{
    if (!h->sink_pending) return CORBO_HIP_OK;
    HIP_TRY(hipEventSynchronize(h->ev_sink));
    h->sink_pending = false;
    return CORBO_HIP_OK;
}

// enqueue that delivery behind everything on the handle's stream
static int deliver_results(corbo_hip_handle h)
{
    const size_t xd = (size_t)h->batch * h->S.nvs, sd = (size_t)h->batch * sizeof(LmState) / sizeof(double);
    if (!h->d_snap) {
        HIP_TRY(hipMalloc((void**)&h->d_snap, (xd + sd) * sizeof(double)));
        HIP_TRY(hipHostMalloc((void**)&h->h_sink, (xd + sd) * sizeof(double)));
        HIP_TRY(hipStreamCreateWithFlags(&h->sink_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_snap, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_sink, hipEventDisableTiming));
    }
    if (h->sink_pending) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_sink, 0));   // the previous delivery has finished reading the snapshot
    launch_copy_rows(h->d_x, h->d_snap, nullptr, xd, h->stream);
    launch_copy_rows(reinterpret_cast<const double*>(h->d_state), h->d_snap + xd, nullptr, sd, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev_snap, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->sink_stream, h->ev_snap, 0));
    // (a copy ENGINE, not a copy kernel: a kernel's posted writes over PCIe fill the write path of the whole chip -- measured: the next step's 7 us re-arm
    //  copy, started while such a kernel ran on the other stream, ended with it 164 us later, and everything queued behind it waited)
    HIP_TRY(hipMemcpyAsync(h->h_sink, h->d_snap, (xd + sd) * sizeof(double), hipMemcpyDeviceToHost, h->sink_stream));
    HIP_TRY(hipEventRecord(h->ev_sink, h->sink_stream));
    h->sink_pending = true;
    return CORBO_HIP_OK;
}

static int finish_async(corbo_hip_handle h, bool report)
{
    auto deferred = [&]() -> int {
        if (!report || !h->async_error_deferred) return CORBO_HIP_OK;
        h->async_error_deferred = false;
        return fail(CORBO_HIP_ERR_DEVICE, "pass limit reached with unfinished instances");
    };
    if (h->async_pending == 0) return deferred();
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (auto& pr : h->async_events) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { h->solve_ms_sum += ms; h->stats.solve_ms = ms; }
        h->event_pool.push_back(pr);
    }
    h->async_events.clear();
    h->async_pending = 0;
    const bool unfinished = h->h_counter[0] != 0 || h->h_counter[1] != 0;
    h->h_counter[0] = h->h_counter[1] = 0;
    if (unfinished) { h->async_error_deferred = true; return deferred(); }
    h->sink_valid = h->result_sink && !h->sink_invalidated;
    h->sink_delivered = false;
    return deferred();
}

static int solve_impl(corbo_hip_handle h, const corbo_hip_lm_opts* o, int new_run, bool async);

int corbo_hip_solve(corbo_hip_handle h, const corbo_hip_lm_opts* o, int new_run)
try {
    if (h) { const int rc = finish_async(h); if (rc) return rc; }
    return solve_impl(h, o, new_run, false);
}
ABI_CATCH

// The same solve without the wait: everything is enqueued on the handle's stream and the call returns -- the host side of the next solve (its re-arm,
// its launch) overlaps this one's kernel.  Run-to-completion handles only (one launch per solve); other handles solve synchronously.  Results, statistics
// and the pass-limit check become available with the next corbo_hip_synchronize / corbo_hip_solve / corbo_hip_get_* call.
int corbo_hip_solve_async(corbo_hip_handle h, const corbo_hip_lm_opts* o, int new_run)
try {
    return solve_impl(h, o, new_run, true);
}
ABI_CATCH

static int solve_impl(corbo_hip_handle h, const corbo_hip_lm_opts* o, int new_run, bool async)
{
    if (!h || !o) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    if (!h->have_data) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_set_instance_data must be called before corbo_hip_solve");
    if (h->S.desc.cost_nonlsq) return fail(CORBO_HIP_ERR_UNSUPPORTED, "not a least-squares problem (cost_nonlsq): LevenbergMarquardtSparse::solve refuses it too (levenberg_marquardt_sparse.cpp:48-55); the Hessian-path operators work on it");
    if (o->iterations < 0 || o->iterations > MAX_PASSES / 8) return fail(CORBO_HIP_ERR_INVALID, "iterations out of range");
    ON_DEVICE_OF(h);
    // new_run = 2: a new run from the iterates of the last corbo_hip_set_instance_data -- corbo_hip_restore_instance_data + new_run = 1 in one call; the
    // run-to-completion kernel reads its start from the shadow copy itself (no copy launch in front of it), other handles copy first
    const bool rearm = (new_run == 2);
    new_run = new_run ? 1 : 0;
    update_penalty_weights(h, o, new_run);
    h->stats = corbo_hip_stats{};
    h->sink_valid = false;
    h->sink_invalidated = false;   // (this solve's results are the newest thing the sink will hold)
    if (h->active == 0) return CORBO_HIP_OK;   // an empty bucket of an adaptive-grid batch
    EventList ev_list;
    std::vector<hipEvent_t>& evs = ev_list.v;
    auto stamp = [&]() {
        if (!h->profile) return;
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, h->stream);
        evs.push_back(e);
    };
    // (handles of the block-tridiagonal route have a run-to-completion kernel only: every per-pass mode runs their passes as separate launches, band kernels)
    const bool split = h->split_passes || h->force_split || (h->bt_rounds > 0 && !(h->loop_mode && o->iterations > 0));
    const bool run_to_completion = !split && h->loop_mode && o->iterations > 0;
    if (rearm && !run_to_completion) { launch_copy_rows(h->d_x0, h->d_x, nullptr, (size_t)h->batch * h->S.nvs, h->stream); HIP_TRY(hipGetLastError()); }
    if (async && !run_to_completion) { const int rc0 = finish_async(h); if (rc0) return rc0; async = false; }   // (host-driven passes: synchronous)
    // (measured and not kept: asynchronous run-to-completion solves delivering through the copy engine as well -- the kernel alone is 7.5 us shorter
    //  without its own stores into pinned host memory, but the cross-stream waits and the engine's traffic cost more: 0.543 -> 0.753 ms per step)
    std::pair<hipEvent_t, hipEvent_t> aev{nullptr, nullptr};
    if (async && h->solve_timing) {
        if (!h->event_pool.empty()) { aev = h->event_pool.back(); h->event_pool.pop_back(); }
        else { HIP_TRY(hipEventCreate(&aev.first)); HIP_TRY(hipEventCreate(&aev.second)); }
        HIP_TRY(hipEventRecord(aev.first, h->stream));
    }
    else if (h->solve_timing) HIP_TRY(hipEventRecord(h->ev0, h->stream));
    if (!run_to_completion)   // per-pass "unfinished instances" counters (the run-to-completion kernel reports through pinned host memory)
    { launch_zero_ints(h->d_counters, (size_t)corbo_hip_solver::MAX_SUB * MAX_PASSES, h->stream); HIP_TRY(hipGetLastError()); }
    stamp();
    // Launch structure.  Run-to-completion (default): ONE launch, every workgroup walks its instance through the prologue sweep
    // and [factor phase -> trial sweep phase] passes until the instance has finished (the instances are independent, nothing has to
    // meet at a grid-wide point).  Per-pass (CORBO_HIP_LOOP=0): every launch is [sweep phase -> factor phase] per instance; the first one runs the
    // prologue sweep (mode 2), the following ones the trial-step sweep (mode 3); an instance that finishes in its sweep phase
    // skips the factor phase.  Split (diagnostics / big-block family): the same phases as separate launches on one stream.
    // The batch is cut into `nsub` contiguous sub-batches, each driven on its own stream.
    // (the big-block family's separate launches on 2 / 4 sub-batch streams were measured: 17.4 / 21.6 ms against 17.1 ms on one; round 6, cfg 5 without the
    //  reject-streak speculation: 8.38 ms on two against 8.50 on one -- a stage kernel's 25 k waves leave the other half's chain nothing to overlap with)
    const int nsub = (split || run_to_completion) ? 1 : h->nsub;
    int pass_of[corbo_hip_solver::MAX_SUB] = {0, 0, 0, 0};
    int left_of[corbo_hip_solver::MAX_SUB] = {0, 0, 0, 0};
    int first_of[corbo_hip_solver::MAX_SUB], count_of[corbo_hip_solver::MAX_SUB];
    hipStream_t st_of[corbo_hip_solver::MAX_SUB];
    {
        const int base = h->active / nsub, rem = h->active % nsub;
        int f = 0;
        for (int i = 0; i < nsub; ++i) {
            count_of[i] = base + (i < rem ? 1 : 0);
            first_of[i] = f;
            f += count_of[i];
            st_of[i]   = (nsub == 1) ? h->stream : h->sub_stream[i];
            left_of[i] = (o->iterations > 0) ? count_of[i] : 0;
        }
        if (nsub > 1) {
            HIP_TRY(hipEventRecord(h->ev_chk[0], h->stream));  // the sub-streams start after everything queued on the main stream
            for (int i = 0; i < nsub; ++i) HIP_TRY(hipStreamWaitEvent(st_of[i], h->ev_chk[0], 0));
        }
    }
    int rc = 0;
    // reject-streak speculation (big-block family, stage / chain path; kernels.hpp SpecParams): the candidates live in the spare rows right behind
    // the batch, so the launches of a pass simply cover batch + spare rows -- possible when every row of the handle is in use and no per-instance table
    // (references per component, model parameters per instance) would have to follow the candidates
    // ... and when a pass is long against the one small launch per pass the bookkeeping costs (7 us: below 128 instances it is 4 - 7 % of a solve
    // and the laggers of a small batch hold back little); option reject_speculation: 0 = off, 1 = this rule, 2 = always, every streak (tests)
    const bool spec = split && h->spare > 0 && h->reject_speculation && (h->batch >= 128 || h->reject_speculation == 2) && h->band.n == 0 &&
                      h->active == h->batch && !h->refvec_on && !h->d_dyn_inst && !h->profile && o->iterations > 0;
    auto spec_params = [&](int mode, int32_t* counter) {
        SpecParams q{};
        q.mode = mode; q.batch = h->batch; q.groups = corbo_hip_solver::SPEC_GROUPS; q.spec = corbo_hip_solver::SPEC_SLOTS;
        q.nvs = h->S.nvs; q.m_pad = h->m_pad; q.xe_row = h->S.N * h->S.nx; q.batch_total = h->batch + h->spare;
        q.x = h->d_x; q.lb = h->d_lb; q.ub = h->d_ub; q.xref = h->d_xref; q.values0 = h->d_values0; q.values1 = h->d_values1; q.xe0 = h->d_xe0; q.chi2 = h->d_chi2;
        q.st = h->d_state; q.parent_of = h->d_spec_parent; q.rej_seen = h->d_spec_seen; q.slot_rej = h->d_spec_slotrej; q.prev_reject = h->d_spec_prev; q.adopted = h->d_spec_adopted;
        q.counter = counter;
        q.max_parents = (h->reject_speculation == 2) ? corbo_hip_solver::SPEC_GROUPS : 3;
        return q;
    };
    auto launch_one = [&](int i, int mode, int32_t* counter) -> int {
        FactorParams fp = h->factor_params();
        SweepParams sp  = h->sweep_params(mode, o->iterations, h->w_eq, h->w_ineq, h->w_b, counter);
        fp.batch = sp.batch = count_of[i] + ((spec && mode == 3) ? h->spare : 0);
        fp.inst0 = sp.inst0 = first_of[i];
        fp.pass_threads = h->pass_threads;
        if (split) {
            if (mode == 3) {
                fp.first_pass = (pass_of[i] == 0) ? 1 : 0;
                if (h->band.n > 0) { if (!launch_band_factor(fp, h->band, st_of[i])) return fail(CORBO_HIP_ERR_UNSUPPORTED, "band factorisation: half-bandwidth beyond 63 or more than 160 KB of LDS"); }   // integral-form constraint edges / control-deviation edges / a free dt around a big-block model
                else if (!launch_factor(h->S.desc, fp, st_of[i], &sp)) return fail(CORBO_HIP_ERR_UNSUPPORTED, "no factor kernel for this nx/nu/N");
                HIP_TRY(hipGetLastError());
                stamp();
            }
            if (!launch_sweep(h->S.desc, sp, st_of[i])) return fail(CORBO_HIP_ERR_UNSUPPORTED, "no sweep kernel for this dynamics/defect");
            HIP_TRY(hipGetLastError());
            if (spec) {   // after the prologue: every slot free; after a pass: candidates merged, the next ones started
                launch_big_spec(spec_params(mode == 3 ? 0 : 1, counter), st_of[i]);
                HIP_TRY(hipGetLastError());
            }
            stamp();
        }
        else {
            if (!launch_pass(h->S.desc, fp, sp, st_of[i])) return fail(CORBO_HIP_ERR_UNSUPPORTED, "no fused pass kernel for this descriptor");
            HIP_TRY(hipGetLastError());
        }
        return 0;
    };
    if (run_to_completion) {
        // one launch per sub-batch walks every instance through the prologue and all of its LM passes
        for (int i = 0; i < nsub; ++i) {
            FactorParams fp = h->factor_params();
            SweepParams sp  = h->sweep_params(2, o->iterations, h->w_eq, h->w_ineq, h->w_b, nullptr);
            if (rearm) sp.x_init = h->d_x0;
            fp.batch = sp.batch = count_of[i];
            fp.inst0 = sp.inst0 = first_of[i];
            fp.loop_passes = MAX_PASSES;
            fp.stagger = h->stagger;
            fp.pass_threads = h->pass_threads;
            if (h->lag_priority && !(count_of[i] > 4 * h->num_cus)) fp.cu_table = h->d_queue + 16;
            if (h->result_sink) { fp.x_host = h->h_stage; fp.st_host = h->h_state; }
            if (count_of[i] > 4 * h->num_cus) {   // more instances than resident workgroups: instance queue
                HIP_TRY(hipMemsetAsync(h->d_queue, 0, sizeof(int32_t), st_of[i]));
                fp.queue = h->d_queue; fp.queue_grid = h->num_cus;
            }
            if (h->pass_limit > 0 && h->pass_limit < MAX_PASSES) fp.loop_passes = h->pass_limit;   // (tests: provoke the "pass limit reached" error path)
            // an instance that runs into the pass limit raises a flag in pinned, device-visible host memory: no memset, no read-back
            // copy and no second synchronisation around the one launch of a solve
            // (asynchronous solves alternate between two flag slots: the flag of the solve still in flight is not cleared under it)
            const int flag_slot = async ? (h->async_pending & 1) : 0;
            if (!async || h->async_pending < 2) h->h_counter[2 * i + flag_slot] = 0;
            fp.unfinished_flag  = h->h_counter + 2 * i + flag_slot;
            if (h->phase_cycles) {
                if (!h->d_phase) HIP_TRY(hipMalloc((void**)&h->d_phase, (size_t)h->batch * 8 * sizeof(long long)));
                // (each sub-batch clears ITS rows on ITS stream: the sub-streams are not ordered against each other)
                HIP_TRY(hipMemsetAsync(h->d_phase + (size_t)first_of[i] * 8, 0, (size_t)count_of[i] * 8 * sizeof(long long), st_of[i]));
                fp.phase_cycles = h->d_phase;
            }
            long long* d_ptl = nullptr;  // CORBO_HIP_PASS_TIMELINE=<instance>: per-pass shader-clock stamps of that instance on stderr
            const bool ptl_on = h->pass_timeline_inst >= 0;
            if (ptl_on && i == 0) {
                HIP_TRY(hipMalloc((void**)&d_ptl, PTL_LEN * sizeof(long long)));
                HIP_TRY(hipMemsetAsync(d_ptl, 0, PTL_LEN * sizeof(long long), st_of[i]));
                fp.pass_timeline      = d_ptl;
                fp.timeline = d_ptl + 130;  // factor phases of that instance [130,138), sweep phases [138,148); copied into the per-pass log [150 + 18 pass, ...)
                sp.timeline = d_ptl + 138;
                fp.timeline_inst = sp.timeline_inst = h->pass_timeline_inst;
                fp.pass_timeline_inst = h->pass_timeline_inst;
            }
            struct PtlGuard { long long* p; hipStream_t s; bool raw; ~PtlGuard() { if (!p) return; (void)hipStreamSynchronize(s); static long long tl[PTL_LEN];
                if (hipMemcpy(tl, p, sizeof(tl), hipMemcpyDeviceToHost) == hipSuccess) {
                    // per-pass phase log: factor stamps F0..F7 (start, loaded, first-mu, controls, blocks, cyclic reduction, root/arrow, back-substitution, trial iterate)
                    // and sweep stamps S0..S9 of the sweep phase that STARTED the pass
                    for (int k = 0; k < 64 && tl[2 * k]; ++k) {
                        const long long* f = tl + 150 + 18 * k; const long long* w = f + 8;
                        if (!f[0] && !w[0]) continue;
                        if (raw) {   // (option raw_stamps: development builds that re-purpose the stamp slots, e.g. one stamp per cyclic-reduction level)
                            fprintf(stderr, "pass %2d raw:", k);
                            const long long base = f[0] ? f[0] : w[0];
                            for (int q = 0; q < 18; ++q) fprintf(stderr, " %lld", f[q] ? f[q] - base : -1);
                            fprintf(stderr, "\n");
                            continue;
                        }
                        fprintf(stderr, "pass %2d sweep:", k);
                        if (w[0]) { const int ids[] = {1, 2, 9, 3, 4, 5, 6, 7, 8}; long long prev = w[0]; for (int q : ids) { if (w[q]) { fprintf(stderr, " s%d+%lld", q, w[q] - prev); prev = w[q]; } } }
                        fprintf(stderr, " | factor:");
                        if (f[0]) { long long prev = f[0]; for (int q = 1; q < 8; ++q) if (f[q]) { fprintf(stderr, " f%d+%lld", q, f[q] - prev); prev = f[q]; } fprintf(stderr, " | sweep-end->factor-start %lld, pass-start->sweep-start %lld", f[0] - tl[2 * k + 1], w[0] ? w[0] - tl[2 * k] : -1); }
                        fprintf(stderr, "\n");
                    }
                    fprintf(stderr, "pass timeline (sweep/factor cycles):");
                    for (int k = 0; k < 64 && tl[2 * k]; ++k) fprintf(stderr, " %lld/%lld", tl[2 * k + 1] ? tl[2 * k + 1] - tl[2 * k] : -1, (k < 63 && tl[2 * k + 2]) ? tl[2 * k + 2] - tl[2 * k + 1] : 0);
                    fprintf(stderr, "\n");
                    if (tl[65] || tl[66]) { fprintf(stderr, "issue priority per pass (+10*slot):"); for (int k = 1; k < 32 && tl[2 * k]; ++k) fprintf(stderr, " %lld", tl[64 + k]); fprintf(stderr, "\n"); }
                    const long long* f = tl + 130;
                    if (f[7]) fprintf(stderr, "factor phases, last pass (cycles): load %lld | controls %lld | blocks + level 0 %lld | cyclic reduction %lld | root %lld | back-substitution %lld | trial iterate %lld\n",
                                      f[1] - f[0], f[2] - f[1], f[3] - f[2], f[4] - f[3], f[5] - f[4], f[6] - f[5], f[7] - f[6]); }
                (void)hipFree(p); } } ptl_guard{d_ptl, st_of[i], h->raw_stamps};
            if (!launch_pass(h->S.desc, fp, sp, st_of[i])) return fail(CORBO_HIP_ERR_UNSUPPORTED, "no fused pass kernel for this descriptor");
            HIP_TRY(hipGetLastError());
            pass_of[i] = 1;
        }
    }
    else
    for (int i = 0; i < nsub; ++i) {  // prologue
        rc = launch_one(i, 2, nullptr);
        if (rc) return rc;
    }
    auto enqueue_pass = [&](int i) -> int {
        if (pass_of[i] >= MAX_PASSES) return 0;
        int r = launch_one(i, 3, h->d_counters + (size_t)i * MAX_PASSES + pass_of[i]);
        ++pass_of[i];
        return r;
    };
    if (o->iterations > 0 && !run_to_completion) {
        // Every instance needs at least `iterations` passes; after that the host reads one "unfinished instances" counter per group
        // of passes and sub-batch, always with the NEXT group already enqueued, so the GPU never waits for the host; finished
        // instances make their workgroups exit at once, so an overshooting group costs a few microseconds.
        for (int p_ = 0; p_ < o->iterations; ++p_)
            for (int i = 0; i < nsub; ++i) { rc = enqueue_pass(i); if (rc) return rc; }
        // (big-block family: a pass is 0.2 - 0.8 ms, the read-back 30 us -- ONE pass ahead keeps the GPU busy just as well, and a batch that is
        //  through leaves one overshooting pass behind instead of three: 28 us each at cfg 5's size, the launches of four grids that exit at once)
        const int GROUP = big_family_dims(h->S.desc.nx, h->S.desc.nu) ? 1 : 2;
        int slot = 0;
        bool any = true;
        while (any) {
            for (int i = 0; i < nsub; ++i) {
                if (left_of[i] == 0) continue;
                HIP_TRY(hipMemcpyAsync(h->h_counter + 2 * i + slot, h->d_counters + (size_t)i * MAX_PASSES + (pass_of[i] - 1), sizeof(int32_t),
                                       hipMemcpyDeviceToHost, st_of[i]));
                HIP_TRY(hipEventRecord(h->sub_chk[i][slot], st_of[i]));
            }
            for (int g = 0; g < GROUP; ++g)
                for (int i = 0; i < nsub; ++i)
                    if (left_of[i] != 0) { rc = enqueue_pass(i); if (rc) return rc; }  // speculative
            any = false;
            for (int i = 0; i < nsub; ++i) {
                if (left_of[i] == 0) continue;
                HIP_TRY(hipEventSynchronize(h->sub_chk[i][slot]));
                left_of[i] = h->h_counter[2 * i + slot];
                if (left_of[i] != 0 && pass_of[i] < MAX_PASSES) any = true;
            }
            slot ^= 1;
        }
    }
    int pass = 0, remaining = 0;
    for (int i = 0; i < nsub; ++i) { if (pass_of[i] > pass) pass = pass_of[i]; remaining += left_of[i]; }
    if (nsub > 1)
        for (int i = 0; i < nsub; ++i) {  // the main stream continues after every sub-batch
            HIP_TRY(hipEventRecord(h->sub_done[i], st_of[i]));
            HIP_TRY(hipStreamWaitEvent(h->stream, h->sub_done[i], 0));
        }
    if (async) {   // no wait: finish_async does the rest
        if (aev.first) { HIP_TRY(hipEventRecord(aev.second, h->stream)); h->async_events.push_back(aev); }
        h->async_pending += 1;
        h->solve_count += 1;
        h->stats.passes = 1;
        return CORBO_HIP_OK;
    }
    if (h->solve_timing) {
        HIP_TRY(hipEventRecord(h->ev1, h->stream));
        HIP_TRY(hipEventSynchronize(h->ev1));
        HIP_TRY(hipEventElapsedTime(&h->stats.solve_ms, h->ev0, h->ev1));
    }
    else HIP_TRY(hipStreamSynchronize(h->stream));
    h->solve_ms_sum += h->stats.solve_ms;
    h->solve_count += 1;
    if (run_to_completion) {
        remaining = 0;
        for (int i = 0; i < nsub; ++i) remaining += h->h_counter[2 * i];
    }
    h->stats.passes = pass;
    if (h->profile && evs.size() >= 3) {
        // evs: [start, after init sweep, after factor, after sweep, after factor, ...]
        float ms = 0;
        (void)hipEventElapsedTime(&ms, evs[0], evs[1]);
        h->stats.sweep_ms += ms;
        for (size_t i = 2; i < evs.size(); ++i) {
            (void)hipEventElapsedTime(&ms, evs[i - 1], evs[i]);
            if (i % 2 == 0) h->stats.factor_ms += ms; else h->stats.sweep_ms += ms;
        }
    }
    if (remaining > 0) return fail(CORBO_HIP_ERR_DEVICE, "pass limit reached with unfinished instances");
    // (host-driven passes: a large batch is delivered by the copy engine on the second stream; a small one -- the drop-in adapter's single OCP -- is
    //  fetched by the copy kernels of corbo_hip_fetch_solution as before: a copy engine takes 0.1 - 0.3 ms to wake up, more than such a solve)
    const bool deliver = !run_to_completion && h->result_sink && h->active == h->batch && (size_t)h->batch * h->S.nvs * sizeof(double) >= ((size_t)1 << 20);
    h->sink_valid     = h->result_sink && (run_to_completion || deliver);
    h->sink_delivered = deliver;
    if (h->sink_delivered) { const int rc = deliver_results(h); if (rc) return rc; }   // (host-driven passes: delivered by a copy)
    return CORBO_HIP_OK;
}

int corbo_hip_set_result_sink(corbo_hip_handle h, int enable)
{
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    h->result_sink = enable != 0;
    h->sink_valid  = false;
    return CORBO_HIP_OK;
}

int corbo_hip_set_references(corbo_hip_handle h, const double* ref)
try {
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    h->sink_valid = false;
    h->ref_T = 0;   // (an explicit set of references ends the stepping of a resident reference trajectory)
    if (!ref) { h->refvec_on = false; return CORBO_HIP_OK; }   // back to the static state reference of corbo_hip_set_instance_data
    const Structure& S = h->S;
    const size_t B = (size_t)h->active, nv = (size_t)S.dims.nv, nvs = (size_t)S.nvs;
    if (!h->d_refvec) HIP_TRY(hipMalloc((void**)&h->d_refvec, (size_t)h->batch * nvs * sizeof(double)));
    std::vector<double> st(B * nvs, 0.0);
    for (size_t b = 0; b < B; ++b) {
        std::memcpy(&st[b * nvs], ref + b * nv, nv * sizeof(double));
        for (int k = 0; k < S.N - 1; ++k)
            for (int i = 0; i < S.nu; ++i)
                if (ref[b * nv + (size_t)k * S.s + S.nx + i] != 0.0)
                    return fail(CORBO_HIP_ERR_INVALID, "corbo_hip_set_references: non-zero control reference (the reference's least-squares control term is not defined for one, quadratic_cost.cpp:160-163)");
    }
    HIP_TRY(hipMemcpyAsync(h->d_refvec, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->refvec_on = true;
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_set_reference_trajectory(corbo_hip_handle h, const double* traj, int T, int step)
try {
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    h->sink_valid = false;
    const Structure& S = h->S;
    if (!traj) {   // back to the static reference
        if (h->d_reftraj) { (void)hipStreamSynchronize(h->stream); (void)hipFree(h->d_reftraj); h->d_reftraj = nullptr; }
        h->ref_T = 0; h->ref_step = 0; h->refvec_on = false;
        return CORBO_HIP_OK;
    }
    if (T < 1 || step < 0) return fail(CORBO_HIP_ERR_INVALID, "corbo_hip_set_reference_trajectory: T >= 1 and step >= 0");
    const size_t B = (size_t)h->batch, bytes = B * (size_t)T * S.nx * sizeof(double);
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->d_reftraj) { (void)hipFree(h->d_reftraj); h->d_reftraj = nullptr; }
    HIP_TRY(hipMalloc((void**)&h->d_reftraj, bytes));
    HIP_TRY(hipMemcpy(h->d_reftraj, traj, bytes, hipMemcpyHostToDevice));
    if (!h->d_refvec) HIP_TRY(hipMalloc((void**)&h->d_refvec, B * (size_t)S.nvs * sizeof(double)));
    HIP_TRY(hipMemsetAsync(h->d_refvec, 0, B * (size_t)S.nvs * sizeof(double), h->stream));   // control / dt entries: zero
    h->ref_T = T; h->ref_step = step;
    launch_reference_window(h->d_reftraj, h->d_refvec, h->batch, T, S.N, S.nx, S.s, S.nvs, step, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->refvec_on = true;
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_restore_instance_data(corbo_hip_handle h)
{
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    if (!h->have_data) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_set_instance_data must be called first");
    ON_DEVICE_OF(h);
    // (no drain: the copy is ordered behind the enqueued solves on the handle's stream, so re-arm + solve_async pipelines; the pinned views of those
    //  solves must not come back as "valid" when they are drained -- the device iterates are the re-armed ones by then)
    h->sink_valid = false;   // the pinned result views are stale from here on
    h->sink_invalidated = true;
    launch_copy_rows(h->d_x0, h->d_x, nullptr, (size_t)h->batch * h->S.nvs, h->stream);
    HIP_TRY(hipGetLastError());
    return CORBO_HIP_OK;
}

static int warm_start_from(corbo_hip_handle h, const double* x0_dev_visible, int shift)
{
    const Structure& S = h->S;
    WarmStartParams p{};
    p.batch = h->active; p.nvs = S.nvs; p.nx = S.nx; p.nu = S.nu; p.N = S.N;
    p.xf_fixed_mask = (int32_t)S.desc.xf_fixed_mask;
    p.shift = (shift != 0 && !S.dt_free) ? 1 : 0;   // variable grids never shift (finite_differences_variable_grid.h:77)
    p.x = h->d_x; p.x0new = x0_dev_visible; p.xref = h->d_xref;
    if (p.batch == 0) return CORBO_HIP_OK;
    launch_warm_start(p, h->stream);
    HIP_TRY(hipGetLastError());
    return CORBO_HIP_OK;
}

int corbo_hip_warm_start(corbo_hip_handle h, const double* x0_new, int shift)
{
    if (!h || !x0_new) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    if (!h->have_data) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_set_instance_data must be called first");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    h->sink_valid = false;   // the pinned result views are stale from here on
    const Structure& S = h->S;
    HIP_TRY(hipStreamSynchronize(h->stream));   // the pinned staging buffer of the previous call has been consumed
    for (int b = 0; b < h->batch; ++b)
        for (int i = 0; i < S.nx; ++i) h->h_xnew[(size_t)b * CORBO_HIP_MAX_NX + i] = x0_new[(size_t)b * S.nx + i];
    // the kernel reads the measured states straight from the pinned (device-visible) buffer: no copy engine in the control loop
    return warm_start_from(h, h->h_xnew, shift);
}

int corbo_hip_warm_start_from_plant(corbo_hip_handle h, int shift)
{
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    if (!h->have_data) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_set_instance_data must be called first");
    if (!h->have_plant) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_plant_set_state must be called first");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    h->sink_valid = false;   // the pinned result views are stale from here on
    return warm_start_from(h, h->d_xplant, shift);
}

int corbo_hip_plant_set_state(corbo_hip_handle h, const double* x)
{
    if (!h || !x) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    h->sink_valid = false;   // the pinned result views are stale from here on
    const Structure& S = h->S;
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int b = 0; b < h->batch; ++b)
        for (int i = 0; i < CORBO_HIP_MAX_NX; ++i) h->h_dist[(size_t)b * CORBO_HIP_MAX_NX + i] = (i < S.nx) ? x[(size_t)b * S.nx + i] : 0.0;
    HIP_TRY(hipMemcpyAsync(h->d_xplant, h->h_dist, (size_t)h->batch * CORBO_HIP_MAX_NX * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->have_plant = true;
    return CORBO_HIP_OK;
}

int corbo_hip_plant_step(corbo_hip_handle h, int integrator, double dt, const double* disturbance)
{
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    if (integrator != CORBO_HIP_INTEGRATOR_EULER && integrator != CORBO_HIP_INTEGRATOR_RK4) return fail(CORBO_HIP_ERR_INVALID, "unknown integrator");
    if (!(dt > 0)) return fail(CORBO_HIP_ERR_INVALID, "dt must be positive");
    if (!h->have_data) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_set_instance_data must be called first");
    if (!h->have_plant) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_plant_set_state must be called first");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    const Structure& S = h->S;
    if (disturbance) {
        HIP_TRY(hipStreamSynchronize(h->stream));   // the previous step has consumed the pinned buffer
        for (int b = 0; b < h->batch; ++b)
            for (int i = 0; i < S.nx; ++i) h->h_dist[(size_t)b * CORBO_HIP_MAX_NX + i] = disturbance[(size_t)b * S.nx + i];
    }
    PlantParams p{};
    p.batch = h->batch; p.nvs = S.nvs; p.nx = S.nx; p.nu = S.nu; p.integrator = integrator; p.dt = dt;
    h->fill_dyn(p.dyn);
    p.x = h->d_x; p.xplant = h->d_xplant; p.disturbance = disturbance ? h->h_dist : nullptr;
    p.dyn_inst = h->d_plant_prm ? h->d_plant_prm : h->d_dyn_inst;   // the plant's own parameters, else the controller's (per instance or the descriptor's)
    if (!launch_plant_step(S.desc, p, h->stream)) return fail(CORBO_HIP_ERR_UNSUPPORTED, "no plant kernel for this dynamics");
    HIP_TRY(hipGetLastError());
    return CORBO_HIP_OK;
}

int corbo_hip_plant_set_params(corbo_hip_handle h, const double* params)
try {
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!params) {   // the plants use the controller's model again
        if (h->d_plant_prm) { (void)hipFree(h->d_plant_prm); h->d_plant_prm = nullptr; }
        return CORBO_HIP_OK;
    }
    if (h->S.desc.dynamics == CORBO_HIP_DYN_LINEAR_STATE_SPACE)
        return fail(CORBO_HIP_ERR_UNSUPPORTED, "corbo_hip_plant_set_params: the linear state-space model keeps its matrices per handle");
    const size_t bytes = (size_t)h->batch * 8 * sizeof(double);
    if (!h->d_plant_prm) HIP_TRY(hipMalloc((void**)&h->d_plant_prm, bytes));
    HIP_TRY(hipMemcpy(h->d_plant_prm, params, bytes, hipMemcpyHostToDevice));
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_set_instance_params(corbo_hip_handle h, const double* params)
try {
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->sink_valid = false;
    if (!params) {   // every instance uses the descriptor's parameters again
        if (h->d_dyn_inst) { (void)hipFree(h->d_dyn_inst); h->d_dyn_inst = nullptr; }
        return CORBO_HIP_OK;
    }
    if (h->S.desc.dynamics == CORBO_HIP_DYN_LINEAR_STATE_SPACE)
        return fail(CORBO_HIP_ERR_UNSUPPORTED, "corbo_hip_set_instance_params: the linear state-space model keeps its matrices per handle");
    for (size_t i = 0; i < (size_t)h->batch * 8; ++i)
        if (!std::isfinite(params[i])) return fail(CORBO_HIP_ERR_INVALID, "corbo_hip_set_instance_params: non-finite parameter");
    const size_t bytes = (size_t)h->batch * 8 * sizeof(double);
    if (!h->d_dyn_inst) HIP_TRY(hipMalloc((void**)&h->d_dyn_inst, bytes));
    std::vector<double> prm(params, params + (size_t)h->batch * 8);
    for (int b = 0; b < h->batch; ++b) prm[(size_t)b * 8 + 7] = (double)h->S.desc.shooting_integrator;   // slot 7 is the library's (fill_dyn)
    HIP_TRY(hipMemcpy(h->d_dyn_inst, prm.data(), bytes, hipMemcpyHostToDevice));
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_plant_get_state(corbo_hip_handle h, double* x_out)
{
    if (!h || !x_out) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    if (!h->have_plant) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_plant_set_state must be called first");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    h->sink_valid = false;   // the pinned result views are stale from here on
    const Structure& S = h->S;
    HIP_TRY(hipMemcpyAsync(h->h_stage, h->d_xplant, (size_t)h->batch * CORBO_HIP_MAX_NX * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int b = 0; b < h->batch; ++b)
        for (int i = 0; i < S.nx; ++i) x_out[(size_t)b * S.nx + i] = h->h_stage[(size_t)b * CORBO_HIP_MAX_NX + i];
    return CORBO_HIP_OK;
}

int corbo_hip_closed_loop(corbo_hip_handle h, const corbo_hip_lm_opts* o, int steps, int ocp_iterations, int integrator, double dt, int shift,
                          const double* disturbance, double* states_out, double* controls_out)
try {
    if (!h || !o || steps < 0 || ocp_iterations < 1) return fail(CORBO_HIP_ERR_INVALID, "bad argument");
    if (h->S.desc.cost_nonlsq) return fail(CORBO_HIP_ERR_UNSUPPORTED, "not a least-squares problem (cost_nonlsq): LevenbergMarquardtSparse::solve refuses it too (levenberg_marquardt_sparse.cpp:48-55); the Hessian-path operators work on it");
    if (integrator != CORBO_HIP_INTEGRATOR_EULER && integrator != CORBO_HIP_INTEGRATOR_RK4) return fail(CORBO_HIP_ERR_INVALID, "unknown integrator");
    if (!(dt > 0)) return fail(CORBO_HIP_ERR_INVALID, "dt must be positive");
    if (o->iterations < 0 || o->iterations > MAX_PASSES / 8) return fail(CORBO_HIP_ERR_INVALID, "iterations out of range");
    if (!h->have_data) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_set_instance_data must be called first");
    if (!h->have_plant) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_plant_set_state must be called first");
    if (steps == 0) return CORBO_HIP_OK;
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    h->sink_valid = false;   // the pinned result views are stale from here on
    const Structure& S = h->S;
    const size_t B = (size_t)h->batch, NXm = CORBO_HIP_MAX_NX;
    const size_t dist_doubles = disturbance ? (size_t)steps * B * NXm : 0;
    const size_t logx = states_out ? (size_t)steps * B * S.nx : 0, logu = controls_out ? (size_t)steps * B * S.nu : 0;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t need_h = dist_doubles > logx + logu ? dist_doubles : logx + logu;
    if (need_h > h->h_loop_doubles) {
        if (h->h_loop) (void)hipHostFree(h->h_loop);
        h->h_loop = nullptr; h->h_loop_doubles = 0;
        HIP_TRY(hipHostMalloc((void**)&h->h_loop, need_h * sizeof(double)));
        h->h_loop_doubles = need_h;
    }
    if (logx + logu > h->d_loop_doubles) {
        if (h->d_loop) (void)hipFree(h->d_loop);
        h->d_loop = nullptr; h->d_loop_doubles = 0;
        HIP_TRY(hipMalloc((void**)&h->d_loop, (logx + logu) * sizeof(double)));
        h->d_loop_doubles = logx + logu;
    }
    if (disturbance)
        for (size_t r = 0; r < (size_t)steps * B; ++r)
            for (size_t i = 0; i < NXm; ++i) h->h_loop[r * NXm + i] = (i < (size_t)S.nx) ? disturbance[r * S.nx + i] : 0.0;
    double* d_logx = logx ? h->d_loop : nullptr;
    double* d_logu = logu ? h->d_loop + logx : nullptr;

    const bool split = h->split_passes || h->force_split;
    const bool rtc   = !split && h->loop_mode && o->iterations > 0;   // run-to-completion solve kernel: the whole loop is asynchronous
    const int limit = (h->pass_limit > 0 && h->pass_limit < MAX_PASSES) ? h->pass_limit : MAX_PASSES;
    h->h_counter[0] = 0;
    float nested_ms = 0.0f;
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    for (int s = 0; s < steps; ++s) {
        PlantParams pp{};
        pp.batch = h->batch; pp.nvs = S.nvs; pp.nx = S.nx; pp.nu = S.nu; pp.integrator = integrator; pp.dt = dt;
        h->fill_dyn(pp.dyn);
        pp.x = h->d_x; pp.xplant = h->d_xplant; pp.dyn_inst = h->d_plant_prm ? h->d_plant_prm : h->d_dyn_inst;
        pp.disturbance = disturbance ? h->h_loop + (size_t)s * B * NXm : nullptr;   // read by the kernel from pinned host memory
        pp.log_x = d_logx ? d_logx + (size_t)s * B * S.nx : nullptr;
        pp.log_u = d_logu ? d_logu + (size_t)s * B * S.nu : nullptr;
        if (!launch_plant_step(S.desc, pp, h->stream)) return fail(CORBO_HIP_ERR_UNSUPPORTED, "no plant kernel for this dynamics");
        HIP_TRY(hipGetLastError());
        if (int rc = warm_start_from(h, h->d_xplant, shift)) return rc;
        if (h->d_reftraj && h->ref_T > 0) {   // tracking: the next control step sees the reference trajectory one sample further on
            ++h->ref_step;
            launch_reference_window(h->d_reftraj, h->d_refvec, h->batch, h->ref_T, S.N, S.nx, S.s, S.nvs, h->ref_step, h->stream);
            HIP_TRY(hipGetLastError());
        }
        for (int it = 0; it < ocp_iterations; ++it) {
            const int new_run = (it == 0) ? 1 : 0;
            if (!rtc) {
                if (int rc = corbo_hip_solve(h, o, new_run)) return rc;
                nested_ms += h->stats.solve_ms;   // (each nested solve re-records ev0 / ev1: the loop's time is the sum)
                continue;
            }
            update_penalty_weights(h, o, new_run);   // host-side state only
            FactorParams fp = h->factor_params();
            SweepParams sp  = h->sweep_params(2, o->iterations, h->w_eq, h->w_ineq, h->w_b, nullptr);
            fp.loop_passes     = limit;
            fp.unfinished_flag = h->h_counter;   // any step's unfinished instance raises it
            if (h->batch > 4 * h->num_cus) {
                HIP_TRY(hipMemsetAsync(h->d_queue, 0, sizeof(int32_t), h->stream));
                fp.queue = h->d_queue; fp.queue_grid = h->num_cus;
            }
            if (!launch_pass(S.desc, fp, sp, h->stream)) return fail(CORBO_HIP_ERR_UNSUPPORTED, "no fused pass kernel for this descriptor");
            HIP_TRY(hipGetLastError());
        }
    }
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    if (logx + logu) HIP_TRY(hipMemcpyAsync(h->h_loop, h->d_loop, (logx + logu) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (rtc) HIP_TRY(hipEventElapsedTime(&h->stats.solve_ms, h->ev0, h->ev1));   // the whole loop
    else h->stats.solve_ms = nested_ms;                                          // (sum of the solves; plant / grid-update kernels are microseconds)
    if (logx) std::memcpy(states_out, h->h_loop, logx * sizeof(double));
    if (logu) std::memcpy(controls_out, h->h_loop + logx, logu * sizeof(double));
    if (rtc && h->h_counter[0] != 0) return fail(CORBO_HIP_ERR_DEVICE, "pass limit reached with unfinished instances");
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_get_first_control(corbo_hip_handle h, double* u0_out)
{
    if (!h || !u0_out) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    if (!h->have_data) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_set_instance_data must be called first");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    h->sink_valid = false;   // the pinned result views are stale from here on
    const Structure& S = h->S;
    // a small kernel packs u_0 of every instance straight into the pinned (device-visible) staging buffer: no copy engine in the
    // loop of a predictive controller (waking the idle DMA engine for 16 KB cost ~0.1 ms per step)
    launch_gather_first_control(h->d_x, h->h_stage, S.nvs, S.nx, S.nu, h->batch, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::memcpy(u0_out, h->h_stage, (size_t)h->batch * S.nu * sizeof(double));
    return CORBO_HIP_OK;
}

int corbo_hip_device_count(int* count)
{
    if (!count) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return CORBO_HIP_OK;
}

int corbo_hip_shard_bounds(int global_batch, int world, int rank, int* first, int* count)
{
    if (!first || !count || world < 1 || rank < 0 || rank >= world || global_batch < 0) return fail(CORBO_HIP_ERR_INVALID, "bad argument");
    const int base = global_batch / world, rem = global_batch % world;
    *count = base + (rank < rem ? 1 : 0);
    *first = rank * base + (rank < rem ? rank : rem);
    return CORBO_HIP_OK;
}

int corbo_hip_prepare_slots(corbo_hip_handle h, int active)
{
    if (!h || active < 0 || active > h->batch) return fail(CORBO_HIP_ERR_INVALID, "bad argument");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    h->sink_valid = false;
    const Structure& S = h->S;
    if (!h->have_data) {   // never uploaded: zero iterates (create cleared them), the descriptor's bound pattern in every slot
        launch_broadcast_rows(h->d_bound_rows, h->d_bound_rows + S.nvs, h->d_lb, h->d_ub, S.nvs, h->batch, h->stream);
        HIP_TRY(hipGetLastError());
        h->have_data = true;
    }
    h->active = active;
    return CORBO_HIP_OK;
}

int corbo_hip_get_dt(corbo_hip_handle h, double* dt_out)
{
    if (!h || !dt_out) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    if (!h->have_data) return fail(CORBO_HIP_ERR_STATE, "no instance data");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    h->sink_valid = false;
    if (h->active == 0) return CORBO_HIP_OK;
    launch_gather_dt(h->d_x, h->h_stage, h->S.nvs, h->S.off_dt, h->active, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::memcpy(dt_out, h->h_stage, (size_t)h->active * sizeof(double));
    return CORBO_HIP_OK;
}

int corbo_hip_resample_into(corbo_hip_handle src, corbo_hip_handle dst, int count, const int32_t* src_index, const int32_t* dst_index)
try {
    if (!src || !dst || count < 0 || (count > 0 && (!src_index || !dst_index))) return fail(CORBO_HIP_ERR_INVALID, "bad argument");
    if (count == 0) return CORBO_HIP_OK;
    // every destination slot is written at most once: more pairs than slots means duplicates, and the two index lists are staged in a
    // buffer of dst->batch rows
    if (count > dst->batch) return fail(CORBO_HIP_ERR_INVALID, "more index pairs than destination slots");
    const Structure &A = src->S, &B = dst->S;
    if (!A.dt_free || !B.dt_free) return fail(CORBO_HIP_ERR_INVALID, "resampling needs free-dt grids on both sides");
    if (A.nx != B.nx || A.nu != B.nu || src->device != dst->device) return fail(CORBO_HIP_ERR_INVALID, "handles do not belong to the same problem family / device");
    if (!src->have_data || !dst->have_data) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_set_instance_data / corbo_hip_prepare_slots must be called on both handles first");
    for (int q = 0; q < count; ++q)
        if (src_index[q] < 0 || src_index[q] >= src->batch || dst_index[q] < 0 || dst_index[q] >= dst->batch) return fail(CORBO_HIP_ERR_INVALID, "instance index out of range");
    ON_DEVICE_OF(dst);
    DRAIN_ASYNC(src);
    DRAIN_ASYNC(dst);
    src->sink_valid = dst->sink_valid = false;
    // the index lists travel through the destination handle's pinned staging buffer (batch x nvs doubles: 2 x count ints always fit; the
    // result views it may have held were invalidated above) -- no allocation on this path
    HIP_TRY(hipStreamSynchronize(dst->stream));   // nothing in flight reads the staging buffer
    struct { int32_t* p; } tmp{reinterpret_cast<int32_t*>(dst->h_stage)};
    std::memcpy(tmp.p, src_index, (size_t)count * sizeof(int32_t));
    std::memcpy(tmp.p + count, dst_index, (size_t)count * sizeof(int32_t));
    HIP_TRY(hipStreamSynchronize(src->stream));   // the source trajectories are final
    ResampleParams p{};
    p.pairs = count; p.nx = A.nx; p.nu = A.nu;
    p.n_src = A.N; p.nvs_src = A.nvs; p.n_dst = B.N; p.nvs_dst = B.nvs;
    p.x_src = src->d_x; p.x_dst = dst->d_x; p.xref_src = src->d_xref; p.xref_dst = dst->d_xref;
    p.src_index = tmp.p; p.dst_index = tmp.p + count;
    launch_resample(p, dst->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(dst->stream));
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_set_option(corbo_hip_handle h, const char* name, int value)
{
    if (!h || !name) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    const std::string n(name);
    if (n == "pass_limit") h->pass_limit = value;
    else if (n == "run_to_completion") h->loop_mode = value != 0;
    else if (n == "pass_timeline") h->pass_timeline_inst = value;
    else if (n == "sweep_timeline") h->sweep_timeline = value != 0;
    else if (n == "chain_variant") h->chain_variant = value;
    else if (n == "band_wide") h->band_wide = value;
    else if (n == "hess_split") h->hess_split = value;
    else if (n == "bt_waves") h->bt_waves = (value == 2 || value == 3) ? value : 0;
    else if (n == "reject_speculation") h->reject_speculation = value;
    else if (n == "stagger") h->stagger = value;
    else if (n == "pass_threads") h->pass_threads = value;
    else if (n == "lag_priority") h->lag_priority = value;
    else if (n == "raw_stamps") h->raw_stamps = value != 0;
    else if (n == "phase_cycles") h->phase_cycles = value != 0;
    else if (n == "ff_converged") h->ff_converged = value;
    else if (n == "solve_timing") h->solve_timing = value;
    else return fail(CORBO_HIP_ERR_INVALID, "unknown option: " + n);
    return CORBO_HIP_OK;
}

int corbo_hip_set_profiling(corbo_hip_handle h, int enable)
{
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    h->profile      = enable != 0;
    h->split_passes = enable != 0;
    return CORBO_HIP_OK;
}

int corbo_hip_synchronize(corbo_hip_handle h)
{
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    ON_DEVICE_OF(h);
    { const int rc = finish_async(h); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return wait_sink(h);
}

int corbo_hip_get_solution(corbo_hip_handle h, double* x_out, double* chi2_out, int32_t* status_out)
try {
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    ON_DEVICE_OF(h);
    { const int rc_a = finish_async(h); if (rc_a) return rc_a; }
    h->sink_valid = false;   // the pinned result views are stale from here on
    HIP_TRY(hipStreamSynchronize(h->stream));
    const Structure& S = h->S;
    const int B = h->batch;
    // both arrays are written into the pinned buffers by copy kernels on the handle's stream (posted writes over PCIe, no copy engine:
    // 30 -> 15 us for one OCP), one synchronisation
    if (x_out) launch_copy_rows(h->d_x, h->h_stage, nullptr, (size_t)B * S.nvs, h->stream);
    if (chi2_out || status_out) launch_copy_rows(reinterpret_cast<const double*>(h->d_state), reinterpret_cast<double*>(h->h_state), nullptr, (size_t)B * sizeof(LmState) / sizeof(double), h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (x_out)
        for (int b = 0; b < B; ++b) std::memcpy(x_out + (size_t)b * S.dims.nv, h->h_stage + (size_t)b * S.nvs, S.dims.nv * sizeof(double));
    if (chi2_out || status_out) {
        const LmState* st = h->h_state;
        for (int b = 0; b < B; ++b) {
            if (chi2_out) chi2_out[b] = st[b].chi2_old;
            if (status_out) status_out[b] = st[b].status;
        }
    }
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_fetch_solution(corbo_hip_handle h, const double** x_pinned, int32_t* x_row_stride, const double** chi2_pinned,
                             const int32_t** status_pinned)
try {
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    ON_DEVICE_OF(h);
    { const int rc_a = finish_async(h); if (rc_a) return rc_a; }
    const Structure& S = h->S;
    const int B = h->batch;
    if (!h->sink_valid) {
        // both copies are queued behind the solve on the handle's stream; one synchronisation
        if (x_pinned) launch_copy_rows(h->d_x, h->h_stage, nullptr, (size_t)B * S.nvs, h->stream);
        if (chi2_pinned || status_pinned)
            launch_copy_rows(reinterpret_cast<const double*>(h->d_state), reinterpret_cast<double*>(h->h_state), nullptr, (size_t)B * sizeof(LmState) / sizeof(double), h->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
    }   // else: the solve kernel has written both itself and corbo_hip_solve has waited for it -- or (host-driven passes) the delivery behind the solve has
    const bool delivered = h->sink_valid && h->sink_delivered;
    if (delivered) { const int rc_s = wait_sink(h); if (rc_s) return rc_s; }
    const LmState* hstate = delivered ? reinterpret_cast<const LmState*>(h->h_sink + (size_t)B * S.nvs) : h->h_state;
    if (chi2_pinned || status_pinned) {
        int32_t* hs = reinterpret_cast<int32_t*>(h->h_chi2 + B);
        for (int b = 0; b < B; ++b) { h->h_chi2[b] = hstate[b].chi2_old; hs[b] = hstate[b].status; }
        if (chi2_pinned) *chi2_pinned = h->h_chi2;
        if (status_pinned) *status_pinned = hs;
    }
    if (x_pinned) *x_pinned = delivered ? h->h_sink : h->h_stage;
    if (x_row_stride) *x_row_stride = S.nvs;
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_get_phase_cycles(corbo_hip_handle h, int64_t* out)
{
    if (!h || !out) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    if (!h->d_phase) return fail(CORBO_HIP_ERR_STATE, "no phase totals: set option \"phase_cycles\" and solve (run-to-completion handles)");
    HIP_TRY(hipStreamSynchronize(h->stream));
    static_assert(sizeof(long long) == sizeof(int64_t), "phase totals are 64-bit");
    HIP_TRY(hipMemcpy(out, h->d_phase, (size_t)h->batch * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    return CORBO_HIP_OK;
}

int corbo_hip_get_timing(corbo_hip_handle h, double* solve_ms_sum, int64_t* solves, int reset)
{
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    { ON_DEVICE_OF(h); const int rc = finish_async(h); if (rc) return rc; }
    if (solve_ms_sum) *solve_ms_sum = h->solve_ms_sum;
    if (solves) *solves = h->solve_count;
    if (reset) { h->solve_ms_sum = 0.0; h->solve_count = 0; }
    return CORBO_HIP_OK;
}

int corbo_hip_factor_route(corbo_hip_handle h)
{
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    if (h->bt_rounds > 0) return CORBO_HIP_FACTOR_BLOCK_TRI;
    if (h->band.n > 0) return CORBO_HIP_FACTOR_BAND;
    return big_family_dims(h->S.nx, h->S.nu) ? CORBO_HIP_FACTOR_STAGE_CHAIN : CORBO_HIP_FACTOR_STAGE_CR;
}

int corbo_hip_get_stats(corbo_hip_handle h, corbo_hip_stats* stats)
try {
    if (!h || !stats) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    ON_DEVICE_OF(h);
    { const int rc_a = finish_async(h); if (rc_a) return rc_a; }
    h->sink_valid = false;   // the pinned result views are stale from here on
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpyAsync(h->h_state, h->d_state, (size_t)h->batch * sizeof(LmState), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    corbo_hip_stats s = h->stats;
    s.lm_iterations = s.accepted_steps = s.rejected_steps = s.jacobian_sweeps = s.residual_sweeps = s.factorizations = 0;
    s.inner_loop_cuts = 0;
    s.counted_iterations = 0;
    int max_fact = 0;
    for (int b = 0; b < h->active; ++b) {
        const LmState& a = h->h_state[b];
        if (a.n_fact > max_fact) max_fact = a.n_fact;
        s.lm_iterations += a.k;
        s.accepted_steps += a.n_accept;
        s.rejected_steps += a.n_reject;
        s.jacobian_sweeps += a.n_jac;
        s.residual_sweeps += a.n_res;
        s.factorizations += a.n_fact;
        s.inner_loop_cuts += a.pad[0];
        s.counted_iterations += a.pad[1];
    }
    s.passes = max_fact;  // inner passes of the slowest instance
    s.speculative_takeovers = 0;
    if (h->d_spec_adopted) { int32_t a = 0; HIP_TRY(hipMemcpy(&a, h->d_spec_adopted, sizeof(a), hipMemcpyDeviceToHost)); s.speculative_takeovers = a; }
    *stats = s;
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_eval(corbo_hip_handle h, double w_eq, double w_ineq, double w_bounds, double* values_out, double* jac_out)
try {
    if (!h || !values_out) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    if (h->S.desc.cost_nonlsq) return fail(CORBO_HIP_ERR_UNSUPPORTED, "not a least-squares problem (cost_nonlsq): LevenbergMarquardtSparse::solve refuses it too (levenberg_marquardt_sparse.cpp:48-55); the Hessian-path operators work on it");
    if (!h->have_data) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_set_instance_data must be called first");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    h->sink_valid = false;   // the pinned result views are stale from here on
    // (Runge-Kutta 5 - 7 around a big-block model: only the residual-only sweep exists; the Jacobian is the stage kernel's, below)
    const bool high_big = big_family_dims(h->S.nx, h->S.nu) && h->S.desc.shooting_integrator >= 5;
    if (jac_out && high_big && (h->band.n != 0 || h->S.desc.final_eq_mask))
        return fail(CORBO_HIP_ERR_UNSUPPORTED, "corbo_hip_eval with a Jacobian: Runge-Kutta 5 - 7 around a big-block model with a partial terminal equality");
    const SweepParams spe = h->sweep_params((jac_out && !high_big) ? 1 : 0, 0, w_eq, w_ineq, w_bounds, nullptr);
    int rc = launch_sweep_checked(h, spe);
    if (rc) return rc;
    // (a partial terminal equality: the stage kernel's dump writes the full constraint's block -- kernels.hip, big_stage_edges --, the sweep kernel's Jacobian is returned)
    if (jac_out && h->d_xe0 && h->band.n == 0 && !h->S.desc.final_eq_mask) {   // (band route: the sweep's stored Jacobian IS what the factorisation reads)
        // big-block family: what an LM pass differentiates is the stage kernel's Jacobian (never stored during a solve); the parity hook
        // returns THAT one -- every value is overwritten (an entry the stage kernel does not produce would come back as NaN)
        HIP_TRY(hipMemsetAsync(h->d_jac, 0xFF, (size_t)h->batch * h->nnz_pad * sizeof(double), h->stream));
        if (!launch_stage_jacobian_dump(h->S.desc, h->factor_params(), spe, h->d_jac, h->stream))
            return fail(CORBO_HIP_ERR_UNSUPPORTED, "no stage kernel for this dynamics");
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    const Structure& S = h->S;
    const int B = h->batch;
    std::vector<double> buf((size_t)B * h->m_pad);
    HIP_TRY(hipMemcpy(buf.data(), h->d_values0, buf.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) std::memcpy(values_out + (size_t)b * S.dims.m, &buf[(size_t)b * h->m_pad], S.dims.m * sizeof(double));
    if (jac_out) {
        buf.resize((size_t)B * h->nnz_pad);
        HIP_TRY(hipMemcpy(buf.data(), h->d_jac, buf.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) {   // device-internal layout -> public value order
            const double* src = &buf[(size_t)b * h->nnz_pad];
            double* dst       = jac_out + (size_t)b * S.dims.nnz;
            for (int i = 0; i < S.dims.nnz; ++i) dst[i] = src[h->jmap[i]];
        }
    }
    return CORBO_HIP_OK;
}
ABI_CATCH

// ---- operators of the exact-Hessian path ----------------------------------------------------------------------
namespace {
struct DevBuf {   // scratch device buffer of one call
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    double* d() const { return static_cast<double*>(p); }
    int32_t* i() const { return static_cast<int32_t*>(p); }
};
}  // namespace

int corbo_hip_hessian_nnz(const corbo_hip_problem_desc* desc, int lower_part_only, int32_t* nnz_out)
try {
    if (!desc || !nnz_out) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    if (desc->stage_ineq_integral || desc->stage_eq || desc->ctrl_dev) return fail(CORBO_HIP_ERR_UNSUPPORTED, "Hessian-path operators with integral-form constraints / a control-deviation term: not built");
    Structure S;
    std::string err = build_structure(*desc, S);
    if (!err.empty()) return fail(CORBO_HIP_ERR_INVALID, err);
    HessianStructure H;
    build_hessian_structure(S, lower_part_only != 0, H);
    for (int c = 0; c < 3; ++c) nnz_out[c] = H.nnz[c];
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_hessian_structure(const corbo_hip_problem_desc* desc, int lower_part_only, int32_t* rows_obj, int32_t* cols_obj, int32_t* rows_eq,
                                int32_t* cols_eq, int32_t* rows_ineq, int32_t* cols_ineq)
try {
    if (!desc) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    Structure S;
    std::string err = build_structure(*desc, S);
    if (!err.empty()) return fail(CORBO_HIP_ERR_INVALID, err);
    HessianStructure H;
    build_hessian_structure(S, lower_part_only != 0, H);
    int32_t* rows[3] = {rows_obj, rows_eq, rows_ineq};
    int32_t* cols[3] = {cols_obj, cols_eq, cols_ineq};
    for (int c = 0; c < 3; ++c) {
        if (H.nnz[c] > 0 && (!rows[c] || !cols[c])) return fail(CORBO_HIP_ERR_INVALID, "null structure array for a non-empty list");
        if (H.nnz[c] > 0) { std::memcpy(rows[c], H.rows[c].data(), H.nnz[c] * sizeof(int32_t)); std::memcpy(cols[c], H.cols[c].data(), H.nnz[c] * sizeof(int32_t)); }
    }
    return CORBO_HIP_OK;
}
ABI_CATCH

static int hessian_common(corbo_hip_handle h, const HessianStructure*& Hout, bool lower, HessParams& hp)
{
    if (!h->have_data) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_set_instance_data must be called first");
    if (h->S.has_extra()) return fail(CORBO_HIP_ERR_UNSUPPORTED, "Hessian-path operators with integral-form constraints / a control-deviation term: not built");
    auto& c = h->hess_cache[lower ? 1 : 0];
    if (!c.valid) {   // once per (handle, lower): the walk over the edges and its two device tables
        build_hessian_structure(h->S, lower, c.H);
        HIP_TRY(c.d_so.need(c.H.stage_off.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(c.d_so.p, c.H.stage_off.data(), c.H.stage_off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(c.d_lo.need(c.H.lin_off.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(c.d_lo.p, c.H.lin_off.data(), c.H.lin_off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        c.valid = true;
    }
    const HessianStructure& H = c.H;
    Hout = &H;
    hp = HessParams{};
    hp.lower = lower ? 1 : 0;
    hp.eq_dim = h->S.dims.eq; hp.ineq_dim = h->S.dims.ineq;
    hp.stage_off = static_cast<const int32_t*>(c.d_so.p); hp.lin_off = static_cast<const int32_t*>(c.d_lo.p);
    for (int q = 0; q < 3; ++q) hp.nnz[q] = H.nnz[q];
    hp.lin_nnz = H.lin_nnz; hp.lin_bounds0 = H.lin_bounds0; hp.bnd_row0 = h->S.bnd_row0; hp.n_bounds = h->S.dims.bounds;
    hp.stage_cost = h->S.desc.stage_cost; hp.stage_ineq = h->S.desc.stage_ineq;
    hp.dt_cost_off = H.dt_cost_off; hp.quad_first_interval = h->S.desc.quad_first_interval; hp.cost_nonlsq = h->S.desc.cost_nonlsq; hp.cost_integral = h->S.desc.cost_integral;
    hp.ms_mixed = (h->S.desc.cost_integral && h->S.desc.grid == CORBO_HIP_GRID_MS) ? 1 : 0;
    return 0;
}

// device -> pinned host: a copy kernel for what a kernel moves faster than a copy engine wakes up (<= 8 MB), the copy engine above
static int copy_to_pinned(corbo_hip_handle h, const double* src_dev, double* dst_pin, size_t doubles)
{
    if (doubles * sizeof(double) <= ((size_t)8 << 20)) { launch_copy_rows(src_dev, dst_pin, nullptr, doubles, h->stream); HIP_TRY(hipGetLastError()); }
    else HIP_TRY(hipMemcpyAsync(dst_pin, src_dev, doubles * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return 0;
}

// multipliers of the caller (pageable host memory) -> device, through the pinned staging buffer and a copy kernel (no copy engine)
static int upload_through_pin(corbo_hip_handle h, const double* src, size_t doubles, corbo_hip_solver::GrowBuf& dev)
{
    const size_t even = (doubles + 1) & ~(size_t)1;
    HIP_TRY(dev.need(even * sizeof(double)));
    HIP_TRY(h->hb_pin.need(even * sizeof(double)));
    HIP_TRY(hipStreamSynchronize(h->stream));   // the staging buffer is free
    std::memcpy(h->hb_pin.p, src, doubles * sizeof(double));
    launch_copy_rows(h->hb_pin.d(), dev.d(), nullptr, even, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// the three value lists on the device (hb_vals); `where`: 0 = leave them there, 1 = also into the pinned buffer [obj | eq | ineq]
static int eval_hessians_device(corbo_hip_handle h, int lower_part_only, double mult_obj, const double* mult_eq, const double* mult_ineq, const HessianStructure*& H,
                                size_t off[4], bool to_pinned)
{
    HessParams hp;
    int rc = hessian_common(h, H, lower_part_only != 0, hp);
    if (rc) return rc;
    const size_t B = (size_t)h->active;
    off[0] = 0;
    for (int c = 0; c < 3; ++c) {
        const size_t n = (B * H->nnz[c] + 1) & ~(size_t)1;
        HIP_TRY(h->hb_vals[c].need(n * sizeof(double)));
        hp.vals[c] = h->hb_vals[c].d();
        off[c + 1] = off[c] + n;
    }
    if (mult_eq && hp.eq_dim > 0) { rc = upload_through_pin(h, mult_eq, B * hp.eq_dim, h->hb_me); if (rc) return rc; hp.mult_eq = h->hb_me.d(); }
    if (mult_ineq && hp.ineq_dim > 0) { rc = upload_through_pin(h, mult_ineq, B * hp.ineq_dim, h->hb_mi); if (rc) return rc; hp.mult_ineq = h->hb_mi.d(); }
    hp.mode = 0;
    hp.mult_obj = mult_obj;
    {   // how the blocks of a stage are spread over waves (hessian_kernel): few instances -> by block, a batch that fills the chip anyway -> not at all
        const long waves = (long)((h->S.N + 63) / 64) * h->active;
        hp.split = h->hess_split >= 0 ? h->hess_split : (waves > 768 ? 0 : (waves > 256 ? 1 : 2));   // (measured, unicycle N = 100: one OCP 171 -> 86 -> 43 us with split 0 / 1 / 2, 32 OCPs 173 / 89 / 66, 1024 OCPs 284 / 380 / 838)
    }
    const SweepParams sp = h->sweep_params(0, 0, 1.0, 1.0, 1.0, nullptr);
    if (to_pinned) HIP_TRY(h->hb_pin.need(off[3] * sizeof(double)));
    if (!launch_hessian(h->S.desc, sp, hp, h->stream)) return fail(CORBO_HIP_ERR_UNSUPPORTED, "no Hessian kernel for this dynamics/defect");
    HIP_TRY(hipGetLastError());
    if (to_pinned)
        for (int c = 0; c < 3; ++c)
            if (H->nnz[c] > 0) { rc = copy_to_pinned(h, h->hb_vals[c].d(), h->hb_pin.d() + off[c], off[c + 1] - off[c]); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int corbo_hip_eval_hessians(corbo_hip_handle h, int lower_part_only, double mult_obj, const double* mult_eq, const double* mult_ineq, double* vals_obj,
                            double* vals_eq, double* vals_ineq)
try {
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    const HessianStructure* H = nullptr;
    size_t off[4];
    double* out[3] = {vals_obj, vals_eq, vals_ineq};
    {
        HessParams probe;
        int rc0 = hessian_common(h, H, lower_part_only != 0, probe);
        if (rc0) return rc0;
        for (int c = 0; c < 3; ++c)
            if (H->nnz[c] > 0 && !out[c]) return fail(CORBO_HIP_ERR_INVALID, "null value array for a non-empty list");
    }
    const size_t B = (size_t)h->active;
    // small results travel through the pinned buffer (copy kernel, no copy engine); large ones (the 36 MB of a 1024-instance batch) go
    // straight into the caller's arrays -- a second pass over them on the host would cost more than the engine's wake-up
    const bool small = B * ((size_t)H->nnz[0] + H->nnz[1] + H->nnz[2]) * sizeof(double) <= ((size_t)8 << 20);
    int rc = eval_hessians_device(h, lower_part_only, mult_obj, mult_eq, mult_ineq, H, off, small);
    if (rc) return rc;
    for (int c = 0; c < 3; ++c) {
        if (H->nnz[c] == 0) continue;
        if (small) std::memcpy(out[c], h->hb_pin.d() + off[c], B * H->nnz[c] * sizeof(double));
        else HIP_TRY(hipMemcpy(out[c], h->hb_vals[c].p, B * H->nnz[c] * sizeof(double), hipMemcpyDeviceToHost));
    }
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_eval_hessians_views(corbo_hip_handle h, int lower_part_only, double mult_obj, const double* mult_eq, const double* mult_ineq, int device_views,
                                  const double** vals_obj, const double** vals_eq, const double** vals_ineq)
try {
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    const HessianStructure* H = nullptr;
    size_t off[4];
    int rc = eval_hessians_device(h, lower_part_only, mult_obj, mult_eq, mult_ineq, H, off, device_views == 0);
    if (rc) return rc;
    const double** out[3] = {vals_obj, vals_eq, vals_ineq};
    for (int c = 0; c < 3; ++c)
        if (out[c]) *out[c] = (H->nnz[c] == 0) ? nullptr : (device_views ? h->hb_vals[c].d() : h->hb_pin.d() + off[c]);
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_eval_objective_gradient(corbo_hip_handle h, double* grad, double* obj)
try {
    if (!h || !grad) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    const HessianStructure* H = nullptr;
    HessParams hp;
    int rc = hessian_common(h, H, false, hp);
    if (rc) return rc;
    const size_t B = (size_t)h->active, n = (size_t)h->S.dims.n, N = (size_t)h->S.N;
    const size_t ng = (B * n + 1) & ~(size_t)1, no = (B * N + 1) & ~(size_t)1;
    HIP_TRY(h->hb_grad.need(ng * sizeof(double)));
    HIP_TRY(h->hb_obj.need(no * sizeof(double)));
    HIP_TRY(h->hb_pin.need((ng + no) * sizeof(double)));
    HIP_TRY(hipMemsetAsync(h->hb_grad.p, 0, ng * sizeof(double), h->stream));
    hp.mode = 2;
    hp.grad = h->hb_grad.d(); hp.obj_part = h->hb_obj.d(); hp.n_params = (int32_t)n;
    const SweepParams sp = h->sweep_params(0, 0, 1.0, 1.0, 1.0, nullptr);
    if (!launch_hessian(h->S.desc, sp, hp, h->stream)) return fail(CORBO_HIP_ERR_UNSUPPORTED, "no Hessian kernel for this dynamics/defect");
    HIP_TRY(hipGetLastError());
    launch_copy_rows(h->hb_grad.d(), h->hb_pin.d(), nullptr, ng, h->stream);
    if (obj) launch_copy_rows(h->hb_obj.d(), h->hb_pin.d() + ng, nullptr, no, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::memcpy(grad, h->hb_pin.p, B * n * sizeof(double));
    if (obj) {
        const double* part = h->hb_pin.d() + ng;
        for (size_t b = 0; b < B; ++b) {   // in edge order: stages 0 .. N-2, then the final cost
            double v = 0.0;
            for (size_t k = 0; k < N; ++k) v += part[b * N + k];
            obj[b] = v;
        }
    }
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_linear_form_structure(const corbo_hip_problem_desc* desc, int32_t* nnz_out, int32_t* n_rows_out, int32_t* rows, int32_t* cols)
try {
    if (!desc) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    Structure S;
    std::string err = build_structure(*desc, S);
    if (!err.empty()) return fail(CORBO_HIP_ERR_INVALID, err);
    HessianStructure H;
    build_hessian_structure(S, false, H);
    if (nnz_out) *nnz_out = H.lin_nnz;
    if (n_rows_out) *n_rows_out = S.dims.eq + S.dims.ineq + S.dims.bounds;
    if (rows && cols && H.lin_nnz > 0) {
        std::memcpy(rows, H.lin_rows.data(), H.lin_nnz * sizeof(int32_t));
        std::memcpy(cols, H.lin_cols.data(), H.lin_nnz * sizeof(int32_t));
    }
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_eval_linear_form(corbo_hip_handle h, double* vals, double* lbA, double* ubA)
try {
    if (!h || !vals || !lbA || !ubA) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    const HessianStructure* H = nullptr;
    HessParams hp;
    int rc = hessian_common(h, H, false, hp);
    if (rc) return rc;
    const size_t B = (size_t)h->active, rows = (size_t)(hp.eq_dim + hp.ineq_dim + hp.n_bounds);
    const size_t nv = (B * H->lin_nnz + 1) & ~(size_t)1, nr = (B * rows + 1) & ~(size_t)1;
    HIP_TRY(h->hb_lin.need(nv * sizeof(double)));
    HIP_TRY(h->hb_lb.need(nr * sizeof(double)));
    HIP_TRY(h->hb_ub.need(nr * sizeof(double)));
    HIP_TRY(h->hb_pin.need((nv + 2 * nr) * sizeof(double)));
    hp.mode = 1;
    hp.lin_vals = h->hb_lin.d(); hp.lbA = h->hb_lb.d(); hp.ubA = h->hb_ub.d();
    const SweepParams sp = h->sweep_params(0, 0, 1.0, 1.0, 1.0, nullptr);
    if (!launch_hessian(h->S.desc, sp, hp, h->stream)) return fail(CORBO_HIP_ERR_UNSUPPORTED, "no Hessian kernel for this dynamics/defect");
    HIP_TRY(hipGetLastError());
    const bool small = (nv + 2 * nr) * sizeof(double) <= ((size_t)8 << 20);   // (see corbo_hip_eval_hessians)
    if (small) {
        if (H->lin_nnz > 0) { rc = copy_to_pinned(h, h->hb_lin.d(), h->hb_pin.d(), nv); if (rc) return rc; }
        if (rows > 0) {
            rc = copy_to_pinned(h, h->hb_lb.d(), h->hb_pin.d() + nv, nr); if (rc) return rc;
            rc = copy_to_pinned(h, h->hb_ub.d(), h->hb_pin.d() + nv + nr, nr); if (rc) return rc;
        }
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (small) {
        if (H->lin_nnz > 0) std::memcpy(vals, h->hb_pin.p, B * H->lin_nnz * sizeof(double));
        if (rows > 0) {
            std::memcpy(lbA, h->hb_pin.d() + nv, B * rows * sizeof(double));
            std::memcpy(ubA, h->hb_pin.d() + nv + nr, B * rows * sizeof(double));
        }
    }
    else {
        if (H->lin_nnz > 0) HIP_TRY(hipMemcpy(vals, h->hb_lin.p, B * H->lin_nnz * sizeof(double), hipMemcpyDeviceToHost));
        if (rows > 0) {
            HIP_TRY(hipMemcpy(lbA, h->hb_lb.p, B * rows * sizeof(double), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(ubA, h->hb_ub.p, B * rows * sizeof(double), hipMemcpyDeviceToHost));
        }
    }
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_device_views(corbo_hip_handle h, double** x_dev, double** chi2_dev, void** hip_stream)
{
    if (!h) return fail(CORBO_HIP_ERR_INVALID, "null handle");
    if (x_dev) *x_dev = h->d_x;
    if (chi2_dev) *chi2_dev = h->d_chi2;
    if (hip_stream) *hip_stream = (void*)h->stream;
    return CORBO_HIP_OK;
}

int corbo_hip_device_row_stride(corbo_hip_handle h, int32_t* row_stride)
{
    if (!h || !row_stride) return fail(CORBO_HIP_ERR_INVALID, "null argument");
    *row_stride = h->S.nvs;
    return CORBO_HIP_OK;
}

int corbo_hip_time_sweep(corbo_hip_handle h, double w_eq, double w_ineq, double w_bounds, int with_jacobian, int repeat, float* ms_per_launch)
{
    if (!h || !ms_per_launch || repeat < 1) return fail(CORBO_HIP_ERR_INVALID, "bad argument");
    if (!h->have_data) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_set_instance_data must be called first");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    if (h->S.desc.cost_nonlsq) return fail(CORBO_HIP_ERR_UNSUPPORTED, "not a least-squares problem (cost_nonlsq): LevenbergMarquardtSparse::solve refuses it too (levenberg_marquardt_sparse.cpp:48-55); the Hessian-path operators work on it");
    SweepParams p = h->sweep_params(with_jacobian ? 1 : 0, 0, w_eq, w_ineq, w_bounds, nullptr);
    long long* d_tl = nullptr;  // CORBO_HIP_SWEEP_TIMELINE=1: shader-clock stamps of the phases of instance 0 on stderr (diagnostics)
    if (h->sweep_timeline) {
        HIP_TRY(hipMalloc((void**)&d_tl, 16 * sizeof(long long)));
        HIP_TRY(hipMemset(d_tl, 0, 16 * sizeof(long long)));
    }
    p.timeline = d_tl;
    int rc = launch_sweep_checked(h, p);  // warm-up
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < repeat; ++i) {
        rc = launch_sweep_checked(h, p);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *ms_per_launch = ms / repeat;
    if (d_tl) {
        long long tl[16];
        HIP_TRY(hipMemcpy(tl, d_tl, sizeof(tl), hipMemcpyDeviceToHost));
        fprintf(stderr, "sweep phases of instance 0 (cycles): stage-x %lld | caches %lld | residual %lld | decision %lld | defect-J %lld | rest-J %lld | barrier %lld | stream-out %lld | total %lld\n",
                tl[1] - tl[0], tl[2] - tl[1], tl[3] - tl[2], tl[4] - tl[3], tl[5] - tl[4], tl[6] - tl[5], tl[7] - tl[6], tl[8] - tl[7], tl[8] - tl[0]);
        fprintf(stderr, "   residual detail: components %lld | stages %lld\n", tl[9] - tl[2], tl[3] - tl[9]);
        (void)hipFree(d_tl);
    }
    return CORBO_HIP_OK;
}

int corbo_hip_time_sweep_each(corbo_hip_handle h, double w_eq, double w_ineq, double w_bounds, int with_jacobian, int repeat, float* ms_each)
try {
    if (!h || !ms_each || repeat < 1) return fail(CORBO_HIP_ERR_INVALID, "bad argument");
    if (h->S.desc.cost_nonlsq) return fail(CORBO_HIP_ERR_UNSUPPORTED, "not a least-squares problem (cost_nonlsq): LevenbergMarquardtSparse::solve refuses it too (levenberg_marquardt_sparse.cpp:48-55); the Hessian-path operators work on it");
    if (!h->have_data) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_set_instance_data must be called first");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    const SweepParams p = h->sweep_params(with_jacobian ? 1 : 0, 0, w_eq, w_ineq, w_bounds, nullptr);
    EventList evs;
    evs.v.reserve((size_t)repeat + 1);
    int rc = launch_sweep_checked(h, p);  // warm-up
    if (rc) return rc;
    for (int i = 0; i <= repeat; ++i) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        evs.v.push_back(e);
        HIP_TRY(hipEventRecord(e, h->stream));
        if (i < repeat) { rc = launch_sweep_checked(h, p); if (rc) return rc; }
    }
    HIP_TRY(hipEventSynchronize(evs.v.back()));
    for (int i = 0; i < repeat; ++i) HIP_TRY(hipEventElapsedTime(&ms_each[i], evs.v[i], evs.v[i + 1]));
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_time_factor(corbo_hip_handle h, int repeat, float* ms_per_launch, long long* timeline8)
{
    if (!h || !ms_per_launch || repeat < 1) return fail(CORBO_HIP_ERR_INVALID, "bad argument");
    if (!h->have_data) return fail(CORBO_HIP_ERR_STATE, "corbo_hip_set_instance_data must be called first");
    ON_DEVICE_OF(h);
    DRAIN_ASYNC(h);
    if (h->S.desc.cost_nonlsq) return fail(CORBO_HIP_ERR_UNSUPPORTED, "not a least-squares problem (cost_nonlsq): LevenbergMarquardtSparse::solve refuses it too (levenberg_marquardt_sparse.cpp:48-55); the Hessian-path operators work on it");
    // LM prologue (residual + Jacobian + state init), then the assemble/factor/solve kernel `repeat` times on that state
    corbo_hip_lm_opts o;
    corbo_hip_default_lm_opts(&o);
    int rc = launch_sweep_checked(h, h->sweep_params(2, o.iterations, h->w_eq, h->w_ineq, h->w_b, nullptr));
    if (rc) return rc;
    FactorParams fp = h->factor_params();
    fp.first_pass = 1;
    rc = launch_factor_checked(h, fp);  // warm-up (also consumes the `first` pass)
    fp.first_pass = 0;
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < repeat; ++i) {
        rc = launch_factor_checked(h, fp);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *ms_per_launch = ms / repeat;
    if (timeline8) {
        long long* d_tl = nullptr;
        HIP_TRY(hipMalloc((void**)&d_tl, 8 * sizeof(long long)));
        HIP_TRY(hipMemsetAsync(d_tl, 0, 8 * sizeof(long long), h->stream));
        fp.timeline = d_tl;
        rc = launch_factor_checked(h, fp);
        if (rc) { (void)hipFree(d_tl); return rc; }
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipMemcpy(timeline8, d_tl, 8 * sizeof(long long), hipMemcpyDeviceToHost));
        (void)hipFree(d_tl);
    }
    return CORBO_HIP_OK;
}

int corbo_hip_stage_function_kind(int id)
{
#if __has_include("stage_functions/_registry.inc")
#define CORBO_HIP_USER_STAGE(NAME, SLOT, KIND_, NXMIN) if (id == CORBO_HIP_STAGE_FN_USER + SLOT) return KIND_;
#include "stage_functions/_registry.inc"
#undef CORBO_HIP_USER_STAGE
#endif
    (void)id;
    return -1;
}

int corbo_hip_eval_stage_function(int id, int dim, int n, const double* v, const double* prm, double* out)
try {
    if (!v || !prm || !out || n < 1 || dim < 1 || dim > CORBO_HIP_MAX_NX) return fail(CORBO_HIP_ERR_INVALID, "bad argument");
    const int kind = (id == CORBO_HIP_INEQ_BALL) ? 0 : corbo_hip_stage_function_kind(id);
    if (kind < 0) return fail(CORBO_HIP_ERR_INVALID, "not a registered stage function");
    switch (dim) {   // (the functions are templates on the vertex dimension, like the kernels instantiate them)
#define CORBO_HIP_SF_DIM(D) case D: eval_stage_fn<D>(id, kind, n, v, prm, out); break;
        CORBO_HIP_SF_DIM(1) CORBO_HIP_SF_DIM(2) CORBO_HIP_SF_DIM(3) CORBO_HIP_SF_DIM(4) CORBO_HIP_SF_DIM(5) CORBO_HIP_SF_DIM(6)
        CORBO_HIP_SF_DIM(7) CORBO_HIP_SF_DIM(8) CORBO_HIP_SF_DIM(9) CORBO_HIP_SF_DIM(10) CORBO_HIP_SF_DIM(11) CORBO_HIP_SF_DIM(12)
#undef CORBO_HIP_SF_DIM
        default: return fail(CORBO_HIP_ERR_INVALID, "dimension out of range");
    }
    return CORBO_HIP_OK;
}
ABI_CATCH

int corbo_hip_eval_dynamics(const corbo_hip_problem_desc* desc, int n, const double* x, const double* u, double* f)
try {
    if (!desc || !x || !u || !f || n < 1) return fail(CORBO_HIP_ERR_INVALID, "bad argument");
    const int nx = desc->nx, nu = desc->nu;
    if (nx < 1 || nx > CORBO_HIP_MAX_NX || nu < 1 || nu > CORBO_HIP_MAX_NU) return fail(CORBO_HIP_ERR_INVALID, "nx / nu out of range");
    // the plant kernel with the "no integrator" mode: state in [n][MAX_NX], control at offset nx of a row of nx + nu doubles
    struct Buffers {
        double *xp = nullptr, *xu = nullptr, *lin = nullptr;
        ~Buffers() { if (xp) (void)hipFree(xp); if (xu) (void)hipFree(xu); if (lin) (void)hipFree(lin); }
    } b;
    const int row = nx + nu;
    std::vector<double> hxp((size_t)n * CORBO_HIP_MAX_NX, 0.0), hxu((size_t)n * row, 0.0);
    for (int p = 0; p < n; ++p) {
        for (int i = 0; i < nx; ++i) hxp[(size_t)p * CORBO_HIP_MAX_NX + i] = x[(size_t)p * nx + i];
        for (int i = 0; i < nu; ++i) hxu[(size_t)p * row + nx + i] = u[(size_t)p * nu + i];
    }
    HIP_TRY(hipMalloc((void**)&b.xp, hxp.size() * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&b.xu, hxu.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(b.xp, hxp.data(), hxp.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b.xu, hxu.data(), hxu.size() * sizeof(double), hipMemcpyHostToDevice));
    PlantParams p{};
    p.batch = n; p.nvs = row; p.nx = nx; p.nu = nu; p.integrator = 2 /* none: f itself */; p.dt = 1.0;
    std::memcpy(p.dyn, desc->dyn_params, sizeof(p.dyn));
    if (desc->dynamics == CORBO_HIP_DYN_LINEAR_STATE_SPACE) {
        std::vector<double> ab((size_t)nx * nx + (size_t)nx * nu);
        for (int i = 0; i < nx * nx; ++i) ab[i] = desc->lin_a[i];
        for (int i = 0; i < nx * nu; ++i) ab[(size_t)nx * nx + i] = desc->lin_b[i];
        HIP_TRY(hipMalloc((void**)&b.lin, ab.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(b.lin, ab.data(), ab.size() * sizeof(double), hipMemcpyHostToDevice));
        const long long bits = (long long)reinterpret_cast<uintptr_t>(b.lin);
        std::memcpy(&p.dyn[0], &bits, sizeof(double));
    }
    p.x = b.xu; p.xplant = b.xp;
    if (!launch_plant_step(*desc, p, nullptr)) return fail(CORBO_HIP_ERR_UNSUPPORTED, "no device model for this dynamics id / (nx, nu)");
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(hxp.data(), b.xp, hxp.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int q = 0; q < n; ++q)
        for (int i = 0; i < nx; ++i) f[(size_t)q * nx + i] = hxp[(size_t)q * CORBO_HIP_MAX_NX + i];
    return CORBO_HIP_OK;
}
ABI_CATCH

const char* corbo_hip_last_error(void) { return g_last_error.c_str(); }

size_t corbo_hip_sizeof(int which)
{
    switch (which) {
        case 0: return sizeof(corbo_hip_problem_desc);
        case 1: return sizeof(corbo_hip_dims);
        case 2: return sizeof(corbo_hip_lm_opts);
        case 3: return sizeof(corbo_hip_stats);
        default: return 0;
    }
}

}  // extern "C"
