// kernels.hpp -- kernel parameter blocks and launch entry points (definitions in kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "model.hpp"
#include "structure.hpp"

namespace corbo_hip {

// Per-instance Levenberg-Marquardt state machine (LevenbergMarquardtSparse::solve's local variables,
// levenberg_marquardt_sparse.cpp:103-127).  Lives in HBM, one per OCP instance; touched by one lane per pass.
struct alignas(16) LmState {
    double mu;         // damping
    double mu_acc;     // damping accumulated on diag(H) since the last Jacobian refresh (quirk i: never undone on reject)
    double rho;
    double chi2_old;   // last accepted chi2 (*obj_value)
    double last_sq;    // |values|^2 of the LAST computeValues call (accepted or not), feeds stop = |values| <= eps3
    double den;        // delta^T (mu delta + rhs) of the pending trial step
    double dnorm;      // |delta| of the pending trial step
    uint32_t v;        // unsigned int v (doubles on reject, wraps like the reference's)
    int32_t k;         // outer iteration
    int32_t stop;
    int32_t fresh;     // Jacobian/values refreshed since the last factorisation -> rebuild H without accumulated damping
    int32_t first;     // first factorisation of this solve: mu = tau * max diag(H), stop = |rhs|_inf <= eps1
    int32_t no_trial;  // |delta| <= eps2: no trial step pending
    int32_t done;
    int32_t status;    // corbo_hip_solver_status
    int32_t vbuf;      // which residual buffer pairs with the resident Jacobian
    int32_t inner;     // inner passes of the current outer iteration
    int32_t n_accept, n_reject, n_jac, n_res, n_fact;
    int32_t pad[3];  // sizeof == 128: whole-struct copies are eight aligned 16-byte moves
};
static_assert(sizeof(LmState) == 128, "LmState layout");

struct SweepParams {
    // static structure
    int32_t batch, nvs, m, nnz, N, s, nx, off_dt, dt_free;
    int32_t inst0;      // first instance of this launch (sub-batch launches); grid = batch
    int32_t batch_total;  // instances of the handle (stride of the per-handle two-buffer arrays)
    int32_t eq_row0, ineq_row0;  // first residual row of the defect / stage-inequality edges (one edge per stage)
    int32_t ineq_stride;         // rows between the stage inequalities of consecutive intervals: 1, or 1 + nu when every interval also has a control-deviation edge behind it
    const StageCols* stage_cols;  // N-1: Jacobian offsets of the defect columns of stage k
    const CompInfo* comp;         // nvs: cost-block offsets per component
    const int32_t* ineq_cols;     // (N-1)*nx or null
    int32_t fin_row;              // residual row of the final-stage inequality (TerminalBall) or -1
    int32_t fin_joff[CORBO_HIP_MAX_NX];   // its Jacobian entries on x_f, -1 = fixed component
    int32_t fin_eq_row0, fin_eq_dim;      // final-stage equality (TerminalEqualityConstraint: nx rows; partial: one per active component): first row, rows
    ModelParams mp;
    double dt_fixed;
    // per-call
    // integral-form constraint edges / control-deviation edges (structure.hpp XEdge; sweep_body<..., XE>), or n_xedges = 0
    const XEdge* xedges;
    int32_t n_xedges;
    const int4* xtasks;                 // or null: one entry per Jacobian COLUMN of the extra edges -- (edge, attached vertex, component, first Jacobian value of the column)
    int32_t n_xtasks;
    int32_t eq_stride, eq_defect_off;   // equality rows per interval and the defect's row inside them (nx, 0 without integral equality rows)
    const double* xparams;              // [stage_eq: a (nx), b (nu), c | ctrl_dev: r_max (nu)]
    const double* uprev;                // [batch_total][CORBO_HIP_MAX_NU + 1]: previously applied control, its age (corbo_hip_set_previous_control)
    int32_t mode;       // 0 = residual only, 1 = residual + Jacobian, 2 = LM init, 3 = LM trial step
    int32_t iterations; // LM: outer iteration count
    int32_t ff_converged;   // 1 (default): count the outer iterations that follow a step of norm <= eps2 / 2 instead of computing them (sweep_body, mode 3)
    double w_eq, w_ineq, w_b;
    double* x;          // [batch][nvs] accepted iterate
    const double* xt;   // [batch][nvs] trial iterate (mode 3)
    const double* lb;   // [batch][nvs]
    const double* ub;
    const double* xref; // [batch][MAX_NX]
    const double* refvec;  // or null: per-component state references in the vertex layout [batch][nvs] (time-varying reference,
    const double* dyn_inst;      // [batch_total][8] per-instance parameters of the dynamics (corbo_hip_set_instance_params) or null: mp.dyn for all
                           // corbo_hip_set_references): the cost row of state component v is w * (x_v - refvec[v]); final-stage terms use the x_f entries
    double* values0;    // [batch][m]   residual buffer 0
    double* values1;    // [batch][m]   residual buffer 1 (LM only)
    double* jac;        // [batch][nnz_pad]
    int32_t m_pad, nnz_pad;
    int32_t jlean_lo2, jlean_hi2;   // [lo, hi) in double2 units: the part of the Jacobian the two-wave run-to-completion shape moves between its phases (defect / inequality / special
                                    // blocks; the cost blocks and bound rows of the stage components travel in registers, StageKeep); 0, 0: everything
    LmState* st;
    int32_t* active_count;  // number of instances not done after this pass (mode 3)
    long long* timeline;    // optional [16] shader-clock stamps of instance `timeline_inst` (diagnostics), may be null
    int32_t timeline_inst;
    double* chi2;           // [batch] dense copy of the accepted chi2 (*obj_value), written by the LM modes
    // big-block family (multiple shooting with RK4, nx > 6): the Jacobian of an LM pass is never stored.  The residual sweep keeps
    // the end state of the unperturbed Runge-Kutta step of every shooting interval, [2][batch][N][nx] (the half paired with the
    // accepted iterate / the trial half, like values0 / values1), and the stage kernel (big_stage_kernel) differentiates and
    // assembles from the accepted iterate on the fly.
    double* xe0;            // or null
    int32_t skip_jac;       // LM modes: leave the Jacobian to the stage kernel
    const double* x_init;   // or null: (run-to-completion kernel, prologue) start from these iterates and make them the accepted ones -- the re-arm copy of
                            // corbo_hip_restore_instance_data done by the solve kernel itself (corbo_hip_solve[_async] with new_run = 2)
};

struct FactorParams {
    int32_t batch, nvs, m, N, nx, nu, s, off_dt, dt_free;
    int32_t inst0;                // first instance of this launch (sub-batch launches); grid = batch
    int32_t eq_row0;
    const StageCols* stage_cols;  // N-1
    const CompInfo* comp;         // nvs
    const int32_t* ineq_cols;     // (N-1)*nx or null
    const int32_t* ineq_rows;     // N-1 or null
    int32_t fin_row;              // final-stage inequality on the last block (see SweepParams)
    int32_t fin_joff[CORBO_HIP_MAX_NX];
    const double* x;              // accepted iterate
    double* xt;                   // trial iterate out
    const double* values0;
    const double* values1;
    const double* jac;
    int32_t m_pad, nnz_pad;
    int32_t jlean_lo2, jlean_hi2;   // [lo, hi) in double2 units: the part of the Jacobian the two-wave run-to-completion shape moves between its phases (defect / inequality / special
                                    // blocks; the cost blocks and bound rows of the stage components travel in registers, StageKeep); 0, 0: everything
    LmState* st;
    double* delta_out;            // optional [batch][nvs] (debug / tests), may be null
    long long* timeline;          // optional [8] shader-clock stamps of instance `timeline_inst` (diagnostics), may be null
    long long* phase_cycles;      // optional [batch][8] (diagnostics, run-to-completion kernel): shader-clock cycles every instance's workgroup spent in sweep phases that
                                  // end with a Jacobian [0] / residual-only sweep phases [1] / factor phases [2], and their counts [3], [4], [5]; may be null
    int32_t timeline_inst;
    double* work;                 // big-block kernel only: per-instance factor workspace in HBM
    int64_t work_stride;          // doubles per instance
    long long* pass_timeline;     // optional [2 * 64 + 1] shader-clock stamps (pass start, sweep end) of one instance (diagnostics)
    int32_t pass_timeline_inst;
    int32_t band_wide;            // band route: 1 = the eight-wave kernel whatever the half-bandwidth (option "band_wide": A/B, tests); 0 = one wave per instance for half-bandwidths <= 7
    int32_t chain_variant;        // big-block family: 0 = automatic (N >= 64: partitioned chain big_chain3_kernel, four segments; else the twisted chain big_chain2_kernel), 1 = the first formulation, 2 = twisted, 3 / 4 / 6 = 4 / 2 / 1 segments
    int32_t first_pass;           // big-block family: this may be the first factorisation of a solve (launches the mu / stop kernels)
    double* stage_cache;          // big-block family: [batch][pairs][2 x BigLds::HALF] the stage waves' local Jacobians of a solve's FIRST factorisation, written by its
                                  //   diag pass (mu = tau max diag(J^T J) needs all of them before any block can be damped) and read back by its assembly pass, or null
    int64_t stage_cache_stride;   // doubles per instance
    int32_t loop_passes;          // fused pass kernel: > 0 = run-to-completion, at most this many LM passes inside one launch
    double* x_host;               // run-to-completion kernel: optional result sink in pinned, device-visible HOST memory [batch][nvs]: every
    LmState* st_host;             //   workgroup writes its instance's accepted iterate and LM state there as soon as the instance has finished
    int32_t queue_grid;           // workgroups of a launch in queue mode
    int32_t* queue;               // run-to-completion kernel, batch > resident workgroups: ticket counter (zeroed before the launch); every
                                  //   workgroup pulls instance after instance from it.  null = workgroup b solves instance inst0 + b
    int32_t wdense_mask;          // non-diagonal weights (SweepParams::wdense_mask): the cost blocks of those vertices are dense
    int32_t* cu_table;            // run-to-completion kernel: per-CU progress table for the lag-based issue priority (see lm_pass_kernel), or null
    int32_t stagger;              // run-to-completion kernel (diagnostics): workgroup b waits (b / 256) * stagger shader cycles before it starts
    int32_t defect;               // corbo_hip_problem_desc::defect (big-block family: which stage kernel)
    int32_t pass_threads;         // run-to-completion kernel: workgroup size 256 (default) / 192 / 128 (option "pass_threads"; headline shape only)
    int32_t* unfinished_flag;  // run-to-completion kernel: set to 1 by an instance that hits the pass limit (may be device-visible pinned host memory)
    // block-tridiagonal route (small-block families with extra edges; bt_factor.hpp, structure.hpp BtTables) or null / 0
    const uint32_t* bt_pairs;
    const int32_t* bt_off;
    const uint32_t* bt_target;
    int32_t bt_rounds;
    // ... the assembled, damped blocks of an instance's last factorisation [batch][bt_snap_stride]: the factorisation after a REJECTED step has the same J and the
    // same right-hand side, it reloads them and adds its mu to the diagonal (levenberg_marquardt_sparse.cpp:135-138: H_ii += mu, never undone) instead of assembling again
    double* bt_snap;
    int32_t bt_snap_stride;
    int32_t bt_waves;             // handle option "bt_waves": 0 = launch_bt_t chooses by batch size, 2 / 3 = the instantiation for that many workgroups per CU (A/B)
    int32_t num_cus;              // compute units of the device (launch_bt_t: how many instances are resident at once)
};

// Reject-streak speculation of the big-block family (big_spec_kernel; VERDICT r3 item 2 a).  An instance whose trial step was rejected re-factorises the
// SAME Jacobian with the damping mu v, then mu v 2v, ... (levenberg_marquardt_sparse.cpp:204-215) until a step is accepted: every one of those passes is a
// full launch group for one or two instances (cfg 5: two streaks of five and four rejections = 5 of the 15 passes of the solve).  The dampings of a streak
// are known in advance, so the next SPEC of them are tried in the same pass as the instance's own: SPEC spare instance slots per rejecting instance
// (rows behind the batch in every per-instance array) start from copies of its accepted iterate with the LM state it WOULD have after 1, 2, ... more
// rejections; after the pass's sweep this kernel walks the candidates in order, the first one that did not reject is what the instance itself would have
// reached that many passes later and is copied back (iterate, end states, residual buffers, LM state -- counters included), the others are dropped.  Same
// arithmetic on the same numbers: bit-identical results.
struct SpecParams {
    int32_t mode;            // 0 = after a pass's sweep: merge + spawn; 1 = start of a solve: every slot free
    int32_t batch;           // real instances (rows [0, batch)); slots are rows [batch, batch + groups * spec)
    int32_t groups, spec;    // slot groups (one rejecting instance each), slots per group
    int32_t nvs, m_pad, xe_row, batch_total;   // row lengths: vertex storage, residual buffer, N * nx end states; rows of the two-buffer end-state array
    double *x, *lb, *ub, *xref, *values0, *values1, *xe0, *chi2;
    LmState* st;
    int32_t* parent_of;      // [groups] instance of the group or -1
    int32_t* rej_seen;       // [groups] the parent's n_reject when its slots were filled
    int32_t* slot_rej;       // [groups * spec] a slot's n_reject when it was filled
    int32_t* prev_reject;    // [batch] n_reject of every instance at the last visit
    int32_t* counter;        // the pass's "unfinished instances" counter (slots that the sweep counted are taken out again) or null
    int32_t max_parents;     // streaks followed at a time (speculation pays when rejections are rare: see big_spec_kernel)
    int32_t* adopted;        // statistics: times an instance took over a candidate's state in this solve (corbo_hip_stats.speculative_takeovers) or null
};
bool launch_big_spec(const SpecParams& p, hipStream_t stream);

// Band factorisation (band_factor_kernel): the generic assemble / factor / solve step for structures the stage-parallel kernels do not
// cover -- integral-form constraint edges, control-deviation edges (they couple the controls of neighbouring intervals).  H = J^T J is
// assembled from static product lists into band storage (natural parameter order: half-bandwidth of a few stage widths), a free dt -- the last
// parameter, a dense row of H -- is carried as a border column.  Tables built once per handle (corbo_hip_create).
struct BandParams {
    int32_t n;            // parameters (columns of J)
    int32_t nb;           // parameters inside the band (n, or n - 1 with a free dt)
    int32_t bw;           // half-bandwidth: H(r, c) is structurally zero for r - c > bw (r, c < nb)
    int32_t n_ent;        // entries of the lower part of H that are assembled: band entries, border entries, the corner
    const int32_t* ent_target;  // [n_ent] r * (bw + 1) + (bw - (r - c)) for a band entry; -1 - c for the border entry (n - 1, c); INT32_MIN for the corner
    const int32_t* ent_ptr;     // [n_ent + 1] into ent_pairs
    const int32_t* ent_pairs;   // [2 * pairs] Jacobian value indices (a, b): H entry = sum J[a] * J[b]
    const int32_t* rhs_ptr;     // [n + 1] into rhs_ent
    const int32_t* rhs_ent;     // [2 * entries] (Jacobian value index, residual row): rhs_c = - sum J[a] * values[row]
    const int32_t* param_voff;  // [n] parameter -> vertex storage offset
    double* work;               // [batch][work_stride] band + vectors (HBM), used when the LDS does not hold them
    int64_t work_stride;
    int32_t use_lds;
};

// Operators of the exact-Hessian path, evaluated at the accepted iterate of every instance (hessian_kernel):
// mode 0: the value lists of computeSparseHessiansValues (objective / equalities / inequalities); mode 1: the two-side-bounded linear
// form (values, lbA, ubA); mode 2: gradient and value of the objective.  Structure: build_hessian_structure (structure.hpp).
struct HessParams {
    int32_t mode, lower;
    int32_t split;              // mode 0: 0 = one lane walks all edges of its stage, 1 = one wave per (edge, row vertex), 2 = per block (hessian_kernel); chosen from the batch size
    double mult_obj;
    const double* mult_eq;      // [batch][eq_dim] or null (= 1)
    const double* mult_ineq;    // [batch][ineq_dim] or null
    int32_t eq_dim, ineq_dim;
    const int32_t* stage_off;   // [N][6] (HessianStructure::stage_off)
    double* vals[3];            // [batch][nnz[c]]
    int32_t nnz[3];
    const int32_t* lin_off;     // [N][2]
    double* lin_vals;           // [batch][lin_nnz]
    double* lbA;                // [batch][eq_dim + ineq_dim + bounds]
    double* ubA;
    int32_t lin_nnz, lin_bounds0, bnd_row0, n_bounds;
    int32_t stage_cost, stage_ineq;   // corbo_hip_cost / corbo_hip_ineq of the descriptor
    int32_t dt_cost_off;              // HessianStructure::dt_cost_off
    int32_t quad_first_interval;      // descriptor field (MinTimeQuadratic::only_last_n)
    int32_t cost_nonlsq;              // descriptor field: the cost edges are plain objective edges (scalar terms)
    int32_t cost_integral;            // descriptor field: 1 / 2 = one trapezoidal / left-sum integral cost edge per interval
    int32_t ms_mixed;                 // cost_integral on a MultipleShootingGrid: one MultipleShootingEdgeSingleControl (mixed edge) per interval
    // mode 2: gradient of the least-squares objective, computeGradientObjective (hyper_graph_optimization_problem_edge_based.cpp:31-102)
    double* grad;               // [batch][n], zeroed by the caller (every parameter is written by the one lane that owns its component)
    double* obj_part;           // [batch][N]: the stage's share of computeValueObjective (sum of the squared norms of its cost edges)
    int32_t n_params;
};
bool launch_hessian(const corbo_hip_problem_desc& d, const SweepParams& sp, const HessParams& hp, hipStream_t stream);

// returns false if the (dynamics, defect) pair has no device instantiation
bool launch_sweep(const corbo_hip_problem_desc& d, const SweepParams& p, hipStream_t stream);
// sp: the sweep parameters of the same pass (big-block family: the stage kernel evaluates the edges itself); may be null for the
// LDS-resident small-block families
bool launch_factor(const corbo_hip_problem_desc& d, const FactorParams& p, hipStream_t stream, const SweepParams* sp = nullptr);
size_t big_stage_cache_doubles(const corbo_hip_problem_desc& d, int N);   // per instance (0: not a big-block descriptor)
// one fused LM pass: [sweep phase (sp.mode 2 = prologue, 3 = trial step) -> factor phase] per workgroup, one launch
struct WarmStartParams {
    int32_t batch, nvs, nx, nu, N, xf_fixed_mask, shift;
    double* x;            // [batch][nvs] accepted iterate, updated in place
    const double* x0new;  // [batch][CORBO_HIP_MAX_NX]
    const double* xref;   // [batch][CORBO_HIP_MAX_NX]
};
void launch_warm_start(const WarmStartParams& p, hipStream_t stream);
// FullDiscretizationGridBase::resampleTrajectory (full_discretization_grid_base.cpp:397-474) between two batches of free-dt grids with
// N_src / N_dst grid points: pair q takes instance src_index[q] of the source arrays to instance dst_index[q] of the destination arrays
struct ResampleParams {
    int32_t pairs, nx, nu;
    int32_t n_src, nvs_src, n_dst, nvs_dst;
    const double* x_src;      // [.][nvs_src]
    double* x_dst;            // [.][nvs_dst]
    const double* xref_src;   // [.][CORBO_HIP_MAX_NX]
    double* xref_dst;
    const int32_t* src_index; // [pairs] (device-visible)
    const int32_t* dst_index;
};
void launch_resample(const ResampleParams& p, hipStream_t stream);
// dst (and dst2, may be null) = src, `doubles` values (even), as a kernel on `stream` (no copy engine: see kernels.hip)
// (src may be pinned, device-visible HOST memory: the upload of a staged array is then the same kernel, no DMA engine either)
void launch_copy_rows(const double* src, double* dst, double* dst2, size_t doubles, hipStream_t stream, double* dst3 = nullptr);
void launch_zero_ints(int32_t* dst, size_t n, hipStream_t stream);
// refvec[b][x_k entries] = traj[b][min(step + k, T - 1)] for k = 0 .. N-1 (the window of a resident reference trajectory that control step
// `step` sees: DiscreteTimeReferenceTrajectory sampled at t + k dt, the last sample held beyond its end); other entries are left alone
void launch_reference_window(const double* traj, double* refvec, int batch, int T, int N, int nx, int s, int nvs, int step, hipStream_t stream);
// dt_out[b] = x[b][off_dt]  (packed; dt_out may be device-visible pinned host memory)
void launch_gather_dt(const double* x, double* out, int nvs, int off_dt, int batch, hipStream_t stream);
// the plant side of a closed loop (SimulatedPlant::control, plants/src/simulated_plant.cpp:97-160, no dead time): one thread per instance
struct PlantParams {
    int32_t batch, nvs, nx, nu, integrator;   // integrator: corbo_hip_integrator
    double dt;
    double dyn[8];
    const double* dyn_inst;     // or null: the plant's OWN model parameters per instance [batch][8] (a plant that differs from the controller's
                                // model: SimulatedPlant takes its own dynamics object, plants/src/simulated_plant.cpp)
    const double* x;            // [batch][nvs] resident trajectories: u_0 = x[b][nx .. nx+nu)
    double* xplant;             // [batch][CORBO_HIP_MAX_NX] plant states, updated in place
    const double* disturbance;  // [batch][CORBO_HIP_MAX_NX] added to the new state (may be pinned host memory), or null
    double* log_x;              // [batch][nx] packed: the new plant states (closed-loop log of this step), or null
    double* log_u;              // [batch][nu] packed: the controls that were applied, or null
};
bool launch_plant_step(const corbo_hip_problem_desc& d, const PlantParams& p, hipStream_t stream);
// out[b][0..nu) = x[b][nx .. nx+nu)  (u_0 of every instance, packed; `out` may be device-visible pinned host memory)
void launch_gather_first_control(const double* x, double* out, int nvs, int nx, int nu, int batch, hipStream_t stream);
// dst_a[b][:] = row_a, dst_b[b][:] = row_b for b < batch (the descriptor's bound pattern repeated for every instance)
void launch_broadcast_rows(const double* row_a, const double* row_b, double* dst_a, double* dst_b, int nvs, int batch, hipStream_t stream);
// corbo_hip_set_instance_data as one launch (see upload_instance_kernel); double2 element counts, nvs is even
struct UploadParams {
    const double2* x; double2 *dx, *dxt, *dx0; size_t n2;
    const double2 *lb_src, *ub_src;   // per-instance bound arrays in pinned staging, or null: the pattern rows
    const double2 *row_lb, *row_ub; double2 *dlb, *dub; size_t nvs2;
    const double2* xref; double2* dxref; size_t nref2;
};
void launch_upload_instance(const UploadParams& p, hipStream_t stream);

// big-block family, parity hook: the Jacobian values the stage kernel differentiates (defect blocks, cost / bound / inequality rows of
// every interval) at the accepted iterate, written into jac_out [batch][nnz_pad] in the public value order; needs a residual sweep
// (mode 0 / 1 with sp.xe0) of the same iterate before it
bool launch_stage_jacobian_dump(const corbo_hip_problem_desc& d, const FactorParams& fp, const SweepParams& sp, double* jac_out, hipStream_t stream);
bool device_kernels_exist(const corbo_hip_problem_desc& d);   // host-only mirror of the dispatch (corbo_hip_create's gate)
bool launch_pass(const corbo_hip_problem_desc& d, const FactorParams& fp, const SweepParams& sp, hipStream_t stream);
bool launch_band_factor(const FactorParams& fp, const BandParams& bp, hipStream_t stream);
// Block-tridiagonal route (lm_bt_kernel): what corbo_hip_create asks before it builds the tables -- block size s = nx + nu, free dt, grid points, the
// padded Jacobian / residual lengths, the dynamics' cache doubles per grid state.  0 = no such kernel (horizon, LDS); else the rounds its assembly holds.
int bt_route_max_rounds(int nx, int nu, bool arrow, int N, int nnz_pad, int m_pad, int nvs);
constexpr int BT_THREADS = 256;
size_t band_work_doubles(int nb, int bw);
bool band_route_supported(int nb, int bw);   // half-bandwidth <= 63 and the sliding window + vectors within the LDS of a CU
size_t sweep_lds_bytes(const SweepParams& p, int nc);
// Small-block families with horizons up to 256 grid points assemble the Jacobian values in an LDS staging area (STAGE in sweep_body); the
// device-internal value layout carries one pad double per defect block for exactly those (corbo_hip_create).
constexpr bool jacobian_staged_in_lds(int nx, int N) { return nx <= 4 && N <= 256; }
size_t factor_lds_bytes(const corbo_hip_problem_desc& d, const FactorParams& p);
// doubles of HBM workspace per instance the factor kernel needs (0 for the LDS-resident small-block kernel)
size_t factor_work_doubles(const corbo_hip_problem_desc& d);

}  // namespace corbo_hip
