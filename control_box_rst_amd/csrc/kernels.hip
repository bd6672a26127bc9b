// kernels.hip -- CDNA4 (gfx950) kernels of the hypergraph NLP inner loop.
//
//  sweep_kernel   one workgroup per OCP instance.  Evaluates every hypergraph edge (stage cost, dynamics defect, stage
//                 inequality, bounds) -> stacked residual, and every central-difference Jacobian column, one lane per
//                 (edge, vertex component) column task, vertices and per-state dynamics caches staged in LDS, every
//                 Jacobian column stored to HBM as soon as it is computed (stores overlap the remaining columns).  In LM mode it also runs the trial-step bookkeeping (rho,
//                 accept/reject, mu update) of LevenbergMarquardtSparse::solve so the residual re-evaluation of the line
//                 search is the same kernel (levenberg_marquardt_sparse.cpp:158-213).
//  factor_kernel  one workgroup per OCP instance.  Assembles H = J^T J + mu I and rhs = -J^T r from the block-sparse
//                 Jacobian (levenberg_marquardt_sparse.cpp:97-100,135-138), eliminates the per-stage controls in parallel,
//                 factors the remaining block-tridiagonal state system by block cyclic reduction (a Cholesky
//                 factorisation in nested-dissection order, replaces Eigen::SimplicialLLT :140-148), handles a free dt as
//                 the last (arrowhead) pivot, back-substitutes and writes the trial iterate x + delta (:158-161).
//
// Both are table driven (structure.hpp); nothing here knows about edge objects.
#include "kernels.hpp"

namespace corbo_hip {

namespace {

constexpr int SWEEP_THREADS = 256;
constexpr int LONG_HORIZON = 256;   // grid points beyond which the small-block families switch to the long-horizon kernels (up to LONG_HORIZON_MAX)
constexpr int LONG_HORIZON_MAX = 1024;
constexpr double LM_EPS1 = 1e-5, LM_EPS2 = 1e-5, LM_EPS3 = 1e-5, LM_EPS4 = 0.0, LM_TAU = 1e-5;  // levenberg_marquardt_sparse.cpp:103-110
constexpr int LM_MAX_INNER = 64;  // guard against an endless reject loop (the reference would spin)

// hipFuncSetAttribute is per DEVICE and handles may live on several: a launcher's "already set" flag is one bit per device.
inline bool first_on_device(unsigned long long& mask)
{
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    if (mask & bit) return false;
    mask |= bit;
    return true;
}

// Workgroup barrier for hand-overs through LDS only.  __syncthreads() is a workgroup-scope release/acquire fence + s_barrier: the
// fence waits for EVERY outstanding memory operation of the wave (s_waitcnt vmcnt(0)), i.e. also for the write acknowledgements of
// global stores that nobody in the workgroup is going to read (gfx9 counts stores in vmcnt) -- an HBM / L2 round trip at every
// barrier that follows a store.  Where the data that crosses the barrier lives in LDS, waiting for the wave's LDS operations is
// enough.  (-DCORBO_HIP_FULL_BARRIERS: the conservative barrier everywhere, for A/B measurements.)
__device__ __forceinline__ void lds_barrier()
{
#ifdef CORBO_HIP_FULL_BARRIERS
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// Wave-wide reductions without the LDS crossbar: __shfl_xor on a double is two ds_bpermute_b32 per step, six dependent LDS round trips per
// reduction (>= 800 cycles on an idle CU, and the LDS pipeline is the resource the four workgroups of a CU share) -- three of them sat on the
// critical path of every LM pass.  Four DPP steps give every lane the sum of its row of 16, four v_readlane pairs combine the rows.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int lane)   // (uniform: lives in scalar registers)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v)   // all 64 lanes active; every lane gets the total
{
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);   // row_half_mirror (the other quad pair of the 8: holds that pair's sum in every lane)
    v += dpp_move<0x140>(v);   // row_mirror (the other half of the row)
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ double wave_max(double v)
{
    v = fmax(v, dpp_move<0xB1>(v));
    v = fmax(v, dpp_move<0x4E>(v));
    v = fmax(v, dpp_move<0x141>(v));
    v = fmax(v, dpp_move<0x140>(v));
    return fmax(fmax(lane_value(v, 0), lane_value(v, 16)), fmax(lane_value(v, 32), lane_value(v, 48)));
}

// end of one outer LM iteration: levenberg_marquardt_sparse.cpp:216-218
// (works on scalars: whole-struct copies of LmState end up in scratch memory)
// A value every lane of the wave loads from the same address: kept in scalar registers like a kernel argument.
__device__ __forceinline__ double uniform_load(const double* q)
{
    const double v = *q;
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// The 8 parameters of the dynamics for one instance: the descriptor's (kernel argument) or the instance's own
// (corbo_hip_set_instance_params; the instance is uniform over the workgroup / wave in every kernel that evaluates dynamics).
#define CORBO_HIP_DYN_OF(dynl, P, INST)                                                                \
    double dynl[8];                                                                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) dynl[i_] = (P).mp.dyn[i_];                         \
    if ((P).dyn_inst) {                                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) dynl[i_] = uniform_load((P).dyn_inst + (size_t)(INST) * 8 + i_); \
    }

__device__ __forceinline__ bool lm_end_outer(LmState* st, double last_sq, double rho, int k, int iterations)
{
    const int stop = (sqrt(last_sq) <= LM_EPS3) ? 1 : 0;
    st->stop  = stop;
    st->inner = 0;
    st->k     = k + 1;
    if (k + 1 >= iterations) {
        st->done   = 1;
        st->status = (stop || rho <= 0) ? CORBO_HIP_SOLVER_CONVERGED : CORBO_HIP_SOLVER_EARLY_TERMINATED;
        return true;
    }
    return false;
}

// The 128-byte LM state of the instance lives in LDS while its workgroup works on it (the pass loop would otherwise pay a global
// round trip at the start of every phase and for every read-modify-write of a counter); 8 lanes move it in / out.
__device__ __forceinline__ void lm_state_in(LmState* sl, const LmState* sg, int tid)
{
    static_assert(sizeof(LmState) == 128, "LmState is moved as 8 x 16 bytes");
    if (tid < 8) reinterpret_cast<double2*>(sl)[tid] = reinterpret_cast<const double2*>(sg)[tid];
}
__device__ __forceinline__ void lm_state_out(LmState* sg, const LmState* sl, int tid)
{
    if (tid < 8) reinterpret_cast<double2*>(sg)[tid] = reinterpret_cast<const double2*>(sl)[tid];
}

// ---------------------------------------------------------------------------------------------------------------------
// edge / Jacobian sweep
// ---------------------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)

// Two-wave shape of the run-to-completion kernel: what lane k's factor phase needs from the residual paired with the resident Jacobian -- the
// single-entry rows of block k's components (values + Jacobian entries) and defect k's residual -- stays in lane k's REGISTERS from the sweep
// phase that evaluated it (the stage-centric component pass and the defect loop run in the same lane) until the next accepted step replaces it:
// the same rule as the flip of the two residual buffers in HBM (LmState::vbuf), without the round trip through them.
template <int NX, int NU>
struct StageKeep {
    double vc[NX + NU], ac[NX + NU], vb[NX + NU], ab[NX + NU], r[NX];
};

// The single-entry rows of ONE vertex component exactly as the edges define them (contraction off: the same IEEE operations wherever this
// is evaluated): cost value w (x - ref) with its central-difference Jacobian entry (edge_interface.cpp:55-96, delta = 1e-9), distance to the
// box bounds times w_b (hyper_graph_optimization_problem_base.cpp:291-315) with its entry -w_b / 0 / +w_b (:1721-1752).  The two-wave shape of
// the run-to-completion kernel evaluates these rows where they are consumed -- chi2 in the sweep phase, H and rhs in the factor phase -- and
// never moves them through memory.
__device__ __forceinline__ void diagonal_rows(double xv, double w, double ref, double lo, double up, double w_b, double& val, double& dv, double& bval, double& bent)
{
    constexpr double delta = 1e-9, neg2delta = -2 * delta, scalar = 1.0 / (2 * delta);
    val = w * (xv - ref);
    const double a = xv + delta, b = a + neg2delta;
    dv   = scalar * (w * (a - ref) - w * (b - ref));
    bval = fmax(fmax(lo - xv, xv - up), 0.0) * w_b;   // (lo <= up: at most one of the two distances is positive)
    bent = (((xv > up) ? 1.0 : 0.0) - ((xv < lo) ? 1.0 : 0.0)) * w_b;
}

constexpr int SWEEP_ROLE_TABLE = 2 * CORBO_HIP_MAX_NX + CORBO_HIP_MAX_NX + CORBO_HIP_MAX_NU;   // doubles: [sq | sr] [sqf] [xref], see sweep_body
#ifdef CORBO_HIP_LEVEL_STAMPS
#define SWEEP_STAMP(id) do { } while (0)
#else
#define SWEEP_STAMP(id)                                                     \
    do {                                                                    \
        if (p.timeline && inst == p.timeline_inst && tid == 0) p.timeline[id] = clock64(); \
    } while (0)
#endif

// Row c of  U (x - ref)  for a vertex with a NON-DIAGONAL weight (quadratic_cost.cpp:116-118, 148-150, final_state_cost.cpp:88-90:
// `cost.noalias() = _Q_sqrt * xd` with xd = x_k - xref(k), U = the upper Cholesky factor kept by setWeightQ / setWeightR / setWeightQf).
// Eigen evaluates the dynamic-size product as a column-major gemv into a zeroed destination: one running sum per row over the columns,
// a full block of FOUR columns added pairwise (the order LinearDynamics restates for f = A x + B u, measured against the compiled
// reference).  U: row-major [dim][dim] with its explicit zeros below the diagonal -- they take part in the sum like in the reference.
// xd: the vertex's differences x_j - ref_j (ref = 0 for controls), xd[c] possibly replaced by a perturbed value (finite differences).
// THREE columns into a destination that starts on an odd row of the solver's residual vector (8 bytes past a 16-byte boundary): the kernel skips
// column 0 to line its packets up, runs columns 1, 2 and adds column 0 last (GeneralMatrixVector.h, skipColumns; alignmentStep = 1 for an odd
// leading dimension) -- `odd`; finite-difference temporaries start aligned.  Pinned bit for bit by unicycle_n300_fullq (150 odd stages).
template <int DM>
__device__ __forceinline__ double dense_weight_row(const double* U, int c, int dim, const double (&xd)[DM], bool odd = false)
{
    double u[DM];
#pragma unroll
    for (int j = 0; j < DM; ++j) u[j] = (j < dim) ? U[c * dim + j] : 0.0;
    if (dim == 3 && odd) {
        if constexpr (DM >= 3) return ((0.0 + u[1] * xd[1]) + u[2] * xd[2]) + u[0] * xd[0];
    }
    if (dim == 4) {
        if constexpr (DM >= 4) return 0.0 + ((u[0] * xd[0] + u[1] * xd[1]) + (u[2] * xd[2] + u[3] * xd[3]));
    }
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < DM; ++j)
        if (j < dim) acc += u[j] * xd[j];
    return acc;
}

// LDS operands: xs [nvs] vertex values of this instance, red [10] reduction scratch + 4 int flags, cs [N*NC] per-grid-state
// dynamics caches, jst [nnz_pad] Jacobian staging.  FUSED: a factor phase follows in the same workgroup and takes the Jacobian
// straight from jst when this phase refreshed it (flag word [0]).
// DENSE: the descriptor has non-diagonal weights (corbo_hip_problem_desc::weights_dense).  A compile-time switch, instantiated for the
// stand-alone kernels only (such handles run the phases as separate launches): inside the fused run-to-completion kernel even the
// never-taken branches cost the headline path a third of its speed (register allocation: 44 -> 82 spilled VGPRs, measured).
// LONG: horizons beyond 256 grid points (up to 1024; FiniteDifferencesVariableGrid's default n_max is 1000,
// finite_differences_variable_grid.h:82): the Jacobian of such an instance does not fit the LDS staging area, its entries go straight
// to HBM like the big-block family's (STAGE = false).  Stand-alone kernels only.
// Values of one extra edge (structure.hpp XEdge: integral-form constraint edges, control-deviation edges) on private copies of its vertices,
// in the reference's operation order (finite_differences_collocation_edges.h:149-459: 0.5 * dt * (c1 + c2) resp. c1, then *= dt; the plug-in
// stage functions as oracle/ref_driver.cpp states them).  loc[vi][c]: component c of attached vertex vi.
template <int NX, int NU>
__device__ __forceinline__ void xedge_values(const XEdge& xe, const double (&loc)[4][(NX > NU ? NX : NU)], const double* xp, const ModelParams& mp, double (&out)[4])
{
    constexpr int MC = (NX > NU) ? NX : NU;   // components of the widest attached vertex
    auto lin = [&](const double (&x)[MC], const double (&u)[MC]) {   // a^T x + b^T u - c, summed left to right
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) acc += xp[i] * x[i];
#pragma unroll
        for (int i = 0; i < NU; ++i) acc += xp[NX + i] * u[i];
        return acc - xp[NX + NU];
    };
    auto ball = [&](const double (&x)[MC]) {   // the stage inequalities' integral term: the keep-out ball or a user state function (model.hpp stage_ineq_state)
        double v[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) v[i] = x[i];
        return stage_ineq_state<NX>(mp.ineq_id, v, mp.ineq);
    };
    out[0] = out[1] = out[2] = out[3] = 0.0;
    switch (xe.kind) {
        case EK_XI_INEQ: {
            const double c1 = ball(loc[0]);
            if (xe.nverts == 4) { const double c2 = ball(loc[2]); out[0] = 0.5 * loc[3][0] * (c1 + c2); }
            else { out[0] = c1; out[0] *= loc[2][0]; }
            break;
        }
        case EK_XI_EQ_LEFT: out[0] = lin(loc[0], loc[1]); out[0] *= loc[2][0]; break;
        case EK_XI_EQ_ROW: { const double e1 = lin(loc[0], loc[1]), e2 = lin(loc[2], loc[1]); out[0] = 0.5 * loc[3][0] * (e1 + e2); break; }
        case EK_U_INEQ: {   // the stage inequalities' non-integral control term c(u_k): a user function (csrc/stage_functions/), parameters behind the rate limits'
            double v[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) v[i] = loc[0][i];
            out[0] = stage_ineq_control<NU>(mp.ineq_ctrl_id, v, xp + NX + 2 * NU + 1);
            break;
        }
        default:   // EK_CTRL_DEV: ((u_k - u_prev) / dt_prev)^2 - r_max^2 per control
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const double dd = (loc[0][i] - loc[1][i]) / loc[2][0];
                out[i] = dd * dd - xp[NX + NU + 1 + i] * xp[NX + NU + 1 + i];
            }
            break;
    }
}

// XE: the descriptor has integral-form constraint edges / control-deviation edges (SweepParams::xedges): one lane per such edge evaluates it
// generically (values before the chi2 reduction, central-difference blocks with the other Jacobian entries).  Stand-alone kernels only.
template <int DYN, int DEFECT, bool FUSED, bool DENSE = false, bool LONG = false, int THREADS = SWEEP_THREADS, bool XE = false, bool NOJAC = false>
__device__ __forceinline__ void sweep_body(const SweepParams& p, const int mode, int32_t* const active_count, LmState* const st, double* xs, double* red, double* cs, double* jst, const int inst, const int tid, const bool xs_ready = false,
                                           StageKeep<Dynamics<DYN>::NX, Dynamics<DYN>::NU>* const keep = nullptr, const bool in_loop = false, int* const sc_out = nullptr)
{
    using Dy          = Dynamics<DYN>;
    constexpr int NX  = Dy::NX;
    constexpr int NU  = Dy::NU;
    constexpr int S   = NX + NU;
    constexpr int W   = S + NX;  // local vertex values of a defect edge: x1 u1 x2
    constexpr int NC  = Dy::NC;
    constexpr bool CACHED = DefectTraits<DEFECT>::cached;
    constexpr bool STAGE  = (NX <= 4) && !LONG;  // (= jacobian_staged_in_lds(nx, N): LONG is N > 256) small models: Jacobian assembled in LDS and streamed out; big ones / long horizons: stored column by column
    double* js  = p.jac + (size_t)inst * p.nnz_pad;  // Jacobian values of this instance (HBM)
    if constexpr (!STAGE) jst = js;
    int* flags  = reinterpret_cast<int*>(red + 8);   // [4]
    const size_t xo = (size_t)inst * p.nvs;
    CORBO_HIP_DYN_OF(dynl, p, inst)

    // Two-wave shape of the run-to-completion kernel (SC, below): the tables of this lane's stage -- component descriptors, bounds, the Jacobian offsets
    // of its defect edge -- are requested FIRST, ahead of the state hand-over and its barriers: they depend on nothing but the lane, and their memory round
    // trip (1.5 - 2 k cycles per pass when requested where they are used) rides under the phase's prologue.  Branch-free, indices clamped.
    constexpr bool SC_EARLY = (THREADS <= 128) && !DENSE && !LONG;   // (= SC)
    int4 pcA[SC_EARLY ? S : 1], pcB[SC_EARLY ? S : 1];
    double plo[SC_EARLY ? S : 1], pup[SC_EARLY ? S : 1];
    int scv_pre[W + 1];                  // StageCols of lane k's defect edge
    if constexpr (SC_EARLY) {
        const int4* comp4e = reinterpret_cast<const int4*>(p.comp);
        const int kb = tid, js_ = kb - p.N;
        const bool isfin = (kb == p.N - 1), reg = (kb < p.N - 1), spec = (js_ >= 0 && js_ <= NX);
        const int vspec = (js_ < NX) ? (p.N - 1) * S + js_ : p.off_dt;
#pragma unroll
        for (int c = 0; c < W + 1; ++c) scv_pre[c] = p.stage_cols[reg ? kb : 0].col[c];
        if (sc_out) {   // (the factor phase of this pass gathers lane k's defect block through the same offsets: handed over in registers, not loaded a second time)
#pragma unroll
            for (int c = 0; c < W + 1; ++c) sc_out[c] = scv_pre[c];
        }
#pragma unroll
        for (int e = 0; e < S; ++e) {   // (an absent slot fetches component 0 and is ignored; slot 0 of a special lane: its component)
            const bool on = reg || (isfin && e < NX);   // (the last block: x_f, no controls)
            const int vc = on ? kb * S + e : ((spec && e == 0) ? vspec : 0);
            pcA[e] = comp4e[2 * vc]; pcB[e] = comp4e[2 * vc + 1]; plo[e] = p.lb[xo + vc]; pup[e] = p.ub[xo + vc];
        }
    }

    const double* xsrc = p.x + xo;
    double* vout       = p.values0 + (size_t)inst * p.m_pad;
    int vsel           = 0;   // which half of the two-buffer arrays (values0 / values1, xe0) this evaluation writes
    if (mode == 3) {
        const int done = st->done, no_trial = st->no_trial, vbuf = st->vbuf;
        lds_barrier();  // everybody has read the state before lane 0 may change it
        if (done) return;
        if (no_trial) {  // |delta| <= eps2 -> stop = true, the do-while ends without a trial step (:151-154,215)
            if (tid == 0) {
                // The reference's outer loop does not look at `stop` (:129): every remaining iteration adds mu to the diagonal of the SAME H again
                // (:135-138 -- mu itself, the right-hand side, the iterate and the values are untouched on this branch), factorises, solves and finds
                // |delta| <= eps2 again: for a positive definite H the norm of (H + s I)^-1 rhs only shrinks as s grows.  None of those
                // iterations changes anything the caller sees -- iterate, chi2, values, rho, status -- so they are COUNTED here, not computed
                // (one factorisation each), once the step is below eps2 by a factor of two (rounding cannot lift a later one above eps2 then).
                // A warm-started moving-horizon solve spends most of its iterations on this branch.
                int k = st->k;
                // (mu > 0: the argument needs H + s I positive definite with s growing; an all-zero Jacobian gives mu = 0 and is computed)
                if (p.ff_converged && st->mu > 0 && st->dnorm <= 0.5 * LM_EPS2 && k + 1 < p.iterations) {
                    const int rest = p.iterations - 1 - k;
                    st->n_fact += rest;
                    st->pad[1] += rest;   // corbo_hip_stats.counted_iterations: these iterations / factorisations were not executed
                    st->mu_acc += rest * st->mu;
                    k += rest;
                }
                const bool fin = lm_end_outer(st, st->last_sq, st->rho, k, p.iterations);
                if (p.chi2) p.chi2[inst] = st->chi2_old;
                if (!fin && active_count) atomicAdd(active_count, 1);
            }
            return;
        }
        xsrc = p.xt + xo;
        vout = (vbuf ? p.values0 : p.values1) + (size_t)inst * p.m_pad;  // the buffer NOT paired with the resident J
        vsel = vbuf ? 0 : 1;
    }

    SWEEP_STAMP(0);
    // ---- stage vertex values in LDS (coalesced 16-byte loads), unless the factor phase of the same launch left its trial iterate
    //      there (run-to-completion kernel)
    if constexpr (FUSED) {
        if (mode == 2 && p.x_init) {   // re-armed solve: the start is the shadow copy of the uploaded iterates, which becomes the accepted iterate here
            const double2* x0s = reinterpret_cast<const double2*>(p.x_init + xo);
            double2* xacc      = reinterpret_cast<double2*>(p.x + xo);
            for (int i = tid; i < p.nvs / 2; i += THREADS) { const double2 v = x0s[i]; reinterpret_cast<double2*>(xs)[i] = v; xacc[i] = v; }
        }
        else if (!xs_ready)
            for (int i = tid; i < p.nvs / 2; i += THREADS)
                reinterpret_cast<double2*>(xs)[i] = reinterpret_cast<const double2*>(xsrc)[i];
    }
    else
    if (!xs_ready)
        for (int i = tid; i < p.nvs / 2; i += THREADS)
            reinterpret_cast<double2*>(xs)[i] = reinterpret_cast<const double2*>(xsrc)[i];
    double xr[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xr[i] = p.refvec ? p.refvec[xo + (size_t)(p.N - 1) * S + i] : p.xref[(size_t)inst * CORBO_HIP_MAX_NX + i];
    // big models (stand-alone kernel): weight and reference of a component by its slot, in LDS in front of the LM state.  Selecting them out of the
    // kernel arguments costs a compare / select chain over NX entries per component, and the 40 scalars it runs over do not fit the scalar
    // register file next to everything else (the compiler parked them in vector-register lanes: 719 v_readlane in the component phase of the
    // quadrotor's kernel, 600 instructions per component -- measured: 33 k of the 52 k cycles of a residual evaluation)
    constexpr bool ROLE_TAB = (NX > 4) && !FUSED;
    double* const role = reinterpret_cast<double*>(st) - SWEEP_ROLE_TABLE;   // [0, S): sqrt(Q_ii) | sqrt(R_jj); [S, S + NX): sqrt(Qf_ii); [S + NX, S + 2 NX): xref
    if constexpr (ROLE_TAB) {
        if (tid < S) role[tid] = (tid < NX) ? p.mp.sq[tid < NX ? tid : 0] : p.mp.sr[tid < NX ? 0 : tid - NX];
        if (tid < NX) { role[S + tid] = p.mp.sqf[tid]; role[S + NX + tid] = p.refvec ? p.refvec[xo + (size_t)(p.N - 1) * S + tid] : p.xref[(size_t)inst * CORBO_HIP_MAX_NX + tid]; }
    }
    lds_barrier();
    SWEEP_STAMP(1);
    // ---- stacked residual (LevenbergMarquardtSparse::computeValues, :222-246)
    constexpr double delta     = 1e-9;
    constexpr double neg2delta = -2 * delta;
    constexpr double scalar    = 1.0 / (2 * delta);
    double sq_acc = 0.0;
    // (a) per vertex component: its cost row (state / control / final-state / dt cost edges are all diagonal in the component)
    //     and its bound row (computeDistanceFiniteCombinedBounds, hyper_graph_optimization_problem_base.cpp:291-315).  The
    //     component's entries of the Jacobian (cost block column, bound row) are functions of the same few numbers; for the models
    //     staged in LDS they are written here as well -- ahead of the accept / reject decision, harmless, the staging area only
    //     reaches HBM when the Jacobian is due.
    const bool jac_with_values = STAGE && (mode != 0);
    // role of component v: index c inside its cost edge, dimension of that edge, weight and reference (branch-free).  Edges
    // without a reference (control cost, dt cost) use ref = 0: w * (x - 0) is w * x exactly.
    // residual entries: streaming stores in the stand-alone kernel (consumed by a later launch), normal ones in the fused kernel
    auto put_value = [&](int row, double val) {
        if constexpr (FUSED || NX > 4) vout[row] = val;
        else __builtin_nontemporal_store(val, &vout[row]);
    };
    constexpr int DM = (NX > NU) ? NX : NU;
    const int wdm = (DENSE && p.mp.wdense) ? p.mp.wdense_mask : 0;   // non-diagonal weights: bit 0 Q, bit 1 R, bit 2 Qf (uniform; 0 for every diagonal problem)
    // differences x_j - ref_j of the whole vertex that component v (index c, class cls: 0 state, 1 control, 2 final state) belongs to
    auto vertex_diffs = [&](int v, int c, int dim, int cls, double (&xd)[DM]) {
#pragma unroll
        for (int j = 0; j < DM; ++j) {
            double rj = 0.0;
            if (cls != 1) {
#pragma unroll
                for (int i = 0; i < NX; ++i) rj = (j == i) ? xr[i] : rj;
                if (p.refvec && j < dim) rj = p.refvec[xo + v - c + j];
            }
            xd[j] = (j < dim) ? xs[v - c + j] - rj : 0.0;
        }
    };
    auto comp_role = [&](int v, int& c, int& dim, double& w, double& ref, bool& fin) {
        const bool is_dt  = (v == p.off_dt);
        const bool is_fin = !is_dt && v >= (p.N - 1) * S;
        const int cs_     = is_fin ? v - (p.N - 1) * S : v % S;
        const bool is_u   = !is_dt && !is_fin && cs_ >= NX;
        const int cu      = cs_ - NX;
        double wq = 0.0, wf = 0.0, wr = 0.0, rf = 0.0;
        if constexpr (ROLE_TAB) {   // (the same numbers, read by slot)
            const int sl_ = is_dt ? 0 : cs_, sx_ = (is_dt || is_u) ? 0 : cs_;
            wq = wr = role[sl_];
            wf = role[S + sx_];
            rf = role[S + NX + sx_];
        }
        else {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const bool hit = (cs_ == i);
            wq = hit ? p.mp.sq[i] : wq;
            wf = hit ? p.mp.sqf[i] : wf;
            rf = hit ? xr[i] : rf;
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) wr = (cu == i) ? p.mp.sr[i] : wr;
        }
        w   = is_dt ? p.mp.dt_weight : (is_fin ? wf : (is_u ? wr : wq));
        ref = (is_dt || is_u) ? 0.0 : rf;
        // time-varying state references (ReferenceTrajectoryInterface::getReferenceCached(k), quadratic_cost.cpp:100-119): one per component.
        // (Controls keep the zero reference: the reference's least-squares control term with a non-zero uref is not a function of the
        // control vector -- quadratic_cost.cpp:160-163 assigns the scalar ud^T R^(1/2) ud to the nu-vector.)
        if (p.refvec && !is_dt && !is_u) ref = p.refvec[xo + v];
        dim = is_dt ? 1 : (is_u ? NU : NX);
        c   = is_dt ? 0 : (is_u ? cu : cs_);
        fin = is_fin;
    };
    auto comp_jac = [&](int v, const CompInfo& ci, double xv, double l, double u, int c, int dim, double w, double ref, bool fin) {
        const int cls = (v == p.off_dt) ? 3 : (fin ? 2 : (v % S >= NX ? 1 : 0));
        if (DENSE && wdm && cls < 3 && ((wdm >> cls) & 1)) {
            // dense weight: column c of the (upper-triangular) block U, every row by central differences of the edge's own value
            if (!ci.fixed && ci.cost_joff >= 0) {
                const double* U = p.mp.wdense + 16 * cls;
                double xd[DM];
                vertex_diffs(v, c, dim, cls, xd);
                double refc = 0.0;   // the component's own reference (the perturbed difference is (x_c +- delta) - ref_c)
                if (cls != 1) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) refc = (c == i) ? xr[i] : refc;
                    if (p.refvec) refc = p.refvec[xo + v];
                }
                const double a = xv + delta, b = a + neg2delta;
                const int col0 = ci.cost_joff - c;
#pragma unroll
                for (int r = 0; r < DM; ++r) {
                    if (r < dim) {
                        double x2[DM], x1[DM];
#pragma unroll
                        for (int j = 0; j < DM; ++j) { x2[j] = (j == c) ? a - refc : xd[j]; x1[j] = (j == c) ? b - refc : xd[j]; }
                        jst[col0 + r] = scalar * (dense_weight_row<DM>(U, r, dim, x2) - dense_weight_row<DM>(U, r, dim, x1));
                    }
                }
            }
        }
        else if (!ci.fixed && ci.cost_joff >= 0) {  // central difference of the diagonal cost block (edge_interface.cpp:55-96)
            const double a = xv + delta, b = a + neg2delta;
            const double dv = scalar * (w * (a - ref) - w * (b - ref));
            const int col0  = ci.cost_joff - c;
#pragma unroll
            for (int r = 0; r < DM; ++r)
                if (r < dim) jst[col0 + r] = (r == c) ? dv : 0.0;  // untouched rows: scalar * (e - e) = 0
            if (!fin && ci.cost2_joff >= 0) jst[ci.cost2_joff] = dv;   // duplicated MinimumTime dt edge (nlp_functions.cpp:91-107)
        }
        if (fin && !ci.fixed && ci.cost2_joff >= 0) {   // Terminal[Partial]EqualityConstraint x_f - xref: one row per (active) component, scaled by w_eq (:1552)
            const double a = xv + delta, b = a + neg2delta;
            const double dv = (scalar * ((a - ref) - (b - ref))) * p.w_eq;
            // the component's row inside the edge (TerminalPartialEqualityConstraint: active components only; an inactive one has no row and
            // its column is explicit zeros -- cost2_joff is then the column's start)
            const int idx  = (ci.cost2_row >= 0) ? ci.cost2_row - p.fin_eq_row0 : -1;
            const int col0 = ci.cost2_joff - (idx >= 0 ? idx : 0);
#pragma unroll
            for (int r = 0; r < NX; ++r)
                if (r < p.fin_eq_dim) jst[col0 + r] = (r == idx) ? dv : 0.0;
        }
        if (ci.bnd_joff >= 0) jst[ci.bnd_joff] = (xv < l) ? -p.w_b : ((xv > u) ? p.w_b : 0.0);  // :1721-1752
    };
    auto comp_values = [&](int v, const int4& ca, const int4& cb, double l, double u) {
        const CompInfo ci{ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
        const double xv = xs[v];
        int c, dim;
        double w, ref;
        bool fin;
        comp_role(v, c, dim, w, ref, fin);
        if (ci.cost_row >= 0) {
            double val = w * (xv - ref);
            if (DENSE && wdm) {
                const int cls = (v == p.off_dt) ? 3 : (fin ? 2 : (v % S >= NX ? 1 : 0));
                if (cls < 3 && ((wdm >> cls) & 1)) {
                    double xd[DM];
                    vertex_diffs(v, c, dim, cls, xd);
                    val = dense_weight_row<DM>(p.mp.wdense + 16 * cls, c, dim, xd, ((ci.cost_row - c) & 1) != 0);   // (the edge's first row)
                }
            }
            put_value(ci.cost_row, val);
            sq_acc += val * val;
            if (!fin && ci.cost2_row >= 0) { put_value(ci.cost2_row, val); sq_acc += val * val; }
        }
        if (fin && ci.cost2_row >= 0) {   // TerminalEqualityConstraint row (equality section: times w_eq)
            const double val = (xv - ref) * p.w_eq;
            put_value(ci.cost2_row, val);
            sq_acc += val * val;
        }
        if (ci.bnd_row >= 0) {
            double val = (xv < l) ? l - xv : ((xv > u) ? xv - u : 0.0);
            val *= p.w_b;
            put_value(ci.bnd_row, val);
            sq_acc += val * val;
        }
        if (jac_with_values) comp_jac(v, ci, xv, l, u, c, dim, w, ref, fin);
    };
    const int4* comp4 = reinterpret_cast<const int4*>(p.comp);
    // Two-wave shape of the run-to-completion kernel: STAGE-CENTRIC component pass.  Lane k owns the components of block k (x_k, u_k; the last
    // block: x_f) and one lane the dt component, the component's role (weight, reference, index inside its cost edge) is a compile-time
    // property of its slot -- no per-component selection -- and the pass is ordered  loads -> arithmetic -> stores: the component rounds
    // of the other shapes interleave global stores (residual rows) with global loads (the next round's descriptors), and on gfx9 a wait for
    // a load issued behind stores waits for the stores' acknowledgements as well (one memory round trip per round: 7.4 k cycles for the
    // four rounds of the headline instance, measured with the phase stamps).
    constexpr bool SC = (THREADS <= 128) && !DENSE && !LONG;
    constexpr bool LEAN = SC && FUSED;   // the diagonal rows of the stage blocks' components are not stored: they reach the factor phase in registers
    StageKeep<NX, NU> kt;                // (LEAN) this evaluation's rows; they become *keep when the step is accepted and the Jacobian refreshed
    if constexpr (SC) {
        // regular slots: lane k < N - 1, slot e of block k (e < NX: state component e, else control component e - NX);
        // special slots: lane N + j -- j < NX the component j of x_f (final cost / terminal equality), j = NX the dt component
        const int kb     = tid;
        const bool isfin = (kb == p.N - 1);
        const bool reg   = (kb < p.N - 1);
        const int js     = kb - p.N;
        const bool spec  = (js >= 0 && js <= NX);
        const int vspec  = (js < NX) ? (p.N - 1) * S + js : p.off_dt;
        // (descriptors, bounds and the defect edge's Jacobian offsets: requested at the top of the phase, see SC_EARLY -- also ahead of the accepted
        //  iterate's copy to HBM: behind those stores the first use of a later load would wait for their acknowledgements as well)
        int4 (&cA)[S] = pcA, (&cB)[S] = pcB;
        double (&lo)[S] = plo, (&up)[S] = pup;
        double rv[NX];
#pragma unroll
        for (int e = 0; e < NX; ++e) rv[e] = xr[e];
        double rspec = 0.0;
        if (p.refvec) {
#pragma unroll
            for (int e = 0; e < NX; ++e) rv[e] = p.refvec[xo + ((reg || isfin) ? kb * S + e : 0)];
            rspec = p.refvec[xo + ((spec && js < NX) ? vspec : 0)];
        }
        if constexpr (CACHED) {  // state-only part of the dynamics, once per grid state
            for (int k = tid; k < p.N; k += THREADS) {
                double c[NC];
                Dy::prepare(xs + k * S, dynl, c);
#pragma unroll
                for (int i = 0; i < NC; ++i) cs[k * NC + i] = c[i];
            }
        }
        SWEEP_STAMP(2);
#pragma unroll
        for (int e = 0; e < S; ++e) {
            constexpr double zero = 0.0;
            const bool is_u   = (e >= NX);
            const bool on     = reg || (isfin && !is_u);
            const double w    = is_u ? p.mp.sr[is_u ? e - NX : 0] : (isfin ? p.mp.sqf[is_u ? 0 : e] : p.mp.sq[is_u ? 0 : e]);   // (compile-time slot: scalar operands)
            const double ref  = is_u ? zero : rv[is_u ? 0 : e];
            const int c       = is_u ? e - NX : e;
            const int dim     = is_u ? NU : NX;
            const double xv   = xs[on ? kb * S + e : 0];
            const int fixed = cA[e].x, cost_joff = cA[e].z, cost_row = cA[e].w, bnd_joff = cB[e].x, bnd_row = cB[e].y;
            double val, dv, bval, bent;
            diagonal_rows(xv, w, ref, lo[e], up[e], p.w_b, val, dv, bval, bent);
            const bool pc = on && cost_row >= 0, pb = on && bnd_row >= 0;
            sq_acc += (pc ? val * val : 0.0) + (pb ? bval * bval : 0.0);
            if constexpr (LEAN) { kt.vc[e] = val; kt.ac[e] = dv; kt.vb[e] = bval; kt.ab[e] = bent; }   // handed to the factor phase in registers (below)
            else {
                if (pc) put_value(cost_row, val);
                if (pb) put_value(bnd_row, bval);
                if (jac_with_values) {
                    if (on && !fixed && cost_joff >= 0) {
                        const int col0 = cost_joff - c;
#pragma unroll
                        for (int r = 0; r < DM; ++r)
                            if (r < dim) jst[col0 + r] = (r == c) ? dv : 0.0;   // untouched rows: scalar * (e - e) = 0
                    }
                    if (on && bnd_joff >= 0) jst[bnd_joff] = bent;
                }
            }
            (void)fixed; (void)cost_joff; (void)bnd_joff; (void)c; (void)dim;
        }
        if (spec) {   // the x_f components and dt: every row kind a component can carry (second rows: terminal equality, duplicated dt edge)
            const bool fin = (js < NX);
            double wf = p.mp.dt_weight, reff = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                wf   = (js == i) ? p.mp.sqf[i] : wf;
                reff = (js == i) ? xr[i] : reff;
            }
            if (p.refvec && fin) reff = rspec;
            const CompInfo ci{cA[0].x, cA[0].y, cA[0].z, cA[0].w, cB[0].x, cB[0].y, cB[0].z, cB[0].w};
            const double xv = xs[vspec];
            if (!fin && ci.cost_row >= 0) {   // (the x_f components' cost and bound rows: the regular slots of lane N - 1)
                const double val = wf * (xv - reff);
                put_value(ci.cost_row, val);
                sq_acc += val * val;
                if (ci.cost2_row >= 0) { put_value(ci.cost2_row, val); sq_acc += val * val; }
            }
            if (fin && ci.cost2_row >= 0) {   // TerminalEqualityConstraint row (equality section: times w_eq)
                const double val = (xv - reff) * p.w_eq;
                put_value(ci.cost2_row, val);
                sq_acc += val * val;
            }
            if (!fin && ci.bnd_row >= 0) {
                double val = (xv < lo[0]) ? lo[0] - xv : ((xv > up[0]) ? xv - up[0] : 0.0);
                val *= p.w_b;
                put_value(ci.bnd_row, val);
                sq_acc += val * val;
            }
            if (jac_with_values) comp_jac(vspec, ci, xv, lo[0], up[0], fin ? js : 0, fin ? NX : 1, wf, reff, fin);
        }
        if constexpr (CACHED) lds_barrier();
    }
    else {
        // Work split of the residual (horizons up to 128 stages, i.e. when half of the workgroup holds one lane per stage): waves 0-1
        // are the stage lanes (dynamics caches, one round of components, then the defects), waves 2-3 take the other rounds of
        // components (cost / bound rows) meanwhile -- the halves run side by side on different SIMDs.  Longer horizons: every lane does
        // both kinds of work, in sequence.
        // (THREADS = 192, the three-wave shape of the run-to-completion kernel: waves 0-1 stage lanes, wave 2 the component rounds)
        constexpr int SLANES = (THREADS == 192) ? 128 : THREADS / 2;   // stage lanes of the split
        const bool split   = (THREADS > 128) && (p.N <= SLANES);
        const int cstr     = split ? THREADS - SLANES : THREADS;   // component stride of the lanes that do several rounds
        const bool cworker = !split || (tid >= SLANES);
        // descriptors and bounds of the first two rounds of components: requested now, consumed below (straight-line code: the waits
        // are exact, nothing waits for the write acknowledgements)
        int4 ca0 = make_int4(1, -1, -1, -1), cb0 = make_int4(-1, -1, -1, -1), ca1 = ca0, cb1 = cb0;
        double l0 = 0.0, u0 = 0.0, l1 = 0.0, u1 = 0.0;
        const int vend = p.off_dt + 1;
        const int v0 = tid, v1 = cworker ? v0 + cstr : vend, v2 = cworker ? v1 + cstr : vend;
        if (v0 < vend) { ca0 = comp4[2 * v0]; cb0 = comp4[2 * v0 + 1]; l0 = p.lb[xo + v0]; u0 = p.ub[xo + v0]; }
        if (v1 < vend) { ca1 = comp4[2 * v1]; cb1 = comp4[2 * v1 + 1]; l1 = p.lb[xo + v1]; u1 = p.ub[xo + v1]; }
        // two-wave shape (256 VGPRs): every lane does four rounds of components one after the other -- all four requested up front (a round that
        // fetches as it goes pays a memory round trip: 7.5 k cycles for the four rounds of the headline instance, measured)
        constexpr bool PF4 = (THREADS <= 128);
        int4 ca2 = ca0, cb2 = cb0, ca3 = ca0, cb3 = cb0;
        double l2 = 0.0, u2 = 0.0, l3 = 0.0, u3 = 0.0;
        const int w2 = v1 + cstr, w3 = w2 + cstr;
        if constexpr (PF4) {
            if (w2 < vend) { ca2 = comp4[2 * w2]; cb2 = comp4[2 * w2 + 1]; l2 = p.lb[xo + w2]; u2 = p.ub[xo + w2]; }
            if (w3 < vend) { ca3 = comp4[2 * w3]; cb3 = comp4[2 * w3 + 1]; l3 = p.lb[xo + w3]; u3 = p.ub[xo + w3]; }
        }
        if (split && cworker) {  // component lanes: rounds 0 and 1 while the stage lanes prepare their caches; round 2 requested
            if (v0 < vend) {
                comp_values(v0, ca0, cb0, l0, u0);
                if (v2 < vend) { ca0 = comp4[2 * v2]; cb0 = comp4[2 * v2 + 1]; l0 = p.lb[xo + v2]; u0 = p.ub[xo + v2]; }
            }
            if (v1 < vend) comp_values(v1, ca1, cb1, l1, u1);
        }
        if constexpr (CACHED) {  // state-only part of the dynamics, once per grid state
            for (int k = tid; k < p.N; k += THREADS) {
                double c[NC];
                Dy::prepare(xs + k * S, dynl, c);
    #pragma unroll
                for (int i = 0; i < NC; ++i) cs[k * NC + i] = c[i];
            }
            if (split && !cworker && v0 < vend) comp_values(v0, ca0, cb0, l0, u0);  // stage lanes: their one round of components
            lds_barrier();
        }
        else if (split && !cworker && v0 < vend) comp_values(v0, ca0, cb0, l0, u0);

        SWEEP_STAMP(2);
        if (split) {
            if (cworker) {   // round 2 is in the registers, later rounds (N > 100 or so) fetch as they go
                if (v2 < vend) comp_values(v2, ca0, cb0, l0, u0);
                for (int v = v2 + cstr; v < vend; v += cstr) comp_values(v, comp4[2 * v], comp4[2 * v + 1], p.lb[xo + v], p.ub[xo + v]);
            }
        }
        else {
            if (v0 < vend) comp_values(v0, ca0, cb0, l0, u0);
            if (v1 < vend) comp_values(v1, ca1, cb1, l1, u1);
            if constexpr (PF4) {
                if (w2 < vend) comp_values(w2, ca2, cb2, l2, u2);
                if (w3 < vend) comp_values(w3, ca3, cb3, l3, u3);
                for (int v = w3 + cstr; v < vend; v += cstr) comp_values(v, comp4[2 * v], comp4[2 * v + 1], p.lb[xo + v], p.ub[xo + v]);
            }
            else {
                // later rounds four at a time: all descriptors and bounds of the four requested first, then the arithmetic and the stores -- a round
                // that fetches as it goes waits for the previous round's store acknowledgements with its loads (one memory round trip per round:
                // 2.3 k cycles each, 11 such rounds on the 12-state quadrotor with N = 200, measured with the phase stamps)
                for (int v = v1 + cstr; v < vend; v += 4 * cstr) {
                    int4 qa[4], qb[4];
                    double ql[4], qu[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int vr = v + r * cstr;
                        const int vc = (vr < vend) ? vr : v;   // (clamped: an absent round re-reads the first one and is skipped below)
                        qa[r] = comp4[2 * vc]; qb[r] = comp4[2 * vc + 1]; ql[r] = p.lb[xo + vc]; qu[r] = p.ub[xo + vc];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (v + r * cstr < vend) comp_values(v + r * cstr, qa[r], qb[r], ql[r], qu[r]);
                }
            }
        }
    }
    SWEEP_STAMP(9);
    // (b) per stage: the dynamics defect (equality rows) and the stage inequality
    for (int k = tid; k < p.N - 1; k += THREADS) {
        const int base = k * S;
        double e[NX];
        if constexpr (CACHED)
            defect_eval_cached<DYN, DEFECT>(xs + base, cs + k * NC, xs + base + NX, xs + base + S, cs + (k + 1) * NC, xs[p.off_dt], dynl, e);
        else if constexpr (DEFECT == CORBO_HIP_DEFECT_RK4_SHOOTING && !STAGE) {
            // same operations as defect_eval (end state of the step, then the subtraction); the end state is kept for the stage kernel
            double ck[4][NC], xe[NX];
            rk4_end_state<DYN, false>(xs + base, xs + base + NX, xs[p.off_dt], dynl, ck, xe);
            double* xeo = p.xe0 ? p.xe0 + (((size_t)vsel * p.batch_total + inst) * p.N + k) * NX : nullptr;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                e[i] = xe[i] - xs[base + S + i];
                if (xeo) xeo[i] = xe[i];
            }
        }
        else if constexpr (DEFECT == DEFECT_SHOOTING_HIGH && !STAGE && NX > 4) {
            // (big-block family with Runge-Kutta 5 / 6 / 7: defect_eval's operations, the end state kept for the stage kernel like above)
            double xe[NX];
            rk_high_order_end_state<DYN>(xs + base, xs + base + NX, xs[p.off_dt], dynl, (int)dynl[7], xe);
            double* xeo = p.xe0 ? p.xe0 + (((size_t)vsel * p.batch_total + inst) * p.N + k) * NX : nullptr;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                e[i] = xe[i]; e[i] -= xs[base + S + i];
                if (xeo) xeo[i] = xe[i];
            }
        }
        else
            defect_eval<DYN, DEFECT>(xs + base, xs + base + NX, xs + base + S, xs[p.off_dt], dynl, e);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double val = e[i] * p.w_eq;
            // (LEAN: the factor phase takes the rows from registers, StageKeep; stored all the same when the next factor phase is another LAUNCH.  Inside the
            //  run-to-completion loop nobody reads them -- and on gfx9 the first load the factor phase waits for would wait for these stores'
            //  acknowledgements too, a memory round trip on the critical path of every pass)
            if (!(LEAN && in_loop)) put_value(p.eq_row0 + (XE ? k * p.eq_stride + p.eq_defect_off : k * NX) + i, val);
            sq_acc += val * val;
            if constexpr (LEAN) kt.r[i] = val;
        }
        {
            if (p.ineq_cols) {  // computeValuesActiveInequality (hyper_graph_optimization_problem_base.cpp:278-289)
                double xl_[NX];   // (a private copy: the function sees one address space at every call site)
#pragma unroll
                for (int i = 0; i < NX; ++i) xl_[i] = xs[base + i];
                double ci = stage_ineq_state<NX>(p.mp.ineq_id, xl_, p.mp.ineq);
                ci        = (ci < 0) ? 0.0 : ci * p.w_ineq;
                put_value(p.ineq_row0 + k * p.ineq_stride, ci);
                sq_acc += ci * ci;
            }
        }
    }
    // (c) the final-stage inequality on x_f (TerminalBall): one lane
    {
        if (p.fin_row >= 0 && tid == THREADS - 1) {
            double cf = terminal_ball<NX>(xs + (p.N - 1) * S, xr, p.mp.fin);
            cf        = (cf < 0) ? 0.0 : cf * p.w_ineq;   // computeValuesActiveInequality
            put_value(p.fin_row, cf);
            sq_acc += cf * cf;
        }
    }

    // (d) integral-form constraint edges, control-deviation edges: one lane per edge
    constexpr int XMC = (NX > NU) ? NX : NU;   // components of the widest vertex an extra edge attaches
    auto xe_load = [&](const XEdge& xe, double (&loc)[4][XMC]) {
#pragma unroll
        for (int vi = 0; vi < 4; ++vi)
#pragma unroll
            for (int c = 0; c < XMC; ++c) {
                double v = 0.0;
                if (vi < xe.nverts && c < xe.vdim[vi]) {
                    const int vo = xe.voff[vi];
                    if (vo >= 0) v = xs[vo + c];
                    else if (vo == -1) v = p.uprev[(size_t)inst * (CORBO_HIP_MAX_NU + 1) + c];            // _u_prev
                    else if (vo == -3) v = p.uprev[(size_t)inst * (CORBO_HIP_MAX_NU + 1) + CORBO_HIP_MAX_NU];   // _u_prev_dt
                    // (-2: _u_ref = the zero control reference)
                }
                loc[vi][c] = v;
            }
    };
    if constexpr (XE) {
        for (int i = tid; i < p.n_xedges; i += THREADS) {
            const XEdge xe = p.xedges[i];
            double loc[4][XMC], out[4];
            xe_load(xe, loc);
            xedge_values<NX, NU>(xe, loc, p.xparams, p.mp, out);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < xe.dim) {
                    const double val = (xe.scale == 1) ? out[j] * p.w_eq : ((out[j] < 0) ? 0.0 : out[j] * p.w_ineq);   // computeValuesActiveInequality
                    put_value(xe.row + j, val);
                    sq_acc += val * val;
                }
        }
    }
    SWEEP_STAMP(3);
    // ---- chi2 = |values|^2 and the LM trial-step decision
    int do_jac = (mode == 1 || mode == 2) ? 1 : 0;
    if (mode >= 2) {
        double ws = wave_sum(sq_acc);
        if ((tid & 63) == 0) red[tid >> 6] = ws;
        lds_barrier();
        if (tid == 0) {
            double chi2 = red[0] + red[1];
            if constexpr (THREADS >= 192) chi2 += red[2];
            if constexpr (THREADS >= 256) chi2 += red[3];
            bool fin = false;
            if (mode == 2) {  // solve() prologue (:89-127)
                st->mu = 0; st->mu_acc = 0; st->rho = 0; st->chi2_old = chi2; st->last_sq = chi2; st->den = 0; st->dnorm = 0;
                st->v = 2; st->k = 0; st->stop = 0; st->fresh = 1; st->first = 1; st->no_trial = 0;
                st->status = CORBO_HIP_SOLVER_CONVERGED;  // iterations == 0: (stop || rho <= 0) with rho = 0
                fin        = (p.iterations <= 0);
                st->done   = fin ? 1 : 0;
                st->vbuf = 0; st->inner = 0; st->n_accept = 0; st->n_reject = 0; st->n_jac = 1; st->n_res = 1; st->n_fact = 0; st->pad[0] = 0; st->pad[1] = 0;
                flags[0] = 1;
                flags[1] = 0;
                if (p.chi2) p.chi2[inst] = chi2;
            }
            else {
                const double chi2_new = chi2, chi2_old = st->chi2_old;
                const int k = st->k;
                double mu   = st->mu;
                unsigned v  = st->v;
                int stop    = st->stop;
                st->last_sq = chi2_new;
                st->n_res += 1;
                const double rho = (chi2_old - chi2_new) / st->den;  // :169
                st->rho          = rho;
                int accept = 0, refresh = 0;
                if (rho > 0 && !isnan(chi2_new) && !isinf(chi2_new)) {  // :171
                    stop   = (sqrt(chi2_old) - sqrt(chi2_new) < LM_EPS4 * sqrt(chi2_old)) ? 1 : 0;
                    accept = 1;
                    st->n_accept += 1;
                    if (!stop && k < p.iterations - 1) {  // :178-199
                        refresh            = 1;
                        // (2 rho - 1)^3 (:195, std::pow(., 3)): the cube with the rounding error of the square carried along (fma) -- within
                        // half an ulp of the exact cube like the math library's pow, a dozen instructions instead of its few hundred on the
                        // one lane every other lane of the workgroup waits for
                        const double t_ = 2 * rho - 1, t2_ = t_ * t_;
                        const double cube = __builtin_fma(t2_, t_, __builtin_fma(t_, t_, -t2_) * t_);
                        const double alpha = fmin(2. / 3., 1 - cube);
                        const double scale = fmax(1. / 3., alpha);
                        mu *= scale;
                        v         = 2;
                        st->fresh = 1;
                        st->vbuf ^= 1;  // the residual just written pairs with the Jacobian about to be written
                        st->n_jac += 1;
                    }
                    st->chi2_old = chi2_new;
                    if (p.chi2) p.chi2[inst] = chi2_new;
                }
                else {  // :204-213
                    st->n_reject += 1;
                    mu = mu * v;
                    v  = 2 * v;
                }
                st->mu   = mu;
                st->v    = v;
                st->stop = stop;
                const bool cont = (rho <= 0) && !stop;  // :215
                if (cont && st->inner >= LM_MAX_INNER) {
                    // guarded deviation from the reference (which would keep rejecting and multiplying the damping): the instance stops
                    // at its last accepted iterate and says so -- status ERROR, counted in corbo_hip_stats.inner_loop_cuts
                    st->done = 1; st->status = CORBO_HIP_SOLVER_ERROR; st->pad[0] = 1;
                    fin = true;
                }
                else if (!cont) fin = lm_end_outer(st, chi2_new, rho, k, p.iterations);
                flags[0] = refresh;
                flags[1] = accept;
            }
            if (!fin && active_count) atomicAdd(active_count, 1);
        }
        lds_barrier();
        do_jac = flags[0];
        if (mode == 3 && flags[1]) {  // accepted: the trial iterate becomes the iterate (discardBackupParameters :176)
            double* xdst = p.x + xo;
            for (int i = tid; i < p.nvs / 2; i += THREADS)
                reinterpret_cast<double2*>(xdst)[i] = reinterpret_cast<const double2*>(xs)[i];
        }
    }
    SWEEP_STAMP(4);
    if constexpr (LEAN) {
        if (keep && do_jac) *keep = kt;   // the residual just evaluated pairs with the Jacobian about to be written (like LmState::vbuf ^= 1)
    }
    if (!do_jac || (p.skip_jac && mode >= 2)) return;
    if constexpr (NOJAC) return;   // residual-only instantiation (big-block family inside the LM loop: the Jacobian is big_stage_kernel's): no Jacobian code, a quarter of the registers

    // ---- combined sparse Jacobian (computeCombinedSparseJacobian, hyper_graph_optimization_problem_edge_based.cpp:1480-1753),
    //      central differences exactly as BaseEdge::computeJacobian (edge_interface.cpp:55-96):
    //      x_i += delta -> v2 ; x_i += -2 delta -> v1 ; col = (1/(2 delta)) (v2 - v1), on private copies of the edge's vertices.
    //      Values are assembled in the LDS staging area and streamed out with coalesced 16-byte stores at the end.
    const double dt0 = xs[p.off_dt];
    // (1) dynamics-defect blocks.  Lane = (stage k, column group g): the perturbed component is a compile-time index inside the
    //     lane's loop, so there is no per-lane selection of what to perturb, no divergence, and for the cached defects only the
    //     parts of the defect that depend on the perturbed component are re-evaluated (the others are bit-identical anyway).
    {
        constexpr int NU0 = (NU + 1) / 2;  // group 0: x_k, u_k[0,NU0) ; group 1: u_k[NU0,NU), x_{k+1}, dt
        // Column sets of a lane (wave-uniform): bit 0 the x_k columns, bit 1 u_k[0,NU0), bit 2 u_k[NU0,NU), bit 3 x_{k+1} and dt.
        //   256 threads: two lanes per stage (groups 0 / 1 = bits 0|1 / 2|3);  128 threads: one lane per stage, every column;
        //   192 threads: wave 0 the x_k columns, wave 1 the u_k columns, wave 2 x_{k+1} / dt, 64 stages per round.
        constexpr int JL = (THREADS == 256) ? 128 : (THREADS == 192 ? 64 : THREADS);   // lanes (= stages per round) of a column group
        const int jw = tid / JL;
        const unsigned cset = (THREADS == 256) ? (jw == 0 ? 0x3u : 0xCu) : (THREADS == 192 ? (jw == 0 ? 0x1u : (jw == 1 ? 0x6u : 0x8u)) : 0xFu);
        const bool cs_x1 = cset & 1u, cs_u0 = cset & 2u, cs_u1 = cset & 4u, cs_x2 = cset & 8u;
        for (int k = tid % JL; k < p.N - 1; k += JL) {
            const int base  = k * S;
            const int* sc   = SC ? scv_pre : p.stage_cols[k].col;   // (SC: one lane per stage, k = tid: the offsets are in registers already)
            double x1[NX], u1[NU], x2[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) { x1[i] = xs[base + i]; x2[i] = xs[base + S + i]; }
#pragma unroll
            for (int i = 0; i < NU; ++i) u1[i] = xs[base + NX + i];
            auto emit = [&](int jo, const double* e2, const double* e1) {
#pragma unroll
                for (int r = 0; r < NX; ++r) jst[jo + r] = (scalar * (e2[r] - e1[r])) * p.w_eq;  // :1552
            };
            if constexpr (CACHED) {
                using DP = DefectParts<DEFECT>;
                double c1[NC], c2[NC], q[NX], f1[NX], f2[NX];
#pragma unroll
                for (int i = 0; i < NC; ++i) { c1[i] = cs[k * NC + i]; c2[i] = cs[(k + 1) * NC + i]; }
#pragma unroll
                for (int i = 0; i < NX; ++i) q[i] = (x2[i] - x1[i]) / dt0;
                Dy::eval(x1, c1, u1, dynl, f1);
                Dy::eval(x2, c2, u1, dynl, f2);
                if (cs_x1) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) {  // d/d x_k[i]: q_i and f(x1,u1) change
                        const int jo = sc[i];
                        if (jo < 0) continue;
                        double e[2][NX];
                        double xa = x1[i];
#pragma unroll
                        for (int side = 0; side < 2; ++side) {
                            xa += (side == 0) ? delta : neg2delta;
                            double xp[NX], cp[NC], fp[NX], qp[NX];
#pragma unroll
                            for (int r = 0; r < NX; ++r) { xp[r] = (r == i) ? xa : x1[r]; qp[r] = q[r]; fp[r] = f1[r]; }
                            qp[i] = (x2[i] - xa) / dt0;
                            if constexpr (DP::uses_f1) {
                                if ((Dy::CACHE_XMASK >> i) & 1u) Dy::prepare(xp, dynl, cp);
                                else {
#pragma unroll
                                    for (int r = 0; r < NC; ++r) cp[r] = c1[r];
                                }
                                Dy::eval(xp, cp, u1, dynl, fp);
                            }
                            defect_combine<NX, DEFECT>(qp, fp, f2, e[side]);
                        }
                        emit(jo, e[0], e[1]);
                    }
                }
#pragma unroll
                for (int j = 0; j < NU; ++j) {  // d/d u_k[j]: both dynamics evaluations change
                    if (!((j < NU0) ? cs_u0 : cs_u1)) continue;
                    const int jo = sc[NX + j];
                    if (jo < 0) continue;
                    double e[2][NX];
                    double ua = u1[j];
#pragma unroll
                    for (int side = 0; side < 2; ++side) {
                        ua += (side == 0) ? delta : neg2delta;
                        double up[NU], g1[NX], g2[NX];
#pragma unroll
                        for (int r = 0; r < NU; ++r) up[r] = (r == j) ? ua : u1[r];
#pragma unroll
                        for (int r = 0; r < NX; ++r) { g1[r] = f1[r]; g2[r] = f2[r]; }
                        if constexpr (DP::uses_f1) Dy::eval(x1, c1, up, dynl, g1);
                        if constexpr (DP::uses_f2) Dy::eval(x2, c2, up, dynl, g2);
                        defect_combine<NX, DEFECT>(q, g1, g2, e[side]);
                    }
                    emit(jo, e[0], e[1]);
                }
                if (cs_x2) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) {  // d/d x_{k+1}[i]: q_i and f(x2,u1) change
                        const int jo = sc[S + i];
                        if (jo < 0) continue;
                        double e[2][NX];
                        double xa = x2[i];
#pragma unroll
                        for (int side = 0; side < 2; ++side) {
                            xa += (side == 0) ? delta : neg2delta;
                            double xp[NX], cp[NC], fp[NX], qp[NX];
#pragma unroll
                            for (int r = 0; r < NX; ++r) { xp[r] = (r == i) ? xa : x2[r]; qp[r] = q[r]; fp[r] = f2[r]; }
                            qp[i] = (xa - x1[i]) / dt0;
                            if constexpr (DP::uses_f2) {
                                if ((Dy::CACHE_XMASK >> i) & 1u) Dy::prepare(xp, dynl, cp);
                                else {
#pragma unroll
                                    for (int r = 0; r < NC; ++r) cp[r] = c2[r];
                                }
                                Dy::eval(xp, cp, u1, dynl, fp);
                            }
                            defect_combine<NX, DEFECT>(qp, f1, fp, e[side]);
                        }
                        emit(jo, e[0], e[1]);
                    }
                    const int jo = sc[S + NX];
                    if (jo >= 0) {  // d/d dt (free-dt grids): every q changes
                        double e[2][NX];
                        double da = dt0;
#pragma unroll
                        for (int side = 0; side < 2; ++side) {
                            da += (side == 0) ? delta : neg2delta;
                            double qp[NX];
#pragma unroll
                            for (int r = 0; r < NX; ++r) qp[r] = (x2[r] - x1[r]) / da;
                            defect_combine<NX, DEFECT>(qp, f1, f2, e[side]);
                        }
                        emit(jo, e[0], e[1]);
                    }
                }
            }
            else if constexpr (DEFECT == CORBO_HIP_DEFECT_RK4_SHOOTING) {
                // e = RK4(x_k, u_k, dt) - x_{k+1} (multiple_shooting_edges.h:125-134).  Per column the reference re-integrates; here
                //  * an x_{k+1} column reuses the end state of the unperturbed step (same inputs, same bits) and only redoes the
                //    subtraction,
                //  * an (x_k, u_k) column that cannot change what prepare() sees at any Runge-Kutta stage (Dynamics<>::
                //    RK4_CACHE_DEP_COLS) re-integrates with the caches of the unperturbed step (quadrotor: 7 of 16 columns skip
                //    24 sin/cos evaluations each),
                //  * the others re-integrate in full.
                double loc[W], ck[4][NC], xe0[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) { loc[i] = x1[i]; loc[S + i] = x2[i]; }
#pragma unroll
                for (int i = 0; i < NU; ++i) loc[NX + i] = u1[i];
                rk4_end_state<DYN, false>(loc, loc + NX, dt0, dynl, ck, xe0);
#pragma unroll
                for (int c = 0; c < S; ++c) {
                    const bool mine = (THREADS == 256) ? ((((Dy::RK4_GROUP1_COLS >> c) & 1u) != 0u) == (jw == 1)) : ((c < NX) ? cs_x1 : ((c - NX < NU0) ? cs_u0 : cs_u1));
                    const int jo    = sc[c];
                    if (!mine || jo < 0) continue;
                    double e[2][NX];
                    const double keep = loc[c];
#pragma unroll
                    for (int side = 0; side < 2; ++side) {
                        loc[c] += (side == 0) ? delta : neg2delta;
                        double xe[NX];
                        if (((Dy::RK4_CACHE_DEP_COLS >> c) & 1u) != 0u) {   // (compile-time after unrolling)
                            double ct[4][NC];
                            rk4_end_state<DYN, false>(loc, loc + NX, dt0, dynl, ct, xe);
                        }
                        else rk4_end_state<DYN, true>(loc, loc + NX, dt0, dynl, ck, xe);
#pragma unroll
                        for (int r = 0; r < NX; ++r) e[side][r] = xe[r] - x2[r];
                    }
                    loc[c] = keep;
                    emit(jo, e[0], e[1]);
                }
#pragma unroll
                for (int c = S; c < W; ++c) {
                    const bool mine = (THREADS == 256) ? ((((Dy::RK4_GROUP1_COLS >> c) & 1u) != 0u) == (jw == 1)) : cs_x2;
                    const int jo    = sc[c];
                    if (!mine || jo < 0) continue;
                    double e[2][NX];
                    double xa = x2[c - S];
#pragma unroll
                    for (int side = 0; side < 2; ++side) {
                        xa += (side == 0) ? delta : neg2delta;
#pragma unroll
                        for (int r = 0; r < NX; ++r) e[side][r] = xe0[r] - ((r == c - S) ? xa : x2[r]);
                    }
                    emit(jo, e[0], e[1]);
                }
                const int jo = sc[S + NX];
                if (cs_x2 && jo >= 0) {  // free dt: every stage changes
                    double e[2][NX];
                    double da = dt0;
#pragma unroll
                    for (int side = 0; side < 2; ++side) {
                        da += (side == 0) ? delta : neg2delta;
                        double ct[4][NC], xe[NX];
                        rk4_end_state<DYN, false>(loc, loc + NX, da, dynl, ct, xe);
#pragma unroll
                        for (int r = 0; r < NX; ++r) e[side][r] = xe[r] - x2[r];
                    }
                    emit(jo, e[0], e[1]);
                }
            }
            else {  // midpoint collocation evaluates the dynamics off the grid states: full re-evaluation per column
                double loc[W];
#pragma unroll
                for (int i = 0; i < NX; ++i) { loc[i] = x1[i]; loc[S + i] = x2[i]; }
#pragma unroll
                for (int i = 0; i < NU; ++i) loc[NX + i] = u1[i];
#pragma unroll
                for (int c = 0; c < W; ++c) {
                    const bool mine = (c < NX) ? cs_x1 : ((c < NX + NU0) ? cs_u0 : ((c < S) ? cs_u1 : cs_x2));
                    const int jo    = sc[c];
                    if (!mine || jo < 0) continue;
                    double e[2][NX];
                    const double keep = loc[c];
#pragma unroll
                    for (int side = 0; side < 2; ++side) {
                        loc[c] += (side == 0) ? delta : neg2delta;
                        defect_eval<DYN, DEFECT>(loc, loc + NX, loc + S, dt0, dynl, e[side]);
                    }
                    loc[c] = keep;
                    emit(jo, e[0], e[1]);
                }
                const int jo = sc[S + NX];
                if (cs_x2 && jo >= 0) {
                    double e[2][NX];
                    double da = dt0;
#pragma unroll
                    for (int side = 0; side < 2; ++side) {
                        da += (side == 0) ? delta : neg2delta;
                        defect_eval<DYN, DEFECT>(loc, loc + NX, loc + S, da, dynl, e[side]);
                    }
                    emit(jo, e[0], e[1]);
                }
            }
        }
    }
    SWEEP_STAMP(5);
    // (2) least-squares cost blocks and bound rows, per vertex component -- unless written together with the residual above
    if (!jac_with_values) {
        for (int v = tid; v <= p.off_dt; v += THREADS) {
            const int4 ca = comp4[2 * v], cb = comp4[2 * v + 1];
            const CompInfo ci{ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
            int c, dim;
            double w, ref;
            bool fin;
            comp_role(v, c, dim, w, ref, fin);
            comp_jac(v, ci, xs[v], p.lb[xo + v], p.ub[xo + v], c, dim, w, ref, fin);
        }
    }
    // (3) stage inequality rows (active rows only, explicit zero otherwise, :1568-1610)
    {
        if (p.ineq_cols) {
            for (int k = tid; k < p.N - 1; k += THREADS) {
                double loc[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) loc[i] = xs[k * S + i];
                const double c0   = stage_ineq_state<NX>(p.mp.ineq_id, loc, p.mp.ineq);
                const bool active = (((c0 < 0) ? 0.0 : c0 * p.w_ineq) > 0.0);
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const int jo = p.ineq_cols[k * NX + i];
                    if (jo < 0) continue;
                    const double keep = loc[i];
                    loc[i] += delta;
                    const double c2 = stage_ineq_state<NX>(p.mp.ineq_id, loc, p.mp.ineq);
                    loc[i] += neg2delta;
                    const double c1 = stage_ineq_state<NX>(p.mp.ineq_id, loc, p.mp.ineq);
                    loc[i]  = keep;
                    jst[jo] = active ? (scalar * (c2 - c1)) * p.w_ineq : 0.0;
                }
            }
        }
    }
    if constexpr (NX <= 4) {   // final-stage inequality row on x_f (same rule: active row or explicit zeros)
        if (p.fin_row >= 0 && tid == THREADS - 1) {
            double loc[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) loc[i] = xs[(p.N - 1) * S + i];
            const double c0   = terminal_ball<NX>(loc, xr, p.mp.fin);
            const bool active = (((c0 < 0) ? 0.0 : c0 * p.w_ineq) > 0.0);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int jo = p.fin_joff[i];
                if (jo < 0) continue;
                const double keep = loc[i];
                loc[i] += delta;
                const double c2 = terminal_ball<NX>(loc, xr, p.mp.fin);
                loc[i] += neg2delta;
                const double c1 = terminal_ball<NX>(loc, xr, p.mp.fin);
                loc[i]  = keep;
                jst[jo] = active ? (scalar * (c2 - c1)) * p.w_ineq : 0.0;
            }
        }
    }
    // (4) extra edges: central differences of every unfixed component of every attached vertex (BaseEdge::computeJacobian, edge_interface.cpp:55-96);
    //     equality rows times w_eq (:1552), inequality rows times w_ineq where active, explicit zeros otherwise (:1568-1610)
    if constexpr (XE) {
        // One lane per COLUMN of an extra edge (round 6; SweepParams::xtasks: edge, attached vertex, component, the column's first Jacobian value): its value, then
        // the central-difference pair of that one component -- three evaluations per lane instead of 1 + 2 x (unfixed components) on one lane per edge (the
        // control-deviation edges of the headline structure: 9 evaluations with two divisions each, 10.3 k cycles of a 30 k sweep phase).  The same
        // operations on the same numbers per column: the values are bit-identical to the edge-per-lane loop below (kept for handles without the task table).
        if (p.xtasks) {
            for (int t = tid; t < p.n_xtasks; t += THREADS) {
                const int4 tk = p.xtasks[t];
                const XEdge xe = p.xedges[tk.x];
                double loc[4][XMC], f0[4], v2[4], v1[4];
                xe_load(xe, loc);
                xedge_values<NX, NU>(xe, loc, p.xparams, p.mp, f0);
                double keep = 0.0;
#pragma unroll
                for (int vi = 0; vi < 4; ++vi)
#pragma unroll
                    for (int c = 0; c < XMC; ++c) keep = (vi == tk.y && c == tk.z) ? loc[vi][c] : keep;
                const double a = keep + delta, b = a + neg2delta;   // x += delta; x += -2 delta (edge_interface.cpp:55-96)
#pragma unroll
                for (int vi = 0; vi < 4; ++vi)
#pragma unroll
                    for (int c = 0; c < XMC; ++c) loc[vi][c] = (vi == tk.y && c == tk.z) ? a : loc[vi][c];
                xedge_values<NX, NU>(xe, loc, p.xparams, p.mp, v2);
#pragma unroll
                for (int vi = 0; vi < 4; ++vi)
#pragma unroll
                    for (int c = 0; c < XMC; ++c) loc[vi][c] = (vi == tk.y && c == tk.z) ? b : loc[vi][c];
                xedge_values<NX, NU>(xe, loc, p.xparams, p.mp, v1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < xe.dim) {
                        const double dj = scalar * (v2[j] - v1[j]);
                        const bool active = (((f0[j] < 0) ? 0.0 : f0[j] * p.w_ineq) > 0.0);
                        jst[tk.w + j] = (xe.scale == 1) ? dj * p.w_eq : (active ? dj * p.w_ineq : 0.0);
                    }
            }
        }
        else
        for (int i = tid; i < p.n_xedges; i += THREADS) {
            const XEdge xe = p.xedges[i];
            double loc[4][XMC], f0[4];
            xe_load(xe, loc);
            xedge_values<NX, NU>(xe, loc, p.xparams, p.mp, f0);
#pragma unroll
            for (int vi = 0; vi < 4; ++vi) {
                if (vi >= xe.nverts || xe.joff[vi] < 0) continue;
                int col = 0;
#pragma unroll
                for (int c = 0; c < XMC; ++c) {
                    if (c >= xe.vdim[vi] || ((xe.fixed[vi] >> c) & 1u)) continue;
                    const double keep = loc[vi][c];
                    double v2[4], v1[4];
                    loc[vi][c] += delta;
                    xedge_values<NX, NU>(xe, loc, p.xparams, p.mp, v2);
                    loc[vi][c] += neg2delta;
                    xedge_values<NX, NU>(xe, loc, p.xparams, p.mp, v1);
                    loc[vi][c] = keep;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < xe.dim) {
                            const double dj = scalar * (v2[j] - v1[j]);
                            const bool active = (((f0[j] < 0) ? 0.0 : f0[j] * p.w_ineq) > 0.0);
                            jst[xe.joff[vi] + col * xe.edim + xe.rie + j] = (xe.scale == 1) ? dj * p.w_eq : (active ? dj * p.w_ineq : 0.0);
                        }
                    ++col;
                }
            }
        }
    }
    SWEEP_STAMP(6);
    if constexpr (STAGE) {
        lds_barrier();
    SWEEP_STAMP(7);
        // (run-to-completion loop, two-wave shape: the factor phase of this pass streams the range out behind its own loads -- stream_jacobian_range --
        //  so that nothing on the pass's critical path waits for the write acknowledgements)
        if (LEAN && in_loop) return;
        // ---- stream the Jacobian values to HBM: 16 bytes per lane, fully coalesced
        // (stand-alone kernel: streaming stores -- the consumer is a later launch and 1024 Jacobians do not fit the L2 anyway, +7 % on the
        //  sweep; fused kernel: normal stores -- the pass after a rejected step reads its Jacobian back from the L2, streaming costs 3 %)
        // (two-wave run-to-completion shape: only the range the factor phase of a later pass re-reads -- SweepParams::jlean_*; the cost blocks and bound
        //  rows of the stage components are not even assembled in the staging area there, they travel in registers)
        const int so_lo = (LEAN && p.jlean_hi2 > 0) ? p.jlean_lo2 : 0, so_hi = (LEAN && p.jlean_hi2 > 0) ? p.jlean_hi2 : p.nnz_pad / 2;
        for (int i = so_lo + tid; i < so_hi; i += THREADS) {
            const double2 v = reinterpret_cast<const double2*>(jst)[i];
            if constexpr (FUSED) reinterpret_cast<double2*>(js)[i] = v;
            else {
                __builtin_nontemporal_store(v.x, &js[2 * i]);
                __builtin_nontemporal_store(v.y, &js[2 * i + 1]);
            }
        }
    SWEEP_STAMP(8);
    }
}

// the lean range of the staged Jacobian (SweepParams::jlean_*) from LDS to HBM, for the passes that follow a rejected step
template <int THREADS>
__device__ __forceinline__ void stream_jacobian_range(const SweepParams& p, const double* jst, const int inst, const int tid)
{
    double* js = p.jac + (size_t)inst * p.nnz_pad;
    const int lo = (p.jlean_hi2 > 0) ? p.jlean_lo2 : 0, hi = (p.jlean_hi2 > 0) ? p.jlean_hi2 : p.nnz_pad / 2;
    for (int i = lo + tid; i < hi; i += THREADS) reinterpret_cast<double2*>(js)[i] = reinterpret_cast<const double2*>(jst)[i];
}

template <int DYN, int DEFECT, bool DENSE = false, bool LONG = false, bool XE = false, bool NOJAC = false>
__global__ __launch_bounds__(SWEEP_THREADS) void sweep_kernel(const SweepParams p)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* xs  = smem;
    double* red = smem + p.nvs;
    double* cs  = red + 10;
    double* jst = cs + ((p.N * Dynamics<DYN>::NC + 1) & ~1);  // Jacobian staging (16-byte aligned; unused by big models)
    LmState* sl = reinterpret_cast<LmState*>(jst + ((p.nx <= 4 && !LONG) ? p.nnz_pad : 0) + (Dynamics<DYN>::NX > 4 ? SWEEP_ROLE_TABLE : 0));   // (big models: the role table in front of it)
    const int inst = blockIdx.x + p.inst0;
    if (p.st) { lm_state_in(sl, p.st + inst, threadIdx.x); __syncthreads(); }
    sweep_body<DYN, DEFECT, false, DENSE, LONG, SWEEP_THREADS, XE, NOJAC>(p, p.mode, p.active_count, sl, xs, red, cs, jst, inst, threadIdx.x);
    if (p.st && p.mode >= 2) { __syncthreads(); lm_state_out(p.st + inst, sl, threadIdx.x); }
}

#pragma clang fp contract(fast)

#ifndef CORBO_HIP_DYN_TU
// ---------------------------------------------------------------------------------------------------------------------
// moving-horizon grid update (FullDiscretizationGridBase::update, new_run branch, full_discretization_grid_base.cpp:91-108)
// ---------------------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
// One workgroup per instance; the new vertex vector is assembled in LDS from the old one and written back.
__global__ __launch_bounds__(256) void warm_start_kernel(const WarmStartParams p)
{
    extern __shared__ __attribute__((aligned(16))) double wsm[];
    double* nw   = wsm;                 // [nvs] new vertex values
    double* dist = wsm + p.nvs;         // [21]  |x0 - x_seq[i]|, i = 0..20
    int* ish     = reinterpret_cast<int*>(dist + 22);
    const int inst = blockIdx.x, tid = threadIdx.x;
    const int nx = p.nx, nu = p.nu, N = p.N, s = nx + nu;
    double* X        = p.x + (size_t)inst * p.nvs;
    const double* x0 = p.x0new + (size_t)inst * CORBO_HIP_MAX_NX;
    const int lookahead = (N - 2 < 20) ? N - 2 : 20;   // min(num_interv - 1, 20), :303
    // findNearestState (:285-317)
    if (p.shift && tid <= lookahead) {
        double acc = 0.0;
        for (int c = 0; c < nx; ++c) { const double d = x0[c] - X[tid * s + c]; acc += d * d; }
        dist[tid] = sqrt(acc);
    }
    __syncthreads();
    if (tid == 0) {
        int num_shift = 0;
        if (p.shift && !(fabs(dist[0]) < 1e-12)) {
            double cache = dist[0];
            for (int i = 1; i <= lookahead; ++i) {
                if (dist[i] < cache) { cache = dist[i]; num_shift = i; }
                else break;
            }
        }
        if (num_shift > N - 2) num_shift = 0;   // "Cannot shift if num_shift > N-2" (:236-240)
        ish[0] = num_shift;
    }
    __syncthreads();
    const int ns = ish[0];
    // shifted copy (:247-260): x_seq[i] = x_seq[i + ns] (or x_f), u_seq[i] = u_seq[i + ns]; everything else stays
    for (int e = tid; e < p.nvs; e += 256) {
        double v = X[e];
        if (ns > 0 && e < (N - 1) * s) {
            const int i = e / s, c = e % s, idx = i + ns;
            if (i < N - ns) {
                if (idx == N - 1) { if (c < nx) v = X[(N - 1) * s + c]; }   // final state reached (u_seq[i] untouched)
                else v = X[idx * s + c];
            }
        }
        nw[e] = v;
    }
    __syncthreads();
    // linear extrapolation of the tail (:262-282), a short sequential chain per component
    if (ns > 0) {
        if (tid < nx) {
            int idx = N - ns;
            for (int i = 0; i < ns; ++i, ++idx) {
                const double a = nw[(idx - 2) * s + tid], b = nw[(idx - 1) * s + tid];
                const double v = a + 2.0 * (b - a);
                if (i == ns - 1) nw[(N - 1) * s + tid] = v;
                else nw[idx * s + tid] = v;
            }
        }
        else if (tid >= 64 && tid < 64 + nu) {
            const int c = nx + (tid - 64);
            int idx = N - ns;
            for (int i = 0; i < ns; ++i, ++idx) nw[(idx - 1) * s + c] = nw[(idx - 2) * s + c];
        }
    }
    __syncthreads();
    if (tid < nx) {
        nw[tid] = x0[tid];                                                                   // x_seq.front() = x0 (:101)
        if ((p.xf_fixed_mask >> tid) & 1) nw[(N - 1) * s + tid] = p.xref[(size_t)inst * CORBO_HIP_MAX_NX + tid];   // :103-106
    }
    __syncthreads();
    for (int e = tid; e < p.nvs; e += 256) X[e] = nw[e];
}
// resampleTrajectory: one workgroup per (source instance, destination instance) pair.  The old trajectory is staged in LDS (the
// destination may be another row of the SAME array: compaction moves), every interior grid point of the new trajectory is one lane:
// idx_old = the first old sample not before t_new (the reference's running while-loop, :428-432, is monotone, so each lane finds it
// on its own), linear interpolation of the state, held control (:440-447), same operation order -> bit-identical to the oracle.
__global__ __launch_bounds__(256) void resample_kernel(const ResampleParams p)
{
    extern __shared__ __attribute__((aligned(16))) double old[];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int nx = p.nx, nu = p.nu, s = nx + nu, n = p.n_src, n_new = p.n_dst;
    const double* X = p.x_src + (size_t)p.src_index[q] * p.nvs_src;
    double* Y       = p.x_dst + (size_t)p.dst_index[q] * p.nvs_dst;
    for (int e = tid; e < p.nvs_src; e += 256) old[e] = X[e];
    double xr = 0.0;
    if (tid < CORBO_HIP_MAX_NX) xr = p.xref_src[(size_t)p.src_index[q] * CORBO_HIP_MAX_NX + tid];
    __syncthreads();
    if (tid < CORBO_HIP_MAX_NX) p.xref_dst[(size_t)p.dst_index[q] * CORBO_HIP_MAX_NX + tid] = xr;
    const double* xf    = old + (n - 1) * s;
    const double dt_old = old[(n - 1) * s + nx];
    if (n == n_new) {   // :400
        for (int e = tid; e < p.nvs_dst; e += 256) Y[e] = old[e];
        return;
    }
    const double dt_new = dt_old * (double)(n - 1) / (double)(n_new - 1);
    for (int e = tid; e < s; e += 256) Y[e] = old[e];                         // start sample untouched
    for (int e = tid; e < nx; e += 256) Y[(n_new - 1) * s + e] = xf[e];       // x_f copied
    if (tid == 0) {
        Y[(n_new - 1) * s + nx] = dt_new;
        for (int e = (n_new - 1) * s + nx + 1; e < p.nvs_dst; ++e) Y[e] = 0.0;
    }
    for (int idx_new = 1 + tid; idx_new < n_new - 1; idx_new += 256) {
        const double t_new = dt_new * (double)idx_new;
        int idx_old = 1;
        while (t_new > (double)idx_old * dt_old && idx_old < n) ++idx_old;
        const double t_old_p1 = (double)idx_old * dt_old;
        const double* x_prev  = old + (idx_old - 1) * s;
        const double* x_cur   = (idx_old < n - 1) ? old + idx_old * s : xf;
        const double f        = (t_new - (t_old_p1 - dt_old)) / dt_old;
        double* yn            = Y + idx_new * s;
        for (int c = 0; c < nx; ++c) yn[c] = x_prev[c] + f * (x_cur[c] - x_prev[c]);
        const double* u_prev = old + ((idx_old - 1 < n - 1) ? idx_old - 1 : n - 2) * s + nx;
        for (int c = 0; c < nu; ++c) yn[nx + c] = u_prev[c];
    }
}
#pragma clang fp contract(fast)
#endif  // !CORBO_HIP_DYN_TU

// ---------------------------------------------------------------------------------------------------------------------
// assemble + factor + solve
// ---------------------------------------------------------------------------------------------------------------------

// Cholesky of a small dense SPD matrix held in registers: lower factor in M, diagonal replaced by 1/L_ii.
template <int Nn>
__device__ __forceinline__ void chol_inv(double (&M)[Nn][Nn])
{
#pragma unroll
    for (int j = 0; j < Nn; ++j) {
        double d = M[j][j];
#pragma unroll
        for (int c = 0; c < j; ++c) d -= M[j][c] * M[j][c];
        const double inv = rsqrt(d);  // 1/L_jj (NaN for a non-positive pivot, like an unchecked LLT: chi2 -> NaN -> step rejected)
        M[j][j]          = inv;
#pragma unroll
        for (int i = j + 1; i < Nn; ++i) {
            double v = M[i][j];
#pragma unroll
            for (int c = 0; c < j; ++c) v -= M[i][c] * M[j][c];
            M[i][j] = v * inv;
        }
    }
}
// X := L^{-1} X for a column block X[Nn][Mm] (L from chol_inv)
template <int Nn, int Mm>
__device__ __forceinline__ void fwd_solve(const double (&L)[Nn][Nn], double (&X)[Nn][Mm])
{
#pragma unroll
    for (int c = 0; c < Mm; ++c)
#pragma unroll
        for (int i = 0; i < Nn; ++i) {
            double v = X[i][c];
#pragma unroll
            for (int j = 0; j < i; ++j) v -= L[i][j] * X[j][c];
            X[i][c] = v * L[i][i];
        }
}
template <int Nn>
__device__ __forceinline__ void fwd_solve_vec(const double (&L)[Nn][Nn], double (&x)[Nn])
{
#pragma unroll
    for (int i = 0; i < Nn; ++i) {
        double v = x[i];
#pragma unroll
        for (int j = 0; j < i; ++j) v -= L[i][j] * x[j];
        x[i] = v * L[i][i];
    }
}
// x := L^{-T} x
template <int Nn>
__device__ __forceinline__ void bwd_solve_vec(const double (&L)[Nn][Nn], double (&x)[Nn])
{
#pragma unroll
    for (int i = Nn - 1; i >= 0; --i) {
        double v = x[i];
#pragma unroll
        for (int j = i + 1; j < Nn; ++j) v -= L[j][i] * x[j];
        x[i] = v * L[i][i];
    }
}

// value of lane SRC of the caller's quad (DPP quad_perm broadcast; all four lanes of the quad must be active)
template <int SRC>
__device__ __forceinline__ double quad_bcast(double v)
{
    constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

#ifndef CORBO_HIP_TWISTED_TOP
#define CORBO_HIP_TWISTED_TOP 1      // (0: cyclic reduction up to the root, for A/B)
#endif
#ifndef CORBO_HIP_SOLO_LAST_LEVEL
#define CORBO_HIP_SOLO_LAST_LEVEL 1  // (0: the last back-substitution level in two rounds of quads, for A/B)
#endif
#ifndef CORBO_HIP_RELOAD_UNR
#define CORBO_HIP_RELOAD_UNR 10      // 16-byte loads in flight per lane when the Jacobian range is staged again after a rejected step (two-wave shape)
#endif
#ifndef CORBO_HIP_COMPACT_LEVELS
#define CORBO_HIP_COMPACT_LEVELS 1   // (0: the two-round h = 2 level of round 4, for A/B)
#endif
#define SOA(arr, e, k) (arr)[(e) * NP + (k)]
#define TRI(i, j) ((i) * ((i) + 1) / 2 + (j))  // packed lower triangle, i >= j
#define STAMP(id)                                                       \
    do {                                                                \
        if (p.timeline && inst == p.timeline_inst && tid == 0) p.timeline[id] = clock64(); \
    } while (0)

// twisted top of the block-tridiagonal elimination (factor_body): the level h (a power of two >= 4) at which at most eight and at least three blocks are left
__host__ __device__ constexpr int twist_level(int N)
{
    for (int h = 4; h < N; h <<= 1) {
        const int nb = (N + h - 1) / h;
        if (nb <= 8) return nb >= 3 ? h : 0;
    }
    return 0;
}
// value of lane 0 of the caller's lane PAIR (0,1), (2,3), ...: DPP quad_perm [0,0,2,2]
__device__ __forceinline__ double pair_bcast0(double v)
{
    constexpr int ctrl = 0 | (0 << 2) | (2 << 4) | (2 << 6);
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// ... and of lane 1 of the pair: quad_perm [1,1,3,3]
__device__ __forceinline__ double pair_bcast1(double v)
{
    constexpr int ctrl = 1 | (1 << 2) | (3 << 4) | (3 << 6);
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// LDS carve of factor_body (doubles), shared with the fused pass kernel
template <int NX, int NU>
struct FactorLds {
    static constexpr int NT = NX * (NX + 1) / 2;
    static constexpr int RED = 24;
    __host__ __device__ static constexpr int off_Wam(int NP) { return (NU * NU + 2 * NU * NX + NU + NT) * NP; }
    __host__ __device__ static constexpr int off_Wbm(int NP) { return off_Wam(NP) + NX * NX * NP; }
    __host__ __device__ static constexpr int off_red(int NP) { return off_Wbm(NP) + NX * NX * NP + NX * NP; }
    __host__ __device__ static constexpr int total(int NP, bool arrow) { return off_red(NP) + RED + (arrow ? (NU + NX) * NP : 0); }
};

// Non-diagonal weights (corbo_hip_problem_desc::weights_dense): the cost edge of a vertex is a dense block C (the finite-difference
// Jacobian of U (x - ref), upper triangular up to rounding) instead of single-entry rows.  Its share of H = J^T J and rhs = -J^T r for
// stage k's state vertex (Q, or Qf on the last block) and control vertex (R):  out = [C^T C packed lower (NT) | -C^T v (NX) | the same for
// the controls (NUT, NU)], zeros for a vertex whose weight is diagonal (those go through the single-entry path) or that has no cost edge.
// Deliberately NOT inlined and handed a private array: the diagonal problems (every BASELINE configuration) pay one uniform branch and
// no registers for it.
template <int NX, int NU>
__device__ __noinline__ void dense_cost_terms(const StageCols*, const CompInfo* comp, int wdense_mask, int N, const double* J, const double* val, int k, double* out)
{
    constexpr int S = NX + NU, NT = NX * (NX + 1) / 2, NUT = NU * (NU + 1) / 2;
#pragma unroll
    for (int i = 0; i < NT + NX + NUT + NU; ++i) out[i] = 0.0;
    if (k >= N) return;
    {   // state vertex of block k
        const int cls = (k == N - 1) ? 2 : 0;
        if (((wdense_mask >> cls) & 1) && comp[k * S].cost_row >= 0) {
            double C[NX][NX], v[NX];
#pragma unroll
            for (int c = 0; c < NX; ++c) {
                const CompInfo ci = comp[k * S + c];
                v[c] = val[ci.cost_row];
                const bool have = !ci.fixed && ci.cost_joff >= 0;
                const int col0  = have ? ci.cost_joff - c : 0;
#pragma unroll
                for (int r = 0; r < NX; ++r) { const double a = J[col0 + r]; C[r][c] = have ? a : 0.0; }
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double g = 0.0;
#pragma unroll
                for (int r = 0; r < NX; ++r) g -= C[r][i] * v[r];
                out[NT + i] = g;
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    double d = 0.0;
#pragma unroll
                    for (int r = 0; r < NX; ++r) d += C[r][i] * C[r][j];
                    out[i * (i + 1) / 2 + j] = d;
                }
            }
        }
    }
    if (k < N - 1 && ((wdense_mask >> 1) & 1) && comp[k * S + NX].cost_row >= 0) {   // control vertex of stage k
        double C[NU][NU], v[NU];
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            const CompInfo ci = comp[k * S + NX + c];
            v[c] = val[ci.cost_row];
            const bool have = !ci.fixed && ci.cost_joff >= 0;
            const int col0  = have ? ci.cost_joff - c : 0;
#pragma unroll
            for (int r = 0; r < NU; ++r) { const double a = J[col0 + r]; C[r][c] = have ? a : 0.0; }
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            double g = 0.0;
#pragma unroll
            for (int r = 0; r < NU; ++r) g -= C[r][i] * v[r];
            out[NT + NX + NUT + i] = g;
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double d = 0.0;
#pragma unroll
                for (int r = 0; r < NU; ++r) d += C[r][i] * C[r][j];
                out[NT + NX + i * (i + 1) / 2 + j] = d;
            }
        }
    }
}

// j_in_lds: the Jacobian values of this instance already sit in smem[0, nnz_pad) (left there by the sweep phase of the same
// workgroup); otherwise they are staged from HBM first.
// NPC > 0: the padded block count N | 1 as a compile-time constant (LDS element strides become immediate offsets of the DS
// instructions instead of two VALU operations per access); 0: taken from the launch parameters.
// GWS: the factor workspace (`smem`) is a per-instance array in HBM instead of LDS, and the Jacobian is read from HBM in place -- the
// long-horizon variant (256 < N <= 1024, one lane per stage in a 1024-thread workgroup; 45 N doubles do not fit 160 KB of LDS then).
// Same code, the barriers become full workgroup barriers (global memory crosses them).
// RECOMP (two-wave shape of the run-to-completion kernel; `sq` = the launch's sweep parameters): the single-entry rows of the lane's components
// -- cost and bound values and their Jacobian entries -- are re-evaluated from the accepted iterate (diagonal_rows, the very function the sweep
// phase sums chi2 with) instead of being fetched: the sweep phase does not store them, and the load phase loses its dependent round trip
// (table entry -> residual row).
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// after_gather: called once every lane holds its Jacobian entries and before the staging area is released -- the run-to-completion kernel streams the
// freshly evaluated Jacobian out there and does its per-CU bookkeeping: memory operations that nothing in the factor phase waits for, issued BEHIND the
// phase's own loads (gfx9 returns vector-memory operations in order: a load's wait includes every store and every slow load issued before it).
// GWS = 2 (round 6; horizons beyond 256 grid points whose state-block arrays fit the CU's LDS: up to ~ 700 grid points for nx = 3): only the arrays of the eliminated
// CONTROLS -- written and read by the lane of their own stage, nothing crosses lanes -- live in the HBM workspace (`ctrl_ws`); D, W_a, W_b, the right-hand sides and the
// reduction scratch, everything the levels of the cyclic reduction exchange between lanes, are in LDS (`smem`).  The Jacobian is read from HBM in place as for GWS = 1.
template <int NX, int NU, int THREADS, bool ARROW, int NPC = 0, bool DENSE = false, int GWS = 0, bool RECOMP = false, class Hook = NoHook>
__device__ __forceinline__ void factor_body(const FactorParams& p, LmState* const st, double* smem, const int inst, const int tid, const bool j_in_lds, double* const xt_lds = nullptr, const SweepParams* const sq = nullptr,
                                            const StageKeep<NX, NU>* const keep = nullptr, const bool keep_valid = false, Hook&& after_gather = Hook{},
                                            const int* const sc_in = nullptr, int* const smask = nullptr, double* const ctrl_ws = nullptr)
{
    constexpr int S  = NX + NU;
    constexpr int NW = THREADS / 64;
    // (two-wave run-to-completion shape) no table load at the start of the factor phase: the defect edge's Jacobian offsets come from the sweep phase of the
    // same pass in registers (sc_in), and what the component tables say about this lane's stage is decoded once per solve into *smask -- bits [0, NX) fixed
    // components, [8, 8 + S) cost rows present, [16, 16 + S) bound rows present, 24: the last block carries rows of a terminal equality, 31: valid
    const bool SM = RECOMP && smask && (*smask < 0);
    constexpr int NT = NX * (NX + 1) / 2;
    const int N  = p.N;
    const int NP = NPC > 0 ? NPC : (N | 1);
    auto fb_barrier = [] {
        if constexpr (GWS == 1) __syncthreads();   // workspace in HBM: the stores of every wave are visible behind the barrier
        else lds_barrier();                          // (GWS = 2: what crosses lanes is in LDS)
    };
    constexpr int REDN = GWS ? 8 * (THREADS / 64) : FactorLds<NX, NU>::RED;   // reduction scratch: 5 doubles per wave
    // SoA arrays, element-major: arr[e][block].  Per state block: D/L (packed lower), W_a, W_b, rhs/y/x; per stage: the
    // eliminated controls.  Slots of block k+1 double as the mailbox for what stage k contributes to it.
    double* Luu = (GWS == 2) ? ctrl_ws : smem;   // NU*NU   L_uu (diag inverted)
    double* Zx  = Luu + NU * NU * NP;      // NU*NX   L_uu^{-1} H(u_k, x_k)
    double* Zp  = Zx + NU * NX * NP;       // NU*NX   L_uu^{-1} H(u_k, x_{k+1})
    double* yu  = Zp + NU * NX * NP;       // NU
    double* Dm  = (GWS == 2) ? smem : yu + NU * NP;   // NT      D_i (packed lower) -> L_i
    double* Wam = Dm + NT * NP;            // NX*NX   W_a = L_i^{-1} H(i, i-h)   (before: mailbox for H(k, k-1))
    double* Wbm = Wam + NX * NX * NP;      // NX*NX   W_b = L_i^{-1} H(i, i+h)
    double* gv  = Wbm + NX * NX * NP;      // NX      rhs -> y -> delta x
    double* red = gv + NX * NP;            // 24
    double* zu  = (GWS == 2) ? yu + NU * NP : red + REDN;  // NU  (arrowhead only from here on; a stage's own like the other arrays of the controls)
    double* bv  = (GWS == 2) ? red + REDN : zu + NU * NP;  // NX      border column -> z
    const int done = st->done, fresh = st->fresh, first = st->first, vbuf = st->vbuf;
    const int stop_in = st->stop;
    double mu = st->mu;
    const double mu_acc_in = st->mu_acc;
    fb_barrier();
    if (done) return;
    STAMP(0);

    const double* val = (vbuf ? p.values1 : p.values0) + (size_t)inst * p.m_pad;
    const double* xin = p.x + (size_t)inst * p.nvs;
    const int k       = tid;
    const bool has_stage = (k < N - 1);
    const bool has_block = (k < N);
    // ---- load phase, organised so that every memory latency is paid once: (1) the static tables of this lane's stage (L2), (2) the
    //      residual entries they point to (global; all loads issued back to back, indices clamped instead of branched), (3) the
    //      Jacobian staging (coalesced 16-byte loads into LDS unless the sweep phase left it there), (4) the per-stage Jacobian
    //      gathers from LDS.  Absent entries (offset -1: fixed component, no cost/bound row) are loaded from index 0 and zeroed.
    constexpr int WL = S + NX + 1;  // local columns of the defect edge: x_k, u_k, x_{k+1}, dt
    int sco[WL];
    if (RECOMP && sc_in) {
#pragma unroll
        for (int c = 0; c < WL; ++c) sco[c] = has_stage ? sc_in[c] : -1;
    }
    else {
        const int ks = has_stage ? k : 0;
#pragma unroll
        for (int c = 0; c < WL; ++c) { const int o = p.stage_cols[ks].col[c]; sco[c] = has_stage ? o : -1; }
    }
    int cj[S], cr[S], bj[S], br[S], xfixed[NX];   // cost / bound row (Jacobian offset, residual row) per component of stage k
    if (SM) {   // (decoded: RECOMP takes the rows' values and entries from registers / recomputes them, only their presence matters here)
        const int m = *smask;
#pragma unroll
        for (int e = 0; e < S; ++e) {
            cj[e] = ((m >> (8 + e)) & 1) ? 0 : -1; bj[e] = ((m >> (16 + e)) & 1) ? 0 : -1; cr[e] = 0; br[e] = 0;
            if (e < NX) xfixed[e] = (m >> e) & 1;
        }
    }
    else {
        const int kb = has_block ? k : 0;
        int m = (int)0x80000000u;
#pragma unroll
        for (int e = 0; e < S; ++e) {
            const bool ok = (e < NX) ? has_block : has_stage;
            const CompInfo ci = p.comp[(ok ? kb : 0) * S + e];
            cj[e] = ok ? ci.cost_joff : -1; cr[e] = ci.cost_row; bj[e] = ok ? ci.bnd_joff : -1; br[e] = ci.bnd_row;
            if (e < NX) xfixed[e] = ok ? ci.fixed : 1;
            m |= (cj[e] >= 0 ? 1 << (8 + e) : 0) | (bj[e] >= 0 ? 1 << (16 + e) : 0);
            if (e < NX) m |= xfixed[e] ? 1 << e : 0;
            if (e < NX && has_block && !has_stage && !ci.fixed && ci.cost2_joff >= 0 && ci.cost2_row >= 0) m |= 1 << 24;
        }
        if (RECOMP && smask) *smask = m;
    }
    int iq_row = -1, iq[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) iq[i] = -1;
    if (p.ineq_cols && has_stage) {
        iq_row = p.ineq_rows[k];
#pragma unroll
        for (int i = 0; i < NX; ++i) iq[i] = p.ineq_cols[k * NX + i];
    }
    else if (p.fin_row >= 0 && k == N - 1) {   // final-stage inequality: a row on the last state block
        iq_row = p.fin_row;
#pragma unroll
        for (int i = 0; i < NX; ++i) iq[i] = p.fin_joff[i];
    }
    // (2) residual entries
    double r[NX], vc[S], vb[S], rin;
    double acr[S], abr[S];   // RECOMP: the Jacobian entries of the same rows
    double xk[S];            // the accepted iterate of this lane's stage (RECOMP: needed here; otherwise requested before the back-substitution)
#pragma unroll
    for (int i = 0; i < NX; ++i) r[i] = (RECOMP && keep_valid) ? keep->r[i] : val[has_stage ? p.eq_row0 + k * NX + i : 0];
    if (RECOMP && keep_valid) {   // the rows lane k's own sweep phase evaluated (run to completion: always; one pass per launch: after an accepted step)
#pragma unroll
        for (int e = 0; e < S; ++e) { vc[e] = keep->vc[e]; acr[e] = keep->ac[e]; vb[e] = keep->vb[e]; abr[e] = keep->ab[e]; }
    }
    else if constexpr (RECOMP) {
        const SweepParams& q = *sq;
        const size_t xo = (size_t)inst * p.nvs;
        double lo[S], up[S], rf[NX];
#pragma unroll
        for (int e = 0; e < S; ++e) {
            const int v = ((e < NX) ? has_block : has_stage) ? k * S + e : 0;
            xk[e] = xin[v]; lo[e] = q.lb[xo + v]; up[e] = q.ub[xo + v];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) rf[i] = q.refvec ? q.refvec[xo + (has_block ? k * S + i : 0)] : q.xref[(size_t)inst * CORBO_HIP_MAX_NX + i];
        const bool lastb = (k == N - 1);
#pragma unroll
        for (int e = 0; e < S; ++e) {
            const bool is_u = (e >= NX);
            const double w  = is_u ? q.mp.sr[is_u ? e - NX : 0] : (lastb ? q.mp.sqf[is_u ? 0 : e] : q.mp.sq[is_u ? 0 : e]);
            diagonal_rows(xk[e], w, is_u ? 0.0 : rf[is_u ? 0 : e], lo[e], up[e], q.w_b, vc[e], acr[e], vb[e], abr[e]);
        }
    }
    else {
#pragma unroll
        for (int e = 0; e < S; ++e) { vc[e] = val[cj[e] >= 0 ? cr[e] : 0]; vb[e] = val[bj[e] >= 0 ? br[e] : 0]; }
    }
    rin = 0.0;
    if (iq_row >= 0) rin = val[iq_row];   // (a real branch: no load at all for a stage without an inequality row)
    if (!has_stage) {
#pragma unroll
        for (int i = 0; i < NX; ++i) r[i] = 0.0;
    }
    // (3) Jacobian staging
    const double* J = GWS ? p.jac + (size_t)inst * p.nnz_pad : smem;
    // (measured and not kept: after a rejected step lane k gathering its stage's defect block straight from HBM / L2 instead of staging the range in LDS
    //  first -- the load phase 4.7 k -> 3.9 k cycles, but one pointer for two address spaces makes every access of J a flat access: the solve 0.471 -> 0.478 ms)
    if (!GWS && !j_in_lds) {
        const double2* src = reinterpret_cast<const double2*>(p.jac + (size_t)inst * p.nnz_pad);
        double2* dst       = reinterpret_cast<double2*>(smem);
        const int lo2      = (RECOMP && p.jlean_hi2 > 0) ? p.jlean_lo2 : 0;                  // (RECOMP: the range the sweep phase streamed out, FactorParams::jlean_*)
        const int n2       = (RECOMP && p.jlean_hi2 > 0) ? p.jlean_hi2 : p.nnz_pad / 2;
        // loads in flight per lane (8 costs 3 % of a solve at the 128-VGPR budget) (branch-free: indices are clamped, the duplicates are harmless);
        // the two-wave shape of the run-to-completion kernel has the registers for the whole headline Jacobian in ONE round trip (17 x 128 x 16 bytes)
        constexpr int UNR  = (THREADS <= 128) ? CORBO_HIP_RELOAD_UNR : 4;   // (17 -- the whole headline Jacobian in one round trip of the two-wave shape -- was measured: 5.2 k -> 7.2 k cycles alone, 14 k+ under load)
        for (int i0 = lo2 + tid; i0 < n2; i0 += THREADS * UNR) {
            double2 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) { const int i = i0 + u * THREADS; v[u] = src[i < n2 ? i : n2 - 1]; }
#pragma unroll
            for (int u = 0; u < UNR; ++u) { const int i = i0 + u * THREADS; dst[i < n2 ? i : n2 - 1] = v[u]; }
        }
    }
    fb_barrier();
    // (4) per-stage gathers from the staging area
    double A[NX][NX], B[NX][NU], Cc[NX][NX], dc[NX];
#pragma unroll
    for (int c = 0; c < NX; ++c) {
        const int oa = sco[c], oc = sco[S + c];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double va = J[(oa >= 0 ? oa : 0) + i], vcc = J[(oc >= 0 ? oc : 0) + i];
            A[i][c]  = (oa >= 0) ? va : 0.0;
            Cc[i][c] = (oc >= 0) ? vcc : 0.0;
        }
    }
#pragma unroll
    for (int c = 0; c < NU; ++c) {
        const int o = sco[NX + c];
#pragma unroll
        for (int i = 0; i < NX; ++i) { const double v = J[(o >= 0 ? o : 0) + i]; B[i][c] = (o >= 0) ? v : 0.0; }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        dc[i] = 0.0;
        if constexpr (ARROW) { const int o = sco[S + NX]; const double v = J[(o >= 0 ? o : 0) + i]; dc[i] = (o >= 0) ? v : 0.0; }
    }
    // diagonal (single-entry) rows of this lane's components: cost rows, bound rows
    // (a vertex with a NON-DIAGONAL weight has a dense cost block instead: its share comes from dense_cost_terms below)
    const int wdm = DENSE ? p.wdense_mask : 0;
    const bool wd_x = DENSE && wdm && ((wdm >> ((k == N - 1) ? 2 : 0)) & 1), wd_u = DENSE && wdm && ((wdm >> 1) & 1);
    double du_diag[NU], gu[NU], dx_diag[NX], gx[NX];
#pragma unroll
    for (int e = 0; e < S; ++e) {
        double ac, ab;
        if constexpr (RECOMP) { ac = acr[e]; ab = abr[e]; }
        else { ac = J[cj[e] >= 0 ? cj[e] : 0]; ab = J[bj[e] >= 0 ? bj[e] : 0]; }
        if (cj[e] < 0 || ((e < NX) ? wd_x : wd_u)) ac = 0.0;
        if (bj[e] < 0) ab = 0.0;
        const double dd = ac * ac + ab * ab, gg = -(ac * vc[e]) - ab * vb[e];
        if (e < NX) { dx_diag[e] = dd; gx[e] = gg; }
        else { du_diag[e - NX] = dd; gu[e - NX] = gg; }
    }
    // non-diagonal weights: the dense cost blocks' shares, computed out of line into a private array (see dense_cost_terms)
    constexpr int NUT = NU * (NU + 1) / 2;
    double dq[NT + NX + NUT + NU];
    if constexpr (DENSE) { if (wdm) dense_cost_terms<NX, NU>(p.stage_cols, p.comp, wdm, N, J, val, k, dq); }
    if (has_block && !has_stage && !(SM && !((*smask >> 24) & 1))) {   // the last block: rows of a TerminalEqualityConstraint on x_f (second diagonal row of a component)
#pragma unroll
        for (int e = 0; e < NX; ++e) {
            const CompInfo ci = p.comp[k * S + e];
            if (!ci.fixed && ci.cost2_joff >= 0 && ci.cost2_row >= 0) { const double a = J[ci.cost2_joff]; dx_diag[e] += a * a; gx[e] -= a * val[ci.cost2_row]; }
        }
    }
    // stage inequality row on x_k: rank-1 contribution c c^T
    double cin[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { const double v = J[iq[i] >= 0 ? iq[i] : 0]; cin[i] = (iq[i] >= 0) ? v : 0.0; }
    // border (dt) scalars: this lane's share of H(dt,dt) and rhs(dt)
    double cdt = 0, gdt = 0;
    if constexpr (ARROW) {
#pragma unroll
        for (int i = 0; i < NX; ++i) { cdt += dc[i] * dc[i]; gdt -= dc[i] * r[i]; }
        if (tid == 0) {
            const CompInfo ci = p.comp[p.off_dt];
            if (ci.cost_joff >= 0) { const double a = J[ci.cost_joff]; cdt += a * a; gdt -= a * val[ci.cost_row]; }
            if (ci.cost2_joff >= 0) { const double a = J[ci.cost2_joff]; cdt += a * a; gdt -= a * val[ci.cost2_row]; }
            if (ci.bnd_joff >= 0) { const double a = J[ci.bnd_joff]; cdt += a * a; gdt -= a * val[ci.bnd_row]; }
        }
    }
    after_gather();
    fb_barrier();  // every lane has taken its Jacobian entries out of the staging area
    STAMP(1);

    // ---- first factorisation of a solve: mu = tau * max diag(J^T J), stop = |rhs|_inf <= eps1 (:115-118)
    int stop = stop_in;
    if (first) {
        // raw neighbour parts of x_{k+1}: diag(C^T C), -C^T r  (scratch: the W_b / rhs slots, rewritten below)
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double dd = 0, gg = 0;
#pragma unroll
            for (int q = 0; q < NX; ++q) { dd += Cc[q][i] * Cc[q][i]; gg -= Cc[q][i] * r[q]; }
            if (has_stage) { SOA(Wbm, i, k) = dd; SOA(Wbm, NX + i, k) = gg; }
        }
        fb_barrier();
        double mx_d = -1e300, mx_g = 0;
#pragma unroll
        for (int j = 0; j < NU; ++j)
            if (has_stage) {
                double dd = du_diag[j], gg = gu[j];
                if constexpr (DENSE) { if (wd_u) { dd += dq[NT + NX + TRI(j, j)]; gg += dq[NT + NX + NUT + j]; } }
#pragma unroll
                for (int q = 0; q < NX; ++q) { dd += B[q][j] * B[q][j]; gg -= B[q][j] * r[q]; }
                mx_d = fmax(mx_d, dd);
                mx_g = fmax(mx_g, fabs(gg));
            }
#pragma unroll
        for (int i = 0; i < NX; ++i)
            if (has_block && !xfixed[i]) {
                double dd = dx_diag[i] + cin[i] * cin[i], gg = gx[i] - cin[i] * rin;
                if constexpr (DENSE) { if (wd_x) { dd += dq[TRI(i, i)]; gg += dq[NT + i]; } }
#pragma unroll
                for (int q = 0; q < NX; ++q) { dd += A[q][i] * A[q][i]; gg -= A[q][i] * r[q]; }
                if (k >= 1) { dd += SOA(Wbm, i, k - 1); gg += SOA(Wbm, NX + i, k - 1); }
                mx_d = fmax(mx_d, dd);
                mx_g = fmax(mx_g, fabs(gg));
            }
        mx_d = wave_max(mx_d);
        mx_g = wave_max(mx_g);
        double sc_ = wave_sum(cdt), sg_ = wave_sum(gdt);
        if ((tid & 63) == 0) { red[(tid >> 6) * 4 + 0] = mx_d; red[(tid >> 6) * 4 + 1] = mx_g; red[(tid >> 6) * 4 + 2] = sc_; red[(tid >> 6) * 4 + 3] = sg_; }
        fb_barrier();
        double s_cdt = 0, s_gdt = 0;
        mx_d = red[0]; mx_g = red[1]; s_cdt = red[2]; s_gdt = red[3];
#pragma unroll
        for (int w = 1; w < NW; ++w) { mx_d = fmax(mx_d, red[w * 4]); mx_g = fmax(mx_g, red[w * 4 + 1]); s_cdt += red[w * 4 + 2]; s_gdt += red[w * 4 + 3]; }
        if (ARROW) { mx_d = fmax(mx_d, s_cdt); mx_g = fmax(mx_g, fabs(s_gdt)); }
        fb_barrier();
        stop = (mx_g <= LM_EPS1) ? 1 : 0;
        mu   = LM_TAU * mx_d;
        if (mu < 0) mu = 0;
    }
    // H_ii += mu on every inner pass, never undone on reject (:135-138 and the comment at :208)
    const double mu_eff = (fresh ? 0.0 : mu_acc_in) + mu;

    // ---- phase A: eliminate the controls of stage k (they couple only to x_k and x_{k+1}); Schur pieces:
    //      own block k stays in registers, the pieces for block k+1 go to that block's slots (mailbox)
    double Dk[NX][NX], gk[NX], bk[NX], Ck[NX][NX];  // Ck = H'(x_{k+1}, x_k)
    double y2 = 0, zz = 0, zy = 0;  // running sums of |y|^2, |z|^2, z.y over the pivots this lane owns
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        gk[i] = 0; bk[i] = 0;
#pragma unroll
        for (int j = 0; j < NX; ++j) { Dk[i][j] = 0; Ck[i][j] = 0; }
    }
    // Register-rich shapes of the run-to-completion kernel (two / three waves per workgroup, headline stride): the pad slot N of the SoA arrays
    // (NP = N | 1, N even) is a ZERO neighbour -- the cyclic-reduction levels fetch absent neighbours from it instead of zeroing 30 registers
    // per side with selects (measured at the 128-VGPR budget of the four-wave shape: -7 % in the levels, +15 % in the stage phases through
    // register allocation; the rich shapes have the registers).  A spare lane clears it while the stage lanes eliminate their controls.
    constexpr bool ZSLOT = (THREADS < 256) && (NPC > 0) && !ARROW && !GWS;
    if constexpr (ZSLOT) {
        if (k == N && N < NP) {
#pragma unroll
            for (int i = 0; i < NX * NX; ++i) { SOA(Wam, i, N) = 0.0; SOA(Wbm, i, N) = 0.0; }
#pragma unroll
            for (int i = 0; i < NX; ++i) SOA(gv, i, N) = 0.0;
        }
    }
    if (has_stage) {
        double Huu[NU][NU];
#pragma unroll
        for (int a = 0; a < NU; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                double v = 0;
#pragma unroll
                for (int q = 0; q < NX; ++q) v += B[q][a] * B[q][b];
                Huu[a][b] = v;
                Huu[b][a] = v;
            }
        double gu_[NU], bu_[NU];
#pragma unroll
        for (int a = 0; a < NU; ++a) {
            Huu[a][a] += du_diag[a] + mu_eff;
            double gg = gu[a], bb = 0;
            if constexpr (DENSE) if (wd_u) {
                gg += dq[NT + NX + NUT + a];
#pragma unroll
                for (int b = 0; b <= a; ++b) { Huu[a][b] += dq[NT + NX + TRI(a, b)]; if (b < a) Huu[b][a] += dq[NT + NX + TRI(a, b)]; }
            }
#pragma unroll
            for (int q = 0; q < NX; ++q) { gg -= B[q][a] * r[q]; bb += B[q][a] * dc[q]; }
            gu_[a] = gg; bu_[a] = bb;
        }
        chol_inv<NU>(Huu);
        double zx[NU][NX], zp[NU][NX];
#pragma unroll
        for (int a = 0; a < NU; ++a)
#pragma unroll
            for (int c = 0; c < NX; ++c) {
                double v1 = 0, v2 = 0;
#pragma unroll
                for (int q = 0; q < NX; ++q) { v1 += B[q][a] * A[q][c]; v2 += B[q][a] * Cc[q][c]; }
                zx[a][c] = v1; zp[a][c] = v2;
            }
        fwd_solve<NU, NX>(Huu, zx);
        fwd_solve<NU, NX>(Huu, zp);
        fwd_solve_vec<NU>(Huu, gu_);
        if constexpr (ARROW) fwd_solve_vec<NU>(Huu, bu_);
#pragma unroll
        for (int a = 0; a < NU; ++a) {
            y2 += gu_[a] * gu_[a];
            if constexpr (ARROW) { zz += bu_[a] * bu_[a]; zy += bu_[a] * gu_[a]; SOA(zu, a, k) = bu_[a]; }
            SOA(yu, a, k) = gu_[a];
#pragma unroll
            for (int b = 0; b < NU; ++b) SOA(Luu, a * NU + b, k) = Huu[a][b];
#pragma unroll
            for (int c = 0; c < NX; ++c) { SOA(Zx, a * NX + c, k) = zx[a][c]; SOA(Zp, a * NX + c, k) = zp[a][c]; }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double g1 = 0, g2 = 0, b1 = 0, b2 = 0;
#pragma unroll
            for (int q = 0; q < NX; ++q) { g1 -= A[q][i] * r[q]; g2 -= Cc[q][i] * r[q]; b1 += A[q][i] * dc[q]; b2 += Cc[q][i] * dc[q]; }
#pragma unroll
            for (int a = 0; a < NU; ++a) { g1 -= zx[a][i] * gu_[a]; g2 -= zp[a][i] * gu_[a]; b1 -= zx[a][i] * bu_[a]; b2 -= zp[a][i] * bu_[a]; }
            gk[i] = g1; bk[i] = b1;
            SOA(gv, i, k + 1) = g2;
            if constexpr (ARROW) SOA(bv, i, k + 1) = b2;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                double d1 = 0, d2 = 0, cc = 0;
#pragma unroll
                for (int q = 0; q < NX; ++q) { d1 += A[q][i] * A[q][j]; d2 += Cc[q][i] * Cc[q][j]; cc += Cc[q][i] * A[q][j]; }
#pragma unroll
                for (int a = 0; a < NU; ++a) { d1 -= zx[a][i] * zx[a][j]; d2 -= zp[a][i] * zp[a][j]; cc -= zp[a][i] * zx[a][j]; }
                Dk[i][j] = d1;
                Ck[i][j] = cc;
                if (j <= i) SOA(Dm, TRI(i, j), k + 1) = d2;   // partial of D_{k+1}
                SOA(Wam, i * NX + j, k + 1) = cc;             // H'(x_{k+1}, x_k), consumed by block k+1 if it is odd
            }
        }
    }
    fb_barrier();
    STAMP(2);
    // ---- phase B + cyclic-reduction level 0: complete state block k; odd blocks are eliminated at once
    if (has_block) {
        // the mailbox of block k (written by stage k-1; block 0 has none: fetched anyway, in one batch, and discarded)
        double mg[NX], mb[NX], mD[NX][NX], Wa[NX][NX], Wb[NX][NX];
        const bool has_mail = (k >= 1);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            mg[i] = SOA(gv, i, k);
            mb[i] = ARROW ? SOA(bv, i, k) : 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                mD[i][j] = (j <= i) ? SOA(Dm, TRI(i, j), k) : 0.0;
                Wa[i][j] = SOA(Wam, i * NX + j, k);               // H(k, k-1) from stage k-1
                Wb[i][j] = (k + 1 < N) ? Ck[j][i] : 0.0;           // H(k, k+1) = H'(x_{k+1}, x_k)^T
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double g = gk[i] + gx[i] - cin[i] * rin, b = bk[i];
            if (has_mail) { g += mg[i]; b += mb[i]; }
            if constexpr (DENSE) { if (wd_x) g += dq[NT + i]; }
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double d = Dk[i][j] + cin[i] * cin[j];
                if (has_mail) d += mD[i][j];
                if (i == j) d += dx_diag[i] + mu_eff;
                if constexpr (DENSE) { if (wd_x) d += dq[TRI(i, j)]; }
                if (xfixed[i] || xfixed[j]) d = (i == j) ? 1.0 : 0.0;
                Dk[i][j] = d;
            }
            if (xfixed[i]) { g = 0; b = 0; }
            gk[i] = g; bk[i] = b;
        }
        if (k & 1) {  // eliminate: neighbours a = k-1, b = k+1
            chol_inv<NX>(Dk);
            fwd_solve<NX, NX>(Dk, Wa);
            fwd_solve<NX, NX>(Dk, Wb);
            fwd_solve_vec<NX>(Dk, gk);
            if constexpr (ARROW) fwd_solve_vec<NX>(Dk, bk);
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                y2 += gk[q] * gk[q];
                if constexpr (ARROW) { zz += bk[q] * bk[q]; zy += bk[q] * gk[q]; }
#pragma unroll
                for (int c = 0; c < NX; ++c) { SOA(Wam, q * NX + c, k) = Wa[q][c]; SOA(Wbm, q * NX + c, k) = Wb[q][c]; }
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            SOA(gv, i, k) = gk[i];
            if constexpr (ARROW) SOA(bv, i, k) = bk[i];
#pragma unroll
            for (int j = 0; j <= i; ++j) SOA(Dm, TRI(i, j), k) = Dk[i][j];
        }
    }
    fb_barrier();
    STAMP(3);

    // ---- cyclic reduction, levels h = 2, 4, ...: lane t owns the active block a = h*t.  It first applies the Schur updates of
    //      the previous level (neighbours a -+ h/2 were eliminated there); odd t then eliminates a against a -+ h, whose
    //      couplings it forms on the fly from those same neighbours.  The first h >= N leaves block 0 alone: the root.
    // Four lanes share a block (a wave64 instruction costs 4 cycles whether 1 or 64 lanes work, so the per-level instruction count
    // is what the chain pays): every lane applies the Schur updates of both neighbours to its own copy of D (redundant, 36 FMAs, no
    // exchange) and factors it; lane j < NX then produces column j of the new couplings W_a and W_b, lane NX the right-hand side.
    // What differs between the lanes is only WHERE their 3-vector operand comes from / goes to (column j of W, or the rhs slot), i.e.
    // a base pointer and a stride -- the instruction stream is the same.
    // One level as a generic lambda: instantiated for the regular lane mapping (the loop below; everything that depends on the lane's role is
    // loop-invariant there) and, two-wave shape only, once more for the COMPACT mapping of the first level (peeled: with the choice inside the loop the
    // role-dependent pointers became loop-variant and every OTHER level paid 150 - 270 cycles for it -- measured with one stamp per level).
    auto cr_level = [&](const int h, const int lg, auto compact_tag, auto top_tag) {
        constexpr bool COMPACT = decltype(compact_tag)::value;
        constexpr bool TOP     = decltype(top_tag)::value;   // assembly of the top system only (twisted top, below): no block is eliminated at this level
        const int hh    = h >> 1;
        const bool root = (h >= N);                                      // only a == 0 is active then
        static_assert(NX <= 4, "four lanes per block: NX columns (+ one right-hand-side lane when NX <= 3)");
        // NX = 4 leaves no spare lane for the right-hand side: every lane carries it as a SECOND operand (redundantly, same
        // instruction stream) and lane 0 stores it.
        constexpr bool RHS2 = (NX == 4);
        const int nblk  = (N + h - 1) >> lg;
        // Lane -> (block t, role j).  Regular: four lanes per block, THREADS / 4 blocks per round.  COMPACT (two-wave shape; a level with more blocks than
        // quads -- h = 2: 50 blocks, 32 quads -- would take two rounds): only the blocks that are ELIMINATED at this level (odd t) need the column lanes; a
        // block that is merely updated (even t) needs its right-hand-side lane alone.  Four lanes per odd block + one per even block make it ONE round.
        // Quads and single lanes are dealt to the two waves half and half: the blocks of a level are 2 doubles apart in the SoA arrays (16 distinct
        // bank groups), so a wave that held all the single lanes would serialise its LDS reads four-fold (measured: +1.5 k cycles per pass).
        int j = tid & 3, t_first = tid >> 2, t_step = THREADS / 4;
        if constexpr (COMPACT) {
            const int n_el = nblk >> 1, n_ne = nblk - n_el;
            const int nq0 = (n_el + 1) >> 1, ns0 = n_ne >> 1;   // wave 0's quads / single lanes
            const int w = tid >> 6, l = tid & 63;
            const int nq = w ? n_el - nq0 : nq0, ns = w ? n_ne - ns0 : ns0;
            const int qb = w ? nq0 : 0, sb = w ? ns0 : 0;
            t_step = nblk;
            if (l < 4 * nq) t_first = 2 * (qb + (l >> 2)) + 1;
            else if (l < 4 * nq + ns) { t_first = 2 * (sb + l - 4 * nq); j = RHS2 ? 0 : NX; }
            else t_first = nblk;
        }
        const bool vec  = !RHS2 && (j == NX);
        const bool col  = (j < NX);
        for (int t = t_first; t < nblk; t += t_step) {   // (one round whenever 4 * ceil(N / h) <= THREADS, or compact)
        const int a      = h * t;
        const bool has_m = (a - hh >= 0), has_p = (a + hh < N);
        const int em = has_m ? a - hh : (ZSLOT ? N : a), ep = has_p ? a + hh : (ZSLOT ? N : a);       // clamped: absent neighbours are fetched from a and zeroed (ZSLOT: from the zero slot)
        const bool elim = (t & 1);
        // operand of this lane: column j of the far coupling of each neighbour, or (lane 3) the neighbour's rhs
        const int jc     = col ? j : 0;               // (a spare lane, NX < 3, mirrors column 0 and stores nothing)
        const double* om = vec ? gv : Wam + jc * NP;  // em: far side is a - h  -> W_a(em)
        const double* op = vec ? gv : Wbm + jc * NP;  // ep: far side is a + h  -> W_b(ep)
        const int os     = vec ? NP : NX * NP;
        double D[NX][NX], g[NX], bb[NX], Xm[NX][NX], Xp[NX][NX], ym[NX], yp[NX], zm[NX], zp[NX];
        double ymr[NX], ypr[NX];   // RHS2: the neighbours' right-hand sides
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            g[q]  = SOA(gv, q, a);
            bb[q] = ARROW ? SOA(bv, q, a) : 0.0;
            ym[q] = om[q * os + em];
            yp[q] = op[q * os + ep];
            ymr[q] = RHS2 ? SOA(gv, q, em) : 0.0;
            ypr[q] = RHS2 ? SOA(gv, q, ep) : 0.0;
            zm[q] = ARROW ? SOA(bv, q, em) : 0.0;
            zp[q] = ARROW ? SOA(bv, q, ep) : 0.0;
#pragma unroll
            for (int c = 0; c < NX; ++c) {
                D[q][c]  = (c <= q) ? SOA(Dm, TRI(q, c), a) : 0.0;
                Xm[q][c] = SOA(Wbm, q * NX + c, em);   // em's coupling to a (a is its b-side)
                Xp[q][c] = SOA(Wam, q * NX + c, ep);   // ep's coupling to a (a is its a-side)
            }
        }
        // absent neighbour: BOTH factors of every product are zeroed -- the clamped loads may have fetched never-written LDS
        // (whatever an earlier kernel left there, possibly NaN bit patterns; 0 x NaN would poison the block)
        if (!ZSLOT && !has_m) {
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                ym[q] = 0.0; zm[q] = 0.0; ymr[q] = 0.0;
#pragma unroll
                for (int c = 0; c < NX; ++c) Xm[q][c] = 0.0;
            }
        }
        if (!ZSLOT && !has_p) {
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                yp[q] = 0.0; zp[q] = 0.0; ypr[q] = 0.0;
#pragma unroll
                for (int c = 0; c < NX; ++c) Xp[q][c] = 0.0;
            }
        }
        // Schur updates: D -= X^T X (both neighbours), v = X^T (operand)
        double vm[NX], vp[NX], wm[NX], wp[NX], vr[NX];
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            double s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
#pragma unroll
            for (int t = 0; t < NX; ++t) {
                s1 += Xm[t][q] * ym[t];
                s2 += Xp[t][q] * yp[t];
                if constexpr (ARROW) { s3 += Xm[t][q] * zm[t]; s4 += Xp[t][q] * zp[t]; }
                if constexpr (RHS2) s5 += Xm[t][q] * ymr[t] + Xp[t][q] * ypr[t];
            }
            vm[q] = s1; vp[q] = s2; wm[q] = s3; wp[q] = s4; vr[q] = s5;
#pragma unroll
            for (int c = 0; c <= q; ++c) {
                double dd = 0;
#pragma unroll
                for (int t = 0; t < NX; ++t) dd += Xm[t][q] * Xm[t][c] + Xp[t][q] * Xp[t][c];
                D[q][c] -= dd;
            }
        }
        // lane 3: rhs (and border column) of block a; lanes 0..2: H(a, a-h) = -W_b(em)^T W_a(em), H(a, a+h) = -W_a(ep)^T W_b(ep)
        double c1[NX], c2[NX], r1[NX], r2[NX];
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            c1[q] = vec ? g[q] - (vm[q] + vp[q]) : -vm[q];
            c2[q] = vec ? bb[q] - (wm[q] + wp[q]) : -vp[q];
            r1[q] = RHS2 ? g[q] - vr[q] : 0.0;
            r2[q] = (RHS2 && ARROW) ? bb[q] - (wm[q] + wp[q]) : 0.0;
        }
        if (!TOP && (elim || root)) {
            chol_inv<NX>(D);
            fwd_solve_vec<NX>(D, c1);
            if (ARROW || !vec) fwd_solve_vec<NX>(D, c2);
            if constexpr (RHS2) {
                fwd_solve_vec<NX>(D, r1);
                if constexpr (ARROW) fwd_solve_vec<NX>(D, r2);
                if (j == 0) {
#pragma unroll
                    for (int q = 0; q < NX; ++q) {
                        y2 += r1[q] * r1[q];
                        if constexpr (ARROW) { zz += r2[q] * r2[q]; zy += r2[q] * r1[q]; }
                    }
                }
                if (root && !ARROW) bwd_solve_vec<NX>(D, r1);
            }
            if (vec) {
#pragma unroll
                for (int q = 0; q < NX; ++q) {
                    y2 += c1[q] * c1[q];
                    if constexpr (ARROW) { zz += c2[q] * c2[q]; zy += c2[q] * c1[q]; }
                }
                if (root && !ARROW) bwd_solve_vec<NX>(D, c1);  // no border: the root is back-substituted right away
            }
        }
        if (vec) {
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                SOA(gv, q, a) = c1[q];
                if constexpr (ARROW) SOA(bv, q, a) = c2[q];
#pragma unroll
                for (int c = 0; c <= q; ++c) SOA(Dm, TRI(q, c), a) = D[q][c];
            }
        }
        else if (col && (TOP || (elim && !root))) {   // (TOP: the raw couplings H(a, a -+ h) of every top block, for the chains)
#pragma unroll
            for (int q = 0; q < NX; ++q) { SOA(Wam, q * NX + j, a) = c1[q]; SOA(Wbm, q * NX + j, a) = c2[q]; }
        }
        if constexpr (RHS2) {
            if (j == 0) {
#pragma unroll
                for (int q = 0; q < NX; ++q) {
                    SOA(gv, q, a) = r1[q];
                    if constexpr (ARROW) SOA(bv, q, a) = r2[q];
#pragma unroll
                    for (int c = 0; c <= q; ++c) SOA(Dm, TRI(q, c), a) = D[q][c];
                }
            }
        }
        }
    };
    int hroot = 2, h_first = 2, lg_first = 1;
    if constexpr (CORBO_HIP_COMPACT_LEVELS && THREADS == 128 && !GWS) {
        const int nblk2 = (N + 1) >> 1, n_el = nblk2 >> 1, n_ne = nblk2 - n_el, nq0 = (n_el + 1) >> 1, ns0 = n_ne >> 1;
        if (4 * nblk2 > THREADS && 4 * nq0 + ns0 <= 64 && 4 * (n_el - nq0) + (n_ne - ns0) <= 64) {   // (more blocks than quads at h = 2: N > 64, so h = 2 is not the root level)
            cr_level(2, 1, std::true_type{}, std::false_type{});
#ifdef CORBO_HIP_LEVEL_STAMPS
            STAMP(8 + 1);
#endif
            fb_barrier();
            h_first = 4; lg_first = 2;
        }
    }
    // TWISTED TOP (two-wave shape, headline stride).  Once at most eight blocks are left (N = 100: the seven blocks 0, 16, ..., 96 at h = 16) the remaining
    // block-tridiagonal system is not reduced by three or four more levels -- each a barrier, 33 LDS loads per lane, two-sided Schur updates, a 3 x 3
    // Cholesky and the stores, ~ 2 k cycles whatever the number of blocks -- but ASSEMBLED once (one level without eliminations: updated diagonal blocks,
    // right-hand sides and the raw couplings between neighbouring top blocks) and eliminated by two lanes from both ends towards the middle block
    // (block Thomas steps: one-sided update, the running block stays in registers, no barrier, the next block's loads ride under the Cholesky), then
    // back-substituted outwards by the same two lanes.  Sequential depth: ceil(nb / 2) steps instead of log2(nb) + 1 levels, but a step is less than
    // half a level.  Same matrix, another elimination order for its last few blocks: rounding-level differences (like the partitioned chain of the
    // big-block family); |y|^2 sums the same quantity g^T H^-1 g.
    // (the running blocks, the factors and the couplings of the chains stay in REGISTERS: the step count is a compile-time constant of the headline
    //  stride -- N is NPC - 1 or NPC --, the loops are unrolled, nothing but the solution of the top blocks goes back to LDS)
    constexpr int TW_HS = twist_level(NPC), TW_NB = TW_HS ? (NPC + TW_HS - 1) / TW_HS : 0;
    constexpr bool TWIST = CORBO_HIP_TWISTED_TOP && THREADS == 128 && NPC > 0 && !ARROW && !GWS && !DENSE && NX <= 3 && TW_HS > 0 &&
                           (NPC - 1 + TW_HS - 1) / TW_HS == TW_NB;   // (the same block count for N = NPC - 1 and N = NPC)
    const int hs = TWIST ? TW_HS : 0;   // the level at which the top system is assembled (0: never)
    for (int h = h_first, lg = lg_first;; h <<= 1, ++lg) {   // h = 2^lg
        if constexpr (TWIST) {
            if (h == hs) {
                cr_level(h, lg, std::false_type{}, std::true_type{});
                fb_barrier();
#ifdef CORBO_HIP_LEVEL_STAMPS
                STAMP(13);
#endif
                if (tid < 2) {   // lane 0: blocks 0 .. m - 1 upwards, then the middle block m; lane 1: blocks nb - 1 .. m + 1 downwards
                    constexpr int nb = TW_NB, m = nb >> 1, NSMAX = m;   // (m >= nb - 1 - m)
                    const bool up = (tid == 0);
                    const int nsteps = up ? m : nb - 1 - m;
                    const int di = up ? 1 : -1;
                    const double* const Wf = up ? Wbm : Wam;   // raw coupling towards the next block in this lane's direction, H(a, a +- hs)
                    const int i0 = up ? 0 : nb - 1;
                    double D[NX][NX], g[NX];
                    double Ls[NSMAX][NX][NX], Ys[NSMAX][NX][NX], ys[NSMAX][NX];   // per eliminated block: L (diagonal inverted), Y = L^-1 H, y = L^-1 g
#pragma unroll
                    for (int q = 0; q < NX; ++q) {
                        g[q] = SOA(gv, q, hs * i0);
#pragma unroll
                        for (int c = 0; c < NX; ++c) D[q][c] = (c <= q) ? SOA(Dm, TRI(q, c), hs * i0) : 0.0;
                    }
#pragma unroll
                    for (int s_ = 0; s_ < NSMAX; ++s_) {
                        if (s_ < nsteps) {
                            const int a = hs * (i0 + s_ * di), an = a + hs * di;
                            double Dn[NX][NX], gn[NX];
#pragma unroll
                            for (int q = 0; q < NX; ++q) {
                                gn[q] = SOA(gv, q, an);
#pragma unroll
                                for (int c = 0; c < NX; ++c) { Ys[s_][q][c] = SOA(Wf, q * NX + c, a); Dn[q][c] = (c <= q) ? SOA(Dm, TRI(q, c), an) : 0.0; }
                            }
                            chol_inv<NX>(D);
                            fwd_solve<NX, NX>(D, Ys[s_]);
                            fwd_solve_vec<NX>(D, g);
#pragma unroll
                            for (int q = 0; q < NX; ++q) {
                                y2 += g[q] * g[q];
                                ys[s_][q] = g[q];
#pragma unroll
                                for (int c = 0; c < NX; ++c) Ls[s_][q][c] = D[q][c];
                            }
#pragma unroll
                            for (int c1_ = 0; c1_ < NX; ++c1_) {
                                double sv = 0;
#pragma unroll
                                for (int q = 0; q < NX; ++q) sv += Ys[s_][q][c1_] * g[q];
                                gn[c1_] -= sv;
#pragma unroll
                                for (int c2_ = 0; c2_ <= c1_; ++c2_) {
                                    double sd = 0;
#pragma unroll
                                    for (int q = 0; q < NX; ++q) sd += Ys[s_][q][c1_] * Ys[s_][q][c2_];
                                    Dn[c1_][c2_] -= sd;
                                }
                            }
#pragma unroll
                            for (int q = 0; q < NX; ++q) {
                                g[q] = gn[q];
#pragma unroll
                                for (int c = 0; c < NX; ++c) D[q][c] = Dn[q][c];
                            }
                        }
                    }
#ifdef CORBO_HIP_LEVEL_STAMPS
                    asm volatile("" :: "v"(D[0][0]), "v"(g[0]));
                    STAMP(14);
#endif
                    // the middle block: both lanes hold their one-sided version  D_m - U_side, g_m - v_side;  lane 0 combines them with the assembled block
                    // (lane 1's version reaches lane 0 through DPP moves inside the pair: no LDS round trip)
                    const int am = hs * m;
                    double xn[NX];
                    {
                        double Do[NX][NX], go[NX];
#pragma unroll
                        for (int q = 0; q < NX; ++q) {
                            go[q] = pair_bcast1(g[q]);
#pragma unroll
                            for (int c = 0; c <= q; ++c) Do[q][c] = pair_bcast1(D[q][c]);
                        }
                        if (up) {
#pragma unroll
                            for (int q = 0; q < NX; ++q) {
                                g[q] += go[q] - SOA(gv, q, am);
#pragma unroll
                                for (int c = 0; c <= q; ++c) D[q][c] += Do[q][c] - SOA(Dm, TRI(q, c), am);
                            }
                            chol_inv<NX>(D);
                            fwd_solve_vec<NX>(D, g);
#pragma unroll
                            for (int q = 0; q < NX; ++q) y2 += g[q] * g[q];
                            bwd_solve_vec<NX>(D, g);
#pragma unroll
                            for (int q = 0; q < NX; ++q) SOA(gv, q, am) = g[q];
                        }
                    }
                    // x_m to the other chain's lane: a quad-local broadcast from lane 0 (both lanes of the pair are active)
#pragma unroll
                    for (int q = 0; q < NX; ++q) xn[q] = pair_bcast0(g[q]);
#ifdef CORBO_HIP_LEVEL_STAMPS
                    STAMP(15);
#endif
                    // back-substitution outwards from the middle:  x_i = L_i^-T (y_i - Y_i x_next), everything in registers
#pragma unroll
                    for (int s_ = NSMAX - 1; s_ >= 0; --s_) {
                        if (s_ < nsteps) {
                            const int a = hs * (i0 + s_ * di);
                            double v[NX];
#pragma unroll
                            for (int q = 0; q < NX; ++q) {
                                double t_ = ys[s_][q];
#pragma unroll
                                for (int c = 0; c < NX; ++c) t_ -= Ys[s_][q][c] * xn[c];
                                v[q] = t_;
                            }
                            bwd_solve_vec<NX>(Ls[s_], v);
#pragma unroll
                            for (int q = 0; q < NX; ++q) { SOA(gv, q, a) = v[q]; xn[q] = v[q]; }
                        }
                    }
                }
                hroot = hs;
#ifdef CORBO_HIP_LEVEL_STAMPS
                STAMP(8 + lg);
#endif
                break;
            }
        }
        cr_level(h, lg, std::false_type{}, std::false_type{});
        hroot = h;
#ifdef CORBO_HIP_LEVEL_STAMPS
        STAMP(8 + lg);
#endif
        if (h >= N) break;
        fb_barrier();
    }
    STAMP(4);
    // ---- reductions: |y|^2 (= delta^T rhs), and for the arrowhead the last pivot
    double ddt = 0;
    {
        double a0 = wave_sum(y2), a1 = 0, a2 = 0, a3 = 0, a4 = 0;
        if constexpr (ARROW) { a1 = wave_sum(zz); a2 = wave_sum(zy); a3 = wave_sum(cdt); a4 = wave_sum(gdt); }
        if ((tid & 63) == 0) { double* rr = red + (tid >> 6) * 5; rr[0] = a0; rr[1] = a1; rr[2] = a2; rr[3] = a3; rr[4] = a4; }
        fb_barrier();
        a0 = a1 = a2 = a3 = a4 = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { a0 += red[w * 5]; a1 += red[w * 5 + 1]; a2 += red[w * 5 + 2]; a3 += red[w * 5 + 3]; a4 += red[w * 5 + 4]; }
        y2 = a0;
        if constexpr (ARROW) {
            const double piv  = (a3 + mu_eff) - a1;   // H(dt,dt) + damping - |z|^2
            const double linv = rsqrt(piv);
            const double ydt  = (a4 - a2) * linv;
            y2 += ydt * ydt;
            ddt = ydt * linv;
        }
    }
    if constexpr (ARROW) {  // y := y - z * delta_dt  (back-substitution of the last pivot), then the root
        if (has_block)
#pragma unroll
            for (int q = 0; q < NX; ++q) SOA(gv, q, k) -= SOA(bv, q, k) * ddt;
        if (has_stage)
#pragma unroll
            for (int a = 0; a < NU; ++a) SOA(yu, a, k) -= SOA(zu, a, k) * ddt;
        fb_barrier();
        if (tid == 0) {
            double L[NX][NX], y[NX];
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                y[q] = SOA(gv, q, 0);
#pragma unroll
                for (int c = 0; c < NX; ++c) L[q][c] = (c <= q) ? SOA(Dm, TRI(q, c), 0) : 0.0;
            }
            bwd_solve_vec<NX>(L, y);
#pragma unroll
            for (int q = 0; q < NX; ++q) SOA(gv, q, 0) = y[q];
        }
        fb_barrier();
    }
    STAMP(5);
    // the accepted iterate of this lane's stage (for x + delta below): requested now, the back-substitution hides the latency
    double xdt = 0.0;
    if (!RECOMP || keep_valid) {
#pragma unroll
        for (int e = 0; e < S; ++e) xk[e] = ((e < NX) ? has_block : has_stage) ? xin[k * S + e] : 0.0;
    }
    if (tid == 0) xdt = xin[p.off_dt];
    // ---- back-substitution down the elimination tree.  Four lanes per block again: lane q forms row q of
    //      v = y - W_a x_a - W_b x_b, the three (two) numbers are broadcast inside the quad with DPP moves, every lane solves
    //      L^T x = v and lane q stores x_q.
    //      (Fetching the factor data of the next level ahead of the barrier was tried: its 26 registers cost more than the LDS
    //      latency it hides, at the 128-VGPR budget.)
    {
        const int q  = tid & 3;
        const int qc = (q < NX) ? q : NX - 1;   // spare lanes mirror the last row and store nothing
        double L[NX][NX], wa[NX], wb[NX], yq = 0.0;
        auto fetch = [&](int hh, int t) {
            const int i = hh * (2 * t + 1);
            if (i < N) {
                yq = SOA(gv, qc, i);
#pragma unroll
                for (int c = 0; c < NX; ++c) {
                    wa[c] = SOA(Wam, qc * NX + c, i);
                    wb[c] = SOA(Wbm, qc * NX + c, i);
#pragma unroll
                    for (int r = 0; r < NX; ++r) L[r][c] = (c <= r) ? SOA(Dm, TRI(r, c), i) : 0.0;
                }
            }
        };
        auto finish = [&](int hh, int t) {
            const int i = hh * (2 * t + 1);
            const int a = i - hh, b = i + hh;
            const int bc = (b < N) ? b : a;  // W_b is zero when there is no right neighbour: any finite operand will do
            double v = yq;
#pragma unroll
            for (int c = 0; c < NX; ++c) v -= wa[c] * SOA(gv, c, a) + wb[c] * SOA(gv, c, bc);
            double x[NX];
            x[0] = quad_bcast<0>(v);
            if constexpr (NX > 1) x[1] = quad_bcast<1>(v);
            if constexpr (NX > 2) x[2] = quad_bcast<2>(v);
            if constexpr (NX > 3) x[3] = quad_bcast<3>(v);
            bwd_solve_vec<NX>(L, x);
            double xq = x[0];
#pragma unroll
            for (int r = 1; r < NX; ++r) xq = (q == r) ? x[r] : xq;
            if (q < NX) SOA(gv, q, i) = xq;
        };
        // SOLO last level (two-wave shape): the level h = 1 has N / 2 blocks -- more than the workgroup has quads (50 against 32 at N = 100), two rounds of
        // the quad scheme.  One lane per block instead: every row of  v = y - W_a x_a - W_b x_b  and the triangular solve in the same lane (no exchange,
        // three times the multiply-adds of a quad lane, but ONE round); the factor data of the block -- final since the forward sweep -- is requested
        // before the levels above run, their barriers hide the latency.  The blocks are dealt to the two waves half and half (LDS bank groups, see the
        // compact level of the forward sweep).
        constexpr bool SOLO_OK = CORBO_HIP_SOLO_LAST_LEVEL && (THREADS == 128) && (NPC > 0) && !GWS && !ARROW;
        const int nb1 = N >> 1, nb1h = (nb1 + 1) >> 1;                       // blocks 1, 3, ..; wave 0's share
        const bool solo = SOLO_OK && (4 * nb1 > THREADS) && nb1h <= 64 && hroot > 2;
        const int tsolo = (tid < 64) ? tid : nb1h + (tid - 64);
        const bool solo_on = solo && ((tid < 64) ? (tid < nb1h) : (tsolo < nb1));
        double sL[NX][NX], sWa[NX][NX], sWb[NX][NX], sy[NX];
        if constexpr (SOLO_OK) {
            if (solo_on) {
                const int i = 2 * tsolo + 1;
#pragma unroll
                for (int r = 0; r < NX; ++r) {
                    sy[r] = SOA(gv, r, i);
#pragma unroll
                    for (int c = 0; c < NX; ++c) { sWa[r][c] = SOA(Wam, r * NX + c, i); sWb[r][c] = SOA(Wbm, r * NX + c, i); sL[r][c] = (c <= r) ? SOA(Dm, TRI(r, c), i) : 0.0; }
                }
            }
        }
        int h = hroot >> 1;
        fetch(h, tid >> 2);
        constexpr bool PREFETCH = false && (THREADS < 256) && (NPC > 0) && !GWS;   // measured in the two-wave shape: back-substitution 5.9 k -> 7.0 k cycles -- off   // register-rich shapes: the factor data of the NEXT level (final since the
        for (; h >= 1; h >>= 1) {                                         // forward sweep) is requested ahead of this level's arithmetic and barrier
            if constexpr (PREFETCH) {
                double L0[NX][NX], wa0[NX], wb0[NX];
                const double yq0 = yq;
#pragma unroll
                for (int c = 0; c < NX; ++c) {
                    wa0[c] = wa[c]; wb0[c] = wb[c];
#pragma unroll
                    for (int r = 0; r < NX; ++r) L0[r][c] = L[r][c];
                }
                if (h > 1) fetch(h >> 1, tid >> 2);   // into L / wa / wb / yq; this level works on the copies
                if (h * (2 * (tid >> 2) + 1) < N) {
                    const int t = tid >> 2, i = h * (2 * t + 1), a = i - h, b = i + h, bc = (b < N) ? b : a;
                    double v = yq0;
#pragma unroll
                    for (int c = 0; c < NX; ++c) v -= wa0[c] * SOA(gv, c, a) + wb0[c] * SOA(gv, c, bc);
                    double x[NX];
                    x[0] = quad_bcast<0>(v);
                    if constexpr (NX > 1) x[1] = quad_bcast<1>(v);
                    if constexpr (NX > 2) x[2] = quad_bcast<2>(v);
                    if constexpr (NX > 3) x[3] = quad_bcast<3>(v);
                    bwd_solve_vec<NX>(L0, x);
                    double xq = x[0];
#pragma unroll
                    for (int r = 1; r < NX; ++r) xq = (q == r) ? x[r] : xq;
                    if (q < NX) SOA(gv, q, i) = xq;
                }
                // more blocks than quads: only at h = 1 (N / 4 <= THREADS / 4 blocks at h = 2), where nothing was prefetched
                for (int t = (tid >> 2) + THREADS / 4; h * (2 * t + 1) < N; t += THREADS / 4) { fetch(h, t); finish(h, t); }
                fb_barrier();
            }
            else {
            if constexpr (SOLO_OK) {
                if (h == 1 && solo) {
                    if (solo_on) {
                        const int i = 2 * tsolo + 1, a = i - 1, b = i + 1, bc = (b < N) ? b : a;
                        double v[NX];
#pragma unroll
                        for (int r = 0; r < NX; ++r) {
                            double t_ = sy[r];
#pragma unroll
                            for (int c = 0; c < NX; ++c) t_ -= sWa[r][c] * SOA(gv, c, a) + sWb[r][c] * SOA(gv, c, bc);
                            v[r] = t_;
                        }
                        bwd_solve_vec<NX>(sL, v);
#pragma unroll
                        for (int r = 0; r < NX; ++r) SOA(gv, r, i) = v[r];
                    }
                    fb_barrier();
                    break;
                }
            }
            if (h * (2 * (tid >> 2) + 1) < N) finish(h, tid >> 2);
            for (int t = (tid >> 2) + THREADS / 4; h * (2 * t + 1) < N; t += THREADS / 4) { fetch(h, t); finish(h, t); }  // long horizons
            fb_barrier();
            if (h > 1 && !(SOLO_OK && solo && h == 2)) fetch(h >> 1, tid >> 2);
            }
        }
    }
    STAMP(6);
    // ---- controls, trial iterate, step norms
    double dn2 = 0;
    // the trial iterate goes to HBM for the sweep kernel / phase of the NEXT launch, or -- run-to-completion kernel -- straight into
    // the LDS array the sweep phase of this launch evaluates it from
    double* xt = xt_lds ? xt_lds : p.xt + (size_t)inst * p.nvs;
    double* dl = p.delta_out ? p.delta_out + (size_t)inst * p.nvs : nullptr;
    if (has_block) {
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const double d = xfixed[q] ? 0.0 : SOA(gv, q, k);
            dn2 += d * d;
            xt[k * S + q] = xk[q] + d;
            if (dl) dl[k * S + q] = d;
        }
    }
    if (has_stage) {
        double L[NU][NU], y[NU];
#pragma unroll
        for (int a = 0; a < NU; ++a) {
            double v = SOA(yu, a, k);
#pragma unroll
            for (int c = 0; c < NX; ++c) v -= SOA(Zx, a * NX + c, k) * SOA(gv, c, k) + SOA(Zp, a * NX + c, k) * SOA(gv, c, k + 1);
            y[a] = v;
#pragma unroll
            for (int b = 0; b < NU; ++b) L[a][b] = SOA(Luu, a * NU + b, k);
        }
        bwd_solve_vec<NU>(L, y);
#pragma unroll
        for (int a = 0; a < NU; ++a) {
            dn2 += y[a] * y[a];
            xt[k * S + NX + a] = xk[NX + a] + y[a];
            if (dl) dl[k * S + NX + a] = y[a];
        }
    }
    if (tid == 0) {
        if (ARROW) { dn2 += ddt * ddt; xt[p.off_dt] = xdt + ddt; }
        else xt[p.off_dt] = xdt;
        if (dl) dl[p.off_dt] = ARROW ? ddt : 0.0;
        if (p.off_dt + 1 < p.nvs) { xt[p.off_dt + 1] = 0.0; if (dl) dl[p.off_dt + 1] = 0.0; }
    }
    {
        double a0 = wave_sum(dn2);
        fb_barrier();
        if ((tid & 63) == 0) red[tid >> 6] = a0;
        fb_barrier();
        if (tid == 0) {
            dn2 = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) dn2 += red[w];
            st->mu     = mu;
            st->mu_acc = mu_eff;
            st->first  = 0;
            st->fresh  = 0;
            st->n_fact += 1;
            st->inner += 1;
            const double dnorm = sqrt(dn2);
            st->dnorm = dnorm;
            int no_trial;
            if (dnorm <= LM_EPS2) { stop = 1; no_trial = 1; }                    // :151-154
            else { no_trial = 0; st->den = mu * dn2 + y2; }                      // delta^T (mu delta + rhs), delta^T rhs = |y|^2
            st->stop     = stop;
            st->no_trial = no_trial;
        }
    }
    STAMP(7);
}

// long horizons: the same phases with the workspace in HBM (FactorParams::work), one lane per stage in a 1024-thread workgroup
// HYB (round 6): the state-block arrays in LDS, only the eliminated controls' arrays in the HBM workspace (factor_body, GWS = 2) -- wherever they fit
// THREADS / MINW: up to 512 grid points eight waves do (one lane per stage) -- with 128 VGPRs where the LDS holds TWO such workgroups per CU (up to ~ 370 grid points
// for nx = 3: 3.77 -> 2.93 ms per solve of 1024 unicycle OCPs at N = 300), with all the registers of two waves per SIMD where it holds one (N = 512: 4.9 -> 4.7)
template <int NX, int NU, bool ARROW, bool DENSE = false, bool HYB = false, int THREADS = 1024, int MINW = 4>
__global__ __launch_bounds__(THREADS, MINW) void factor_long_kernel(const FactorParams p)
{
    __shared__ __attribute__((aligned(16))) LmState sl_;
    extern __shared__ __attribute__((aligned(16))) double long_lds[];   // (HYB)
    const int inst = blockIdx.x + p.inst0;
    lm_state_in(&sl_, p.st + inst, threadIdx.x);
    __syncthreads();
    if constexpr (HYB)
        factor_body<NX, NU, THREADS, ARROW, 0, DENSE, 2>(p, &sl_, long_lds, inst, threadIdx.x, false, nullptr, nullptr, nullptr, false, NoHook{}, nullptr, nullptr, p.work + (size_t)inst * p.work_stride);
    else
        factor_body<NX, NU, THREADS, ARROW, 0, DENSE, 1>(p, &sl_, p.work + (size_t)inst * p.work_stride, inst, threadIdx.x, false);
    __syncthreads();
    lm_state_out(p.st + inst, &sl_, threadIdx.x);
}
// LDS of the HYB variant: D (packed), W_a, W_b, rhs per block, the reduction scratch of sixteen waves, the border column (free dt)
template <int NX, int NU>
__host__ __device__ constexpr size_t factor_long_hyb_lds_doubles(int N, bool arrow)
{
    return (size_t)(NX * (NX + 1) / 2 + 2 * NX * NX + NX) * (N | 1) + 8 * 16 + (arrow ? (size_t)NX * (N | 1) : 0) + 2;
}
template <int NX, int NU>
__host__ __device__ constexpr size_t factor_long_work_doubles(int N, bool arrow)
{
    return (size_t)FactorLds<NX, NU>::off_red(N | 1) + 8 * 16 + (arrow ? (size_t)(NU + NX) * (N | 1) : 0) + 2;
}

template <int NX, int NU, int THREADS, bool ARROW, bool DENSE = false>
__global__ __launch_bounds__(THREADS) void factor_kernel(const FactorParams p)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int inst = blockIdx.x + p.inst0;
    const int ftot = FactorLds<NX, NU>::total(p.N | 1, ARROW);
    LmState* sl = reinterpret_cast<LmState*>(smem + ((ftot > p.nnz_pad ? ftot : p.nnz_pad) + 1) / 2 * 2);
    lm_state_in(sl, p.st + inst, threadIdx.x);
    __syncthreads();
    factor_body<NX, NU, THREADS, ARROW, 0, DENSE>(p, sl, smem, inst, threadIdx.x, false);
    __syncthreads();
    lm_state_out(p.st + inst, sl, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// assemble + factor + solve for LARGE stage blocks (cfg 5: quadrotor, nx = 12, nu = 4, N = 200)
//
// The block-tridiagonal system does not fit LDS (3 x 144 doubles per stage), and the lane-per-block scheme does not apply to
// 12 x 12 blocks.  Three kernels per factorisation, data handed over in an HBM workspace (BigWs, 572 doubles per stage):
//   big_first_kernel    (first factorisation of a solve only; one wave per instance) mu = tau max diag(J^T J), stop flag;
//   big_assemble_kernel (one wave per (stage, instance) -- everything that does not depend on the neighbours, N x batch waves):
//                       G^T G of the local defect Jacobian G = [A | B | C] (12 x 28 -> 28 x 28; fp64 MFMA
//                       v_mfma_f64_16x16x4f64, a real contraction here), elimination of the controls, the stage's parts of the
//                       diagonal blocks of x_k and x_{k+1}, their coupling, the right-hand sides;
//   big_chain_kernel    (one wave per instance) the sequential part: block Cholesky in natural order over the 12 x 12 state
//                       blocks (Schur complement Y Y^T on the matrix cores), backward sweep, controls, trial iterate, LM state.
// Same LM bookkeeping as factor_body.
// the two sizes of BigLds the host needs for a descriptor's (nx, nu) (checked against the structure in every unit that uses it)
constexpr int big_ws_stage(int nx, int nu)
{
    const int base = (3 * nx * nx + nu * nu + 2 * nu * nx + 2 * nx + nu + 2) & ~1;
    return (base + 3 * nx + nu + 4 + 1) & ~1;   // + the border (free dt) parts: BX, BN, ZU, four scalars, YV2
}
constexpr int big_lds_total(int nx, int nu)
{
    const int half = (5 * nx + 2 * (nx + nu) + nx * (nx + nu) + 8 + nx * nx + 1) & ~1, s = nx + nu;
    return 2 * half + ((s * s + 2 * s + nu * nu + 2 * nu * nx + 2 * nu + 1) & ~1);
}

template <int NX, int NU>
struct BigLds {
    // LDS of the stage kernel.  The x_{k+1} block C of the local defect Jacobian [A | B | C] is DIAGONAL (only e_i depends on
    // x_{k+1,i}), so C lives as NX numbers and G^T G is the S x S product [A B]^T [A B] plus row / column scalings.
    // per interval (two per wave):
    static constexpr int G = 0;                        // [NX][S]  [A | B]
    static constexpr int CD = G + NX * (NX + NU);      // [NX]     diagonal of C
    static constexpr int R = CD + NX;                  // [NX]     defect residual
    static constexpr int DIAG = R + NX;                // [NX+NU]  diagonal (single-entry row) contributions of the interval's components
    static constexpr int GDIAG = DIAG + NX + NU;       // [NX+NU]
    static constexpr int CIN = GDIAG + NX + NU;        // [NX]     inequality row
    static constexpr int FIX = CIN + NX;               // [NX]     fixed flags (as doubles)
    static constexpr int RED = FIX + NX;               // [8]
    static constexpr int CM = RED + 8;                 // [NX][NX] the x_{k+1} block C when it is dense (collocation defects; shooting: diagonal, CD)
    static constexpr int DCV = CM + NX * NX;           // [NX]     the dt column of the defect edge (free-dt grids)
    static constexpr int HALF = (DCV + NX + 1) & ~1;
    // shared by the two assemble turns of a wave:
    static constexpr int M = 0;                        // [S][S]   [A B]^T [A B]
    static constexpr int GM = M + (NX + NU) * (NX + NU);   // [S] -[A B]^T r
    static constexpr int LUU = GM + NX + NU;           // [NU][NU]
    static constexpr int ZX = LUU + NU * NU;           // [NU][NX]
    static constexpr int ZP = ZX + NU * NX;            // [NU][NX]
    static constexpr int YU = ZP + NU * NX;            // [NU]
    static constexpr int HB = YU + NU;                 // [S]  border (free dt): [A B]^T d,  d = the dt column
    static constexpr int ZUB = HB + NX + NU;           // [NU] L_uu^{-1} (border's control part)
    static constexpr int SHARED = (ZUB + NU + 1) & ~1;
    static constexpr int TOTAL = 2 * HALF + SHARED;    // doubles per wave (7.4 KB for nx = 12, nu = 4: the register budget, not LDS, bounds the occupancy)
    // HBM workspace per stage
    static constexpr int WS_L = 0;                    // [NX][NX] assemble: own parts of the diagonal block of x_k ; chain: L_k
    static constexpr int WS_Y = WS_L + NX * NX;       // [NX][NX] assemble: coupling H'(x_{k+1}, x_k)              ; chain: Y_k
    static constexpr int WS_DN = WS_Y + NX * NX;      // [NX][NX] this stage's contribution to the diagonal block of x_{k+1}
    static constexpr int WS_LUU = WS_DN + NX * NX;
    static constexpr int WS_ZX = WS_LUU + NU * NU, WS_ZP = WS_ZX + NU * NX;
    static constexpr int WS_YV = WS_ZP + NU * NX;     // [NX] assemble: rhs part of x_k ; chain: y_k
    static constexpr int WS_GN = WS_YV + NX;          // [NX] rhs contribution to x_{k+1}
    static constexpr int WS_YU = WS_GN + NX;          // [NU]
    static constexpr int WS_Y2 = WS_YU + NU;          // [1]  |y_u|^2 of the stage
    // border of a free dt (the arrowhead's last column, carried through the chain as a second right-hand side):
    static constexpr int WS_BX = (WS_Y2 + 2) & ~1;    // [NX] border part of x_k (controls eliminated) ; chain: unchanged
    static constexpr int WS_BN = WS_BX + NX;          // [NX] border contribution to x_{k+1}
    static constexpr int WS_ZU = WS_BN + NX;          // [NU] L_uu^{-1} (border's control part)
    static constexpr int WS_SC = WS_ZU + NU;          // [4]  the stage's parts of H(dt,dt), rhs(dt), |z_u|^2, z_u . y_u
    static constexpr int WS_YV2 = WS_SC + 4;          // [NX] chain: W z (the second right-hand side's a_k)
    static constexpr int WS_STAGE = (WS_YV2 + NX + 1) & ~1;
    static_assert(WS_STAGE == big_ws_stage(NX, NU) && TOTAL == big_lds_total(NX, NU), "host-side mirrors of the sizes");
    static_assert(HALF == ((5 * NX + 2 * (NX + NU) + NX * (NX + NU) + 8 + NX * NX + 1) & ~1), "big_stage_cache_doubles mirrors HALF");
};

// LDS pointers of one interval of the stage kernel (half = its per-interval area, shared = the wave's assemble scratch)
template <int NX, int NU>
struct BigCtx {
    using BL = BigLds<NX, NU>;
    static constexpr int S = NX + NU;
    double *Gm, *cd, *rv, *dg, *gd, *cin, *fx, *red, *cm, *dcv, *Mm, *gm, *Luu, *Zx, *Zp, *yu, *hb, *zub;
    __device__ __forceinline__ BigCtx(double* half, double* shared)
        : Gm(half + BL::G), cd(half + BL::CD), rv(half + BL::R), dg(half + BL::DIAG), gd(half + BL::GDIAG), cin(half + BL::CIN), fx(half + BL::FIX),
          red(half + BL::RED), cm(half + BL::CM), dcv(half + BL::DCV), Mm(shared + BL::M), gm(shared + BL::GM), Luu(shared + BL::LUU), Zx(shared + BL::ZX), Zp(shared + BL::ZP),
          yu(shared + BL::YU), hb(shared + BL::HB), zub(shared + BL::ZUB) {}
};

// ---- first factorisation of a solve: mu = tau * max diag(J^T J), stop = |rhs|_inf <= eps1 (:115-118), in two steps:
//      big_diag_stage    (stage kernel, diag pass) per (stage, instance): the stage's parts of diag(J^T J) and of rhs = -J^T r, left in the
//                        (not yet used) factor slot of the stage's workspace:  [0,NX) diag of x_k without the C-part of stage k-1,
//                        [NX,2NX) same for rhs, [2NX,3NX) / [3NX,4NX) the C-parts this stage adds to x_{k+1}, [4NX], [4NX+1] the
//                        maxima over the stage's controls;
//      big_first_kernel  one wave per instance, lanes over the stages: adds the neighbouring parts, takes the maxima.
template <int NX, int NU, bool DENSEC = false, bool ARROW = false>
__device__ __forceinline__ void big_diag_stage(const BigCtx<NX, NU>& c, const int N, const int k, const int lane, double* wk)
{
    if constexpr (ARROW) {   // the stage's parts of H(dt,dt) and rhs(dt) (the dt vertex' own rows sit in stage 0's red[5], red[6])
        if (lane == 63) {
            double cdt = 0.0, gdt = 0.0;
#pragma unroll
            for (int r = 0; r < NX; ++r) { cdt += c.dcv[r] * c.dcv[r]; gdt -= c.dcv[r] * c.rv[r]; }
            wk[4 * NX + 2] = cdt + c.red[5]; wk[4 * NX + 3] = gdt + c.red[6];
        }
    }
    constexpr int S = NX + NU, W = 2 * NX + NU;
    double mu_d = -1e300, mu_g = 0.0;
    if (lane < W) {
        double dd = 0.0, gg = 0.0;
        if (lane < S) {
            for (int r = 0; r < NX; ++r) { const double a = c.Gm[r * S + lane]; dd += a * a; gg -= a * c.rv[r]; }
        }
        else if constexpr (DENSEC) {   // column i of the dense block C (collocation defects)
            for (int r = 0; r < NX; ++r) { const double a = c.cm[r * NX + lane - S]; dd += a * a; gg -= a * c.rv[r]; }
        }
        else { const double a = c.cd[lane - S]; dd = a * a; gg = -(a * c.rv[lane - S]); }   // column i of the diagonal block C
        if (lane < NX) {
            dd += c.dg[lane] + c.cin[lane] * c.cin[lane]; gg += c.gd[lane] - c.cin[lane] * c.red[7];
            wk[lane] = dd; wk[NX + lane] = gg;
        }
        else if (lane < S) {
            if (k < N - 1) { dd += c.dg[lane]; gg += c.gd[lane]; mu_d = dd; mu_g = fabs(gg); }
        }
        else { wk[2 * NX + lane - S] = dd; wk[3 * NX + lane - S] = gg; }
    }
    mu_d = wave_max(mu_d);
    mu_g = wave_max(mu_g);
    if (lane == 0) { wk[4 * NX] = mu_d; wk[4 * NX + 1] = mu_g; }
}

template <int NX, int NU, bool ARROW = false>
__global__ __launch_bounds__(64) void big_first_kernel(const FactorParams p)
{
    using BL = BigLds<NX, NU>;
    constexpr int S = NX + NU;
    static_assert(4 * NX + 4 <= NX * NX, "the diag parts live in the factor slot of the stage");
    const int inst = blockIdx.x + p.inst0, lane = threadIdx.x;
    LmState* st = p.st + inst;
    if (st->done || !st->first) return;
    const int N = p.N;
    const double* ws = p.work + (size_t)inst * p.work_stride + BL::WS_L;
    double mx_d = -1e300, mx_g = 0.0, s_cdt = 0.0, s_gdt = 0.0;
    for (int k = lane; k < N; k += 64) {
        const double* wk = ws + (size_t)k * BL::WS_STAGE;
        const double* wp = wk - BL::WS_STAGE;   // stage k-1 (only read for k > 0)
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double dd = wk[i], gg = wk[NX + i];
            if (k > 0) { dd += wp[2 * NX + i]; gg += wp[3 * NX + i]; }
            if (!p.comp[k * S + i].fixed) { mx_d = fmax(mx_d, dd); mx_g = fmax(mx_g, fabs(gg)); }
        }
        if (k < N - 1) { mx_d = fmax(mx_d, wk[4 * NX]); mx_g = fmax(mx_g, wk[4 * NX + 1]); }
        if constexpr (ARROW) { s_cdt += wk[4 * NX + 2]; s_gdt += wk[4 * NX + 3]; }
    }
    mx_d = wave_max(mx_d);
    mx_g = wave_max(mx_g);
    if constexpr (ARROW) { mx_d = fmax(mx_d, wave_sum(s_cdt)); mx_g = fmax(mx_g, fabs(wave_sum(s_gdt))); }   // (:115-118 incl. the dt vertex)
    if (lane == 0) {
        double mu = LM_TAU * mx_d;
        if (mu < 0) mu = 0;
        st->mu   = mu;
        st->stop = (mx_g <= LM_EPS1) ? 1 : 0;
    }
}

// ---- per (stage, instance): everything of the factorisation that does not depend on the neighbouring stages.  The local Jacobian
//      G = [A | B | C], the defect residual and the single-entry rows of the stage's components are in the LDS context c (written by
//      the stage kernel straight from the finite differences: the Jacobian of this family never exists in HBM).
template <int NX, int NU, bool USE_MFMA, bool DENSEC = false, bool ARROW = false>
__device__ __forceinline__ void big_assemble_stage(const BigCtx<NX, NU>& c, const int N, const int k, const int lane, double* wk, const double mu_eff)
{
    using BL = BigLds<NX, NU>;
    constexpr int S = NX + NU;
    const double* cm = c.cm;   // (DENSEC) the x_{k+1} block, [NX][NX]
    double *Gm = c.Gm, *cd = c.cd, *rv = c.rv, *Mm = c.Mm, *gm = c.gm, *Luu = c.Luu, *Zx = c.Zx, *Zp = c.Zp, *yu = c.yu, *dg = c.dg, *gd = c.gd,
           *cin = c.cin, *red = c.red;
    const bool stage = (k < N - 1);
    double y2 = 0.0;
    if (stage) {
        // M = [A B]^T [A B]  (S x NX times NX x S): fp64 matrix cores, one 16 x 16 tile, K = NX in steps of 4.  v_mfma_f64_16x16x4f64
        // layout (probed on gfx950 with tools/mfma_f64_layout.hip): A[i][k] and B[k][j] live in lane l with i|j = l % 16, k = l / 16;
        // the result register r of lane l is D[4 r + l / 16][l % 16].
        if constexpr (USE_MFMA && S <= 16 && NX % 4 == 0) {
            typedef double d4_t __attribute__((ext_vector_type(4)));
            const int lj = lane & 15, lk = lane >> 4;
            d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k0 = 0; k0 < NX; k0 += 4) {
                const double a = (lj < S) ? Gm[(k0 + lk) * S + lj] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * r + lk, j = lj;
                if (i < S && j < S) Mm[i * S + j] = acc[r];
            }
        }
        else {
            for (int e = lane; e < S * S; e += 64) {
                const int i = e / S, j = e % S;
                double v = 0.0;
#pragma unroll
                for (int r = 0; r < NX; ++r) v += Gm[r * S + i] * Gm[r * S + j];
                Mm[i * S + j] = v;
            }
        }
        if (lane < S) {
            double v = 0.0;
#pragma unroll
            for (int r = 0; r < NX; ++r) v -= Gm[r * S + lane] * rv[r];
            gm[lane] = v;
            if constexpr (ARROW) {   // border: [A B]^T d
                double hv = 0.0;
#pragma unroll
                for (int r = 0; r < NX; ++r) hv += Gm[r * S + lane] * c.dcv[r];
                c.hb[lane] = hv;
            }
        }
        __syncthreads();
        // controls: Huu = M[uu] + diag + damping, Cholesky by one lane (NU x NU)
        if (lane == 0) {
            double H[NU][NU];
#pragma unroll
            for (int a = 0; a < NU; ++a)
#pragma unroll
                for (int b = 0; b < NU; ++b) H[a][b] = Mm[(NX + a) * S + NX + b] + ((a == b) ? dg[NX + a] + mu_eff : 0.0);
            chol_inv<NU>(H);
#pragma unroll
            for (int a = 0; a < NU; ++a)
#pragma unroll
                for (int b = 0; b < NU; ++b) Luu[a * NU + b] = H[a][b];
        }
        __syncthreads();
        // Zx = L^{-1} H(u, x_k), Zp = L^{-1} H(u, x_{k+1}) = L^{-1} B^T C (column j of B^T scaled by c_j), yu = L^{-1} gu: one lane per column
        if (lane < 2 * NX + 1 + (ARROW ? 1 : 0)) {
            double col[NU];
#pragma unroll
            for (int a = 0; a < NU; ++a) {
                if constexpr (DENSEC) {   // B^T C, column lane - NX
                    double bc = 0.0;
                    if (lane >= NX && lane < 2 * NX)
                        for (int r = 0; r < NX; ++r) bc += Gm[r * S + NX + a] * cm[r * NX + lane - NX];
                    col[a] = (lane < NX) ? Mm[(NX + a) * S + lane] : (lane < 2 * NX) ? bc : gm[NX + a] + gd[NX + a];
                }
                else
                col[a] = (lane < NX) ? Mm[(NX + a) * S + lane] : (lane < 2 * NX) ? Gm[(lane - NX) * S + NX + a] * cd[lane - NX] : gm[NX + a] + gd[NX + a];
                if constexpr (ARROW) { if (lane == 2 * NX + 1) col[a] = c.hb[NX + a]; }
            }
#pragma unroll
            for (int a = 0; a < NU; ++a) {
                double v = col[a];
#pragma unroll
                for (int b = 0; b < a; ++b) v -= Luu[a * NU + b] * col[b];
                col[a] = v * Luu[a * NU + a];
            }
#pragma unroll
            for (int a = 0; a < NU; ++a) {
                if (lane < NX) Zx[a * NX + lane] = col[a];
                else if (lane < 2 * NX) Zp[a * NX + lane - NX] = col[a];
                else if (lane == 2 * NX) { yu[a] = col[a]; y2 += col[a] * col[a]; }
                else c.zub[a] = col[a];
            }
        }
        __syncthreads();
    }
    // own parts of state block k: M[xx] - Zx^T Zx + diag + c c^T + damping ; the coupling C^T A - Zp^T Zx (row i of A scaled by c_i) ;
    // the contribution to block k+1: C^T C - Zp^T Zp
    for (int e = lane; e < NX * NX; e += 64) {
        const int i = e / NX, j = e % NX;
        double d = cin[i] * cin[j], cx = 0.0, dn = 0.0;   // (the inequality row of the block: keep-out ball, or the TerminalBall on x_f)
        if (stage) {
            d  = Mm[i * S + j] + d;
            if constexpr (DENSEC) {   // C^T A and C^T C
                for (int r = 0; r < NX; ++r) { cx += cm[r * NX + i] * Gm[r * S + j]; dn += cm[r * NX + i] * cm[r * NX + j]; }
            }
            else {
                cx = cd[i] * Gm[i * S + j];
                dn = (i == j) ? cd[i] * cd[i] : 0.0;
            }
#pragma unroll
            for (int a = 0; a < NU; ++a) {
                d -= Zx[a * NX + i] * Zx[a * NX + j];
                cx -= Zp[a * NX + i] * Zx[a * NX + j];
                dn -= Zp[a * NX + i] * Zp[a * NX + j];
            }
        }
        if (i == j) d += dg[i] + mu_eff;
        wk[BL::WS_L + e] = d; wk[BL::WS_Y + e] = cx; wk[BL::WS_DN + e] = dn;
    }
    if (lane < NX) {
        double g = gd[lane], g2 = 0.0;
        if (!stage) g += -(cin[lane] * red[7]);
        if (stage) {
            g += gm[lane] - cin[lane] * red[7];
            if constexpr (DENSEC) { for (int r = 0; r < NX; ++r) g2 -= cm[r * NX + lane] * rv[r]; }
            else g2 = -(cd[lane] * rv[lane]);
#pragma unroll
            for (int a = 0; a < NU; ++a) { g -= Zx[a * NX + lane] * yu[a]; g2 -= Zp[a * NX + lane] * yu[a]; }
        }
        wk[BL::WS_YV + lane] = g; wk[BL::WS_GN + lane] = g2;
        if constexpr (ARROW) {   // the border's parts of x_k and x_{k+1}, controls eliminated
            double bx = 0.0, bn = 0.0;
            if (stage) {
                bx = c.hb[lane];
                if constexpr (DENSEC) { for (int r = 0; r < NX; ++r) bn += cm[r * NX + lane] * c.dcv[r]; }
                else bn = cd[lane] * c.dcv[lane];
#pragma unroll
                for (int a = 0; a < NU; ++a) { bx -= Zx[a * NX + lane] * c.zub[a]; bn -= Zp[a * NX + lane] * c.zub[a]; }
            }
            wk[BL::WS_BX + lane] = bx; wk[BL::WS_BN + lane] = bn;
        }
    }
    if constexpr (ARROW) {
        if (lane == 63) {   // the stage's parts of H(dt,dt), rhs(dt) (the dt vertex' own rows: red[5], red[6] of stage 0), |z_u|^2, z_u . y_u
            double cdt = red[5], gdt = red[6], zzu = 0.0, zyu = 0.0;
            if (stage) {
#pragma unroll
                for (int r = 0; r < NX; ++r) { cdt += c.dcv[r] * c.dcv[r]; gdt -= c.dcv[r] * rv[r]; }
#pragma unroll
                for (int a = 0; a < NU; ++a) { zzu += c.zub[a] * c.zub[a]; zyu += c.zub[a] * yu[a]; }
            }
            wk[BL::WS_SC] = cdt; wk[BL::WS_SC + 1] = gdt; wk[BL::WS_SC + 2] = zzu; wk[BL::WS_SC + 3] = zyu;
        }
        if (lane < NU) wk[BL::WS_ZU + lane] = stage ? c.zub[lane] : 0.0;
    }
    if (stage) {
        if (lane < NU * NU) wk[BL::WS_LUU + lane] = Luu[lane];
        if (lane < NU * NX) { wk[BL::WS_ZX + lane] = Zx[lane]; wk[BL::WS_ZP + lane] = Zp[lane]; }
        if (lane < NU) wk[BL::WS_YU + lane] = yu[lane];
    }
    if (lane == 2 * NX) wk[BL::WS_Y2] = y2;   // (the lane that formed y_u; 0 for the last block)
    __syncthreads();   // the shared scratch is free for the wave's second interval
}

// ---- the stage kernel of the big-block family: ONE wave per PAIR of shooting intervals (k, k+1) of one instance.
//      Phase 1 (edges): the finite-difference Jacobian of the two defect edges, one lane per (interval, column, side) -- 2 x 16 x 2 =
//      64 Runge-Kutta integrations side by side, each with its own perturbed copy of (x_k, u_k), exactly BaseEdge::computeJacobian
//      (edge_interface.cpp:55-96: x_i += delta -> v2, x_i += -2 delta -> v1, (1 / (2 delta)) (v2 - v1)); the two sides of a column meet
//      through a lane swap.  The x_{k+1} columns need no integration: e = RK4(x_k, u_k) - x_{k+1} with the end state of the UNPERTURBED
//      step, which the residual sweep has kept (SweepParams::xe0).  Cost rows, bound rows and the stage inequality of the intervals'
//      components are evaluated by the first lanes of each half (same formulas, same operation order as sweep_body: bit-identical).
//      Phase 2 (assemble): big_diag_stage (first factorisation of a solve: diag(J^T J), rhs) or big_assemble_stage, interval by
//      interval on the whole wave.  The Jacobian lives in LDS for the length of this kernel and nowhere else; a rejected step is
//      re-assembled from the (unchanged) accepted iterate with the larger damping -- same bits, no Jacobian traffic at all.
//      jac_dump (parity hook, corbo_hip_eval): the Jacobian values this kernel works with, written in the public value order.
#pragma clang fp contract(off)
// USERINEQ: the stage inequality is a user state function (csrc/stage_functions/) -- an instantiation of its own, so that the keep-out ball's kernels (cfg 5)
// are what they were: its three-component copies, no branch (an out-of-line helper behind a uniform branch cost cfg 5 8.05 -> 8.41 ms per solve: 192 bytes of scratch per lane)
template <int DYN, int DEFECT = CORBO_HIP_DEFECT_RK4_SHOOTING, bool ARROW = false, bool USERINEQ = false>
__device__ __forceinline__ void big_stage_edges(const FactorParams& p, const SweepParams& sp, const BigCtx<Dynamics<DYN>::NX, Dynamics<DYN>::NU>& c,
                                                const int k, const int l32, const int inst, const int vsel, double* jac_dump)
{
    using Dy = Dynamics<DYN>;
    constexpr int NX = Dy::NX, NU = Dy::NU, S = NX + NU, NC = Dy::NC;
    constexpr bool SHOOT = (DEFECT == CORBO_HIP_DEFECT_RK4_SHOOTING || DEFECT == DEFECT_SHOOTING_HIGH);
    constexpr double delta = 1e-9, neg2delta = -2 * delta, scalar = 1.0 / (2 * delta);
    const int N      = p.N;
    const bool stage = (k < N - 1), block = (k < N);
    const int kk     = stage ? k : 0;   // (clamped: absent intervals load interval 0 and write zeros)
    const size_t xo  = (size_t)inst * sp.nvs;
    const double* X  = sp.x + xo;
    const double dt0 = X[sp.off_dt];
    const int* sc    = p.stage_cols[kk].col;
    if constexpr (!SHOOT) {
        // ---- collocation defects (FDCollocationEdge, finite_differences_collocation_edges.h:43-80): every one of the 2 nx + nu columns
        //      (x_k | u_k | x_{k+1}) is a pair of evaluations of the defect formula (the x_{k+1} block is dense: LDS area cm), 16 columns
        //      per round of the interval's 32 lanes; the round behind the last column evaluates the unperturbed defect (the residual the
        //      assembly needs: same inputs and operations as the sweep's, same bits)
        constexpr int W = 2 * NX + NU;
        double base[W];
#pragma unroll
        for (int i = 0; i < W; ++i) base[i] = X[kk * S + i];
        CORBO_HIP_DYN_OF(dynl, sp, inst)
        const bool minus = (l32 & 1) != 0;
#pragma unroll 1
        for (int c0 = 0; c0 <= W + (ARROW ? 1 : 0); c0 += 16) {   // (free dt: column W + 1 is the dt column, every q of the formula changes)
            const int col = c0 + (l32 >> 1);
            double loc[W], e[NX];
            double pert = 0.0;
#pragma unroll
            for (int i = 0; i < W; ++i) { loc[i] = base[i]; pert = (i == col) ? base[i] : pert; }
            pert += delta;
            if (minus) pert += neg2delta;
#pragma unroll
            for (int i = 0; i < W; ++i) loc[i] = (i == col) ? pert : loc[i];
            double dtl = dt0;
            if constexpr (ARROW) {
                double da = dt0 + delta;
                if (minus) da += neg2delta;
                dtl = (col == W + 1) ? da : dt0;
            }
            defect_eval<DYN, DEFECT>(loc, loc + NX, loc + S, dtl, dynl, e);
            int jo = -1;
#pragma unroll
            for (int i = 0; i < W; ++i) jo = (i == col) ? sc[i] : jo;
            if constexpr (ARROW) jo = (col == W + 1) ? sc[W] : jo;
            const bool present = stage && col < W && jo >= 0;
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                const double eo = __shfl_xor(e[r], 1);   // the other side of the same column
                const double cv = (scalar * (e[r] - eo)) * sp.w_eq;
                if (!minus && col < W) {
                    if (col < S) c.Gm[r * S + col] = present ? cv : 0.0;
                    else c.cm[r * NX + col - S] = present ? cv : 0.0;
                    if (jac_dump && present) jac_dump[jo + r] = cv;
                }
                if (col == W && !minus) c.rv[r] = stage ? e[r] * sp.w_eq : 0.0;
                if constexpr (ARROW) {
                    if (col == W + 1 && !minus) c.dcv[r] = (stage && jo >= 0) ? cv : 0.0;
                    e[r] = cv;
                }
            }
            if constexpr (ARROW) {   // (the parity hook's copy of the dt column in a block of its own: see the shooting branch)
                asm volatile("" ::: "memory");
                if (jac_dump && col == W + 1 && !minus && stage && jo >= 0) {
#pragma unroll
                    for (int r = 0; r < NX; ++r) jac_dump[jo + r] = e[r];
                }
            }
        }
    }
    // ---- (x_k, u_k) columns: lane = (column, side)
    if constexpr (SHOOT) {
        double loc[S], ck[4][NC], xe[NX];
#pragma unroll
        for (int i = 0; i < S; ++i) loc[i] = X[kk * S + i];
        const int col    = l32 >> 1;
        const bool minus = (l32 & 1) != 0;
        double pert = 0.0;
#pragma unroll
        for (int i = 0; i < S; ++i) pert = (i == col) ? loc[i] : pert;
        pert += delta;
        if (minus) pert += neg2delta;
#pragma unroll
        for (int i = 0; i < S; ++i) loc[i] = (i == col) ? pert : loc[i];
        CORBO_HIP_DYN_OF(dynl, sp, inst)
        if constexpr (DEFECT == DEFECT_SHOOTING_HIGH) rk_high_order_end_state<DYN>(loc, loc + NX, dt0, dynl, (int)dynl[7], xe);   // Runge-Kutta 5 / 6 / 7
        else
        rk4_end_state<DYN, false>(loc, loc + NX, dt0, dynl, ck, xe);
        int jo = 0;
#pragma unroll
        for (int i = 0; i < S; ++i) jo = (i == col) ? sc[i] : jo;
        const bool present = stage && col < S && jo >= 0;   // (models with fewer than 16 columns: the lanes beyond them integrate the unperturbed step and write nothing)
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            const double ev = xe[r] - X[kk * S + S + r];   // (x_{k+1} is read here, not kept in registers across the integration)
            const double eo = __shfl_xor(ev, 1);   // the other side of the same column
            const double cv = (scalar * (ev - eo)) * sp.w_eq;   // (plus lanes: v2 - v1; hyper_graph_optimization_problem_edge_based.cpp:1552)
            if (!minus && col < S) {
                c.Gm[r * S + col] = present ? cv : 0.0;
                if (jac_dump && present) jac_dump[jo + r] = cv;
            }
        }
        if constexpr (ARROW) {   // free dt: one more pair of integrations (lanes 0 / 1 of the interval matter), the step length perturbed
#pragma unroll
            for (int i = 0; i < S; ++i) loc[i] = X[kk * S + i];
            double da = dt0 + delta;
            if (minus) da += neg2delta;
            if constexpr (DEFECT == DEFECT_SHOOTING_HIGH) rk_high_order_end_state<DYN>(loc, loc + NX, da, dynl, (int)dynl[7], xe);
            else
            rk4_end_state<DYN, false>(loc, loc + NX, da, dynl, ck, xe);
            const int jd    = sc[S + NX];
            const bool pdt  = stage && jd >= 0;
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                const double ev = xe[r] - X[kk * S + S + r];
                const double eo = __shfl_xor(ev, 1);
                const double cv = (scalar * (ev - eo)) * sp.w_eq;
                xe[r] = cv;
                if (l32 == 0) c.dcv[r] = pdt ? cv : 0.0;
            }
            // (the parity hook's copy in a block of its own: with the LDS and the global store under one condition the 6-state unit's compile ends in
            //  "Illegal instruction detected: Operand has incorrect register class" -- a flat store through a select of the two address spaces)
            asm volatile("" ::: "memory");
            if (jac_dump && pdt && l32 == 0) {
#pragma unroll
                for (int r = 0; r < NX; ++r) jac_dump[jd + r] = xe[r];
            }
        }
    }
    // ---- x_{k+1} columns (diagonal: only e_i depends on x_{k+1,i}), defect residual, stage inequality: lane i < NX
    if (l32 < NX) {
        const int i      = l32;
        if constexpr (SHOOT) {
        const double xei = sp.xe0[(((size_t)vsel * sp.batch_total + inst) * sp.N + kk) * NX + i];
        const double x2i = X[kk * S + S + i];
        const double a = x2i + delta, b = a + neg2delta;
        const double cd = (scalar * ((xei - a) - (xei - b))) * sp.w_eq;
        const int jo    = sc[S + i];
        const bool present = stage && jo >= 0;
        c.cd[i] = present ? cd : 0.0;
        if (jac_dump && present) {
#pragma unroll
            for (int r = 0; r < NX; ++r) jac_dump[jo + r] = (r == i) ? cd : 0.0;   // rows r != i: scalar * (e_r - e_r) = 0
        }
        c.rv[i] = stage ? (xei - x2i) * sp.w_eq : 0.0;
        }
        double cinv = 0.0, rin = 0.0;
        if (stage && p.ineq_cols) {   // computeValuesActiveInequality + its active-row Jacobian (sweep_body (b), (3))
            double c0, c2, c1;
            if constexpr (USERINEQ) {   // a user state function: all NX components in private copies, perturbed like BaseEdge::computeJacobian perturbs them
                double q[NX];
#pragma unroll
                for (int t = 0; t < NX; ++t) q[t] = X[kk * S + t];
                c0 = stage_ineq_state<NX>(sp.mp.ineq_id, q, sp.mp.ineq);
                const double up = X[kk * S + i] + delta;
#pragma unroll
                for (int t = 0; t < NX; ++t) q[t] = (t == i) ? up : X[kk * S + t];
                c2 = stage_ineq_state<NX>(sp.mp.ineq_id, q, sp.mp.ineq);
#pragma unroll
                for (int t = 0; t < NX; ++t) q[t] = (t == i) ? up + neg2delta : X[kk * S + t];
                c1 = stage_ineq_state<NX>(sp.mp.ineq_id, q, sp.mp.ineq);
            }
            else {   // the keep-out ball reads three components: three registers per copy (cfg 5's kernel)
                double q[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) q[t] = X[kk * S + t];
                c0 = ineq_ball(q, sp.mp.ineq);
                double q2[3], q1[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) { const double up = q[t] + delta; q2[t] = (t == i) ? up : q[t]; q1[t] = (t == i) ? up + neg2delta : q[t]; }
                c2 = ineq_ball(q2, sp.mp.ineq); c1 = ineq_ball(q1, sp.mp.ineq);
            }
            rin             = (c0 < 0) ? 0.0 : c0 * sp.w_ineq;
            const bool active = rin > 0.0;
            const int jq = p.ineq_cols[kk * NX + i];
            cinv = (jq >= 0 && active) ? (scalar * (c2 - c1)) * sp.w_ineq : 0.0;
            if (jac_dump && jq >= 0) jac_dump[jq] = active ? (scalar * (c2 - c1)) * sp.w_ineq : 0.0;
        }
        if (!stage && block && sp.fin_row >= 0) {   // final block: TerminalBall on x_f (sweep_body (c) + its active-row Jacobian), same slot
            double loc[NX], xrf[NX];
#pragma unroll
            for (int t = 0; t < NX; ++t) {
                loc[t] = X[(size_t)(N - 1) * S + t];
                xrf[t] = sp.refvec ? sp.refvec[xo + (size_t)(N - 1) * S + t] : sp.xref[(size_t)inst * CORBO_HIP_MAX_NX + t];
            }
            const double c0 = terminal_ball<NX>(loc, xrf, sp.mp.fin);
            rin             = (c0 < 0) ? 0.0 : c0 * sp.w_ineq;
            const bool active = rin > 0.0;
            double keep = 0.0;
#pragma unroll
            for (int t = 0; t < NX; ++t) keep = (t == i) ? loc[t] : keep;
            const double up = keep + delta, dn = up + neg2delta;
#pragma unroll
            for (int t = 0; t < NX; ++t) loc[t] = (t == i) ? up : loc[t];
            const double c2 = terminal_ball<NX>(loc, xrf, sp.mp.fin);
#pragma unroll
            for (int t = 0; t < NX; ++t) loc[t] = (t == i) ? dn : loc[t];
            const double c1 = terminal_ball<NX>(loc, xrf, sp.mp.fin);
            const int jq = p.fin_joff[i];
            cinv = (jq >= 0 && active) ? (scalar * (c2 - c1)) * sp.w_ineq : 0.0;
            if (jac_dump && jq >= 0) jac_dump[jq] = active ? (scalar * (c2 - c1)) * sp.w_ineq : 0.0;
        }
        c.cin[i] = cinv;
        if (i == 0) c.red[7] = rin;
    }
    // ---- single-entry rows of the interval's components (cost row, bound row): lane e < S  (sweep_body comp_values / comp_jac)
    if (l32 < S) {
        const int e    = l32;
        const bool isx = e < NX;
        double dd = 0.0, gg = 0.0;
        if (block && (isx || stage)) {
            const int v       = k * S + e;
            const CompInfo ci = p.comp[v];
            if (isx) c.fx[e] = ci.fixed ? 1.0 : 0.0;
            const double xv = X[v];
            const bool fin  = (k == N - 1);
            double w = 0.0, ref = 0.0;
#pragma unroll
            for (int t = 0; t < NX; ++t)
                if (isx && e == t) { w = fin ? sp.mp.sqf[t] : sp.mp.sq[t]; ref = sp.xref[(size_t)inst * CORBO_HIP_MAX_NX + t]; }
#pragma unroll
            for (int t = 0; t < NU; ++t)
                if (!isx && e - NX == t) w = sp.mp.sr[t];
            if (sp.refvec && isx) ref = sp.refvec[xo + v];   // time-varying state references
            if (!ci.fixed && ci.cost_joff >= 0) {
                const double a = xv + delta, b = a + neg2delta;
                const double dv  = scalar * (w * (a - ref) - w * (b - ref));
                const double val = w * (xv - ref);
                dd += dv * dv;
                gg -= dv * val;
                if (jac_dump) {   // the cost block column incl. its explicit zeros
                    const int cdim = isx ? NX : NU, cc = isx ? e : e - NX;
                    for (int r = 0; r < cdim; ++r) jac_dump[ci.cost_joff - cc + r] = (r == cc) ? dv : 0.0;
                }
            }
            if (fin && isx && !ci.fixed && ci.cost2_joff >= 0) {   // Terminal[Partial]EqualityConstraint x_f - xref: a second diagonal row, times w_eq
                const double a = xv + delta, b = a + neg2delta;
                const double dv  = (scalar * ((a - ref) - (b - ref))) * sp.w_eq;
                const double val = (xv - ref) * sp.w_eq;
                // (TerminalPartialEqualityConstraint, final_state_constraints.h:219-252: a row for the ACTIVE components only; an inactive component has no
                //  row -- cost2_row < 0.  The Jacobian dump below writes the FULL constraint's nx x nx block: corbo_hip_eval takes the Jacobian of a handle
                //  with a partial mask from the sweep kernel instead, corbo_hip.hip)
                if (ci.cost2_row >= 0) { dd += dv * dv; gg -= dv * val; }
                if (jac_dump)
                    for (int r = 0; r < NX; ++r) jac_dump[ci.cost2_joff - e + r] = (r == e) ? dv : 0.0;
            }
            if (ci.bnd_joff >= 0) {
                const double l = sp.lb[xo + v], u = sp.ub[xo + v];
                const double ab = (xv < l) ? -sp.w_b : ((xv > u) ? sp.w_b : 0.0);
                double vb = (xv < l) ? l - xv : ((xv > u) ? xv - u : 0.0);
                vb *= sp.w_b;
                dd += ab * ab;
                gg -= ab * vb;
                if (jac_dump) jac_dump[ci.bnd_joff] = ab;
            }
        }
        c.dg[e] = dd;
        c.gd[e] = gg;
    }
    if constexpr (ARROW) {   // the dt vertex' own rows (cost row, the duplicated MinimumTime row, bound row): once per instance, with stage 0
        if (l32 == S) {
            double dd = 0.0, gg = 0.0;
            if (k == 0) {
                const CompInfo ci = p.comp[sp.off_dt];
                const double xv = dt0, w = sp.mp.dt_weight, ref = 0.0;
                if (!ci.fixed && ci.cost_joff >= 0) {
                    const double a = xv + delta, b = a + neg2delta;
                    const double dv  = scalar * (w * (a - ref) - w * (b - ref));
                    const double val = w * (xv - ref);
                    dd += dv * dv; gg -= dv * val;
                    if (jac_dump) jac_dump[ci.cost_joff] = dv;
                    if (ci.cost2_joff >= 0) { dd += dv * dv; gg -= dv * val; if (jac_dump) jac_dump[ci.cost2_joff] = dv; }
                }
                if (ci.bnd_joff >= 0) {
                    const double l = sp.lb[xo + sp.off_dt], u = sp.ub[xo + sp.off_dt];
                    const double ab = (xv < l) ? -sp.w_b : ((xv > u) ? sp.w_b : 0.0);
                    double vb = (xv < l) ? l - xv : ((xv > u) ? xv - u : 0.0);
                    vb *= sp.w_b;
                    dd += ab * ab; gg -= ab * vb;
                    if (jac_dump) jac_dump[ci.bnd_joff] = ab;
                }
            }
            c.red[5] = dd; c.red[6] = gg;
        }
    }
}
#pragma clang fp contract(fast)

template <int DYN, bool USE_MFMA, int DEFECT = CORBO_HIP_DEFECT_RK4_SHOOTING, bool ARROW = false, bool USERINEQ = false>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(3, 3)))   // 168 registers: three waves per SIMD (170 without the cap, i.e. two; a cap of four spills 270 bytes and loses)
void big_stage_kernel(const FactorParams p, const SweepParams sp, const int diag_only, double* jac_dump)
{
    constexpr bool DENSEC = (DEFECT != CORBO_HIP_DEFECT_RK4_SHOOTING && DEFECT != DEFECT_SHOOTING_HIGH);   // collocation: the x_{k+1} block of the local Jacobian is dense
    using Dy = Dynamics<DYN>;
    constexpr int NX = Dy::NX, NU = Dy::NU;
    using BL = BigLds<NX, NU>;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int pair = blockIdx.x, inst = blockIdx.y + p.inst0, lane = threadIdx.x;
    const LmState* st = p.st + inst;
    int vsel = 0;
    double mu_eff = 0.0;
    if (!jac_dump) {
        if (st->done) return;
        if (diag_only && !st->first) return;
        vsel   = st->vbuf;
        mu_eff = (st->fresh ? 0.0 : st->mu_acc) + st->mu;   // H_ii += mu on every inner pass, never undone (:135-138)
    }
    const BigCtx<NX, NU> c0(sm, sm + 2 * BL::HALF), c1(sm + BL::HALF, sm + 2 * BL::HALF);
    const int half = lane >> 5;
    // The first factorisation of a solve needs every stage's diag(J^T J) before any block can be damped (mu = tau max diag, :117): two passes of this
    // kernel.  The diag pass leaves the wave's LDS context -- the finite-difference Jacobians of its two intervals, 6.8 KB -- in HBM and the assembly
    // pass reads it back instead of integrating the 64 perturbed Runge-Kutta steps a second time (same numbers: bit-identical).
    double2* const cache = (p.stage_cache && !jac_dump && p.first_pass)
                               ? reinterpret_cast<double2*>(p.stage_cache + (size_t)inst * p.stage_cache_stride + (size_t)pair * 2 * BL::HALF) : nullptr;
    const bool cached = cache && !diag_only && st->first;
    if (cached) {
        for (int i = lane; i < BL::HALF; i += 64) reinterpret_cast<double2*>(sm)[i] = cache[i];
    }
    else
    big_stage_edges<DYN, DEFECT, ARROW, USERINEQ>(p, sp, half ? c1 : c0, 2 * pair + half, lane & 31, inst, vsel, jac_dump ? jac_dump + (size_t)inst * sp.nnz_pad : nullptr);
    __syncthreads();
    if (jac_dump) return;
    if (cache && diag_only)
        for (int i = lane; i < BL::HALF; i += 64) cache[i] = reinterpret_cast<const double2*>(sm)[i];
    for (int h = 0; h < 2; ++h) {
        const int k = 2 * pair + h;
        if (k >= p.N) break;
        double* wk = p.work + (size_t)inst * p.work_stride + (size_t)k * BL::WS_STAGE;
        if (diag_only) big_diag_stage<NX, NU, DENSEC, ARROW>(h ? c1 : c0, p.N, k, lane, wk + BL::WS_L);
        else big_assemble_stage<NX, NU, USE_MFMA, DENSEC, ARROW>(h ? c1 : c0, p.N, k, lane, wk, mu_eff);
    }
}

// value of lane `src` (uniform / compile-time index) for every lane: two v_readlane_b32, the result lives in scalar registers
__device__ __forceinline__ double lane_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// ---- per instance: the sequential part, from both ends ("twisted" block Cholesky).  Two waves per instance: wave 0 eliminates the
//      state blocks 0, 1, ... m-1 upwards (block k against k+1), wave 1 the blocks N-1, N-2, ... m+1 downwards (block k against
//      k-1, the mirrored step: coupling transposed), they meet at block m = N/2, which wave 0 factors with both Schur
//      complements; the back-substitution then runs outwards on both sides at once.  Half the sequential depth of a one-sided sweep.
//      Inside a wave lane i < NX owns ROW i of the state block in registers: the 12 x 12 Cholesky, the triangular solves and the
//      back-substitution run on register rows with v_readlane broadcasts -- no LDS round trip and no barrier per pivot; only the
//      Schur complement Y Y^T goes through LDS (operands of the matrix-core instruction).  The assembled data of the next block
//      (elimination) / the factors of the next block (back-substitution) are in flight while the current one is worked on.
template <int NX, int NU, bool USE_MFMA>
__global__ __launch_bounds__(128) void big_chain_kernel(const FactorParams p)
{
    using BL = BigLds<NX, NU>;
    constexpr int S = NX + NU;
    static_assert(NX <= 16 && NU <= NX, "row-per-lane mapping of the chain kernel");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int side = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* Dn = sm + side * 2 * NX * NX;        // Schur mailbox of this wave  [NX][NX]
    double* Cx = Dn + NX * NX;                   // Y of the current block       [NX][NX]
    double* xch = sm + 4 * NX * NX;              // exchange between the waves: [0,NX) rhs mailbox of wave 1, [NX,2NX) x_m, [2NX..] sums
    const int inst = blockIdx.x + p.inst0;
    const int row = (lane < NX) ? lane : NX - 1;   // lanes >= NX mirror the last row (their results are never stored)
    const bool own = (lane < NX);
    LmState* st = p.st + inst;
    const int done = st->done;
    int stop = st->stop;
    const double mu = st->mu;
    const double mu_eff = (st->fresh ? 0.0 : st->mu_acc) + mu;
    if (done) return;
    const int N = p.N;
    const int m = N / 2;                                   // meeting block
    const int nsteps = (m > N - 1 - m) ? m : N - 1 - m;    // both waves run the same number of (possibly idle) steps: barriers
    const double* xin = p.x + (size_t)inst * p.nvs;
    double* xt        = p.xt + (size_t)inst * p.nvs;
    double* ws        = p.work + (size_t)inst * p.work_stride;

    // diagnostics (p.timeline): cycles of wave 0 per part of a step, summed over the steps
    const bool tl_on = p.timeline && blockIdx.x == 0 && threadIdx.x == 0;
    long long tl_acc[6] = {0, 0, 0, 0, 0, 0}, tl_t = 0;
#define CHAIN_STAMP(slot) do { if (p.timeline) { const long long now_ = clock64(); tl_acc[slot] += now_ - tl_t; tl_t = now_; } } while (0)
    if (p.timeline) tl_t = clock64();
    double y2 = 0.0;
    for (int e = lane; e < NX * NX; e += 64) Dn[e] = 0.0;
    double gn_r = 0.0;   // Schur mailbox of the right-hand side (row of this lane)
    // ---- elimination.  Block k of this wave's sequence: wave 0: k = s, wave 1: k = N-1-s.
    //      pd: own parts of D_k (row), pc: coupling towards the block eliminated against (wave 0: row of C_k = H(k+1,k);
    //      wave 1: row of C_{k-1}^T = H(k-1,k)), pn/pgn: the assemble-time contribution DN/GN that belongs to the NEXT block of wave
    //      0's sequence (k+1) resp. to THIS block of wave 1's sequence (DN_{k-1} adds to D_k).
    double pd[NX], pc[NX], pn[NX], pg = 0.0, pgn = 0.0, py2 = 0.0;
    int pfix = 0;
    auto fetch = [&](int k) {
        const double* wk = ws + (size_t)k * BL::WS_STAGE;
        const double* wc = (side == 0) ? wk : wk - BL::WS_STAGE;   // stage whose C / DN / GN this block uses (wave 1: k-1 >= m)
#pragma unroll
        for (int cc = 0; cc < NX; ++cc) {
            pd[cc] = wk[BL::WS_L + row * NX + cc];
            pc[cc] = (side == 0) ? wc[BL::WS_Y + row * NX + cc] : wc[BL::WS_Y + cc * NX + row];
            pn[cc] = wc[BL::WS_DN + row * NX + cc];
        }
        pg  = wk[BL::WS_YV + row];
        pgn = wc[BL::WS_GN + row];
        py2 = wk[BL::WS_Y2];
        pfix = p.comp[k * S + row].fixed;
    };
    auto block_of = [&](int s) { return (side == 0) ? s : N - 1 - s; };
    const int mysteps = (side == 0) ? m : N - 1 - m;
    fetch(block_of(0));   // (mysteps >= 1 for every horizon this kernel is launched with; block 0 / N-1 always exist)
    lds_barrier();
    // one elimination step on register rows; returns y_k[row]; leaves Y in yr, L in d
    auto factor_rows = [&](double (&d)[NX], double (&cr)[NX], double g, double (&yr)[NX]) -> double {
        // Cholesky, right-looking over register rows: after step j, d[j] of lane i > j is L[i][j], of lane j it is 1 / L[j][j]
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const double inv = rsqrt(lane_bcast(d[j], j));
            d[j] = (lane == j) ? inv : d[j] * inv;
#pragma unroll
            for (int cc = j + 1; cc < NX; ++cc) d[cc] -= d[j] * lane_bcast(d[j], cc);   // (rows above cc hold unused upper entries)
        }
        // Y = C L^{-T}: row i of Y from row i of C; y = L^{-1} g across the lanes
#pragma unroll
        for (int cc = 0; cc < NX; ++cc) {
            double v = cr[cc];
#pragma unroll
            for (int t = 0; t < cc; ++t) v -= yr[t] * lane_bcast(d[t], cc);
            yr[cc] = v * lane_bcast(d[cc], cc);
        }
        double yk = g;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const double yj = lane_bcast(yk, j) * lane_bcast(d[j], j);   // y_j = (g_j - sum_{t<j} L[j][t] y_t) / L[j][j]
            yk = (lane == j) ? yj : ((lane > j) ? yk - d[j] * yj : yk);
        }
        return yk;
    };
    for (int s = 0; s < nsteps; ++s) {
        const bool active = (s < mysteps);
        const int k = active ? block_of(s) : 0;
        double d[NX], cr[NX], yr[NX], g = 0.0, yk = 0.0;
        if (active) {
            // state block k = own parts + Schur mailbox (+ DN_{k-1} for wave 1); fixed components: identity rows / columns, zero rhs
            const unsigned long long fmask = __ballot(pfix != 0 && own);
            const bool fixed_r = (fmask >> row) & 1ull;
#pragma unroll
            for (int cc = 0; cc < NX; ++cc) {
                double v = pd[cc] + Dn[row * NX + cc];
                if (side == 1) v += pn[cc];
                if (fixed_r || ((fmask >> cc) & 1ull)) v = (row == cc) ? 1.0 : 0.0;
                d[cc]  = v;
                cr[cc] = pc[cc];
            }
            g = pg + gn_r;
            if (side == 1) g += pgn;
            if (fixed_r) g = 0.0;
            y2 += (lane == 0) ? py2 : 0.0;
        }
        const double gnk = pgn;
        double dnk[NX];
#pragma unroll
        for (int cc = 0; cc < NX; ++cc) dnk[cc] = pn[cc];
        CHAIN_STAMP(0);   // waited for the block's data, combined it
        lds_barrier();   // every lane has taken its mailbox row
        fetch(block_of((s + 1 < mysteps) ? s + 1 : ((mysteps > 0) ? mysteps - 1 : 0)));   // unconditional (see the back-substitution)
        if (active) {
            if (own) {
#pragma unroll
                for (int cc = 0; cc < NX; ++cc) Dn[row * NX + cc] = (side == 0) ? dnk[cc] : 0.0;
            }
            yk = factor_rows(d, cr, g, yr);
            CHAIN_STAMP(1);   // Cholesky + triangular solves on register rows
            if (own) y2 += yk * yk;
            double* wk = ws + (size_t)k * BL::WS_STAGE;
            if (own) {
#pragma unroll
                for (int cc = 0; cc < NX; ++cc) {
                    Cx[row * NX + cc] = yr[cc];
                    wk[BL::WS_L + row * NX + cc] = d[cc];
                    wk[BL::WS_Y + row * NX + cc] = yr[cc];
                }
                wk[BL::WS_YV + row] = yk;
            }
            double v = (side == 0) ? gnk : 0.0;
#pragma unroll
            for (int t = 0; t < NX; ++t) v -= yr[t] * lane_bcast(yk, t);
            gn_r = v;
        }
        CHAIN_STAMP(2);   // stores + right-hand-side update
        lds_barrier();
        if (active) {   // Schur complement to the next block of the sequence: mailbox -= Y Y^T (matrix cores, operands through LDS)
            if constexpr (USE_MFMA && NX <= 16 && NX % 4 == 0) {
                typedef double d4_t __attribute__((ext_vector_type(4)));
                const int lj = lane & 15, lk = lane >> 4;
                d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k0 = 0; k0 < NX; k0 += 4) {   // (Y Y^T)[i][j] = sum_t Y[i][t] Y[j][t]: A[i][t] = Y[i][t], B[t][j] = Y[j][t]
                    const double a = (lj < NX) ? Cx[lj * NX + k0 + lk] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 4 * r + lk, j = lj;
                    if (i < NX && j < NX) Dn[i * NX + j] -= acc[r];
                }
            }
            else {
                for (int e = lane; e < NX * NX; e += 64) {
                    const int i = e / NX, j = e % NX;
                    double v2 = Dn[e];
#pragma unroll
                    for (int t = 0; t < NX; ++t) v2 -= Cx[i * NX + t] * Cx[j * NX + t];
                    Dn[e] = v2;
                }
            }
        }
        lds_barrier();
        CHAIN_STAMP(3);   // Schur complement (matrix cores) + barriers
    }
    // ---- the meeting block m: own parts + both mailboxes (wave 0's holds DN_{m-1} - Y Y^T, wave 1's -Y' Y'^T)
    if (side == 1 && own) xch[row] = gn_r;
    lds_barrier();
    double xm_r = 0.0;
    if (side == 0) {
        const double* wk = ws + (size_t)m * BL::WS_STAGE;
        const int fx = p.comp[m * S + row].fixed;
        const unsigned long long fmask = __ballot(fx != 0 && own);
        const bool fixed_r = (fmask >> row) & 1ull;
        double d[NX], cr[NX], yr[NX];
        const double* DnR = sm + 2 * NX * NX;
#pragma unroll
        for (int cc = 0; cc < NX; ++cc) {
            double v = wk[BL::WS_L + row * NX + cc] + Dn[row * NX + cc] + DnR[row * NX + cc];
            if (fixed_r || ((fmask >> cc) & 1ull)) v = (row == cc) ? 1.0 : 0.0;
            d[cc] = v; cr[cc] = 0.0;
        }
        double g = fixed_r ? 0.0 : wk[BL::WS_YV + row] + gn_r + xch[row];
        y2 += (lane == 0) ? wk[BL::WS_Y2] : 0.0;
        double yk = factor_rows(d, cr, g, yr);
        if (own) y2 += yk * yk;
        // x_m = L^{-T} y across the lanes (row i of the register block holds L[i][.]; column access through broadcasts)
        double v = yk, xk = 0.0;
#pragma unroll
        for (int j = NX - 1; j >= 0; --j) {
            const double xj = lane_bcast(v, j) * lane_bcast(d[j], j);
            if (lane == j) xk = xj;
            // lane i < j subtracts L[j][i] x_j: L[j][i] lives in lane j's register d[i] -> broadcast it
#pragma unroll
            for (int i = 0; i < NX; ++i)
                if (i < j) { const double lji = lane_bcast(d[i], j); if (lane == i) v -= lji * xj; }
        }
        if (fixed_r || !own) xk = 0.0;
        xm_r = xk;
        if (own) {
            double* wkm = ws + (size_t)m * BL::WS_STAGE;
#pragma unroll
            for (int cc = 0; cc < NX; ++cc) wkm[BL::WS_L + row * NX + cc] = d[cc];
            xch[NX + row] = xk;
            xt[m * S + row] = xin[m * S + row] + xk;
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- back-substitution outwards: wave 0 blocks m-1 .. 0 (x_k from x_{k+1}), wave 1 blocks m+1 .. N-1 (x_k from x_{k-1}).
    //      Lane i owns COLUMN i of L_k and Y_k; the neighbour's solution stays in the lanes' registers.  Controls of stage q need
    //      x_q and x_{q+1}: wave 0 does stage k with block k, wave 1 stage k-1 with block k.
    double dn2 = (side == 0) ? xm_r * xm_r : 0.0, xn_r = xch[NX + row];
    if (!own) xn_r = 0.0;
    // The factor data of a block is requested TWO steps ahead (two register buffers, the loop is unrolled by two): one step of
    // arithmetic (~1.5k cycles) does not cover an HBM round trip, and with one step of lookahead the wave sat in s_waitcnt for more
    // than half of this loop.
    struct BackBuf { double bl[NX], by[NX], bzx[NX], bzp[NX], bluu[NU], byv, byu, bx, bxu; int bfix; };
    BackBuf b0{}, b1{};
    const int urow = (lane < NU) ? lane : NU - 1;
    auto blk_of = [&](int s) { return (side == 0) ? m - 1 - s : m + 1 + s; };
    auto fetch_b = [&](BackBuf& bb, int k) {
        const double* wk = ws + (size_t)k * BL::WS_STAGE;
        const int q = (side == 0) ? k : k - 1;                       // stage whose controls go with this block
        const double* wq = ws + (size_t)q * BL::WS_STAGE;
#pragma unroll
        for (int t = 0; t < NX; ++t) {
            bb.bl[t] = wk[BL::WS_L + t * NX + row];   // L[t][row]  (t >= row; 1 / L[row][row] at t == row)
            bb.by[t] = wk[BL::WS_Y + t * NX + row];   // Y[t][row]
            bb.bzx[t] = wq[BL::WS_ZX + urow * NX + t];
            bb.bzp[t] = wq[BL::WS_ZP + urow * NX + t];
        }
#pragma unroll
        for (int b = 0; b < NU; ++b) bb.bluu[b] = wq[BL::WS_LUU + b * NU + urow];   // Luu[b][urow]
        bb.byv  = wk[BL::WS_YV + row];
        bb.byu  = wq[BL::WS_YU + urow];
        bb.bfix = p.comp[k * S + row].fixed;
        bb.bx   = (lane < NX) ? xin[k * S + lane] : 0.0;
        bb.bxu  = (lane >= NX && lane < S) ? xin[q * S + lane] : 0.0;
    };
    // (The prefetch is UNCONDITIONAL -- clamped block index -- and a step beyond the end is computed and discarded: loads issued on
    // one side of a branch only make the compiler fall back to s_waitcnt vmcnt(0) at the next use, which serialises the prefetch.)
    const int last_s = (mysteps > 0) ? mysteps - 1 : 0;
    auto back_step = [&](BackBuf& bb, int s) {
        const bool valid = (s < mysteps);
        const int k = blk_of(valid ? s : last_s);
        const int q = (side == 0) ? k : k - 1;
        double Lc[NX], v = bb.byv, zx[NX], zp[NX], lu[NU];
        const double yu_r = bb.byu, xin_r = bb.bx, xinu_r = bb.bxu;
        const bool fixed_r = (bb.bfix != 0);
#pragma unroll
        for (int t = 0; t < NX; ++t) {
            Lc[t] = bb.bl[t]; zx[t] = bb.bzx[t]; zp[t] = bb.bzp[t];
            v -= bb.by[t] * lane_bcast(xn_r, t);   // t = y - Y^T x_neighbour
        }
#pragma unroll
        for (int b = 0; b < NU; ++b) lu[b] = bb.bluu[b];
        fetch_b(bb, blk_of((s + 2 < mysteps) ? s + 2 : last_s));
        // x_k = L^{-T} t across the lanes
        double xk = 0.0;
#pragma unroll
        for (int j = NX - 1; j >= 0; --j) {
            const double xj = lane_bcast(v * Lc[j], j);       // lane j: v_j / L[j][j]
            if (lane == j) xk = xj;
            v -= (lane < j) ? Lc[j] * xj : 0.0;               // lane i < j: L[j][i] x_j
        }
        if (fixed_r || !own) xk = 0.0;
        // u_q = Luu^{-T} (yu - Zx x_q - Zp x_{q+1}) : lane a < NU owns row a of Zx, Zp and column a of Luu
        //   wave 0: x_q = x_k (just computed), x_{q+1} = neighbour; wave 1: x_q = neighbour (x_{k-1}), x_{q+1} = x_k
        double w = yu_r;
#pragma unroll
        for (int t = 0; t < NX; ++t) {
            const double xa = lane_bcast(xk, t), xb = lane_bcast(xn_r, t);
            w -= (side == 0) ? zx[t] * xa + zp[t] * xb : zx[t] * xb + zp[t] * xa;
        }
        double uk = 0.0;
#pragma unroll
        for (int a = NU - 1; a >= 0; --a) {
            const double ua = lane_bcast(w * lu[a], a);   // lane a: w_a / Luu[a][a]
            if (lane == a) uk = ua;
            w -= (lane < a) ? lu[a] * ua : 0.0;           // lane b < a: Luu[a][b] u_a
        }
        if (lane >= NU) uk = 0.0;
        dn2 += valid ? xk * xk + uk * uk : 0.0;
        double ush = 0.0;
#pragma unroll
        for (int a = 0; a < NU; ++a) { const double ua = lane_bcast(uk, a); if (lane == NX + a) ush = ua; }
        if (valid && lane < NX) xt[k * S + lane] = xin_r + xk;
        else if (valid && lane < S) xt[q * S + lane] = xinu_r + ush;
        xn_r = valid ? xk : xn_r;
    };
    fetch_b(b0, blk_of(0 < mysteps ? 0 : last_s));
    fetch_b(b1, blk_of(1 < mysteps ? 1 : last_s));
    CHAIN_STAMP(4);   // meeting block
    for (int s = 0; s < mysteps; s += 2) {
        back_step(b0, s);
        back_step(b1, s + 1);
    }
    CHAIN_STAMP(5);   // back-substitution
    if (tl_on)
        for (int i = 0; i < 6; ++i) p.timeline[i] = tl_acc[i];
    if (threadIdx.x == 0) {
        xt[p.off_dt] = xin[p.off_dt];
        if (p.off_dt + 1 < p.nvs) xt[p.off_dt + 1] = 0.0;
    }
    y2  = wave_sum(y2);
    dn2 = wave_sum(dn2);
    if (lane == 0) { xch[2 * NX + 2 * side] = y2; xch[2 * NX + 2 * side + 1] = dn2; }
    lds_barrier();
    if (threadIdx.x == 0) {
        y2  = xch[2 * NX] + xch[2 * NX + 2];
        dn2 = xch[2 * NX + 1] + xch[2 * NX + 3];
        st->mu_acc = mu_eff;
        st->first  = 0;
        st->fresh  = 0;
        st->n_fact += 1;
        st->inner += 1;
        const double dnorm = sqrt(dn2);
        st->dnorm = dnorm;
        int no_trial;
        if (dnorm <= LM_EPS2) { stop = 1; no_trial = 1; }
        else { no_trial = 0; st->den = mu * dn2 + y2; }
        st->stop     = stop;
        st->no_trial = no_trial;
    }
}

// ---- per instance: the sequential part, second formulation ("stacked" twisted block Cholesky with explicit back-substitution
//      operators).  Same two waves from both ends as big_chain_kernel, but
//      * ONE right-looking Cholesky pass over register rows factors the block AND produces everything that depends on its factor: the
//        wave holds the stacked matrix  [ D_k ; C ; g^T ; I ]  (lanes 0..NX-1: rows of the state block, 16..16+NX-1: rows of the
//        coupling to the next block of the sequence, lane 28: the right-hand side as a row, 32..32+NX-1: rows of the identity); the column
//        operations of the factorisation turn it into  [ L ; Y = C L^{-T} ; y^T = g^T L^{-T} ; W = L^{-T} ]  -- the rows are independent,
//        so the extra rows ride in lanes that were idle (the separate triangular solves of big_chain_kernel: 350 instructions per step);
//      * the back-substitution  x_k = L^{-T} (y - Y^T x_next)  is prepared as  x_k = a_k - G_k x_next  with  a_k = W y  (one more dot
//        product per lane in the same loop that forms Y y for the right-hand-side mailbox and |y|^2) and  G_k = W Y^T  (three
//        matrix-core instructions next to the three of the Schur complement Y Y^T).  A back-substitution step is then ONE 12 x 12
//        matrix-vector product on data that was prefetched steps ahead -- 13 doubles per lane instead of 57 -- with no triangular
//        solve in the dependent chain (big_chain_kernel: 6 k cycles per step, here a few hundred);
//      * the controls and the trial iterate are not part of the chain at all: the state increments are collected in LDS and a
//        stage-parallel epilogue (all lanes of both waves) forms  u_q = L_uu^{-T} (y_u - Z_x dx_q - Z_p dx_{q+1}),  x + delta,  |delta|^2;
//      * each wave owns its LDS areas, so the elimination loop has NO workgroup barrier (the waves meet twice: at the middle block
//        and before the epilogue).
//      W = L^{-T} is formed explicitly (condition of a damped 12 x 12 block: <= 1e5; the step changes at the 1e-11 level, five orders
//      below the finite-difference noise of the Jacobian).
template <int NX, int NU>
struct Chain2Lds {
    static constexpr int DN = 0;                     // [NX][NX] Schur mailbox of the wave
    static constexpr int GN = DN + NX * NX;          // [16]     right-hand-side mailbox
    static constexpr int YL = GN + 16;               // [16][NX] Y rows (rows >= NX stay zero: matrix-core operand padding)
    static constexpr int WL = YL + 16 * NX;          // [16][NX] W rows
    static constexpr int PER_WAVE = WL + 16 * NX;
    __host__ __device__ static constexpr int total(int N) { return 2 * PER_WAVE + N * NX + 16; }   // + state increments + sums
};

template <int NX, int NU>
__global__ __launch_bounds__(256) void big_chain2_kernel(const FactorParams p)
{
    using BL = BigLds<NX, NU>;
    using CL = Chain2Lds<NX, NU>;
    constexpr int S = NX + NU;
    static_assert(NX <= 12 && NX % 4 == 0 && NU <= NX, "lane roles of the stacked pass: D rows 0.., C rows 16.., rhs row 28, identity rows 32..");
    extern __shared__ __attribute__((aligned(16))) double sm_all[];
    // one workgroup = TWO instances x two waves: its four waves land on the four SIMDs of a compute unit, one each.  (Workgroups of two
    // waves were placed two to a SIMD pair: at 512 instances half of the SIMDs idled while the others were shared -- 461 us per launch
    // against 369 us; rocprofv3 kernel trace over the batch size, DESIGN.md.)
    const int pair = threadIdx.x >> 7, tid = threadIdx.x & 127;
    const int side = tid >> 6, lane = tid & 63;
    if (2 * (int)blockIdx.x + pair >= p.batch) return;   // odd batch: the last workgroup holds one instance (ended waves do not hold barriers)
    double* sm  = sm_all + (size_t)pair * CL::total(p.N);
    double* wsm = sm + side * CL::PER_WAVE;
    double *Dn = wsm + CL::DN, *gnl = wsm + CL::GN, *Yl = wsm + CL::YL, *Wl = wsm + CL::WL;
    double* dxs = sm + 2 * CL::PER_WAVE;             // [N][NX] delta x of every block
    const int inst = 2 * blockIdx.x + pair + p.inst0;
    LmState* st = p.st + inst;
    const int done = st->done;
    int stop = st->stop;
    const double mu = st->mu;
    const double mu_eff = (st->fresh ? 0.0 : st->mu_acc) + mu;
    if (done) return;
    const int N = p.N;
    double* sums = dxs + N * NX;                     // [8] reductions across the waves
    const int m = N / 2;                             // meeting block
    const double* xin = p.x + (size_t)inst * p.nvs;
    double* xt        = p.xt + (size_t)inst * p.nvs;
    double* ws        = p.work + (size_t)inst * p.work_stride;
    // lane roles
    const bool isD = lane < NX, isC = lane >= 16 && lane < 16 + NX, isG = lane == 28, isI = lane >= 32 && lane < 32 + NX;
    const int row = isD ? lane : (isC ? lane - 16 : (isI ? lane - 32 : 0));
    for (int e = lane; e < CL::PER_WAVE; e += 64) wsm[e] = 0.0;   // mailboxes empty, operand padding zero
    const int mysteps = (side == 0) ? m : N - 1 - m;
    auto block_of = [&](int s) { return (side == 0) ? s : N - 1 - s; };
    // ---- prefetch of a block's assembled data.  BRANCH-FREE: every lane loads NX + NX + 3 values through per-lane offsets that encode
    //      its role (lanes without a role read the block's first words and ignore them) -- loads inside divergent branches made the
    //      compiler put an s_waitcnt vmcnt(0) behind every one of them (measured: 12 k cycles per step instead of 2.5 k).
    //      pm: D rows: own parts of D_k | C rows: coupling (wave 1: transposed) | rhs row: own rhs.   pe: D rows: DN of the stage that
    //      feeds this block (wave 0: into the mailbox of the next block; wave 1: added to this block) | rhs row: GN likewise.
    const int back = (side == 0) ? 0 : BL::WS_STAGE;   // wave 1 takes coupling / DN / GN from stage k-1
    int off_a = 0, str_a = 1, off_b = 0, off_g = 0;
    if (isD) { off_a = BL::WS_L + row * NX; off_b = BL::WS_DN + row * NX - back; }
    if (isC) { off_a = (side == 0) ? BL::WS_Y + row * NX : BL::WS_Y + row - back; str_a = (side == 0) ? 1 : NX; off_g = BL::WS_GN + row - back; }
    if (isG) { off_a = BL::WS_YV; off_b = BL::WS_GN - back; }
    // (lanes without a role and wave 1's dummy offsets never go below the instance's workspace: wave 1 starts at block N-1 >= 1)
    double pm[NX], pe[NX], pgn = 0.0, py2 = 0.0;
    int pfix = 0;
    auto fetch = [&](int k) {
        const double* wk = ws + (size_t)k * BL::WS_STAGE;
#pragma unroll
        for (int cc = 0; cc < NX; ++cc) { pm[cc] = wk[off_a + cc * str_a]; pe[cc] = wk[off_b + cc]; }
        pgn  = wk[off_g];
        py2  = wk[BL::WS_Y2];
        pfix = p.comp[k * S + row].fixed;
    };
    // the stacked right-looking pass (see the head comment); returns this lane's dot product with y
    // Look-ahead order: column j+1 is updated first, then the reciprocal square root of its pivot (v_rsq_f64 + the math library's
    // Newton step, written out: five dependent instructions) is interleaved with the remaining column updates of step j -- a wave that
    // has its SIMD to itself has nobody else to hide that chain behind.  Same operations as rsqrt(): bit-identical results.
    auto stacked_pass = [&](double (&mrow)[NX]) -> double {
        double inv = rsqrt(lane_bcast(mrow[0], 0));
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            mrow[j] = (lane == j) ? inv : mrow[j] * inv;
            if (j + 1 < NX) {
                mrow[j + 1] -= mrow[j] * lane_bcast(mrow[j], j + 1);
                const double d = lane_bcast(mrow[j + 1], j + 1);
                double y = 0.0, t = 0.0, e = 0.0, u = 0.0, w = 0.0, r = 0.0;
                const int REM = NX - (j + 2);   // (compile-time after unrolling)
#pragma unroll
                for (int q = 0; q < NX; ++q) {
                    if (q == 0) y = __builtin_amdgcn_rsq(d);
                    if (q == 1) t = y * (-d);
                    if (q == 2) e = __builtin_fma(t, y, 1.0);
                    if (q == 3) { u = y * e; w = __builtin_fma(e, 0.375, 0.5); }
                    if (q == 4) r = __builtin_fma(u, w, y);
                    if (q == 5) inv = __builtin_amdgcn_class(y, 0x180) ? r : y;   // (+normal | +denormal: refined; else rsq's own inf / nan / 0)
                    if (q < REM) mrow[j + 2 + q] -= mrow[j] * lane_bcast(mrow[j], j + 2 + q);   // (the pivot lanes' upper entries are unused)
                }
            }
        }
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < NX; ++t) acc += mrow[t] * lane_bcast(mrow[t], 28);
        return acc;
    };
    typedef double d4_t __attribute__((ext_vector_type(4)));
    const int lj = lane & 15, lk = lane >> 4;
    double y2 = 0.0;
    long long tl_acc[6] = {0, 0, 0, 0, 0, 0}, tl_t = 0;   // diagnostics (p.timeline): cycles of wave 0 per part, summed over the steps
#define CHAIN2_STAMP(slot) do { if (p.timeline) { const long long now_ = clock64(); tl_acc[slot] += now_ - tl_t; tl_t = now_; } } while (0)
    if (p.timeline) tl_t = clock64();
    fetch(block_of(0));
    // ---- elimination (no workgroup barrier: the LDS areas belong to the wave)
    for (int s = 0; s < mysteps; ++s) {
        const int k = block_of(s);
        const unsigned long long fmask = __ballot(pfix != 0 && isD);
        const bool fixed_r = (fmask >> row) & 1ull;
        double mrow[NX];
#pragma unroll
        for (int cc = 0; cc < NX; ++cc) {   // (selects, no branches)
            const bool fixed_c = (fmask >> cc) & 1ull;
            const double unit  = (row == cc) ? 1.0 : 0.0;
            const double mail  = isD ? Dn[row * NX + cc] : gnl[cc];
            const double sum   = pm[cc] + mail + ((side == 1) ? pe[cc] : 0.0);
            double v = 0.0;
            v = isC ? pm[cc] : v;
            v = isI ? unit : v;
            v = isD ? ((fixed_r || fixed_c) ? unit : sum) : v;
            v = isG ? (fixed_c ? 0.0 : sum) : v;
            mrow[cc] = v;
        }
        if (isD) {   // the mailbox of the next block starts from this stage's contribution (wave 0) / empty (wave 1)
#pragma unroll
            for (int cc = 0; cc < NX; ++cc) Dn[row * NX + cc] = (side == 0) ? pe[cc] : 0.0;
        }
        const double gn_base = (side == 0) ? pgn : 0.0;
        y2 += (lane == 0) ? py2 : 0.0;
        CHAIN2_STAMP(0);   // waited for the block's data, combined it
        fetch(block_of((s + 1 < mysteps) ? s + 1 : s));
        const double acc = stacked_pass(mrow);
        CHAIN2_STAMP(1);   // stacked Cholesky pass + dot products
        if (isG) y2 += acc;                               // |y|^2
        if (isC) gnl[row] = gn_base - acc;                // rhs mailbox: GN - Y y
        double* wk = ws + (size_t)k * BL::WS_STAGE;
        if (isI) wk[BL::WS_YV + row] = acc;               // a_k = W y
        if (isC || isI) {
            double* dst = (isC ? Yl : Wl) + row * NX;
#pragma unroll
            for (int cc = 0; cc < NX; ++cc) dst[cc] = mrow[cc];
        }
        // matrix cores: S = Y Y^T -> mailbox, G = W Y^T -> workspace (operand (i, k) of lane l: i = l % 16, k = k0 + l / 16)
        d4_t accS = {0.0, 0.0, 0.0, 0.0}, accG = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k0 = 0; k0 < NX; k0 += 4) {
            const double yv = Yl[lj * NX + k0 + lk];
            const double wv = Wl[lj * NX + k0 + lk];
            accS = __builtin_amdgcn_mfma_f64_16x16x4f64(yv, yv, accS, 0, 0, 0);
            accG = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, yv, accG, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * r + lk, j = lj;
            if (i < NX && j < NX) {
                Dn[i * NX + j] -= accS[r];
                wk[BL::WS_L + i * NX + j] = accG[r];      // G_k
            }
        }
        CHAIN2_STAMP(2);   // LDS hand-over, matrix cores, stores
    }
    // ---- the meeting block m (wave 0): own parts + both mailboxes
    __syncthreads();
    if (side == 0) {
        const double* wk = ws + (size_t)m * BL::WS_STAGE;
        const int fx = isD ? p.comp[m * S + row].fixed : 0;
        const unsigned long long fmask = __ballot(fx != 0);
        const bool fixed_r = (fmask >> row) & 1ull;
        const double* Dn1 = sm + CL::PER_WAVE + CL::DN;
        const double* gn1 = sm + CL::PER_WAVE + CL::GN;
        double mrow[NX];
#pragma unroll
        for (int cc = 0; cc < NX; ++cc) {
            double v = 0.0;
            if (isD) {
                v = wk[BL::WS_L + row * NX + cc] + Dn[row * NX + cc] + Dn1[row * NX + cc];
                if (fixed_r || ((fmask >> cc) & 1ull)) v = (row == cc) ? 1.0 : 0.0;
            }
            else if (isG) {
                v = wk[BL::WS_YV + cc] + gnl[cc] + gn1[cc];
                if ((fmask >> cc) & 1ull) v = 0.0;
            }
            else if (isI) v = (row == cc) ? 1.0 : 0.0;
            mrow[cc] = v;
        }
        y2 += (lane == 0) ? wk[BL::WS_Y2] : 0.0;
        const double acc = stacked_pass(mrow);
        if (isG) y2 += acc;
        if (isI) dxs[m * NX + row] = acc;                 // x_m = W y
    }
    __syncthreads();
    CHAIN2_STAMP(3);   // meeting block
    // ---- back-substitution outwards: x_k = a_k - G_k x_neighbour, lane r < NX owns component r; data two steps ahead
    {
        struct BackBuf { double g[NX], a; };
        BackBuf b0{}, b1{};
        auto blk_of = [&](int s) { return (side == 0) ? m - 1 - s : m + 1 + s; };
        const int last_s = (mysteps > 0) ? mysteps - 1 : 0;
        const int rr = isD ? lane : 0;
        auto fetch_b = [&](BackBuf& bb, int s) {
            const double* wk = ws + (size_t)blk_of(s < mysteps ? s : last_s) * BL::WS_STAGE;
#pragma unroll
            for (int i = 0; i < NX; ++i) bb.g[i] = wk[BL::WS_L + rr * NX + i];
            bb.a = wk[BL::WS_YV + rr];
        };
        double xn = isD ? dxs[m * NX + lane] : 0.0;
        auto back_step = [&](BackBuf& bb, int s) {
            const bool valid = (s < mysteps);
            double v = bb.a;
#pragma unroll
            for (int i = 0; i < NX; ++i) v -= bb.g[i] * lane_bcast(xn, i);
            fetch_b(bb, s + 2);
            if (valid && isD) dxs[blk_of(s) * NX + lane] = v;
            xn = valid ? v : xn;
        };
        if (mysteps > 0) {
            fetch_b(b0, 0);
            fetch_b(b1, 1);
            for (int s = 0; s < mysteps; s += 2) {
                back_step(b0, s);
                back_step(b1, s + 1);
            }
        }
    }
    __syncthreads();
    CHAIN2_STAMP(4);   // back-substitution
    // ---- epilogue, stage-parallel over both waves: trial iterate of the states, controls, step norm
    double dn2 = 0.0;
    for (int e = tid; e < N * NX; e += 128) {
        const int k = e / NX, r = e - k * NX;
        const double d = p.comp[k * S + r].fixed ? 0.0 : dxs[e];
        dn2 += d * d;
        xt[k * S + r] = xin[k * S + r] + d;
    }
    for (int q = tid; q < N - 1; q += 128) {
        const double* wq = ws + (size_t)q * BL::WS_STAGE;
        double w[NU];
#pragma unroll
        for (int a = 0; a < NU; ++a) {
            double v = wq[BL::WS_YU + a];
#pragma unroll
            for (int t = 0; t < NX; ++t) v -= wq[BL::WS_ZX + a * NX + t] * dxs[q * NX + t] + wq[BL::WS_ZP + a * NX + t] * dxs[(q + 1) * NX + t];
            w[a] = v;
        }
#pragma unroll
        for (int a = NU - 1; a >= 0; --a) {   // u = L_uu^{-T} w (diagonal stored inverted)
            double v = w[a];
#pragma unroll
            for (int b = a + 1; b < NU; ++b) v -= wq[BL::WS_LUU + b * NU + a] * w[b];
            w[a] = v * wq[BL::WS_LUU + a * NU + a];
            dn2 += w[a] * w[a];
            xt[q * S + NX + a] = xin[q * S + NX + a] + w[a];
        }
    }
    if (tid == 0) {
        xt[p.off_dt] = xin[p.off_dt];
        if (p.off_dt + 1 < p.nvs) xt[p.off_dt + 1] = 0.0;
    }
    CHAIN2_STAMP(5);   // epilogue
    if (p.timeline && blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) p.timeline[i] = tl_acc[i];
    y2  = wave_sum(y2);
    dn2 = wave_sum(dn2);
    if (lane == 0) { sums[2 * side] = y2; sums[2 * side + 1] = dn2; }
    __syncthreads();
    if (tid == 0) {
        y2  = sums[0] + sums[2];
        dn2 = sums[1] + sums[3];
        st->mu_acc = mu_eff;
        st->first  = 0;
        st->fresh  = 0;
        st->n_fact += 1;
        st->inner += 1;
        const double dnorm = sqrt(dn2);
        st->dnorm = dnorm;
        int no_trial;
        if (dnorm <= LM_EPS2) { stop = 1; no_trial = 1; }
        else { no_trial = 0; st->den = mu * dn2 + y2; }
        st->stop     = stop;
        st->no_trial = no_trial;
    }
}

// ---- per instance: the sequential part, third formulation ("partitioned" stacked block Cholesky).  The twisted elimination of
//      big_chain2_kernel has a dependent depth of N / 2 block steps per instance (cfg 5: 100 steps of ~5.8 k cycles = the 0.29 ms a
//      single instance needs per LM pass, whatever the batch).  Here the chain is cut by NSEG - 1 SEPARATOR blocks s_j = j N / NSEG into
//      NSEG segments, every segment is eliminated from both of its ends by a pair of waves (2 NSEG waves per instance, N / (2 NSEG)
//      steps each), the separators are solved as a small reduced chain, and the back-substitution runs outwards in every segment at once:
//      * a wave that starts next to a separator carries the fill that its eliminations create towards that separator (the "spike"
//        T_k = H[sep, k], one more 12 x 12 block) along the segment: the stacked matrix of a step is  [ D_k ; C ; S ; g^T ; I ]  with the
//        rows of the spike in lanes that were idle in big_chain2_kernel (D rows: lanes 0.., C: 16.., g: 28, S: 32.., I: 48..), the same
//        right-looking pass turns it into  [ L ; Y = C L^-T ; Z = S L^-T ; y^T ; W = L^-T ];
//      * per step five products on the matrix cores instead of two:  Y Y^T (next diagonal block),  Z Y^T (the next block's spike),
//        Z Z^T (accumulated for the separator),  G = W Y^T and Gs = W Z^T (back-substitution:  x_k = a_k - G_k x_next - Gs_k x_sep);
//      * the meeting block of a segment is the same step with "next" = the segment's other separator: its Y Z^T is the coupling of the
//        two separators in the reduced chain;
//      * reduced chain (NSEG - 1 blocks, one wave), then x_meeting, then both waves of every segment outwards.
//      Same numbers as a Cholesky factorisation in the order [segment interiors, outside in | meeting blocks | separators]: the step
//      differs from big_chain2_kernel's at rounding level (1e-13 relative), like that one's from Eigen's AMD order.
template <int NX, int NU, int NSEG, bool ARROW = false>
struct Chain3Lds {
    static constexpr int NN = NX * NX;
    static constexpr int DN = 0;                     // [NX][NX] Schur mailbox: -Y Y^T (+ the stage's contribution) for the next block of the sequence
    static constexpr int TM = DN + NN;               // [NX][NX] spike mailbox: rows of the separator, columns of the next block
    static constexpr int DS = TM + NN;               // [NX][NX] what this wave adds to the diagonal block of its separator
    static constexpr int YL = DS + NN;               // [NX][NX] Y rows   (matrix-core operands)
    static constexpr int ZL = YL + NN;               // [NX][NX] Z rows
    static constexpr int WL = ZL + NN;               // [NX][NX] W rows
    static constexpr int GN = WL + NN;               // [16] right-hand-side mailbox
    static constexpr int GS = GN + 16;               // [16] what this wave adds to the right-hand side of its separator
    static constexpr int GN2 = GS + 16;              // (free dt) the same two mailboxes of the second right-hand side: the border column
    static constexpr int GS2 = GN2 + 16;
    static constexpr int PER_WAVE = ARROW ? GS2 + 16 : GS + 16;
    // Nothing else is shared but the state increments: what the phases after the elimination need lives in wave areas that are dead by then --
    //   the separator behind a downward wave is staged by that wave in its own YL (own parts of the diagonal block) and ZL ([0,12) right-hand side,
    //   [12] |y_u|^2 of the stage, [13] fixed mask), the coupling of a segment's two separators (-Z Y^T of its meeting block; rows: left
    //   separator) replaces the upward wave's spike mailbox TM, and the reduced chain (wave 1) works with wave 0's YL / WL as operands, wave 0's ZL /
    //   GN as its Schur / right-hand-side mailbox and keeps separator j's back-substitution operator / vector in ZL / GN of the upward wave 2 j.
    static constexpr int IDM = 2 * NSEG * PER_WAVE;  // [NX][NX] identity: the I rows of the stacked matrix read their start values like the other rows read their mailboxes
    static constexpr int DXS = IDM + NN;
    static constexpr int BSUM = ARROW ? 12 * NSEG + 8 : 0;   // (free dt) per wave: z.y, |z|^2 and the four stage scalars; [12 NSEG] delta_dt, [12 NSEG + 1] y_dt^2
    __host__ __device__ static constexpr int total(int N) { return DXS + N * NX + 4 * NSEG + 8 + BSUM; }   // + state increments + sums (NSEG = 4, N = 200: 76.7 KB, two workgroups per CU)
};

// (the body: `st` = the instance's LM state in LDS, `sm` = the chain's LDS area, `xs` = optional LDS copy of the trial iterate.  A residual sweep fused
// behind the chain in the same launch -- sweep_body on that copy, 8 waves -- was built and measured: cfg 5 9.09 -> 9.36 ms, one OCP 1.83 -> 1.96 ms: the
// sweep then runs at the chain's occupancy, one workgroup per CU, and its dependent descriptor loads are no longer hidden by co-resident workgroups; removed)
template <int NX, int NU, int NSEG, bool ARROW = false>
__device__ __forceinline__ void big_chain3_body(const FactorParams& p, LmState* const st, double* const sm, const int inst, double* const xs)
{
    using BL = BigLds<NX, NU>;
    using CL = Chain3Lds<NX, NU, NSEG, ARROW>;
    constexpr int S = NX + NU, NN = NX * NX, NW = 2 * NSEG, THREADS = 128 * NSEG;
    static_assert(NX <= 12 && NX % 2 == 0 && NU <= NX, "lane roles of the stacked pass: D rows 0.., C rows 16.., rhs row 28, spike rows 32.., identity rows 48..; rows move as double2");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int seg = wave >> 1, side = wave & 1;
    int stop = st->stop;
    const double mu = st->mu;
    const double mu_eff = (st->fresh ? 0.0 : st->mu_acc) + mu;
    const int N = p.N;
    double* wsm = sm + wave * CL::PER_WAVE;
    double *Dn = wsm + CL::DN, *Tm = wsm + CL::TM, *Ds = wsm + CL::DS, *Yl = wsm + CL::YL, *Zl = wsm + CL::ZL, *Wl = wsm + CL::WL, *gnl = wsm + CL::GN, *gs = wsm + CL::GS;
    double* dxs  = sm + CL::DXS;                     // [N][NX] delta x of every block
    double* sums = dxs + N * NX;                     // [2 NW + ..] reductions across the waves
    double* bsum = sums + 4 * NSEG + 8;              // (free dt) [6 NW] per wave: z.y, |z|^2, sums of the stage scalars; then delta_dt, y_dt^2
    double *gnl2 = wsm + CL::GN2, *gs2 = wsm + CL::GS2;   // (only touched when ARROW)
    const double* xin = p.x + (size_t)inst * p.nvs;
    double* xt        = p.xt + (size_t)inst * p.nvs;
    double* ws        = p.work + (size_t)inst * p.work_stride;
    // ---- the wave's part of the chain: segment [a, b] between the separators a - 1 and b + 1, meeting block m
    auto sep_of = [&](int j) { return (int)(((long long)j * N) / NSEG); };   // j = 1 .. NSEG - 1
    const int a = (seg == 0) ? 0 : sep_of(seg) + 1;
    const int b = (seg == NSEG - 1) ? N - 1 : sep_of(seg + 1) - 1;
    const int m = a + (b - a + 1) / 2;
    const int mysteps = (side == 0) ? m - a : b - m;
    const bool spike  = (side == 0) ? (seg > 0) : (seg < NSEG - 1);   // (wave-uniform) there is a separator behind the wave's first block
    auto block_of = [&](int s) { return (side == 0) ? a + s : b - s; };
    // lane roles
    const bool isD = lane < NX, isC = lane >= 16 && lane < 16 + NX, isG = lane == 28, isS = lane >= 32 && lane < 32 + NX, isI = lane >= 48 && lane < 48 + NX;
    const bool isG2 = ARROW && lane == 29;           // free dt: the border column rides along as a second right-hand-side row
    const int row = isD ? lane : (isC ? lane - 16 : (isS ? lane - 32 : (isI ? lane - 48 : 0)));
    // ---- mailboxes: empty, except next to a separator (the separator's stage contributes to the first block of an upward wave; the
    //      direct coupling of the first block to the separator is the spike the wave starts with)
    for (int e = lane; e < CL::PER_WAVE; e += 64) wsm[e] = 0.0;
    if (spike) {
        if (side == 0) {
            const double* wq = ws + (size_t)(a - 1) * BL::WS_STAGE;   // the separator's own stage record
            for (int e = lane; e < NN; e += 64) {
                const int r = e / NX, c = e - r * NX;
                Dn[e] = wq[BL::WS_DN + e];
                Tm[e] = wq[BL::WS_Y + c * NX + r];                    // H[sep, a] = H[a, sep]^T
            }
            if (lane < NX) gnl[lane] = wq[BL::WS_GN + lane];
            if constexpr (ARROW) { if (lane < NX) gnl2[lane] = wq[BL::WS_BN + lane]; }
        }
        else {
            const double* wq = ws + (size_t)b * BL::WS_STAGE;         // stage b: couples block b to the separator b + 1
            for (int e = lane; e < NN; e += 64) {
                Tm[e] = wq[BL::WS_Y + e];                             // H[sep, b]
                Ds[e] = wq[BL::WS_DN + e];
            }
            if (lane < NX) gs[lane] = wq[BL::WS_GN + lane];
            if constexpr (ARROW) { if (lane < NX) gs2[lane] = wq[BL::WS_BN + lane]; }
        }
    }
    if constexpr (ARROW) {   // the stage scalars of the border: every wave sums its share of the stages
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int k = tid; k < N; k += THREADS) {
            const double* sc4 = ws + (size_t)k * BL::WS_STAGE + BL::WS_SC;
            s0 += sc4[0]; s1 += sc4[1]; s2 += sc4[2]; s3 += sc4[3];
        }
        s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
        if (lane == 0) { bsum[6 * wave + 2] = s0; bsum[6 * wave + 3] = s1; bsum[6 * wave + 4] = s2; bsum[6 * wave + 5] = s3; }
    }
    if (wave == 0)
        for (int e = lane; e < NN; e += 64) sm[CL::IDM + e] = (e / NX == e % NX) ? 1.0 : 0.0;
    // how a lane forms its row of the stacked matrix from what it prefetched (pm, pe) and its mailbox: coefficients instead of selects
    //   D: pm + mailbox (+ pe downward) | C: pm | S: mailbox | g: pm + mailbox (+ pe downward) | I: "mailbox" = a row of the identity
    const double cpm   = (isD || isC || isG || isG2) ? 1.0 : 0.0;
    const double cmail = (isD || isS || isG || isG2 || isI) ? 1.0 : 0.0;
    const double cpe   = (side == 1 && (isD || isG || isG2)) ? 1.0 : 0.0;
    const double2* mailp = reinterpret_cast<const double2*>(isD ? Dn + row * NX : (isS ? Tm + row * NX : (isI ? sm + CL::IDM + row * NX : (isG2 ? gnl2 : gnl))));
    // ---- prefetch of a block's assembled data, branch-free (see big_chain2_kernel)
    const int back = (side == 0) ? 0 : BL::WS_STAGE;   // downward waves take coupling / DN / GN from stage k-1
    int off_a = 0, str_a = 1, off_b = 0, off_g = 0;
    if (isD) { off_a = BL::WS_L + row * NX; off_b = BL::WS_DN + row * NX - back; }
    if (isC) { off_a = (side == 0) ? BL::WS_Y + row * NX : BL::WS_Y + row - back; str_a = (side == 0) ? 1 : NX; off_g = BL::WS_GN + row - back; }
    if (isG) { off_a = BL::WS_YV; off_b = BL::WS_GN - back; }
    if (isG2) { off_a = BL::WS_BX; off_b = BL::WS_BN - back; }
    const int off_g2 = isC ? BL::WS_BN + row - back : 0;
    double pm[NX], pe[NX], pgn = 0.0, py2 = 0.0, pbn = 0.0;
    int pfix = 0;
    // (per-lane pointers and a per-lane byte stride, advanced load by load: indexing wk[off + cc * stride] cost 130 address instructions per block)
    const double* const pa0 = ws + off_a;
    const double* const pb0 = ws + off_b;
    const double* const pg0 = ws + off_g;
    const double* const pg20 = ws + off_g2;
    const ptrdiff_t sa = str_a;
    const int32_t* const fx0 = &p.comp[row].fixed;
    auto fetch = [&](int k) {
        const size_t ko = (size_t)k * BL::WS_STAGE;
        const double* pa = pa0 + ko;
        const double* pb = pb0 + ko;
#pragma unroll
        for (int cc = 0; cc < NX; ++cc) { pm[cc] = *pa; pa += sa; pe[cc] = pb[cc]; }
        pgn  = pg0[ko];
        if constexpr (ARROW) pbn = pg20[ko];
        py2  = ws[ko + BL::WS_Y2];
        pfix = fx0[(size_t)k * S * (sizeof(CompInfo) / sizeof(int32_t))];
    };
    double acc2 = 0.0;   // (free dt) second result of stacked_pass: the row's product with z
    auto stacked_pass = [&](double (&mrow)[NX]) -> double {   // (big_chain2_kernel: look-ahead order, v_rsq_f64 + the library's Newton step)
        double inv = rsqrt(lane_bcast(mrow[0], 0));
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            mrow[j] *= inv;   // (the pivot lane keeps sqrt(d) instead of its reciprocal: a D row's entries at and right of its pivot are never read again)
            if (j + 1 < NX) {
                mrow[j + 1] -= mrow[j] * lane_bcast(mrow[j], j + 1);
                const double d = lane_bcast(mrow[j + 1], j + 1);
                double y = 0.0, t = 0.0, e = 0.0, u = 0.0, w = 0.0, r = 0.0;
                const int REM = NX - (j + 2);
#pragma unroll
                for (int q = 0; q < NX; ++q) {
                    if (q == 0) y = __builtin_amdgcn_rsq(d);
                    if (q == 1) t = y * (-d);
                    if (q == 2) e = __builtin_fma(t, y, 1.0);
                    if (q == 3) { u = y * e; w = __builtin_fma(e, 0.375, 0.5); }
                    if (q == 4) r = __builtin_fma(u, w, y);
                    if (q == 5) inv = __builtin_amdgcn_class(y, 0x180) ? r : y;
                    if (q < REM) mrow[j + 2 + q] -= mrow[j] * lane_bcast(mrow[j], j + 2 + q);
                }
            }
        }
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < NX; ++t) acc += mrow[t] * lane_bcast(mrow[t], 28);
        if constexpr (ARROW) {   // the same with the second right-hand-side row z^T (lane 29)
            double a2 = 0.0;
#pragma unroll
            for (int t = 0; t < NX; ++t) a2 += mrow[t] * lane_bcast(mrow[t], 29);
            acc2 = a2;
        }
        return acc;
    };
    typedef double d4_t __attribute__((ext_vector_type(4)));
    const int lj = lane & 15, lk = lane >> 4, ljc = (lj < NX) ? lj : 0;
    const bool ljin = lj < NX;
    // the five products of a step: Y Y^T -> dS (-=), Z Y^T -> dT (= -), Z Z^T -> dZ (-=), W Y^T -> gG, W Z^T -> gH (operands: the wave's Yl / Zl / Wl)
    // (Z Z^T is not stored step by step: the wave's accumulator accZ collects it over all of its steps and reaches the separator's area once, at the end)
    auto products = [&](const double* yop, const double* zop, const double* wop, double* dS, double* dT, d4_t& accZ, double* gG, double* gH, const bool with_spike) {
        d4_t accS = {0.0, 0.0, 0.0, 0.0}, accG = accS, accT = accS, accH = accS;
#pragma unroll
        for (int k0 = 0; k0 < NX; k0 += 4) {
            // (block sizes that are not a multiple of the instruction's K = 4: the last step's surplus columns are zeros)
            const bool kin = ljin && (NX % 4 == 0 || k0 + lk < NX);
            const int kc   = (NX % 4 == 0 || k0 + lk < NX) ? k0 + lk : 0;
            double yv = yop[ljc * NX + kc], wv = wop[ljc * NX + kc];
            yv = kin ? yv : 0.0; wv = kin ? wv : 0.0;
            accS = __builtin_amdgcn_mfma_f64_16x16x4f64(yv, yv, accS, 0, 0, 0);
            accG = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, yv, accG, 0, 0, 0);
            if (with_spike) {
                double zv = zop[ljc * NX + kc];
                zv = kin ? zv : 0.0;
                accT = __builtin_amdgcn_mfma_f64_16x16x4f64(zv, yv, accT, 0, 0, 0);
                accZ = __builtin_amdgcn_mfma_f64_16x16x4f64(zv, zv, accZ, 0, 0, 0);
                accH = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, zv, accH, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * r + lk, j = lj;
            if (i < NX && j < NX) {
                const int e = i * NX + j;
                dS[e] -= accS[r];
                gG[e] = accG[r];
                if (with_spike) dT[e] = -accT[r];
                gH[e] = with_spike ? accH[r] : 0.0;
            }
        }
    };
    double y2 = 0.0, zy = 0.0;            // zy: S lanes, (Z y)[row] summed over the wave's steps
    double zy2 = 0.0, bsc = 0.0;          // (free dt) S lanes: (Z z)[row]; lane 28: z . y, lane 29: |z|^2 -- over the wave's pivots
    d4_t accZ = {0.0, 0.0, 0.0, 0.0};
    fetch(block_of(0));
    __syncthreads();   // the identity rows are in place
    // ---- elimination of the wave's blocks (no workgroup barrier: the LDS areas belong to the wave)
    for (int s = 0; s < mysteps; ++s) {
        const int k = block_of(s);
        const unsigned long long fmask = __ballot(pfix != 0 && isD);
        double mrow[NX];
#pragma unroll
        for (int c2 = 0; c2 < NX / 2; ++c2) {   // (pm + mailbox) + pe, the operation order of big_chain2_kernel; the products by 0 / 1 are exact
            const double2 mail = mailp[c2];
            mrow[2 * c2]     = __builtin_fma(cpe, pe[2 * c2], __builtin_fma(cpm, pm[2 * c2], cmail * mail.x));
            mrow[2 * c2 + 1] = __builtin_fma(cpe, pe[2 * c2 + 1], __builtin_fma(cpm, pm[2 * c2 + 1], cmail * mail.y));
        }
        if (fmask) {   // (wave-uniform; block 0 and a terminal equality) fixed components: unit row and column in D, no right-hand side
            asm volatile("" ::: "memory");   // (a real branch: if-converted, these selects ran for every block)
            const bool fixed_r = (fmask >> row) & 1ull;
#pragma unroll
            for (int cc = 0; cc < NX; ++cc) {
                const bool fixed_c = (fmask >> cc) & 1ull;
                const double unit  = (row == cc) ? 1.0 : 0.0;
                double v = mrow[cc];
                v = (isD && (fixed_r || fixed_c)) ? unit : v;
                v = ((isG || isG2) && fixed_c) ? 0.0 : v;
                mrow[cc] = v;
            }
        }
        if (isD) {   // the mailbox of the next block starts from this stage's contribution (upward) / empty (downward)
            double2* dn2 = reinterpret_cast<double2*>(Dn + row * NX);
#pragma unroll
            for (int c2 = 0; c2 < NX / 2; ++c2) dn2[c2] = (side == 0) ? double2{pe[2 * c2], pe[2 * c2 + 1]} : double2{0.0, 0.0};
        }
        const double gn_base = (side == 0) ? pgn : 0.0, gn2_base = (side == 0) ? pbn : 0.0;
        y2 += (lane == 0) ? py2 : 0.0;
        // next block of the wave; behind the last one: the segment's meeting block (its own parts: the upward wave takes it after the barrier)
        fetch((s + 1 < mysteps) ? block_of(s + 1) : m);
        const double acc = stacked_pass(mrow);
        if (isG) y2 += acc;                               // |y|^2
        if (isC) gnl[row] = gn_base - acc;                // rhs mailbox: GN - Y y
        zy += acc;                                        // (S lanes) the separator's right-hand side collects - Z y
        double* wk = ws + (size_t)k * BL::WS_STAGE;
        if (isI) wk[BL::WS_YV + row] = acc;               // a_k = W y
        if constexpr (ARROW) {
            bsc += acc2;                                  // lane 28: y . z ; lane 29: z . z
            if (isC) gnl2[row] = gn2_base - acc2;
            zy2 += acc2;
            if (isI) wk[BL::WS_YV2 + row] = acc2;         // W z
        }
        if (isC || isS || isI) {
            double2* dst = reinterpret_cast<double2*>((isC ? Yl : (isS ? Zl : Wl)) + row * NX);
#pragma unroll
            for (int c2 = 0; c2 < NX / 2; ++c2) dst[c2] = double2{mrow[2 * c2], mrow[2 * c2 + 1]};
        }
        products(Yl, Zl, Wl, Dn, Tm, accZ, wk + BL::WS_L, wk + BL::WS_DN, spike);
    }
    auto flush_sep = [&]() {   // what the wave collected for its separator
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * r + lk, j = lj;
            if (i < NX && j < NX) Ds[i * NX + j] -= accZ[r];
        }
        if (isS) gs[row] -= zy;
        if constexpr (ARROW) { if (isS) gs2[row] -= zy2; }
    };
    if (side == 1) flush_sep();
    if (mysteps == 0 && side == 0) fetch(m);   // (a segment of one or two blocks)
    // a downward wave stages the separator behind its first block for the reduced chain (its loop is over: its operand areas are dead, and
    // the meeting blocks do not touch them)
    if (side == 1 && spike) {
        const int sj = b + 1;
        const double* wq = ws + (size_t)sj * BL::WS_STAGE;
        for (int e = lane; e < NN; e += 64) Yl[e] = wq[BL::WS_L + e];
        if (lane < NX) Zl[lane] = wq[BL::WS_YV + lane];
        if constexpr (ARROW) { if (lane < NX) Zl[16 + lane] = wq[BL::WS_BX + lane]; }
        const unsigned long long fm = __ballot(lane < NX && p.comp[sj * S + (lane < NX ? lane : 0)].fixed != 0);
        if (lane == 0) { Zl[12] = wq[BL::WS_Y2]; Zl[13] = (double)(unsigned)fm; }
    }
    __syncthreads();
    // ---- the meeting block of every segment (upward wave): own parts + both mailboxes, "next" = the right separator, spike = the left one
    if (side == 0) {
        double* wo = wsm + CL::PER_WAVE;                  // the downward wave's areas
        const double *Dn1 = wo + CL::DN, *gn1 = wo + CL::GN, *Tm1 = wo + CL::TM, *gn1b = wo + CL::GN2;
        double *Ds1 = wo + CL::DS, *gs1 = wo + CL::GS, *gs1b = wo + CL::GS2;
        const unsigned long long fmask = __ballot(pfix != 0 && isD);
        const bool fixed_r = (fmask >> row) & 1ull;
        double mrow[NX];
#pragma unroll
        for (int cc = 0; cc < NX; ++cc) {
            const bool fixed_c = (fmask >> cc) & 1ull;
            const double unit  = (row == cc) ? 1.0 : 0.0;
            double v = 0.0;
            if (isD) v = (fixed_r || fixed_c) ? unit : pm[cc] + Dn[row * NX + cc] + Dn1[row * NX + cc];
            else if (isC) v = Tm1[row * NX + cc];
            else if (isS) v = Tm[row * NX + cc];
            else if (isG) v = fixed_c ? 0.0 : pm[cc] + gnl[cc] + gn1[cc];
            else if (isG2) v = fixed_c ? 0.0 : pm[cc] + gnl2[cc] + gn1b[cc];
            else if (isI) v = unit;
            mrow[cc] = v;
        }
        y2 += (lane == 0) ? py2 : 0.0;
        const double acc = stacked_pass(mrow);
        if (isG) y2 += acc;
        if (isC) gs1[row] -= acc;                         // right separator: - Y y
        zy += acc;                                        // left separator:  - Z y
        double* wk = ws + (size_t)m * BL::WS_STAGE;
        if (isI) wk[BL::WS_YV + row] = acc;
        if constexpr (ARROW) {
            bsc += acc2;
            if (isC) gs1b[row] -= acc2;
            zy2 += acc2;
            if (isI) wk[BL::WS_YV2 + row] = acc2;
        }
        if (isC || isS || isI) {
            double* dst = (isC ? Yl : (isS ? Zl : Wl)) + row * NX;
#pragma unroll
            for (int cc = 0; cc < NX; ++cc) dst[cc] = mrow[cc];
        }
        products(Yl, Zl, Wl, Ds1, Tm, accZ, wk + BL::WS_L, wk + BL::WS_DN, true);   // (the spike mailbox has been read: it takes the separators' coupling)
        flush_sep();
    }
    if constexpr (ARROW) {
        if (wave != 1 && (isG || isG2)) bsum[6 * wave + (isG ? 0 : 1)] = bsc;
    }
    __syncthreads();
    // ---- reduced chain over the separators (wave 1): D_j = own + what the two neighbouring waves collected, coupling to the next
    //      separator = (segment j's  -Z Y^T)^T, plain twisted-free elimination (NSEG - 1 blocks), then its back-substitution
    if (wave == 1) {
        double* w0 = sm;                                  // wave 0's areas (its meeting block is done)
        double *Dr = w0 + CL::ZL, *gr = w0 + CL::GN, *yop = w0 + CL::YL, *wop = w0 + CL::WL, *gr2 = w0 + CL::GN2;
        auto area = [&](int w) { return sm + w * CL::PER_WAVE; };
        for (int e = lane; e < NN; e += 64) Dr[e] = 0.0;
        if (lane < 16) gr[lane] = 0.0;
        if constexpr (ARROW) { if (lane < 16) gr2[lane] = 0.0; }
        double last_a = 0.0, last_a2 = 0.0, ddt_r = 0.0;
        for (int j = 1; j < NSEG; ++j) {
            const double* DsL = area(2 * j - 1) + CL::DS;   // downward wave of the segment on the left
            const double* DsR = area(2 * j) + CL::DS;       // upward wave of the segment on the right
            const double* gsL = area(2 * j - 1) + CL::GS;
            const double* gsR = area(2 * j) + CL::GS;
            const double* own = area(2 * j - 1) + CL::YL;   // staged by the downward wave on the left
            const double* og  = area(2 * j - 1) + CL::ZL;
            const double* gsL2 = area(2 * j - 1) + CL::GS2;
            const double* gsR2 = area(2 * j) + CL::GS2;
            const double* rc  = area(2 * j) + CL::TM;       // segment j lies between separator j and j + 1
            const unsigned long long fmask = (unsigned long long)(unsigned)og[13];
            const bool fixed_r = (fmask >> row) & 1ull;
            const bool last = (j == NSEG - 1);
            double mrow[NX];
#pragma unroll
            for (int cc = 0; cc < NX; ++cc) {
                const bool fixed_c = (fmask >> cc) & 1ull;
                const double unit  = (row == cc) ? 1.0 : 0.0;
                double v = 0.0;
                if (isD) v = (fixed_r || fixed_c) ? unit : own[row * NX + cc] + DsL[row * NX + cc] + DsR[row * NX + cc] + Dr[row * NX + cc];
                else if (isC) v = last ? 0.0 : rc[cc * NX + row];   // H[sep j+1, sep j]
                else if (isG) v = fixed_c ? 0.0 : og[cc] + gsL[cc] + gsR[cc] + gr[cc];
                else if (isG2) v = fixed_c ? 0.0 : og[16 + cc] + gsL2[cc] + gsR2[cc] + gr2[cc];
                else if (isI) v = unit;
                mrow[cc] = v;
            }
            y2 += (lane == 0) ? og[12] : 0.0;
            const double acc = stacked_pass(mrow);
            if (isG) y2 += acc;
            if constexpr (ARROW) bsc += acc2;
            if (last) {
                if constexpr (ARROW) { last_a = acc; last_a2 = acc2; }   // (x = W y - delta_dt W z: behind the last pivot, below)
                else
                if (isI) dxs[sep_of(j) * NX + row] = acc;   // x = W y
            }
            else {
                double* GRj = area(2 * j) + CL::ZL;         // (upward wave 2 j: dead since its meeting block)
                double* ARj = area(2 * j) + CL::GN;
                if (isC) gr[row] = -acc;
                if (isI) ARj[row] = acc;
                if constexpr (ARROW) {
                    if (isC) gr2[row] = -acc2;
                    if (isI) (area(2 * j) + CL::GN2)[row] = acc2;
                }
                if (isC || isI) {
                    double* dst = (isC ? yop : wop) + row * NX;
#pragma unroll
                    for (int cc = 0; cc < NX; ++cc) dst[cc] = mrow[cc];
                }
                if (isD) {
#pragma unroll
                    for (int cc = 0; cc < NX; ++cc) Dr[row * NX + cc] = 0.0;
                }
                products(yop, yop, wop, Dr, nullptr, accZ, GRj, w0 + CL::DN, false);   // (gH: scratch)
            }
        }
        if constexpr (ARROW) {
            // the last pivot: H(dt,dt) + damping - |z|^2  (factor_body's arrowhead, same formulas);  every forward elimination is behind us
            double zyt = lane_bcast(bsc, 28), zzt = lane_bcast(bsc, 29), cdt = 0.0, gdt = 0.0;
            for (int w = 0; w < NW; ++w) {
                if (w != 1) { zyt += bsum[6 * w]; zzt += bsum[6 * w + 1]; }
                cdt += bsum[6 * w + 2]; gdt += bsum[6 * w + 3]; zzt += bsum[6 * w + 4]; zyt += bsum[6 * w + 5];
            }
            const double piv  = (cdt + mu_eff) - zzt;
            const double linv = rsqrt(piv);
            const double ydt  = (gdt - zyt) * linv;
            ddt_r = ydt * linv;
            if (lane == 0) { bsum[6 * NW] = ddt_r; bsum[6 * NW + 1] = ydt * ydt; }
            if constexpr (NSEG > 1) { if (isI) dxs[sep_of(NSEG - 1) * NX + row] = last_a - ddt_r * last_a2; }
        }
        for (int j = NSEG - 2; j >= 1; --j) {
            const double xn = isD ? dxs[sep_of(j + 1) * NX + lane] : 0.0;
            double v = isD ? (area(2 * j) + CL::GN)[lane] : 0.0;
            if constexpr (ARROW) { if (isD) v -= ddt_r * (area(2 * j) + CL::GN2)[lane]; }
            const double* g = area(2 * j) + CL::ZL + (isD ? lane : 0) * NX;
#pragma unroll
            for (int i = 0; i < NX; ++i) v -= g[i] * lane_bcast(xn, i);
            if (isD) dxs[sep_of(j) * NX + lane] = v;
        }
    }
    __syncthreads();
    // ---- back-substitution.  Meeting block (upward wave):  x_m = a_m - G_m x_right - Gs_m x_left
    const double ddt = ARROW ? bsum[6 * NW] : 0.0;   // (written by wave 1 before the barrier above)
    const int rr = isD ? lane : (isC ? lane - 16 : 0);
    const int off_back = isC ? BL::WS_DN + rr * NX : BL::WS_L + rr * NX;   // D lanes: rows of G_k, C lanes: rows of Gs_k
    if (side == 0) {
        const double* wk = ws + (size_t)m * BL::WS_STAGE;
        double g[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) g[i] = wk[off_back + i];
        double am = wk[BL::WS_YV + rr];
        if constexpr (ARROW) am -= ddt * wk[BL::WS_YV2 + rr];
        const double xr = (isD && seg < NSEG - 1) ? dxs[(b + 1) * NX + lane] : 0.0;
        const double xl = (isD && seg > 0) ? dxs[(a - 1) * NX + lane] : 0.0;
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) v += g[i] * (isC ? lane_bcast(xl, i) : lane_bcast(xr, i));
        const double vs = __shfl(v, (lane + 16) & 63);   // D lane r: the spike part computed by lane 16 + r
        if (isD) dxs[m * NX + lane] = am - v - vs;
    }
    __syncthreads();
    // ---- outwards in every segment: x_k = a_k - G_k x_neighbour - Gs_k x_sep, data two steps ahead
    {
        struct BackBuf { double g[NX], a, a2; };
        BackBuf b0{}, b1{};
        auto blk_of = [&](int s) { return (side == 0) ? m - 1 - s : m + 1 + s; };
        const int last_s = (mysteps > 0) ? mysteps - 1 : 0;
        auto fetch_b = [&](BackBuf& bb, int s) {
            const double* wk = ws + (size_t)blk_of(s < mysteps ? s : last_s) * BL::WS_STAGE;
#pragma unroll
            for (int i = 0; i < NX; ++i) bb.g[i] = wk[off_back + i];
            bb.a = wk[BL::WS_YV + rr];
            if constexpr (ARROW) bb.a2 = wk[BL::WS_YV2 + rr];
        };
        const double xsep = (isD && spike) ? dxs[((side == 0) ? a - 1 : b + 1) * NX + lane] : 0.0;
        double xn = isD ? dxs[m * NX + lane] : 0.0;
        auto back_step = [&](BackBuf& bb, int s) {
            const bool valid = (s < mysteps);
            double t = 0.0;   // C lanes: (Gs_k x_sep)[row] -- does not depend on the neighbour's solution
#pragma unroll
            for (int i = 0; i < NX; ++i) t += bb.g[i] * lane_bcast(xsep, i);
            double v = bb.a - __shfl(t, (lane + 16) & 63);
            if constexpr (ARROW) v -= ddt * bb.a2;
#pragma unroll
            for (int i = 0; i < NX; ++i) v -= bb.g[i] * lane_bcast(xn, i);
            fetch_b(bb, s + 2);
            if (valid && isD) dxs[blk_of(s) * NX + lane] = v;
            xn = valid ? v : xn;
        };
        if (mysteps > 0) {
            fetch_b(b0, 0);
            fetch_b(b1, 1);
            for (int s = 0; s < mysteps; s += 2) {
                back_step(b0, s);
                back_step(b1, s + 1);
            }
        }
    }
    __syncthreads();
    // ---- epilogue, stage-parallel over all waves: trial iterate of the states, controls, step norm (big_chain2_kernel)
    double dn2 = 0.0;
    for (int e = tid; e < N * NX; e += THREADS) {
        const int k = e / NX, r = e - k * NX;
        const double d = p.comp[k * S + r].fixed ? 0.0 : dxs[e];
        dn2 += d * d;
        const double v = xin[k * S + r] + d;
        xt[k * S + r] = v;
        if (xs) xs[k * S + r] = v;
    }
    for (int q = tid; q < N - 1; q += THREADS) {
        const double* wq = ws + (size_t)q * BL::WS_STAGE;
        double w[NU];
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            double v = wq[BL::WS_YU + c];
            if constexpr (ARROW) v -= wq[BL::WS_ZU + c] * ddt;
#pragma unroll
            for (int t = 0; t < NX; ++t) v -= wq[BL::WS_ZX + c * NX + t] * dxs[q * NX + t] + wq[BL::WS_ZP + c * NX + t] * dxs[(q + 1) * NX + t];
            w[c] = v;
        }
#pragma unroll
        for (int c = NU - 1; c >= 0; --c) {   // u = L_uu^{-T} w (diagonal stored inverted)
            double v = w[c];
#pragma unroll
            for (int d = c + 1; d < NU; ++d) v -= wq[BL::WS_LUU + d * NU + c] * w[d];
            w[c] = v * wq[BL::WS_LUU + c * NU + c];
            dn2 += w[c] * w[c];
            const double un = xin[q * S + NX + c] + w[c];
            xt[q * S + NX + c] = un;
            if (xs) xs[q * S + NX + c] = un;
        }
    }
    if (tid == 0) {
        xt[p.off_dt] = xin[p.off_dt] + ddt;
        if (xs) xs[p.off_dt] = xin[p.off_dt] + ddt;
        if (p.off_dt + 1 < p.nvs) { xt[p.off_dt + 1] = 0.0; if (xs) xs[p.off_dt + 1] = 0.0; }
    }
    y2  = wave_sum(y2);
    dn2 = wave_sum(dn2);
    if (lane == 0) { sums[2 * wave] = y2; sums[2 * wave + 1] = dn2; }
    __syncthreads();
    if (tid == 0) {
        y2 = 0.0; dn2 = 0.0;
        for (int w = 0; w < NW; ++w) { y2 += sums[2 * w]; dn2 += sums[2 * w + 1]; }
        if constexpr (ARROW) { y2 += bsum[6 * NW + 1]; dn2 += ddt * ddt; }
        st->mu_acc = mu_eff;
        st->first  = 0;
        st->fresh  = 0;
        st->n_fact += 1;
        st->inner += 1;
        const double dnorm = sqrt(dn2);
        st->dnorm = dnorm;
        int no_trial;
        if (dnorm <= LM_EPS2) { stop = 1; no_trial = 1; }
        else { no_trial = 0; st->den = mu * dn2 + y2; }
        st->stop     = stop;
        st->no_trial = no_trial;
    }
}

template <int NX, int NU, int NSEG, bool ARROW = false>
__global__ __launch_bounds__(128 * NSEG)
__attribute__((amdgpu_waves_per_eu(2, 2)))   // 196 registers: two waves per SIMD (four segments: one workgroup per CU; two: two).  A 128-register build
                                             // (two 8-wave workgroups per CU) spills 290 - 340 bytes per lane into the dependent chain: 0.73 -> 0.88 ms per factor launch group at cfg 5, measured;
                                             // six segments (12 waves at 168 registers, 68 bytes of scratch): 0.698 -> 0.723 ms per group, one OCP 0.149 -> 0.158 -- measured, not kept
void big_chain3_kernel(const FactorParams p)
{
    extern __shared__ __attribute__((aligned(16))) double sm3[];
    LmState* sl = reinterpret_cast<LmState*>(sm3 + ((Chain3Lds<NX, NU, NSEG, ARROW>::total(p.N) + 1) & ~1));
    const int inst = blockIdx.x + p.inst0;
    lm_state_in(sl, p.st + inst, threadIdx.x);
    __syncthreads();
    if (sl->done) return;
    big_chain3_body<NX, NU, NSEG, ARROW>(p, sl, sm3, inst, nullptr);
    __syncthreads();
    lm_state_out(p.st + inst, sl, threadIdx.x);
}

// One Levenberg-Marquardt pass of every unfinished instance in ONE launch:  [sweep phase -> factor phase]  per workgroup.
//   sweep phase  : residual at the trial iterate, accept / reject, and on an accepted step the new Jacobian (mode 3); for the
//                  first launch of a solve the prologue instead (mode 2: residual + Jacobian at the start iterate);
//   factor phase : assemble, factor, solve -> next trial iterate.  When the sweep phase has just refreshed the Jacobian it is
//                  still in the LDS staging area, so accepted steps never re-read it from HBM (it is still streamed out once,
//                  for the passes that follow a rejected step).
// While some workgroups are in the throughput-bound sweep phase others are in the latency-bound factor phase, so the two overlap
// across the chip.
// LOOP = run-to-completion (default of corbo_hip_solve): the workgroup repeats the pass until its instance has finished, one
// launch per solve -- the instances are independent, so there is no grid-wide meeting point, no launch gap between passes, and
// the slow instances of the tail run at single-instance latency (0.81 ms instead of 1.09 ms per headline solve).  The loop needs
// two precautions against the compiler carrying state around it: the kernel arguments are re-read from the kernarg segment each
// pass, and the library is built with -disable-machine-licm (hoisted math-library constants were spilled to scratch otherwise).
// row of this wave's CU in the per-CU progress table (16 ints): [0] workgroups present, [2..3] = eight BYTES, their outer iteration + 1 (0 = empty slot)
__device__ __forceinline__ int32_t* cu_row_of(int32_t* table)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // HW_ID (gfx9): cu_id[11:8] sh_id[12] se_id[15:13]
    const unsigned idx = ((xcc & 7u) << 8) | ((hw >> 8) & 0xFFu);
    return table + (size_t)idx * 16;
}

#ifndef CORBO_HIP_PASS_WAVES
#define CORBO_HIP_PASS_WAVES 4   // waves per SIMD the fused kernel is compiled for: 4 = 128 VGPRs (44 spilled), four workgroups per CU.  -DCORBO_HIP_PASS_WAVES=3:
#endif                           // 168 VGPRs, no spills, three workgroups per CU (diagnostics: attribution of the scratch traffic, profiles/r03_spill_attribution.json)
// THREADS: 256 (four waves, 128 VGPRs at four workgroups per CU), 192 (three waves: 168 VGPRs at the same four workgroups per CU, i.e. the
// same 1024 resident instances without the register spills of the 128-VGPR build -- VERDICT r3 item 1a) or 128 (two waves, 256 VGPRs).
// DENSE (round 6): the instantiation for non-diagonal weights (four-wave shape only; a separate instantiation, so the diagonal problems' kernels carry none of its
// registers) -- such handles ran every LM pass as two launches before.
template <int DYN, int DEFECT, bool ARROW, bool LOOP, int NPC, int THREADS = SWEEP_THREADS, bool DENSE = false>
__global__ __launch_bounds__(THREADS, (THREADS == SWEEP_THREADS ? CORBO_HIP_PASS_WAVES : THREADS / 64)) void lm_pass_kernel(const FactorParams fp, const SweepParams sp)
{
    static_assert(!DENSE || THREADS == SWEEP_THREADS || THREADS == 128, "non-diagonal weights: the four-wave shape, or two waves WITHOUT the stage-centric component pass (its lean staging and re-evaluated rows are the diagonal problem's)");
    constexpr int BK_LANE = (THREADS > 128) ? 128 : THREADS - 1;   // the lane that keeps the CU's progress row (a spare lane of wave 2 / the last lane)
    using Dy = Dynamics<DYN>;
    using FL = FactorLds<Dy::NX, Dy::NU>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    int inst      = blockIdx.x + fp.inst0;
    const int NP = NPC > 0 ? NPC : (fp.N | 1);
    // LDS map: Jacobian staging [0, nnz_pad) (= where the factor phase expects it), dynamics caches right behind it, vertex values
    // behind the factor carve, then the LM state and the sweep phase's reduction scratch + flags.  (The scratch must not live inside
    // the factor carve: a Jacobian with many bound and inequality entries is longer than the carve's prefix in front of it.)
    const int ftot = FL::total(NP, ARROW);
    double* jst = smem;
    double* cs  = smem + sp.nnz_pad;
    double* xs  = smem + ((ftot > sp.nnz_pad + fp.N * Dy::NC ? ftot : sp.nnz_pad + fp.N * Dy::NC) + 1) / 2 * 2;
    LmState* sl = reinterpret_cast<LmState*>(xs + sp.nvs);   // LM state of the instance, resident in LDS (nvs is even: 16-byte aligned)
    double* red = reinterpret_cast<double*>(sl + 1);         // [12]
    int* flags  = reinterpret_cast<int*>(red + 8);
    if constexpr (!LOOP) {
        lm_state_in(sl, fp.st + inst, tid);
        if (tid == 0) flags[0] = 0;
        __syncthreads();
        if (sp.mode == 3 && sl->done) return;
        sweep_body<DYN, DEFECT, true, DENSE, false, THREADS>(sp, sp.mode, sp.active_count, sl, xs, red, cs, jst, inst, tid);
        __threadfence_block();  // this workgroup's residual / iterate stores are visible to its factor phase
        __syncthreads();
        if (!sl->done) {
            factor_body<Dy::NX, Dy::NU, THREADS, ARROW, NPC, DENSE>(fp, sl, smem, inst, tid, flags[0] != 0);
            __syncthreads();
        }
        lm_state_out(fp.st + inst, sl, tid);
    }
    else {
        // run-to-completion: the instances are independent, so the workgroup walks its instance through the prologue and every LM
        // pass without leaving the chip (no launch gaps, workgroups drift apart so that latency-bound factor phases overlap
        // throughput-bound sweep phases of their neighbours).
        // Batches larger than what the chip holds at once (4 workgroups per CU): the launch has as many workgroups as fit, and every
        // workgroup pulls instance after instance from a ticket counter -- a finished instance's slot is refilled at once instead of
        // idling until the slowest instance of its round is through (instances need 10 .. 23+ passes).
        // the kernel arguments are re-read from the kernarg segment in every pass (scalar loads) instead of being kept live in
        // ~200 SGPRs around the loop
        struct Args { FactorParams f; SweepParams s; };
        typedef const __attribute__((address_space(4))) Args* ArgsPtr;
        ArgsPtr ka = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
        int tid_v = tid;
        if (fp.stagger == -1 || (fp.cu_table && fp.batch <= 1024)) {   // before the first table-based priority: by dispatch generation, youngest first (age favours the oldest)
            switch (blockIdx.x >> 8) {
                case 3: __builtin_amdgcn_s_setprio(3); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                default: __builtin_amdgcn_s_setprio(0); break;
            }
        }
        if (fp.stagger > 0) {   // diagnostics: phase-shift the workgroups that share a CU
            const long long t0 = clock64(), d = (long long)(blockIdx.x >> 8) * fp.stagger;
            while (clock64() - t0 < d) __builtin_amdgcn_s_sleep(16);
        }
#pragma nounroll
        for (;;) {
            // (queue mode is a launch parameter, not an instantiation: one uniform branch per instance; the library carries half as many kernels)
            const bool queue_mode = ((const FactorParams&)ka->f).queue != nullptr;
            if (queue_mode) {
                asm volatile("" : "+v"(tid_v), "+s"(ka) : : "memory");   // (as in the pass loop: nothing is carried around this loop either)
                const FactorParams& fq = (const FactorParams&)ka->f;
                __syncthreads();   // the previous instance's last readers of flags[] are through
                if (tid_v == 0) flags[2] = atomicAdd(fq.queue, 1);
                __syncthreads();
                const int ticket = __builtin_amdgcn_readfirstlane(flags[2]);   // (uniform: into a scalar register)
                if (ticket >= fq.batch) break;
                inst = ticket + fq.inst0;
            }
            int inst_v = inst;
            {
                const FactorParams& fq = (const FactorParams&)ka->f;
                lm_state_in(sl, fq.st + inst_v, tid_v);
            }
            if (tid_v == 0) { flags[0] = 0; flags[3] = -1; }
            __syncthreads();
            int mode = ((const SweepParams&)ka->s).mode;
            const int max_passes = ((const FactorParams&)ka->f).loop_passes;
            StageKeep<Dy::NX, Dy::NU> keep;   // (two-wave shape) the residual rows paired with the resident Jacobian, in registers across the passes
            int smask = 0;                    // (two-wave shape) what the component tables say about this lane's stage, decoded once per solve (factor_body)
#pragma nounroll
            for (int pass = 0; pass <= max_passes; ++pass) {
                asm volatile("" : "+s"(inst_v), "+v"(tid_v), "+s"(ka) : : "memory");  // nothing derived from them is carried around the loop
                const FactorParams& fpl = (const FactorParams&)ka->f;
                const SweepParams& spl  = (const SweepParams&)ka->s;
                const bool stamp = fpl.pass_timeline && inst_v == fpl.pass_timeline_inst && tid_v == 0 && pass < 64;
                if (stamp) fpl.pass_timeline[2 * pass] = clock64();
                const bool pcyc = fpl.phase_cycles && tid_v == 0;   // (diagnostics: per-instance phase totals, corbo_hip_get_phase_cycles)
                long long pc_t0 = 0;
                if (pcyc) pc_t0 = clock64();
                if (pass > 0) {
                    // (no barrier: lane 0 itself is the next writer of flags[0] -- the sweep phase's decision -- and every reader sits behind that
                    //  phase's barriers; the previous pass's readers are through, the pass ended with a workgroup barrier)
                    if (tid_v == 0) flags[0] = 0;
                    if (fpl.cu_table) {
                        if (stamp && pass < 32) fpl.pass_timeline[64 + pass] = flags[2] + 10 * flags[3];   // (diagnostics: valid for < 32 passes)
                        switch (__builtin_amdgcn_readfirstlane(flags[2])) {
                            case 3: __builtin_amdgcn_s_setprio(3); break;
                            case 2: __builtin_amdgcn_s_setprio(2); break;
                            case 1: __builtin_amdgcn_s_setprio(1); break;
                            default: __builtin_amdgcn_s_setprio(0); break;
                        }
                    }
                }
                constexpr bool TWOW = (THREADS <= 128) && !DENSE;   // two-wave shape: Jacobian stream-out and bookkeeping ride inside the factor phase (factor_body, after_gather); non-diagonal weights: the sweep phase stores and streams everything itself
                int scv[2 * Dy::NX + Dy::NU + 1];   // (two-wave shape) Jacobian offsets of lane k's defect edge: requested by the sweep phase, used by this pass's factor phase as well
                sweep_body<DYN, DEFECT, true, DENSE, false, THREADS>(spl, mode, spl.active_count, sl, xs, red, cs, jst, inst_v, tid_v, pass > 0, &keep, TWOW && max_passes > 0, TWOW ? scv : nullptr);   // (active_count: per-pass launches only)
                __threadfence_block();
                __syncthreads();
                if (stamp) fpl.pass_timeline[2 * pass + 1] = clock64();
                if (pcyc) {
                    const long long t1 = clock64();
                    long long* row = fpl.phase_cycles + (size_t)inst_v * 8;
                    const int w = flags[0] != 0 ? 0 : 1;
                    row[w] += t1 - pc_t0; row[3 + w] += 1;
                    pc_t0 = t1;
                }
                if (sl->done) break;
                // lag-based issue priority: the SIMD arbiter prefers older waves, so the workgroups dispatched last to a CU run every
                // pass ~40 % slower than the first ones while the CU is full -- and the launch ends with the slowest chain.  Every
                // workgroup publishes its outer iteration in a per-CU row; the one that is furthest behind gets the highest user
                // priority (s_setprio beats age).  One lane does the bookkeeping: it publishes and REQUESTS the row before the factor phase
                // and ranks the workgroup after it -- the memory round trip rides under the factor phase instead of holding every wave
                // of the workgroup at the phase's first barrier (measured: 2 k of the 4.3 k cycles between the two phases).
                // (four-wave shape: ranked at once -- eleven more live registers across the factor phase cost 26 more spills at its 128-VGPR budget)
                constexpr bool BK_DEFER = THREADS < SWEEP_THREADS;
                int bk_slot = -1, bk_k = 0;
                unsigned long long bk_word = 0;
                auto bk_rank = [&] {
                    int rank = 0, bk_cnt = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) bk_cnt += ((bk_word >> (8 * q)) & 0xFFull) ? 1 : 0;
                    const int rot = (bk_slot > 3 || bk_cnt > 4) ? 7 : 3;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int kq = (int)((bk_word >> (8 * q)) & 0xFFull) - 1;   // -1: empty slot
                        if (q != bk_slot && kq >= 0 && (kq < bk_k || (kq == bk_k && ((pass - q) & rot) < ((pass - bk_slot) & rot)))) ++rank;   // ties: rotating order (the arbiter's own tie-break is age)
                    }
                    flags[2] = rank > 3 ? 0 : 3 - rank;
                };
                auto bk_publish = [&] {
                    // (measured and not kept: the bookkeeping in every other pass only -- 0.463 ms either way)
                    if (fpl.cu_table && tid_v == BK_LANE) {
                        int32_t* row = cu_row_of(fpl.cu_table);
                        bk_slot = flags[3];
                        if (bk_slot < 0) { bk_slot = atomicAdd(row, 1) & 7; flags[3] = bk_slot; }
                        bk_k = sl->k;
                        // the eight slots of the row are BYTES of one 64-bit word (outer iteration + 1 <= 255; 0 = empty): one byte store and one 8-byte load per
                        // pass instead of one store and nine loads -- every one of them a trip to the L2 that the lane's later loads queue up behind
                        __hip_atomic_store(reinterpret_cast<unsigned char*>(row + 2) + bk_slot, (unsigned char)((bk_k + 1) & 0xFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        bk_word = __hip_atomic_load(reinterpret_cast<unsigned long long*>(row + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if constexpr (!BK_DEFER) bk_rank();
                    }
                };
                if constexpr (!TWOW) bk_publish();
                const bool j_fresh = flags[0] != 0;
                auto hook = [&] {   // (two-wave shape: behind the factor phase's own loads)
                    if constexpr (TWOW) {
                        if (max_passes > 0 && j_fresh) stream_jacobian_range<THREADS>(spl, jst, inst_v, tid_v);
                        bk_publish();
                    }
                };
                // (loop_passes = 0: ONE pass per launch -- the per-pass mode of corbo_hip_solve and the profiling mode; the trial iterate then
                //  goes to HBM for the next launch instead of staying in the LDS array the next sweep phase evaluates it from)
                factor_body<Dy::NX, Dy::NU, THREADS, ARROW, NPC, DENSE, false, TWOW>(fpl, sl, smem, inst_v, tid_v, j_fresh, max_passes > 0 ? xs : nullptr, &spl, &keep, max_passes > 0 || j_fresh, hook, TWOW ? scv : nullptr, &smask);
                if constexpr (BK_DEFER) { if (bk_slot >= 0) bk_rank(); }
                __threadfence_block();
                __syncthreads();
                if (pcyc) { long long* row = fpl.phase_cycles + (size_t)inst_v * 8; row[2] += clock64() - pc_t0; row[5] += 1; }
                if (stamp && fpl.timeline) {   // diagnostics: the phase stamps of this pass (factor [0,8), sweep [8,18)) into the per-pass log
                    long long* lg = fpl.pass_timeline + 150 + 18 * pass;
                    for (int q = 0; q < 18; ++q) { lg[q] = fpl.timeline[q]; fpl.timeline[q] = 0; }
                }
                mode = 3;
            }
            asm volatile("" : "+s"(inst_v), "+v"(tid_v), "+s"(ka) : : "memory");
            const FactorParams& fe = (const FactorParams&)ka->f;
            if (fe.cu_table && tid_v == BK_LANE && flags[3] >= 0) {   // leave the CU's progress row (the table is all zeros again when the launch retires)
                int32_t* row = cu_row_of(fe.cu_table);
                __hip_atomic_store(reinterpret_cast<unsigned char*>(row + 2) + flags[3], (unsigned char)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicSub(row, 1);
            }
            lm_state_out(fe.st + inst_v, sl, tid_v);   // the host reads status and counters from HBM
            if (fe.x_host) {
                // result sink: the finished instance's accepted iterate and LM state go straight to host-visible memory (posted PCIe
                // writes, overlapped with the instances that are still iterating) -- no copy-engine pass after the launch
                const double2* src = reinterpret_cast<const double2*>(fe.x + (size_t)inst_v * fe.nvs);
                double2* dst       = reinterpret_cast<double2*>(fe.x_host + (size_t)inst_v * fe.nvs);
                for (int i = tid_v; i < fe.nvs / 2; i += THREADS) dst[i] = src[i];
                lm_state_out(fe.st_host + inst_v, sl, tid_v);
            }
            if (tid_v == 0 && !sl->done && fe.unfinished_flag) *(volatile int32_t*)fe.unfinished_flag = 1;  // pass limit hit
            asm volatile("" : "+s"(ka) : : "memory");
            if (((const FactorParams&)ka->f).queue == nullptr) break;
        }
    }
}

#include "bt_factor.hpp"
#ifndef CORBO_HIP_BT_WAVES
#define CORBO_HIP_BT_WAVES 3   // waves per SIMD lm_bt_kernel is compiled for: 3 = 168 VGPRs, three workgroups per CU where the LDS permits (the headline structure with a rate
                               // limit: 52.6 KB each); 2 = 256 VGPRs.  Measured, ms per solve at batch 512 / 768 / 1024 / 2048: 1.00 / 1.15 / 1.83 / 2.91 against 0.96 / 1.67 / 1.80 / 3.45
#endif

// Run-to-completion kernel of the block-tridiagonal route (small-block families with extra edges: DESIGN.md 3.5d): lm_pass_kernel's loop -- prologue
// sweep, then [factor phase -> trial sweep phase] until the instance has finished its outer iterations -- with the sweep phase's XE instantiation (one
// lane per extra edge) and bt_factor_body.  Four waves per instance; the LDS carve holds the operands [J | values] of the assembly and, overlaid, the
// block storage of the cyclic reduction.  Queue mode, re-armed start, result sink and the pass limit as in lm_pass_kernel.
// (the operands [J | values | 0] of the assembly may run on into the vertex array behind the carve: the trial iterate is dead while the factor phase
//  assembles -- it is what that phase writes last -- so the carve only has to hold what does not fit there: three workgroups per CU for the headline structure)
template <int NX, int NU, bool ARROW>
__host__ __device__ constexpr int bt_carve_doubles(int N, int nnz_pad, int m_pad, int nc, int nvs)
{
    int c = BtLayout<NX + NU, ARROW>::carve(N);
    if (c < nnz_pad + m_pad + 2 - nvs) c = nnz_pad + m_pad + 2 - nvs;
    if (c < nnz_pad + N * nc) c = nnz_pad + N * nc;
    return (c + 1) / 2 * 2;
}

// BIG: horizons of 129 .. 256 grid points (bt_factor_body<.., BIG>): the block storage alone is 113 KB for the unicycle at N = 256 -- one workgroup per CU, all its registers
// WAVES: waves per SIMD = workgroups per CU the instantiation is compiled for -- 3: 168 VGPRs (46 spilled), 768 resident instances; 2: 256 VGPRs, no spills, 512 resident
// instances, 6 - 9 % fewer cycles per instance.  launch_bt_t picks by the number of ROUNDS the batch needs (1024 = 2 x 512: two; 768, 1536, the instance queue: three).
template <int DYN, int DEFECT, bool ARROW, bool BIG = false, int WAVES = CORBO_HIP_BT_WAVES>
__global__ __launch_bounds__(BT_THREADS, (BIG ? 1 : WAVES)) void lm_bt_kernel(const FactorParams fp, const SweepParams sp)
{
    using Dy = Dynamics<DYN>;
    constexpr int THREADS = BT_THREADS;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    int inst      = blockIdx.x + fp.inst0;
    double* jst = smem;
    double* cs  = smem + sp.nnz_pad;
    double* xs  = smem + bt_carve_doubles<Dy::NX, Dy::NU, ARROW>(fp.N, sp.nnz_pad, sp.m_pad, Dy::NC, sp.nvs);
    LmState* sl = reinterpret_cast<LmState*>(xs + sp.nvs);
    double* red = reinterpret_cast<double*>(sl + 1);   // [12]: reductions [0, 8), the sweep phase's flag words behind them
    int* flags  = reinterpret_cast<int*>(red + 8);
    struct Args { FactorParams f; SweepParams s; };
    typedef const __attribute__((address_space(4))) Args* ArgsPtr;
    ArgsPtr ka = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    int tid_v = tid;
#pragma nounroll
    for (;;) {
        const bool queue_mode = ((const FactorParams&)ka->f).queue != nullptr;
        if (queue_mode) {
            asm volatile("" : "+v"(tid_v), "+s"(ka) : : "memory");
            const FactorParams& fq = (const FactorParams&)ka->f;
            __syncthreads();
            if (tid_v == 0) flags[2] = atomicAdd(fq.queue, 1);
            __syncthreads();
            const int ticket = __builtin_amdgcn_readfirstlane(flags[2]);
            if (ticket >= fq.batch) break;
            inst = ticket + fq.inst0;
        }
        int inst_v = inst;
        {
            const FactorParams& fq = (const FactorParams&)ka->f;
            lm_state_in(sl, fq.st + inst_v, tid_v);
        }
        if (tid_v == 0) flags[0] = 0;
        __syncthreads();
        int mode = ((const SweepParams&)ka->s).mode;
        const int max_passes = ((const FactorParams&)ka->f).loop_passes;
#pragma nounroll
        for (int pass = 0; pass <= max_passes; ++pass) {
            asm volatile("" : "+s"(inst_v), "+v"(tid_v), "+s"(ka) : : "memory");   // nothing derived from them is carried around the loop
            const FactorParams& fpl = (const FactorParams&)ka->f;
            const SweepParams& spl  = (const SweepParams&)ka->s;
            const bool stamp = fpl.pass_timeline && inst_v == fpl.pass_timeline_inst && tid_v == 0 && pass < 64;   // (diagnostics: option pass_timeline)
            if (stamp) fpl.pass_timeline[2 * pass] = clock64();
            const bool pcyc = fpl.phase_cycles && tid_v == 0;
            long long pc_t0 = 0;
            if (pcyc) pc_t0 = clock64();
            if (pass > 0 && tid_v == 0) flags[0] = 0;
            sweep_body<DYN, DEFECT, true, false, false, THREADS, true>(spl, mode, spl.active_count, sl, xs, red, cs, jst, inst_v, tid_v, pass > 0);
            __threadfence_block();
            __syncthreads();
            if (stamp) fpl.pass_timeline[2 * pass + 1] = clock64();
            if (pcyc) {
                const long long t1 = clock64();
                long long* row = fpl.phase_cycles + (size_t)inst_v * 8;
                const int w = flags[0] != 0 ? 0 : 1;
                row[w] += t1 - pc_t0; row[3 + w] += 1;
                pc_t0 = t1;
            }
            if (sl->done) break;
            const bool j_fresh = flags[0] != 0;
            bt_factor_body<Dy::NX + Dy::NU, Dy::NX, ARROW, THREADS, BIG>(fpl, sl, smem, xs, red, inst_v, tid_v, j_fresh, spl.eq_stride, spl.eq_defect_off);
            __threadfence_block();
            __syncthreads();
            if (pcyc) { long long* row = fpl.phase_cycles + (size_t)inst_v * 8; row[2] += clock64() - pc_t0; row[5] += 1; }
            if (stamp && fpl.timeline) {   // diagnostics: the phase stamps of this pass (factor [0,8), sweep [8,18)) into the per-pass log
                long long* lg = fpl.pass_timeline + 150 + 18 * pass;
                for (int q = 0; q < 18; ++q) { lg[q] = fpl.timeline[q]; fpl.timeline[q] = 0; }
            }
            mode = 3;
        }
        asm volatile("" : "+s"(inst_v), "+v"(tid_v), "+s"(ka) : : "memory");
        const FactorParams& fe = (const FactorParams&)ka->f;
        lm_state_out(fe.st + inst_v, sl, tid_v);
        if (fe.x_host) {
            const double2* src = reinterpret_cast<const double2*>(fe.x + (size_t)inst_v * fe.nvs);
            double2* dst       = reinterpret_cast<double2*>(fe.x_host + (size_t)inst_v * fe.nvs);
            for (int i = tid_v; i < fe.nvs / 2; i += THREADS) dst[i] = src[i];
            lm_state_out(fe.st_host + inst_v, sl, tid_v);
        }
        if (tid_v == 0 && !sl->done && fe.unfinished_flag) *(volatile int32_t*)fe.unfinished_flag = 1;  // pass limit hit
        asm volatile("" : "+s"(ka) : : "memory");
        if (((const FactorParams&)ka->f).queue == nullptr) break;
    }
}

template <int DYN, int DEFECT>
bool launch_bt_t(const FactorParams& fp, const SweepParams& sp, hipStream_t stream)
{
    using Dy = Dynamics<DYN>;
    if constexpr (DEFECT == DEFECT_SHOOTING_HIGH) return false;
    else {
        constexpr int NBM = BtLayout<Dy::NX + Dy::NU, false>::NB_MAX;
        if (!fp.bt_pairs || fp.N > 2 * NBM || fp.loop_passes <= 0) return false;
        const bool arrow = fp.dt_free != 0, big = fp.N > NBM;
        const size_t carve = arrow ? bt_carve_doubles<Dy::NX, Dy::NU, true>(fp.N, fp.nnz_pad, fp.m_pad, Dy::NC, fp.nvs) : bt_carve_doubles<Dy::NX, Dy::NU, false>(fp.N, fp.nnz_pad, fp.m_pad, Dy::NC, fp.nvs);
        const size_t lds = sizeof(double) * (carve + fp.nvs + 12) + sizeof(LmState);
        if (lds > (size_t)160 * 1024) return false;
        // (four waves per workgroup: workgroups per CU = waves per SIMD the kernel is compiled for, if the LDS holds as many)
        const int lds_per_cu = (int)((size_t)160 * 1024 / lds) < 1 ? 1 : (int)((size_t)160 * 1024 / lds);
        const int cus = fp.num_cus > 0 ? fp.num_cus : 256;
        const int res3 = (lds_per_cu < 3 ? lds_per_cu : 3) * cus, res2 = (lds_per_cu < 2 ? lds_per_cu : 2) * cus;   // resident instances of the two instantiations
        // two workgroups per CU whenever the batch needs no more rounds that way (<= 512, 769 .. 1024 on 256 CUs; whenever the LDS holds two at most) -- the register-rich
        // instantiation is 6 - 9 % faster per instance; three for the batches in between and for the instance queue (1.39 against 1.68 us per instance sustained)
        bool two = !fp.queue && ((fp.batch + res2 - 1) / res2 <= (fp.batch + res3 - 1) / res3);
        if (fp.bt_waves == 2) two = true;
        if (fp.bt_waves == 3) two = false;
        int grid = fp.batch;
        if (fp.queue) {
            int per_cu = lds_per_cu;
            const int cap = big ? 1 : (two ? 2 : 3);
            if (per_cu > cap) per_cu = cap;
            grid = fp.queue_grid * per_cu;
            if (grid > fp.batch) grid = fp.batch;
        }
        static unsigned long long attr_set[6] = {0, 0, 0, 0, 0, 0};   // (per device)
        auto go = [&](auto kernel, int slot) {
            if (first_on_device(attr_set[slot])) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(BT_THREADS), lds, stream, fp, sp);
        };
        if (big) { if (arrow) go(lm_bt_kernel<DYN, DEFECT, true, true, 1>, 3); else go(lm_bt_kernel<DYN, DEFECT, false, true, 1>, 2); }
        else if (two) { if (arrow) go(lm_bt_kernel<DYN, DEFECT, true, false, 2>, 5); else go(lm_bt_kernel<DYN, DEFECT, false, false, 2>, 4); }
        else { if (arrow) go(lm_bt_kernel<DYN, DEFECT, true, false, 3>, 1); else go(lm_bt_kernel<DYN, DEFECT, false, false, 3>, 0); }
        return true;
    }
}

template <int DYN, int DEFECT>
void launch_sweep_t(const SweepParams& p, hipStream_t stream)
{
    if constexpr (Dynamics<DYN>::NX <= 4) {
        if (p.N > LONG_HORIZON) {   // long horizon: Jacobian straight to HBM
            if constexpr (DEFECT != DEFECT_SHOOTING_HIGH) {
                if (p.n_xedges > 0) {   // ... with integral-form constraint edges / control-deviation edges (and non-diagonal weights: the band route reads any pattern)
                    if (p.mp.wdense) hipLaunchKernelGGL((sweep_kernel<DYN, DEFECT, true, true, true>), dim3(p.batch), dim3(SWEEP_THREADS), sweep_lds_bytes(p, Dynamics<DYN>::NC), stream, p);
                    else hipLaunchKernelGGL((sweep_kernel<DYN, DEFECT, false, true, true>), dim3(p.batch), dim3(SWEEP_THREADS), sweep_lds_bytes(p, Dynamics<DYN>::NC), stream, p);
                    return;
                }
            }
            if (p.mp.wdense) hipLaunchKernelGGL((sweep_kernel<DYN, DEFECT, true, true>), dim3(p.batch), dim3(SWEEP_THREADS), sweep_lds_bytes(p, Dynamics<DYN>::NC), stream, p);
            else hipLaunchKernelGGL((sweep_kernel<DYN, DEFECT, false, true>), dim3(p.batch), dim3(SWEEP_THREADS), sweep_lds_bytes(p, Dynamics<DYN>::NC), stream, p);
            return;
        }
        if constexpr (DEFECT != DEFECT_SHOOTING_HIGH) {
            if (p.n_xedges > 0) {   // integral-form constraint edges (finite-differences grids) / control-deviation edges (every grid): the XE instantiation
                if (p.mp.wdense) hipLaunchKernelGGL((sweep_kernel<DYN, DEFECT, true, false, true>), dim3(p.batch), dim3(SWEEP_THREADS), sweep_lds_bytes(p, Dynamics<DYN>::NC), stream, p);
                else hipLaunchKernelGGL((sweep_kernel<DYN, DEFECT, false, false, true>), dim3(p.batch), dim3(SWEEP_THREADS), sweep_lds_bytes(p, Dynamics<DYN>::NC), stream, p);
                return;
            }
        }
        if (p.mp.wdense) {   // non-diagonal weights: the DENSE instantiation
            hipLaunchKernelGGL((sweep_kernel<DYN, DEFECT, true>), dim3(p.batch), dim3(SWEEP_THREADS), sweep_lds_bytes(p, Dynamics<DYN>::NC), stream, p);
            return;
        }
    }
    if constexpr (Dynamics<DYN>::NX > 4) {
        if (p.n_xedges > 0) {   // big-block family with extra edges (the band factorisation reads the stored Jacobian): the XE instantiation
            hipLaunchKernelGGL((sweep_kernel<DYN, DEFECT, false, false, true>), dim3(p.batch), dim3(SWEEP_THREADS), sweep_lds_bytes(p, Dynamics<DYN>::NC), stream, p);
            return;
        }
        if (p.mode == 0 || (p.skip_jac && p.mode >= 2)) {   // no Jacobian due in this launch, whatever the decision: the residual-only instantiation
            hipLaunchKernelGGL((sweep_kernel<DYN, DEFECT, false, false, false, true>), dim3(p.batch), dim3(SWEEP_THREADS), sweep_lds_bytes(p, Dynamics<DYN>::NC), stream, p);
            return;
        }
    }
    hipLaunchKernelGGL((sweep_kernel<DYN, DEFECT>), dim3(p.batch), dim3(SWEEP_THREADS), sweep_lds_bytes(p, Dynamics<DYN>::NC), stream, p);
}

template <int DYN>
bool launch_sweep_d(int defect, const SweepParams& p, hipStream_t stream)
{
#ifdef CORBO_HIP_DEV_FAST   // development builds: only the headline defect formula is instantiated (compile time)
    if (defect != CORBO_HIP_DEFECT_CRANK_NICOLSON) return false;
    launch_sweep_t<DYN, CORBO_HIP_DEFECT_CRANK_NICOLSON>(p, stream);
    return true;
#else
    if (defect == CORBO_HIP_DEFECT_RK4_SHOOTING && (int)p.mp.dyn[7] >= 5) {   // Runge-Kutta 5 / 6 / 7: a defect formula of its own (model.hpp)
        if constexpr (Dynamics<DYN>::NX <= 4) { launch_sweep_t<DYN, DEFECT_SHOOTING_HIGH>(p, stream); return true; }
        else {
            // big-block family: the residual-only instantiation (the Jacobian of these handles is the stage kernel's; no band route with these integrators)
            if (p.n_xedges > 0 || !(p.mode == 0 || (p.skip_jac && p.mode >= 2))) return false;
            hipLaunchKernelGGL((sweep_kernel<DYN, DEFECT_SHOOTING_HIGH, false, false, false, true>), dim3(p.batch), dim3(SWEEP_THREADS), sweep_lds_bytes(p, Dynamics<DYN>::NC), stream, p);
            return true;
        }
    }
    switch (defect) {
        case CORBO_HIP_DEFECT_FORWARD: launch_sweep_t<DYN, CORBO_HIP_DEFECT_FORWARD>(p, stream); return true;
        case CORBO_HIP_DEFECT_BACKWARD: launch_sweep_t<DYN, CORBO_HIP_DEFECT_BACKWARD>(p, stream); return true;
        case CORBO_HIP_DEFECT_MIDPOINT: launch_sweep_t<DYN, CORBO_HIP_DEFECT_MIDPOINT>(p, stream); return true;
        case CORBO_HIP_DEFECT_CRANK_NICOLSON: launch_sweep_t<DYN, CORBO_HIP_DEFECT_CRANK_NICOLSON>(p, stream); return true;
        case CORBO_HIP_DEFECT_RK4_SHOOTING: launch_sweep_t<DYN, CORBO_HIP_DEFECT_RK4_SHOOTING>(p, stream); return true;
        default: return false;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Exact-Hessian path: computeSparseHessiansValues (hyper_graph_optimization_problem_edge_based.cpp:3491-3760)
// and the two-side-bounded linear form (:4904-4968, optimization_problem_interface.cpp:1141-1183) on the device.
// One lane per (instance, stage): the lane owns private copies of the vertices of its stage's edges -- x_k, u_k, x_{k+1}, dt -- and
// runs, edge by edge, the reference's own sequence on them: the central-difference Jacobian block of vertex i (BaseEdge::computeJacobian,
// edge_interface.cpp:55-96, delta = 1e-9, perturbed in place and reverted by a third addition), then for every component of vertex j a
// forward step of HESSIAN_DELTA = 1e-2 (:32), the Jacobian block again, and  (1 / delta) * multiplier_r * (J2 - J1)  summed over the
// edge's rows r (computeHessian / computeHessianInc, :151-255).  Least-squares cost edges contribute the Gauss-Newton block
// 2 m J_i^T J_j (:3566-3606).  Value layout and entry order: build_hessian_structure.  The only thing a lane cannot reproduce is the
// drift the reference's in-place perturbations leave in vertices SHARED with edges evaluated earlier (x_{k+1} of stage k is x_k of
// stage k+1): a few ulps of the point, i.e. 1e-7 in a Jacobian entry and 1e-5 in a Hessian entry -- the size of the difference
// between two consecutive calls of the reference itself (tests/test_gpu_hessian.py).
// ---------------------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
template <int DYN, int DEFECT>
struct HessEdge {
    using Dy = Dynamics<DYN>;
    static constexpr int NX = Dy::NX, NU = Dy::NU, S = NX + NU, W = S + NX + 1;
    static constexpr int MAXD = (NX > NU) ? NX : NU;
    static constexpr int MAXE = MAXD + 1;   // rows of the widest edge: the joint view of a mixed edge (1 objective value + NX equality values)
    // (round 4: state blocks up to 8 rows -- the planar quadrotor; the 12-state instantiation compiles but does not finish on the hardware: > 100 s for a
    // 4-interval problem, the same pathology as the four-call-site version of round 3 -- structure.cpp refuses it)
    static constexpr bool MIXED_OK = (DEFECT == CORBO_HIP_DEFECT_RK4_SHOOTING || DEFECT == DEFECT_SHOOTING_HIGH) && NX <= 8;
    // precompute() of MultipleShootingEdgeSingleControl (multiple_shooting_edges.h:214-229, 251-281): the grid's integrator on current = [cost; x]
    // with the integrand [c(x, u_k) against reference k; f(x, u_k)], c = QuadraticFormCost::computeIntegralStateControlTerm (quadratic_cost.cpp:
    // 186-230, diagonal weights in mp.sq / mp.sr).  The solveIVP overload for a generic integrand has the expressions of the one for the system
    // dynamics (explicit_integrators.h), so the state part is the defect edge's own integration.
    __device__ static void aug_rhs(const double* X, const double* u, const double* xr, const ModelParams& mp, double* F)
    {
        dyn_full<DYN>(X + 1, u, mp.dyn, F + 1);
        double cost = 0.0, acc = 0.0;
        { double t[NX]; for (int i = 0; i < NX; ++i) { const double xd = X[1 + i] - xr[i]; t[i] = (xd * mp.sq[i]) * xd; } acc = eigen_sum<NX>(t); }   // (Eigen's reduction order: model.hpp)
        cost += acc;
        { double t[NU]; for (int i = 0; i < NU; ++i) t[i] = (u[i] * mp.sr[i]) * u[i]; acc = eigen_sum<NU>(t); }
        cost += acc;
        F[0] = cost;
    }
    __device__ static void aug_end_state(const double* x1, const double* u1, double dt, const double* xr, const ModelParams& mp, double* xe)
    {
        constexpr int n = NX + 1;
        const int order = (int)mp.dyn[7];
        double t[n];
#define AUG_STAGE(K, EXPR)                               \
    {                                                    \
        for (int i = 0; i < n; ++i) t[i] = EXPR;         \
        aug_rhs(t, u1, xr, mp, K);                       \
        for (int i = 0; i < n; ++i) K[i] *= dt;          \
    }
        if constexpr (DEFECT == DEFECT_SHOOTING_HIGH) {   // Runge-Kutta 5 / 6 / 7 (explicit_integrators.h:371-394, 479-503, 600-628)
            double k1[n], k2[n], k3[n], k4[n], k5[n], k6[n], k7[n], k8[n];
            AUG_STAGE(k1, x1[i])
            if (order == 5) {
                const double s6 = 2.449489742783178;
                AUG_STAGE(k2, x1[i] + 4.0 * k1[i] / 11.0)
                AUG_STAGE(k3, x1[i] + (9.0 * k1[i] + 11.0 * k2[i]) / 50.0)
                AUG_STAGE(k4, x1[i] + (-11.0 * k2[i] + 15.0 * k3[i]) / 4.0)
                AUG_STAGE(k5, x1[i] + ((81.0 + 9.0 * s6) * k1[i] + (255.0 - 55.0 * s6) * k3[i] + (24.0 - 14.0 * s6) * k4[i]) / 600.0)
                AUG_STAGE(k6, x1[i] + ((81.0 - 9.0 * s6) * k1[i] + (255.0 + 55.0 * s6) * k3[i] + (24.0 + 14.0 * s6) * k4[i]) / 600.0)
                for (int i = 0; i < n; ++i) xe[i] = x1[i] + (4.0 * k1[i] + (16.0 + s6) * k5[i] + (16.0 - s6) * k6[i]) / 36.0;
            }
            else if (order == 6) {
                AUG_STAGE(k2, x1[i] + 2.0 * k1[i] / 33.0)
                AUG_STAGE(k3, x1[i] + 4.0 * k2[i] / 33.0)
                AUG_STAGE(k4, x1[i] + (k1[i] + 3.0 * k3[i]) / 22.0)
                AUG_STAGE(k5, x1[i] + (43.0 * k1[i] - 165.0 * k3[i] + 144.0 * k4[i]) / 64.0)
                AUG_STAGE(k6, x1[i] + (-4053483.0 * k1[i] + 16334703.0 * k3[i] - 12787632.0 * k4[i] + 1057536.0 * k5[i]) / 826686.0)
                AUG_STAGE(k7, x1[i] + (169364139.0 * k1[i] - 663893307.0 * k3[i] + 558275718.0 * k4[i] - 29964480.0 * k5[i] + 35395542.0 * k6[i]) / 80707214.0)
                AUG_STAGE(k8, x1[i] + (-733.0 * k1[i] + 3102.0 * k3[i]) / 176.0 - (335763.0 * k4[i] / 23296.0) + (216.0 * k5[i] / 77.0) - (4617.0 * k6[i] / 2816.0) + (7203.0 * k7[i] / 9152.0))
                for (int i = 0; i < n; ++i)
                    xe[i] = x1[i] + (336336.0 * k1[i] + 1771561.0 * k4[i] + 1916928.0 * k5[i] + 597051.0 * k6[i] + 1411788.0 * k7[i] + 256256.0 * k8[i]) / 6289920.0;
            }
            else {
                double k9[n], k10[n], k11[n];
                AUG_STAGE(k2, x1[i] + 2.0 * k1[i] / 27.0)
                AUG_STAGE(k3, x1[i] + (k1[i] + 3.0 * k2[i]) / 36.0)
                AUG_STAGE(k4, x1[i] + (k1[i] + 3.0 * k3[i]) / 24.0)
                AUG_STAGE(k5, x1[i] + (80.0 * k1[i] - 300.0 * k3[i] + 300.0 * k4[i]) / 192.0)
                AUG_STAGE(k6, x1[i] + (k1[i] + 5.0 * k4[i] + 4.0 * k5[i]) / 20.0)
                AUG_STAGE(k7, x1[i] + (-25.0 * k1[i] + 125.0 * k4[i] - 260.0 * k5[i] + 250.0 * k6[i]) / 108.0)
                AUG_STAGE(k8, x1[i] + (93.0 * k1[i] + 244.0 * k5[i] - 200.0 * k6[i] + 13.0 * k7[i]) / 900.0)
                AUG_STAGE(k9, x1[i] + (12.0 * k1[i] - 53.0 * k4[i]) / 6.0 + (1408.0 * k5[i] - 1070.0 * k6[i] + 67.0 * k7[i] + 270.0 * k8[i]) / 90.0)
                AUG_STAGE(k10, x1[i] + (-12285.0 * k1[i] + 3105.0 * k4[i] - 105408.0 * k5[i] + 83970.0 * k6[i] - 4617.0 * k7[i] + 41310.0 * k8[i] - 1215.0 * k9[i]) / 14580.0)
                AUG_STAGE(k11, x1[i] + (2383.0 * k1[i] - 8525.0 * k4[i] + 17984.0 * k5[i] - 15050.0 * k6[i] + 2133.0 * k7[i] + 2250.0 * k8[i] + 1125.0 * k9[i] + 1800.0 * k10[i]) / 4100.0)
                for (int i = 0; i < n; ++i)
                    xe[i] = x1[i] + (41.0 * k1[i] + 272.0 * k6[i] + 216.0 * k7[i] + 216.0 * k8[i] + 27.0 * k9[i] + 27.0 * k10[i] + 41.0 * k11[i]) / 840.0;
            }
        }
        else {   // Euler (:66-72), Runge-Kutta 2 (:127-138), 3 (:200-213), 4 (:280-295)
            double k1[n], k2[n], k3[n], k4[n];
            AUG_STAGE(k1, x1[i])
            if (order == 1) { for (int i = 0; i < n; ++i) xe[i] = k1[i] + x1[i]; }
            else if (order == 2) {
                AUG_STAGE(k2, x1[i] + k1[i])
                for (int i = 0; i < n; ++i) xe[i] = x1[i] + (k1[i] + k2[i]) / 2.0;
            }
            else if (order == 3) {
                AUG_STAGE(k2, x1[i] + (k1[i] / 2.0))
                AUG_STAGE(k3, x1[i] - k1[i] + 2.0 * k2[i])
                for (int i = 0; i < n; ++i) xe[i] = x1[i] + (k1[i] + 4.0 * k2[i] + k3[i]) / 6.0;
            }
            else {
                AUG_STAGE(k2, x1[i] + k1[i] / 2.0)
                AUG_STAGE(k3, x1[i] + k2[i] / 2.0)
                AUG_STAGE(k4, x1[i] + k3[i])
                for (int i = 0; i < n; ++i) xe[i] = x1[i] + (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]) / 6.0;
            }
        }
#undef AUG_STAGE
    }
    // BaseEdge::computeValues of the edge kinds on the path, on the lane's private vertex copies (xl: x_k | u_k | x_{k+1} | dt)
    __device__ static void values(int kind, const double* xl, const double* xr, const ModelParams& mp, double* out)
    {
        switch (kind) {
            case EK_STATE_COST: case EK_FINAL_COST: {
                const int cls = (kind == EK_FINAL_COST) ? 2 : 0;
                if (mp.wdense && ((mp.wdense_mask >> cls) & 1)) {   // non-diagonal weight: U (x - ref), Eigen's gemv order (dense_weight_row)
                    double xd[MAXD];
                    for (int i = 0; i < MAXD; ++i) xd[i] = (i < NX) ? xl[i] - xr[i] : 0.0;
                    for (int i = 0; i < NX; ++i) out[i] = dense_weight_row<MAXD>(mp.wdense + 16 * cls, i, NX, xd);
                }
                else
                    for (int i = 0; i < NX; ++i) out[i] = (cls ? mp.sqf[i] : mp.sq[i]) * (xl[i] - xr[i]);
                break;
            }
            case EK_CONTROL_COST:
                if (mp.wdense && (mp.wdense_mask & 2)) {
                    double ud[MAXD];
                    for (int i = 0; i < MAXD; ++i) ud[i] = (i < NU) ? xl[NX + i] : 0.0;
                    for (int i = 0; i < NU; ++i) out[i] = dense_weight_row<MAXD>(mp.wdense + 16, i, NU, ud);
                }
                else
                    for (int i = 0; i < NU; ++i) out[i] = mp.sr[i] * xl[NX + i];
                break;
            case EK_DT_COST: case EK_DT_QCOST: out[0] = mp.dt_weight * xl[W - 1]; break;   // (plain form: dt_weight = N - 1)
            // plain objective edges, lsq_form = false (quadratic_cost.cpp:133-138,165-170, final_state_cost.cpp:102-108): xd^T * W_diag * xd, the
            // expression of TerminalBall; mp.sq / sr / sqf hold the weights themselves for such a descriptor
            case EK_STATE_QCOST: { double t[NX]; for (int i = 0; i < NX; ++i) { const double xd = xl[i] - xr[i]; t[i] = (xd * mp.sq[i]) * xd; } out[0] = 0.0 + eigen_sum<NX>(t); break; }
            case EK_CONTROL_QCOST: { double t[NU]; for (int i = 0; i < NU; ++i) t[i] = (xl[NX + i] * mp.sr[i]) * xl[NX + i]; out[0] = 0.0 + eigen_sum<NU>(t); break; }
            case EK_FINAL_QCOST: { double t[NX]; for (int i = 0; i < NX; ++i) { const double xd = xl[i] - xr[i]; t[i] = (xd * mp.sqf[i]) * xd; } out[0] = 0.0 + eigen_sum<NX>(t); break; }
            // QuadraticFormCost::computeIntegralStateControlTerm (quadratic_cost.cpp:186-230): cost = 0; cost += xd^T Q xd; cost += u^T R u, at
            // (x_k, u_k) and -- trapezoidal rule -- at (x_{k+1}, u_k), both against reference k; 0.5 dt (c1 + c2) resp. c1 *= dt
            case EK_INTEGRAL_TRAP: case EK_INTEGRAL_LEFT: {
                double c[2] = {0.0, 0.0};
                for (int end = 0; end < (kind == EK_INTEGRAL_TRAP ? 2 : 1); ++end) {
                    const double* xe = end ? xl + S : xl;
                    double cost = 0.0, acc = 0.0;
                    { double t[NX]; for (int i = 0; i < NX; ++i) { const double xd = xe[i] - xr[i]; t[i] = (xd * mp.sq[i]) * xd; } acc = eigen_sum<NX>(t); }
                    cost += acc;
                    { double t[NU]; for (int i = 0; i < NU; ++i) t[i] = (xl[NX + i] * mp.sr[i]) * xl[NX + i]; acc = eigen_sum<NU>(t); }
                    cost += acc;
                    c[end] = cost;
                }
                if (kind == EK_INTEGRAL_TRAP) out[0] = 0.5 * xl[W - 1] * (c[0] + c[1]);
                else { out[0] = c[0]; out[0] *= xl[W - 1]; }
                break;
            }
            case EK_DEFECT: defect_eval<DYN, DEFECT>(xl, xl + NX, xl + S, xl[W - 1], mp.dyn, out); break;
            // MultipleShootingEdgeSingleControl (multiple_shooting_edges.h:214-242): objective value = values[0], equality values = values[1 .. nx] - x_{k+1}
            // EK_MIXED_JOINT: both parts, [objective value; equality values] -- what BaseMixedEdge::computeJacobians differentiates in ONE perturbation cycle
            case EK_MIXED_OBJ: case EK_MIXED_EQ: case EK_MIXED_JOINT:
                if constexpr (MIXED_OK) {
                    double cur[NX + 1], val[NX + 1];
                    cur[0] = 0;
                    for (int i = 0; i < NX; ++i) cur[1 + i] = xl[i];
                    aug_end_state(cur, xl + NX, xl[W - 1], xr, mp, val);
                    if (kind != EK_MIXED_EQ) out[0] = val[0];
                    const int o = (kind == EK_MIXED_JOINT) ? 1 : 0;
                    if (kind != EK_MIXED_OBJ)
                        for (int i = 0; i < NX; ++i) out[o + i] = val[1 + i] - xl[S + i];
                }
                break;
            case EK_STAGE_INEQ:
                out[0] = stage_ineq_state<NX>(mp.ineq_id, xl, mp.ineq);
                break;
            case EK_FINAL_EQ:
                if (mp.fin_eq_mask) {   // TerminalPartialEqualityConstraint (final_state_constraints.h:236-252): the active components only, in order
                    // (static indices only: a running output index puts the lane's arrays into scratch memory -- measured on this kernel: 448 -> 1408 bytes
                    //  per lane, 273 -> 458 us for 1024 OCPs)
                    int idx = 0;
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        const bool a = ((mp.fin_eq_mask >> i) & 1) != 0;
                        const double v = xl[i] - xr[i];
#pragma unroll
                        for (int r = 0; r < NX; ++r) out[r] = (a && idx == r) ? v : out[r];
                        idx += a ? 1 : 0;
                    }
                }
                else
                    for (int i = 0; i < NX; ++i) out[i] = xl[i] - xr[i];
                break;
            case EK_FINAL_INEQ: out[0] = terminal_ball<NX>(xl, xr, mp.fin); break;
            default: break;
        }
    }
    __device__ static int edge_dim(int kind, const ModelParams& mp) { return (kind == EK_FINAL_EQ && mp.fin_eq_mask) ? __popc((unsigned)mp.fin_eq_mask) : kind == EK_MIXED_JOINT ? NX + 1 : kind == EK_MIXED_EQ ? NX : kind == EK_CONTROL_COST ? NU : (kind == EK_DT_COST || kind == EK_STAGE_INEQ || kind == EK_FINAL_INEQ || kind >= EK_STATE_QCOST) ? 1 : NX; }
    __device__ static int n_verts(int kind) { return (kind == EK_DEFECT || kind == EK_INTEGRAL_TRAP || kind >= EK_MIXED_OBJ) ? 4 : kind == EK_INTEGRAL_LEFT ? 3 : 1; }
    __device__ static int vert_off(int kind, int vi) { return (kind >= EK_MIXED_OBJ) ? (vi == 0 ? 0 : vi == 1 ? NX : vi == 2 ? W - 1 : S) : kind == EK_INTEGRAL_LEFT ? (vi == 0 ? 0 : vi == 1 ? NX : W - 1) : (kind == EK_DEFECT || kind == EK_INTEGRAL_TRAP) ? (vi == 0 ? 0 : vi == 1 ? NX : vi == 2 ? S : W - 1) : (kind == EK_CONTROL_COST || kind == EK_CONTROL_QCOST) ? NX : (kind == EK_DT_COST || kind == EK_DT_QCOST) ? W - 1 : 0; }
    __device__ static int vert_dim(int kind, int vi)
    {
        if (kind == EK_DEFECT || kind == EK_INTEGRAL_TRAP) return vi == 0 ? NX : vi == 1 ? NU : vi == 2 ? NX : 1;
        if (kind == EK_INTEGRAL_LEFT) return vi == 0 ? NX : vi == 1 ? NU : 1;
        if (kind >= EK_MIXED_OBJ) return vi == 0 ? NX : vi == 1 ? NU : vi == 2 ? 1 : NX;   // (x_k, u_k, dt, x_{k+1})
        return (kind == EK_CONTROL_COST || kind == EK_CONTROL_QCOST) ? NU : (kind == EK_DT_COST || kind == EK_DT_QCOST) ? 1 : NX;   // every other edge hangs on one state vertex
    }
    __device__ static int unfixed(unsigned fm, int off, int dim) { int n = 0; for (int i = 0; i < dim; ++i) n += ((fm >> (off + i)) & 1u) ? 0 : 1; return n; }
    // BaseEdge::computeJacobian (edge_interface.cpp:55-96): block [dim x n_unfixed], column-major
    __device__ static void jacobian(int kind, int vi, unsigned fm, double* xl, const double* xr, const ModelParams& mp, double* blk)
    {
        constexpr double delta = 1e-9, neg2delta = -2 * delta, scalar = 1.0 / (2 * delta);
        const int off = vert_off(kind, vi), dim = vert_dim(kind, vi), ed = edge_dim(kind, mp);
        double v1[MAXE], v2[MAXE];
        int col = 0;
        for (int i = 0; i < dim; ++i) {
            if ((fm >> (off + i)) & 1u) continue;
            xl[off + i] += delta;
            values(kind, xl, xr, mp, v2);
            xl[off + i] += neg2delta;
            values(kind, xl, xr, mp, v1);
            for (int r = 0; r < ed; ++r) blk[col * ed + r] = scalar * (v2[r] - v1[r]);
            xl[off + i] += delta;
            ++col;
        }
    }
    // all blocks of one edge, in the order of the reference's walk; returns the number of values written (per list).
    // cat 0: least-squares objective edge; 1 / 2: equality / inequality edge; 3: plain objective edge; 4: a MIXED edge with a plain objective part
    // and an equality part (kind EK_MIXED_JOINT; the last branch of the mixed loop, :3941-3988): BaseMixedEdge::computeJacobians once per vertex i
    // (ONE perturbation cycle for both parts, edge_interface.cpp:394-465), then per vertex j computeObjectiveHessian[Inc](.., nullptr, multiplier_obj)
    // into `out` and computeEqualityHessian[Inc](.., mult_eq_part) into `out2` (edge_interface.cpp:525-634) -- both lists advance by the same
    // blocks; the Jacobian at the perturbed point is the joint one again (the other part's rows are not read: same perturbation cycle, same values).
    __device__ static int hessian_blocks(int kind, int cat, bool lower, unsigned fm, double* xl, const double* xr, const ModelParams& mp, double mult_obj,
                                         const double* mult, double* out, double* out2, double* out3, int vi_only = -1, int vj_only = -1)
    {
        constexpr double hdelta = 1e-2;
        const int ed = edge_dim(kind, mp), nv = n_verts(kind);
        const int nparts = (cat == 4) ? 2 : 1;
        double jac1[MAXE * MAXD], jac2[MAXE * MAXD], blk[MAXD * MAXD];
        int at = 0;
        for (int vi = 0; vi < nv; ++vi) {
            const int oi = vert_off(kind, vi), di = vert_dim(kind, vi), ni = unfixed(fm, oi, di);
            if (ni == 0) continue;
            const int vend = lower ? vi + 1 : nv;
            const bool mine = (vi_only < 0 || vi == vi_only);   // (the kernel splits an edge's blocks over waves: by row vertex, or by block)
            if (mine) jacobian(kind, vi, fm, xl, xr, mp, jac1);
            for (int vj = 0; vj < vend; ++vj) {
                const int oj = vert_off(kind, vj), dj = vert_dim(kind, vj), nj = unfixed(fm, oj, dj);
                if (nj == 0) continue;
                const bool diag_lower = lower && vi == vj;
                if (!mine || (vj_only >= 0 && vj != vj_only)) { at += diag_lower ? ni * (ni + 1) / 2 : ni * nj; continue; }
                for (int part = 0; part < nparts; ++part) {
                    if (cat == 0) {   // least-squares objective edge: 2 m J_i^T J_j.  Eigen evaluates small products (rows + cols + depth < 20)
                        // coefficient-based with (2 m J_i^T) as the left factor -- every term scaled first -- and larger ones through its GEMM
                        // kernel, which scales the finished sum (the 12 x 12 state-cost block of the quadrotor)
                        jacobian(kind, vj, fm, xl, xr, mp, jac2);
                        const bool small = ni + nj + ed < 20;
                        for (int c = 0; c < nj; ++c)
                            for (int r = 0; r < ni; ++r) {
                                double acc = 0.0;
                                if (small) for (int q = 0; q < ed; ++q) acc += ((2.0 * mult_obj) * jac1[r * ed + q]) * jac2[c * ed + q];
                                else { for (int q = 0; q < ed; ++q) acc += jac1[r * ed + q] * jac2[c * ed + q]; acc = (2.0 * mult_obj) * acc; }
                                blk[c * ni + r] = acc;
                            }
                    }
                    else {   // BaseEdge::computeHessian[Inc] (edge_interface.cpp:151-255); cat 3: a plain objective edge, weighted with the objective
                        // multiplier instead of row multipliers (…edge_based.cpp:2363-2410); cat 4: rows [0, 1) like cat 3, rows [1, ed) like cat 1
                        const int r0 = (cat == 4 && part == 1) ? 1 : 0, r1 = (cat == 4 && part == 0) ? 1 : ed;
                        const double* pm = (cat == 4 && part == 0) ? nullptr : mult;
                        double scalar = 1.0 / hdelta;
                        if ((cat == 3 || (cat == 4 && part == 0)) && mult_obj != 1.0) scalar *= mult_obj;
                        for (int q = 0; q < ni * nj; ++q) blk[q] = 0.0;
                        int cj = 0;
                        for (int j = 0; j < dj; ++j) {
                            if ((fm >> (oj + j)) & 1u) continue;
                            xl[oj + j] += hdelta;
                            jacobian(kind, vi, fm, xl, xr, mp, jac2);
                            for (int r = r0; r < r1; ++r) {
                                const double f = pm ? scalar * pm[r - r0] : scalar;
                                for (int c = 0; c < ni; ++c) {
                                    const double t = f * (jac2[c * ed + r] - jac1[c * ed + r]);
                                    if (r == r0 && diag_lower) blk[cj * ni + c] = t;
                                    else blk[cj * ni + c] += t;
                                }
                            }
                            xl[oj + j] += -hdelta;
                            ++cj;
                        }
                    }
                    double* o = (part ? out2 : out) + at;
                    if (diag_lower) {
                        int w = 0;
                        for (int i = 0; i < ni; ++i)
                            for (int j = 0; j <= i; ++j) o[w++] = 0.0 + blk[j * ni + i];
                    }
                    else
                        for (int q = 0; q < ni * nj; ++q) o[q] = 0.0 + blk[q];
                }
                if (out3)   // (mixed edge in a problem that has inequalities: its blocks exist in the inequality list as well -- the reference never writes them)
                    for (int q = 0; q < (diag_lower ? ni * (ni + 1) / 2 : ni * nj); ++q) out3[at + q] = 0.0;
                at += diag_lower ? ni * (ni + 1) / 2 : ni * nj;
            }
        }
        return at;
    }
    // the unweighted Jacobian blocks of a constraint edge in the order of computeSparseJacobianTwoSideBoundedLinearFormValues (:4904-4968)
    __device__ static void linear_blocks(int kind, unsigned fm, double* xl, const double* xr, const ModelParams& mp, double* out)
    {
        const int ed = edge_dim(kind, mp), nv = n_verts(kind);
        double blk[MAXD * MAXD];
        int at = 0;
        for (int vi = 0; vi < nv; ++vi) {
            const int ni = unfixed(fm, vert_off(kind, vi), vert_dim(kind, vi));
            if (ni == 0) continue;
            jacobian(kind, vi, fm, xl, xr, mp, blk);
            for (int q = 0; q < ni * ed; ++q) out[at + q] = blk[q];
            at += ni * ed;
        }
    }
};

template <int DYN, int DEFECT>
__global__ __launch_bounds__(64) void hessian_kernel(const SweepParams p, const HessParams hp)
{
    using HE = HessEdge<DYN, DEFECT>;
    constexpr int NX = HE::NX, NU = HE::NU, S = HE::S, W = HE::W;
    const int k = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y, inst = b + p.inst0;
    if (k >= p.N) return;
    const bool final_stage = (k == p.N - 1);
    const double* xg = p.x + (size_t)inst * p.nvs;
    double xl[W], xr[NX];
    unsigned fm = 0;
    for (int i = 0; i < W - 1; ++i) {
        const int v = k * S + i;
        const bool in = final_stage ? (i < NX) : true;
        xl[i] = in ? xg[v] : 0.0;
        if (!in || p.comp[v].fixed) fm |= 1u << i;
    }
    xl[W - 1] = p.dt_free ? xg[p.off_dt] : p.dt_fixed;
    if (!p.dt_free) fm |= 1u << (W - 1);
    for (int i = 0; i < NX; ++i) xr[i] = p.refvec ? p.refvec[(size_t)inst * p.nvs + k * S + i] : p.xref[(size_t)inst * CORBO_HIP_MAX_NX + i];
    const int32_t* so = hp.stage_off + (size_t)k * 6;
    ModelParams mpl = p.mp;   // the instance's own parameters of the dynamics (corbo_hip_set_instance_params)
    if (p.dyn_inst)
        for (int i = 0; i < 8; ++i) mpl.dyn[i] = p.dyn_inst[(size_t)inst * 8 + i];
    if (hp.mode == 0) {
        double* vo = hp.vals[0] + (size_t)b * hp.nnz[0];
        double* ve = hp.vals[1] + (size_t)b * hp.nnz[1];
        double* vi = hp.vals[2] + (size_t)b * hp.nnz[2];
        const double* me = (hp.mult_eq && so[4] >= 0) ? hp.mult_eq + (size_t)b * hp.eq_dim + so[4] : nullptr;
        const double* mi = (hp.mult_ineq && so[5] >= 0) ? hp.mult_ineq + (size_t)b * hp.ineq_dim + so[5] : nullptr;
        const bool lower = hp.lower != 0;
        // The stage's edges in the order of the reference's walk (the three lists one after the other: objective edges first, then the
        // equality edges, then the inequality edges) -- ONE call site in a loop.  (With four inlined call sites the 12-state instantiation
        // never finished on the hardware -- > 120 s for one interval; the compiler had merged the copies into an exec-mask dispatch loop.)
        int kinds[6], cats[6], n_edges = 0;
        double* outs[6];
        double* outs2[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // (mixed edge: its second list)
        double* outs3[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // (mixed edge: its zero blocks in the inequality list)
        const double* mults[6];
        auto add = [&](int kind, int cat, double* out, const double* mult) { kinds[n_edges] = kind; cats[n_edges] = cat; outs[n_edges] = out; mults[n_edges] = mult; ++n_edges; };
        const bool nl = hp.cost_nonlsq != 0;   // plain objective edges: category 3 (same output list)
        const int stage_kind = hp.cost_integral == 1 ? EK_INTEGRAL_TRAP : hp.cost_integral == 2 ? EK_INTEGRAL_LEFT : (nl ? EK_STATE_QCOST : EK_STATE_COST);
        if (hp.ms_mixed && !final_stage) { add(EK_MIXED_JOINT, 4, vo + so[0], me); outs2[0] = ve + so[2]; if (so[3] >= 0) outs3[0] = vi + so[3]; }   // shooting grid + integral-form cost: the interval's only edge is the mixed one
        else {
        if (so[0] >= 0) add(final_stage ? (nl ? EK_FINAL_QCOST : EK_FINAL_COST) : stage_kind, nl ? 3 : 0, vo + so[0], nullptr);
        if (so[1] >= 0) add(nl ? EK_CONTROL_QCOST : EK_CONTROL_COST, nl ? 3 : 0, vo + so[1], nullptr);
        if (k == 0 && hp.dt_cost_off >= 0) { add(nl ? EK_DT_QCOST : EK_DT_COST, nl ? 3 : 0, vo + hp.dt_cost_off, nullptr); add(nl ? EK_DT_QCOST : EK_DT_COST, nl ? 3 : 0, nullptr, nullptr); }
        if (so[2] >= 0) add(final_stage ? EK_FINAL_EQ : EK_DEFECT, 1, ve + so[2], me);
        if (so[3] >= 0) add(final_stage ? EK_FINAL_INEQ : EK_STAGE_INEQ, 2, vi + so[3], mi);
        }
        // Work split (HessParams::split, uniform per wave: blockIdx.z): 0 = this lane walks all edges of its stage (round 3: two waves per
        // instance at N = 100 and a dependent chain of 35 k instructions per lane, whatever the batch); 1 = one wave per (edge, row vertex i):
        // the base Jacobian J_i once, then its blocks (i, j); 2 = one wave per block (i, j) (J_i recomputed per block: + 30 % evaluations, a
        // quarter of the depth).  The lanes of a wave are 64 stages of the SAME edge and vertex: no divergence.  The host picks by batch size.
        const int gz = blockIdx.z;
        const int ge = hp.split ? gz / (hp.split == 2 ? 16 : 4) : -1, gvi = hp.split ? (hp.split == 2 ? (gz / 4) % 4 : gz % 4) : -1, gvj = (hp.split == 2) ? gz % 4 : -1;
        double* next = nullptr;
        for (int e = 0; e < n_edges; ++e) {
            double* out = outs[e] ? outs[e] : next;   // (the duplicated dt edge follows the first one)
            if (ge >= 0 && e != ge) {   // another wave's edge: only the second dt edge needs to know where the first one ends (one 1 x 1 block)
                next = out + (((fm >> (W - 1)) & 1u) ? 0 : 1);
                continue;
            }
            const int n = HE::hessian_blocks(kinds[e], cats[e], lower, fm, xl, xr, mpl, hp.mult_obj, mults[e], out, outs2[e], outs3[e], gvi, gvj);
            next = out + n;
        }
    }
    else if (hp.mode == 2) {
        // computeGradientObjective: per least-squares edge the Jacobian block, then the values, gradient += (2 values^T) J; the stage's
        // share of computeValueObjective.  Every component belongs to exactly one lane: no atomics.
        double* gr = hp.grad + (size_t)b * hp.n_params;
        int kinds[4], n_edges = 0;
        const int terms = CORBO_HIP_COST_TERMS(hp.stage_cost);
        const bool nl = hp.cost_nonlsq != 0;
        if (final_stage) { if (so[0] >= 0) kinds[n_edges++] = nl ? EK_FINAL_QCOST : EK_FINAL_COST; }
        else {
            if (hp.ms_mixed) kinds[n_edges++] = EK_MIXED_OBJ;   // computeObjectiveJacobian per vertex, column sums (:75-101)
            else if (hp.cost_integral) { if (k >= hp.quad_first_interval) kinds[n_edges++] = hp.cost_integral == 1 ? EK_INTEGRAL_TRAP : EK_INTEGRAL_LEFT; }
            else {
                if ((terms & 1) && k >= hp.quad_first_interval) kinds[n_edges++] = nl ? EK_STATE_QCOST : EK_STATE_COST;
                if ((terms & 2) && k >= hp.quad_first_interval) kinds[n_edges++] = nl ? EK_CONTROL_QCOST : EK_CONTROL_COST;
            }
            if ((terms & 4) && k == 0) { kinds[n_edges++] = nl ? EK_DT_QCOST : EK_DT_COST; kinds[n_edges++] = nl ? EK_DT_QCOST : EK_DT_COST; }
        }
        double obj = 0.0;
        for (int e = 0; e < n_edges; ++e) {
            const int kind = kinds[e], ed = HE::edge_dim(kind, mpl), off = HE::vert_off(kind, 0), dim = HE::vert_dim(kind, 0);
            double blk[HE::MAXD * HE::MAXD], vals[HE::MAXD];
            const int nu_ = HE::unfixed(fm, off, dim);
            if (kind >= EK_STATE_QCOST) {   // plain objective edge: gradient += the Jacobian's column sums per attached vertex, value += the sum of
                // the values (…edge_based.cpp:43-56, hyper_graph_optimization_problem_base.cpp:136-141).  An integral cost edge reaches into
                // x_{k+1}, whose other contribution comes from the next lane: atomic adds (two terms per component: either order gives the same sum)
                for (int vi = 0; vi < HE::n_verts(kind); ++vi) {
                    const int vo_ = HE::vert_off(kind, vi), vd_ = HE::vert_dim(kind, vi);
                    if (HE::unfixed(fm, vo_, vd_) == 0) continue;
                    HE::jacobian(kind, vi, fm, xl, xr, mpl, blk);
                    int col = 0;
                    for (int i = 0; i < vd_; ++i) {
                        if ((fm >> (vo_ + i)) & 1u) continue;
                        double acc = 0.0;
                        for (int r = 0; r < ed; ++r) acc += blk[col * ed + r];
                        const int L = vo_ + i;   // local slot -> vertex-storage offset: x_k u_k | x_{k+1} | dt
                        const int v = (L == HE::W - 1) ? p.off_dt : (L < S ? k * S + L : (k + 1) * S + (L - S));
                        atomicAdd(&gr[p.comp[v].param], acc);
                        ++col;
                    }
                }
                HE::values(kind, xl, xr, mpl, vals);
                for (int r = 0; r < ed; ++r) obj += vals[r];
                continue;
            }
            if (nu_ > 0) HE::jacobian(kind, 0, fm, xl, xr, mpl, blk);
            HE::values(kind, xl, xr, mpl, vals);
            int col = 0;
            for (int i = 0; i < dim; ++i) {
                if ((fm >> (off + i)) & 1u) continue;
                double acc = 0.0;
                for (int r = 0; r < ed; ++r) acc += (2.0 * vals[r]) * blk[col * ed + r];
                const int v = (kind == EK_DT_COST) ? p.off_dt : k * S + off + i;
                gr[p.comp[v].param] += acc;
                ++col;
            }
            double sq = 0.0;
            for (int r = 0; r < ed; ++r) sq += vals[r] * vals[r];
            obj += sq;
        }
        hp.obj_part[(size_t)b * p.N + k] = obj;
    }
    else {
        const int32_t* lo = hp.lin_off + (size_t)k * 2;
        const int rows = hp.eq_dim + hp.ineq_dim + hp.n_bounds;
        double* lv = hp.lin_vals + (size_t)b * hp.lin_nnz;
        double* lb = hp.lbA + (size_t)b * rows;
        double* ub = hp.ubA + (size_t)b * rows;
        double c[NX];
        if (lo[0] >= 0) {   // computeBoundsForTwoSideBoundedLinearForm: lbA = ubA = -c_eq
            const int kind = final_stage ? EK_FINAL_EQ : (hp.ms_mixed ? EK_MIXED_EQ : EK_DEFECT);   // (mixed edge: computeConstraintJacobians, :4944-4960)
            HE::values(kind, xl, xr, mpl, c);
            const int ed = HE::edge_dim(kind, mpl);
            for (int r = 0; r < NX; ++r)
                if (r < ed) { lb[so[4] + r] = c[r] * -1; ub[so[4] + r] = c[r] * -1; }
            HE::linear_blocks(kind, fm, xl, xr, mpl, lv + lo[0]);
        }
        if (lo[1] >= 0) {   // (-inf, -c_ineq]
            const int kind = final_stage ? EK_FINAL_INEQ : EK_STAGE_INEQ;
            HE::values(kind, xl, xr, mpl, c);
            lb[hp.eq_dim + so[5]] = -CORBO_HIP_INF;
            ub[hp.eq_dim + so[5]] = c[0] * -1;
            HE::linear_blocks(kind, fm, xl, xr, mpl, lv + lo[1]);
        }
        // finite bounds of this stage's components (and of dt, with the final stage): identity rows; lbA = lb - x, ubA = x - ub (sic, :1177-1178)
        const int ncomp = final_stage ? NX + 1 : S;
        for (int i = 0; i < ncomp; ++i) {
            const int v = (final_stage && i == NX) ? p.off_dt : k * S + i;
            const int br = p.comp[v].bnd_row;
            if (br < 0) continue;
            const int idx = br - hp.bnd_row0;
            const double xv = xg[v];
            lv[hp.lin_bounds0 + idx] = 1.0;
            lb[hp.eq_dim + hp.ineq_dim + idx] = p.lb[(size_t)inst * p.nvs + v] - xv;
            ub[hp.eq_dim + hp.ineq_dim + idx] = xv - p.ub[(size_t)inst * p.nvs + v];
        }
    }
}
#pragma clang fp contract(fast)

template <int DYN, int DEFECT>
void launch_hessian_t(const SweepParams& p, const HessParams& hp, hipStream_t stream)
{
    const int gz = (hp.mode == 0 && hp.split) ? 6 * (hp.split == 2 ? 16 : 4) : 1;   // Hessian values: waves per (edge, row vertex [, column vertex]) -- at most 6 edges per stage, 4 vertices per edge
    hipLaunchKernelGGL((hessian_kernel<DYN, DEFECT>), dim3((p.N + 63) / 64, p.batch, gz), dim3(64), 0, stream, p, hp);
}

template <int DYN>
bool launch_hessian_d(int defect, const SweepParams& p, const HessParams& hp, hipStream_t stream)
{
    {
        if (defect == CORBO_HIP_DEFECT_RK4_SHOOTING && (int)p.mp.dyn[7] >= 5) {
            if constexpr (Dynamics<DYN>::NX <= 4) { launch_hessian_t<DYN, DEFECT_SHOOTING_HIGH>(p, hp, stream); return true; }
            else return false;
        }
        switch (defect) {
            case CORBO_HIP_DEFECT_FORWARD: launch_hessian_t<DYN, CORBO_HIP_DEFECT_FORWARD>(p, hp, stream); return true;
            case CORBO_HIP_DEFECT_BACKWARD: launch_hessian_t<DYN, CORBO_HIP_DEFECT_BACKWARD>(p, hp, stream); return true;
            case CORBO_HIP_DEFECT_MIDPOINT: launch_hessian_t<DYN, CORBO_HIP_DEFECT_MIDPOINT>(p, hp, stream); return true;
            case CORBO_HIP_DEFECT_CRANK_NICOLSON: launch_hessian_t<DYN, CORBO_HIP_DEFECT_CRANK_NICOLSON>(p, hp, stream); return true;
            case CORBO_HIP_DEFECT_RK4_SHOOTING: launch_hessian_t<DYN, CORBO_HIP_DEFECT_RK4_SHOOTING>(p, hp, stream); return true;
            default: return false;
        }
    }
}

template <int NX, int NU>
size_t factor_lds(int N, bool arrow)
{
    return sizeof(double) * (size_t)FactorLds<NX, NU>::total(N | 1, arrow);
}

template <int DYN, int DEFECT>
bool launch_pass_t(const FactorParams& fp, const SweepParams& sp, hipStream_t stream)
{
    using Dy = Dynamics<DYN>;
    if (fp.N > SWEEP_THREADS) return false;
    if (sp.n_xedges > 0) return launch_bt_t<DYN, DEFECT>(fp, sp, stream);   // extra edges: the block-tridiagonal route (bt_factor.hpp)
    size_t dbl = FactorLds<Dy::NX, Dy::NU>::total(fp.N | 1, fp.dt_free != 0);
    if (dbl < (size_t)fp.nnz_pad + (size_t)fp.N * Dy::NC) dbl = (size_t)fp.nnz_pad + (size_t)fp.N * Dy::NC;  // staging + caches
    const size_t lds = sizeof(double) * (((dbl + 1) & ~(size_t)1) + fp.nvs + 12) + sizeof(LmState);           // + vertex values + LM state + scratch
    // queue mode: as many workgroups as the chip holds at once (register budget: 4 of these workgroups per CU; LDS: 160 KB per CU)
    int grid = fp.batch;
    if (fp.queue) {
        int per_cu = (int)((size_t)160 * 1024 / lds);
        if (per_cu > 4) per_cu = 4;
        if (per_cu < 1) per_cu = 1;
        grid = fp.queue_grid * per_cu;   // queue_grid = compute units of the device
        if (grid > fp.batch) grid = fp.batch;
    }
    // Workgroup shape of the run-to-completion kernel: TWO waves (128 threads, 256 VGPRs at the same four workgroups per CU) whenever every
    // block and the special component lanes fit (N + NX + 1 <= 128) -- fewer waves per SIMD, no register spills, the stage-centric component
    // pass, the zero neighbour slot (0.66 -> 0.55 ms per headline solve); four waves for longer horizons.  Handle option
    // "pass_threads" (256 / 128) forces a shape (A/B measurements).
    const bool two_ok = fp.N + Dy::NX + 1 <= 128;
    const bool two    = two_ok && fp.pass_threads != 256;
    if (fp.wdense_mask) {   // non-diagonal weights: the DENSE instantiation (four waves)
        if constexpr (Dy::NX <= 4) {
            if (two) {   // two waves, 256 VGPRs (the four-wave instantiation spills 130 registers at its 128)
                const dim3 b(128);
                if (fp.dt_free) hipLaunchKernelGGL((lm_pass_kernel<DYN, DEFECT, true, true, 0, 128, true>), dim3(grid), b, lds, stream, fp, sp);
                else hipLaunchKernelGGL((lm_pass_kernel<DYN, DEFECT, false, true, 0, 128, true>), dim3(grid), b, lds, stream, fp, sp);
                return true;
            }
            const dim3 b(SWEEP_THREADS);
            if (fp.dt_free) hipLaunchKernelGGL((lm_pass_kernel<DYN, DEFECT, true, true, 0, SWEEP_THREADS, true>), dim3(grid), b, lds, stream, fp, sp);
            else hipLaunchKernelGGL((lm_pass_kernel<DYN, DEFECT, false, true, 0, SWEEP_THREADS, true>), dim3(grid), b, lds, stream, fp, sp);
            return true;
        }
        else return false;
    }
    {
        if (two) {
            const dim3 b(128);
            if (fp.dt_free) hipLaunchKernelGGL((lm_pass_kernel<DYN, DEFECT, true, true, 0, 128>), dim3(grid), b, lds, stream, fp, sp);
            else if (fp.N == 100) hipLaunchKernelGGL((lm_pass_kernel<DYN, DEFECT, false, true, 101, 128>), dim3(grid), b, lds, stream, fp, sp);   // the headline horizon: LDS strides as immediates, zero neighbour slot
            else hipLaunchKernelGGL((lm_pass_kernel<DYN, DEFECT, false, true, 0, 128>), dim3(grid), b, lds, stream, fp, sp);
        }
        else {
            const dim3 b(SWEEP_THREADS);
            if (fp.dt_free) hipLaunchKernelGGL((lm_pass_kernel<DYN, DEFECT, true, true, 0>), dim3(grid), b, lds, stream, fp, sp);
            else hipLaunchKernelGGL((lm_pass_kernel<DYN, DEFECT, false, true, 0>), dim3(grid), b, lds, stream, fp, sp);
        }
    }
    return true;
}

template <int DYN>
bool launch_pass_d(int defect, const FactorParams& fp, const SweepParams& sp, hipStream_t stream)
{
#ifdef CORBO_HIP_DEV_FAST
    if (defect != CORBO_HIP_DEFECT_CRANK_NICOLSON) return false;
    return launch_pass_t<DYN, CORBO_HIP_DEFECT_CRANK_NICOLSON>(fp, sp, stream);
#else
    if (defect == CORBO_HIP_DEFECT_RK4_SHOOTING && (int)sp.mp.dyn[7] >= 5) return false;   // Runge-Kutta 5 / 6 / 7: separate launches only
    switch (defect) {
        case CORBO_HIP_DEFECT_FORWARD: return launch_pass_t<DYN, CORBO_HIP_DEFECT_FORWARD>(fp, sp, stream);
        case CORBO_HIP_DEFECT_BACKWARD: return launch_pass_t<DYN, CORBO_HIP_DEFECT_BACKWARD>(fp, sp, stream);
        case CORBO_HIP_DEFECT_MIDPOINT: return launch_pass_t<DYN, CORBO_HIP_DEFECT_MIDPOINT>(fp, sp, stream);
        case CORBO_HIP_DEFECT_CRANK_NICOLSON: return launch_pass_t<DYN, CORBO_HIP_DEFECT_CRANK_NICOLSON>(fp, sp, stream);
        case CORBO_HIP_DEFECT_RK4_SHOOTING: return launch_pass_t<DYN, CORBO_HIP_DEFECT_RK4_SHOOTING>(fp, sp, stream);
        default: return false;
    }
#endif
}

template <int NX, int NU, bool ARROW>
bool launch_factor_a(const FactorParams& p, hipStream_t stream)
{
    size_t lds = factor_lds<NX, NU>(p.N, ARROW);
    if (lds < sizeof(double) * (size_t)p.nnz_pad) lds = sizeof(double) * (size_t)p.nnz_pad;  // Jacobian staging area
    lds = ((lds + 15) & ~(size_t)15) + sizeof(LmState);                                      // + LM state
    if (p.N > LONG_HORIZON) {   // long horizon: workspace in HBM
        if (p.N > LONG_HORIZON_MAX || !p.work) return false;
        const size_t hyb = sizeof(double) * factor_long_hyb_lds_doubles<NX, NU>(p.N, ARROW);
        if (hyb + sizeof(LmState) + 64 <= (size_t)160 * 1024) {   // the state-block arrays fit the LDS of a CU: only the controls' arrays stay in the HBM workspace
            static unsigned long long attr_set[4] = {0, 0, 0, 0};   // (per device)
            auto go = [&](auto kernel, int slot, int threads) {
                if (first_on_device(attr_set[slot])) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)sizeof(LmState) - 64);
                hipLaunchKernelGGL(kernel, dim3(p.batch), dim3(threads), hyb, stream, p);
            };
            const bool two_per_cu = 2 * (hyb + sizeof(LmState) + 64) <= (size_t)160 * 1024;
            if (p.wdense_mask) go(factor_long_kernel<NX, NU, ARROW, true, true>, 1, 1024);
            else if (p.N <= 512 && p.pass_threads != 1024) {   // (option pass_threads = 1024: the sixteen-wave shape, A/B)
                if (two_per_cu) go(factor_long_kernel<NX, NU, ARROW, false, true, 512, 4>, 2, 512);
                else go(factor_long_kernel<NX, NU, ARROW, false, true, 512, 2>, 3, 512);
            }
            else go(factor_long_kernel<NX, NU, ARROW, false, true>, 0, 1024);
            return true;
        }
        if (p.wdense_mask) hipLaunchKernelGGL((factor_long_kernel<NX, NU, ARROW, true>), dim3(p.batch), dim3(1024), 0, stream, p);   // non-diagonal weights
        else hipLaunchKernelGGL((factor_long_kernel<NX, NU, ARROW>), dim3(p.batch), dim3(1024), 0, stream, p);
        return true;
    }
    if (p.wdense_mask) {   // non-diagonal weights: the DENSE instantiation
        if (p.N <= 128) hipLaunchKernelGGL((factor_kernel<NX, NU, 128, ARROW, true>), dim3(p.batch), dim3(128), lds, stream, p);
        else if (p.N <= 256) hipLaunchKernelGGL((factor_kernel<NX, NU, 256, ARROW, true>), dim3(p.batch), dim3(256), lds, stream, p);
        else return false;
        return true;
    }
    if (p.N <= 128) hipLaunchKernelGGL((factor_kernel<NX, NU, 128, ARROW>), dim3(p.batch), dim3(128), lds, stream, p);
    else if (p.N <= 256) hipLaunchKernelGGL((factor_kernel<NX, NU, 256, ARROW>), dim3(p.batch), dim3(256), lds, stream, p);
    else return false;
    return true;
}

template <int NX, int NU>
bool launch_factor_t(const FactorParams& p, hipStream_t stream)
{
    return p.dt_free ? launch_factor_a<NX, NU, true>(p, stream) : launch_factor_a<NX, NU, false>(p, stream);
}

}  // namespace

#pragma clang fp contract(off)
// SimulatedPlant::control without dead time: x+ = integrator.solveIVP(x, u_0, dt) (explicit Euler, explicit_integrators.h:66-72:
// f * dt + x; Runge-Kutta 4, :280-295), then the state disturbance.  Operation for operation the host formulas: bit-identical.
template <int DYN>
__global__ __launch_bounds__(256) void plant_step_kernel(const PlantParams p)
{
    using D = Dynamics<DYN>;
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= p.batch) return;
    double x1[D::NX], u[D::NU], xe[D::NX];
    double* xp = p.xplant + (size_t)b * CORBO_HIP_MAX_NX;
#pragma unroll
    for (int i = 0; i < D::NX; ++i) x1[i] = xp[i];
#pragma unroll
    for (int i = 0; i < D::NU; ++i) u[i] = p.x[(size_t)b * p.nvs + D::NX + i];
    double prm[8];   // model parameters of THIS plant
#pragma unroll
    for (int i = 0; i < 8; ++i) prm[i] = p.dyn_inst ? p.dyn_inst[(size_t)b * 8 + i] : p.dyn[i];
    prm[7] = 0.0;   // (slot 7 selects the shooting grids' integrator for the defect edges; the plant's own integrator is p.integrator)
    if (p.integrator == CORBO_HIP_INTEGRATOR_RK4) {
        double ck[4][D::NC];
        rk4_end_state<DYN, false>(x1, u, p.dt, prm, ck, xe);
    }
    else {
        dyn_full<DYN>(x1, u, prm, xe);
        if (p.integrator == CORBO_HIP_INTEGRATOR_EULER) {
#pragma unroll
            for (int i = 0; i < D::NX; ++i) { xe[i] *= p.dt; xe[i] += x1[i]; }
        }   // else (internal, corbo_hip_eval_dynamics): the right-hand side f(x, u) itself
    }
    if (p.disturbance) {
#pragma unroll
        for (int i = 0; i < D::NX; ++i) xe[i] = xe[i] + p.disturbance[(size_t)b * CORBO_HIP_MAX_NX + i];
    }
#pragma unroll
    for (int i = 0; i < D::NX; ++i) xp[i] = xe[i];
    if (p.log_x) {
#pragma unroll
        for (int i = 0; i < D::NX; ++i) p.log_x[(size_t)b * D::NX + i] = xe[i];
    }
    if (p.log_u) {
#pragma unroll
        for (int i = 0; i < D::NU; ++i) p.log_u[(size_t)b * D::NU + i] = u[i];
    }
}
#pragma clang fp contract(fast)

template <int DYN>
static void launch_plant_step_t(const PlantParams& p, hipStream_t stream)
{
    hipLaunchKernelGGL(plant_step_kernel<DYN>, dim3((p.batch + 255) / 256), dim3(256), 0, stream, p);
}


// ---------------------------------------------------------------------------------------------------------------------
// Translation units.  This file is compiled once as the MAIN unit (CORBO_HIP_DYN_TU undefined): the kernels that do not depend on
// the dynamics model (factor / big-block / warm start / helpers) and the dispatch; and once PER DYNAMICS MODEL with
//   -DCORBO_HIP_DYN_TU=<template id> -DCORBO_HIP_DYN_TU_NAME=<suffix> [-DCORBO_HIP_DYN_TU_BIG]
// : the sweep, fused-pass and plant kernels of that model behind three plain entry functions.  The models compile in parallel
// (__graft_entry__.build()); a new model costs one more unit, not a longer critical path.
// ---------------------------------------------------------------------------------------------------------------------
#define CORBO_HIP_DYN_ENTRIES(NAME)                                                                              \
    bool sweep_entry_##NAME(int defect, const SweepParams& p, hipStream_t stream);                               \
    bool pass_entry_##NAME(int defect, const FactorParams& fp, const SweepParams& sp, hipStream_t stream);       \
    void plant_entry_##NAME(const PlantParams& p, hipStream_t stream);                                           \
    bool hessian_entry_##NAME(int defect, const SweepParams& p, const HessParams& hp, hipStream_t stream);
CORBO_HIP_DYN_ENTRIES(vdp)
CORBO_HIP_DYN_ENTRIES(integ2)
CORBO_HIP_DYN_ENTRIES(integ3)
CORBO_HIP_DYN_ENTRIES(unicycle)
CORBO_HIP_DYN_ENTRIES(quadrotor)
CORBO_HIP_DYN_ENTRIES(duffing)
CORBO_HIP_DYN_ENTRIES(rocket)
CORBO_HIP_DYN_ENTRIES(pendulum)
CORBO_HIP_DYN_ENTRIES(mpendulum)
CORBO_HIP_DYN_ENTRIES(toy)
CORBO_HIP_DYN_ENTRIES(artstein)
CORBO_HIP_DYN_ENTRIES(cartpole)
CORBO_HIP_DYN_ENTRIES(par2)
CORBO_HIP_DYN_ENTRIES(par3)
CORBO_HIP_DYN_ENTRIES(lin21)
CORBO_HIP_DYN_ENTRIES(lin22)
CORBO_HIP_DYN_ENTRIES(lin31)
CORBO_HIP_DYN_ENTRIES(lin32)
CORBO_HIP_DYN_ENTRIES(lin33)
CORBO_HIP_DYN_ENTRIES(lin41)

// user models (csrc/models/*.hpp, registry generated by the build): one set of entries each
#if __has_include("models/_registry.inc")
#define CORBO_HIP_USER_MODEL(NAME, SLOT, NX_, NU_, P0, P1, P2, P3) CORBO_HIP_DYN_ENTRIES(user_##NAME)
#include "models/_registry.inc"
#undef CORBO_HIP_USER_MODEL
#endif

// big-block family (5 <= nx <= 12; quadrotor and the user models of that size): stage kernel (Jacobian dump / assemble) and the factorisation
#define CORBO_HIP_BIG_ENTRIES(NAME)                                                                                                       \
    bool stage_entry_##NAME(const FactorParams& fp, const SweepParams& sp, int diag_only, double* jac_dump, hipStream_t stream);          \
    bool factor_entry_##NAME(const FactorParams& fp, const SweepParams& sp, hipStream_t stream);
CORBO_HIP_BIG_ENTRIES(quadrotor)
#if __has_include("models/_registry.inc")
#define CORBO_HIP_USER_MODEL(NAME, SLOT, NX_, NU_, P0, P1, P2, P3) CORBO_HIP_BIG_ENTRIES(user_##NAME)
#include "models/_registry.inc"
#undef CORBO_HIP_USER_MODEL
#endif

#ifdef CORBO_HIP_DYN_TU
#define CORBO_HIP_CAT2(a, b) a##b
#define CORBO_HIP_CAT(a, b) CORBO_HIP_CAT2(a, b)
bool CORBO_HIP_CAT(sweep_entry_, CORBO_HIP_DYN_TU_NAME)(int defect, const SweepParams& p, hipStream_t stream)
{
    return launch_sweep_d<CORBO_HIP_DYN_TU>(defect, p, stream);   // (big-block family: shooting with its integrators, and the four collocation formulas)
}
bool CORBO_HIP_CAT(pass_entry_, CORBO_HIP_DYN_TU_NAME)(int defect, const FactorParams& fp, const SweepParams& sp, hipStream_t stream)
{
#ifdef CORBO_HIP_DYN_TU_BIG   // no fused pass kernel for the big-block family
    (void)defect; (void)fp; (void)sp; (void)stream;
    return false;
#else
    return launch_pass_d<CORBO_HIP_DYN_TU>(defect, fp, sp, stream);
#endif
}
void CORBO_HIP_CAT(plant_entry_, CORBO_HIP_DYN_TU_NAME)(const PlantParams& p, hipStream_t stream) { launch_plant_step_t<CORBO_HIP_DYN_TU>(p, stream); }
bool CORBO_HIP_CAT(hessian_entry_, CORBO_HIP_DYN_TU_NAME)(int defect, const SweepParams& p, const HessParams& hp, hipStream_t stream)
{
    return launch_hessian_d<CORBO_HIP_DYN_TU>(defect, p, hp, stream);
}
#ifdef CORBO_HIP_DYN_TU_BIG
bool CORBO_HIP_CAT(stage_entry_, CORBO_HIP_DYN_TU_NAME)(const FactorParams& fp, const SweepParams& sp, int diag_only, double* jac_dump, hipStream_t stream)
{
    using Dy = Dynamics<CORBO_HIP_DYN_TU>;
    if (!sp.xe0 || (!fp.work && !jac_dump)) return false;
    const size_t lds = sizeof(double) * (size_t)BigLds<Dy::NX, Dy::NU>::TOTAL;
    const dim3 g((fp.N + 1) / 2, fp.batch), b(64);
    // (a user state function as the stage inequality -- csrc/stage_functions/, only where one is registered for this state dimension -- : the USERINEQ instantiation)
    const bool user_ineq = has_user_state_ineq<Dy::NX>() && sp.mp.ineq_id >= CORBO_HIP_STAGE_FN_USER;
#define CORBO_HIP_STAGE_LAUNCH(DEFECT_, ARROW_)                                                                                                                             \
    do {                                                                                                                                                                    \
        if constexpr (has_user_state_ineq<Dy::NX>()) {                                                                                                                      \
            if (user_ineq) { hipLaunchKernelGGL((big_stage_kernel<CORBO_HIP_DYN_TU, true, DEFECT_, ARROW_, true>), g, b, lds, stream, fp, sp, diag_only, jac_dump); return true; } \
        }                                                                                                                                                                   \
        hipLaunchKernelGGL((big_stage_kernel<CORBO_HIP_DYN_TU, true, DEFECT_, ARROW_, false>), g, b, lds, stream, fp, sp, diag_only, jac_dump);                             \
        return true;                                                                                                                                                        \
    } while (0)
    (void)user_ineq;
    if (fp.dt_free) {   // free dt: the dt column of every defect edge and the border parts (second right-hand side of the chain)
        // (even block sizes only -- the partitioned chain carries the border)
        if constexpr (Dy::NX % 2 != 0) return false;
        else
        switch (fp.defect) {
            case CORBO_HIP_DEFECT_RK4_SHOOTING:
                if ((int)sp.mp.dyn[7] >= 5) CORBO_HIP_STAGE_LAUNCH(DEFECT_SHOOTING_HIGH, true);   // Runge-Kutta 5 / 6 / 7
                else CORBO_HIP_STAGE_LAUNCH(CORBO_HIP_DEFECT_RK4_SHOOTING, true);
            case CORBO_HIP_DEFECT_FORWARD: CORBO_HIP_STAGE_LAUNCH(CORBO_HIP_DEFECT_FORWARD, true);
            case CORBO_HIP_DEFECT_BACKWARD: CORBO_HIP_STAGE_LAUNCH(CORBO_HIP_DEFECT_BACKWARD, true);
            case CORBO_HIP_DEFECT_MIDPOINT: CORBO_HIP_STAGE_LAUNCH(CORBO_HIP_DEFECT_MIDPOINT, true);
            case CORBO_HIP_DEFECT_CRANK_NICOLSON: CORBO_HIP_STAGE_LAUNCH(CORBO_HIP_DEFECT_CRANK_NICOLSON, true);
            default: return false;
        }
    }
    switch (fp.defect) {   // shooting (Runge-Kutta 4 / 3 / 2, Euler; 5 / 6 / 7: an instantiation of its own), or a collocation formula on the FiniteDifferencesGrid
        case CORBO_HIP_DEFECT_RK4_SHOOTING:
            if ((int)sp.mp.dyn[7] >= 5) CORBO_HIP_STAGE_LAUNCH(DEFECT_SHOOTING_HIGH, false);
            else CORBO_HIP_STAGE_LAUNCH(CORBO_HIP_DEFECT_RK4_SHOOTING, false);
        case CORBO_HIP_DEFECT_FORWARD: CORBO_HIP_STAGE_LAUNCH(CORBO_HIP_DEFECT_FORWARD, false);
        case CORBO_HIP_DEFECT_BACKWARD: CORBO_HIP_STAGE_LAUNCH(CORBO_HIP_DEFECT_BACKWARD, false);
        case CORBO_HIP_DEFECT_MIDPOINT: CORBO_HIP_STAGE_LAUNCH(CORBO_HIP_DEFECT_MIDPOINT, false);
        case CORBO_HIP_DEFECT_CRANK_NICOLSON: CORBO_HIP_STAGE_LAUNCH(CORBO_HIP_DEFECT_CRANK_NICOLSON, false);
        default: return false;
    }
#undef CORBO_HIP_STAGE_LAUNCH
}
// one factorisation of the big-block family: (first factorisation of a solve: diag pass + mu / stop) stage kernel, then the chain.  The
// stacked chain kernel (big_chain2_kernel) is laid out for state blocks of 4, 8 or 12 rows; other sizes take the first formulation
// (big_chain_kernel, without the matrix-core tiles).
bool CORBO_HIP_CAT(factor_entry_, CORBO_HIP_DYN_TU_NAME)(const FactorParams& p, const SweepParams& sp, hipStream_t stream)
{
    using Dy = Dynamics<CORBO_HIP_DYN_TU>;
    constexpr int NX = Dy::NX, NU = Dy::NU;
    static_assert(big_family_dims(NX, NU), "big-block family: 5 <= nx <= 12, nu <= 4, nx + nu <= 16");
    if (!p.work) return false;
    if (p.dt_free && NX % 2 != 0) return false;   // (a free dt rides through the partitioned chain only: even block sizes; others take the band route)
    if (p.first_pass) {   // (the kernels themselves also check LmState::first)
        if (!CORBO_HIP_CAT(stage_entry_, CORBO_HIP_DYN_TU_NAME)(p, sp, 1, nullptr, stream)) return false;
        if (p.dt_free) hipLaunchKernelGGL((big_first_kernel<NX, NU, true>), dim3(p.batch), dim3(64), 0, stream, p);
        else hipLaunchKernelGGL((big_first_kernel<NX, NU>), dim3(p.batch), dim3(64), 0, stream, p);
    }
    if (!CORBO_HIP_CAT(stage_entry_, CORBO_HIP_DYN_TU_NAME)(p, sp, 0, nullptr, stream)) return false;
    if constexpr (NX % 2 == 0) {
        if (p.dt_free) {   // free dt: the partitioned chain with the border column as a second right-hand side; short horizons: one segment (two waves, from both ends)
            int nseg = (p.N >= 64) ? 4 : 1;
            if (p.chain_variant == 3) nseg = 4;
            if (p.chain_variant == 4) nseg = 2;
            if (p.chain_variant == 6) nseg = 1;
            if (p.N < 4 * nseg) nseg = 1;
            auto launch3a = [&](auto kernel, int nseg_, size_t lds3) {
                static unsigned long long attr_set[9] = {};   // (a kernel's attributes are per DEVICE: one bit per device, handles may live on several)
                if (first_on_device(attr_set[nseg_])) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipLaunchKernelGGL(kernel, dim3(p.batch), dim3(128 * nseg_), ((lds3 + 15) & ~(size_t)15) + sizeof(LmState), stream, p);
            };
            if (nseg == 4) launch3a(big_chain3_kernel<NX, NU, 4, true>, 4, sizeof(double) * (size_t)Chain3Lds<NX, NU, 4, true>::total(p.N));
            else if (nseg == 2) launch3a(big_chain3_kernel<NX, NU, 2, true>, 2, sizeof(double) * (size_t)Chain3Lds<NX, NU, 2, true>::total(p.N));
            else launch3a(big_chain3_kernel<NX, NU, 1, true>, 1, sizeof(double) * (size_t)Chain3Lds<NX, NU, 1, true>::total(p.N));
            return true;
        }
        if (p.chain_variant == 1)   // (diagnostics: the first formulation)
            hipLaunchKernelGGL((big_chain_kernel<NX, NU, NX % 4 == 0>), dim3(p.batch), dim3(128), sizeof(double) * (4 * NX * NX + 2 * NX + 8), stream, p);
        else {
            // partitioned chain (big_chain3_kernel): NSEG segments = 2 NSEG waves per instance.  chain_variant 0 = automatic (horizons of 64 grid points
            // and more: four segments), 2 = the twisted chain (big_chain2_kernel) whatever the horizon, 3 / 4 / 5 / 6 = 4 / 2 / 8 / 1 segments (tests, A/B)
            int nseg = 0;
            if (p.chain_variant == 0 && p.N >= 64) nseg = 4;
            if (p.chain_variant == 3) nseg = 4;
            if (p.chain_variant == 4) nseg = 2;
            if (p.chain_variant == 5) nseg = 8;
            if (p.chain_variant == 6) nseg = 1;
            if (nseg > 0 && p.N < 4 * nseg) nseg = 0;   // (every segment needs a block of its own next to its separators)
            if (NX % 4 != 0 && nseg == 0) nseg = 1;     // (6- and 10-row blocks: the twisted chain's matrix-core tiling assumes multiples of four; one segment is the same elimination)
            auto launch3 = [&](auto kernel, int nseg_, size_t lds3) {
                static unsigned long long attr_set[9] = {};   // (a kernel's attributes are per DEVICE: one bit per device, handles may live on several)
                if (first_on_device(attr_set[nseg_])) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipLaunchKernelGGL(kernel, dim3(p.batch), dim3(128 * nseg_), ((lds3 + 15) & ~(size_t)15) + sizeof(LmState), stream, p);
            };
            if (nseg == 4) launch3(big_chain3_kernel<NX, NU, 4>, 4, sizeof(double) * (size_t)Chain3Lds<NX, NU, 4>::total(p.N));
            else if (nseg == 2) launch3(big_chain3_kernel<NX, NU, 2>, 2, sizeof(double) * (size_t)Chain3Lds<NX, NU, 2>::total(p.N));
            else if (nseg == 8) launch3(big_chain3_kernel<NX, NU, 8>, 8, sizeof(double) * (size_t)Chain3Lds<NX, NU, 8>::total(p.N));
            else if (nseg == 1) launch3(big_chain3_kernel<NX, NU, 1>, 1, sizeof(double) * (size_t)Chain3Lds<NX, NU, 1>::total(p.N));
            else if constexpr (NX % 4 == 0) {
                const size_t lds2 = 2 * sizeof(double) * (size_t)Chain2Lds<NX, NU>::total(p.N);   // two instances per workgroup
                hipLaunchKernelGGL((big_chain2_kernel<NX, NU>), dim3((p.batch + 1) / 2), dim3(256), lds2, stream, p);
            }
        }
    }
    else hipLaunchKernelGGL((big_chain_kernel<NX, NU, false>), dim3(p.batch), dim3(128), sizeof(double) * (4 * NX * NX + 2 * NX + 8), stream, p);
    return true;
}
#else   // small-block family: the dispatch tables name these entries for every user model
bool CORBO_HIP_CAT(stage_entry_, CORBO_HIP_DYN_TU_NAME)(const FactorParams&, const SweepParams&, int, double*, hipStream_t) { return false; }
bool CORBO_HIP_CAT(factor_entry_, CORBO_HIP_DYN_TU_NAME)(const FactorParams&, const SweepParams&, hipStream_t) { return false; }
#endif
#endif  // CORBO_HIP_DYN_TU

#ifndef CORBO_HIP_DYN_TU
__global__ __launch_bounds__(256) void reference_window_kernel(const double* __restrict__ traj, double* __restrict__ refvec, int batch, int T, int N, int nx,
                                                               int s, int nvs, int step)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= batch * N * nx) return;
    const int b = t / (N * nx), r = t - b * (N * nx), k = r / nx, i = r - k * nx;
    const int smp = (step + k < T - 1) ? step + k : T - 1;
    refvec[(size_t)b * nvs + (size_t)k * s + i] = traj[((size_t)b * T + smp) * nx + i];
}

void launch_reference_window(const double* traj, double* refvec, int batch, int T, int N, int nx, int s, int nvs, int step, hipStream_t stream)
{
    hipLaunchKernelGGL(reference_window_kernel, dim3((batch * N * nx + 255) / 256), dim3(256), 0, stream, traj, refvec, batch, T, N, nx, s, nvs, step);
}

__global__ __launch_bounds__(256) void broadcast_rows_kernel(const double* __restrict__ row_a, const double* __restrict__ row_b,
                                                             double* __restrict__ dst_a, double* __restrict__ dst_b, int nvs)
{
    const size_t base = (size_t)blockIdx.x * nvs;
    for (int i = threadIdx.x; i < nvs; i += 256) {
        dst_a[base + i] = row_a[i];
        dst_b[base + i] = row_b[i];
    }
}

bool launch_plant_step(const corbo_hip_problem_desc& d, const PlantParams& p, hipStream_t stream)
{
#if __has_include("models/_registry.inc")
#define CORBO_HIP_USER_MODEL(NAME, SLOT, NX_, NU_, P0, P1, P2, P3) if (d.dynamics == CORBO_HIP_DYN_USER + SLOT) { plant_entry_user_##NAME(p, stream); return true; }
#include "models/_registry.inc"
#undef CORBO_HIP_USER_MODEL
#endif
    switch (d.dynamics) {
        case CORBO_HIP_DYN_VAN_DER_POL: plant_entry_vdp(p, stream); return true;
        case CORBO_HIP_DYN_SERIAL_INTEGRATOR:
            if (d.nx == 3) { plant_entry_integ3(p, stream); return true; }
            if (d.nx != 2) return false;
            plant_entry_integ2(p, stream);
            return true;
        case CORBO_HIP_DYN_UNICYCLE: plant_entry_unicycle(p, stream); return true;
        case CORBO_HIP_DYN_QUADROTOR: plant_entry_quadrotor(p, stream); return true;
        case CORBO_HIP_DYN_DUFFING: plant_entry_duffing(p, stream); return true;
        case CORBO_HIP_DYN_FREE_SPACE_ROCKET: plant_entry_rocket(p, stream); return true;
        case CORBO_HIP_DYN_SIMPLE_PENDULUM: plant_entry_pendulum(p, stream); return true;
        case CORBO_HIP_DYN_MASSLESS_PENDULUM: plant_entry_mpendulum(p, stream); return true;
        case CORBO_HIP_DYN_TOY_EXAMPLE: plant_entry_toy(p, stream); return true;
        case CORBO_HIP_DYN_ARTSTEINS_CIRCLE: plant_entry_artstein(p, stream); return true;
        case CORBO_HIP_DYN_CART_POLE: plant_entry_cartpole(p, stream); return true;
        case CORBO_HIP_DYN_PARALLEL_INTEGRATOR:
            if (d.nx == 2) plant_entry_par2(p, stream); else plant_entry_par3(p, stream);
            return true;
        case CORBO_HIP_DYN_LINEAR_STATE_SPACE:
            if (d.nx == 2 && d.nu == 1) { plant_entry_lin21(p, stream); return true; }
            if (d.nx == 2 && d.nu == 2) { plant_entry_lin22(p, stream); return true; }
            if (d.nx == 3 && d.nu == 1) { plant_entry_lin31(p, stream); return true; }
            if (d.nx == 3 && d.nu == 2) { plant_entry_lin32(p, stream); return true; }
            if (d.nx == 3 && d.nu == 3) { plant_entry_lin33(p, stream); return true; }
            if (d.nx == 4 && d.nu == 1) { plant_entry_lin41(p, stream); return true; }
            return false;
        default: return false;
    }
}

__global__ __launch_bounds__(256) void gather_first_control_kernel(const double* __restrict__ x, double* __restrict__ out, int nvs, int nx, int nu,
                                                                   int batch)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= batch * nu) return;
    const int b = t / nu, i = t - b * nu;
    out[t] = x[(size_t)b * nvs + nx + i];
}

void launch_gather_first_control(const double* x, double* out, int nvs, int nx, int nu, int batch, hipStream_t stream)
{
    hipLaunchKernelGGL(gather_first_control_kernel, dim3((batch * nu + 255) / 256), dim3(256), 0, stream, x, out, nvs, nx, nu, batch);
}

void launch_broadcast_rows(const double* row_a, const double* row_b, double* dst_a, double* dst_b, int nvs, int batch, hipStream_t stream)
{
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3(batch), dim3(256), 0, stream, row_a, row_b, dst_a, dst_b, nvs);
}

// dst[i] = src[i] (and dst2[i] = src[i] when given), n2 double2 elements: device-to-device copies on the handle's own stream as a plain
// kernel -- hipMemcpyAsync(DeviceToDevice) wakes a copy engine, 0.1 - 0.3 ms before the next kernel of the stream may start (measured:
// re-arm + solve of one OCP 0.44 ms per step with the copy engine, 0.13 ms with this kernel)
__global__ __launch_bounds__(256) void copy_rows_kernel(const double2* __restrict__ src, double2* __restrict__ dst, double2* __restrict__ dst2, double2* __restrict__ dst3, size_t n2)
{
    // four elements per lane and round, all four requested before the first store (one element per round left the copy latency-bound:
    // cfg 5's re-arm, 14 MB into two destinations, 72 us = 0.58 TB/s)
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += 4 * stride) {
        double2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const size_t j = i + r * stride; v[r] = src[j < n2 ? j : i]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t j = i + r * stride;
            if (j >= n2) break;
            dst[j] = v[r];
            if (dst2) dst2[j] = v[r];
            if (dst3) dst3[j] = v[r];
        }
    }
}
// n 32-bit zeros as a plain kernel on the caller's stream (hipMemsetAsync was seen to hold the stream back until a copy kernel on ANOTHER
// stream of the handle had finished: the result delivery of the host-driven handles, corbo_hip.hip deliver_results)
__global__ __launch_bounds__(256) void zero_ints_kernel(int32_t* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = 0;
}
void launch_zero_ints(int32_t* dst, size_t n, hipStream_t stream)
{
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_ints_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, n);
}
void launch_copy_rows(const double* src, double* dst, double* dst2, size_t doubles, hipStream_t stream, double* dst3)
{
    const size_t n2 = doubles / 2;   // (row strides are even: 16-byte elements)
    size_t blocks = (n2 + 1023) / 1024;   // (four elements per lane)
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const double2*>(src), reinterpret_cast<double2*>(dst),
                       reinterpret_cast<double2*>(dst2), reinterpret_cast<double2*>(dst3), n2);
}

// The whole upload of corbo_hip_set_instance_data as ONE launch (the per-solve path of the drop-in adapter: five small launches back to back
// cost ~5 us each on an otherwise idle stream): iterate -> accepted / trial / re-arm copies, bounds from the per-instance staging arrays or
// from the descriptor's pattern rows, state references.  All sources are device-visible (pinned host memory or device arrays).
__global__ __launch_bounds__(256) void upload_instance_kernel(const UploadParams p)
{
    const size_t stride = (size_t)gridDim.x * 256, t0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = t0; i < p.n2; i += stride) {
        const double2 v = p.x[i];
        p.dx[i] = v;
        p.dxt[i] = v;
        if (p.dx0) p.dx0[i] = v;
        p.dlb[i] = p.lb_src ? p.lb_src[i] : p.row_lb[i % p.nvs2];
        p.dub[i] = p.ub_src ? p.ub_src[i] : p.row_ub[i % p.nvs2];
    }
    for (size_t i = t0; i < p.nref2; i += stride) p.dxref[i] = p.xref[i];
}
void launch_upload_instance(const UploadParams& p, hipStream_t stream)
{
    size_t blocks = (p.n2 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(upload_instance_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
}

void launch_resample(const ResampleParams& p, hipStream_t stream)
{
    hipLaunchKernelGGL(resample_kernel, dim3(p.pairs), dim3(256), sizeof(double) * (size_t)p.nvs_src, stream, p);
}

__global__ __launch_bounds__(256) void gather_dt_kernel(const double* __restrict__ x, double* __restrict__ out, int nvs, int off_dt, int batch)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < batch) out[b] = x[(size_t)b * nvs + off_dt];
}

void launch_gather_dt(const double* x, double* out, int nvs, int off_dt, int batch, hipStream_t stream)
{
    hipLaunchKernelGGL(gather_dt_kernel, dim3((batch + 255) / 256), dim3(256), 0, stream, x, out, nvs, off_dt, batch);
}

void launch_warm_start(const WarmStartParams& p, hipStream_t stream)
{
    const size_t lds = sizeof(double) * ((size_t)p.nvs + 24);
    hipLaunchKernelGGL(warm_start_kernel, dim3(p.batch), dim3(256), lds, stream, p);
}

size_t sweep_lds_bytes(const SweepParams& p, int nc)
{
    // vertex values + reduction scratch + dynamics caches + Jacobian staging; the headline family must stay below 40 KB so that
    // four workgroups share a CU (1024 instances = one round over 256 CUs)
    static_assert(LONG_HORIZON == 256, "jacobian_staged_in_lds");
    const size_t stage = jacobian_staged_in_lds(p.nx, p.N) ? (size_t)p.nnz_pad : 0;  // see STAGE in sweep_body
    const size_t role_table = (p.nx > 4) ? SWEEP_ROLE_TABLE : 0;   // big models: weights / reference by slot (sweep_body, comp_role), in front of the LM state
    return sizeof(double) * ((size_t)p.nvs + 10 + (((size_t)p.N * nc + 1) & ~(size_t)1) + stage + role_table) + sizeof(LmState);
}

size_t factor_work_doubles(const corbo_hip_problem_desc& d)
{
    if (big_family_dims(d.nx, d.nu)) return (size_t)d.N * big_ws_stage(d.nx, d.nu);
    if (d.N > LONG_HORIZON && d.N <= LONG_HORIZON_MAX) {   // small-block families, long horizon: the factor carve lives in HBM
        const bool arrow = (d.grid == CORBO_HIP_GRID_FD_VARIABLE || d.grid == CORBO_HIP_GRID_MS_VARIABLE);
        if (d.nx == 2 && d.nu == 1) return factor_long_work_doubles<2, 1>(d.N, arrow);
        if (d.nx == 3 && d.nu == 2) return factor_long_work_doubles<3, 2>(d.N, arrow);
        if (d.nx == 3 && d.nu == 1) return factor_long_work_doubles<3, 1>(d.N, arrow);
        if (d.nx == 4 && d.nu == 1) return factor_long_work_doubles<4, 1>(d.N, arrow);
        if (d.nx == 2 && d.nu == 2) return factor_long_work_doubles<2, 2>(d.N, arrow);
        if (d.nx == 3 && d.nu == 3) return factor_long_work_doubles<3, 3>(d.N, arrow);
    }
    return 0;
}

size_t factor_lds_bytes(const corbo_hip_problem_desc& d, const FactorParams& p)
{
    if (big_family_dims(d.nx, d.nu)) return sizeof(double) * big_lds_total(d.nx, d.nu);
    const bool arrow = (d.grid == CORBO_HIP_GRID_FD_VARIABLE || d.grid == CORBO_HIP_GRID_MS_VARIABLE);
    if (d.nx == 2 && d.nu == 1) return factor_lds<2, 1>(p.N, arrow);
    if (d.nx == 3 && d.nu == 2) return factor_lds<3, 2>(p.N, arrow);
    if (d.nx == 3 && d.nu == 1) return factor_lds<3, 1>(p.N, arrow);
    if (d.nx == 4 && d.nu == 1) return factor_lds<4, 1>(p.N, arrow);
    if (d.nx == 2 && d.nu == 2) return factor_lds<2, 2>(p.N, arrow);
    if (d.nx == 3 && d.nu == 3) return factor_lds<3, 3>(p.N, arrow);
    return 0;
}

// Host-only mirror of the launch_sweep / launch_pass / launch_factor dispatch below: is there a device kernel set for this
// (dynamics, nx, nu, defect, grid)?  corbo_hip_create refuses a descriptor at once instead of failing in the first solve.
bool device_kernels_exist(const corbo_hip_problem_desc& d)
{
    const bool known_defect = d.defect >= CORBO_HIP_DEFECT_FORWARD && d.defect <= CORBO_HIP_DEFECT_RK4_SHOOTING;
    if (!known_defect) return false;
    auto is = [&](int nx, int nu) { return d.nx == nx && d.nu == nu; };
    // big-block family: the shooting grids (explicit integrators) or the collocation grids (the four formulas); with a fixed dt the stage / chain
    // kernels, with a free dt (...VariableGrid) the sweep kernel's stored Jacobian and the band factorisation (corbo_hip_create, band_route)
    const bool big_grid = (d.defect == CORBO_HIP_DEFECT_RK4_SHOOTING && (d.grid == CORBO_HIP_GRID_MS || d.grid == CORBO_HIP_GRID_MS_VARIABLE)) ||
                          (d.defect != CORBO_HIP_DEFECT_RK4_SHOOTING && (d.grid == CORBO_HIP_GRID_FD || d.grid == CORBO_HIP_GRID_FD_VARIABLE));
#if __has_include("models/_registry.inc")
#define CORBO_HIP_USER_MODEL(NAME, SLOT, NX_, NU_, P0, P1, P2, P3) \
    if (d.dynamics == CORBO_HIP_DYN_USER + SLOT)                     \
        return is(NX_, NU_) && (NX_ <= 4 || (big_family_dims(NX_, NU_) && big_grid));   // (big-block family: as the quadrotor)
#include "models/_registry.inc"
#undef CORBO_HIP_USER_MODEL
#endif
    switch (d.dynamics) {
        case CORBO_HIP_DYN_VAN_DER_POL: case CORBO_HIP_DYN_DUFFING: case CORBO_HIP_DYN_SIMPLE_PENDULUM: case CORBO_HIP_DYN_MASSLESS_PENDULUM:
        case CORBO_HIP_DYN_TOY_EXAMPLE: case CORBO_HIP_DYN_ARTSTEINS_CIRCLE: return is(2, 1);
        case CORBO_HIP_DYN_FREE_SPACE_ROCKET: return is(3, 1);
        case CORBO_HIP_DYN_CART_POLE: return is(4, 1);
        case CORBO_HIP_DYN_SERIAL_INTEGRATOR: return is(2, 1) || is(3, 1);
        case CORBO_HIP_DYN_PARALLEL_INTEGRATOR: return is(2, 2) || is(3, 3);
        case CORBO_HIP_DYN_UNICYCLE: return is(3, 2);
        case CORBO_HIP_DYN_LINEAR_STATE_SPACE: return is(2, 1) || is(2, 2) || is(3, 1) || is(3, 2) || is(3, 3) || is(4, 1);
        case CORBO_HIP_DYN_QUADROTOR:   // big-block family: multiple shooting with RK4, fixed dt
            return is(12, 4) && big_grid;
        default: return false;
    }
}

bool launch_sweep(const corbo_hip_problem_desc& d, const SweepParams& p, hipStream_t stream)
{
#if __has_include("models/_registry.inc")
#define CORBO_HIP_USER_MODEL(NAME, SLOT, NX_, NU_, P0, P1, P2, P3) if (d.dynamics == CORBO_HIP_DYN_USER + SLOT) return sweep_entry_user_##NAME(d.defect, p, stream);
#include "models/_registry.inc"
#undef CORBO_HIP_USER_MODEL
#endif
    switch (d.dynamics) {
        case CORBO_HIP_DYN_VAN_DER_POL: return sweep_entry_vdp(d.defect, p, stream);
        case CORBO_HIP_DYN_SERIAL_INTEGRATOR:
            if (d.nx == 3) return sweep_entry_integ3(d.defect, p, stream);
            if (d.nx != 2) return false;
            return sweep_entry_integ2(d.defect, p, stream);
        case CORBO_HIP_DYN_UNICYCLE: return sweep_entry_unicycle(d.defect, p, stream);
        case CORBO_HIP_DYN_QUADROTOR: return sweep_entry_quadrotor(d.defect, p, stream);
        case CORBO_HIP_DYN_DUFFING: return sweep_entry_duffing(d.defect, p, stream);
        case CORBO_HIP_DYN_FREE_SPACE_ROCKET: return sweep_entry_rocket(d.defect, p, stream);
        case CORBO_HIP_DYN_SIMPLE_PENDULUM: return sweep_entry_pendulum(d.defect, p, stream);
        case CORBO_HIP_DYN_MASSLESS_PENDULUM: return sweep_entry_mpendulum(d.defect, p, stream);
        case CORBO_HIP_DYN_TOY_EXAMPLE: return sweep_entry_toy(d.defect, p, stream);
        case CORBO_HIP_DYN_ARTSTEINS_CIRCLE: return sweep_entry_artstein(d.defect, p, stream);
        case CORBO_HIP_DYN_CART_POLE: return sweep_entry_cartpole(d.defect, p, stream);
        case CORBO_HIP_DYN_PARALLEL_INTEGRATOR: return d.nx == 2 ? sweep_entry_par2(d.defect, p, stream) : sweep_entry_par3(d.defect, p, stream);
        case CORBO_HIP_DYN_LINEAR_STATE_SPACE:
            if (d.nx == 2 && d.nu == 1) return sweep_entry_lin21(d.defect, p, stream);
            if (d.nx == 2 && d.nu == 2) return sweep_entry_lin22(d.defect, p, stream);
            if (d.nx == 3 && d.nu == 1) return sweep_entry_lin31(d.defect, p, stream);
            if (d.nx == 3 && d.nu == 2) return sweep_entry_lin32(d.defect, p, stream);
            if (d.nx == 3 && d.nu == 3) return sweep_entry_lin33(d.defect, p, stream);
            if (d.nx == 4 && d.nu == 1) return sweep_entry_lin41(d.defect, p, stream);
            return false;
        default: return false;
    }
}

bool launch_hessian(const corbo_hip_problem_desc& d, const SweepParams& p, const HessParams& hp, hipStream_t stream)
{
#if __has_include("models/_registry.inc")
#define CORBO_HIP_USER_MODEL(NAME, SLOT, NX_, NU_, P0, P1, P2, P3) if (d.dynamics == CORBO_HIP_DYN_USER + SLOT) return hessian_entry_user_##NAME(d.defect, p, hp, stream);
#include "models/_registry.inc"
#undef CORBO_HIP_USER_MODEL
#endif
    switch (d.dynamics) {
        case CORBO_HIP_DYN_VAN_DER_POL: return hessian_entry_vdp(d.defect, p, hp, stream);
        case CORBO_HIP_DYN_SERIAL_INTEGRATOR:
            if (d.nx == 3) return hessian_entry_integ3(d.defect, p, hp, stream);
            if (d.nx != 2) return false;
            return hessian_entry_integ2(d.defect, p, hp, stream);
        case CORBO_HIP_DYN_UNICYCLE: return hessian_entry_unicycle(d.defect, p, hp, stream);
        case CORBO_HIP_DYN_QUADROTOR: return hessian_entry_quadrotor(d.defect, p, hp, stream);
        case CORBO_HIP_DYN_DUFFING: return hessian_entry_duffing(d.defect, p, hp, stream);
        case CORBO_HIP_DYN_FREE_SPACE_ROCKET: return hessian_entry_rocket(d.defect, p, hp, stream);
        case CORBO_HIP_DYN_SIMPLE_PENDULUM: return hessian_entry_pendulum(d.defect, p, hp, stream);
        case CORBO_HIP_DYN_MASSLESS_PENDULUM: return hessian_entry_mpendulum(d.defect, p, hp, stream);
        case CORBO_HIP_DYN_TOY_EXAMPLE: return hessian_entry_toy(d.defect, p, hp, stream);
        case CORBO_HIP_DYN_ARTSTEINS_CIRCLE: return hessian_entry_artstein(d.defect, p, hp, stream);
        case CORBO_HIP_DYN_CART_POLE: return hessian_entry_cartpole(d.defect, p, hp, stream);
        case CORBO_HIP_DYN_PARALLEL_INTEGRATOR: return d.nx == 2 ? hessian_entry_par2(d.defect, p, hp, stream) : hessian_entry_par3(d.defect, p, hp, stream);
        case CORBO_HIP_DYN_LINEAR_STATE_SPACE:
            if (d.nx == 2 && d.nu == 1) return hessian_entry_lin21(d.defect, p, hp, stream);
            if (d.nx == 2 && d.nu == 2) return hessian_entry_lin22(d.defect, p, hp, stream);
            if (d.nx == 3 && d.nu == 1) return hessian_entry_lin31(d.defect, p, hp, stream);
            if (d.nx == 3 && d.nu == 2) return hessian_entry_lin32(d.defect, p, hp, stream);
            if (d.nx == 3 && d.nu == 3) return hessian_entry_lin33(d.defect, p, hp, stream);
            if (d.nx == 4 && d.nu == 1) return hessian_entry_lin41(d.defect, p, hp, stream);
            return false;
        default: return false;
    }
}

bool launch_stage_jacobian_dump(const corbo_hip_problem_desc& d, const FactorParams& fp, const SweepParams& sp, double* jac_out, hipStream_t stream)
{
    if (!big_family_dims(d.nx, d.nu)) return false;
    if (d.dynamics == CORBO_HIP_DYN_QUADROTOR) return stage_entry_quadrotor(fp, sp, 0, jac_out, stream);
#if __has_include("models/_registry.inc")
#define CORBO_HIP_USER_MODEL(NAME, SLOT, NX_, NU_, P0, P1, P2, P3) if (d.dynamics == CORBO_HIP_DYN_USER + SLOT) return stage_entry_user_##NAME(fp, sp, 0, jac_out, stream);
#include "models/_registry.inc"
#undef CORBO_HIP_USER_MODEL
#endif
    return false;
}

bool launch_pass(const corbo_hip_problem_desc& d, const FactorParams& fp, const SweepParams& sp, hipStream_t stream)
{
#if __has_include("models/_registry.inc")
#define CORBO_HIP_USER_MODEL(NAME, SLOT, NX_, NU_, P0, P1, P2, P3) if (d.dynamics == CORBO_HIP_DYN_USER + SLOT) return pass_entry_user_##NAME(d.defect, fp, sp, stream);
#include "models/_registry.inc"
#undef CORBO_HIP_USER_MODEL
#endif
    switch (d.dynamics) {
        case CORBO_HIP_DYN_VAN_DER_POL: return pass_entry_vdp(d.defect, fp, sp, stream);
        case CORBO_HIP_DYN_SERIAL_INTEGRATOR:
            if (d.nx == 3) return pass_entry_integ3(d.defect, fp, sp, stream);
            if (d.nx != 2) return false;
            return pass_entry_integ2(d.defect, fp, sp, stream);
        case CORBO_HIP_DYN_UNICYCLE: return pass_entry_unicycle(d.defect, fp, sp, stream);
        case CORBO_HIP_DYN_DUFFING: return pass_entry_duffing(d.defect, fp, sp, stream);
        case CORBO_HIP_DYN_FREE_SPACE_ROCKET: return pass_entry_rocket(d.defect, fp, sp, stream);
        case CORBO_HIP_DYN_SIMPLE_PENDULUM: return pass_entry_pendulum(d.defect, fp, sp, stream);
        case CORBO_HIP_DYN_MASSLESS_PENDULUM: return pass_entry_mpendulum(d.defect, fp, sp, stream);
        case CORBO_HIP_DYN_TOY_EXAMPLE: return pass_entry_toy(d.defect, fp, sp, stream);
        case CORBO_HIP_DYN_ARTSTEINS_CIRCLE: return pass_entry_artstein(d.defect, fp, sp, stream);
        case CORBO_HIP_DYN_CART_POLE: return pass_entry_cartpole(d.defect, fp, sp, stream);
        case CORBO_HIP_DYN_PARALLEL_INTEGRATOR: return d.nx == 2 ? pass_entry_par2(d.defect, fp, sp, stream) : pass_entry_par3(d.defect, fp, sp, stream);
        case CORBO_HIP_DYN_LINEAR_STATE_SPACE:
            if (d.nx == 2 && d.nu == 1) return pass_entry_lin21(d.defect, fp, sp, stream);
            if (d.nx == 2 && d.nu == 2) return pass_entry_lin22(d.defect, fp, sp, stream);
            if (d.nx == 3 && d.nu == 1) return pass_entry_lin31(d.defect, fp, sp, stream);
            if (d.nx == 3 && d.nu == 2) return pass_entry_lin32(d.defect, fp, sp, stream);
            if (d.nx == 3 && d.nu == 3) return pass_entry_lin33(d.defect, fp, sp, stream);
            if (d.nx == 4 && d.nu == 1) return pass_entry_lin41(d.defect, fp, sp, stream);
            return false;
        default: return false;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Band factorisation: assemble H = J^T J + mu I and rhs = -J^T r from the stored Jacobian values through static product lists, Cholesky in
// band storage (natural parameter order), solve, trial iterate -- levenberg_marquardt_sparse.cpp:97-100, 135-161 for ANY sparsity of J whose
// normal matrix is banded (plus a free dt as a border).  One wave per instance: every elimination step updates the bw (bw + 1) / 2 trailing
// entries with one lane each; the steps are sequential (n of them), so this is the slow, general path -- the stage-parallel kernels cover
// the structures that matter for throughput.  Replaces Eigen::SimplicialLLT (:140-148) like factor_body does: ordering differences are
// rounding-level.
// ---------------------------------------------------------------------------------------------------------------------
// ---- reject-streak speculation (SpecParams, kernels.hpp): one workgroup, after every pass's sweep
__global__ __launch_bounds__(1024) void big_spec_kernel(const SpecParams p)
{
    constexpr int MAXG = 16;
    __shared__ int a_parent[MAXG];   // the instance that takes over the state of its group's slot a_take (this pass), or -1
    __shared__ int a_take[MAXG];
    __shared__ int a_fill[MAXG];     // 1 = fill the group's slots from its parent (new group, or a streak that outlasted its candidates)
    __shared__ int n_new, new_list[MAXG];
    const int tid = threadIdx.x, G = p.groups < MAXG ? p.groups : MAXG, SP = p.spec;
    const int slot0 = p.batch;
    if (p.mode == 1) {   // start of a solve
        for (int g = tid; g < G; g += 1024) { p.parent_of[g] = -1; p.rej_seen[g] = 0; }
        for (int i = tid; i < p.batch; i += 1024) p.prev_reject[i] = 0;
        for (int q = tid; q < G * SP; q += 1024) { p.st[slot0 + q].done = 1; p.slot_rej[q] = 0; }
        if (tid == 0 && p.adopted) *p.adopted = 0;
        return;
    }
    if (tid < MAXG) { a_parent[tid] = -1; a_take[tid] = 0; a_fill[tid] = 0; }
    if (tid == 0) n_new = 0;
    __syncthreads();
    // ---- A: groups in use -- what did the parent and its candidates do in this pass?
    if (tid < G && p.parent_of[tid] >= 0) {
        const int g = tid, P = p.parent_of[g];
        int uncount = 0;   // slots the sweep counted as unfinished
        for (int c = 0; c < SP; ++c) uncount += p.st[slot0 + g * SP + c].done ? 0 : 1;
        const bool parent_rejected = !p.st[P].done && p.st[P].n_reject > p.rej_seen[g];
        if (!parent_rejected) p.parent_of[g] = -1;   // its own pass ended the streak (or the solve): the candidates are void
        else {
            int take = SP - 1;
            bool goes_on = true;
            for (int c = 0; c < SP; ++c) {
                const LmState& sc = p.st[slot0 + g * SP + c];
                const bool rejected = !sc.done && sc.n_reject > p.slot_rej[g * SP + c];
                if (!rejected) { take = c; goes_on = false; break; }
            }
            a_parent[g] = P;
            a_take[g]   = take;
            if (goes_on) a_fill[g] = 1;   // every candidate rejected: the parent takes the last one's state, the next dampings are tried
            else p.parent_of[g] = -1;
        }
        if (p.counter && uncount) atomicSub(p.counter, uncount);
    }
    __syncthreads();
    // ---- B: instances whose step was rejected in this pass and that have no group yet
    for (int i = tid; i < p.batch; i += 1024) {
        const int nr = p.st[i].n_reject;
        const bool rejected_now = !p.st[i].done && nr > p.prev_reject[i];
        p.prev_reject[i] = nr;
        if (!rejected_now) continue;
        bool has_group = false;
        for (int g = 0; g < G; ++g) has_group = has_group || (p.parent_of[g] == i) || (a_parent[g] == i);
        if (!has_group) { const int q = atomicAdd(&n_new, 1); if (q < MAXG) new_list[q] = i; }
    }
    __syncthreads();
    if (tid == 0) {   // C: free groups for them (a group that hands a state over in this pass is busy until the copy is done: next pass)
        // Speculation pays when rejections are RARE (a few laggers hold the whole batch back): every candidate is a full instance of work in the
        // next pass.  With many instances rejecting at once the slowest one is not helped and the candidates only add work (64 quadrotor OCPs at N = 40,
        // 83 rejections: 1.69 -> 2.12 ms when every rejecting instance got candidates) -- at most max_parents streaks are followed at a time.
        const int SPEC_MAX_PARENTS = p.max_parents;   // (3 by default; tests lift it)
        int busy = 0;
        for (int g = 0; g < G; ++g) busy += (p.parent_of[g] >= 0 || a_parent[g] >= 0) ? 1 : 0;
        int nn = n_new < MAXG ? n_new : MAXG;
        // (option reject_speculation = 2 -- tests -- lifts the limit to the number of groups: whoever finds a free group gets it)
        if (busy + nn > SPEC_MAX_PARENTS) nn = (SPEC_MAX_PARENTS >= G && busy < G) ? G - busy : 0;
        int g = 0;
        for (int q = 0; q < nn; ++q) {
            while (g < G && (p.parent_of[g] >= 0 || a_parent[g] >= 0)) ++g;
            if (g >= G) break;
            p.parent_of[g] = new_list[q];
            a_fill[g] = 1;
            ++g;
        }
    }
    __syncthreads();
    auto copy_row = [&](double* base, size_t row_len, int dst, int src) {
        const double* s_ = base + (size_t)src * row_len;
        double* d_       = base + (size_t)dst * row_len;
        for (size_t i = tid; i < row_len; i += 1024) d_[i] = s_[i];
    };
    // ---- D: take over a candidate's state (iterate, end states of both buffers, residual buffers, chi2, LM state)
    for (int g = 0; g < G; ++g) {
        const int P = a_parent[g];
        if (P < 0) continue;
        const int src = slot0 + g * SP + a_take[g];
        // (the residual buffers are scratch of the sweep for this family -- the stage kernel recomputes what it needs -- and are not moved; of the
        // two end-state buffers the one paired with the candidate's accepted iterate is what the next stage kernel reads: both halves are 2 x N nx doubles)
        copy_row(p.x, p.nvs, P, src);
        copy_row(p.xe0, p.xe_row, P, src);
        copy_row(p.xe0, p.xe_row, p.batch_total + P, p.batch_total + src);
        copy_row(reinterpret_cast<double*>(p.st), sizeof(LmState) / sizeof(double), P, src);
        if (tid == 0) { p.chi2[P] = p.chi2[src]; if (p.adopted) *p.adopted += 1; }   // (one workgroup, groups in sequence: no atomic needed)
    }
    __syncthreads();
    // ---- E: fill slots: copies of the parent with the LM state it would have after 1, 2, ... more rejected steps
    for (int g = 0; g < G; ++g) {
        if (!a_fill[g]) continue;
        const int P = p.parent_of[g];
        const LmState s0 = p.st[P];
        const bool ok = !s0.done && (s0.inner + SP + 2 < LM_MAX_INNER) && !s0.no_trial;   // (near the inner-loop guard: no speculation)
        __syncthreads();
        if (!ok) { if (tid == 0) p.parent_of[g] = -1; continue; }
        for (int c = 0; c < SP; ++c) {
            const int dst = slot0 + g * SP + c;
            copy_row(p.x, p.nvs, dst, P);
            copy_row(p.lb, p.nvs, dst, P);
            copy_row(p.ub, p.nvs, dst, P);
            copy_row(p.xref, CORBO_HIP_MAX_NX, dst, P);
            copy_row(p.xe0, p.xe_row, dst, P);
            copy_row(p.xe0, p.xe_row, p.batch_total + dst, p.batch_total + P);
            if (tid == c) {
                LmState sc = s0;
                for (int r = 0; r <= c; ++r) {   // one more rejected pass: the chain's bookkeeping, then the sweep's reject branch (:204-213)
                    sc.mu_acc = (sc.fresh ? 0.0 : sc.mu_acc) + sc.mu;
                    sc.fresh  = 0;
                    sc.n_fact += 1; sc.inner += 1;
                    sc.n_res += 1;  sc.n_reject += 1;
                    sc.mu = sc.mu * sc.v;
                    sc.v  = 2 * sc.v;
                }
                p.st[dst] = sc;
                p.slot_rej[g * SP + c] = sc.n_reject;
            }
        }
        if (tid == 0) p.rej_seen[g] = s0.n_reject;
    }
    __syncthreads();
    // ---- slots of free groups do nothing in the next pass
    for (int q = tid; q < G * SP; q += 1024)
        if (p.parent_of[q / SP] < 0) p.st[slot0 + q].done = 1;
}

bool launch_big_spec(const SpecParams& p, hipStream_t stream)
{
    hipLaunchKernelGGL(big_spec_kernel, dim3(1), dim3(1024), 0, stream, p);
    return true;
}

size_t band_work_doubles(int nb, int bw) { return (size_t)nb * (bw + 1) + 2 * (size_t)nb + 8; }   // the band, then rhs [nb], border [nb], corner, rhs of dt
static size_t band_lds_doubles(int nb, int bw) { return (size_t)(bw + 1) * (bw + 1) + 2 * (size_t)nb + 32; }   // window, rhs, border, scratch [corner, rhs of dt, per-wave partial results]

// ---- assembly of H = J^T J and rhs = -J^T r from the static product lists, spread over the chip: grid (chunks, instances).  (Inside the factor kernel
//      -- four waves per instance -- it was a third of the launch: every product is an index load and two dependent value loads.)  Every position of
//      the band has a list (possibly empty): no zero fill.
__global__ __launch_bounds__(256) void band_assemble_kernel(const FactorParams p, const BandParams bp)
{
    const int inst = blockIdx.y + p.inst0, tid = threadIdx.x, chunk = blockIdx.x, nchunk = gridDim.x;
    const LmState* st = p.st + inst;
    if (st->done) return;
    const int n = bp.n, nb = bp.nb, W = bp.bw + 1;
    double* Hb = bp.work + (size_t)inst * bp.work_stride;
    double* gz = Hb + (size_t)nb * W;                         // rhs [nb], border [nb], corner, rhs of dt
    const double* val = (st->vbuf ? p.values1 : p.values0) + (size_t)inst * p.m_pad;
    const double* J   = p.jac + (size_t)inst * p.nnz_pad;
    // (the product lists are walked four products at a time: eight independent Jacobian loads in flight -- the sums keep their order)
    const int2* pairs2 = reinterpret_cast<const int2*>(bp.ent_pairs);
    for (int e = chunk * 256 + tid; e < bp.n_ent; e += nchunk * 256) {
        double acc = 0.0;
        int q = bp.ent_ptr[e];
        const int qe = bp.ent_ptr[e + 1];
        for (; q + 4 <= qe; q += 4) {
            const int2 i0 = pairs2[q], i1 = pairs2[q + 1], i2 = pairs2[q + 2], i3 = pairs2[q + 3];
            const double a0 = J[i0.x], b0 = J[i0.y], a1 = J[i1.x], b1 = J[i1.y], a2 = J[i2.x], b2 = J[i2.y], a3 = J[i3.x], b3 = J[i3.y];
            acc += a0 * b0; acc += a1 * b1; acc += a2 * b2; acc += a3 * b3;
        }
        for (; q < qe; ++q) { const int2 i0 = pairs2[q]; acc += J[i0.x] * J[i0.y]; }
        const int t = bp.ent_target[e];
        if (t >= 0) Hb[t] = acc;
        else if (t == INT32_MIN) gz[2 * nb] = acc;
        else gz[nb + (-1 - t)] = acc;
    }
    const int2* rent2 = reinterpret_cast<const int2*>(bp.rhs_ent);
    for (int c = chunk * 256 + tid; c < n; c += nchunk * 256) {
        double acc = 0.0;
        int q = bp.rhs_ptr[c];
        const int qe = bp.rhs_ptr[c + 1];
        for (; q + 4 <= qe; q += 4) {
            const int2 i0 = rent2[q], i1 = rent2[q + 1], i2 = rent2[q + 2], i3 = rent2[q + 3];
            const double a0 = J[i0.x], b0 = val[i0.y], a1 = J[i1.x], b1 = val[i1.y], a2 = J[i2.x], b2 = val[i2.y], a3 = J[i3.x], b3 = val[i3.y];
            acc -= a0 * b0; acc -= a1 * b1; acc -= a2 * b2; acc -= a3 * b3;
        }
        for (; q < qe; ++q) { const int2 i0 = rent2[q]; acc -= J[i0.x] * val[i0.y]; }
        if (c < nb) gz[c] = acc; else gz[2 * nb + 1] = acc;
    }
}

// (round 4) Four waves per instance and a SLIDING WINDOW: pivot j only touches rows j .. j + bw of the band, so those bw + 1 rows live in LDS (a ring of
// row slots, (bw + 1)^2 doubles) while the band itself stays in HBM -- a finished row is written out once, the row that enters the window is requested one
// pivot ahead.  Right-hand side and border column stay in LDS for the whole factorisation; the back-substitution walks the rows of L from the last one
// (x_i = y_i / L_ii, then y_c -= L_ic x_i for the row's band: contiguous rows, requested two ahead) instead of gathering columns.  Before: one wave,
// every pivot three dependent trips to wherever the band was (the 12-state quadrotor with a free dt at N = 100: 13 ms per factorisation).
__global__ __launch_bounds__(512) void band_factor_kernel(const FactorParams p, const BandParams bp)
{
    extern __shared__ __attribute__((aligned(16))) double band_smem[];
    __shared__ __attribute__((aligned(16))) LmState sl_;
    constexpr int T = 512, NWV = T / 64;   // (eight waves: the trailing update of a pivot -- up to 63 * 64 / 2 entries -- in two rounds)
    const int inst = blockIdx.x + p.inst0, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    LmState* st = &sl_;
    lm_state_in(st, p.st + inst, tid);
    __syncthreads();
    if (st->done) return;
    const int n = bp.n, nb = bp.nb, bw = bp.bw, W = bw + 1;
    const bool arrow = (nb < n);
    double* Hb  = bp.work + (size_t)inst * bp.work_stride;   // [nb][W]: column r - bw + d of row r at d; d = bw is the diagonal
    const double* gz = Hb + (size_t)nb * W;                   // what band_assemble_kernel left: rhs, border, corner, rhs of dt
    double* win = band_smem;                                  // [W][W] ring of row slots: row i lives in slot i % W
    double* g   = win + (size_t)W * W;                        // rhs -> y -> delta
    double* z   = g + nb;                                     // border column (free dt) -> L^-1 border
    double* sc  = z + nb;                                     // scratch: [0] corner, [1] rhs of dt, [2..] reductions
    const int fresh = st->fresh, first = st->first;
    int stop = st->stop;
    double mu = st->mu;
    const double mu_acc_in = st->mu_acc;
#define BAND_STAMP(i) do { if (p.timeline && blockIdx.x == 0 && tid == 0) p.timeline[i] = clock64(); } while (0)
    BAND_STAMP(0);
    // ---- right-hand side and border into LDS
    for (int i = tid; i < nb; i += T) { g[i] = gz[i]; z[i] = arrow ? gz[nb + i] : 0.0; }
    if (tid == 0) { sc[0] = arrow ? gz[2 * nb] : 0.0; sc[1] = arrow ? gz[2 * nb + 1] : 0.0; }
    __syncthreads();
    if (first) {   // mu = tau * max diag(J^T J), stop = |rhs|_inf <= eps1 (:115-118)
        double mx_d = -1e300, mx_g = 0.0;
        for (int c = tid; c < nb; c += T) { mx_d = fmax(mx_d, Hb[(size_t)c * W + bw]); mx_g = fmax(mx_g, fabs(g[c])); }
        mx_d = wave_max(mx_d); mx_g = wave_max(mx_g);
        if (lane == 0) { sc[2 + 2 * wave] = mx_d; sc[3 + 2 * wave] = mx_g; }
        __syncthreads();
        mx_d = sc[2]; mx_g = sc[3];
        for (int w = 1; w < NWV; ++w) { mx_d = fmax(mx_d, sc[2 + 2 * w]); mx_g = fmax(mx_g, sc[3 + 2 * w]); }
        if (arrow) { mx_d = fmax(mx_d, sc[0]); mx_g = fmax(mx_g, fabs(sc[1])); }
        stop = (mx_g <= LM_EPS1) ? 1 : 0;
        mu   = LM_TAU * mx_d;
        if (mu < 0) mu = 0;
        __syncthreads();
    }
    const double mu_eff = (fresh ? 0.0 : mu_acc_in) + mu;   // H_ii += mu on every inner pass, never undone on reject (:135-138)
    BAND_STAMP(1);   // assembled
    // ---- the window: rows 0 .. bw (with the damping on their diagonals), the next row requested
    for (int i = tid; i < W * W; i += T) {
        const int r = i / W, d = i - r * W;
        double v = (r < nb) ? Hb[(size_t)r * W + d] : 0.0;
        if (d == bw) v += mu_eff;
        win[i] = v;
    }
    // wave 7: lane d writes entry d of the finished row out and carries entry d of the row that enters next.  (The LAST wave: with the quadrotor's
    //  half-bandwidth of 27 the trailing update is 378 entries, waves 6 and 7 have none of it -- the wait for the requested row rides under the other
    //  waves' update.  Per-wave clock sums, cycles per pivot, with the loader on wave 3 and the right-hand-side updates on wave 2: update phase
    //  875 / 875 / 1160 / 1630 / 1030 / 1030 / 210 / 210 -- everybody waited for wave 3.)
    const bool loader = (wave == 7) && lane < W;
    int rnext = W;                                  // (rows 0 .. bw are in)
    double pre = 0.0;
    if (loader && rnext < nb) { pre = Hb[(size_t)rnext * W + lane]; if (lane == bw) pre += mu_eff; }
    double* dpiv = sc + 30;                         // the next pivot's diagonal, double-buffered (the pivot row's slot is recycled inside the phase)
    if (tid == 0) dpiv[0] = win[bw];
    __syncthreads();
    BAND_STAMP(2);   // window loaded
    // ---- band Cholesky (lower), forward substitution of rhs and border fused into the elimination.  ONE LDS-only barrier per pivot: column j is used
    //      unscaled inside phase j (every product scales its two factors itself: the same numbers as scaling first), its entries are scaled in place
    //      during phase j + 1 (wave 6), when nobody reads them any more; the pivot row is written out and its slot refilled by wave 7, and the next
    //      pivot's diagonal travels through a two-slot side buffer.  (LDS-only barriers: a full __syncthreads() waits for the wave's global operations
    //      too -- the write-out of row j and the request of the entering row would cost a memory round trip per pivot each.)
    // Index arithmetic out of the pivot loop (round 5: the loop was 2.1 k cycles per pivot, most of it integer work -- a square root, two corrections and
    // three modulo operations by the run-time window size per updated entry): entry q of the trailing triangle is (row j + 1 + a, column j + 1 + rem) for
    // EVERY pivot j, so (a, rem) and the three column offsets inside the rows are per-lane constants; only the rows' ring slots move, by one per pivot.
    constexpr int QMAX = 4;   // rounds of the trailing update: 63 * 64 / 2 entries at the largest half-bandwidth, 512 lanes
    const int ntri_full = bw * (bw + 1) / 2;
    int qa[QMAX], qoff_ij[QMAX], qoff_cj[QMAX], qoff_ic[QMAX], qsi[QMAX], qsc[QMAX];   // a, offsets of L(i,j) / L(c,j) / H(i,c) inside their rows, slot * W of rows i and c
#pragma unroll
    for (int r = 0; r < QMAX; ++r) {
        const int q = tid + r * T;
        int a = (int)((__builtin_sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);   // q = a (a + 1) / 2 + rem, rem <= a (single-precision estimate, corrected)
        while (a * (a + 1) / 2 > q) --a;
        while ((a + 1) * (a + 2) / 2 <= q) ++a;
        const int rem = q - a * (a + 1) / 2;
        qa[r] = (q < ntri_full) ? a : bw;   // (bw: never below any pivot's row count -- the entry does not exist)
        qoff_ij[r] = bw - (1 + a); qoff_cj[r] = bw - (1 + rem); qoff_ic[r] = bw - (a - rem);
        qsi[r] = ((1 + a) % W) * W; qsc[r] = ((1 + rem) % W) * W;
    }
    const int WW = W * W;
    int s6 = ((1 + lane) % W) * W;          // wave 6: slot * W of row j + 1 + lane
    int sj = 0;                             // slot * W of the pivot row j
    double inv_prev = 1.0;
    // The entering rows are requested TWO pivots ahead into two registers that take turns (the loop body is instantiated twice per iteration): a pivot is
    // ~ 600 cycles of LDS work, a row from HBM / L2 takes longer than that -- with one pivot of distance the loader wave was late at every barrier.  The
    // request itself is BRANCH-FREE (every lane of every wave issues it, the non-loader lanes for row 0: one cache line): a load inside a conditional
    // region makes the compiler wait for it at the join, i.e. at once.
    const int lcol = loader ? lane : 0;
    double preA = pre, preB = 0.0;
    {
        const int rb = (W + 1 < nb) ? W + 1 : 0;
        preB = Hb[(size_t)(loader ? rb : 0) * W + lcol];
        if (lcol == bw) preB += mu_eff;
    }
    auto pivot_step = [&](const int j, double& pre_now) {
        const bool live = (j < nb);                                            // (uniform; false only for the odd step behind the last pivot)
        const int cnt = live ? ((nb - 1 - j < bw) ? nb - 1 - j : bw) : 0;   // rows below the pivot inside the band
        double* rowj = win + sj;
        const double dj  = dpiv[j & 1];
        const double inv = rsqrt(dj);              // (v_rsq_f64 + one Newton step: half the dependent instructions of sqrt and a division)
        // the row that enters two pivots from now: requested first (see above; row index clamped, the value is only used when the row exists)
        const int rreq = rnext + 2;
        double pre_new = Hb[(size_t)((loader && rreq < nb) ? rreq : 0) * W + lcol];
        if (lcol == bw) pre_new += mu_eff;
        // trailing update: H(i, c) -= L(i, j) L(c, j), j < c <= i <= j + cnt; rhs / border: g_i -= L(i, j) y_j
#pragma unroll
        for (int r = 0; r < QMAX; ++r) {
            if (r * T >= ntri_full) break;   // (uniform: rounds that hold no entry at this half-bandwidth cost nothing -- 27: one round)
            if (qa[r] < cnt) {
                double* ri = win + qsi[r];
                const double lij = ri[qoff_ij[r]] * inv, lcj = win[qsc[r] + qoff_cj[r]] * inv;
                const double v = ri[qoff_ic[r]] - lij * lcj;
                ri[qoff_ic[r]] = v;
                if (r == 0 && tid == 0) dpiv[(j + 1) & 1] = v;     // (i = c = j + 1: the next pivot's diagonal)
            }
            qsi[r] += W; if (qsi[r] == WW) qsi[r] = 0;
            qsc[r] += W; if (qsc[r] == WW) qsc[r] = 0;
        }
        if (wave == 6) {
            if (lane < cnt) {   // rhs / border updates of the rows below the pivot
                const int i = j + 1 + lane;
                const double lij = win[s6 + bw - (1 + lane)] * inv;
                g[i] -= lij * (g[j] * inv);
                if (arrow) z[i] -= lij * (z[j] * inv);
            }
            // column j - 1, rows j + 1 .. : scaled in place now (phase j - 1 used it unscaled; row j's entry leaves with the row, see the loader)
            if (live && j > 0 && lane < bw) {
                const int i = j + 1 + lane;
                if (i < nb && lane + 2 <= bw) win[s6 + bw - (lane + 2)] *= inv_prev;
            }
            if (live && lane == 63 && j > 0) { g[j - 1] *= inv_prev; if (arrow) z[j - 1] *= inv_prev; }   // y_{j-1}, (L^-1 border)_{j-1}
            s6 += W; if (s6 == WW) s6 = 0;
        }
        if (loader && live) {
            // row j is final: L(j, j - bw .. j) goes out (its entry in column j - 1 scaled on the way; the diagonal INVERTED: the back-substitution
            // multiplies instead of dividing), the row that enters for the next pivot takes its slot.  (The write-out is issued BEHIND the request above:
            // gfx9 returns vector-memory operations in order, a wait for a load issued behind a store waits for the store's acknowledgement too.)
            const double old = rowj[lane];
            Hb[(size_t)j * W + lane] = (lane == bw) ? inv : ((lane == bw - 1) ? old * inv_prev : old);
            if (rnext < nb) rowj[lane] = pre_now;
        }
        ++rnext;
        pre_now = pre_new;
        if (live) inv_prev = inv;
        sj += W; if (sj == WW) sj = 0;
        lds_barrier();   // (what crosses it lives in LDS: the row written out and the rows requested stay in flight)
    };
    for (int j = 0; j < nb; j += 2) {
        pivot_step(j, preA);
        pivot_step(j + 1, preB);
    }
    if (tid == 64 && nb > 0) g[nb - 1] *= inv_prev;
    if (tid == 65 && arrow && nb > 0) z[nb - 1] *= inv_prev;
    __syncthreads();
    BAND_STAMP(3);   // factorised
    double y2 = 0.0, zz = 0.0, zy = 0.0;
    for (int c = tid; c < nb; c += T) { y2 += g[c] * g[c]; if (arrow) { zz += z[c] * z[c]; zy += z[c] * g[c]; } }
    y2 = wave_sum(y2); zz = wave_sum(zz); zy = wave_sum(zy);
    if (lane == 0) { sc[2 + 3 * wave] = y2; sc[3 + 3 * wave] = zz; sc[4 + 3 * wave] = zy; }
    __syncthreads();
    y2 = 0.0; zz = 0.0; zy = 0.0;
    for (int w = 0; w < NWV; ++w) { y2 += sc[2 + 3 * w]; zz += sc[3 + 3 * w]; zy += sc[4 + 3 * w]; }
    double ddt = 0.0;
    if (arrow) {   // the last pivot: H(dt, dt) + damping - |z|^2
        const double piv  = (sc[0] + mu_eff) - zz;
        const double linv = 1.0 / sqrt(piv);
        const double ydt  = (sc[1] - zy) * linv;
        y2 += ydt * ydt;
        ddt = ydt * linv;
        for (int c = tid; c < nb; c += T) g[c] -= z[c] * ddt;
    }
    __syncthreads();
    // ---- back-substitution L^T delta = y, row by row from the last one (wave 0; rows of L from HBM, two ahead).  The part of the right-hand side a row can
    //      touch -- the bw + 1 entries up to its own -- rides in REGISTERS, lane d = column i - bw + d of the current row i: one multiply for x_i, one
    //      multiply-add for the others, then everything moves one lane up (a wave-wide DPP shift) and the next entry enters at lane 0.  (Round 5: with
    //      the vector in LDS every row was a dependent LDS read-modify-write round trip and a division: 420 -> 350 -> ~ 100 cycles per row.)
    if (wave == 0) {
        __threadfence_block();
        const bool on = lane < W;
        // (the rows of L come from HBM / L2: eight of them in flight -- with two the loop waited ~ 350 cycles per row for its operand)
        constexpr int PF = 8;
        // (BRANCH-FREE loads: a load inside a divergent or conditional region makes the compiler wait for every outstanding memory operation at the join --
        //  measured here: one s_waitcnt vmcnt(0) per row, the prefetch depth did not matter.  Indices are clamped instead, the duplicates are harmless.)
        const int lc = on ? lane : 0;
        double rr[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) { const int ir = nb - 1 - u; rr[u] = Hb[(size_t)(ir >= 0 ? ir : 0) * W + lc]; }
        const int c0 = nb - 1 - bw + lane;
        double a   = (on && c0 >= 0 && nb >= 1) ? g[c0] : 0.0;
        double gin = g[nb - 2 - bw >= 0 ? nb - 2 - bw : 0];   // the entry that enters with the next row (same address in every lane)
        if (nb - 2 - bw < 0) gin = 0.0;
        for (int i0 = nb - 1; i0 >= 0; i0 -= PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int i = i0 - u;
                const bool act = (i >= 0);                              // (uniform; false only behind row 0 in the last chunk)
                const double row = (on && act) ? rr[u] : 0.0;
                rr[u] = Hb[(size_t)(i - PF >= 0 ? i - PF : 0) * W + lc];
                const double xi = lane_bcast(a * row, bw);          // x_i = y_i / L(i, i): the row carries 1 / L(i, i) at its diagonal   (every lane the same number)
                if (lane == bw && act) g[i] = xi;
                const double upd = a - row * xi;                    // lane d < bw: y_c -= L(i, c) x_i
                const double kept = (lane < bw) ? upd : 0.0;
                const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(kept), 0x138, 0xF, 0xF, false);   // wave_shr:1 -- lane d takes lane d - 1's value
                const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(kept), 0x138, 0xF, 0xF, false);
                const double moved = (lane == 0) ? gin : __hiloint2double(hi, lo);
                a = act ? moved : a;
                const int ig = i - 2 - bw;
                const double gl = g[ig >= 0 ? ig : 0];
                gin = (ig >= 0) ? gl : 0.0;
            }
        }
    }
    __syncthreads();
    BAND_STAMP(4);   // back-substituted
    // ---- trial iterate x + delta (applyIncrementNonFixed, vertex_set.cpp:357-367), step norms
    const double* xin = p.x + (size_t)inst * p.nvs;
    double* xt        = p.xt + (size_t)inst * p.nvs;
    double* dl        = p.delta_out ? p.delta_out + (size_t)inst * p.nvs : nullptr;
    for (int v = tid; v < p.nvs; v += T) { xt[v] = xin[v]; if (dl) dl[v] = 0.0; }
    __syncthreads();
    double dn2 = 0.0;
    for (int c = tid; c < n; c += T) {
        const double d = (c < nb) ? g[c] : ddt;
        const int v    = bp.param_voff[c];
        xt[v] = xin[v] + d;
        if (dl) dl[v] = d;
        dn2 += d * d;
    }
    dn2 = wave_sum(dn2);
    if (lane == 0) sc[2 + wave] = dn2;
    __syncthreads();
    if (tid == 0) {
        dn2 = 0.0;
        for (int w = 0; w < NWV; ++w) dn2 += sc[2 + w];
        st->mu     = mu;
        st->mu_acc = mu_eff;
        st->first  = 0;
        st->fresh  = 0;
        st->n_fact += 1;
        st->inner += 1;
        const double dnorm = sqrt(dn2);
        st->dnorm = dnorm;
        int no_trial;
        if (dnorm <= LM_EPS2) { stop = 1; no_trial = 1; }                    // :151-154
        else { no_trial = 0; st->den = mu * dn2 + y2; }                      // delta^T (mu delta + rhs), delta^T rhs = |y|^2
        st->stop     = stop;
        st->no_trial = no_trial;
    }
    BAND_STAMP(5);
#undef BAND_STAMP
    __syncthreads();
    lm_state_out(p.st + inst, st, tid);
}

// ---- (round 5) NARROW bands, half-bandwidth <= 7: ONE WAVE per instance, the 8 x 8 window of the elimination in REGISTERS (a ring of row / column slots,
//      both triangles), no barrier, no LDS traffic in the pivot loop.  The small-block families with integral-form constraint edges / the control-deviation
//      term are exactly this case (unicycle: five parameters per stage, half-bandwidth 7; 495 pivots per factorisation): with the eight-wave kernel above a
//      pivot is ~ 1.7 k cycles of barriers and LDS round trips whatever the bandwidth (the headline batch with a rate limit on the controls: 18.3 ms per solve
//      against 0.48 ms without).  Per pivot here: the pivot and the right-hand side's entry by v_readlane, column p to the rows and columns by one round of
//      lane permutations, one multiply-subtract; the finished pivot's row and column slots take the row that enters (requested 24 pivots ahead, branch-free).
//      Same products and the same order of operations per entry as band_factor_kernel; L leaves in the same row form (diagonal inverted), so the
//      back-substitution is that kernel's, verbatim.  (docs/measurements/r05.md 9: the versions measured on the way.)
template <bool ARROW>
__global__ __launch_bounds__(64) void band_narrow_kernel(const FactorParams p, const BandParams bp)
{
    extern __shared__ __attribute__((aligned(16))) double nar_smem[];
    __shared__ __attribute__((aligned(16))) LmState sl_;
    const int inst = blockIdx.x + p.inst0, lane = threadIdx.x;
    LmState* st = &sl_;
    lm_state_in(st, p.st + inst, lane);
    __syncthreads();
    if (st->done) return;
    const int n = bp.n, nb = bp.nb, bw = bp.bw, W = bw + 1;
    constexpr bool arrow = ARROW;                             // (nb < n: a free dt as a border)
    double* Hb  = bp.work + (size_t)inst * bp.work_stride;   // [nb][W]: column r - bw + d of row r at d; d = bw is the diagonal
    const double* gz = Hb + (size_t)nb * W;                   // what band_assemble_kernel left: rhs, border, corner, rhs of dt
    double* g = nar_smem;                                     // y -> delta
    double* z = g + nb;                                       // L^-1 border (free dt)
    const int fresh = st->fresh, first = st->first;
    int stop = st->stop;
    double mu = st->mu;
    const double mu_acc_in = st->mu_acc;
    const double corner = arrow ? gz[2 * nb] : 0.0, gdt = arrow ? gz[2 * nb + 1] : 0.0;
    if (first) {   // mu = tau * max diag(J^T J), stop = |rhs|_inf <= eps1 (:115-118)
        double mx_d = -1e300, mx_g = 0.0;
        for (int c = lane; c < nb; c += 64) { mx_d = fmax(mx_d, Hb[(size_t)c * W + bw]); mx_g = fmax(mx_g, fabs(gz[c])); }
        mx_d = wave_max(mx_d); mx_g = wave_max(mx_g);
        if (arrow) { mx_d = fmax(mx_d, corner); mx_g = fmax(mx_g, fabs(gdt)); }
        stop = (mx_g <= LM_EPS1) ? 1 : 0;
        mu   = LM_TAU * mx_d;
        if (mu < 0) mu = 0;
    }
    const double mu_eff = (fresh ? 0.0 : mu_acc_in) + mu;   // H_ii += mu on every inner pass, never undone on reject (:135-138)
#define NAR_STAMP(i) do { if (p.timeline && blockIdx.x == 0 && lane == 0) p.timeline[i] = clock64(); } while (0)
    NAR_STAMP(0); NAR_STAMP(1);
    // ---- the window: a RING -- row p + i of the band lives in row slot (p + i) & 7, column p + j in column slot (p + j) & 7; lane 8 a + b holds the
    //      entry (row slot a, column slot b), both triangles.  Nothing moves between pivots: the finished pivot's row and column slots take the row that
    //      enters.  (A window that is shifted up its diagonal every pivot puts a second lane permutation on the dependent chain: measured 950 cycles per
    //      pivot incl. the back-substitution against ~ 1.7 k of the eight-wave kernel; the ring: one permutation round per pivot.)
    const int wa = lane >> 3, wb = lane & 7;
    const int whi = wa > wb ? wa : wb, wdd = wa > wb ? wa - wb : wb - wa;
    double w = 0.0;
    {
        const bool in = (wdd <= bw) && (whi < nb);
        const double v = Hb[(size_t)(in ? whi : 0) * W + (in ? bw - wdd : 0)];
        w = in ? v : 0.0;
        if (in && wdd == 0) w += mu_eff;
    }
    double gi = gz[wa < nb ? wa : 0], zi = arrow ? gz[nb + (wa < nb ? wa : 0)] : 0.0;   // lane (a, b): the right-hand side's / the border's entry of the row in slot a
    if (wa >= nb) { gi = 0.0; zi = 0.0; }
    // the row that enters behind pivot p is row p + 8: requested PD chunks of eight pivots ahead, element t of the band row by lane t (branch-free, clamped),
    // already in the form the window takes it in: zero for a row behind the band's end and in the lanes beyond the row's length, the damping on its diagonal.
    // (The band was written by another kernel on other compute units: it comes from the fabric-side cache or HBM, 1.5 - 2 us away.)
    constexpr int PF = 8, PD = 3;
    const int et = lane <= bw ? lane : 0;
    const double emu = (lane == bw) ? mu_eff : 0.0;
    // (the loaded values are NOT touched here: an operation on a value that is still in flight makes the wave wait for it at once -- measured: 585 -> 811
    //  cycles per pivot with the masking done at the request; it is done where the row is used, PD chunks later)
    auto fetch_row = [&](int r, double& vr, double& vg, double& vz) {
        const int rc = r < nb ? r : 0;
        vr = Hb[(size_t)rc * W + et]; vg = gz[rc];
        if constexpr (arrow) vz = gz[nb + rc]; else vz = 0.0;
    };
    double pr[PD][PF], pg[PD][PF], pz[PD][PF];
#pragma unroll
    for (int q = 0; q < PD; ++q)
#pragma unroll
        for (int u = 0; u < PF; ++u) fetch_row(8 + q * PF + u, pr[q][u], pg[q][u], pz[q][u]);
    // per-lane constants of the eight pivot slots: where column p of the window is (for this lane's row / column), whether the lane's entry belongs to the
    // pivot's row / column slot (it takes the entering row then) and which element of that row it takes (lane 63 -- always zero -- for an entry outside the band)
    int a_row[PF], a_col[PF], a_ent[PF], ia_[PF];
    bool m_row[PF], m_any[PF], m_st[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int ia = (wa - u) & 7, ib = (wb - u) & 7;
        const bool erow = (ia == 0), ecol = (ib == 0);
        const int io = erow ? ib : ia;
        const int eoff = (io == 0) ? 0 : 8 - io;
        a_row[u] = ((lane & 56) | u) << 2; a_col[u] = ((u << 3) | wb) << 2;
        a_ent[u] = ((eoff <= bw) ? bw - eoff : 63) << 2;
        m_row[u] = erow; m_any[u] = erow || ecol; m_st[u] = ecol && ia <= bw; ia_[u] = ia;
    }
    auto perm = [](int addr, double v) {
        const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
        const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
        return __hiloint2double(hi, lo);
    };
    double y2 = 0.0, zz = 0.0, zy = 0.0;
    NAR_STAMP(2);
    bool more = nb > 0;
    for (int p0 = 0; more; p0 += PD * PF) {
#pragma unroll
        for (int q = 0; q < PD; ++q)
#pragma unroll
        for (int u = 0; u < PF; ++u) {   // (PF = 8: the pivot's slot is the compile-time constant u)
            const int pv = p0 + q * PF + u;
            if (pv >= nb) { more = false; break; }                  // (uniform)
            double nr, ng, nz;
            fetch_row(pv + 8 + PD * PF, nr, ng, nz);
            const double d   = lane_bcast(w, 9 * u);
            const double inv = rsqrt(d);                            // (v_rsq_f64 + one Newton step, like band_factor_kernel)
            const double ci = perm(a_row[u], w), cj = perm(a_col[u], w);   // H'(row of slot a, p), H'(row of slot b, p)
            const bool rin = pv + 8 < nb;                           // (uniform) the entering row exists
            const double ev = perm(a_ent[u], (rin && lane <= bw) ? pr[q][u] + emu : 0.0);   // (off the dependent chain)
            const double li = ci * inv, lj = cj * inv;
            const double wu = w - li * lj;
            const double y  = lane_bcast(gi, 8 * u) * inv;
            const double gu = gi - li * y;
            // column p of L leaves in ROW form: L(p + i, p) sits at offset bw - i of row p + i; the diagonal inverted (the back-substitution multiplies)
            if (m_st[u] && pv + ia_[u] < nb) Hb[(size_t)(pv + ia_[u]) * W + bw - ia_[u]] = (ia_[u] == 0) ? inv : li;
            if (lane == 0) g[pv] = y;
            y2 += y * y;
            if constexpr (arrow) {
                const double zp = lane_bcast(zi, 8 * u) * inv, zu = zi - li * zp;
                if (lane == 0) z[pv] = zp;
                zz += zp * zp; zy += zp * y;
                zi = m_row[u] ? (rin ? pz[q][u] : 0.0) : zu;
            }
            w  = m_any[u] ? ev : wu;
            gi = m_row[u] ? (rin ? pg[q][u] : 0.0) : gu;
            pr[q][u] = nr; pg[q][u] = ng; pz[q][u] = nz;   // (in place: the slot is due again PD chunks from now)
        }
    }
    NAR_STAMP(3);
    double ddt = 0.0;
    __threadfence_block();
    if (arrow) {   // the last pivot: H(dt, dt) + damping - |z|^2
        const double piv  = (corner + mu_eff) - zz;
        const double linv = 1.0 / sqrt(piv);
        const double ydt  = (gdt - zy) * linv;
        y2 += ydt * ydt;
        ddt = ydt * linv;
        for (int c = lane; c < nb; c += 64) g[c] -= z[c] * ddt;
    }
    __threadfence();
    __syncthreads();
    // ---- back-substitution L^T delta = y: band_factor_kernel's (rows of L from HBM / L2, eight in flight, the touched part of the vector in registers)
    {
        const bool on = lane < W;
        const int lc = on ? lane : 0;
        double rr[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) { const int ir = nb - 1 - u; rr[u] = Hb[(size_t)(ir >= 0 ? ir : 0) * W + lc]; }
        const int c0 = nb - 1 - bw + lane;
        double a   = (on && c0 >= 0 && nb >= 1) ? g[c0] : 0.0;
        // the entry that enters at lane 0 behind row i is y of column i - 1 - bw -- untouched by the rows above it: read eight rows ahead like the rows of L
        // (one row ahead, its LDS round trip sat on every row's dependent chain)
        double gq[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) { const int ig = nb - 1 - u - 1 - bw; gq[u] = g[ig >= 0 ? ig : 0]; }
        for (int i0 = nb - 1; i0 >= 0; i0 -= PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int i = i0 - u;
                const bool act = (i >= 0);
                const double row = (on && act) ? rr[u] : 0.0;
                rr[u] = Hb[(size_t)(i - PF >= 0 ? i - PF : 0) * W + lc];
                const double xi = lane_bcast(a * row, bw);
                if (lane == bw && act) g[i] = xi;
                const double upd = a - row * xi;
                const double kept = (lane < bw) ? upd : 0.0;
                const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(kept), 0x138, 0xF, 0xF, false);   // wave_shr:1
                const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(kept), 0x138, 0xF, 0xF, false);
                const double gin = (i - 1 - bw >= 0) ? gq[u] : 0.0;
                const double moved = (lane == 0) ? gin : __hiloint2double(hi, lo);
                a = act ? moved : a;
                const int ig = i - PF - 1 - bw;
                gq[u] = g[ig >= 0 ? ig : 0];
            }
        }
    }
    __syncthreads();
    NAR_STAMP(4);
    // ---- trial iterate x + delta (applyIncrementNonFixed, vertex_set.cpp:357-367), step norms
    const double* xin = p.x + (size_t)inst * p.nvs;
    double* xt        = p.xt + (size_t)inst * p.nvs;
    double* dl        = p.delta_out ? p.delta_out + (size_t)inst * p.nvs : nullptr;
    for (int v = lane; v < p.nvs; v += 64) { xt[v] = xin[v]; if (dl) dl[v] = 0.0; }
    __syncthreads();
    double dn2 = 0.0;
    for (int c = lane; c < n; c += 64) {
        const double d = (c < nb) ? g[c] : ddt;
        const int v    = bp.param_voff[c];
        xt[v] = xin[v] + d;
        if (dl) dl[v] = d;
        dn2 += d * d;
    }
    dn2 = wave_sum(dn2);
    if (lane == 0) {
        st->mu     = mu;
        st->mu_acc = mu_eff;
        st->first  = 0;
        st->fresh  = 0;
        st->n_fact += 1;
        st->inner += 1;
        const double dnorm = sqrt(dn2);
        st->dnorm = dnorm;
        int no_trial;
        if (dnorm <= LM_EPS2) { stop = 1; no_trial = 1; }                    // :151-154
        else { no_trial = 0; st->den = mu * dn2 + y2; }                      // delta^T (mu delta + rhs), delta^T rhs = |y|^2
        st->stop     = stop;
        st->no_trial = no_trial;
    }
    NAR_STAMP(5);
#undef NAR_STAMP
    __syncthreads();
    lm_state_out(p.st + inst, st, lane);
}

static constexpr size_t BAND_LDS_MAX = 160 * 1024 - 256;   // (the kernel also has 128 bytes of static LDS: the LM state)
// what band_factor_kernel can take: one wave writes a finished row out / walks a row in the back-substitution (half-bandwidth <= 63), and the sliding
// window + right-hand side + border live in LDS.  corbo_hip_create asks, so that an unsupported descriptor is refused THERE (include/corbo_hip.h).
// (host mirror of BtLayout / bt_carve_doubles / launch_bt_t for corbo_hip_create; nc: an upper bound of the dynamics' cache doubles per grid state)
int bt_route_max_rounds(int nx, int nu, bool arrow, int N, int nnz_pad, int m_pad, int nvs)
{
    const int s = nx + nu;
    if (nx < 1 || nx > 4 || nu < 1 || N > 256) return 0;
    const int szp = (2 * s * s + s + (arrow ? s : 0)) | 1;
    long carve = (long)N * szp + 3;
    if (carve < (long)nnz_pad + m_pad + 2 - nvs) carve = (long)nnz_pad + m_pad + 2 - nvs;
    if (carve < (long)nnz_pad + (long)N * 8) carve = (long)nnz_pad + (long)N * 8;
    carve = (carve + 1) / 2 * 2;
    const size_t lds = sizeof(double) * (size_t)(carve + nvs + 12) + sizeof(LmState);
    if (lds > (size_t)160 * 1024) return 0;
    const int epb = s * (s + 1) / 2 + s * s + s + (arrow ? s : 0);
    (void)epb;
    return N > 128 ? 8 : 4;   // super-rounds of the product lists (BtLayout::max_rounds: sixteen rounds -- the BIG instantiation: thirty-two --, four each)
}

bool band_route_supported(int nb, int bw) { return bw + 1 <= 64 && sizeof(double) * (band_lds_doubles(nb, bw) + 8) <= BAND_LDS_MAX; }

bool launch_band_factor(const FactorParams& fp, const BandParams& bp, hipStream_t stream)
{
    const size_t lds = sizeof(double) * (band_lds_doubles(bp.nb, bp.bw) + 8);   // window + rhs + border + scratch (a horizon of 256 twelve-state intervals: 85 KB)
    static unsigned long long attr_set = 0;   // (per device)
    constexpr size_t LDS_MAX = BAND_LDS_MAX;
    if (first_on_device(attr_set)) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(band_factor_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) (void)hipGetLastError(); }
    if (!bp.work || !band_route_supported(bp.nb, bp.bw)) return false;
    {
        // list entries per thread: an entry is a chain of dependent loads (index, pair, two values).  Small batches are latency-bound -- two per thread (one OCP:
        // 5.09 -> 4.78 ms per extra-edge solve); from a few hundred instances on the chip is full either way and fewer, longer threads win (1024: 9.8 vs 10.2 ms)
        const int per_wg = (fp.batch >= 256 || bp.bw > 7) ? 2048 : 512;   // (wide bands -- long product lists per entry -- lose with the finer split at every batch size)
        int chunks = (bp.n_ent + per_wg - 1) / per_wg;
        if (chunks > 64) chunks = 64;
        if (chunks < 1) chunks = 1;
        hipLaunchKernelGGL(band_assemble_kernel, dim3(chunks, fp.batch), dim3(256), 0, stream, fp, bp);
    }
    // half-bandwidth <= 7: one wave per instance, window in registers (band_narrow_kernel); option "band_wide" keeps the eight-wave kernel (A/B, tests)
    const size_t lds_n = sizeof(double) * (2 * (size_t)bp.nb + 8);
    if (bp.bw <= 7 && !fp.band_wide && lds_n <= BAND_LDS_MAX) {
        static unsigned long long attr_n = 0;   // (per device)
        if (first_on_device(attr_n)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(band_narrow_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) (void)hipGetLastError();
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(band_narrow_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX) != hipSuccess) (void)hipGetLastError();
        }
        if (bp.nb < bp.n) hipLaunchKernelGGL(band_narrow_kernel<true>, dim3(fp.batch), dim3(64), lds_n, stream, fp, bp);
        else hipLaunchKernelGGL(band_narrow_kernel<false>, dim3(fp.batch), dim3(64), lds_n, stream, fp, bp);
        return true;
    }
    hipLaunchKernelGGL(band_factor_kernel, dim3(fp.batch), dim3(512), lds, stream, fp, bp);
    return true;
}

size_t big_stage_cache_doubles(const corbo_hip_problem_desc& d, int N)
{
    if (!big_family_dims(d.nx, d.nu)) return 0;
    const int half = (5 * d.nx + 2 * (d.nx + d.nu) + d.nx * (d.nx + d.nu) + 8 + d.nx * d.nx + 1) & ~1;   // BigLds::HALF (checked there)
    return (size_t)((N + 1) / 2) * 2 * (size_t)half;
}

bool launch_factor(const corbo_hip_problem_desc& d, const FactorParams& p, hipStream_t stream, const SweepParams* sp)
{
    if (big_family_dims(d.nx, d.nu)) {   // big-block family: the model's own unit
        if (!sp) return false;
        if (d.dynamics == CORBO_HIP_DYN_QUADROTOR) return factor_entry_quadrotor(p, *sp, stream);
#if __has_include("models/_registry.inc")
#define CORBO_HIP_USER_MODEL(NAME, SLOT, NX_, NU_, P0, P1, P2, P3) if (d.dynamics == CORBO_HIP_DYN_USER + SLOT) return factor_entry_user_##NAME(p, *sp, stream);
#include "models/_registry.inc"
#undef CORBO_HIP_USER_MODEL
#endif
        return false;
    }
    if (d.nx == 2 && d.nu == 1) return launch_factor_t<2, 1>(p, stream);
    if (d.nx == 3 && d.nu == 2) return launch_factor_t<3, 2>(p, stream);
    if (d.nx == 3 && d.nu == 1) return launch_factor_t<3, 1>(p, stream);
    if (d.nx == 4 && d.nu == 1) return launch_factor_t<4, 1>(p, stream);
    if (d.nx == 2 && d.nu == 2) return launch_factor_t<2, 2>(p, stream);
    if (d.nx == 3 && d.nu == 3) return launch_factor_t<3, 3>(p, stream);
    return false;
}

#endif  // !CORBO_HIP_DYN_TU

}  // namespace corbo_hip
