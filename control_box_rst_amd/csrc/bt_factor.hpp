// bt_factor.hpp -- block-tridiagonal factor phase of the run-to-completion kernel for the small-block families WITH extra edges
// (control-deviation term, integral-form constraint edges).  Included by kernels.hip inside namespace corbo_hip { namespace {.
//
// With a control-deviation edge on (u_k, u_{k-1}) (nlp_functions.cpp:117-131,152-186) or an integral-form constraint edge on
// (x_k, u_k, x_{k+1}) (finite_differences_collocation_edges.h:149-459) the controls of a stage no longer couple to x_k / x_{k+1} alone, so
// factor_body's "controls first" elimination does not apply; but H = J^T J (levenberg_marquardt_sparse.cpp:97-100) is still BLOCK TRIDIAGONAL in the
// stage blocks z_k = (x_k, u_k) of S = nx + nu rows (the last block: x_f, padded), plus a dense border for a free dt.  This phase
//   (1) assembles the blocks from the Jacobian values the sweep phase left in LDS: the dynamics-defect edges (two thirds of the products) from
//       their dense local Jacobians [A | B | C] by a lane pair per stage, every other row of J (cost, bound, inequality rows, the extra edges of
//       whatever kind) through static product lists (BtTables, structure.hpp: an entry of H or of rhs = -J^T r is a sum of products of two
//       operands of the array [J | values | 0]; the lists are padded to equal length per super-round -- ELL layout, coalesced, branch-free),
//   (2) factors by block cyclic reduction on S x S blocks: level h eliminates the blocks k = h (2t + 1), S lanes per block (lane j: column j
//       of the couplings), D_k^-1 applied through a Cholesky factor each lane computes redundantly,
//   (3) back-substitutes down the tree, closes the arrowhead (free dt: the border rides as a second right-hand side) and writes the trial
//       iterate x + delta into the LDS array the next sweep phase evaluates.
// Replaces Eigen::SimplicialLLT + solve (levenberg_marquardt_sparse.cpp:140-158) like factor_body does; elimination-order differences are rounding level.
#pragma once

template <int S, bool ARROW>
struct BtLayout {
    static constexpr int A   = 0;              // D_k (S x S, lower part valid)  -> after elimination: W_a = D_k^-1 H(k, k - h)   [row * S + column]
    static constexpr int B   = S * S;          // F_k = H(next remaining block, k) [row (next) * S + column (k)] -> W_b = D_k^-1 H(k, k + h)
    static constexpr int G   = 2 * S * S;      // rhs -> D_k^-1 rhs -> delta
    static constexpr int Z   = 2 * S * S + S;  // border column (free dt) -> D_k^-1 border -> H^-1 border
    static constexpr int SZ  = 2 * S * S + S + (ARROW ? S : 0);
    static constexpr int SZP = SZ | 1;         // odd stride: consecutive blocks spread over the LDS banks
    static constexpr int EPB = S * (S + 1) / 2 + S * S + S + (ARROW ? S : 0);   // assembled entries per block
    static constexpr int NB_MAX = 128;         // blocks (= grid points) of the regular instantiation (a lane PAIR per stage, sixteen list rounds); BIG: 256
    // rounds of the product lists a lane holds in registers: the lists cover every row of J but the defect edges' (diagonals, right-hand sides, extra edges,
    // inequality rows) -- sixteen entries per lane at most; a structure that needs more stays on the band route (corbo_hip_create asks bt_route_max_rounds)
    __host__ __device__ static constexpr int max_rounds(bool big) { return big ? 32 : 16; }
    __host__ __device__ static constexpr int carve(int nb) { return nb * SZP + 3; }   // + corner, rhs of dt, trash slot
};

// flags of BtTables::target (bits 28..31; bits 0..27: the LDS slot -- an entry of the padding points at the trash slot behind the two scalars)
constexpr unsigned BT_DIAG = 1u << 28, BT_ONE = 1u << 29, BT_RHS = 1u << 30, BT_CORNER = 1u << 31, BT_SLOT = 0x0FFFFFFFu;

// BIG: horizons of 129 .. 256 grid points -- one lane per stage does both halves of the defect edge's assembly, twice the list rounds; one workgroup per CU (LDS)
template <int S, int NX, bool ARROW, int THREADS, bool BIG = false>
__device__ __forceinline__ void bt_factor_body(const FactorParams& p, LmState* const st, double* const smem, double* const xs, double* const red, const int inst, const int tid, const bool j_in_lds,
                                               const int eq_stride, const int eq_defect_off)
{
    using BL = BtLayout<S, ARROW>;
    constexpr int SZP = BL::SZP, MAXE = BL::max_rounds(BIG), NW = THREADS / 64;
    constexpr int oA = BL::A, oB = BL::B, oG = BL::G, oZ = BL::Z;
    const int NB = p.N;
    const int done = st->done, fresh = st->fresh, first = st->first, vbuf = st->vbuf;
    int stop = st->stop;
    double mu = st->mu;
    const double mu_acc_in = st->mu_acc;
    lds_barrier();   // everybody has read the state
    if (done) return;
#define BT_STAMP(id) do { if (p.timeline && inst == p.timeline_inst && tid == 0) p.timeline[id] = clock64(); } while (0)
    BT_STAMP(0);
    // the defect edge of stage ks is assembled by TWO lanes from its dense local Jacobian (below); its column offsets depend on nothing but the lane: requested first
    constexpr int WL = S + NX, HALF = BIG ? THREADS : THREADS / 2;
    static_assert((BIG ? 2 : 1) * BL::NB_MAX <= HALF, "a lane (pair) per stage");
    const int ks = tid & (HALF - 1), half = BIG ? 0 : tid / HALF;
    constexpr bool BOTH = BIG;   // the lane of a stage does both halves
    const bool st_on = ks < NB - 1;
    // The factorisation after a REJECTED step (the Jacobian was not refreshed: j_in_lds is false) has the same J and the same right-hand side as the one before it:
    // the assembled blocks of that one -- damping included -- come back from HBM / L2 (FactorParams::bt_snap: written below by every factorisation that assembles)
    // and take this pass's mu on top of their diagonal, the reference's own H_ii += mu (levenberg_marquardt_sparse.cpp:135-138; never undone, :208).
    const bool reuse = !j_in_lds && !first && p.bt_snap != nullptr;
    double* const snap = p.bt_snap ? p.bt_snap + (size_t)inst * p.bt_snap_stride : nullptr;
    const int snap2 = (NB * SZP + 3 + 1) / 2;   // double2 words of the block storage (the carve is even)
    int sco[WL + 1];
#pragma unroll
    for (int c = 0; c < WL + 1; ++c) sco[c] = reuse ? 0 : p.stage_cols[st_on ? ks : 0].col[c];
    // ... and so do the first steps of the product lists (a window of four 32-byte word groups per lane, kept in flight across the list loop below)
    const uint4* const pw = reinterpret_cast<const uint4*>(p.bt_pairs);
    const char* const Jb = reinterpret_cast<const char*>(smem);
    auto fetch = [&](int step, uint4& lo, uint4& hi) { const uint4* q = pw + ((size_t)step * THREADS + tid) * 2; lo = q[0]; hi = q[1]; };   // (the table is four steps longer than its last step)
    uint4 w0a, w0b, w1a, w1b, w2a, w2b, w3a, w3b;
    if (!reuse) { fetch(0, w0a, w0b); fetch(1, w1a, w1b); fetch(2, w2a, w2b); fetch(3, w3a, w3b); }
    else {
        w0a = w0b = w1a = w1b = w2a = w2b = w3a = w3b = uint4{0u, 0u, 0u, 0u};
        // the blocks of the previous factorisation straight into the block storage (nothing of the sweep phase lives there any more; no operands are staged on this path):
        // the headline horizon's 44 KB in ONE round trip, eleven 16-byte loads per lane, in flight next to the table loads below
        const double2* s2 = reinterpret_cast<const double2*>(snap);
        double2* b2       = reinterpret_cast<double2*>(smem);
        constexpr int UN = 12;
        for (int i0 = tid; i0 < snap2; i0 += THREADS * UN) {
            double2 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) { const int i = i0 + u * THREADS; v[u] = s2[i < snap2 ? i : snap2 - 1]; }
#pragma unroll
            for (int u = 0; u < UN; ++u) { const int i = i0 + u * THREADS; if (i < snap2) b2[i] = v[u]; }
        }
    }
    // ---- (1) operands [J | values | 0] in LDS: the Jacobian is there after an accepted step (the sweep phase of this pass assembled it), after a
    //      rejected one it is staged again from HBM / L2; the residual paired with it always comes from HBM / L2 (10 KB, written by this workgroup)
    double* const Jv = smem;
    if (!j_in_lds && !reuse) {
        const double2* src = reinterpret_cast<const double2*>(p.jac + (size_t)inst * p.nnz_pad);
        double2* dst       = reinterpret_cast<double2*>(Jv);
        const int n2       = p.nnz_pad / 2;
        for (int i0 = tid; i0 < n2; i0 += THREADS * 4) {
            double2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * THREADS; v[u] = src[i < n2 ? i : n2 - 1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * THREADS; dst[i < n2 ? i : n2 - 1] = v[u]; }
        }
    }
    if (!reuse) {
        const double* val = (vbuf ? p.values1 : p.values0) + (size_t)inst * p.m_pad;
        double* dst       = Jv + p.nnz_pad;
        for (int i0 = tid; i0 < p.m; i0 += THREADS * 4) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * THREADS; v[u] = val[i < p.m ? i : p.m - 1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * THREADS; dst[i < p.m ? i : p.m - 1] = v[u]; }
        }
        if (tid == 0) dst[p.m_pad] = 0.0;   // the operand the padding of the product lists points to
    }
    lds_barrier();
    BT_STAMP(1);
    // ---- (2a) the product lists into registers (the block storage overlays the operands: nothing is written before every lane is through with them).  Entry
    //      e of the table = lane e % THREADS, round e / THREADS; FOUR rounds form a super-round whose lists are padded to one length, a multiple of four: step i
    //      of a super-round is the i-th product of the lane's four entries -- eight operand BYTE OFFSETS into the LDS array (two 16-byte loads: no address
    //      arithmetic between the table and the LDS reads) and four independent multiply-adds.  The loads depend on nothing but the step: a window of FOUR
    //      steps is in flight with STATIC registers -- the loop body is four steps, each reloads the registers it has just consumed.  (A rotating window,
    //      w0 = w1; w1 = w2; ..., does not pipeline: a register move of a value that is still in flight waits for it.)
    constexpr int MAXSR = (MAXE + 3) / 4;
    double acc[4 * MAXSR];
    unsigned tg[4 * MAXSR];
    const int R = p.bt_rounds;            // super-rounds of this handle
    int step = 0;
#pragma unroll
    for (int sr = 0; sr < MAXSR; ++sr) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) tg[4 * sr + c] = (unsigned)(NB * SZP + 2);   // (no entry: the trash slot)
        if (sr < R) {
            const uint4 t4 = reinterpret_cast<const uint4*>(p.bt_target)[(size_t)sr * THREADS + tid];
            tg[4 * sr] = t4.x; tg[4 * sr + 1] = t4.y; tg[4 * sr + 2] = t4.z; tg[4 * sr + 3] = t4.w;
            const int s1 = reuse ? 0 : p.bt_off[sr + 1];
            // (per step: eight LDS operands in ONE batch -- the empty asm takes all of them, so all eight reads are issued before the first product)
            auto products = [&](const uint4& lo, const uint4& hi) {
                auto ld = [&](unsigned off) { return *reinterpret_cast<const double*>(Jb + off); };
                double x0 = ld(lo.x), x1 = ld(lo.y), x2 = ld(lo.z), x3 = ld(lo.w), x4 = ld(hi.x), x5 = ld(hi.y), x6 = ld(hi.z), x7 = ld(hi.w);
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
                a0 += x0 * x1; a1 += x2 * x3; a2 += x4 * x5; a3 += x6 * x7;
            };
            // (the scheduling barriers pin the reloads where they stand: the machine scheduler sinks them to the end of the body otherwise -- next to their
            //  uses in the NEXT iteration --, and the window is one step deep instead of four)
            for (; step < s1; step += 4) {
                products(w0a, w0b); fetch(step + 4, w0a, w0b); __builtin_amdgcn_sched_barrier(0);
                products(w1a, w1b); fetch(step + 5, w1a, w1b); __builtin_amdgcn_sched_barrier(0);
                products(w2a, w2b); fetch(step + 6, w2a, w2b); __builtin_amdgcn_sched_barrier(0);
                products(w3a, w3b); fetch(step + 7, w3a, w3b); __builtin_amdgcn_sched_barrier(0);
            }
        }
        acc[4 * sr]     = (tg[4 * sr] & BT_RHS) ? -a0 : a0;
        acc[4 * sr + 1] = (tg[4 * sr + 1] & BT_RHS) ? -a1 : a1;
        acc[4 * sr + 2] = (tg[4 * sr + 2] & BT_RHS) ? -a2 : a2;
        acc[4 * sr + 3] = (tg[4 * sr + 3] & BT_RHS) ? -a3 : a3;
    }
    BT_STAMP(2);
    // ---- (2b) the defect edges -- two thirds of the products of H = J^T J -- from their DENSE local Jacobians: the lane pair (ks, ks + HALF) of stage ks takes
    //      G = [A | B | C] (NX x (S + NX): x_k, u_k, x_{k+1}) and the edge's residual out of the operand area into registers now, and forms its share of
    //      G^T G once the operands are dead: half 0 the diagonal block of stage ks and its right-hand side, half 1 the coupling F_ks = H(x_{ks+1}, (x, u)_ks)
    //      and, in a second write phase, what the edge adds to the NEXT block (C^T C, -C^T r).  (Through the product lists these were 2 x 132 scattered LDS
    //      reads per stage, and the LDS pipe is what the three workgroups of a CU share.)
    // H_ii += mu on every inner pass, never undone on reject (:135-138 and the comment at :208); the first factorisation of a solve learns its mu below
    double mu_eff = first ? 0.0 : (fresh ? 0.0 : mu_acc_in) + mu;
    double* const blk = smem;
    if (!reuse) {
    double G[NX][WL], rv[NX], dc[NX];
#pragma unroll
    for (int c = 0; c < WL; ++c) {
        const int o = sco[c];
#pragma unroll
        for (int r = 0; r < NX; ++r) { const double v = Jv[(o >= 0 ? o : 0) + r]; G[r][c] = (st_on && o >= 0) ? v : 0.0; }
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) {
        const double v = Jv[p.nnz_pad + p.eq_row0 + (st_on ? ks : 0) * eq_stride + eq_defect_off + r];
        rv[r] = st_on ? v : 0.0;
        const int o = sco[WL];
        const double d = Jv[(ARROW && o >= 0 ? o : 0) + r];
        dc[r] = (ARROW && st_on && o >= 0) ? d : 0.0;
    }
    lds_barrier();   // every lane is through with the operands
    // ---- write phase X: complete diagonal blocks (lower part), right-hand sides, borders and couplings of the stages, plain stores
    if (st_on && half == 0) {
        double* sk = blk + ks * SZP;
#pragma unroll
        for (int i = 0; i < S; ++i) {
#pragma unroll
            for (int c = 0; c <= i; ++c) {
                double v = 0.0;
#pragma unroll
                for (int r = 0; r < NX; ++r) v += G[r][i] * G[r][c];
                sk[oA + i * S + c] = v;
            }
            double g = 0.0, z = 0.0;
#pragma unroll
            for (int r = 0; r < NX; ++r) { g -= G[r][i] * rv[r]; z += G[r][i] * dc[r]; }
            sk[oG + i] = g;
            if constexpr (ARROW) sk[oZ + i] = z;
        }
    }
    if (half == 0 && ks == NB - 1) {   // the last block (x_f): nothing but what its neighbour and the lists add
        double* sk = blk + ks * SZP;
#pragma unroll
        for (int i = 0; i < S; ++i) {
#pragma unroll
            for (int c = 0; c <= i; ++c) sk[oA + i * S + c] = 0.0;
            sk[oG + i] = 0.0;
            if constexpr (ARROW) sk[oZ + i] = 0.0;
        }
    }
    if (st_on && (BOTH || half == 1)) {
        double* sk = blk + ks * SZP;
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
            for (int c = 0; c < S; ++c) {
                double v = 0.0;
                if (i < NX) {
#pragma unroll
                    for (int r = 0; r < NX; ++r) v += G[r][S + (i < NX ? i : 0)] * G[r][c];
                }
                sk[oB + i * S + c] = v;
            }
    }
    if (ARROW && tid == THREADS - 1) { blk[NB * SZP] = 0.0; blk[NB * SZP + 1] = 0.0; }
    lds_barrier();
    // ---- write phase Y: what the edge of stage ks adds to block ks + 1
    if (st_on && (BOTH || half == 1)) {
        double* sn = blk + (ks + 1) * SZP;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int c = 0; c <= i; ++c) {
                double v = 0.0;
#pragma unroll
                for (int r = 0; r < NX; ++r) v += G[r][S + i] * G[r][S + c];
                sn[oA + i * S + c] += v;
            }
            double g = 0.0, z = 0.0;
#pragma unroll
            for (int r = 0; r < NX; ++r) { g -= G[r][S + i] * rv[r]; z += G[r][S + i] * dc[r]; }
            sn[oG + i] += g;
            if constexpr (ARROW) sn[oZ + i] += z;
        }
    }
    double cdt = 0.0, gdtp = 0.0;   // (free dt) the edges' share of the corner H(dt, dt) and of rhs(dt)
    if constexpr (ARROW) {
        if (st_on && half == 0) {
#pragma unroll
            for (int r = 0; r < NX; ++r) { cdt += dc[r] * dc[r]; gdtp -= dc[r] * rv[r]; }
        }
    }
    lds_barrier();
    // ---- write phase T: the sums of the product lists on top (every other row of J: cost, bound, inequality rows, the extra edges), the damping
    // (every entry of H / rhs is the target of exactly one list, the padding's targets are the trash slot: all reads first -- one LDS round trip --, then the writes)
    {
        double cur[4 * MAXSR];
#pragma unroll
        for (int q = 0; q < 4 * MAXSR; ++q) cur[q] = (q < 4 * R) ? blk[tg[q] & BT_SLOT] : 0.0;
#pragma unroll
        for (int q = 0; q < 4 * MAXSR; ++q) {
            if (q < 4 * R) {   // (uniform)
                double v = cur[q] + acc[q];
                v += (tg[q] & BT_DIAG) ? mu_eff : 0.0;
                v = (tg[q] & BT_ONE) ? 1.0 : v;       // a fixed component / a pad row of the last block: an identity row, its increment is zero
                blk[tg[q] & BT_SLOT] = v;
            }
        }
    }
    lds_barrier();
    if constexpr (ARROW) {
        const double s0 = wave_sum(cdt), s1 = wave_sum(gdtp);
        if ((tid & 63) == 0) { red[2 * (tid >> 6)] = s0; red[2 * (tid >> 6) + 1] = s1; }
        lds_barrier();
        if (tid == 0) {
            double c0 = 0.0, g0 = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { c0 += red[2 * w]; g0 += red[2 * w + 1]; }
            blk[NB * SZP] += c0;
            blk[NB * SZP + 1] += g0;
        }
        lds_barrier();
    }
    // ---- first factorisation of a solve: mu = tau * max diag(J^T J), stop = |rhs|_inf <= eps1 (:115-118) -- from the assembled entries; then the damping
    if (first) {
        double mx_d = -1e300, mx_g = 0.0;
#pragma unroll
        for (int q = 0; q < 4 * MAXSR; ++q) {
            if (q < 4 * R) {   // (uniform)
                const double v = blk[tg[q] & BT_SLOT];
                if (tg[q] & (BT_DIAG | BT_CORNER)) mx_d = fmax(mx_d, v);
                if (tg[q] & BT_RHS) mx_g = fmax(mx_g, fabs(v));
            }
        }
        mx_d = wave_max(mx_d);
        mx_g = wave_max(mx_g);
        if ((tid & 63) == 0) { red[2 * (tid >> 6)] = mx_d; red[2 * (tid >> 6) + 1] = mx_g; }
        lds_barrier();
        mx_d = red[0]; mx_g = red[1];
#pragma unroll
        for (int w = 1; w < NW; ++w) { mx_d = fmax(mx_d, red[2 * w]); mx_g = fmax(mx_g, red[2 * w + 1]); }
        stop = (mx_g <= LM_EPS1) ? 1 : 0;
        mu   = LM_TAU * mx_d;
        if (mu < 0) mu = 0;
        mu_eff = (fresh ? 0.0 : mu_acc_in) + mu;
#pragma unroll
        for (int q = 0; q < 4 * MAXSR; ++q)
            if (q < 4 * R && (tg[q] & BT_DIAG)) blk[tg[q] & BT_SLOT] += mu_eff;
        lds_barrier();
    }
    if (snap) {   // the assembled blocks, for the factorisation that follows a rejected step (stores nothing waits for; the reads precede the elimination's first writes)
        const double2* b2 = reinterpret_cast<const double2*>(blk);
        double2* s2       = reinterpret_cast<double2*>(snap);
        for (int i0 = tid; i0 < snap2; i0 += THREADS * 4) {
            double2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * THREADS; v[u] = b2[i < snap2 ? i : snap2 - 1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * THREADS; if (i < snap2) s2[i] = v[u]; }
        }
        lds_barrier();
    }
    }
    else {
        // ---- after a rejected step: the blocks of the previous factorisation (requested at the top of the phase), this pass's mu on top of their diagonal (the
        //      identity rows of fixed components stay)
        lds_barrier();
        {   // (reads first, then the writes: the diagonal slots are distinct, every other entry goes through the trash slot)
            double cur[4 * MAXSR];
#pragma unroll
            for (int q = 0; q < 4 * MAXSR; ++q) {
                const bool d = (q < 4 * R) && (tg[q] & BT_DIAG) && !(tg[q] & BT_ONE);
                cur[q] = blk[d ? (tg[q] & BT_SLOT) : (unsigned)(NB * SZP + 2)];
            }
#pragma unroll
            for (int q = 0; q < 4 * MAXSR; ++q) {
                const bool d = (q < 4 * R) && (tg[q] & BT_DIAG) && !(tg[q] & BT_ONE);
                if (d) {
                    const unsigned slot = tg[q] & BT_SLOT;
                    const double v = cur[q] + mu;
                    blk[slot]  = v;
                    snap[slot] = v;   // (a streak of rejected steps: the next one adds its mu to this)
                }
            }
        }
        lds_barrier();
    }
    BT_STAMP(3);
    // ---- (3) block cyclic reduction.  Level h: the blocks k = h (2 t + 1) are eliminated, neighbours a = k - h, b = k + h (remaining blocks).
    //      Lane (t, j): Cholesky of D_k (redundant in the S lanes), column j of W_a = D_k^-1 H(k, a) and W_b = D_k^-1 H(k, b), column j of the Schur
    //      updates of D_a, D_b and of the new coupling H(b, a) = -H(b, k) W_a.  Phase R reads, W1 writes slot k, slot a; W2 slot b (every remaining
    //      block is the left neighbour of one eliminated block and the right neighbour of another: two write phases, no conflicts).
    constexpr int GR = THREADS / S;   // blocks per round
    const int grp = tid / S, j = tid - grp * S;
    const bool lane_on = grp < GR;
    double gy = 0.0, gz = 0.0, zz = 0.0;   // pivot sums: g^T D^-1 g, g^T D^-1 b, b^T D^-1 b (this lane's terms)
    for (int lg = 0; (1 << lg) < NB; ++lg) {   // h = 2^lg (shifts: a division by the runtime stride is forty instructions)
        const int h = 1 << lg;
        const int cnt = (((NB - 1) >> lg) + 1) >> 1;
        for (int t0 = 0; t0 < cnt; t0 += GR) {
            const int t   = t0 + grp;
            const bool on = lane_on && t < cnt;
            const int k = on ? ((2 * t + 1) << lg) : 0, a = on ? k - h : 0, b = k + h;
            const bool has_b = on && b < NB;
            double wa[S], wb[S], dDa[S], dDb[S], fn[S];
            double yj = 0.0, zj = 0.0, dga = 0.0, dgb = 0.0, dba = 0.0, dbb = 0.0;
            if (on) {
                // Two halves with a scheduling fence between them: (1) factor D_k, the right-hand side(s), column j of W_a and W_b -- the factor and four
                // vectors live; (2) the Schur products, the couplings streamed from LDS -- the factor is dead by then.  (Without the fence the compiler
                // clusters every LDS read of both halves at the top: 180+ live registers, 72 spilled at the 168 of three workgroups per CU.)
                const double* sk = blk + k * SZP;
                const double* sa = blk + a * SZP;
                double yh[S], zh[S];
                {
                    double L[S][S];
#pragma unroll
                    for (int i = 0; i < S; ++i)
#pragma unroll
                        for (int c = 0; c < S; ++c) L[i][c] = (c <= i) ? sk[oA + i * S + c] : 0.0;
#pragma unroll
                    for (int r = 0; r < S; ++r) {
                        yh[r] = sk[oG + r]; zh[r] = ARROW ? sk[oZ + r] : 0.0;
                        wa[r] = sa[oB + r * S + j];                                            // H(k, a)[r][j]
                        const double f = sk[oB + j * S + r];                                   // H(k, b)[r][j] = H(b, k)[j][r]
                        wb[r] = has_b ? f : 0.0;
                    }
                    chol_inv<S>(L);
                    double gk[S], bk[S];
#pragma unroll
                    for (int r = 0; r < S; ++r) { gk[r] = yh[r]; bk[r] = zh[r]; }
                    // (the three / four solves back to back: independent dependent chains the scheduler interleaves)
                    fwd_solve_vec<S>(L, wa); fwd_solve_vec<S>(L, wb); fwd_solve_vec<S>(L, yh);
                    if constexpr (ARROW) fwd_solve_vec<S>(L, zh);
                    bwd_solve_vec<S>(L, wa); bwd_solve_vec<S>(L, wb); bwd_solve_vec<S>(L, yh);
                    if constexpr (ARROW) bwd_solve_vec<S>(L, zh);
                    // (D_k^-1 is symmetric: H(k, a)[:, j] . (D_k^-1 g) = (D_k^-1 H(k, a)[:, j]) . g -- the updates of the neighbours' right-hand sides need W and the raw g)
                    double gkj = 0.0, bkj = 0.0;
#pragma unroll
                    for (int r = 0; r < S; ++r) {
                        dga += wa[r] * gk[r]; dgb += wb[r] * gk[r];
                        if constexpr (ARROW) { dba += wa[r] * bk[r]; dbb += wb[r] * bk[r]; }
                        gkj = (r == j) ? gk[r] : gkj; bkj = (r == j) ? bk[r] : bkj;
                        yj = (r == j) ? yh[r] : yj; zj = (r == j) ? zh[r] : zj;
                    }
                    gy += gkj * yj;
                    if constexpr (ARROW) { gz += gkj * zj; zz += bkj * zj; }
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < S; ++i) { dDa[i] = 0.0; dDb[i] = 0.0; fn[i] = 0.0; }
#pragma unroll
                for (int r = 0; r < S; ++r)
#pragma unroll
                    for (int i = 0; i < S; ++i) dDa[i] += sa[oB + r * S + i] * wa[r];          // (H(k, a)^T W_a)[i][j]
                if (has_b) {
#pragma unroll
                    for (int i = 0; i < S; ++i)
#pragma unroll
                        for (int r = 0; r < S; ++r) {
                            const double f = sk[oB + i * S + r];                               // H(b, k)[i][r]
                            dDb[i] += f * wb[r];                                               // (H(k, b)^T W_b)[i][j]
                            fn[i] -= f * wa[r];                                                // new H(b, a)[i][j]
                        }
                }
            }
            lds_barrier();
            if (on) {
                double* sk = blk + k * SZP;
                double* sa = blk + a * SZP;
#pragma unroll
                for (int r = 0; r < S; ++r) { sk[oA + r * S + j] = wa[r]; sk[oB + r * S + j] = wb[r]; }
                sk[oG + j] = yj;
                if constexpr (ARROW) sk[oZ + j] = zj;
#pragma unroll
                for (int i = 0; i < S; ++i)
                    if (i >= j) sa[oA + i * S + j] -= dDa[i];
                sa[oG + j] -= dga;
                if constexpr (ARROW) sa[oZ + j] -= dba;
                if (has_b) {
#pragma unroll
                    for (int i = 0; i < S; ++i) sa[oB + i * S + j] = fn[i];
                }
            }
            lds_barrier();
            if (has_b) {
                double* sb = blk + b * SZP;
#pragma unroll
                for (int i = 0; i < S; ++i)
                    if (i >= j) sb[oA + i * S + j] -= dDb[i];
                sb[oG + j] -= dgb;
                if constexpr (ARROW) sb[oZ + j] -= dbb;
            }
            lds_barrier();
        }
    }
    BT_STAMP(4);
    // (the accepted iterate of this lane's components: requested here, consumed behind the back-substitution)
    const double* xin = p.x + (size_t)inst * p.nvs;
    double xv_pre[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int v = tid + u * THREADS; xv_pre[u] = xin[v <= p.off_dt ? v : p.off_dt]; }
    // ---- root: block 0 is what is left
    if (tid < S) {
        double L[S][S], x[S], z[S], g0[S], b0[S];
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
            for (int c = 0; c < S; ++c) L[i][c] = (c <= i) ? blk[oA + i * S + c] : 0.0;
#pragma unroll
        for (int r = 0; r < S; ++r) { g0[r] = blk[oG + r]; b0[r] = ARROW ? blk[oZ + r] : 0.0; x[r] = g0[r]; z[r] = b0[r]; }
        chol_inv<S>(L);
        fwd_solve_vec<S>(L, x); bwd_solve_vec<S>(L, x);
        if constexpr (ARROW) { fwd_solve_vec<S>(L, z); bwd_solve_vec<S>(L, z); }
        gy += g0[j] * x[j];
        if constexpr (ARROW) { gz += g0[j] * z[j]; zz += b0[j] * z[j]; }
        blk[oG + j] = x[j];                     // (the S lanes sit in one wave: its LDS operations are in order, every read above precedes these writes)
        if constexpr (ARROW) blk[oZ + j] = z[j];
    }
    lds_barrier();
    BT_STAMP(5);
    // ---- back-substitution down the tree: x_k = D_k^-1 g_k - W_a x_a - W_b x_b (lane (t, i): row i); the border's column likewise
    int lgtop = 0;
    while ((2 << lgtop) < NB) ++lgtop;
    for (int lg = lgtop; lg >= 0; --lg) {
        const int h = 1 << lg;
        const int cnt = (((NB - 1) >> lg) + 1) >> 1;
        for (int t0 = 0; t0 < cnt; t0 += GR) {
            const int t = t0 + grp;
            if (lane_on && t < cnt) {
                const int k = (2 * t + 1) << lg, a = k - h, b = k + h;
                double* sk = blk + k * SZP;
                const double* sa = blk + a * SZP;
                const double* sb = blk + (b < NB ? b : a) * SZP;
                const bool has_b = b < NB;
                double x = sk[oG + j], z = ARROW ? sk[oZ + j] : 0.0;
#pragma unroll
                for (int c = 0; c < S; ++c) {
                    const double w1 = sk[oA + j * S + c], w2 = has_b ? sk[oB + j * S + c] : 0.0;
                    x -= w1 * sa[oG + c];
                    x -= w2 * sb[oG + c];
                    if constexpr (ARROW) { z -= w1 * sa[oZ + c]; z -= w2 * sb[oZ + c]; }
                }
                sk[oG + j] = x;
                if constexpr (ARROW) sk[oZ + j] = z;
            }
        }
        lds_barrier();
    }
    BT_STAMP(6);
    // ---- pivot sums, the arrowhead's last pivot (free dt), trial iterate x + delta (applyIncrementNonFixed, vertex_set.cpp:357-367), step norms
    {
        const double s0 = wave_sum(gy), s1 = ARROW ? wave_sum(gz) : 0.0, s2 = ARROW ? wave_sum(zz) : 0.0;
        if ((tid & 63) == 0) {
            red[tid >> 6] = s0;
            if constexpr (ARROW) { red[NW + (tid >> 6)] = s1; }
        }
        lds_barrier();
        gy = red[0];
        if constexpr (ARROW) gz = red[NW];
#pragma unroll
        for (int w = 1; w < NW; ++w) { gy += red[w]; if constexpr (ARROW) gz += red[NW + w]; }
        lds_barrier();
        if constexpr (ARROW) {
            if ((tid & 63) == 0) red[tid >> 6] = s2;
            lds_barrier();
            zz = red[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) zz += red[w];
            lds_barrier();
        }
    }
    double y2 = gy, ddt = 0.0;
    if constexpr (ARROW) {
        const double corner = blk[NB * SZP], gdt = blk[NB * SZP + 1];
        const double piv = (corner + mu_eff) - zz;
        const double num = gdt - gz;
        ddt = num / piv;
        y2 += num * ddt;
    }
    double dn2 = 0.0;
    const int off_xf = (NB - 1) * S;
    int vround = 0;
    for (int v = tid; v < p.nvs; v += THREADS, ++vround) {
        double d = 0.0;
        const bool in_stage = v < off_xf, in_xf = !in_stage && v < off_xf + NX;
        if (in_stage || in_xf) {
            const int k = in_stage ? v / S : NB - 1, e = v - k * S;
            d = blk[k * SZP + oG + e];
            if constexpr (ARROW) d -= blk[k * SZP + oZ + e] * ddt;
        }
        else if (v == p.off_dt) d = ddt;
        const double xv = (vround == 0) ? xv_pre[0] : ((vround == 1) ? xv_pre[1] : xin[v <= p.off_dt ? v : p.off_dt]);
        xs[v] = (v <= p.off_dt) ? xv + d : 0.0;
        dn2 += d * d;
    }
    {
        const double a0 = wave_sum(dn2);
        if ((tid & 63) == 0) red[tid >> 6] = a0;
        lds_barrier();
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
            else { no_trial = 0; st->den = mu * dn2 + y2; }                      // delta^T (mu delta + rhs), delta^T rhs = the pivot sums
            st->stop     = stop;
            st->no_trial = no_trial;
        }
    }
    BT_STAMP(7);
#undef BT_STAMP
}
