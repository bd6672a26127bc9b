// corbo-hip-stage: name=control_norm slot=1 kind=control_ineq
//
// A USER stage inequality on the CONTROLS, dropped into csrc/stage_functions/: an input-magnitude bound
//     c(u) = u[0]^2 + ... + u[nu-1]^2 - prm[0]^2 <= 0          (summed left to right)
// -- the NON-INTEGRAL CONTROL TERM (dimension 1) of a user's corbo::StageInequalityConstraint subclass (getNonIntegralControlTermDimension /
// computeNonIntegralControlTerm: one UnaryVectorVertexEdge on u_k per interval behind the state term's edge, nlp_functions.cpp:82-89).  Host class:
// oracle/ref_driver.cpp (UserStageInequalities, unorm=); fixtures sf_*_unorm*.json.  Public id CORBO_HIP_STAGE_FN_USER + slot in
// corbo_hip_problem_desc::stage_ineq_control, parameters in ineq_control_params.  An edge on u_k alone: the sweep kernel's extra-edge
// instantiation evaluates it, the block-tridiagonal / band routes factorise (it adds to the diagonal block of the stage only).
template <> struct StageFunction<1> {
    static constexpr int KIND = CORBO_HIP_STAGE_FN_CONTROL_INEQ;
    template <int NV>
    __host__ __device__ static __forceinline__ double value(const double* u, const double* prm)
    {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) acc += u[i] * u[i];
        return acc - prm[0] * prm[0];
    }
};
