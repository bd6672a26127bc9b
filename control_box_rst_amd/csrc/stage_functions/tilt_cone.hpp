// corbo-hip-stage: name=tilt_cone slot=0 kind=state_ineq nx_min=8
//
// A USER stage inequality, dropped into csrc/stage_functions/ (see README.md there): a tilt cone on roll and pitch,
//     c(x) = x[6]^2 + x[7]^2 - prm[0]^2 <= 0          (the 12-state quadrotor's Euler angles phi = x[6], theta = x[7]; prm[0] = the cone's half angle)
// -- the NON-INTEGRAL STATE TERM (dimension 1) of a user's own corbo::StageInequalityConstraint subclass (stage_functions.h:276-310:
// getNonIntegralStateTermDimension / computeNonIntegralStateTerm; the grid creates one UnaryVectorVertexEdge on x_k per interval,
// nlp_functions.cpp:70-80).  The class this one mirrors lives in oracle/ref_driver.cpp (UserStageInequalities, tilt=) and is what the golden
// fixtures sf_quad_tilt*.json were generated with.  The first line is read by __graft_entry__.build(): it registers the function under the
// public id CORBO_HIP_STAGE_FN_USER + slot (corbo_hip_problem_desc::stage_ineq), compiled into every kernel that evaluates a stage
// inequality (sweep, stage kernel of the big-block family, the Hessian-path operators).  Written operation for operation like the host
// class: contraction is off here.
template <> struct StageFunction<0> {
    static constexpr int KIND = CORBO_HIP_STAGE_FN_STATE_INEQ;
    template <int NV>
    __host__ __device__ static __forceinline__ double value(const double* x, const double* prm)
    {
        if constexpr (NV >= 8) return (x[6] * x[6] + x[7] * x[7]) - prm[0] * prm[0];
        else return 0.0;
    }
};
