// corbo-hip-model: name=kinematic_car slot=0 nx=3 nu=2 params=2.5,0,0,0
//
// A USER dynamics model, dropped into csrc/models/ (see README.md there): the kinematic car (bicycle) model
//     x' = v cos(theta),  y' = v sin(theta),  theta' = v / L tan(delta)          state (x, y, theta), controls (v, delta), prm[0] = wheelbase L
// -- the counterpart of a user's own corbo::SystemDynamicsInterface subclass (system_dynamics_interface.h:66-121; the class this one
// mirrors lives in oracle/ref_driver.cpp, scenario "kcar", and is what the golden fixtures kcar*.json were generated with).
// The first line is read by __graft_entry__.build(): it registers the model under the public dynamics id CORBO_HIP_DYN_USER + slot,
// compiles the sweep / pass / plant / Hessian kernels for it as one more translation unit and adds it to the dispatch tables.  The
// formula must be written operation for operation like the host class (contraction is off here): that is what bit-exact parity with
// the reference means for user code.  prepare() = the part that depends on the state only (evaluated once per grid state and once per
// perturbed state instead of once per finite-difference column); CACHE_XMASK bit i = prepare() reads x[i].
template <> struct Dynamics<CORBO_HIP_DYN_USER + 0> {
    static constexpr int NX = 3, NU = 2, NC = 2;
    static constexpr unsigned CACHE_XMASK = 0b100u;                                          // cos / sin of theta
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0b0011100u, RK4_GROUP1_COLS = 0b11110000u;   // theta at the later stages depends on theta, v, delta
    __device__ static __forceinline__ void prepare(const double* x, const double*, double* c)
    {
        c[0] = cos(x[2]);
        c[1] = sin(x[2]);
    }
    __device__ static __forceinline__ void eval(const double*, const double* c, const double* u, const double* prm, double* f)
    {
        f[0] = u[0] * c[0];
        f[1] = u[0] * c[1];
        f[2] = u[0] / prm[0] * tan(u[1]);
    }
};
