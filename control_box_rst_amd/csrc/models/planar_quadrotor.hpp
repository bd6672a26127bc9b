// corbo-hip-model: name=planar_quadrotor slot=1 nx=6 nu=2 params=1.0,0.05,0.25,9.81
//
// A USER dynamics model of the BIG-BLOCK family (5 <= nx <= 12; see README.md here and DESIGN.md §3.12): the planar quadrotor
//     x'' = -(u1 + u2) sin(theta) / m,   z'' = (u1 + u2) cos(theta) / m - g,   theta'' = (u1 - u2) l / I
// state (x, z, theta, x', z', theta'), controls = the two rotor thrusts, prm = m, I, l (arm), g.  The host class it mirrors is
// PlanarQuadrotorRef of oracle/ref_driver.cpp (scenario "pquad"), which generated the golden fixtures pquad_*.json.
// Models of this size run on multiple-shooting grids with Runge-Kutta integration (the family of the 12-state quadrotor: stage kernels
// that integrate one perturbed copy of (x_k, u_k) per lane, block-tridiagonal factorisation with the workspace in HBM); the build
// compiles this header's unit with -DCORBO_HIP_DYN_TU_BIG because nx > 4.
template <> struct Dynamics<CORBO_HIP_DYN_USER + 1> {
    static constexpr int NX = 6, NU = 2, NC = 2;
    static constexpr unsigned CACHE_XMASK = 0b000100u;                 // sin / cos of theta
    // theta at the later Runge-Kutta stages depends on theta, theta' and (through theta') the two thrusts
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0b11100100u;
    static constexpr unsigned RK4_GROUP1_COLS    = 0b11100011011000u;  // x', z', u1, u2 and half of the (trivial) x_{k+1} columns
    __device__ static __forceinline__ void prepare(const double* x, const double*, double* c)
    {
        c[0] = sin(x[2]);
        c[1] = cos(x[2]);
    }
    __device__ static __forceinline__ void eval(const double* x, const double* c, const double* u, const double* prm, double* f)
    {
        const double m = prm[0], I = prm[1], l = prm[2], g = prm[3];
        const double T = u[0] + u[1];
        f[0] = x[3];
        f[1] = x[4];
        f[2] = x[5];
        f[3] = -(T * c[0]) / m;
        f[4] = (T * c[1]) / m - g;
        f[5] = (u[0] - u[1]) * l / I;
    }
};
