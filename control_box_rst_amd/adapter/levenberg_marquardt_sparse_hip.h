// levenberg_marquardt_sparse_hip.h -- the reference-side binding of the C-ABI (include/corbo_hip.h).
//
// corbo::LevenbergMarquardtSparseHip implements the reference's solver plug-in interface
//     corbo::NlpSolverInterface   (src/optimization/include/corbo-optimization/solver/nlp_solver_interface.h:67-115)
// with the same setters as corbo::LevenbergMarquardtSparse (levenberg_marquardt_sparse.h:86-90), so it is injected with
//     StructuredOptimalControlProblem(grid, dynamics, hypergraph, std::make_shared<LevenbergMarquardtSparseHip>())
// and called, unchanged, from src/optimal_control/src/structured_ocp/structured_optimal_control_problem.cpp:61,134,204.
//
// This file is compiled INSIDE the reference's build tree (it needs the reference's headers); it is not part of
// libcorbo_hip.so.  There is no CPU fallback: a hypergraph the device cannot describe makes solve() return
// SolverStatus::Error with a message on stderr.
#ifndef CONTROL_BOX_RST_AMD_ADAPTER_LEVENBERG_MARQUARDT_SPARSE_HIP_H_
#define CONTROL_BOX_RST_AMD_ADAPTER_LEVENBERG_MARQUARDT_SPARSE_HIP_H_

#include <corbo-optimization/hyper_graph/vertex_interface.h>
#include <corbo-optimization/solver/nlp_solver_interface.h>

#include <Eigen/Core>
#include <memory>
#include <vector>

#include "corbo_hip.h"

namespace corbo {

class LevenbergMarquardtSparseHip : public NlpSolverInterface
{
 public:
    using Ptr = std::shared_ptr<LevenbergMarquardtSparseHip>;

    LevenbergMarquardtSparseHip();
    ~LevenbergMarquardtSparseHip() override;

    // ---- NlpSolverInterface
    NlpSolverInterface::Ptr getInstance() const override { return std::make_shared<LevenbergMarquardtSparseHip>(); }
    bool isLsqSolver() const override { return true; }
    bool initialize(OptimizationProblemInterface* problem = nullptr) override;
    SolverStatus solve(OptimizationProblemInterface& problem, bool new_structure, bool new_run = true, double* obj_value = nullptr) override;
    void clear() override;

    // ---- LevenbergMarquardtSparse's parameters (same names, same meaning)
    void setIterations(int iterations) { _opts.iterations = iterations; }
    void setPenaltyWeights(double weight_eq, double weight_ineq, double weight_bounds);
    void setWeightAdapation(double factor_eq, double factor_ineq, double factor_bounds, double max_eq, double max_ineq, double max_bounds);

    // ---- By default NOTHING else has to be called: on every new structure the device model (grid kind, collocation scheme /
    //      integrator, dynamics and its parameters, cost weights, reference, constraints) is derived from the hypergraph itself
    //      (graph_recogniser.h), dimensions / vertex values / bounds / fixed flags / dt are read from the vertices, and on every new run
    //      the model (references, weights, parameters) is derived again and compared with the resident one (setTrackModel).  A graph the device cannot describe makes solve() return
    //      SolverStatus::Error with the reason on stderr.
    //      setDeviceModel() is the override for system dynamics the recogniser cannot know (user classes other than the device
    //      library's plug-in models): the caller states the model; setStateReference() then states the reference.
    void setDeviceModel(const corbo_hip_problem_desc& desc) { _desc = desc; _have_desc = true; _xref_traj.resize(0, 0); releaseHandle(); }
    void setStateReference(const Eigen::Ref<const Eigen::VectorXd>& xref) { _xref = xref; }
    void setDevice(int device) { _device = device; releaseHandle(); }
    // On every new structure the device model -- recognised or stated -- is checked against the graph's own edges (evaluated on the
    // host through the reference's computeValues* / computeCombinedSparseJacobian methods): stacked residual AND combined sparse
    // Jacobian, at the current vertex values and at a deterministically perturbed point (every unfixed component moved, controls
    // included -- at the reference's initial guess u = 0 and x_f = xref many descriptor errors are invisible).  A model that does not
    // describe what the graph's edges compute is refused instead of silently solving a different problem.  On by default; costs two
    // host evaluations and two device sweeps per structure change.
    void setVerifyModel(bool verify) { _verify = verify; }
    // On every new RUN with an unchanged structure the model is derived from the graph again and compared with the resident one: a
    // reference, weight or parameter that changed without a structure change (QuadraticFormCost::update returns false) is followed --
    // moved references keep the device handle, anything else rebuilds (and re-verifies) it.  On by default; costs one pass of the
    // recogniser over the edges per run.  Off: the caller vouches that only the vertex values change between runs.
    void setTrackModel(bool track) { _tracking = track; }
    // Hessian-path entry points (computeGradientObjective, computeSparseHessians*): by default EVERY call tracks the model -- derived again from the
    // graph and compared with the resident one -- like the reference, which always evaluates the live cost objects (a setpoint or weight that moves
    // between two calls of an external interior-point / SQP loop is followed).  setHessianTrackOncePerRun(true) relaxes this to the FIRST
    // such call of an outer run (one recogniser pass per run instead of one per call); the caller then ends a run with the next solve() / clear(),
    // or explicitly with newHessianRun() whenever references, weights or parameters may have changed.
    void setHessianTrackOncePerRun(bool once) { _hess_track_once = once; }
    void newHessianRun() { _hess_run_tracked = false; }

    const corbo_hip_stats& getStatistics() const { return _stats; }

    // ---- Operators of the exact-Hessian path, on the device, for the same hypergraph (recognised / verified like in solve()):
    //      OptimizationProblemInterface::computeSparseHessians{NNZ,Structure,Values} as IpoptWrapper::eval_h calls them
    //      (nlp_solver_ipopt_wrapper.cpp:249-271), entry order of the reference's lists.  false = refused (reason on stderr).
    bool computeSparseHessiansNNZ(OptimizationProblemInterface& problem, int& nnz_obj, int& nnz_eq, int& nnz_ineq, bool lower_part_only = false);
    bool computeSparseHessiansStructure(OptimizationProblemInterface& problem, Eigen::Ref<Eigen::VectorXi> i_row_obj, Eigen::Ref<Eigen::VectorXi> j_col_obj,
                                        Eigen::Ref<Eigen::VectorXi> i_row_eq, Eigen::Ref<Eigen::VectorXi> j_col_eq, Eigen::Ref<Eigen::VectorXi> i_row_ineq,
                                        Eigen::Ref<Eigen::VectorXi> j_col_ineq, bool lower_part_only = false);
    // eval_grad_f / eval_f: OptimizationProblemInterface::computeGradientObjective (+ computeValueObjective into *obj_value if given)
    bool computeGradientObjective(OptimizationProblemInterface& problem, Eigen::Ref<Eigen::VectorXd> gradient, double* obj_value = nullptr);
    bool computeSparseHessiansValues(OptimizationProblemInterface& problem, Eigen::Ref<Eigen::VectorXd> values_obj, Eigen::Ref<Eigen::VectorXd> values_eq,
                                     Eigen::Ref<Eigen::VectorXd> values_ineq, double multiplier_obj = 1.0, const double* multipliers_eq = nullptr,
                                     const double* multipliers_ineq = nullptr, bool lower_part_only = false);

 private:
    void releaseHandle();
    bool attach(OptimizationProblemInterface& problem, bool new_structure, bool new_run);
    bool attachHessianPath(OptimizationProblemInterface& problem);
    bool modelMatchesGraph(OptimizationProblemInterface& problem, bool perturbed);
    bool uploadVertices(const std::vector<VertexInterface*>& xs, const std::vector<VertexInterface*>& us, VertexInterface* xf, VertexInterface* dt);

    corbo_hip_lm_opts _opts;
    corbo_hip_problem_desc _desc;
    bool _have_desc = false;
    bool _verify    = true;
    bool _tracking  = true;
    bool _hess_run_tracked = false;
    bool _hess_track_once  = false;
    Eigen::VectorXd _xref;
    Eigen::MatrixXd _xref_traj;   // recognised time-varying state reference [N][nx] (empty: static)
    std::vector<double> _ref;
    int _device = 0;
    corbo_hip_handle _handle = nullptr;
    corbo_hip_dims _dims;
    corbo_hip_stats _stats;
    std::vector<double> _x, _lb, _ub;
    std::vector<VertexInterface*> _xs, _us;   // the grid's vertices of the attached structure (valid during a call)
    VertexInterface* _xf_v = nullptr;
    VertexInterface *_uprev_v = nullptr, *_uprev_dt_v = nullptr;   // the grid's fixed vertices _u_prev / _u_prev_dt (control-deviation term)
    VertexInterface* _dt_v = nullptr;
    bool _recognised = false;               // _desc comes from the recogniser (not from setDeviceModel)
    double _w_eq = 2, _w_ineq = 2, _w_b = 2;  // current (adapted) penalty weights: survive a structure change like the reference's _weight_*
};

FACTORY_REGISTER_NLP_SOLVER(LevenbergMarquardtSparseHip)

}  // namespace corbo

#endif
