// graph_recogniser.h -- hypergraph -> device descriptor ("graph -> descriptor recogniser").
//
// The reference hands a solver nothing but the hypergraph (corbo::NlpSolverInterface::solve, nlp_solver_interface.h:105): vertices
// with values / bounds / fixed flags and opaque edge objects.  This unit works out, from that graph alone, which of the
// device-describable problems it is:
//   * the vertex set IS the grid object (DiscretizationGridInterface derives from VertexSetInterface): RTTI gives the grid kind
//     (FiniteDifferencesGrid, FiniteDifferencesVariableGrid, MultipleShootingGrid);
//   * the equality edges are FDCollocationEdge / MSVariableDynamicsOnlyEdge objects (RTTI); the dynamics object, the collocation
//     scheme and the integrator they hold are read out of them (the reference keeps them private and has no getters: they are
//     reached through pointers-to-member published by explicit template instantiations, [temp.spec]/6 -- no reference header is
//     modified); the benchmark systems of the reference are mapped to the device's dynamics ids with their parameters, user systems
//     are matched by evaluating them (unicycle, quadrotor);
//   * cost / constraint edges are generic adapters around StageFunction members (generic_edge.h:294-498): their weights and
//     references are IDENTIFIED through the edges' own public computeValues() on temporarily modified vertex values -- the weight
//     vectors and references come out bit-exact (evaluation points are chosen so that every product is a power-of-two scaling).
// Anything else (non-diagonal weights, control references, integral cost edges, other edge types) is refused with a reason:
// the adapter has no CPU fallback.
#ifndef CONTROL_BOX_RST_AMD_ADAPTER_GRAPH_RECOGNISER_H_
#define CONTROL_BOX_RST_AMD_ADAPTER_GRAPH_RECOGNISER_H_

#include <corbo-optimization/hyper_graph/hyper_graph_optimization_problem_base.h>

#include <Eigen/Core>
#include <string>

#include "corbo_hip.h"

namespace corbo {

struct HipRecognisedModel
{
    corbo_hip_problem_desc desc;   // everything but N, bounds, xf_fixed_mask, dt_ref / dt bounds (the adapter reads those from the vertices)
    Eigen::VectorXd xref;          // static state reference all cost / constraint terms agree on (time-varying: of the final-stage terms)
    Eigen::MatrixXd xref_traj;     // empty, or [N][nx]: the state reference of every grid point (time-varying ReferenceTrajectoryInterface)
    Eigen::VectorXd u_prev;        // control-deviation term (desc.ctrl_dev): the previously applied control and its age, the values of the grid's fixed
    double u_prev_dt = 0.0;        //   vertices _u_prev / _u_prev_dt as the edge of interval 0 sees them (empty / 0: no such term)
};

// false: *reason says what the device cannot describe
// hint: the model a previous call derived from the same graph (the adapter's model tracking, once per new run) -- cost terms are CHECKED against it
// (n + 1 edge evaluations each) before they are identified from scratch (~ 7 n); the result is the same model either way
bool recogniseHyperGraphForHip(BaseHyperGraphOptimizationProblem& hg, HipRecognisedModel* model, std::string* reason, const HipRecognisedModel* hint = nullptr);

// the state reference alone (cheap: a few evaluations of one cost edge); used on every solve to follow a reference that changed
// between runs without a structure change.  false if the graph has no state-dependent least-squares term.
bool readStateReferenceForHip(BaseHyperGraphOptimizationProblem& hg, int nx, Eigen::VectorXd* xref);
// the same for a time-varying reference: row k of traj ([N][nx], sized by the caller) = reference of the state cost term of grid point k
// (row N-1: of the final cost term, left untouched if the graph has none)
bool readStateReferenceTrajectoryForHip(BaseHyperGraphOptimizationProblem& hg, int nx, Eigen::MatrixXd* traj);
// the previously applied control and its age as the graph's first control-deviation edge holds them (cheap; read on every solve); false: no such edge
bool readPreviousControlForHip(BaseHyperGraphOptimizationProblem& hg, int nu, Eigen::VectorXd* u_prev, double* dt_prev);

}  // namespace corbo

#endif
