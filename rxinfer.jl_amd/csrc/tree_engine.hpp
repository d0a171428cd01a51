// tree_engine.hpp — the seam between the C ABI (rxhip.hip) and the level-scheduled node-array executor (tree_engine.hip, tree_kernels.hpp).
#pragma once
#include <cstdint>
#include <string>

#include "../../include/rxhip.h"

namespace rxhip {
namespace tree {

struct Engine;

// compile `g` (host: classification, acyclicity, dependency levels, op tables) and allocate the device state.  RXHIP_ERR_UNSUPPORTED: a node type
// outside the Gaussian tree family, a cycle among the Gaussian variables, a dimension above the executor's; RXHIP_ERR_BADARG: malformed tables
rxhip_status create(const rxhip_graph_desc* g, int device, void* stream, Engine** out, std::string& err);
void destroy(Engine* e);
rxhip_status set_data(Engine* e, const int64_t* vars, int64_t n_vars, const double* host, std::string& err);
rxhip_status run(Engine* e, int iterations, int want_fe, std::string& err);
rxhip_status get_marginals(Engine* e, const int64_t* vars, int64_t n_vars, double* mean, double* cov, std::string& err);
rxhip_status get_precision(Engine* e, int64_t var, double* nu, double* V, std::string& err);
rxhip_status get_discrete(Engine* e, int64_t var, double* out, int32_t* n_components, std::string& err);
rxhip_status get_free_energy(Engine* e, double* per_iteration, std::string& err);
rxhip_status get_free_energy_per_replica(Engine* e, double* per_replica, std::string& err);
void counters(Engine* e, uint64_t* rule_calls, uint64_t* products, uint64_t* marginals);
void info(Engine* e, rxhip_tree_info* out);
// on: every later run() continues from the q(W) the previous run ended with (rxhip_tree_continue)
void set_continue(Engine* e, bool on);
int device_of(Engine* e);
void* stream_of(Engine* e);
rxhip_status sync(Engine* e, std::string& err);
// the per-iteration free energies of the last run where they live on the device (summed over ranks in place by rxhip_allreduce_free_energy); nullptr / 0
// when the last run did not compute them
double* free_energy_device(Engine* e, int* iterations);
rxhip_status rule_eval(const rxhip_rule_call* c, int device, std::string& err);
// the graph compiler alone (host, no device): the schedule's static figures and the reference-equivalent counts of ONE replica and iteration
rxhip_status plan(const rxhip_graph_desc* g, rxhip_tree_info* out, uint64_t* rule_calls, uint64_t* products, uint64_t* marginals, std::string& err);

}  // namespace tree
}  // namespace rxhip
