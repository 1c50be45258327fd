// Internal interface of the AQL submission path (aql.hip); the C ABI on top of it is in forward.hip.
#pragma once
#include <vector>
#include "common.h"

namespace vog {

struct AqlProgram;

// rows[r] = kernels with no dependency on each other; everything in row r depends (at most) on rows < r
int aql_open(int n_queues);
int aql_num_queues();
int aql_program_build(const std::vector<std::vector<LaunchRecord>>& rows, AqlProgram** out);
int aql_program_packets(const AqlProgram* p);
int aql_program_rows(const AqlProgram* p);
int aql_submit(AqlProgram* const* progs, int n, int queue);
int aql_wait(AqlProgram* p, uint64_t timeout_us);
int aql_program_destroy(AqlProgram* p);

}  // namespace vog
