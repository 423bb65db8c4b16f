// crp_plan.h -- the cut of a genome over the devices of a node (crp_plan.cpp; no HIP in either file: the sanitizer tests
// build them with g++).
#pragma once
#include <array>
#include <cstdint>
#include <vector>

namespace crp {

// pieces {contig, start, end, device}, in contig order; throws std::bad_alloc only
void plan_shares(const uint64_t *lens, uint64_t n, int world, uint64_t min_piece, std::vector<std::array<uint64_t, 4>> &out);

}  // namespace crp
