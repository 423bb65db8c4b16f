// crp_plan.h -- the cut of a genome over the devices of a node (crp_plan.cpp; no HIP in either file: the sanitizer tests
// build them with g++).
#pragma once
#include <array>
#include <cstdint>
#include <vector>

namespace crp {

// pieces {contig, start, end, device}, in contig order; throws std::bad_alloc only
void plan_shares(const uint64_t *lens, uint64_t n, int world, uint64_t min_piece, std::vector<std::array<uint64_t, 4>> &out);

// One run of contigs (in order) as SLICES of at most limit_words arena words each (crp_arena_words_for per text, + 1 per
// slice): a slice is filled with whole pieces; a piece that would not fit goes on in the next slice, and one that no slice
// can hold is cut to what a slice takes -- every piece carries `halo` characters of context either side inside its
// contig, a hit belongs to the piece its match index falls in.  Only a slice's FIRST piece can begin inside a contig and only
// its LAST can end inside one, so the owned rows of a slice's tables are one run.  pieces {contig, start, end, slice}.
// limit_words must hold a piece of 64 owned characters between two halos (slice_words_min).  Throws std::bad_alloc only.
uint64_t slice_words_min(uint64_t halo);
void plan_slices(const uint64_t *lens, uint64_t n, uint64_t limit_words, uint64_t halo, std::vector<std::array<uint64_t, 4>> &out);

}  // namespace crp
