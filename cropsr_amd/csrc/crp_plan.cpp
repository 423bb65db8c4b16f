// crp_plan.cpp -- how ONE genome is spread over the devices of a node: crp_plan_shares of the C ABI, and what crp_node_load
// (crp_node.cpp) cuts by.  The reference's contig loop (CROPSR.py:409) carries no state from one contig to the next, and
// everything it looks at around a match lies within 58 characters of it: any cut with a halo of CRP_HALO = 128 characters
// gives the same hits.  The rule (the same as cropsr_amd/parallel.py split_evenly, which the process-per-GPU path uses;
// tests/test_node.py holds the two against each other): the contigs, in their order, as CONTIGUOUS runs of equal size --
// device r gets the characters [r, r + 1) * total / world of the concatenation; a contig that straddles a boundary is cut
// there unless one side would be shorter than min_piece, in which case it stays whole on the side that holds most of it.
// At most world - 1 cuts; only a device's FIRST piece can begin inside a contig and only its LAST can end inside one -- so
// the rows a device owns are ONE run of each of its tables, which is what crp_node_gather sends.
#include "crp_plan.h"

#include <algorithm>
#include <cstring>

#include "cropsr_hip.h"

namespace crp {

void plan_shares(const uint64_t *lens, uint64_t n, int world, uint64_t min_piece, std::vector<std::array<uint64_t, 4>> &out)
{
    out.clear();
    int64_t total = 0;
    for (uint64_t k = 0; k < n; ++k) total += (int64_t)lens[k];
    std::vector<int64_t> bounds((size_t)world);
    for (int r = 0; r < world; ++r)
        bounds[(size_t)r] = (int64_t)(((unsigned __int128)(r + 1) * (unsigned __int128)total) / (unsigned)world);
    const int64_t minp = (int64_t)min_piece;
    int r = 0;
    int64_t acc = 0;
    for (uint64_t k = 0; k < n; ++k) {
        const int64_t len = (int64_t)lens[k];
        int64_t start = 0;
        for (;;) {
            const int64_t rest = len - start, room = bounds[(size_t)r] - acc;
            if (r == world - 1 || rest <= room) {
                out.push_back({k, (uint64_t)start, (uint64_t)len, (uint64_t)r});
                acc += rest;
                break;
            }
            if (room >= minp && rest - room >= minp) {  // cut at the boundary
                out.push_back({k, (uint64_t)start, (uint64_t)(start + room), (uint64_t)r});
                acc += room;
                start += room;
                r += 1;
            } else if (2 * room >= rest) {  // a sliver would be left over: the rest of the contig stays here
                out.push_back({k, (uint64_t)start, (uint64_t)len, (uint64_t)r});
                acc += rest;
                break;
            } else {  // a sliver would be cut off: the next device takes the contig from here
                r += 1;
            }
        }
        while (r < world - 1 && acc >= bounds[(size_t)r]) r += 1;
    }
}

static inline uint64_t words_for(uint64_t len) { return (len + 63) / 64 + 1; }  // = crp_arena_words_for (crp_api.cpp; no HIP here)

uint64_t slice_words_min(uint64_t halo) { return words_for(2 * halo + 64) + 2; }

void plan_slices(const uint64_t *lens, uint64_t n, uint64_t limit_words, uint64_t halo, std::vector<std::array<uint64_t, 4>> &out)
{
    out.clear();
    uint64_t slice = 0, used = 1;  // (word 0 of an arena is its leading separator)
    bool any = false;              // the current slice holds a piece
    for (uint64_t k = 0; k < n; ++k) {
        const uint64_t len = lens[k];
        uint64_t start = 0;
        for (;;) {
            uint64_t end = len;
            const uint64_t text_lo = start > halo ? start - halo : 0;
            uint64_t need = words_for(len - text_lo);  // (a piece that ends its contig has no right halo)
            bool cut = false;
            if (used + need > limit_words) {
                if (any) {  // the run goes on in a new slice
                    slice += 1;
                    used = 1;
                    any = false;
                }
                if (used + need > limit_words) {  // not even an empty slice holds it: cut to what one takes
                    const uint64_t chars = (limit_words - used - 1) * 64;
                    const uint64_t own = (chars - (start - text_lo) - halo) & ~(uint64_t)63;
                    end = start + own;
                    need = words_for(std::min(len, end + halo) - text_lo);
                    cut = true;
                }
            }
            out.push_back({k, start, end, slice});
            used += need;
            any = true;
            if (cut) {  // a piece that ends inside its contig closes its slice: one run per table
                slice += 1;
                used = 1;
                any = false;
            }
            if (end == len) break;
            start = end;
        }
    }
}

}  // namespace crp

extern "C" int crp_plan_shares(const uint64_t *lens, uint64_t n, int world, uint64_t min_piece, uint64_t *pieces, uint64_t cap,
                               uint64_t *n_pieces)
{
    if ((n && !lens) || world < 1 || !n_pieces || (cap && !pieces)) return CRP_ERR_INVALID;
    for (uint64_t k = 0; k < n; ++k)
        if (lens[k] >> 62) return CRP_ERR_INVALID;
    std::vector<std::array<uint64_t, 4>> out;
    try {
        crp::plan_shares(lens, n, world, min_piece ? min_piece : 4096, out);
    } catch (...) {
        return CRP_ERR_NOMEM;
    }
    *n_pieces = out.size();
    if (out.size() > cap) return CRP_ERR_CAPACITY;
    for (size_t q = 0; q < out.size(); ++q) std::memcpy(pieces + 4 * q, out[q].data(), 4 * sizeof(uint64_t));
    return CRP_OK;
}
