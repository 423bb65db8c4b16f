// Sanitizer + fuzz driver for the node's cut (crp_plan.cpp): random contig-length lists over 1..17 devices through
// crp_plan_shares under ASan + UBSan; every plan must cover each contig once and in order, give the devices non-decreasing
// contiguous runs, cut at most world - 1 times, and let only a device's first piece begin -- and only its last piece end --
// inside a contig (what crp_node_gather's "one run of owned rows per table" rests on).  Also the capacity protocol.
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "cropsr_hip.h"
#include "crp_plan.h"

#define REQUIRE(c)                                                       \
    do {                                                                 \
        if (!(c)) {                                                      \
            std::printf("FAILED %s (line %d, trial %d)\n", #c, __LINE__, trial); \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main()
{
    std::mt19937_64 rng(20251004);
    for (int trial = 0; trial < 20000; ++trial) {
        const int world = 1 + (int)(rng() % 17);
        const uint64_t n = rng() % 40;
        std::vector<uint64_t> lens(n);
        const int kind = (int)(rng() % 4);
        for (auto &l : lens)
            l = kind == 0 ? rng() % 50000 : kind == 1 ? (rng() % 8 == 0 ? 1000000 + rng() % 8000000 : rng() % 30000)
                : kind == 2 ? rng() % 5000000 : (uint64_t)1 << (rng() % 34);
        const uint64_t minp = (uint64_t[]){0, 1, 64, 4096, 100000}[rng() % 5];
        uint64_t need = 0;
        REQUIRE(crp_plan_shares(lens.data(), n, world, minp, nullptr, 0, &need) == (need ? CRP_ERR_CAPACITY : CRP_OK));
        REQUIRE(need >= n && need <= n + (uint64_t)world - (n ? 1 : 0) + (n ? 0 : 0));
        std::vector<uint64_t> p(4 * need + 4);
        uint64_t got = 0;
        if (need > 1) REQUIRE(crp_plan_shares(lens.data(), n, world, minp, p.data(), need - 1, &got) == CRP_ERR_CAPACITY && got == need);
        REQUIRE(crp_plan_shares(lens.data(), n, world, minp, p.data(), need, &got) == CRP_OK && got == need);
        uint64_t q = 0, prev_dev = 0;
        for (uint64_t k = 0; k < n; ++k) {
            uint64_t at = 0;
            bool first = true;
            while (q < got && p[4 * q] == k) {
                const uint64_t s = p[4 * q + 1], e = p[4 * q + 2], d = p[4 * q + 3];
                REQUIRE(s == at && e >= s && e <= lens[k] && (e > s || lens[k] == 0) && d < (uint64_t)world && d >= prev_dev);
                if (!first) REQUIRE(d > prev_dev);               // a cut moves on to a later device
                if (s > 0) REQUIRE(q == 0 || p[4 * (q - 1) + 3] != d);  // begins inside a contig: first piece of its device
                if (e < lens[k]) REQUIRE(q + 1 == got || p[4 * (q + 1) + 3] != d);  // ends inside one: last piece of its device
                at = e;
                prev_dev = d;
                first = false;
                ++q;
            }
            REQUIRE(!first && at == lens[k]);
        }
        REQUIRE(q == got);
    }
    // plan_slices (crp_scan_stream's cut; the node handle packs a device's share into arenas by the same rule): every contig
    // covered once and in order, slices in order, no slice over its word limit, and only a slice's FIRST piece begins -- only
    // its LAST ends -- inside a contig
    for (int trial = 0; trial < 20000; ++trial) {
        const uint64_t halo = 128;
        const uint64_t n = rng() % 30;
        std::vector<uint64_t> lens(n);
        const int kind = (int)(rng() % 3);
        for (auto &l : lens) l = kind == 0 ? rng() % 3000 : kind == 1 ? rng() % 200000 : (rng() % 6 == 0 ? 500000 + rng() % 3000000 : rng() % 20000);
        const uint64_t lo = crp::slice_words_min(halo);
        const uint64_t limit = rng() % 4 == 0 ? lo + rng() % 8 : lo + rng() % 20000;
        std::vector<std::array<uint64_t, 4>> out;
        crp::plan_slices(lens.data(), n, limit, halo, out);
        size_t q = 0;
        uint64_t prev_slice = 0, used = 1;
        for (uint64_t k = 0; k < n; ++k) {
            uint64_t at = 0;
            bool first = true;
            while (q < out.size() && out[q][0] == k) {
                const uint64_t s = out[q][1], e = out[q][2], sl = out[q][3];
                REQUIRE(s == at && e >= s && e <= lens[k] && (e > s || lens[k] == 0) && sl >= prev_slice && sl <= prev_slice + 1);
                if (sl != prev_slice) used = 1;
                const uint64_t text_lo = s > halo ? s - halo : 0, text_end = std::min(lens[k], e + halo);
                used += (text_end - text_lo + 63) / 64 + 1;
                REQUIRE(used <= limit);
                if (s > 0) REQUIRE(q == 0 || out[q - 1][3] != sl);                  // begins inside a contig: first piece of its slice
                if (e < lens[k]) REQUIRE(q + 1 == out.size() || out[q + 1][3] != sl);  // ends inside one: last piece of its slice
                at = e;
                prev_slice = sl;
                first = false;
                ++q;
            }
            REQUIRE(!first && at == lens[k]);
        }
        REQUIRE(q == out.size());
    }
    int trial = -1;
    uint64_t x = 0, one = (uint64_t)1 << 63;
    REQUIRE(crp_plan_shares(&one, 1, 2, 0, nullptr, 0, &x) == CRP_ERR_INVALID);
    REQUIRE(crp_plan_shares(nullptr, 0, 3, 0, nullptr, 0, &x) == CRP_OK && x == 0);
    std::printf("OK\n");
    return 0;
}
