// Sanitizer + fuzz driver for the node's cut (crp_plan.cpp): random contig-length lists over 1..17 devices through
// crp_plan_shares under ASan + UBSan; every plan must cover each contig once and in order, give the devices non-decreasing
// contiguous runs, cut at most world - 1 times, and let only a device's first piece begin -- and only its last piece end --
// inside a contig (what crp_node_gather's "one run of owned rows per table" rests on).  Also the capacity protocol.
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "cropsr_hip.h"

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
    int trial = -1;
    uint64_t x = 0, one = (uint64_t)1 << 63;
    REQUIRE(crp_plan_shares(&one, 1, 2, 0, nullptr, 0, &x) == CRP_ERR_INVALID);
    REQUIRE(crp_plan_shares(nullptr, 0, 3, 0, nullptr, 0, &x) == CRP_OK && x == 0);
    std::printf("OK\n");
    return 0;
}
