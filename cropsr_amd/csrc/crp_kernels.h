// crp_kernels.h -- host-visible launch interface of crp_kernels.hip.
//
// Geometry and the few build switches that ship.  Everything that was measured and did not ship (ticket
// numbering, dynamic chunk draw, balanced work-list build, other tile shapes, the timing-only ablations, ...) is
// described with its numbers in profiles/EXPERIMENTS.md and lives in the git history, not here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crp {

constexpr int BLOCK = 256;       // threads per workgroup (4 wavefronts of 64) of the streaming kernels (pack, score30, ...)
// The two kernels that work tile by tile (count, emit) run EIGHT wavefronts per workgroup on a tile of 1 024 words:
// 52.5 KB of LDS per workgroup = three per CU = six waves per SIMD (the emit kernel is held to 80 VGPRs for that),
// where 256 threads on 512 words (31.6 KB: the scorer's tables do not shrink with the tile) stop at five.
// Measured -4.4 % on the emit kernel (profiles/EXPERIMENTS.md, round 3).  1 024 words is also the largest tile whose
// positions fit the 16-bit entries of the hit list.
constexpr int TILE_BLOCK = 512;
constexpr int TILE_WPT = 2;      // 64-position words per thread in the emit pass
constexpr int TILE_WORDS = TILE_BLOCK * TILE_WPT;  // one workgroup = one tile of 1 024 words = 65 536 positions
constexpr int ARENA_ALIGN_WORDS = 1024;       // arena planes are padded to this many words
static_assert(ARENA_ALIGN_WORDS % TILE_WORDS == 0 && TILE_WORDS * 64 <= 65536, "tile geometry");
// LDS hit-list entries per round.  The workgroup's LDS (planes 32.8 KB, scorer tables 10.8 KB, this list) must stay within
// 53 760 B = 42 allocation units of 1 280 B: one unit more and only two workgroups fit a CU (measured: 0.42 -> 0.52 ms).
// What the planes leave is split between the scorer's chain tables and the list: 5 016 entries with the tables of
// gen_score_terms.py (4 504 entries and one more table bit measured the same; 5 376 with one table bit less +1.5 %).
// The bench genome has ~3 040 kept hits per tile on average (soft-masked runs of ~2 kb); a tile with more kept hits
// than the list holds takes a second round (one per strand when each strand fits).
constexpr int LIST_CAP = 5016;
constexpr int TILE_LDS_LIMIT = 53760;  // checked in emit_kernel
#ifndef CRP_NT_STORES
#define CRP_NT_STORES 1  // hit-table stores with the non-temporal hint: -1.5 % at steady clocks
#endif
struct Planes {
    const uint64_t *plane[4];  // hi, lo, up, ac
};

struct HitTables {
    uint32_t *pos_plus;
    double *score_plus;
    double *pre_plus;   // may be null
    uint32_t *pos_minus;
    double *score_minus;
    double *pre_minus;  // may be null
    uint64_t cap_plus;  // table capacities in rows: a row past the capacity is dropped by the store's own range check
    uint64_t cap_minus;
    uint32_t *seed_plus;   // off-target scan: raw seed word per hit (crp_offtarget.hip), or null
    uint32_t *seed_minus;
};

// three-launch mode: per-tile counts, their exclusive scan (one workgroup per SCAN_CHUNK_TILES tiles), emit
constexpr uint32_t SCAN_CHUNK_TILES = 8192;
hipError_t launch_count(hipStream_t s, const Planes &pl, uint64_t n_words_padded, int l, uint2 *tile_cnt,
                        uint32_t n_tiles);
hipError_t launch_tile_scan(hipStream_t s, const uint2 *tile_cnt, uint32_t n_tiles, uint2 *tile_off, uint64_t *totals);
hipError_t launch_emit(hipStream_t s, const Planes &pl, uint64_t n_words_padded, int l, const uint2 *tile_off,
                       const HitTables &out);
// single-pass mode: `chain` and `chain_next` = chain_bytes(n_tiles) bytes of device scratch each; `chain`
// must be all zero, the kernel leaves `chain_next` all zero; on return chain[0] = fail << 32,
// chain[1], chain[2] = the '+' and '-' table totals
size_t chain_bytes(uint32_t n_tiles);
// mute_tile: 0xffffffff, or (tests) the index of a tile that withholds its counts so that the look-back times out
// timeout_ticks: how long a look-back may wait, in ticks of the 100 MHz real-time counter
hipError_t launch_emit_chained(hipStream_t s, const Planes &pl, uint64_t n_words_padded, int l, uint64_t *chain,
                               uint64_t *chain_next, const HitTables &out, uint32_t mute_tile, uint32_t timeout_ticks);
// adds the number of entries of score[0..n) that are not -1 to *out (device memory)
hipError_t launch_count_scored(hipStream_t s, const double *score, uint64_t n, uint64_t *out);
hipError_t launch_score30(hipStream_t s, const uint8_t *rows, uint64_t n, int order, double *pre, double *score);
hipError_t launch_pack(hipStream_t s, const uint8_t *text, uint64_t len, uint64_t n_words, uint64_t *hi,
                       uint64_t *lo, uint64_t *up, uint64_t *ac);
// one wave's work in pack_groups_kernel: <= 64 words of one contig of a batch
struct PackGroup {
    uint32_t src;       // byte offset of the group's first character in the device text buffer (multiple of 16)
    uint32_t n_chars;   // real characters of the group (<= 4096); the rest of its words is void
    uint64_t dst_word;  // first plane word it fills
    uint32_t n_words;   // <= 64
    uint32_t pad;
};
hipError_t launch_pack_groups(hipStream_t s, const uint8_t *text, const PackGroup *groups, uint32_t n_groups, uint64_t *hi,
                              uint64_t *lo, uint64_t *up, uint64_t *ac);
uint8_t host_classify_char(uint32_t ch);

// Raw seed word the emit kernel writes per hit when HitTables::seed_* are set (guide lengths >= 20):
// bits 11..0 = low code bits, bits 23..12 = high code bits of the 12 seed characters (character k at bit k and
// 12 + k, k = 0 next to the PAM), already oriented like the scoring string; SEED_RAW_NONE = not 12 bases.
constexpr uint32_t SEED_RAW_NONE = 0xffffffffu;

}  // namespace crp
