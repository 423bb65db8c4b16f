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

// ---- tile geometry of the two kernels that work tile by tile (count, emit)
// A workgroup of BLOCK threads handles one tile of WORDS 64-position words.  The first OWNERS = WORDS / WPT threads own WPT
// words each: they load them, derive the hit masks and build their part of the LDS hit list; ALL threads score, one kept hit
// per lane and iteration.  LIST = hit-list entries per round (a tile with more kept hits takes several rounds); LDS_LIMIT =
// what the workgroup's LDS must stay within for the intended number of workgroups per CU (checked in emit_kernel).
// Tile-local positions are 16-bit list entries: WORDS * 64 <= 65 536.
template <int BLOCK_, int WORDS_, int WPT_, int LIST_, int LDS_LIMIT_>
struct TileGeo {
    static constexpr int BLOCK = BLOCK_, WORDS = WORDS_, WPT = WPT_, OWNERS = WORDS_ / WPT_, LIST = LIST_, LDS_LIMIT = LDS_LIMIT_;
    static_assert(OWNERS * WPT_ == WORDS_ && OWNERS % 64 == 0 && OWNERS <= BLOCK_ && BLOCK_ % 64 == 0, "whole waves own words");
    static_assert(WORDS_ * 64 <= 65536 && WORDS_ % 64 == 0, "16-bit list entries");
};
// LARGE -- the throughput shape, for arenas that fill the GPU several times over (the >= 1 Gb genome the headline is quoted
// on): eight waves on 1 024 words, two words per lane.  52.5 KB of LDS per workgroup = three per CU = six waves per SIMD (the
// emit kernel is held to 80 VGPRs for that), where 256 threads on 512 words (31.6 KB: the scorer's tables do not shrink with
// the tile) stop at five: -4.4 % (profiles/EXPERIMENTS.md, round 3).  The LDS (planes 32.8 KB, scorer tables 10.8 KB, the
// list) must stay within 53 760 B = 42 allocation units of 1 280 B: one unit more and only two workgroups fit a CU (0.42 ->
// 0.52 ms).  What the planes leave is split between the scorer's chain tables and the list: 5 016 entries with the tables
// of gen_score_terms.py (4 504 entries and one more table bit measured the same; 5 376 with one table bit less +1.5 %).
// The bench genome has ~3 040 kept hits per tile (soft-masked runs of ~2 kb); a tile with more kept hits than the list
// holds takes a second round (one per strand when each strand fits).
using GeoLarge = TileGeo<512, 1024, 2, 5016, 53760>;
// SMALL -- arenas that give a CU one or two LARGE tiles at most (E. coli-like: 71 of them on 256 CUs), where one tile's life is
// the whole kernel: the same eight waves on HALF the words, one word per lane.  A lane has half the list to build and half
// the rows to score, so a tile lives about half as long (E. coli-like: 24.1 -> 17.6 us per launch, of which ~7.5 us are what
// any launch costs).  Throughput is lower (every tile stages the scorer's tables and looks back: +40 % on the 1.13 Gb
// genome, still +15 % at 1 800 LARGE tiles), so it is used below 1.5 LARGE tiles per CU only; smaller shapes still (256
// threads on 128 words, 512 on 256, 128 on 64 ...) measured within 1.5 us of this one where it wins and worse everywhere
// else (profiles/EXPERIMENTS.md, round 4).
#ifndef CRP_GEO_SMALL
#define CRP_GEO_SMALL 512, 512, 1, 3072, 53760  // (overridable for A/B builds: make EXTRA='-DCRP_GEO_SMALL=...' OUT=...)
#endif
using GeoSmall = TileGeo<CRP_GEO_SMALL>;
enum { GEO_LARGE = 0, GEO_SMALL = 1, GEO_COUNT = 2 };
int tile_words(int geo);
constexpr int ARENA_ALIGN_WORDS = 1024;       // arena planes are padded to this many words (a multiple of every geometry's tile)
static_assert(ARENA_ALIGN_WORDS % GeoLarge::WORDS == 0 && ARENA_ALIGN_WORDS % GeoSmall::WORDS == 0, "tile geometry");
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
// geo: GEO_* -- the geometry the arena was sealed with (n_tiles = n_words_padded / tile_words(geo))
hipError_t launch_count(hipStream_t s, int geo, const Planes &pl, uint64_t n_words_padded, int l, uint2 *tile_cnt,
                        uint32_t n_tiles);
hipError_t launch_tile_scan(hipStream_t s, const uint2 *tile_cnt, uint32_t n_tiles, uint2 *tile_off, uint64_t *totals);
hipError_t launch_emit(hipStream_t s, int geo, const Planes &pl, uint64_t n_words_padded, int l, const uint2 *tile_off,
                       const HitTables &out);
// single-pass mode: `chain` and `chain_next` = chain_bytes(n_tiles) bytes of device scratch each; `chain`
// must be all zero, the kernel leaves `chain_next` all zero; on return chain[0] = fail << 32,
// chain[1], chain[2] = the '+' and '-' table totals
size_t chain_bytes(uint32_t n_tiles);
// mute_tile: 0xffffffff, or (tests) the index of a tile that withholds its counts so that the look-back times out
// timeout_ticks: how long a look-back may wait, in ticks of the 100 MHz real-time counter
hipError_t launch_emit_chained(hipStream_t s, int geo, const Planes &pl, uint64_t n_words_padded, int l, uint64_t *chain,
                               uint64_t *chain_next, const HitTables &out, uint32_t mute_tile, uint32_t timeout_ticks);
// adds the number of entries of score[0..n) that are not -1 to *out (device memory)
hipError_t launch_count_scored(hipStream_t s, const double *score, uint64_t n, uint64_t *out);
// adds the number of upper-case A/C/G/T characters of the planes' first n_words words to *out (device memory)
hipError_t launch_count_plain(hipStream_t s, const uint64_t *up, const uint64_t *ac, uint64_t n_words, uint64_t *out);
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
