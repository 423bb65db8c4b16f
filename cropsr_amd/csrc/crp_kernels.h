// crp_kernels.h -- host-visible launch interface of crp_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crp {

constexpr int BLOCK = 256;     // threads per workgroup (4 wavefronts of 64)
#ifndef CRP_CHAIN_TICKET
// single-launch mode: 0 = tile id is blockIdx (the hardware dispatches workgroups in index order, so every
// tile a workgroup waits for has started; costs nothing), 1 = tile ids from an atomic ticket (start order
// by construction; one more memory round trip per tile, measured +5 % on the kernel).  Either way every
// wait is bounded and a timed-out scan is repeated with the three-launch sequence (crp_api.cpp).
#define CRP_CHAIN_TICKET 0
#endif
#ifndef CRP_DYN_CHUNKS
// single-launch mode: 1 = the four waves of a workgroup draw chunks of 64 hits from a counter in LDS (the
// look-back of wave 0 is absorbed by the other three), 0 = every wave owns every fourth chunk
#define CRP_DYN_CHUNKS 0
#endif
#ifndef CRP_LB_FIRST
// with CRP_DYN_CHUNKS: 1 = wave 0 looks back before it scores anything, 0 = after its first chunk
#define CRP_LB_FIRST 0
#endif
#ifndef CRP_NT_STORES
#define CRP_NT_STORES 1  // hit-table stores with the non-temporal hint: -1 % (0.470 vs 0.475 ms at steady clocks)
#endif
#ifndef CRP_ROTATE_WAVES
#define CRP_ROTATE_WAVES 1  // single-launch mode: wave 0 (which also resolves the prefix) owns the last chunk of every four
#endif
#ifndef CRP_LB_FIRST_STATIC
#define CRP_LB_FIRST_STATIC 1  // single-launch mode: wave 0 resolves the prefix before it scores its first hits
#endif
#ifndef CRP_LB_NOINLINE
#define CRP_LB_NOINLINE 0  // single-launch mode: the look-back as a real function call (keeps its registers out of the scorer's loop)
#endif
#ifndef CRP_PIPE_UNROLL
#define CRP_PIPE_UNROLL 0  // single-launch mode: the store-one-behind loop unrolled by two (no copies of the parked hit)
#endif
#ifndef CRP_LB_EARLY
#define CRP_LB_EARLY 0  // single-launch mode: request the look-back's descriptors before the first hits are scored
#endif
#ifndef CRP_EXPERIMENT_LDS_PAD
#define CRP_EXPERIMENT_LDS_PAD 0  // timing-only: unused LDS per workgroup, to measure the sensitivity to occupancy
#endif
#ifndef CRP_EXPERIMENT_NO_STORE
#define CRP_EXPERIMENT_NO_STORE 0  // timing-only: the table stores are skipped
#endif
#ifndef CRP_EXPERIMENT_STOP
#define CRP_EXPERIMENT_STOP 0  // timing-only: 1 = tiles stop after publishing their counts, 2 = after the hit list
#endif
#ifndef CRP_EXPERIMENT_NO_LB
#define CRP_EXPERIMENT_NO_LB 0  // timing-only ablations of the look-back (results are wrong when set)
#endif
#ifndef CRP_STREAM_MASKS
// emit pass: 1 = hit masks from the registers the planes were loaded into + wave shuffles (needs two words
// per thread), 0 = from the LDS copy after a barrier
#define CRP_STREAM_MASKS (CRP_TILE_WPT == 2)
#endif
#ifndef CRP_PRIO_UNTIL_PUBLISH
// single-launch mode: 1 = raised wave priority until the tile's counts are published, 2 = and, for the wave
// that resolves the tile's prefix, until it has; 0 = off
#define CRP_PRIO_UNTIL_PUBLISH 1
#endif
#ifndef CRP_PRIO_LEVEL
#define CRP_PRIO_LEVEL 3
#endif
#ifndef CRP_TABLES_AFTER_PUBLISH
#define CRP_TABLES_AFTER_PUBLISH 1  // stage the scorer's LDS tables after the block scan instead of before it
#endif
#ifndef CRP_LIST_COMPACT
// hit-list build: 1 = the non-empty 32-bit mask halves of a wave are compacted into a work list first and the
// bit peeling runs over items (no lane idles on an empty half); 0 = every lane peels its own halves
#define CRP_LIST_COMPACT 0
#endif
#ifndef CRP_LIST_BY_STRAND
#define CRP_LIST_BY_STRAND 1  // a tile whose hits overflow the list but fit it strand by strand takes one round per strand
#endif
#ifndef CRP_LIST_FASTPATH
#define CRP_LIST_FASTPATH 1  // hit-list build without the capacity test when the tile's hits all fit
#endif
#ifndef CRP_TILE_WPT
#define CRP_TILE_WPT 2
#endif
#ifndef CRP_LIST_CAP_PER_WPT
// LDS hit-list entries per round and per word-per-thread.  2 x 1344 = 2688 entries: the most that leaves the workgroup
// (31.6 KB of LDS) at five per CU with some margin -- 2 x 1472 no longer fits five.  A tile with more kept hits than the
// list holds pays a second list build; on the bench genome (1 520 hits per tile on average, soft-masked runs of ~2 kb)
// 2 048 entries left about one tile in six in that state: 0.468 -> 0.457 ms.  Unmasked genomes (TAIR10-like: 2 100 per
// tile) gain more.
#define CRP_LIST_CAP_PER_WPT 1344
#endif
#ifndef CRP_EMIT_BLOCK
#define CRP_EMIT_BLOCK 256
#endif
constexpr int EMIT_BLOCK = CRP_EMIT_BLOCK;  // threads per workgroup of the emit pass
constexpr int TILE_WPT = CRP_TILE_WPT;  // 64-position words per thread in the emit pass (1 or 2)
constexpr int TILE_WORDS = EMIT_BLOCK * TILE_WPT;
constexpr int ARENA_ALIGN_WORDS = 1024;  // arena planes are padded to this many words

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
    uint64_t cap_plus;  // table capacities in elements (single-pass mode checks them)
    uint64_t cap_minus;
};

// three-launch mode: per-tile counts, their exclusive scan (one workgroup per SCAN_CHUNK_TILES tiles), emit
constexpr uint32_t SCAN_CHUNK_TILES = 8192;
hipError_t launch_count(hipStream_t s, const Planes &pl, uint64_t n_words_padded, int l, uint2 *tile_cnt,
                        uint32_t n_tiles);
hipError_t launch_tile_scan(hipStream_t s, const uint2 *tile_cnt, uint32_t n_tiles, uint2 *tile_off, uint64_t *totals);
hipError_t launch_emit(hipStream_t s, const Planes &pl, uint64_t n_words_padded, int l, const uint2 *tile_off,
                       const HitTables &out);
// single-pass mode: `chain` and `chain_next` = chain_bytes(n_tiles) bytes of device scratch each; `chain`
// must be all zero, the kernel leaves `chain_next` all zero; on return chain[0] = ticket | fail << 32,
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
uint8_t host_classify_char(uint32_t ch);

}  // namespace crp
