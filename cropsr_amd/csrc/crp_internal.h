// crp_internal.h -- the two opaque handles of the C ABI (include/cropsr_hip.h), shared by the
// translation units that implement it: crp_api.cpp (arena, scan), crp_comm.cpp (RCCL),
// crp_offtarget.hip (off-target seed scan).
#pragma once
#include "cropsr_hip.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "crp_kernels.h"

struct crp_comm;  // crp_comm.cpp: the RCCL communicator and its scratch

struct crp_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string last_error;
    char name[128] = {0};
    int n_cu = 0;
    uint64_t hbm = 0;
    // upload staging (characters) and seam-2 scratch
    uint8_t *d_text = nullptr;
    uint64_t d_text_cap = 0;
    uint8_t *d_rows = nullptr;
    double *d_rpre = nullptr, *d_rscore = nullptr;
    uint64_t d_rows_cap = 0;
    bool two_pass = false;          // CRP_OPT_TWO_PASS as configured
    bool two_pass_latched = false;  // three look-back time-outs in a row: stay with three launches
    int timeout_streak = 0;
    uint64_t chain_timeouts = 0;    // single-launch scans that were repeated with three launches
    uint32_t chain_timeout_ticks = 2000000;  // CRP_OPT_CHAIN_TIMEOUT_US in ticks of the 100 MHz real-time counter
    int geometry = 0;  // CRP_OPT_TILE_GEOMETRY: 0 = by arena size, else GEO_* + 1
    uint32_t mute_tile = 0xffffffffu;  // test hook (environment CRP_TEST_MUTE_TILE): see crp_kernels.h
    // host <-> device staging for large transfers from / to pageable caller memory (crp_api.cpp: staged_h2d / staged_d2h):
    // two pinned buffers, filled / drained by a few host threads while the other one is on the link
    // (three: the uploads ring through all of them, so the host is a whole round ahead of the link and the copy engine
    // always has the next round queued; the staged copies of tables use the first two)
    static constexpr int N_PIN = 3;
    uint8_t *pin[N_PIN] = {nullptr, nullptr, nullptr};
    hipEvent_t pin_done[N_PIN] = {nullptr, nullptr, nullptr};
    bool pin_busy[N_PIN] = {false, false, false};
    int pin_next = 0;  // the buffer the next upload round takes
    int copy_threads = 8;  // environment CRP_COPY_THREADS
    // measurement
    int profiling = 0;  // 0 off, 1 emit kernel only, 2 all kernels
    hipEvent_t ev[2 * CRP_K_KINDS] = {};  // start/stop pair per kernel kind
    double ms[CRP_K_KINDS] = {};
    uint64_t launches[CRP_K_KINDS] = {};
    uint64_t *d_scalar = nullptr;  // 8 x u64 device scratch for small reductions
    uint64_t *h_scalar = nullptr;  // pinned mirror
    // off-target seed scan (crp_offtarget.hip)
    uint32_t *d_ot_hist = nullptr;  // 4^12 site counts
    uint4 *d_ot_ball = nullptr;     // 4^12 x {sites at distance 0, 1, 2, 3}
    uint16_t *d_ot_part = nullptr;  // partition scratch: low 12 seed bits of one strand's sites, bucket by bucket
    uint64_t ot_part_cap = 0;
    uint32_t *d_ot_part1 = nullptr;  // first partition level: whole codes, grouped by the top 6 seed bits
    uint64_t ot_part1_cap = 0;
    uint32_t *d_ot_bucket = nullptr;  // 4096 bucket totals, 4096 + 1 bucket starts, 4096 write cursors
    bool ot_solved = false;
    uint64_t ot_epoch = 0;          // bumped by crp_offtarget_reset: arenas added before it are stale
    // multi-GPU (crp_comm.cpp)
    crp_comm *comm = nullptr;
    // pipelined seam 1 (crp_stream.cpp): the lanes -- further contexts on this device, each with a reusable arena
    struct crp_stream_state *stream_state = nullptr;
};

struct crp_arena {
    crp_ctx *ctx = nullptr;
    uint64_t cap_words = 0;     // what the caller asked for
    uint64_t padded_words = 0;  // allocation per plane
    uint64_t *d_plane[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t used_words = 1;    // word 0 is the leading separator
    uint64_t n_contigs = 0, n_chars = 0;
    bool sealed = false;
    // per-tile scratch
    uint2 *d_tile_cnt = nullptr, *d_tile_off = nullptr;
    uint64_t *d_chain[2] = {nullptr, nullptr};  // single-pass mode: header + tile descriptors, used alternately
    int chain_cur = 0;                          // the buffer the next single-pass launch uses (all zero)
    uint64_t *d_totals = nullptr;
    uint64_t *h_totals = nullptr;  // pinned
    uint32_t n_tiles = 0;
    uint32_t tile_cap = 0;  // tiles the per-tile scratch has room for (an arena that is reset and refilled keeps it)
    int geo = 0;  // GEO_*: tile geometry, chosen at seal
    // hit tables: [0] = '+', [1] = '-'
    uint32_t *d_pos[2] = {nullptr, nullptr};
    double *d_score[2] = {nullptr, nullptr};
    double *d_pre[2] = {nullptr, nullptr};
    uint64_t tab_cap[2] = {0, 0};
    uint64_t pre_cap[2] = {0, 0};
    uint64_t n_hits[2] = {0, 0};
    bool have_hits = false, have_pre = false;
    // a scan whose launch has been queued and not yet collected (crp::scan_begin / scan_finish): 0 none, 1 the single
    // launch is on the stream, 2 three-launch mode (scan_finish runs the whole sequence)
    int scan_pending = 0, pend_guide_len = 0, pend_flags = 0;
    crp::HitTables pend_out = {};
    // off-target: raw seed words written by the scan itself (CRP_SCAN_SEEDS; same order and capacity as the hit tables)
    uint32_t *d_ot_raw[2] = {nullptr, nullptr};
    uint64_t raw_cap[2] = {0, 0};
    bool have_raw = false;  // the last scan wrote them
    // off-target: per-hit seed codes and counts, same order as the hit tables
    uint32_t *d_ot_seed[2] = {nullptr, nullptr};
    uint4 *d_ot_cnt[2] = {nullptr, nullptr};
    uint64_t ot_cap[2] = {0, 0};
    uint64_t *d_ot_own = nullptr;  // own_ranges of the last crp_offtarget_add
    uint64_t ot_own_cap = 0;
    uint64_t ot_epoch = 0;         // ctx->ot_epoch at the time of crp_offtarget_add (0: never added)
    // annotation join (crp_annotate.hip): the track of this arena and the per-hit label-set ids of the last look-up
    uint32_t *d_ann_points = nullptr, *d_ann_ids = nullptr;  // cut points (arena positions, ascending) and their ids
    uint64_t track_cap = 0, n_ann_points = 0;
    uint32_t *d_ann_bucket = nullptr;  // points before every bucket of 2 048 arena positions
    uint64_t bucket_cap = 0, n_ann_entries = 0;
    uint32_t *d_feat[2] = {nullptr, nullptr};  // same order as the hit tables
    uint64_t feat_cap[2] = {0, 0};
    bool have_track = false;
    bool have_feat = false;  // d_feat belongs to the current tables (cleared by the next scan)
};

#define CRP_HIP(ctx, call)                                                              \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) {                                                        \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e__);     \
            return e__ == hipErrorOutOfMemory ? CRP_ERR_NOMEM : CRP_ERR_HIP;            \
        }                                                                               \
    } while (0)

namespace crp {

// HIP-event bracket of one kernel kind (CRP_K_*) on the context's stream
inline bool prof_on(const crp_ctx *ctx, int kind) { return ctx->profiling >= 2 || (ctx->profiling == 1 && kind == CRP_K_EMIT); }
inline void prof_begin(crp_ctx *ctx, int kind)
{
    if (prof_on(ctx, kind)) (void)hipEventRecord(ctx->ev[kind * 2], ctx->stream);
}
inline void prof_end(crp_ctx *ctx, int kind)
{
    if (prof_on(ctx, kind)) (void)hipEventRecord(ctx->ev[kind * 2 + 1], ctx->stream);
}
// after the stream has been synchronised
inline void prof_collect(crp_ctx *ctx, int kind)
{
    float ms = 0.f;
    if (prof_on(ctx, kind) && hipEventElapsedTime(&ms, ctx->ev[kind * 2], ctx->ev[kind * 2 + 1]) == hipSuccess) {
        ctx->ms[kind] += ms;
        ctx->launches[kind] += 1;
    }
}

// Large copies between pageable host memory and the device, pipelined through the context's two pinned buffers: a few
// host threads move chunk k + 1 between the caller's pages and a pinned buffer while chunk k crosses the link (the
// runtime's own pageable path does this with one thread: ~17 GB/s up, ~10 GB/s down into untouched pages on the
// MI355X boxes).  Small copies go straight through hipMemcpyAsync.  Both are synchronous for the caller's memory:
// on return of staged_h2d `src` may be reused (the device copy is ordered on the stream); staged_d2h returns with
// `dst` filled.
constexpr size_t STAGE_CHUNK = 32ull << 20;
int staged_h2d(crp_ctx *ctx, void *d_dst, const void *src, size_t n);
int staged_d2h(crp_ctx *ctx, void *dst, const void *d_src, size_t n);
int staging_ready(crp_ctx *ctx);
void parallel_copy(void *dst, const void *src, size_t n, int threads);
struct CopySeg {
    void *dst;
    const void *src;
    size_t bytes;
};
void parallel_copy_multi(const CopySeg *segs, int n_segs, int threads);  // all of them at once, bytes dealt evenly to the threads
bool is_pinned_host(const void *p);  // memory the GPU reaches by DMA (crp_host_alloc): the staged copies go direct

// crp_stream.cpp's arena reuse: empty an arena for the next slice / seal without waiting (crp_api.cpp)
int arena_reset(crp_arena *a);
int arena_seal_async(crp_arena *a);
int arena_reserve_tables(crp_arena *a, uint64_t chars, bool want_pre);

// grow-only device buffer: *p holds at least `need` elements of `elem` bytes afterwards
int grow(crp_ctx *ctx, void **p, uint64_t *cap, uint64_t need, size_t elem);

// crp_scan_score in two halves (crp_api.cpp): scan_begin queues the single-launch scan on the context's stream and returns,
// scan_finish waits for it, repeats it where it must (tables too small, look-back time-out) and publishes the tables --
// a caller with several devices queues all of them before it waits for any (crp_node.cpp)
int scan_begin(crp_arena *a, int guide_len, int flags);
int scan_finish(crp_arena *a, uint64_t *n_plus, uint64_t *n_minus, bool kernel_done = false);  // kernel_done: an event behind the launch has been waited for

// crp_gather.hip: the kernels either side of the gatherv (ownership cuts, 16-bit position packing, rebasing at the root)
struct PieceMap {
    const uint32_t *begin;  // device: arena position each piece's OWNED range begins at, ascending
    const uint32_t *sub;    // device: what to subtract from a position inside that piece (mod 2^32)
    uint32_t n;             // 0: positions pass unchanged
};
constexpr int POS16_SHIFT = 16;
inline uint32_t pos16_buckets_for(uint64_t padded_words) { return (uint32_t)((padded_words * 64) >> POS16_SHIFT) + 1; }
hipError_t launch_lower_bound(hipStream_t s, const uint32_t *pos, uint64_t n, const uint32_t *needles, uint32_t n_needles,
                              uint32_t *out);
hipError_t launch_pos16_buckets(hipStream_t s, const uint32_t *pos, uint64_t n, uint64_t first, uint64_t last, uint32_t *bstart,
                                uint32_t n_buckets);
hipError_t launch_pos16_pack(hipStream_t s, const uint32_t *pos, uint64_t n, uint16_t *lo16);
hipError_t launch_pos16_expand(hipStream_t s, const uint16_t *lo16, uint64_t n, const uint32_t *bstart, uint32_t n_buckets,
                               const PieceMap &map, uint32_t *out);
hipError_t launch_pos_rebase(hipStream_t s, const uint32_t *pos, uint64_t n, const PieceMap &map, uint32_t *out);
hipError_t launch_add_u32(hipStream_t s, uint32_t *dst, const uint32_t *src, uint64_t n);  // dst += src (n: a multiple of 4)
constexpr uint64_t OT_HIST_ENTRIES = 1ull << 24;  // 4^12 seeds (crp_offtarget.hip: OT_SEEDS)

void comm_release(crp_ctx *ctx);  // crp_comm.cpp; called by crp_destroy
void stream_release(crp_ctx *ctx);  // crp_stream.cpp; called by crp_destroy
void comm_forget_arena(crp_ctx *ctx, const crp_arena *a);
int comm_allreduce_u32(crp_ctx *ctx, uint32_t *d_buf, uint64_t n);  // in-place sum over the ranks; no-op without a communicator
int comm_world(const crp_ctx *ctx);  // 0 without a communicator
int comm_rank(const crp_ctx *ctx);
uint64_t comm_gather_bytes(const crp_ctx *ctx);  // bytes this rank sent to (peer) or received as (root) the root in the last crp_gather_hits

}  // namespace crp
