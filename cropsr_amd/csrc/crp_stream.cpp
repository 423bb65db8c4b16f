// crp_stream.cpp -- seam 1 as a PIPELINE: crp_scan_stream of the C ABI (include/cropsr_hip.h).
//
// The reference's loop is per contig (CROPSR.py:409-474: scan, score, write, next): it produces and consumes as it goes.
// The arena calls of the ABI take all contigs, scan once and then hand the tables over -- upload, scan and fetch strictly one
// after the other, the full-duplex host link used in one direction at a time.  Here the genome goes through in SLICES
// (crp_plan.cpp plan_slices: whole contigs while they fit, a contig longer than a slice cut with CRP_HALO characters of
// context either side, a hit owned by the piece its match index falls in -- the node handle's rule, crp_node.cpp), every
// slice in an arena of its own (LANES: a few reusable arenas, slice k in lane k mod L), and three host threads keep three
// HIP streams busy -- one stream per DIRECTION of the link, because that is what lets the two directions run at the same
// time (profiles/microbench/duplex_copy.hip: 48 + 48 GB/s on two one-way streams, 56 GB/s in all when every stream carries
// both directions in turn):
//
//     uploader (the calling thread)   UP stream:   slice k + 2: host pages -> pinned staging -> H2D -> pack kernel -> the
//                                                  scan, queued behind it; an event marks the scan
//     drainer  (a helper thread)      CUT stream:  slice k + 1: wait for the scan's event, cut the owned rows (lower_bound
//                                                  kernel), rebase to contig-local positions;
//                                     DOWN stream: the D2H of the owned rows, behind an event of the CUT stream -- straight
//                                                  into the caller's tables when those are pinned, else into pinned landing
//                                                  buffers
//     copier   (a helper thread)                   slice k: when its D2H has landed, copy it out of the landing buffers into
//                                                  its place in the caller's (pageable) tables with a few threads, and hand
//                                                  the lane back
//
// What comes back is ONE table per strand, contig after contig, ascending inside a contig, positions local to the contig
// string: the reference's own order (CROPSR.py:417-434), bit for bit what crp_scan_score + a host-side split by contig give.
#include <algorithm>
#include <array>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "crp_internal.h"
#include "crp_plan.h"
#include "crp_roctx.h"

namespace {

constexpr int N_LAND = 4;  // pinned landing buffers of STAGE_CHUNK bytes each (pageable tables only)

struct Lane {
    crp_arena *arena = nullptr;
    uint64_t arena_words = 0;
    uint32_t *d_needles = nullptr, *d_bounds = nullptr, *d_map = nullptr;
    uint64_t needles_cap = 0, bounds_cap = 0, map_cap = 0;
    uint32_t *h_small = nullptr;  // pinned: needles (2 np) | map (2 np) | bounds (4 np)
    uint64_t h_small_cap = 0;     // in pieces
    uint32_t *d_lpos[2] = {nullptr, nullptr};
    uint64_t lpos_cap[2] = {0, 0};
    hipEvent_t scanned = nullptr;  // UP stream: this lane's scan is behind here
    hipEvent_t cut = nullptr;      // CUT stream: the rebased positions are in place
    hipEvent_t down = nullptr;     // DOWN stream: the slice's last copy is behind here
};

// a batch of D2H copies queued on the DOWN stream: at most one landing buffer's worth
struct CopyJob {
    size_t slice = 0, lane = 0;
    int buf = -1;  // the landing buffer the segments arrive in; -1: the copies go straight into pinned tables
    struct Seg {
        size_t land_off;
        void *dst;
        size_t bytes;
    } seg[8];
    int n_seg = 0;
    bool last_of_slice = false;
};

struct Piece {
    uint64_t contig, start, end, text_lo, text_len, arena_off;
};

struct Slice {
    size_t first_piece = 0, n_pieces = 0;
};

double seconds_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

struct crp_stream_state {
    std::vector<Lane> lanes;
    hipStream_t s_cut = nullptr, s_down = nullptr;  // (the UP stream is the context's own)
    uint8_t *land[N_LAND] = {};
    hipEvent_t landed[N_LAND] = {};
    bool land_free[N_LAND] = {};  // guarded by Shared::m during a call
};

namespace crp {

void stream_release(crp_ctx *ctx)
{
    crp_stream_state *st = ctx->stream_state;
    if (!st) return;
    ctx->stream_state = nullptr;
    (void)hipSetDevice(ctx->device);
    if (st->s_cut) (void)hipStreamSynchronize(st->s_cut);
    if (st->s_down) (void)hipStreamSynchronize(st->s_down);
    for (Lane &l : st->lanes) {
        if (l.arena) (void)crp_arena_destroy(l.arena);
        (void)hipFree(l.d_needles);
        (void)hipFree(l.d_bounds);
        (void)hipFree(l.d_map);
        (void)hipFree(l.d_lpos[0]);
        (void)hipFree(l.d_lpos[1]);
        if (l.h_small) (void)hipHostFree(l.h_small);
        if (l.scanned) (void)hipEventDestroy(l.scanned);
        if (l.cut) (void)hipEventDestroy(l.cut);
        if (l.down) (void)hipEventDestroy(l.down);
    }
    for (int b = 0; b < N_LAND; ++b) {
        if (st->land[b]) (void)hipHostFree(st->land[b]);
        if (st->landed[b]) (void)hipEventDestroy(st->landed[b]);
    }
    if (st->s_cut) (void)hipStreamDestroy(st->s_cut);
    if (st->s_down) (void)hipStreamDestroy(st->s_down);
    delete st;
}

}  // namespace crp

namespace {

size_t lanes_wanted()
{
    if (const char *e = std::getenv("CRP_STREAM_LANES")) return (size_t)std::max(2, std::min(8, std::atoi(e)));
    return 4;
}

uint64_t slice_words_for(uint64_t slice_chars)
{
    if (!slice_chars) slice_chars = 64ull << 20;
    return std::min<uint64_t>(crp_arena_max_words(), std::max<uint64_t>(crp::slice_words_min(CRP_HALO), slice_chars / 64 + 2));
}

int small_host_buffer(crp_ctx *ctx, Lane &l, uint64_t np)
{
    if (l.h_small_cap >= np && l.h_small) return CRP_OK;
    if (l.h_small) (void)hipHostFree(l.h_small);
    l.h_small = nullptr;
    l.h_small_cap = 0;
    const uint64_t cap = np + np / 2 + 1024;
    CRP_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&l.h_small), 8 * cap * sizeof(uint32_t), hipHostMallocDefault));
    l.h_small_cap = cap;
    return CRP_OK;
}

// the two further streams, `want` lanes with an arena of exactly slice_words each (tables and per-lane buffers sized for a
// full slice now: a lane's first slice must not stall the pipeline on an allocation), and -- for pageable tables -- the
// landing buffers
int ensure_lanes(crp_ctx *ctx, size_t want, uint64_t slice_words, bool landing)
{
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->stream_state) {
        ctx->stream_state = new (std::nothrow) crp_stream_state();
        if (!ctx->stream_state) return CRP_ERR_NOMEM;
    }
    crp_stream_state *st = ctx->stream_state;
    if (!st->s_cut) CRP_HIP(ctx, hipStreamCreateWithFlags(&st->s_cut, hipStreamNonBlocking));
    if (!st->s_down) CRP_HIP(ctx, hipStreamCreateWithFlags(&st->s_down, hipStreamNonBlocking));
    if (st->lanes.size() < want) st->lanes.resize(want);
    for (size_t k = 0; k < want; ++k) {
        Lane &l = st->lanes[k];
        if (l.arena && l.arena_words < slice_words) {  // (a roomier arena than asked for serves as well)
            (void)crp_arena_destroy(l.arena);
            l.arena = nullptr;
        }
        if (!l.arena) {
            int rc = crp_arena_create(ctx, slice_words, &l.arena);
            if (rc != CRP_OK) return rc;
            l.arena_words = slice_words;
            const uint64_t chars = slice_words * 64;
            rc = crp::arena_reserve_tables(l.arena, chars, false);
            for (int s = 0; s < 2 && rc == CRP_OK; ++s)
                rc = crp::grow(ctx, reinterpret_cast<void **>(&l.d_lpos[s]), &l.lpos_cap[s], chars / 8 + 1024, sizeof(uint32_t));
            if (rc == CRP_OK) rc = small_host_buffer(ctx, l, 4096);
            if (rc == CRP_OK) rc = crp::grow(ctx, reinterpret_cast<void **>(&l.d_needles), &l.needles_cap, 2 * 4096, sizeof(uint32_t));
            if (rc == CRP_OK) rc = crp::grow(ctx, reinterpret_cast<void **>(&l.d_bounds), &l.bounds_cap, 4 * 4096, sizeof(uint32_t));
            if (rc == CRP_OK) rc = crp::grow(ctx, reinterpret_cast<void **>(&l.d_map), &l.map_cap, 2 * 4096, sizeof(uint32_t));
            if (rc != CRP_OK) return rc;
        }
        if (!l.scanned) CRP_HIP(ctx, hipEventCreateWithFlags(&l.scanned, hipEventDisableTiming));
        if (!l.cut) CRP_HIP(ctx, hipEventCreateWithFlags(&l.cut, hipEventDisableTiming));
        if (!l.down) CRP_HIP(ctx, hipEventCreateWithFlags(&l.down, hipEventDisableTiming));
    }
    for (int b = 0; b < N_LAND; ++b)
        if (!st->landed[b]) CRP_HIP(ctx, hipEventCreateWithFlags(&st->landed[b], hipEventDisableTiming));
    for (int b = 0; b < N_LAND && landing; ++b)  // (crp_scan_stream_prepare; a scan allocates them as it first needs them)
        if (!st->land[b]) CRP_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&st->land[b]), crp::STAGE_CHUNK, hipHostMallocDefault));
    return CRP_OK;
}

struct Shared {
    std::mutex m;
    std::condition_variable cv;
    size_t launched = 0, drained = 0;
    std::vector<CopyJob> jobs;  // drainer -> copier, in order
    size_t jobs_taken = 0;
    bool drainer_done = false;
    int failed = CRP_OK;
    std::string error;
};

void fail(Shared &sh, int rc, const std::string &what)
{
    std::lock_guard<std::mutex> lk(sh.m);
    if (sh.failed == CRP_OK) {
        sh.failed = rc;
        sh.error = what;
    }
    sh.cv.notify_all();
}

}  // namespace

extern "C" {

int crp_host_alloc(uint64_t bytes, void **out)
{
    if (!out) return CRP_ERR_INVALID;
    *out = nullptr;
    if (!bytes) return CRP_OK;
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return e == hipErrorOutOfMemory ? CRP_ERR_NOMEM : CRP_ERR_HIP;
    }
    return CRP_OK;
}

int crp_host_free(void *p)
{
    if (p && hipHostFree(p) != hipSuccess) {
        (void)hipGetLastError();
        return CRP_ERR_INVALID;
    }
    return CRP_OK;
}

static int scan_stream_impl(crp_ctx *ctx, const uint8_t *const *texts, const uint64_t *lens, uint64_t n, int guide_len, int flags,
                            uint64_t slice_chars, uint32_t *pos_plus, double *score_plus, uint64_t cap_plus, uint32_t *pos_minus,
                            double *score_minus, uint64_t cap_minus, uint64_t *per_contig, uint64_t *n_plus, uint64_t *n_minus, double *stats)
{
    crp::Range roctx_range("crp: scan stream (H2D | scan | D2H pipelined)");
    if (!ctx || (n && (!texts || !lens)) || (flags & ~CRP_SCAN_PRE)) return CRP_ERR_INVALID;
    if (guide_len < 0 || guide_len > 50) return CRP_ERR_UNSUPPORTED;
    for (uint64_t k = 0; k < n; ++k) {
        if (lens[k] && !texts[k]) return CRP_ERR_INVALID;
        if (lens[k] > 0xFFFFFFFFull) {
            ctx->last_error = "crp_scan_stream: contig " + std::to_string(k) + " has more than 2^32 - 1 characters; positions are 32-bit";
            return CRP_ERR_CAPACITY;
        }
    }
    const auto t_call = std::chrono::steady_clock::now();
    const bool want_pre = (flags & CRP_SCAN_PRE) != 0;
    const uint64_t slice_words = slice_words_for(slice_chars);

    // ---- the plan: pieces in contig order, slice by slice
    std::vector<Piece> pieces;
    std::vector<Slice> slices;
    {
        std::vector<std::array<uint64_t, 4>> cut;
        crp::plan_slices(lens, n, slice_words, CRP_HALO, cut);
        pieces.reserve(cut.size());
        for (const auto &c : cut) {
            Piece p;
            p.contig = c[0];
            p.start = c[1];
            p.end = c[2];
            p.text_lo = p.start > CRP_HALO ? p.start - CRP_HALO : 0;
            p.text_len = std::min<uint64_t>(lens[p.contig], p.end + CRP_HALO) - p.text_lo;
            p.arena_off = 0;
            if (c[3] >= slices.size()) {
                slices.resize((size_t)c[3] + 1);
                slices[(size_t)c[3]].first_piece = pieces.size();
            }
            slices[(size_t)c[3]].n_pieces += 1;
            pieces.push_back(p);
        }
    }
    const size_t n_slices = slices.size();
    if (per_contig && n) std::memset(per_contig, 0, 2 * n * sizeof(uint64_t));
    if (n_plus) *n_plus = 0;
    if (n_minus) *n_minus = 0;
    if (stats) std::memset(stats, 0, 12 * sizeof(double));
    if (!n_slices) return CRP_OK;

    uint32_t *host_pos[2] = {pos_plus, pos_minus};
    double *host_score[2] = {score_plus, score_minus};
    const uint64_t host_cap[2] = {cap_plus, cap_minus};
    // pinned tables (crp_host_alloc) take the rows by DMA directly; anything else goes through the landing buffers
    bool tables_pinned = true, any_table = false;
    for (int s = 0; s < 2; ++s) {
        if (host_pos[s] && host_cap[s]) any_table = true, tables_pinned = tables_pinned && crp::is_pinned_host(host_pos[s]);
        if (host_score[s] && host_cap[s]) any_table = true, tables_pinned = tables_pinned && crp::is_pinned_host(host_score[s]);
    }
    tables_pinned = tables_pinned && any_table;

    const size_t n_lanes = std::min<size_t>(lanes_wanted(), n_slices);
    // (a genome smaller than a slice: the one lane it needs is sized for it, not for 64 Mi characters)
    uint64_t lane_words = slice_words;
    if (n_slices == 1) {
        uint64_t need = 1;
        for (const Piece &p : pieces) need += crp_arena_words_for(p.text_len);
        lane_words = std::max<uint64_t>(crp::slice_words_min(CRP_HALO), std::min(slice_words, need + 1));
    }
    int rc = ensure_lanes(ctx, n_lanes, lane_words, false);
    if (rc != CRP_OK) return rc;
    crp_stream_state *st = ctx->stream_state;
    std::vector<Lane> &lanes = st->lanes;
    for (int b = 0; b < N_LAND; ++b) st->land_free[b] = true;

    Shared sh;
    uint64_t prefix[2] = {0, 0};
    bool overflow = false;
    double t_first_drained = 0, drain_busy = 0, drain_wait = 0, copy_busy = 0, copy_wait = 0;
    uint64_t copied = 0;

    // ---- the drainer: slice after slice, in order
    auto drain_slices = [&]() {
        int next_land = 0;
        for (size_t k = 0; k < n_slices; ++k) {
            {
                std::unique_lock<std::mutex> lk(sh.m);
                sh.cv.wait(lk, [&] { return sh.launched > k || sh.failed != CRP_OK; });
                if (sh.failed != CRP_OK) return;
            }
            Lane &l = lanes[k % n_lanes];
            Slice &sl = slices[k];
            crp_arena *a = l.arena;
            auto hip_fail = [&](hipError_t e, const char *what) {
                fail(sh, e == hipErrorOutOfMemory ? CRP_ERR_NOMEM : CRP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
            };
            hipError_t e = hipSetDevice(ctx->device);
            if (e == hipSuccess) e = hipEventSynchronize(l.scanned);  // the upload and the scan of THIS slice, nothing behind them
            if (e != hipSuccess) return hip_fail(e, "waiting for a slice's scan");
            const auto t0 = std::chrono::steady_clock::now();
            uint64_t x = 0, y = 0;
            int r = crp::scan_finish(a, &x, &y, true);
            if (r != CRP_OK) {
                fail(sh, r, std::string("scan of slice ") + std::to_string(k) + ": " + crp_last_error(ctx));
                return;
            }
            const size_t np = sl.n_pieces;
            const uint32_t nn = (uint32_t)(2 * np);
            uint32_t *h_bounds = l.h_small + 4 * l.h_small_cap;
            for (int s = 0; s < 2 && e == hipSuccess; ++s)
                e = crp::launch_lower_bound(st->s_cut, a->d_pos[s], a->n_hits[s], l.d_needles, nn, l.d_bounds + (size_t)s * nn);
            if (e == hipSuccess) e = hipMemcpyAsync(h_bounds, l.d_bounds, 2 * (size_t)nn * sizeof(uint32_t), hipMemcpyDeviceToHost, st->s_cut);
            if (e == hipSuccess) e = hipStreamSynchronize(st->s_cut);
            if (e != hipSuccess) return hip_fail(e, "ownership cuts");
            // what goes down: per strand the rebased positions and the f64 column of the owned run
            struct Col {
                const uint8_t *d_src;
                uint8_t *dst;
                size_t bytes;
            } cols[4];
            int n_cols = 0;
            for (int s = 0; s < 2; ++s) {
                const uint32_t *b = h_bounds + (size_t)s * 2 * np;
                for (size_t j = 0; j < np; ++j) {
                    if (b[2 * j + 1] < b[2 * j] || (j + 1 < np && b[2 * j + 2] != b[2 * j + 1])) {
                        fail(sh, CRP_ERR_STATE, "crp_scan_stream: the owned rows of slice " + std::to_string(k) + " are not one run");
                        return;
                    }
                    if (per_contig) per_contig[2 * pieces[sl.first_piece + j].contig + (size_t)s] += b[2 * j + 1] - b[2 * j];
                }
                const uint64_t first = b[0], cnt = b[2 * np - 1] - first, off = prefix[s];
                prefix[s] += cnt;
                if (!cnt) continue;
                if (!host_pos[s] && !host_score[s]) continue;  // (this strand's columns are not wanted)
                if (off + cnt > host_cap[s]) {  // the caller's tables are too small: the totals are still counted to the end
                    overflow = true;
                    continue;
                }
                if (host_pos[s]) {
                    if (l.lpos_cap[s] < cnt) {  // (denser than the lane was sized for; the hipFree inside grow waits for the device)
                        r = crp::grow(ctx, reinterpret_cast<void **>(&l.d_lpos[s]), &l.lpos_cap[s], cnt, sizeof(uint32_t));
                        if (r != CRP_OK) {
                            fail(sh, r, std::string("rebased positions: ") + crp_last_error(ctx));
                            return;
                        }
                    }
                    e = crp::launch_pos_rebase(st->s_cut, a->d_pos[s] + first, cnt, crp::PieceMap{l.d_map, l.d_map + np, (uint32_t)np}, l.d_lpos[s]);
                    if (e != hipSuccess) return hip_fail(e, "rebase kernel");
                    cols[n_cols++] = Col{reinterpret_cast<const uint8_t *>(l.d_lpos[s]), reinterpret_cast<uint8_t *>(host_pos[s] + off), cnt * sizeof(uint32_t)};
                }
                if (host_score[s])
                    cols[n_cols++] = Col{reinterpret_cast<const uint8_t *>((want_pre ? a->d_pre[s] : a->d_score[s]) + first),
                                         reinterpret_cast<uint8_t *>(host_score[s] + off), cnt * sizeof(double)};
            }
            // the copies, queued on the DOWN stream behind the CUT stream's kernels -- nothing here waits for them: the copier
            // does, and hands the lane back
            e = hipEventRecord(l.cut, st->s_cut);
            if (e == hipSuccess) e = hipStreamWaitEvent(st->s_down, l.cut, 0);
            if (e != hipSuccess) return hip_fail(e, "ordering the copies behind the cuts");
            CopyJob job;
            job.slice = k;
            job.lane = k % n_lanes;
            auto submit = [&](bool last) -> bool {
                job.last_of_slice = last;
                if (job.buf >= 0) e = hipEventRecord(st->landed[job.buf], st->s_down);
                if (e == hipSuccess && last) e = hipEventRecord(l.down, st->s_down);
                if (e != hipSuccess) {
                    hip_fail(e, "landing event");
                    return false;
                }
                {
                    std::lock_guard<std::mutex> lk(sh.m);
                    sh.jobs.push_back(job);
                }
                sh.cv.notify_all();
                job = CopyJob();
                job.slice = k;
                job.lane = k % n_lanes;
                return true;
            };
            size_t land_used = 0;
            for (int ci = 0; ci < n_cols; ++ci) {
                size_t done = 0;
                while (done < cols[ci].bytes) {
                    if (tables_pinned) {
                        e = hipMemcpyAsync(cols[ci].dst, cols[ci].d_src, cols[ci].bytes, hipMemcpyDeviceToHost, st->s_down);
                        if (e != hipSuccess) return hip_fail(e, "D2H into pinned tables");
                        done = cols[ci].bytes;
                        continue;
                    }
                    if (job.buf < 0) {  // the next landing buffer of the ring, once the copier has emptied it
                        const auto tw = std::chrono::steady_clock::now();
                        std::unique_lock<std::mutex> lk(sh.m);
                        sh.cv.wait(lk, [&] { return st->land_free[next_land] || sh.failed != CRP_OK; });
                        if (sh.failed != CRP_OK) return;
                        st->land_free[next_land] = false;
                        lk.unlock();
                        drain_wait += seconds_since(tw);
                        job.buf = next_land;
                        next_land = (next_land + 1) % N_LAND;
                        land_used = 0;
                        if (!st->land[job.buf]) {  // (a small genome never needs more than the first)
                            e = hipHostMalloc(reinterpret_cast<void **>(&st->land[job.buf]), crp::STAGE_CHUNK, hipHostMallocDefault);
                            if (e != hipSuccess) return hip_fail(e, "landing buffer");
                        }
                    }
                    const size_t take = std::min(cols[ci].bytes - done, crp::STAGE_CHUNK - land_used);
                    e = hipMemcpyAsync(st->land[job.buf] + land_used, cols[ci].d_src + done, take, hipMemcpyDeviceToHost, st->s_down);
                    if (e != hipSuccess) return hip_fail(e, "D2H into a landing buffer");
                    job.seg[job.n_seg++] = CopyJob::Seg{land_used, cols[ci].dst + done, take};
                    done += take;
                    land_used = (land_used + take + 63) & ~(size_t)63;
                    if (land_used >= crp::STAGE_CHUNK || job.n_seg == 8)
                        if (!submit(false)) return;
                }
            }
            if (!submit(true)) return;
            drain_busy += seconds_since(t0);
        }
    };
    // ---- the copier: job after job, in order
    auto copy_jobs = [&]() {
        for (;;) {
            CopyJob job;
            {
                std::unique_lock<std::mutex> lk(sh.m);
                sh.cv.wait(lk, [&] { return sh.jobs_taken < sh.jobs.size() || sh.drainer_done || sh.failed != CRP_OK; });
                if (sh.failed != CRP_OK) return;
                if (sh.jobs_taken == sh.jobs.size()) return;  // (the drainer is done and so is its queue)
                job = sh.jobs[sh.jobs_taken++];
            }
            Lane &l = lanes[job.lane];
            const auto t_wait = std::chrono::steady_clock::now();
            hipError_t e = hipSetDevice(ctx->device);
            if (e == hipSuccess && job.buf >= 0) e = hipEventSynchronize(st->landed[job.buf]);
            if (e == hipSuccess && job.last_of_slice) e = hipEventSynchronize(l.down);
            if (e != hipSuccess) {
                fail(sh, CRP_ERR_HIP, std::string("waiting for a slice's tables: ") + hipGetErrorString(e));
                return;
            }
            copy_wait += seconds_since(t_wait);
            const auto t0 = std::chrono::steady_clock::now();
            crp::CopySeg segs[8];
            for (int i = 0; i < job.n_seg; ++i) {
                copied += job.seg[i].bytes;
                segs[i] = crp::CopySeg{job.seg[i].dst, st->land[job.buf] + job.seg[i].land_off, job.seg[i].bytes};
            }
            crp::parallel_copy_multi(segs, job.n_seg, ctx->copy_threads);
            copy_busy += seconds_since(t0);
            {
                std::lock_guard<std::mutex> lk(sh.m);
                if (job.buf >= 0) st->land_free[job.buf] = true;
                if (job.last_of_slice) {
                    sh.drained = job.slice + 1;
                    if (job.slice == 0) t_first_drained = seconds_since(t_call);
                }
            }
            sh.cv.notify_all();
        }
    };
    auto drain = [&]() {
        try {
            drain_slices();
        } catch (...) {
            fail(sh, CRP_ERR_NOMEM, "crp_scan_stream: out of host memory while draining");
        }
        {
            std::lock_guard<std::mutex> lk(sh.m);
            sh.drainer_done = true;
        }
        sh.cv.notify_all();
    };
    auto copy = [&]() {
        try {
            copy_jobs();
        } catch (...) {
            fail(sh, CRP_ERR_NOMEM, "crp_scan_stream: out of host memory while copying");
        }
    };
    std::thread drainer, copier;
    try {
        sh.jobs.reserve(4 * n_slices + 8);
        drainer = std::thread(drain);
        copier = std::thread(copy);
    } catch (...) {
        fail(sh, CRP_ERR_NOMEM, "crp_scan_stream: no helper thread to be had");
        if (drainer.joinable()) drainer.join();
        ctx->last_error = "crp_scan_stream: no helper thread to be had";
        return CRP_ERR_NOMEM;
    }

    // ---- the uploader (this thread)
    double up_busy = 0, up_wait = 0;
    std::vector<const uint8_t *> ptrs;
    std::vector<uint64_t> plen, offs;
    auto upload_slices = [&]() {
        for (size_t k = 0; k < n_slices; ++k) {
            {
                const auto t0 = std::chrono::steady_clock::now();
                std::unique_lock<std::mutex> lk(sh.m);
                sh.cv.wait(lk, [&] { return sh.drained + n_lanes > k || sh.failed != CRP_OK; });  // lane k mod n_lanes is free again
                up_wait += seconds_since(t0);
                if (sh.failed != CRP_OK) return;
            }
            const auto t0 = std::chrono::steady_clock::now();
            Lane &l = lanes[k % n_lanes];
            Slice &sl = slices[k];
            const size_t np = sl.n_pieces;
            ptrs.resize(np);
            plen.resize(np);
            offs.resize(np);
            for (size_t j = 0; j < np; ++j) {
                const Piece &p = pieces[sl.first_piece + j];
                ptrs[j] = texts[p.contig] + p.text_lo;
                plen[j] = p.text_len;
            }
            int r = crp::arena_reset(l.arena);
            if (r == CRP_OK) r = crp_arena_add_contigs_ascii(l.arena, ptrs.data(), plen.data(), np, offs.data());
            if (r == CRP_OK) r = crp::arena_seal_async(l.arena);
            if (r == CRP_OK) r = crp::scan_begin(l.arena, guide_len, want_pre ? CRP_SCAN_PRE : 0);
            if (r == CRP_OK && l.h_small_cap < np) r = small_host_buffer(ctx, l, np);
            if (r == CRP_OK) r = crp::grow(ctx, reinterpret_cast<void **>(&l.d_needles), &l.needles_cap, 2 * np, sizeof(uint32_t));
            if (r == CRP_OK) r = crp::grow(ctx, reinterpret_cast<void **>(&l.d_bounds), &l.bounds_cap, 4 * np, sizeof(uint32_t));
            if (r == CRP_OK) r = crp::grow(ctx, reinterpret_cast<void **>(&l.d_map), &l.map_cap, 2 * np, sizeof(uint32_t));
            if (r != CRP_OK) {
                fail(sh, r, std::string("upload of slice ") + std::to_string(k) + ": " + crp_last_error(ctx));
                return;
            }
            // the ownership needles and the piece map of this slice: written into the lane's pinned scratch, uploaded behind the scan
            uint32_t *needles = l.h_small, *map = l.h_small + 2 * l.h_small_cap;
            for (size_t j = 0; j < np; ++j) {
                Piece &p = pieces[sl.first_piece + j];
                p.arena_off = offs[j];
                const uint64_t begin = p.arena_off + (p.start - p.text_lo);
                needles[2 * j] = (uint32_t)begin;
                needles[2 * j + 1] = (uint32_t)(begin + (p.end - p.start));
                map[j] = (uint32_t)begin;
                map[np + j] = (uint32_t)(begin - p.start);  // (mod 2^32)
            }
            hipError_t e = hipMemcpyAsync(l.d_needles, needles, 2 * np * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(l.d_map, map, 2 * np * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipEventRecord(l.scanned, ctx->stream);
            if (e != hipSuccess) {
                fail(sh, e == hipErrorOutOfMemory ? CRP_ERR_NOMEM : CRP_ERR_HIP, std::string("slice set-up: ") + hipGetErrorString(e));
                return;
            }
            up_busy += seconds_since(t0);
            {
                std::lock_guard<std::mutex> lk(sh.m);
                sh.launched = k + 1;
            }
            sh.cv.notify_all();
        }
    };
    try {
        upload_slices();
    } catch (...) {
        fail(sh, CRP_ERR_NOMEM, "crp_scan_stream: out of host memory while uploading");
    }
    drainer.join();
    copier.join();
    if (sh.failed != CRP_OK) {
        // whatever is still queued must not outlive this call (the caller's texts and tables are its to free)
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(st->s_cut);
        (void)hipStreamSynchronize(st->s_down);
        for (size_t k = 0; k < n_lanes; ++k) lanes[k].arena->scan_pending = 0;
        ctx->last_error = sh.error;
        return sh.failed;
    }
    if (n_plus) *n_plus = prefix[0];
    if (n_minus) *n_minus = prefix[1];
    if (stats) {
        stats[0] = seconds_since(t_call);
        stats[1] = up_busy;
        stats[2] = drain_busy;
        stats[3] = (double)n_slices;
        stats[4] = (double)n_lanes;
        stats[5] = t_first_drained;
        stats[6] = up_wait;
        stats[7] = drain_wait;
        stats[8] = copy_busy;
        stats[9] = copy_wait;
        stats[10] = (double)copied;
        stats[11] = tables_pinned ? 1.0 : 0.0;
    }
    if (overflow) {
        ctx->last_error = "crp_scan_stream: the caller's tables are too small: " + std::to_string(prefix[0]) + " '+' and " + std::to_string(prefix[1]) + " '-' rows are needed";
        return CRP_ERR_CAPACITY;
    }
    return CRP_OK;
}

int crp_scan_stream(crp_ctx *ctx, const uint8_t *const *texts, const uint64_t *lens, uint64_t n, int guide_len, int flags, uint64_t slice_chars,
                    uint32_t *pos_plus, double *score_plus, uint64_t cap_plus, uint32_t *pos_minus, double *score_minus, uint64_t cap_minus,
                    uint64_t *per_contig, uint64_t *n_plus, uint64_t *n_minus, double *stats)
{
    try {  // (vectors, strings and threads: nothing may throw across the C ABI)
        return scan_stream_impl(ctx, texts, lens, n, guide_len, flags, slice_chars, pos_plus, score_plus, cap_plus, pos_minus, score_minus,
                                cap_minus, per_contig, n_plus, n_minus, stats);
    } catch (...) {
        return CRP_ERR_NOMEM;
    }
}

int crp_scan_stream_prepare(crp_ctx *ctx, uint64_t slice_chars)
{
    if (!ctx) return CRP_ERR_INVALID;
    try {
        return ensure_lanes(ctx, lanes_wanted(), slice_words_for(slice_chars), true);
    } catch (...) {
        return CRP_ERR_NOMEM;
    }
}

}  // extern "C"
