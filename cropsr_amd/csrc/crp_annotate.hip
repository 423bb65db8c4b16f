// crp_annotate.hip -- the opt-in annotation join on the device (BASELINE.json configs[2], [3]).
//
// The reference parses the GFF (CROPSR.py:77-95), drops the table (:375) and writes '' into `features`
// (:466, :468); the join the -g / -p flags promise is this engine's own (cropsr_amd/annotate.py states it; parity
// unpinned).  What runs here is its per-hit half: the host turns the gene / CDS intervals of every contig into a
// TRACK -- ascending cut points in arena positions, each opening an elementary interval with one label-set id --
// and this file maps every kept hit of the resident tables to the id of the interval its cut site falls in:
//
//   cut site    end_pos - 3 (CROPSR.py:155-158): i - 3 for a '+' row (end_pos = i, :418), j for a '-' row
//               (end_pos = j + 3, :429, :433) -- independent of the guide length
//   has one     only a row whose long_sequence has 30 characters (CROPSR.py:466); those are exactly the rows the
//               scan scored (score != -1, include/cropsr_hip.h crp_scan_score)
//   id          ids[k] for the largest k with points[k] <= cut site, CRP_NO_FEATURE if there is none
//
// Both tables are ascending in arena position and so is the track: a merge.  It is done without walking: a bucket
// index (one entry per 2 048 arena positions: how many points lie before the bucket) is built once per track, so a
// hit reads two neighbouring bucket entries -- shared with the ~100-200 hits around it, i.e. served by the caches
// -- and finishes with a binary search over the handful of points inside its bucket.  The hit-table traffic is the
// algorithmic traffic: 4 B position + 8 B score in, 4 B id out per hit, streamed with 16-byte accesses; bound: HBM.
#include <algorithm>
#include <cstring>

#include "crp_internal.h"
#include "crp_roctx.h"

namespace crp {

constexpr int ANN_SHIFT = 11;  // one bucket = 2 048 arena positions (~150 kept hits, ~1 cut point of a plant GFF)
constexpr int ANN_ROWS = 4;    // table rows per thread: one 16-byte load of positions, one 16-byte store of ids
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// bucket[b] = number of points below b << ANN_SHIFT = index of the first point inside or after bucket b
__global__ __launch_bounds__(BLOCK) void annot_bucket_kernel(const uint32_t *__restrict__ points, uint32_t n_points,
                                                             uint32_t *__restrict__ bucket, uint32_t n_entries)
{
    const uint32_t b = blockIdx.x * BLOCK + threadIdx.x;
    if (b >= n_entries) return;
    const uint64_t key = (uint64_t)b << ANN_SHIFT;
    uint32_t lo = 0, hi = n_points;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (points[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    bucket[b] = lo;
}

__device__ __forceinline__ uint32_t annot_one(uint32_t pos, double score, uint32_t back, const uint32_t *__restrict__ points,
                                              const uint32_t *__restrict__ ids, const uint32_t *__restrict__ bucket,
                                              uint32_t last_bucket)
{
    if (score == -1.0) return CRP_NO_FEATURE;  // an 11-field row: no cut site (CROPSR.py:466-468)
    const uint32_t cut = pos - back;           // (pos >= l + 5 on the '+' strand: never wraps)
    const uint32_t b = min(cut >> ANN_SHIFT, last_bucket);
    uint32_t lo = bucket[b], hi = bucket[b + 1];
    while (lo < hi) {  // points[lo..hi) lie inside the bucket: how many of them are <= cut
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (points[mid] <= cut) lo = mid + 1;
        else hi = mid;
    }
    return lo ? ids[lo - 1] : CRP_NO_FEATURE;
}

// One launch for both tables: the first blocks_plus workgroups walk the '+' table (cut site = i - 3), the others the '-'
// table (cut site = j).
struct AnnotTable {
    const uint32_t *pos;
    const double *score;
    uint32_t *feat;
    uint64_t n;
};

__global__ __launch_bounds__(BLOCK) void annot_lookup_kernel(AnnotTable plus, AnnotTable minus, uint32_t blocks_plus,
                                                             const uint32_t *__restrict__ points, const uint32_t *__restrict__ ids,
                                                             const uint32_t *__restrict__ bucket, uint32_t last_bucket)
{
    const bool is_minus = blockIdx.x >= blocks_plus;  // (uniform per workgroup)
    const AnnotTable t = is_minus ? minus : plus;
    const uint32_t back = is_minus ? 0u : 3u;
    const uint64_t first = ((uint64_t)(blockIdx.x - (is_minus ? blocks_plus : 0u)) * BLOCK + threadIdx.x) * ANN_ROWS;
    if (first >= t.n) return;
    if (first + ANN_ROWS <= t.n) {
        // the tables are read once and the ids written once: keep them out of the way of the track and its index
        const u32x4 p = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(t.pos + first));
        const f64x2 s0 = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(t.score + first));
        const f64x2 s1 = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(t.score + first + 2));
        u32x4 f;
        f.x = annot_one(p.x, s0.x, back, points, ids, bucket, last_bucket);
        f.y = annot_one(p.y, s0.y, back, points, ids, bucket, last_bucket);
        f.z = annot_one(p.z, s1.x, back, points, ids, bucket, last_bucket);
        f.w = annot_one(p.w, s1.y, back, points, ids, bucket, last_bucket);
        __builtin_nontemporal_store(f, reinterpret_cast<u32x4 *>(t.feat + first));
    } else {
        for (uint64_t r = first; r < t.n; ++r) t.feat[r] = annot_one(t.pos[r], t.score[r], back, points, ids, bucket, last_bucket);
    }
}

}  // namespace crp

extern "C" {

int crp_annotate_set_track(crp_arena *a, const uint32_t *points, const uint32_t *ids, uint64_t n_points)
{
    crp::Range roctx_range("crp: annotation track");
    if (!a || (n_points && (!points || !ids)) || n_points > 0x7fffffffull) return CRP_ERR_INVALID;
    if (!a->sealed) return CRP_ERR_STATE;
    for (uint64_t k = 0; k < n_points; ++k)
        if (points[k] > 0x7fffffffu || (k && points[k] <= points[k - 1])) return CRP_ERR_INVALID;  // strictly ascending
    crp_ctx *ctx = a->ctx;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    a->have_track = false;
    a->have_feat = false;
    uint64_t cap_p = a->track_cap, cap_i = a->track_cap;
    int rc = crp::grow(ctx, reinterpret_cast<void **>(&a->d_ann_points), &cap_p, n_points, sizeof(uint32_t));
    if (rc == CRP_OK) rc = crp::grow(ctx, reinterpret_cast<void **>(&a->d_ann_ids), &cap_i, n_points, sizeof(uint32_t));
    if (rc != CRP_OK) {
        a->track_cap = 0;
        return rc;
    }
    a->track_cap = std::min(cap_p, cap_i);
    // one entry per bucket of the arena's positions, plus the end of the last bucket
    const uint64_t n_entries = ((a->padded_words * 64) >> crp::ANN_SHIFT) + 2;
    rc = crp::grow(ctx, reinterpret_cast<void **>(&a->d_ann_bucket), &a->bucket_cap, n_entries, sizeof(uint32_t));
    if (rc != CRP_OK) return rc;
    if (n_points) {
        rc = crp::staged_h2d(ctx, a->d_ann_points, points, n_points * sizeof(uint32_t));
        if (rc == CRP_OK) rc = crp::staged_h2d(ctx, a->d_ann_ids, ids, n_points * sizeof(uint32_t));
        if (rc != CRP_OK) return rc;
    }
    hipLaunchKernelGGL(crp::annot_bucket_kernel, dim3((uint32_t)((n_entries + crp::BLOCK - 1) / crp::BLOCK)), dim3(crp::BLOCK), 0,
                       ctx->stream, a->d_ann_points, (uint32_t)n_points, a->d_ann_bucket, (uint32_t)n_entries);
    CRP_HIP(ctx, hipGetLastError());
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    a->n_ann_points = n_points;
    a->n_ann_entries = n_entries;
    a->have_track = true;
    return CRP_OK;
}

int crp_annotate_lookup(crp_arena *a, uint32_t *feat_plus, uint32_t *feat_minus)
{
    crp::Range roctx_range("crp: annotation join");
    if (!a) return CRP_ERR_INVALID;
    if (!a->have_hits || !a->have_track) return CRP_ERR_STATE;
    crp_ctx *ctx = a->ctx;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    a->have_feat = false;
    for (int s = 0; s < 2; ++s) {
        const int rc = crp::grow(ctx, reinterpret_cast<void **>(&a->d_feat[s]), &a->feat_cap[s], a->n_hits[s], sizeof(uint32_t));
        if (rc != CRP_OK) return rc;
    }
    crp::prof_begin(ctx, CRP_K_ANNOTATE);
    {
        const uint64_t per_block = (uint64_t)crp::BLOCK * crp::ANN_ROWS;
        const uint32_t bp = (uint32_t)((a->n_hits[0] + per_block - 1) / per_block), bm = (uint32_t)((a->n_hits[1] + per_block - 1) / per_block);
        const crp::AnnotTable plus{a->d_pos[0], a->d_score[0], a->d_feat[0], a->n_hits[0]};
        const crp::AnnotTable minus{a->d_pos[1], a->d_score[1], a->d_feat[1], a->n_hits[1]};
        if (bp + bm)
            hipLaunchKernelGGL(crp::annot_lookup_kernel, dim3(bp + bm), dim3(crp::BLOCK), 0, ctx->stream, plus, minus, bp,
                               a->d_ann_points, a->d_ann_ids, a->d_ann_bucket, (uint32_t)(a->n_ann_entries - 2));
    }
    CRP_HIP(ctx, hipGetLastError());
    crp::prof_end(ctx, CRP_K_ANNOTATE);
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    crp::prof_collect(ctx, CRP_K_ANNOTATE);
    a->have_feat = true;
    uint32_t *host[2] = {feat_plus, feat_minus};
    for (int s = 0; s < 2; ++s)
        if (host[s] && a->n_hits[s]) {
            const int rc = crp::staged_d2h(ctx, host[s], a->d_feat[s], a->n_hits[s] * sizeof(uint32_t));
            if (rc != CRP_OK) return rc;
        }
    return CRP_OK;
}

}  // extern "C"
