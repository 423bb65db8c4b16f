// crp_gather.hip -- the small kernels either side of the path's one exchange, the gatherv of the per-device hit tables
// to a root (BASELINE.json configs[3], [4]: "final RCCL gatherv of the per-shard hit tables"; the reference's side of it
// is the append at CROPSR.py:423 / :434 inside the contig loop of :409).
//
// What crosses xGMI per hit is the position (u32) and the score (f64).  The score is incompressible if the result is to
// stay bit-exact; the position is not: a table is ascending, so the HIGH 16 bits of its entries change once per 65 536
// arena positions.  POS16 sends the low 16 bits of every position (2 B per hit, exact) plus, per table, one u32 per
// 65 536 positions of the arena: bstart[k] = index of the first hit at or after position k << 16.  Hit i then lies in
// bucket b(i) = (number of k with bstart[k] <= i) - 1 and its position is (b(i) << 16) | lo16[i] -- no escapes, no
// ordering between workgroups, nothing to agree on: 10 B per hit on the links instead of 12 (+ ~9 KB per table).
//
//   lower_bound_kernel     out[k] = first index whose position is >= needles[k]       (ownership cuts, per-contig counts)
//   pos16_buckets_kernel   bstart[] of the rows [first, last) of a table               (sender)
//   pos16_pack_kernel      lo16[i] = pos[first + i] & 0xffff                           (sender; 4 B in, 2 B out per hit)
//   pos16_expand_kernel    lo16 + bstart -> u32 positions, rebased piece by piece      (root; 2 B in, 4 B out per hit)
//   pos_rebase_kernel      u32 -> u32, rebased piece by piece                          (root's own rows; raw transport)
//
// Rebasing: the single-process node handle (crp_node.cpp) returns positions LOCAL to their contig string -- the regex
// match indices the reference iterates over -- while a device's table holds positions of its own arena.  A device's
// arena is a list of pieces (contig pieces with their halo); PieceMap gives, per piece, the arena position its OWNED
// range begins at and the constant to subtract.  All five kernels are streaming or trivially small; bound: HBM.
#include "crp_internal.h"

namespace crp {

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16x8_a2 __attribute__((ext_vector_type(8), aligned(2)));

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *__restrict__ v, uint32_t n, uint32_t key)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (v[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// number of entries of v[0..n) that are <= key
__device__ __forceinline__ uint32_t upper_bound_u32(const uint32_t *__restrict__ v, uint32_t n, uint32_t key)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (v[mid] <= key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(BLOCK) void lower_bound_kernel(const uint32_t *__restrict__ pos, uint32_t n,
                                                            const uint32_t *__restrict__ needles, uint32_t n_needles,
                                                            uint32_t *__restrict__ out)
{
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k < n_needles) out[k] = lower_bound_u32(pos, n, needles[k]);
}

__global__ __launch_bounds__(BLOCK) void pos16_buckets_kernel(const uint32_t *__restrict__ pos, uint32_t n, uint32_t first,
                                                              uint32_t last, uint32_t *__restrict__ bstart, uint32_t n_buckets)
{
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n_buckets) return;
    // (k << 16 stays below 2^32: n_buckets <= 2^15 + 1, and the last key of a full-size arena is 2^31)
    const uint32_t t = lower_bound_u32(pos, n, k << 16);
    bstart[k] = min(max(t, first), last) - first;
}

constexpr int G_ROWS = 8;  // rows per thread of the three streaming kernels: one 16-byte access on the 16-bit side

__global__ __launch_bounds__(BLOCK) void pos16_pack_kernel(const uint32_t *__restrict__ pos, uint64_t n, uint16_t *__restrict__ lo16)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * G_ROWS;
    if (i0 >= n) return;
    if (i0 + G_ROWS <= n) {
        // (`pos` starts at the first OWNED row of a table: 4-byte aligned, not more)
        const u32x4_a4 a = *reinterpret_cast<const u32x4_a4 *>(pos + i0);
        const u32x4_a4 b = *reinterpret_cast<const u32x4_a4 *>(pos + i0 + 4);
        u16x8 o;
        o[0] = (uint16_t)a.x; o[1] = (uint16_t)a.y; o[2] = (uint16_t)a.z; o[3] = (uint16_t)a.w;
        o[4] = (uint16_t)b.x; o[5] = (uint16_t)b.y; o[6] = (uint16_t)b.z; o[7] = (uint16_t)b.w;
        *reinterpret_cast<u16x8 *>(lo16 + i0) = o;
    } else {
        for (uint64_t i = i0; i < n; ++i) lo16[i] = (uint16_t)pos[i];
    }
}

// position -> position - sub[piece], piece = the last one whose owned range begins at or before the position
struct Rebaser {
    const uint32_t *__restrict__ begin;
    const uint32_t *__restrict__ sub;
    uint32_t n, p;
    __device__ __forceinline__ Rebaser(const PieceMap &m, uint32_t first_pos) : begin(m.begin), sub(m.sub), n(m.n), p(0)
    {
        if (n) {
            const uint32_t u = upper_bound_u32(begin, n, first_pos);
            p = u ? u - 1 : 0;  // (a row before the first owned range does not occur: the caller sends owned rows only)
        }
    }
    __device__ __forceinline__ uint32_t operator()(uint32_t pos)
    {
        if (!n) return pos;
        while (p + 1 < n && begin[p + 1] <= pos) ++p;  // rows ascend: walk forward
        return pos - sub[p];
    }
};

__global__ __launch_bounds__(BLOCK) void pos16_expand_kernel(const uint16_t *__restrict__ lo16, uint64_t n,
                                                             const uint32_t *__restrict__ bstart, uint32_t n_buckets, PieceMap map,
                                                             uint32_t *__restrict__ out)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * G_ROWS;
    if (i0 >= n) return;
    const uint32_t rows = (uint32_t)min((uint64_t)G_ROWS, n - i0);
    uint16_t lo[G_ROWS];
    if (rows == G_ROWS) {
        const u16x8 v = *reinterpret_cast<const u16x8 *>(lo16 + i0);  // (the staging side is 16-byte aligned by construction)
#pragma unroll
        for (int r = 0; r < G_ROWS; ++r) lo[r] = v[r];
    } else {
#pragma unroll
        for (int r = 0; r < G_ROWS; ++r) lo[r] = (uint32_t)r < rows ? lo16[i0 + r] : (uint16_t)0;
    }
    // bucket of the first row: the last k with bstart[k] <= i0 (bstart[0] = 0); the others walk forward from it
    uint32_t b = upper_bound_u32(bstart, n_buckets, (uint32_t)i0) - 1;
    uint32_t p[G_ROWS];
#pragma unroll
    for (int r = 0; r < G_ROWS; ++r) {
        const uint32_t i = (uint32_t)i0 + (uint32_t)r;
        while (b + 1 < n_buckets && bstart[b + 1] <= i) ++b;
        p[r] = (b << 16) | lo[r];
    }
    Rebaser rebase(map, p[0]);
#pragma unroll
    for (int r = 0; r < G_ROWS; ++r) p[r] = (uint32_t)r < rows ? rebase(p[r]) : 0u;
    if (rows == G_ROWS) {
        u32x4_a4 a, c;
        a.x = p[0]; a.y = p[1]; a.z = p[2]; a.w = p[3];
        c.x = p[4]; c.y = p[5]; c.z = p[6]; c.w = p[7];
        *reinterpret_cast<u32x4_a4 *>(out + i0) = a;  // (`out` starts in the middle of the root's table: 4-byte aligned)
        *reinterpret_cast<u32x4_a4 *>(out + i0 + 4) = c;
    } else {
        for (uint32_t r = 0; r < rows; ++r) out[i0 + r] = p[r];
    }
}

// (pos and out may be the same array -- crp_node_gather rebases raw positions where they landed; every thread reads its rows
// before it writes them -- so neither is __restrict__)
__global__ __launch_bounds__(BLOCK) void pos_rebase_kernel(const uint32_t *pos, uint64_t n, PieceMap map, uint32_t *out)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * G_ROWS;
    if (i0 >= n) return;
    const uint32_t rows = (uint32_t)min((uint64_t)G_ROWS, n - i0);
    Rebaser rebase(map, pos[i0]);
    if (rows == G_ROWS) {
        u32x4_a4 a = *reinterpret_cast<const u32x4_a4 *>(pos + i0);
        u32x4_a4 c = *reinterpret_cast<const u32x4_a4 *>(pos + i0 + 4);
        a.x = rebase(a.x); a.y = rebase(a.y); a.z = rebase(a.z); a.w = rebase(a.w);
        c.x = rebase(c.x); c.y = rebase(c.y); c.z = rebase(c.z); c.w = rebase(c.w);
        *reinterpret_cast<u32x4_a4 *>(out + i0) = a;
        *reinterpret_cast<u32x4_a4 *>(out + i0 + 4) = c;
    } else {
        for (uint32_t r = 0; r < rows; ++r) out[i0 + r] = rebase(pos[i0 + r]);
    }
}

// dst[i] += src[i]: the site histograms of the node's devices summed at the root when RCCL cannot run (crp_node.cpp)
__global__ __launch_bounds__(BLOCK) void add_u32_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t n4)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * BLOCK) {
        u32x4 a = reinterpret_cast<u32x4 *>(dst)[i];
        const u32x4 b = reinterpret_cast<const u32x4 *>(src)[i];
        a += b;
        reinterpret_cast<u32x4 *>(dst)[i] = a;
    }
}

static inline uint32_t blocks_for(uint64_t n, uint64_t per_block) { return (uint32_t)((n + per_block - 1) / per_block); }

hipError_t launch_lower_bound(hipStream_t s, const uint32_t *pos, uint64_t n, const uint32_t *needles, uint32_t n_needles,
                              uint32_t *out)
{
    if (!n_needles) return hipSuccess;
    hipLaunchKernelGGL(lower_bound_kernel, dim3(blocks_for(n_needles, BLOCK)), dim3(BLOCK), 0, s, pos, (uint32_t)n, needles,
                       n_needles, out);
    return hipGetLastError();
}

hipError_t launch_pos16_buckets(hipStream_t s, const uint32_t *pos, uint64_t n, uint64_t first, uint64_t last, uint32_t *bstart,
                                uint32_t n_buckets)
{
    if (!n_buckets) return hipSuccess;
    hipLaunchKernelGGL(pos16_buckets_kernel, dim3(blocks_for(n_buckets, BLOCK)), dim3(BLOCK), 0, s, pos, (uint32_t)n,
                       (uint32_t)first, (uint32_t)last, bstart, n_buckets);
    return hipGetLastError();
}

hipError_t launch_pos16_pack(hipStream_t s, const uint32_t *pos, uint64_t n, uint16_t *lo16)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(pos16_pack_kernel, dim3(blocks_for(n, (uint64_t)BLOCK * G_ROWS)), dim3(BLOCK), 0, s, pos, n, lo16);
    return hipGetLastError();
}

hipError_t launch_pos16_expand(hipStream_t s, const uint16_t *lo16, uint64_t n, const uint32_t *bstart, uint32_t n_buckets,
                               const PieceMap &map, uint32_t *out)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(pos16_expand_kernel, dim3(blocks_for(n, (uint64_t)BLOCK * G_ROWS)), dim3(BLOCK), 0, s, lo16, n, bstart,
                       n_buckets, map, out);
    return hipGetLastError();
}

hipError_t launch_pos_rebase(hipStream_t s, const uint32_t *pos, uint64_t n, const PieceMap &map, uint32_t *out)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(pos_rebase_kernel, dim3(blocks_for(n, (uint64_t)BLOCK * G_ROWS)), dim3(BLOCK), 0, s, pos, n, map, out);
    return hipGetLastError();
}

// n: a multiple of 4, both arrays 16-byte aligned (the 4^12-entry site histograms)
hipError_t launch_add_u32(hipStream_t s, uint32_t *dst, const uint32_t *src, uint64_t n)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(add_u32_kernel, dim3(4096), dim3(BLOCK), 0, s, dst, src, n / 4);
    return hipGetLastError();
}

}  // namespace crp
