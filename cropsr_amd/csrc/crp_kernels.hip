// crp_kernels.hip -- gfx950 kernels of the PAM-scan + on-target-score path.
//
// Replaces, for a whole arena of contigs in one pass, the reference's per-contig
// hot loops:
//   CROPSR.py:415-416, :426-427   re.finditer('(?=.GG)') / ('(?=CC.)')
//   CROPSR.py:417-423, :428-434   window arithmetic + keep-filter
//   CROPSR.py:458-461, :285-313   scoring-string build + rs1_score
//
// Device representation (DESIGN.md "Data layout"): four bit-planes, 64 characters
// per 64-bit word, bit k of word w = arena position 64*w + k.
//   hi, lo : base code A=00 T=01 C=10 G=11      up : upper-case base
//   ac     : scoring base (acgtACGT, U == A)
//   void   : hi & lo & ~up & ~ac  -- positions outside every contig string
//   'Z'    : hi & ~lo & up & ~ac  -- scores as C but never matches the PAM regex
// Contigs are separated by >= 64 void positions, so every reference bounds test
// (i-l >= 5, j+3 >= 5, j+3+l <= len+10, j+2 < len, complete 30-window) becomes a
// test of a void bit at a fixed distance, and no kernel needs a contig table.
//
// Work decomposition: one 256-thread workgroup per tile of 256*WPT words.  PAM
// masks are 64-wide bit-parallel per lane; kept hits are ranked with popcounts and
// a block scan, compacted to an LDS list, and then scored one hit per lane so the
// f64 work is balanced and the table stores are coalesced.  Output order is
// ascending arena position per strand (two-pass count / scan / emit, no atomics):
// bitwise reproducible and identical to the reference's row order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "crp_kernels.h"
#include "crp_score.h"
#include "crp_score_generic.h"

#if CRP_NT_STORES  // the tables are written once and not read again by this kernel
#define CRP_TABLE_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define CRP_TABLE_STORE(ptr, val) (*(ptr) = (val))
#endif

namespace crp {

static constexpr uint64_t ALL = ~0ull;

// ------------------------------------------------------------------ helpers
// 32-bit funnel shift: bits [s+31 : s] of the 64-bit value {hi, lo}, 0 <= s < 32 (v_alignbit_b32)
__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, int s)
{
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)s);
}
__device__ __forceinline__ uint64_t pair64(uint32_t hi, uint32_t lo) { return ((uint64_t)hi << 32) | lo; }

// bit k of result = bit (k + d) of the stream (cur, next), 0 < d < 64.  Written on 32-bit halves:
// one v_alignbit_b32 per half instead of two 64-bit shifts and an OR.
__device__ __forceinline__ uint64_t ahead(uint64_t cur, uint64_t next, int d)
{
    const uint32_t c0 = (uint32_t)cur, c1 = (uint32_t)(cur >> 32), n0 = (uint32_t)next, n1 = (uint32_t)(next >> 32);
    if (d < 32) return pair64(funnel(n0, c1, d), funnel(c1, c0, d));
    return pair64(funnel(n1, n0, d - 32), funnel(n0, c1, d - 32));
}
// bit k of result = bit (k - e) of the stream (prev, cur), 0 < e < 64
__device__ __forceinline__ uint64_t behind(uint64_t prev, uint64_t cur, int e)
{
    const uint32_t p0 = (uint32_t)prev, p1 = (uint32_t)(prev >> 32), c0 = (uint32_t)cur, c1 = (uint32_t)(cur >> 32);
    if (e == 32) return pair64(c0, p1);
    if (e < 32) return pair64(funnel(c1, c0, 32 - e), funnel(c0, p1, 32 - e));
    return pair64(funnel(c0, p1, 64 - e), funnel(p1, p0, 64 - e));
}

struct WordTriple {
    uint64_t p, c, n;
};

// Kept-hit masks of one 64-position word.
//   plus : (?=.GG) at i  <=>  G(i+1) & G(i+2);  keep  i - l >= 5   (CROPSR.py:419)
//   minus: (?=CC.) at j  <=>  C(j) & C(j+1) & exists(j+2);
//          keep  j+3 >= 5  and  j+3+l <= len+10               (CROPSR.py:430)
// "exists"/"index >= 0"/"index < len" are void tests at fixed distances.
__device__ __forceinline__ void word_masks(const WordTriple &hi, const WordTriple &lo,
                                           const WordTriple &up, const WordTriple &ac, int l,
                                           uint64_t &mplus, uint64_t &mminus)
{
    const uint64_t g_c = hi.c & lo.c & up.c & ac.c, g_n = hi.n & lo.n & up.n & ac.n;
    const uint64_t c_c = hi.c & ~lo.c & up.c & ac.c, c_n = hi.n & ~lo.n & up.n & ac.n;
    const uint64_t v_p = hi.p & lo.p & ~up.p & ~ac.p;
    const uint64_t v_c = hi.c & lo.c & ~up.c & ~ac.c;
    const uint64_t v_n = hi.n & lo.n & ~up.n & ~ac.n;

    mplus = ahead(g_c, g_n, 1) & ahead(g_c, g_n, 2) & ~behind(v_p, v_c, l + 5);

    uint64_t m = c_c & ahead(c_c, c_n, 1) & ~ahead(v_c, v_n, 2) & ~behind(v_p, v_c, 2);
    if (l > 8) m &= ~ahead(v_c, v_n, l - 8);
    mminus = m;
}

// Stage TW words (+ one halo word each side) of the four planes into LDS.
// sh[p][0] = word t0-1, sh[p][1+k] = word t0+k, sh[p][TW+1] = word t0+TW.
// The arena allocation is padded to a multiple of the tile, so the body is
// always in range; halos beyond the arena read as void.
template <int TW>
__device__ __forceinline__ void stage_tile(const Planes &pl, uint64_t t0, uint64_t n_words_padded,
                                           uint64_t (*sh)[TW + 2])
{
    const int tid = threadIdx.x;
    constexpr int PAIRS = TW / 2;  // 16-byte units per plane
#pragma unroll
    for (int it = 0; it < (4 * PAIRS) / EMIT_BLOCK; ++it) {
        const int q = tid + it * EMIT_BLOCK;
        const int p = q / PAIRS, k = q % PAIRS;
        const uint64_t *src = pl.plane[p] + t0 + 2 * k;
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src);
        sh[p][1 + 2 * k] = v.x;
        sh[p][2 + 2 * k] = v.y;
    }
    if (tid < 8) {
        const int p = tid >> 1;
        const bool right = tid & 1;
        const uint64_t voidw = (p < 2) ? ALL : 0ull;
        uint64_t w;
        if (right) {
            const uint64_t idx = t0 + TW;
            w = idx < n_words_padded ? pl.plane[p][idx] : voidw;
            sh[p][TW + 1] = w;
        } else {
            w = t0 > 0 ? pl.plane[p][t0 - 1] : voidw;
            sh[p][0] = w;
        }
    }
}

// 64-bit inclusive scan across the wave.
__device__ __forceinline__ uint64_t wave_inclusive_scan(uint64_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Block-wide exclusive scan of a packed (plus | minus << 32) count.
// `wave_tot` is LDS scratch of BLOCK/64 entries.  Returns the exclusive prefix,
// sets `total` to the block total.
// The per-lane counts are small (<= 64*WPT per strand), so both fit one 32-bit
// word as 16-bit fields and the wave scan is six DPP adds (row_shr 1,2,4,8 inside
// each row of 16 lanes, then row_bcast:15 / row_bcast:31 across rows) instead of
// twelve 64-bit shuffles through the LDS crossbar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v)
{
    // lanes without a source (or outside ROW_MASK) contribute 0
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}

__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v)
{
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// WIDE: a wave's per-strand total can reach 2^16 (tiles of 1024 words or more): scan the halves apart
template <bool WIDE>
__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t *wave_tot, uint64_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc;
    if (WIDE) {
        inc = (uint64_t)wave_inclusive_scan_u32((uint32_t)v) | ((uint64_t)wave_inclusive_scan_u32((uint32_t)(v >> 32)) << 32);
    } else {
        const uint32_t v16 = (uint32_t)v | ((uint32_t)(v >> 32) << 16);  // plus | minus << 16
        const uint32_t inc16 = wave_inclusive_scan_u32(v16);
        inc = (uint64_t)(inc16 & 0xffffu) | ((uint64_t)(inc16 >> 16) << 32);
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < EMIT_BLOCK / 64; ++w) {
        const uint64_t t = wave_tot[w];
        if (w < wave) base += t;
        tot += t;
    }
    total = tot;
    return base + inc - v;
}

template <int WPT, int TW>
__device__ __forceinline__ void thread_masks(uint64_t (*sh)[TW + 2], int l, uint64_t (&mp)[WPT],
                                             uint64_t (&mm)[WPT])
{
    const int w0 = threadIdx.x * WPT;  // tile-local first word of this thread
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
        const int s = 1 + w0 + k;
        WordTriple t[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) t[p] = WordTriple{sh[p][s - 1], sh[p][s], sh[p][s + 1]};
        word_masks(t[0], t[1], t[2], t[3], l, mp[k], mm[k]);
    }
}

// -------------------------------------------------------------- pass 1: count
// Pure streaming pass: no LDS staging, no barrier before the loads.  A wavefront
// covers 128 consecutive words (two per lane, one 16-byte load per plane and lane
// = 64 bytes in flight per lane); the words a lane does not own come from its
// neighbours by wave shuffles of the DERIVED G / C / void masks, and only lanes 0
// and 63 touch memory for the words just outside the wave.  A workgroup of four
// waves produces the counts of two emit tiles (TILE_WORDS = 256 words each).
__device__ __forceinline__ void derive(uint64_t hi, uint64_t lo, uint64_t up, uint64_t ac, uint64_t &g,
                                       uint64_t &c, uint64_t &v)
{
    g = hi & lo & up & ac;
    c = hi & ~lo & up & ac;
    v = hi & lo & ~up & ~ac;
}

__device__ __forceinline__ uint64_t counts_of(uint64_t g_c, uint64_t g_n, uint64_t c_c, uint64_t c_n,
                                              uint64_t v_p, uint64_t v_c, uint64_t v_n, int l)
{
    const uint64_t mplus = ahead(g_c, g_n, 1) & ahead(g_c, g_n, 2) & ~behind(v_p, v_c, l + 5);
    uint64_t m = c_c & ahead(c_c, c_n, 1) & ~ahead(v_c, v_n, 2) & ~behind(v_p, v_c, 2);
    if (l > 8) m &= ~ahead(v_c, v_n, l - 8);
    return (uint64_t)__popcll(mplus) | ((uint64_t)__popcll(m) << 32);
}

// the same two masks for one word, from the derived G / C / void masks of the word and its neighbours
__device__ __forceinline__ void masks_of(uint64_t g_c, uint64_t g_n, uint64_t c_c, uint64_t c_n, uint64_t v_p,
                                         uint64_t v_c, uint64_t v_n, int l, uint64_t &mplus, uint64_t &mminus)
{
    mplus = ahead(g_c, g_n, 1) & ahead(g_c, g_n, 2) & ~behind(v_p, v_c, l + 5);
    uint64_t m = c_c & ahead(c_c, c_n, 1) & ~ahead(v_c, v_n, 2) & ~behind(v_p, v_c, 2);
    if (l > 8) m &= ~ahead(v_c, v_n, l - 8);
    mminus = m;
}

// LFIX > 0: guide length known at compile time (20, the reference's default), so
// every funnel shift has a constant amount and becomes one v_alignbit_b32 per half.
template <int LFIX>
__global__ __launch_bounds__(BLOCK) void count_kernel(Planes pl, uint64_t n_words_padded, int l_arg,
                                                       uint2 *__restrict__ tile_cnt)
{
    const int l = LFIX > 0 ? LFIX : l_arg;
    // a workgroup covers COUNT_WORDS words = COUNT_TPB emit tiles; a wave takes 128 words at a time
    constexpr int COUNT_WORDS = TILE_WORDS > 512 ? TILE_WORDS : 512;
    constexpr int COUNT_TPB = COUNT_WORDS / TILE_WORDS;
    constexpr int REPS = COUNT_WORDS / 512;
    static_assert(BLOCK == 256 && (TILE_WORDS == 256 || TILE_WORDS % 512 == 0), "count pass geometry");
    const uint32_t pair = blockIdx.x;
    __shared__ uint64_t wave_tot[BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t c = 0;
#pragma unroll
    for (int rep = 0; rep < REPS; ++rep) {
    const uint64_t w0 = (uint64_t)pair * COUNT_WORDS + (uint64_t)(wave * REPS + rep) * 128;  // first word of this pass
    if (w0 < n_words_padded) {
        const uint64_t wa = w0 + 2 * lane;  // this lane owns words wa, wa+1
        ulonglong2 q[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) q[p] = *reinterpret_cast<const ulonglong2 *>(pl.plane[p] + wa);
        // words just outside the wave: void beyond the arena
        uint64_t e[4] = {ALL, ALL, 0, 0};
        if (lane == 0 && w0 > 0) {
#pragma unroll
            for (int p = 0; p < 4; ++p) e[p] = pl.plane[p][w0 - 1];
        } else if (lane == 63 && w0 + 128 < n_words_padded) {
#pragma unroll
            for (int p = 0; p < 4; ++p) e[p] = pl.plane[p][w0 + 128];
        }
        uint64_t ga, ca, va, gb, cb, vb, ge, ce, ve;
        derive(q[0].x, q[1].x, q[2].x, q[3].x, ga, ca, va);
        derive(q[0].y, q[1].y, q[2].y, q[3].y, gb, cb, vb);
        derive(e[0], e[1], e[2], e[3], ge, ce, ve);
        // left neighbour's second word, right neighbour's first word
        uint64_t v_left = __shfl_up(vb, 1, 64);
        uint64_t g_right = __shfl_down(ga, 1, 64), c_right = __shfl_down(ca, 1, 64), v_right = __shfl_down(va, 1, 64);
        if (lane == 0) v_left = ve;
        if (lane == 63) { g_right = ge; c_right = ce; v_right = ve; }
        c += counts_of(ga, gb, ca, cb, v_left, va, vb, l) + counts_of(gb, g_right, cb, c_right, va, vb, v_right, l);
    }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if (lane == 0) wave_tot[wave] = c;
    __syncthreads();
    if (threadIdx.x < COUNT_TPB) {
        const uint64_t tile = (uint64_t)pair * COUNT_TPB + threadIdx.x;
        if (tile * TILE_WORDS < n_words_padded) {
            constexpr int WAVES = (BLOCK / 64) / COUNT_TPB;
            uint64_t t = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) t += wave_tot[WAVES * threadIdx.x + w];
            tile_cnt[tile] = make_uint2((uint32_t)t, (uint32_t)(t >> 32));
        }
    }
}

// ------------------------------------------------- pass 2: tile offset scan
// Exclusive scan of the per-tile counts, one workgroup per chunk of 8192 tiles.
// A workgroup first sums the counts of all EARLIER chunks itself (a few hundred KB
// of L2-resident reads) instead of waiting for other workgroups, so the chunks run
// in parallel with no hand-off; the last workgroup also publishes the totals.
__global__ __launch_bounds__(1024) void tile_scan_kernel(const uint2 *__restrict__ tile_cnt, uint32_t n_tiles,
                                                          uint2 *__restrict__ tile_off,
                                                          uint64_t *__restrict__ totals)
{
    const uint32_t chunk = blockIdx.x;
    // Chunks of 8192 tiles go through LDS: coalesced load, every thread scans its 8
    // consecutive entries, one scan of the 1024 thread sums, coalesced store.
    // Per-strand totals stay below 2^31, so the packed halves never carry into
    // each other.
    constexpr int PER = 8, CHUNK = 1024 * PER;
    __shared__ uint64_t buf[CHUNK];
    __shared__ uint64_t wave_tot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base = chunk * CHUNK;
    uint64_t carry = 0;
    {
        uint64_t acc = 0;
        // `base` is a multiple of 8192: eight independent 16-byte loads in flight per thread
        const ulonglong2 *cnt2 = reinterpret_cast<const ulonglong2 *>(tile_cnt);
        for (uint32_t i = threadIdx.x; i < base / 2; i += 4096) {
            ulonglong2 c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = cnt2[i + k * 1024];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // each 64-bit element is one uint2 {plus, minus}: already the packed form
                acc += c[k].x + c[k].y;
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if (lane == 0) wave_tot[wave] = acc;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 16; ++w) carry += wave_tot[w];
        __syncthreads();
    }
    {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t i = base + k * 1024 + threadIdx.x;
            uint64_t v = 0;
            if (i < n_tiles) {
                const uint2 c = tile_cnt[i];
                v = (uint64_t)c.x | ((uint64_t)c.y << 32);
            }
            buf[k * 1024 + threadIdx.x] = v;
        }
        __syncthreads();
        uint64_t loc[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            loc[k] = buf[threadIdx.x * PER + k];
            sum += loc[k];
        }
        const uint64_t inc = wave_inclusive_scan(sum);
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        uint64_t pre = carry, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint64_t t = wave_tot[w];
            if (w < wave) pre += t;
            tot += t;
        }
        uint64_t ex = pre + inc - sum;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            buf[threadIdx.x * PER + k] = ex;
            ex += loc[k];
        }
        carry += tot;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t i = base + k * 1024 + threadIdx.x;
            const uint64_t v = buf[k * 1024 + threadIdx.x];
            if (i < n_tiles) tile_off[i] = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
        }
    }
    if (threadIdx.x == 0 && base + CHUNK >= n_tiles) {  // the chunk that holds the last tile
        totals[0] = carry & 0xffffffffull;
        totals[1] = carry >> 32;
    }
}

// ------------------------------------------------------- pass 3: emit + score
// 31 bits of an LDS plane starting at bit position q (q counted from bit 0 of
// the left halo word): the 30-character window plus the character after it.
__device__ __forceinline__ uint32_t window31(const uint64_t *plane, uint32_t q)
{
    const uint32_t *p32 = reinterpret_cast<const uint32_t *>(plane);
    const uint32_t i = q >> 5;
    return __builtin_amdgcn_alignbit(p32[i + 1], p32[i], q & 31) & 0x7fffffffu;
}

__device__ __forceinline__ uint32_t reverse30(uint32_t x) { return __brev(x) >> 2; }

// Chained-scan descriptors (single-pass mode): one 64-bit word per tile,
//   bits 63..62 status (0 = not ready, 1 = tile aggregate, 2 = inclusive prefix),
//   bits 61..31 '-' count, bits 30..0 '+' count.
// The word is the whole message (value and flag travel in ONE relaxed agent-scope
// 8-byte store / load), so no fence is needed around it.
static constexpr uint64_t DESC_AGG = 1ull << 62, DESC_PREFIX = 2ull << 62, DESC_VALUE = (1ull << 62) - 1;

__device__ __forceinline__ uint64_t desc_pack(uint64_t packed32)  // plus | minus << 32  ->  31-bit fields
{
    return (packed32 & 0x7fffffffull) | ((packed32 >> 32) << 31);
}
__device__ __forceinline__ uint64_t desc_unpack(uint64_t v)
{
    return (v & 0x7fffffffull) | (((v >> 31) & 0x7fffffffull) << 32);
}

// Decoupled look-back over the tile descriptors (single-pass mode).  A tile publishes its
// (plus | minus << 32) counts as an AGGREGATE as soon as its block scan is done, and later
// -- once it knows the sum over all earlier tiles -- as an inclusive PREFIX.  A tile only
// ever waits for tiles with a lower number, and those have started before it: workgroups are
// dispatched in index order (or, with CRP_CHAIN_TICKET, numbered by an atomic ticket in start
// order).  Spins are bounded all the same: on timeout *fail is set, every later look-back
// gives up at once, and the host repeats the scan with the count / scan / emit sequence
// instead of using the result -- a wrong assumption about dispatch order would cost time
// once, never a wrong table.
constexpr int CHAIN_HEADER_WORDS = 4;
constexpr int LB_DEPTH = 2;  // descriptors per lane and round trip: 128 tiles (measured: 1 and 2 equal, 4 and 8 slower)

__device__ __forceinline__ void lookback_publish(uint64_t *desc, uint32_t tile, uint64_t total)
{
    const uint64_t tag = tile == 0 ? DESC_PREFIX : DESC_AGG;
    __hip_atomic_store(&desc[tile], tag | desc_pack(total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lane k, slot j looks at tile base - k - 64*j; all loads in flight together
__device__ __forceinline__ void lookback_load(const uint64_t *desc, int64_t base, uint64_t (&v)[LB_DEPTH])
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < LB_DEPTH; ++j) {
        const int64_t idx = base - lane - 64 * j;
        v[j] = DESC_PREFIX;  // before tile 0: an empty prefix
        if (idx >= 0) v[j] = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Called by ONE wave; `v` holds the descriptors lookback_load fetched earlier for
// base = tile - 1 (the loads were issued before the wave scored its first hits, so they
// cost no wait here).  Returns the exclusive prefix in every lane and publishes the
// inclusive one.
#if CRP_LB_NOINLINE
__device__ __attribute__((noinline)) uint64_t lookback_resolve(uint64_t *desc, uint32_t tile, uint64_t total,
#else
__device__ __forceinline__ uint64_t lookback_resolve(uint64_t *desc, uint32_t tile, uint64_t total,
#endif
                                                     uint64_t (&v)[LB_DEPTH], uint32_t *fail, bool muted,
                                                     uint32_t timeout_ticks)
{
    const int lane = threadIdx.x & 63;
    if (tile == 0) return 0;
#if CRP_EXPERIMENT_NO_LB == 1  // TIMING ONLY (wrong tables): what the look-back costs altogether
    return 0;
#endif
    uint64_t excl = 0;
    int64_t base = (int64_t)tile - 1;
    uint32_t spins = 0;
    uint64_t t_first_stall = 0;  // 100 MHz real-time counter at the first stalled look of this tile
    bool stalled_before = false;
    while (true) {
        uint64_t contrib = 0;
        bool found = false, stall = false;
        int64_t missing = 0;  // the nearest tile that has not published yet
#pragma unroll
        for (int j = 0; j < LB_DEPTH; ++j) {
            const uint64_t st = v[j] >> 62;
            const uint64_t not_ready = __ballot(st == 0);
            const uint64_t is_prefix = __ballot(st == 2);
            // slots nearer than the nearest prefix must all be ready
            const int p = is_prefix ? __builtin_ctzll(is_prefix) : 64;
            const uint64_t need = p >= 64 ? ~0ull : ((1ull << p) - 1);
            if (!found && !stall) {
                if (not_ready & need) {
                    stall = true;
                    missing = base - 64 * j - __builtin_ctzll(not_ready & need);
                } else {
                    if (lane <= p) contrib += desc_unpack(v[j] & DESC_VALUE);
                    found = p < 64;
                }
            }
        }
#if CRP_EXPERIMENT_NO_LB == 2  // TIMING ONLY (wrong tables): the look-back's loads and analysis, but no waiting
        if (stall) break;
#endif
        if (stall) {
            // Wait on that ONE descriptor (a single 8-byte load per look instead of the whole
            // window and its analysis), then read the window again.
            while ((__hip_atomic_load(&desc[missing], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 62) == 0) {
                // The bound is WALL TIME (legitimate waits are tens of microseconds; the default
                // allowance is 20 ms), read every 16th look from the constant-rate counter, so it
                // means the same on a throttled, shared or pre-empted GPU.
                bool give_up = false;
                if ((spins++ & 15u) == 0) {
                    const uint64_t now = __builtin_amdgcn_s_memrealtime();
                    if (!stalled_before) {
                        t_first_stall = now;
                        stalled_before = true;
                    }
                    give_up = now - t_first_stall > timeout_ticks;
                    // someone else already timed out: drain
                    give_up |= __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                }
                if (give_up) {
                    if (lane == 0) {
                        atomicExch(fail, 1u);
                        // header word 3: where the host reads the outcome without a copy (see chain_resolve)
                        uint64_t *const host = reinterpret_cast<uint64_t *>(reinterpret_cast<uint64_t *>(fail - 1)[3]);
                        if (host) host[0] = 1ull << 32;
                    }
                    return 0;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        } else {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) contrib += __shfl_xor(contrib, d, 64);
            excl += contrib;
            if (found) break;
            base -= 64 * LB_DEPTH;
        }
        lookback_load(desc, base, v);
    }
    if (lane == 0 && !muted)
        __hip_atomic_store(&desc[tile], DESC_PREFIX | desc_pack(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// what the single-pass mode needs inside emit_rounds
struct ChainArgs {
    uint64_t *desc;     // one descriptor per tile
    uint32_t *fail;
    uint64_t *totals;   // written by the last tile
    uint64_t *s_excl;   // LDS hand-over from the resolving wave to the workgroup
    uint32_t *s_flag;   // LDS: s_excl is valid
    uint32_t *s_next;   // LDS: next chunk of 64 list entries to hand out (CRP_DYN_CHUNKS)
    uint32_t tile, n_tiles;
    uint64_t total;     // this tile's (plus | minus << 32)
    bool muted;         // test hook: this tile publishes nothing
    uint32_t timeout_ticks;  // look-back allowance in ticks of the 100 MHz real-time counter
};

__device__ __forceinline__ void chain_resolve(const ChainArgs &ch, uint64_t (&lb)[LB_DEPTH])
{
    const uint64_t e = lookback_resolve(ch.desc, ch.tile, ch.total, lb, ch.fail, ch.muted, ch.timeout_ticks);
    if ((threadIdx.x & 63) == 0) {
        *ch.s_excl = e;
        if (ch.tile == ch.n_tiles - 1) {
            const uint64_t all = e + ch.total;
            ch.totals[0] = all & 0xffffffffull;
            ch.totals[1] = all >> 32;
            // header word 3 (set once by the host, never zeroed): a pinned host copy of the header, so that the
            // host needs no device-to-host copy between the launch and its synchronisation
            uint64_t *const host = reinterpret_cast<uint64_t *>(ch.totals[2]);
            if (host) {
                host[1] = all & 0xffffffffull;
                host[2] = all >> 32;
            }
        }
    }
}

template <int WPT, int TW, int CAP, bool PAM, bool CHAINED>
__device__ __forceinline__ void emit_rounds(uint64_t (*sh)[TW + 2], uint16_t *list, uint64_t *exp_tab, double *score_tab,
                                            const uint64_t (&mp)[WPT], const uint64_t (&mm)[WPT], uint64_t ex,
                                            uint32_t n_plus, uint32_t n_minus, int l, uint32_t tile_pos,
                                            uint64_t off_plus, uint64_t off_minus, const HitTables &out,
                                            const ChainArgs &ch);

// CHAINED = true : single pass.  Table offsets come from the decoupled look-back above;
//                  `chain` holds a 32-byte header -- tile ticket (u32), fail flag (u32), the two
//                  table totals (u64 each, written by the last tile), a device pointer to a pinned
//                  host copy of the first three words (or 0; never touched by the kernel) -- then one
//                  descriptor per tile; it must be all zero at launch.  Launches alternate
//                  between two such buffers and every tile zeroes its slot of the other one
//                  (`chain_next`), so no memset runs between scans.  The tile
//                  publishes its counts right after the block scan, builds its hit list
//                  and scores the first hit of every lane BEFORE it needs its offsets, so
//                  the look-back's round trip hides under that work.  Stores are
//                  bounds-checked against the table capacities, totals are published by
//                  the last tile.
// CHAINED = false: third pass of the count / scan / emit sequence (offsets from tile_off).
template <int WPT, bool CHAINED, int LFIX>
__global__ __launch_bounds__(EMIT_BLOCK) __attribute__((amdgpu_waves_per_eu(5, 8))) void emit_kernel(Planes pl, uint64_t n_words_padded, int l_arg,
                                                      const uint2 *__restrict__ tile_off, uint64_t *chain,
                                                      uint64_t *__restrict__ chain_next, HitTables out,
                                                      uint32_t mute_tile, uint32_t timeout_ticks)
{
    const int l = LFIX > 0 ? LFIX : l_arg;
    constexpr int TW = EMIT_BLOCK * WPT;
    constexpr int CAP = CRP_LIST_CAP_PER_WPT * WPT;  // list entries per round; typical tiles need one round
    __shared__ uint64_t sh[4][TW + 2];
    // exp table (256 words) + chain-prefix tables of the scorer, one block: until the hit list is built the
    // same bytes are the list build's work-list scratch (CRP_LIST_COMPACT)
    __shared__ uint64_t tabs[256 + CRP_SCORE_TAB_N];
    uint64_t *const exp_tab = tabs;
    double *const score_tab = reinterpret_cast<double *>(tabs + 256);
    __shared__ uint64_t wave_tot[EMIT_BLOCK / 64];
    __shared__ uint64_t s_excl;
    __shared__ uint32_t s_flag;
    __shared__ uint32_t s_next;
#if CRP_CHAIN_TICKET
    __shared__ uint32_t s_tile;
#endif
    __shared__ uint16_t list[CAP];
#if CRP_EXPERIMENT_LDS_PAD  // TIMING ONLY: occupancy sensitivity (bytes of unused LDS per workgroup)
    __shared__ uint8_t lds_pad[CRP_EXPERIMENT_LDS_PAD];
    if (n_words_padded == 12345) lds_pad[threadIdx.x] = 1;
#endif

    const int tid = threadIdx.x;
    uint32_t tile = blockIdx.x;
    if (CHAINED) {
#if CRP_CHAIN_TICKET
        // tile ids in START order, whatever order the hardware dispatches workgroups in
        if (tid == 0) {
            s_flag = 0;
            s_tile = atomicAdd(reinterpret_cast<uint32_t *>(chain), 1u);
        }
        __syncthreads();
        tile = s_tile;
#else
        if (tid == 0) s_flag = 0;
#endif
    }
    const uint64_t t0 = (uint64_t)tile * TW;
#if CRP_PRIO_UNTIL_PUBLISH
    // single-launch mode: later tiles wait for this tile's counts, so the short phase that produces
    // them (stage, masks, block scan) runs ahead of the scoring of the other workgroups on this CU
    if (CHAINED) __builtin_amdgcn_s_setprio(CRP_PRIO_LEVEL);
#endif
#if CRP_STREAM_MASKS
    // Every thread loads the two words it owns of each plane (one 16-byte load per plane) and derives
    // its hit masks from those REGISTERS, taking the neighbouring words' G / C / void masks from the
    // adjacent lanes by wave shuffles (lanes 0 and 63 read the word just outside the wave from
    // memory) -- the count pass's scheme.  The words also go to LDS, but only the window extraction
    // after the block scan reads them there: no barrier and no LDS round trip stand between the loads
    // and the tile's counts, which in single-launch mode is what later tiles wait for.
    static_assert(WPT == 2 && EMIT_BLOCK % 64 == 0, "streamed masks: two words per thread");
    uint64_t mp[WPT], mm[WPT];
    {
        const int lane = tid & 63;
        const uint64_t wa = t0 + 2 * (uint64_t)tid;          // this thread owns words wa, wa + 1
        const uint64_t w0 = t0 + 128 * (uint64_t)(tid >> 6);  // first word of this wave
        ulonglong2 q[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) q[p] = *reinterpret_cast<const ulonglong2 *>(pl.plane[p] + wa);
        uint64_t e[4] = {ALL, ALL, 0, 0};  // void beyond the arena
        if (lane == 0 && w0 > 0) {
#pragma unroll
            for (int p = 0; p < 4; ++p) e[p] = pl.plane[p][w0 - 1];
        } else if (lane == 63 && w0 + 128 < n_words_padded) {
#pragma unroll
            for (int p = 0; p < 4; ++p) e[p] = pl.plane[p][w0 + 128];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            sh[p][1 + 2 * tid] = q[p].x;
            sh[p][2 + 2 * tid] = q[p].y;
        }
        if (tid < 8) {  // the tile's own halo words, for windows that reach across its ends
            const int p = tid >> 1;
            const uint64_t voidw = (p < 2) ? ALL : 0ull;
            if (tid & 1) sh[p][TW + 1] = t0 + TW < n_words_padded ? pl.plane[p][t0 + TW] : voidw;
            else sh[p][0] = t0 > 0 ? pl.plane[p][t0 - 1] : voidw;
        }
        uint64_t ga, ca, va, gb, cb, vb, ge, ce, ve;
        derive(q[0].x, q[1].x, q[2].x, q[3].x, ga, ca, va);
        derive(q[0].y, q[1].y, q[2].y, q[3].y, gb, cb, vb);
        derive(e[0], e[1], e[2], e[3], ge, ce, ve);
        uint64_t v_left = __shfl_up(vb, 1, 64);
        uint64_t g_right = __shfl_down(ga, 1, 64), c_right = __shfl_down(ca, 1, 64), v_right = __shfl_down(va, 1, 64);
        if (lane == 0) v_left = ve;
        if (lane == 63) { g_right = ge; c_right = ce; v_right = ve; }
        masks_of(ga, gb, ca, cb, v_left, va, vb, l, mp[0], mm[0]);
        masks_of(gb, g_right, cb, c_right, va, vb, v_right, l, mp[1], mm[1]);
    }
#else
    stage_tile<TW>(pl, t0, n_words_padded, sh);
#endif
#if !CRP_LIST_COMPACT
    auto stage_tables = [&]() {
        for (int k = tid; k < 256; k += EMIT_BLOCK) exp_tab[k] = CRP_EXP_TAB[k];
        if (LFIX == 20)
            for (int k = tid; k < CRP_SCORE_TAB_N; k += EMIT_BLOCK) score_tab[k] = CRP_SCORE_TAB[k];
    };
#endif
#if !CRP_TABLES_AFTER_PUBLISH && !CRP_LIST_COMPACT
    stage_tables();
#endif
#if !CRP_STREAM_MASKS
    __syncthreads();

    uint64_t mp[WPT], mm[WPT];
    thread_masks<WPT, TW>(sh, l, mp, mm);
#endif
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < WPT; ++k) c += (uint64_t)__popcll(mp[k]) | ((uint64_t)__popcll(mm[k]) << 32);
    uint64_t total;
    const uint64_t ex = block_exclusive_scan<(EMIT_BLOCK * WPT >= 1024)>(c, wave_tot, total);
    const uint32_t n_plus = (uint32_t)total, n_minus = (uint32_t)(total >> 32);
    const uint32_t n_all = n_plus + n_minus;
    uint64_t off_plus = 0, off_minus = 0;
    ChainArgs ch{};
    if (CHAINED) {
        ch = ChainArgs{chain + CHAIN_HEADER_WORDS, reinterpret_cast<uint32_t *>(chain) + 1, chain + 1, &s_excl, &s_flag, &s_next,
                       tile, gridDim.x, total, tile == mute_tile, timeout_ticks};
        if (tid == 0) {
            // mute_tile (normally none): a tile that never publishes, to exercise the time-out path
            if (tile != mute_tile) lookback_publish(ch.desc, tile, total);
            // leave the OTHER descriptor buffer zeroed for the next launch (no memset between scans)
            chain_next[CHAIN_HEADER_WORDS + tile] = 0;
            if (tile == 0) chain_next[0] = chain_next[1] = chain_next[2] = 0;  // (word 3 is the host's)
        }
#if CRP_PRIO_UNTIL_PUBLISH == 1
        __builtin_amdgcn_s_setprio(0);
#elif CRP_PRIO_UNTIL_PUBLISH == 2  // the wave that will resolve the tile's prefix keeps its priority until it has
        if (tid >= 64) __builtin_amdgcn_s_setprio(0);
#endif
        if (n_all == 0) {
            // nothing to store: the aggregate (0) is all later tiles need; only the last tile
            // must still learn its prefix, to publish the totals
            if (tile == ch.n_tiles - 1 && tid < 64) {
                uint64_t lb[LB_DEPTH];
                lookback_load(ch.desc, (int64_t)tile - 1, lb);
                chain_resolve(ch, lb);
            }
            return;
        }
    } else {
        const uint2 off = tile_off[tile];
        off_plus = off.x;
        off_minus = off.y;
        if (n_all == 0) return;
    }
#if CRP_EXPERIMENT_STOP == 1  // TIMING ONLY: load + masks + block scan + publish, nothing else
    if (CHAINED) return;
#endif
#if CRP_TABLES_AFTER_PUBLISH && !CRP_LIST_COMPACT
    // the scorer's tables are first read after the barrier that follows the hit-list build
    stage_tables();
#endif
    emit_rounds<WPT, TW, CAP, LFIX == 20, CHAINED>(sh, list, exp_tab, score_tab, mp, mm, ex, n_plus, n_minus, l, (uint32_t)(t0 * 64),
                                                   off_plus, off_minus, out, ch);
}

// Compact the kept hits of one staged tile and score them, CAP list entries per round.
template <int WPT, int TW, int CAP, bool PAM, bool CHAINED>
__device__ __forceinline__ void emit_rounds(uint64_t (*sh)[TW + 2], uint16_t *list, uint64_t *exp_tab, double *score_tab,
                                            const uint64_t (&mp)[WPT], const uint64_t (&mm)[WPT], uint64_t ex,
                                            uint32_t n_plus, uint32_t n_minus, int l, uint32_t tile_pos,
                                            uint64_t off_plus, uint64_t off_minus, const HitTables &out,
                                            const ChainArgs &ch)
{
    const int tid = threadIdx.x;
    const uint32_t n_all = n_plus + n_minus;
    bool resolved = !CHAINED;  // single pass: this lane has not picked up off_plus / off_minus yet
    // A tile with more kept hits than the list holds takes several rounds.  When each STRAND's hits fit (the usual
    // overflow: an unmasked tile of a GC-rich genome has ~1 700 + 1 700 of them), the rounds are the two strands:
    // each round peels only its own strand's masks, without capacity tests -- one list build's work in all, not two.
    const bool by_strand = CRP_LIST_BY_STRAND && n_all > (uint32_t)CAP && n_plus <= (uint32_t)CAP && n_minus <= (uint32_t)CAP;
    uint32_t hi_rank = 0;
    for (uint32_t lo_rank = 0; lo_rank < n_all; lo_rank = hi_rank) {
        hi_rank = by_strand ? (lo_rank == 0 ? n_plus : n_all) : min(lo_rank + (uint32_t)CAP, n_all);
        if (lo_rank) __syncthreads();  // previous round's readers are done
        // ---- compact: rank -> tile-local position, '+' hits first, then '-'
        // A tile whose hits all fit the list (the normal case; uniform over the workgroup) writes
        // without the per-entry capacity test.
        auto compact = [&](auto check, auto with_plus, auto with_minus) {
            constexpr bool CHECK = decltype(check)::value;
            constexpr bool PLUS = decltype(with_plus)::value, MINUS = decltype(with_minus)::value;
            uint32_t rp = (uint32_t)ex - lo_rank;                    // rank of next '+' hit, window-relative
            uint32_t rm = n_plus + (uint32_t)(ex >> 32) - lo_rank;  // same for '-'
#pragma unroll
            for (int k = 0; k < WPT; ++k) {
                const uint32_t wbase = (uint32_t)(tid * WPT + k) * 64u;
                // 32-bit halves: find-first-bit, clear-lowest and the compare are one
                // VALU instruction each instead of two
                if (PLUS) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        uint32_t m = (uint32_t)(mp[k] >> (32 * half));
                        while (m) {
                            const uint32_t b = __builtin_ctz(m);
                            m &= m - 1;
                            if (!CHECK || rp < (uint32_t)CAP) list[rp] = (uint16_t)(wbase + 32 * half + b);
                            ++rp;
                        }
                    }
                }
                if (MINUS) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        uint32_t m = (uint32_t)(mm[k] >> (32 * half));
                        while (m) {
                            const uint32_t b = __builtin_ctz(m);
                            m &= m - 1;
                            if (!CHECK || rm < (uint32_t)CAP) list[rm] = (uint16_t)(wbase + 32 * half + b);
                            ++rm;
                        }
                    }
                }
            }
        };
#if CRP_LIST_COMPACT
        // Balanced build (a tile whose hits all fit the list).  The peeling loops above run to the wave's
        // MAXIMUM popcount while most lanes have nothing left -- soft-masking clusters the hits, half the lanes
        // of a wave own no hit at all.  So the non-empty 32-bit halves are first compacted into a work list
        // (one 8-byte item each: mask, position base, rank base; slots from the compare's own lane mask +
        // v_mbcnt, no loop), and the peeling then runs over ITEMS, 64 at a time: no lane idles on an empty
        // half.  Scratch = the scorer's table block of LDS, which is staged only after the list is complete;
        // one word of every thread per pass, so a wave never has more than 256 items (its share holds 304).
        auto compact_balanced = [&]() {
            const int lane = tid & 63;
            constexpr uint32_t PER_WAVE = (256 + CRP_SCORE_TAB_N) / (EMIT_BLOCK / 64);
            static_assert(PER_WAVE >= 256, "work-list scratch: 4 halves x 64 lanes per pass");
            uint2 *const scr = reinterpret_cast<uint2 *>(exp_tab) + (tid >> 6) * PER_WAVE;
            uint32_t rp = (uint32_t)ex;                    // rank of this thread's next '+' hit
            uint32_t rm = n_plus + (uint32_t)(ex >> 32);  // same for '-'
#pragma unroll
            for (int k = 0; k < WPT; ++k) {
                const uint32_t wbase = (uint32_t)(tid * WPT + k) * 64u;
                uint32_t n_items = 0;  // wave-uniform
                auto push = [&](uint32_t m, uint32_t pbase, uint32_t r) {
                    const uint64_t nz = __ballot(m != 0);
                    const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nz, n_items));
                    if (m) scr[slot] = make_uint2(m, pbase | (r << 16));
                    n_items += (uint32_t)__popcll(nz);
                };
                const uint32_t p0 = (uint32_t)mp[k], p1 = (uint32_t)(mp[k] >> 32);
                const uint32_t m0 = (uint32_t)mm[k], m1 = (uint32_t)(mm[k] >> 32);
                push(p0, wbase, rp);
                rp += __popc(p0);
                push(p1, wbase + 32, rp);
                rp += __popc(p1);
                push(m0, wbase, rm);
                rm += __popc(m0);
                push(m1, wbase + 32, rm);
                rm += __popc(m1);
                // (LDS serves a wave's accesses in order: the items are there when they are read back)
                for (uint32_t b = 0; b < n_items; b += 64) {
                    uint2 it = make_uint2(0, 0);
                    if (b + lane < n_items) it = scr[b + lane];
                    uint32_t m = it.x;
                    const uint32_t pbase = it.y & 0xffffu;
                    uint16_t *dst = list + (it.y >> 16);
                    while (m) {
                        const uint32_t bit = __builtin_ctz(m);
                        m &= m - 1;
                        *dst++ = (uint16_t)(pbase + bit);
                    }
                }
            }
        };
        if (n_all <= (uint32_t)CAP) compact_balanced();
        else compact(std::true_type{}, std::true_type{}, std::true_type{});
#elif CRP_LIST_FASTPATH
        if (n_all <= (uint32_t)CAP) compact(std::false_type{}, std::true_type{}, std::true_type{});
        else if (by_strand && lo_rank == 0) compact(std::false_type{}, std::true_type{}, std::false_type{});
        else if (by_strand) compact(std::false_type{}, std::false_type{}, std::true_type{});
        else compact(std::true_type{}, std::true_type{}, std::true_type{});
#else
        compact(std::true_type{}, std::true_type{}, std::true_type{});
#endif
        if (CHAINED && tid == 0) *ch.s_next = 0;  // chunk counter of this round (read after the barrier below)
#if CRP_EXPERIMENT_STOP == 2  // TIMING ONLY: ... + the hit list
        if (CHAINED) return;
#endif
        __syncthreads();
#if CRP_LIST_COMPACT
        if (lo_rank == 0) {  // the list is complete, its scratch is free: now the scorer's tables move in
            for (int k = tid; k < 256; k += EMIT_BLOCK) exp_tab[k] = CRP_EXP_TAB[k];
            if (PAM)
                for (int k = tid; k < CRP_SCORE_TAB_N; k += EMIT_BLOCK) score_tab[k] = CRP_SCORE_TAB[k];
            __syncthreads();
        }
#endif
        // ---- one hit per lane: extract the 30-window, score, store
        const uint32_t n_round = hi_rank - lo_rank;
        struct Hit {
            uint32_t e, r;
            double pre, score;
        };
        auto compute = [&](uint32_t k) -> Hit {  // list entry k of this round
            Hit hit{list[k], lo_rank + k, -1.0, -1.0};
            const bool minus = hit.r >= n_plus;
            // '+': long_sequence = T(s[i-l-5 : i+5])       (CROPSR.py:421)
            // '-': long_sequence = T(R(s[j-2 : j+l+8]))     (CROPSR.py:432)
            // Python clamps the slice at len(s); the row is scored iff the result
            // has exactly 30 characters (CROPSR.py:458,466): for l = 20 a complete
            // window, for l > 20 a window cut to 30 by the end of the string, for
            // l < 20 never.
            const uint32_t q = 64u + hit.e - (minus ? 2u : (uint32_t)(l + 5));
            if (l >= 20) {
                uint32_t h = window31(sh[0], q), w = window31(sh[1], q);
                uint32_t u = window31(sh[2], q), a = window31(sh[3], q);
                const uint32_t vd = h & w & ~u & ~a;  // void positions
                const bool complete = (vd & 0x3fffffffu) == 0 && (l == 20 || (vd >> 30));
                h &= 0x3fffffffu;
                w &= 0x3fffffffu;
                u &= 0x3fffffffu;
                a &= 0x3fffffffu;
                uint32_t valid = a | u;  // acgtACGT, U, Z
                if (!minus) {
                    // get_gRNA_sequence (CROPSR.py:128): complement upper-case
                    // bases only, then reverse.  Complement = flip the low code bit.
                    h = reverse30(h);
                    w = reverse30(w ^ u);
                    valid = reverse30(valid);
                }
                if (complete) {
                    const uint32_t mG = h & w & valid, mC = h & ~w & valid;
                    const uint32_t mT = ~h & w & valid, mA = ~h & ~w & valid;
#if defined(CRP_EXPERIMENT_NO_SCORE)
                    hit.score = __hiloint2double((int)(mA ^ mT), (int)(mC ^ mG));
#else
                    crp_score_masks<PAM>(mA, mT, mC, mG, exp_tab, score_tab, hit.pre, hit.score);
#endif
                }
            }
            return hit;
        };
        auto store = [&](const Hit &hit) {
#if CRP_EXPERIMENT_NO_STORE  // TIMING ONLY: nothing is written unless an impossible score turns up
            if (hit.score != 12345.0) return;
#endif
            const uint32_t pos = tile_pos + hit.e;
            if (hit.r >= n_plus) {
                const uint64_t o = off_minus + (hit.r - n_plus);
                if (o < out.cap_minus) {
                    CRP_TABLE_STORE(&out.pos_minus[o], pos);
                    CRP_TABLE_STORE(&out.score_minus[o], hit.score);
                    if (out.pre_minus) CRP_TABLE_STORE(&out.pre_minus[o], hit.pre);
                }
            } else {
                const uint64_t o = off_plus + hit.r;
                if (o < out.cap_plus) {
                    CRP_TABLE_STORE(&out.pos_plus[o], pos);
                    CRP_TABLE_STORE(&out.score_plus[o], hit.score);
                    if (out.pre_plus) CRP_TABLE_STORE(&out.pre_plus[o], hit.pre);
                }
            }
        };
        if (!CHAINED) {
            for (uint32_t k = tid; k < n_round; k += EMIT_BLOCK) store(compute(k));
        } else {
            // Single pass: the table offsets are not known yet.  Every wave scores its first
            // hits; wave 0 then walks the descriptors (its round trip is exposed to that wave
            // only) and raises the LDS flag; the other waves go on scoring and store one
            // iteration behind, so they look at the flag one full iteration (~4 us) later and
            // normally find it set.  No workgroup barrier is involved.
            auto settle = [&]() {
                if (resolved) return;
                while (__hip_atomic_load(ch.s_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
                    __builtin_amdgcn_s_sleep(1);
                const uint64_t e = *ch.s_excl;
                off_plus = e & 0xffffffffull;
                off_minus = e >> 32;
                resolved = true;
            };
#if CRP_DYN_CHUNKS
            // The waves draw chunks of 64 list entries from a counter in LDS instead of owning every
            // fourth one: whatever time wave 0 spends in the look-back (round trip + waiting for
            // predecessors), the other three take over its share of the hits, so the look-back
            // lengthens no wave's critical path; the tail of the list is balanced the same way.
            const int lane = tid & 63;
            auto grab = [&]() -> uint32_t {
                uint32_t c = 0;
                if (lane == 0) c = atomicAdd(ch.s_next, 1u);
                return (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
            };
            auto look_back = [&]() {
                uint64_t lb[LB_DEPTH];
                lookback_load(ch.desc, (int64_t)ch.tile - 1, lb);
                chain_resolve(ch, lb);
                __hip_atomic_store(ch.s_flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            const bool scanner = lo_rank == 0 && tid < 64;  // wave-uniform: once per tile
#if CRP_LB_FIRST
            if (scanner) look_back();
#endif
            uint32_t c = grab();
            bool have_cur = false, first = true;
            Hit cur{};
            while (c * 64u < n_round) {
                const uint32_t k = c * 64u + (uint32_t)lane;
                Hit nxt{0xffffffffu, 0, -1.0, -1.0};
                if (k < n_round) nxt = compute(k);
                c = grab();  // (the atomic's round trip hides under the stores below)
#if !CRP_LB_FIRST
                if (scanner && first) look_back();
#endif
                first = false;
                if (have_cur) {
                    settle();
                    if (cur.e != 0xffffffffu) store(cur);
                }
                cur = nxt;
                have_cur = true;
            }
#if !CRP_LB_FIRST
            if (scanner && first) look_back();  // wave 0 drew no chunk at all
#endif
            if (have_cur) {
                settle();
                if (cur.e != 0xffffffffu) store(cur);
            }
#else
#if CRP_ROTATE_WAVES
            // The workgroup's slot (LDS, wave slots) is held until its LAST wave is done, and wave 0 also
            // resolves the tile's prefix.  So wave 0 takes the chunks of 64 rows nobody would miss: waves
            // 1, 2, 3 own chunks 0, 1, 2 (mod 4) and wave 0 chunk 3 -- when the row count is not a multiple
            // of 256 it is wave 0 that has one chunk less, not one more.
            uint32_t k = (uint32_t)(tid & 63) | ((((uint32_t)tid >> 6) + (EMIT_BLOCK / 64 - 1)) % (EMIT_BLOCK / 64)) << 6;
#else
            uint32_t k = tid;
#endif
            const bool any = k < n_round;
            Hit cur{};
#if CRP_LB_FIRST_STATIC
            // the whole look-back BEFORE the wave scores anything (nothing of the scorer is live then)
            if (lo_rank == 0 && tid < 64) {
                uint64_t lb0[LB_DEPTH];
                lookback_load(ch.desc, (int64_t)ch.tile - 1, lb0);
                chain_resolve(ch, lb0);
                __hip_atomic_store(ch.s_flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#if CRP_PRIO_UNTIL_PUBLISH == 2
                __builtin_amdgcn_s_setprio(0);
#endif
            }
#endif
#if CRP_LB_EARLY
            // the descriptors are requested BEFORE the wave scores its first hits and looked at after:
            // their round trip hides under that work (a snapshot that turns out too old costs a second one)
            uint64_t lb[LB_DEPTH];
            if (lo_rank == 0 && tid < 64) lookback_load(ch.desc, (int64_t)ch.tile - 1, lb);
#endif
            if (any) cur = compute(k);
            if (!CRP_LB_FIRST_STATIC && lo_rank == 0 && tid < 64) {  // wave-uniform: once per tile
#if !CRP_LB_EARLY
                uint64_t lb[LB_DEPTH];
                lookback_load(ch.desc, (int64_t)ch.tile - 1, lb);
#endif
                chain_resolve(ch, lb);
                __hip_atomic_store(ch.s_flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#if CRP_PRIO_UNTIL_PUBLISH == 2
                __builtin_amdgcn_s_setprio(0);
#endif
            }
            if (any) {
#if CRP_PIPE_UNROLL
                // two iterations per trip with the roles of the two parked hits swapped: no register copies
                Hit alt{};
                while (true) {
                    k += EMIT_BLOCK;
                    if (k >= n_round) { settle(); store(cur); break; }
                    alt = compute(k);
                    settle();
                    store(cur);
                    k += EMIT_BLOCK;
                    if (k >= n_round) { store(alt); break; }
                    cur = compute(k);
                    store(alt);
                }
#else
                for (k += EMIT_BLOCK; k < n_round; k += EMIT_BLOCK) {
                    const Hit nxt = compute(k);
                    settle();
                    store(cur);
                    cur = nxt;
                }
                settle();
                store(cur);
#endif
            }
#endif
        }
    }
}

// ------------------------------------------------------------ seam 2 kernel
// rs1_score on rows of 30 raw bytes: compare with 'A','T','C','G' exactly as
// CROPSR.py:300-309 does; no case folding here (the caller did it, :458).
__global__ __launch_bounds__(BLOCK) void score30_kernel(const uint8_t *__restrict__ rows, uint64_t n,
                                                         double *__restrict__ pre_out,
                                                         double *__restrict__ score_out)
{
    __shared__ uint64_t exp_tab[256];
    exp_tab[threadIdx.x] = CRP_EXP_TAB[threadIdx.x];
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        const uint8_t *r = rows + 30 * i;
        uint32_t mA = 0, mT = 0, mC = 0, mG = 0;
#pragma unroll
        for (int p = 0; p < 30; ++p) {
            const uint32_t ch = r[p];
            mA |= (uint32_t)(ch == 'A') << p;
            mT |= (uint32_t)(ch == 'T') << p;
            mC |= (uint32_t)(ch == 'C') << p;
            mG |= (uint32_t)(ch == 'G') << p;
        }
        double pre, score;
        crp_score_masks<false>(mA, mT, mC, mG, exp_tab, nullptr, pre, score);
        if (pre_out) pre_out[i] = pre;
        score_out[i] = score;
    }
}

// Same rows, scored in one of the reference's other accumulation orders
// (crp_score_generic.h).  Slow and generic on purpose: the host sends at most a
// few rows per written chunk here.
__global__ __launch_bounds__(64) void score30_order_kernel(const uint8_t *__restrict__ rows, uint64_t n, int order,
                                                            double *__restrict__ pre_out,
                                                            double *__restrict__ score_out)
{
    __shared__ uint64_t exp_tab[256];
    for (int k = threadIdx.x; k < 256; k += 64) exp_tab[k] = CRP_EXP_TAB[k];
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 64) {
        int8_t code[30];
        for (int p = 0; p < 30; ++p) {
            const uint32_t ch = rows[30 * i + p];
            code[p] = ch == 'A' ? 0 : ch == 'T' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : -1;
        }
        double s1, s2;
        if (order == CRP_ORDER_TAIL2) {
            s1 = crp_sum_tail2<false>(code);
            s2 = crp_sum_tail2<true>(code);
        } else {
            s1 = crp_sum_dot1<false>(code);
            s2 = crp_sum_dot1<true>(code);
        }
        const double pre = (((s1 + s2) + CRP_INTERSECT) + CRP_LOW_GC) * -1.0;
        if (pre_out) pre_out[i] = pre;
        score_out[i] = 1.0 / (1.0 + crp_exp(pre, exp_tab));
    }
}

// -------------------------------------------------------------- pack kernel
// Characters -> bit-planes with wavefront ballots: 64 lanes classify 64
// characters, four __ballot()s ARE the four plane words.
__host__ __device__ inline uint8_t classify_char(uint32_t ch)
{
    // bit0 = hi, bit1 = lo, bit2 = up, bit3 = ac
    switch (ch) {
        case 'A': case 'U': return 0xC;       // 'U' behaves as 'A': both str.replace chains
        case 'T': return 0xE;                 //   (CROPSR.py:120,128) start with A -> U
        case 'C': return 0xD;
        case 'G': return 0xF;
        case 'a': return 0x8;
        case 't': return 0xA;
        case 'c': return 0x9;
        case 'g': return 0xB;
        case 'Z': return 0x5;                 // C -> Z is the chains' second step: scores as C,
        default:  return 0x0;                 //   but 'Z' is not matched by the PAM regexes
    }
}

__global__ __launch_bounds__(BLOCK) void pack_kernel(const uint8_t *__restrict__ text, uint64_t len,
                                                      uint64_t n_words, uint64_t *__restrict__ hi,
                                                      uint64_t *__restrict__ lo, uint64_t *__restrict__ up,
                                                      uint64_t *__restrict__ ac)
{
    __shared__ uint8_t lut[256];
    __shared__ __attribute__((aligned(16))) uint8_t buf[BLOCK / 64][4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    lut[tid] = classify_char(tid);
    const uint64_t n_groups = (n_words + 63) / 64;             // 64 words = 4096 characters per wave
    const uint64_t n_block_iters = (n_groups + BLOCK / 64 - 1) / (BLOCK / 64);
    for (uint64_t bi = blockIdx.x; bi < n_block_iters; bi += gridDim.x) {
        const uint64_t group = bi * (BLOCK / 64) + wave;
        const uint64_t base = group * 4096;
        __syncthreads();  // lut ready / previous iteration's reads done
        if (group < n_groups) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const uint64_t off = base + (uint64_t)it * 1024 + (uint64_t)lane * 16;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (off + 16 <= len) {
                    v = *reinterpret_cast<const uint4 *>(text + off);
                } else if (off < len) {
                    uint8_t tmp[16];
                    for (int b = 0; b < 16; ++b) tmp[b] = off + b < len ? text[off + b] : 0;
                    v = *reinterpret_cast<const uint4 *>(tmp);
                }
                *reinterpret_cast<uint4 *>(&buf[wave][it * 1024 + lane * 16]) = v;
            }
        }
        __syncthreads();
        if (group < n_groups) {
            uint64_t w_hi = 0, w_lo = 0, w_up = 0, w_ac = 0;
            for (int t = 0; t < 64; ++t) {
                const uint64_t idx = base + (uint64_t)t * 64 + lane;
                const uint32_t nib = idx < len ? lut[buf[wave][t * 64 + lane]] : 0x3u;  // void past the end
                const uint64_t b0 = __ballot(nib & 1), b1 = __ballot(nib & 2);
                const uint64_t b2 = __ballot(nib & 4), b3 = __ballot(nib & 8);
                if (lane == t) { w_hi = b0; w_lo = b1; w_up = b2; w_ac = b3; }
            }
            const uint64_t w = group * 64 + lane;
            if (w < n_words) { hi[w] = w_hi; lo[w] = w_lo; up[w] = w_up; ac[w] = w_ac; }
        }
    }
}

// ------------------------------------------------------------ launch wrappers
hipError_t launch_count(hipStream_t s, const Planes &pl, uint64_t n_words_padded, int l, uint2 *tile_cnt,
                        uint32_t n_tiles)
{
    if (n_tiles == 0) return hipSuccess;
    constexpr uint32_t TPB = (TILE_WORDS > 512 ? TILE_WORDS : 512) / TILE_WORDS;  // emit tiles per count workgroup
    const dim3 grid((n_tiles + TPB - 1) / TPB);
    if (l == 20)
        hipLaunchKernelGGL(count_kernel<20>, grid, dim3(BLOCK), 0, s, pl, n_words_padded, l, tile_cnt);
    else
        hipLaunchKernelGGL(count_kernel<0>, grid, dim3(BLOCK), 0, s, pl, n_words_padded, l, tile_cnt);
    return hipGetLastError();
}

// one workgroup per chunk of SCAN_CHUNK_TILES tiles
hipError_t launch_tile_scan(hipStream_t s, const uint2 *tile_cnt, uint32_t n_tiles, uint2 *tile_off, uint64_t *totals)
{
    if (n_tiles == 0) return hipSuccess;
    const uint32_t n_chunks = (n_tiles + SCAN_CHUNK_TILES - 1) / SCAN_CHUNK_TILES;
    hipLaunchKernelGGL(tile_scan_kernel, dim3(n_chunks), dim3(1024), 0, s, tile_cnt, n_tiles, tile_off, totals);
    return hipGetLastError();
}

hipError_t launch_emit(hipStream_t s, const Planes &pl, uint64_t n_words_padded, int l, const uint2 *tile_off,
                       const HitTables &out)
{
    constexpr int TW = EMIT_BLOCK * TILE_WPT;
    const uint32_t n_tiles = (uint32_t)(n_words_padded / TW);
    if (n_tiles == 0) return hipSuccess;
    if (l == 20)
        hipLaunchKernelGGL((emit_kernel<TILE_WPT, false, 20>), dim3(n_tiles), dim3(EMIT_BLOCK), 0, s, pl, n_words_padded,
                           l, tile_off, (uint64_t *)nullptr, (uint64_t *)nullptr, out, 0xffffffffu, 0u);
    else
        hipLaunchKernelGGL((emit_kernel<TILE_WPT, false, 0>), dim3(n_tiles), dim3(EMIT_BLOCK), 0, s, pl, n_words_padded,
                           l, tile_off, (uint64_t *)nullptr, (uint64_t *)nullptr, out, 0xffffffffu, 0u);
    return hipGetLastError();
}

size_t chain_bytes(uint32_t n_tiles) { return (CHAIN_HEADER_WORDS + (size_t)n_tiles) * sizeof(uint64_t); }

hipError_t launch_emit_chained(hipStream_t s, const Planes &pl, uint64_t n_words_padded, int l, uint64_t *chain,
                               uint64_t *chain_next, const HitTables &out, uint32_t mute_tile, uint32_t timeout_ticks)
{
    constexpr int TW = EMIT_BLOCK * TILE_WPT;
    const uint32_t n_tiles = (uint32_t)(n_words_padded / TW);
    if (l == 20)
        hipLaunchKernelGGL((emit_kernel<TILE_WPT, true, 20>), dim3(n_tiles), dim3(EMIT_BLOCK), 0, s, pl, n_words_padded,
                           l, (const uint2 *)nullptr, chain, chain_next, out, mute_tile, timeout_ticks);
    else
        hipLaunchKernelGGL((emit_kernel<TILE_WPT, true, 0>), dim3(n_tiles), dim3(EMIT_BLOCK), 0, s, pl, n_words_padded,
                           l, (const uint2 *)nullptr, chain, chain_next, out, mute_tile, timeout_ticks);
    return hipGetLastError();
}

hipError_t launch_score30(hipStream_t s, const uint8_t *rows, uint64_t n, int order, double *pre, double *score)
{
    if (n == 0) return hipSuccess;
    if (order != CRP_ORDER_BODY4) {
        const uint64_t b = (n + 63) / 64;
        hipLaunchKernelGGL(score30_order_kernel, dim3((uint32_t)(b < 1024 ? b : 1024)), dim3(64), 0, s, rows, n,
                           order, pre, score);
        return hipGetLastError();
    }
    const uint64_t blocks = (n + BLOCK - 1) / BLOCK;
    const uint32_t grid = (uint32_t)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(score30_kernel, dim3(grid), dim3(BLOCK), 0, s, rows, n, pre, score);
    return hipGetLastError();
}

hipError_t launch_pack(hipStream_t s, const uint8_t *text, uint64_t len, uint64_t n_words, uint64_t *hi,
                       uint64_t *lo, uint64_t *up, uint64_t *ac)
{
    if (n_words == 0) return hipSuccess;
    const uint64_t iters = ((n_words + 63) / 64 + BLOCK / 64 - 1) / (BLOCK / 64);
    const uint32_t grid = (uint32_t)(iters < 4096 ? iters : 4096);
    hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(BLOCK), 0, s, text, len, n_words, hi, lo, up, ac);
    return hipGetLastError();
}

// ------------------------------------------------------- scored-row counter
// rows of a score table that carry a real score (the reference writes -1 for the others,
// CROPSR.py:466-468): the unit of the "gRNAs scored" metric, counted where the table lives.
__global__ __launch_bounds__(BLOCK) void count_scored_kernel(const double *__restrict__ score, uint64_t n,
                                                              unsigned long long *__restrict__ out)
{
    uint64_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK)
        c += score[i] != -1.0;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

hipError_t launch_count_scored(hipStream_t s, const double *score, uint64_t n, uint64_t *out)
{
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(count_scored_kernel, dim3((uint32_t)(blocks < 2048 ? blocks : 2048)), dim3(BLOCK), 0, s, score, n,
                       reinterpret_cast<unsigned long long *>(out));
    return hipGetLastError();
}

uint8_t host_classify_char(uint32_t ch) { return classify_char(ch); }

}  // namespace crp
