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
// Work decomposition: one 512-thread workgroup (eight wavefronts) per tile of 1 024 words.  PAM
// masks are 64-wide bit-parallel per lane; kept hits are ranked with popcounts and
// a block scan, compacted to an LDS list, and then scored one hit per lane so the
// f64 work is balanced and the table stores are coalesced.  Output order is
// ascending arena position per strand (offsets from a chained scan across tiles, or
// from the count / scan passes): bitwise reproducible, identical to the reference's row order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "crp_kernels.h"
#include "crp_score.h"
#include "crp_score_generic.h"

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
// `wave_tot` is LDS scratch of WAVES entries.  Returns the exclusive prefix,
// sets `total` to the block total.
// The per-lane counts are small (<= 128 per strand), so both fit one 32-bit
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

template <int WAVES>
__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t *wave_tot, uint64_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t v16 = (uint32_t)v | ((uint32_t)(v >> 32) << 16);  // plus | minus << 16
    const uint32_t inc16 = wave_inclusive_scan_u32(v16);
    const uint64_t inc = (uint64_t)(inc16 & 0xffffu) | ((uint64_t)(inc16 >> 16) << 32);
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const uint64_t t = wave_tot[w];
        if (w < wave) base += t;
        tot += t;
    }
    total = tot;
    return base + inc - v;
}

// -------------------------------------------------------------- pass 1: count
// Pure streaming pass: no LDS staging, no barrier before the loads.  A wavefront
// covers 128 consecutive words (two per lane, one 16-byte load per plane and lane
// = 64 bytes in flight per lane); the words a lane does not own come from its
// neighbours by wave shuffles of the DERIVED G / C / void masks, and only lanes 0
// and 63 touch memory for the words just outside the wave.  A workgroup of eight
// waves produces the counts of one emit tile.
__device__ __forceinline__ void derive(uint64_t hi, uint64_t lo, uint64_t up, uint64_t ac, uint64_t &g,
                                       uint64_t &c, uint64_t &v)
{
    g = hi & lo & up & ac;
    c = hi & ~lo & up & ac;
    v = hi & lo & ~up & ~ac;
}

// Kept-hit masks of one 64-position word, from the derived G / C / void masks of the word and its neighbours:
//   plus : (?=.GG) at i  <=>  G(i+1) & G(i+2);  keep  i - l >= 5   (CROPSR.py:419)
//   minus: (?=CC.) at j  <=>  C(j) & C(j+1) & exists(j+2);
//          keep  j+3 >= 5  and  j+3+l <= len+10               (CROPSR.py:430)
// "exists" / "index >= 0" / "index < len" are void tests at fixed distances.
__device__ __forceinline__ void masks_of(uint64_t g_c, uint64_t g_n, uint64_t c_c, uint64_t c_n, uint64_t v_p,
                                         uint64_t v_c, uint64_t v_n, int l, uint64_t &mplus, uint64_t &mminus)
{
    mplus = ahead(g_c, g_n, 1) & ahead(g_c, g_n, 2) & ~behind(v_p, v_c, l + 5);
    uint64_t m = c_c & ahead(c_c, c_n, 1) & ~ahead(v_c, v_n, 2) & ~behind(v_p, v_c, 2);
    if (l > 8) m &= ~ahead(v_c, v_n, l - 8);
    mminus = m;
}

// The words a lane owns (WPT = 2: one 16-byte load per plane; WPT = 1: one 8-byte load) and the kept-hit masks of those
// words.  w0 = first word of the lane's WAVE, which covers 64 * WPT consecutive words; the neighbouring words' derived
// masks come from the adjacent lanes, and lanes 0 / 63 read the word just outside the wave from memory (void beyond the
// arena).  Must be called by whole waves.
template <int WPT>
__device__ __forceinline__ void own_words_and_masks(const Planes &pl, uint64_t n_words_padded, uint64_t w0, int lane, int l,
                                                    uint64_t (&q)[4][WPT], uint64_t (&mp)[WPT], uint64_t (&mm)[WPT])
{
    static_assert(WPT == 1 || WPT == 2, "one or two words per owning lane");
    const uint64_t wa = w0 + (uint64_t)WPT * lane;
    if constexpr (WPT == 2) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(pl.plane[p] + wa);
            q[p][0] = v.x;
            q[p][WPT - 1] = v.y;
        }
    } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) q[p][0] = pl.plane[p][wa];
    }
    uint64_t e[4] = {ALL, ALL, 0, 0};  // void beyond the arena
    if (lane == 0 && w0 > 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) e[p] = pl.plane[p][w0 - 1];
    } else if (lane == 63 && w0 + 64 * WPT < n_words_padded) {
#pragma unroll
        for (int p = 0; p < 4; ++p) e[p] = pl.plane[p][w0 + 64 * WPT];
    }
    uint64_t ga, ca, va, gb, cb, vb, ge, ce, ve;
    derive(q[0][0], q[1][0], q[2][0], q[3][0], ga, ca, va);
    derive(q[0][WPT - 1], q[1][WPT - 1], q[2][WPT - 1], q[3][WPT - 1], gb, cb, vb);  // (WPT = 1: the same word)
    derive(e[0], e[1], e[2], e[3], ge, ce, ve);
    // left neighbour's last word, right neighbour's first word
    uint64_t v_left = __shfl_up(vb, 1, 64);
    uint64_t g_right = __shfl_down(ga, 1, 64), c_right = __shfl_down(ca, 1, 64), v_right = __shfl_down(va, 1, 64);
    if (lane == 0) v_left = ve;
    if (lane == 63) { g_right = ge; c_right = ce; v_right = ve; }
    if constexpr (WPT == 2) {
        masks_of(ga, gb, ca, cb, v_left, va, vb, l, mp[0], mm[0]);
        masks_of(gb, g_right, cb, c_right, va, vb, v_right, l, mp[WPT - 1], mm[WPT - 1]);
    } else {
        masks_of(ga, g_right, ca, c_right, v_left, va, v_right, l, mp[0], mm[0]);
    }
}

// LFIX > 0: guide length known at compile time (20, the reference's default), so
// every funnel shift has a constant amount and becomes one v_alignbit_b32 per half.
// G: the tile geometry (crp_kernels.h); one workgroup produces the counts of one emit tile, its owner waves only.
template <class G, int LFIX>
__global__ __launch_bounds__(G::BLOCK) void count_kernel(Planes pl, uint64_t n_words_padded, int l_arg,
                                                         uint2 *__restrict__ tile_cnt)
{
    const int l = LFIX > 0 ? LFIX : l_arg;
    const uint32_t tile = blockIdx.x;
    __shared__ uint64_t wave_tot[G::BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t c = 0;
    const uint64_t w0 = (uint64_t)tile * G::WORDS + (uint64_t)wave * 64 * G::WPT;  // first word of this wave
    if (wave < G::OWNERS / 64 && w0 < n_words_padded) {
        uint64_t q[4][G::WPT], mp[G::WPT], mm[G::WPT];
        own_words_and_masks<G::WPT>(pl, n_words_padded, w0, lane, l, q, mp, mm);
#pragma unroll
        for (int k = 0; k < G::WPT; ++k) c += (uint64_t)__popcll(mp[k]) | ((uint64_t)__popcll(mm[k]) << 32);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if (lane == 0) wave_tot[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0 && (uint64_t)tile * G::WORDS < n_words_padded) {
        uint64_t t = 0;
#pragma unroll
        for (int w = 0; w < G::BLOCK / 64; ++w) t += wave_tot[w];
        tile_cnt[tile] = make_uint2((uint32_t)t, (uint32_t)(t >> 32));
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
// dispatched in index order.  Spins are bounded all the same: on timeout *fail is set, every
// later look-back gives up at once, and the host repeats the scan with the count / scan / emit
// sequence instead of using the result -- a wrong assumption about dispatch order would cost
// time once, never a wrong table.
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

// Called by ONE wave.  Returns the exclusive prefix in every lane and publishes the inclusive one.
__device__ __forceinline__ uint64_t lookback_resolve(uint64_t *desc, uint32_t tile, uint64_t total, uint32_t *fail,
                                                     bool muted, uint32_t timeout_ticks)
{
    const int lane = threadIdx.x & 63;
    if (tile == 0) return 0;
    uint64_t v[LB_DEPTH];
    uint64_t excl = 0;
    int64_t base = (int64_t)tile - 1;
    lookback_load(desc, base, v);
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
    uint32_t tile, n_tiles;
    uint64_t total;     // this tile's (plus | minus << 32)
    bool muted;         // test hook: this tile publishes nothing
    uint32_t timeout_ticks;  // look-back allowance in ticks of the 100 MHz real-time counter
};

// ONE wave: look back, hand the tile's exclusive prefix to the workgroup through LDS, raise the flag;
// the last tile also publishes the table totals
__device__ __forceinline__ void chain_resolve(const ChainArgs &ch)
{
    const uint64_t e = lookback_resolve(ch.desc, ch.tile, ch.total, ch.fail, ch.muted, ch.timeout_ticks);
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
    __hip_atomic_store(ch.s_flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// exp table + chain-prefix tables as ONE image in global memory, in the order the emit kernel keeps them in LDS
struct TabsImage {
    uint64_t exp_tab[256];
    double score_tab[CRP_SCORE_TAB_N];
};
static_assert(sizeof(TabsImage) % 16 == 0, "staged in 16-byte units");
__device__ const TabsImage CRP_TABS = {{
#include "exp_table.inc"
                                       },
                                       CRP_SCORE_TAB_DATA};

// ---- table stores: buffer stores through per-tile descriptors
// A tile's rows go to table rows [first, first + n): the descriptor's base is moved to row `first - skip`
// (`skip` = the tile-local rank of the strand's first row: 0 for '+', n_plus for '-'), so that BOTH strands
// address their rows with the same per-lane offset, rank * element size, and the hardware's range check
// (offset + size <= num_records) is the capacity test: a row past the table's end is dropped by the store
// itself -- no 64-bit address arithmetic and no compare per row.  All of it is scalar (SALU): the inputs
// are wave-uniform.  num_records saturates at 2^32 - 1 bytes; a tile's own offsets stay below 2^20.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(void *table, uint64_t cap, uint64_t first, uint32_t skip,
                                                            uint32_t elem)
{
    const uint64_t room = cap > first ? cap - first : 0;
    uint64_t bytes = room ? (room + skip) * elem : 0;
    if (bytes > 0xffffffffull) bytes = 0xffffffffull;
    char *base = reinterpret_cast<char *>(table) + ((int64_t)first - (int64_t)skip) * (int64_t)elem;
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(uint32_t)bytes, 0x00020000);
}

struct TileStores {
    __amdgpu_buffer_rsrc_t pos[2], score[2], pre[2], seed[2];
};

constexpr int STORE_AUX = 2;  // gfx950 cache policy bits of a buffer store: bit 1 = nt (hit-table stores are non-temporal: -1.5 % at steady clocks)
typedef uint32_t u32x2 __attribute__((__vector_size__(2 * sizeof(uint32_t))));
__device__ __forceinline__ u32x2 f64_words(double v)
{
    u32x2 w;
    w[0] = (uint32_t)__double2loint(v);
    w[1] = (uint32_t)__double2hiint(v);
    return w;
}

__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

template <bool PRE, bool SEEDS>
__device__ __forceinline__ void tile_stores(TileStores &ts, const HitTables &out, uint64_t off_plus, uint64_t off_minus,
                                            uint32_t n_plus)
{
    ts.pos[0] = rows_rsrc(out.pos_plus, out.cap_plus, off_plus, 0, 4);
    ts.pos[1] = rows_rsrc(out.pos_minus, out.cap_minus, off_minus, n_plus, 4);
    ts.score[0] = rows_rsrc(out.score_plus, out.cap_plus, off_plus, 0, 8);
    ts.score[1] = rows_rsrc(out.score_minus, out.cap_minus, off_minus, n_plus, 8);
    if (PRE) {
        ts.pre[0] = rows_rsrc(out.pre_plus, out.cap_plus, off_plus, 0, 8);
        ts.pre[1] = rows_rsrc(out.pre_minus, out.cap_minus, off_minus, n_plus, 8);
    }
    if (SEEDS) {
        ts.seed[0] = rows_rsrc(out.seed_plus, out.cap_plus, off_plus, 0, 4);
        ts.seed[1] = rows_rsrc(out.seed_minus, out.cap_minus, off_minus, n_plus, 4);
    }
}

// ---- hit list: rank -> tile-local position, built by the lanes that own the mask words
// Appends the positions of the set bits of m (ascending) as 16-bit entries at LDS byte address `addr`, which
// advances past them.  `base` = tile-local position of bit 0 of m (a multiple of 32, so OR = ADD).
// The loop in ISA: find-first-bit, clear-lowest (2), position, address, compare-into-exec -- 6 VALU per trip
// (the compiler's version of the same C loop carries a separate trip counter and re-derives `base | 32`: 7-8 per
// trip, +1.7 % on the kernel).  A wave runs to the
// largest popcount among its lanes; lanes that are done sit out with their exec bit cleared.
__device__ __forceinline__ void peel32(uint32_t m, uint32_t base, uint32_t &addr)
{
    uint32_t b, t;
    uint64_t saved;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "v_cmpx_ne_u32_e32 vcc, 0, %[m]\n\t"
        "s_cbranch_execz 2f\n"
        "1:\n\t"
        "v_ffbl_b32_e32 %[b], %[m]\n\t"
        "v_add_u32_e32 %[t], -1, %[m]\n\t"
        "v_or_b32_e32 %[b], %[b], %[base]\n\t"
        "v_and_b32_e32 %[m], %[t], %[m]\n\t"
        "ds_write_b16 %[addr], %[b]\n\t"
        "v_add_u32_e32 %[addr], 2, %[addr]\n\t"
        "v_cmpx_ne_u32_e32 vcc, 0, %[m]\n\t"
        "s_cbranch_execnz 1b\n"
        "2:\n\t"
        "s_mov_b64 exec, %[saved]"
        : [m] "+v"(m), [addr] "+v"(addr), [b] "=&v"(b), [t] "=&v"(t), [saved] "=&s"(saved)
        : [base] "v"(base)
        : "vcc", "memory");
}

template <class G, bool PAM, bool CHAINED, bool PRE, bool SEEDS>
__device__ __forceinline__ void emit_rounds(uint64_t (*sh)[G::WORDS + 2], uint16_t *list, uint64_t *exp_tab, double *score_tab,
                                            const uint64_t (&mp)[G::WPT], const uint64_t (&mm)[G::WPT], uint64_t ex,
                                            uint32_t n_plus, uint32_t n_minus, int l, uint32_t tile_pos,
                                            uint64_t off_plus, uint64_t off_minus, const HitTables &out,
                                            const ChainArgs &ch);

// CHAINED = true : single pass.  Table offsets come from the decoupled look-back above;
//                  `chain` holds a 32-byte header -- unused (u32), fail flag (u32), the two
//                  table totals (u64 each, written by the last tile), a device pointer to a pinned
//                  host copy of the first three words (or 0; never touched by the kernel) -- then one
//                  descriptor per tile; it must be all zero at launch.  Launches alternate
//                  between two such buffers and every tile zeroes its slot of the other one
//                  (`chain_next`), so no memset runs between scans.  The tile
//                  publishes its counts right after the block scan -- reached without a barrier or an
//                  LDS round trip, at raised wave priority: later tiles wait for exactly that --, then
//                  stages the scorer's tables and builds its hit list; wave 0 resolves the prefix
//                  before it scores anything and raises an LDS flag, the other waves score their
//                  first hits meanwhile and look at the flag only when those are ready to be stored.
//                  Rows past the table capacities are dropped by the stores' range check, totals are
//                  published by the last tile.
// CHAINED = false: third pass of the count / scan / emit sequence (offsets from tile_off).
// PRE: the pre-sigmoid column is written too.  SEEDS: so is the off-target scan's raw seed word.
// G: the tile geometry (crp_kernels.h): G::BLOCK threads on a tile of G::WORDS words, of which the first G::OWNERS threads
//    own G::WPT words each (load them, derive their masks, build their part of the hit list); all threads score.
template <class G, bool CHAINED, int LFIX, bool PRE, bool SEEDS>
__global__ __launch_bounds__(G::BLOCK) __attribute__((amdgpu_waves_per_eu(6, 8))) void emit_kernel(
    Planes pl, uint64_t n_words_padded, int l_arg, const uint2 *__restrict__ tile_off, uint64_t *chain,
    uint64_t *__restrict__ chain_next, HitTables out, uint32_t mute_tile, uint32_t timeout_ticks)
{
    const int l = LFIX > 0 ? LFIX : l_arg;
    constexpr int TW = G::WORDS;
    __shared__ uint64_t sh[4][TW + 2];
    // exp table (256 words) + chain-prefix tables of the scorer
    __shared__ uint64_t tabs[256 + CRP_SCORE_TAB_N];
    uint64_t *const exp_tab = tabs;
    double *const score_tab = reinterpret_cast<double *>(tabs + 256);
    __shared__ uint64_t wave_tot[G::BLOCK / 64];
    __shared__ uint64_t s_excl;
    __shared__ uint32_t s_flag;
    __shared__ uint16_t list[G::LIST];
    static_assert(sizeof(sh) + sizeof(tabs) + sizeof(wave_tot) + sizeof(s_excl) + 8 + sizeof(list) <= G::LDS_LIMIT,
                  "workgroups per CU: see the geometry's LIST");

    const int tid = threadIdx.x;
    const uint32_t tile = blockIdx.x;  // dispatch order = index order: every tile this one waits for has started
    if (CHAINED && tid == 0) s_flag = 0;
    const uint64_t t0 = (uint64_t)tile * TW;
    // single-launch mode: later tiles wait for this tile's counts, so the short phase that produces
    // them (loads, masks, block scan) runs ahead of the scoring of the other workgroups on this CU
    if (CHAINED) __builtin_amdgcn_s_setprio(3);
    // Every owning thread loads the words it owns of each plane (one load per plane) and derives
    // its hit masks from those REGISTERS, taking the neighbouring words' G / C / void masks from the
    // adjacent lanes by wave shuffles (lanes 0 and 63 read the word just outside the wave from
    // memory) -- the count pass's scheme.  The words also go to LDS, but only the window extraction
    // after the block scan reads them there: no barrier and no LDS round trip stand between the loads
    // and the tile's counts.
    uint64_t mp[G::WPT], mm[G::WPT];
#pragma unroll
    for (int k = 0; k < G::WPT; ++k) mp[k] = mm[k] = 0;
    if (G::OWNERS == G::BLOCK || tid < G::OWNERS) {  // (whole waves: OWNERS is a multiple of 64)
        const int lane = tid & 63;
        const uint64_t w0 = t0 + (uint64_t)(64 * G::WPT) * (uint64_t)(tid >> 6);  // first word of this wave
        uint64_t q[4][G::WPT];
        own_words_and_masks<G::WPT>(pl, n_words_padded, w0, lane, l, q, mp, mm);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int k = 0; k < G::WPT; ++k) sh[p][1 + G::WPT * tid + k] = q[p][k];
        if (tid < 8) {  // the tile's own halo words, for windows that reach across its ends
            const int p = tid >> 1;
            const uint64_t voidw = (p < 2) ? ALL : 0ull;
            if (tid & 1) sh[p][TW + 1] = t0 + TW < n_words_padded ? pl.plane[p][t0 + TW] : voidw;
            else sh[p][0] = t0 > 0 ? pl.plane[p][t0 - 1] : voidw;
        }
    }
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < G::WPT; ++k) c += (uint64_t)__popcll(mp[k]) | ((uint64_t)__popcll(mm[k]) << 32);
    uint64_t total;
    const uint64_t ex = block_exclusive_scan<G::BLOCK / 64>(c, wave_tot, total);
    const uint32_t n_plus = (uint32_t)total, n_minus = (uint32_t)(total >> 32);
    const uint32_t n_all = n_plus + n_minus;
    uint64_t off_plus = 0, off_minus = 0;
    ChainArgs ch{};
    if (CHAINED) {
        ch = ChainArgs{chain + CHAIN_HEADER_WORDS, reinterpret_cast<uint32_t *>(chain) + 1, chain + 1, &s_excl, &s_flag,
                       tile, gridDim.x, total, tile == mute_tile, timeout_ticks};
        if (tid == 0) {
            // mute_tile (normally none): a tile that never publishes, to exercise the time-out path
            if (tile != mute_tile) lookback_publish(ch.desc, tile, total);
            // leave the OTHER descriptor buffer zeroed for the next launch (no memset between scans)
            chain_next[CHAIN_HEADER_WORDS + tile] = 0;
            if (tile == 0) chain_next[0] = chain_next[1] = chain_next[2] = 0;  // (word 3 is the host's)
        }
        __builtin_amdgcn_s_setprio(0);
        if (n_all == 0) {
            // nothing to store: the aggregate (0) is all later tiles need; only the last tile
            // must still learn its prefix, to publish the totals
            if (tile == ch.n_tiles - 1 && tid < 64) chain_resolve(ch);
            return;
        }
    } else {
        const uint2 off = tile_off[tile];
        off_plus = off.x;
        off_minus = off.y;
        if (n_all == 0) return;
    }
    // The scorer's tables (exp: 2 KiB; chain prefixes: 7.5 KiB, l = 20 only) are first read after the barrier that
    // follows the hit-list build; staging them only now keeps them out of the way of the counts that later tiles wait
    // for (-3.5 %).  (Staging by LDS-DMA, or requesting the words earlier and writing them here, both measured
    // slower: profiles/EXPERIMENTS.md.)
    for (int k = tid; k < 256; k += G::BLOCK) exp_tab[k] = CRP_TABS.exp_tab[k];
    if (LFIX == 20)
        for (int k = tid; k < CRP_SCORE_TAB_N; k += G::BLOCK) score_tab[k] = CRP_TABS.score_tab[k];
    emit_rounds<G, LFIX == 20, CHAINED, PRE, SEEDS>(sh, list, exp_tab, score_tab, mp, mm, ex, n_plus, n_minus, l, (uint32_t)(t0 * 64),
                                                 off_plus, off_minus, out, ch);
}

// Compact the kept hits of one staged tile and score them, G::LIST list entries per round.
template <class G, bool PAM, bool CHAINED, bool PRE, bool SEEDS>
__device__ __forceinline__ void emit_rounds(uint64_t (*sh)[G::WORDS + 2], uint16_t *list, uint64_t *exp_tab, double *score_tab,
                                            const uint64_t (&mp)[G::WPT], const uint64_t (&mm)[G::WPT], uint64_t ex,
                                            uint32_t n_plus, uint32_t n_minus, int l, uint32_t tile_pos,
                                            uint64_t off_plus, uint64_t off_minus, const HitTables &out,
                                            const ChainArgs &ch)
{
    constexpr uint32_t CAP = G::LIST;
    const int tid = threadIdx.x;
    const uint32_t n_all = n_plus + n_minus;
    TileStores ts;
    bool resolved = !CHAINED;  // single pass: this wave has not picked up the tile's table offsets yet
    if (!CHAINED) tile_stores<PRE, SEEDS>(ts, out, uniform64(off_plus), uniform64(off_minus), n_plus);
    // A tile with more kept hits than the list holds takes several rounds.  When each STRAND's hits fit (the usual
    // overflow: an unmasked tile of a GC-rich genome has ~1 700 + 1 700 of them), the rounds are the two strands:
    // each round peels only its own strand's masks, without capacity tests -- one list build's work in all, not two.
    const bool by_strand = n_all > CAP && n_plus <= CAP && n_minus <= CAP;
    const uint32_t list_lds = (uint32_t)(uintptr_t)list;  // LDS byte address of the list
    uint32_t hi_rank = 0;
    for (uint32_t lo_rank = 0; lo_rank < n_all; lo_rank = hi_rank) {
        hi_rank = by_strand ? (lo_rank == 0 ? n_plus : n_all) : min(lo_rank + CAP, n_all);
        if (lo_rank) __syncthreads();  // previous round's readers are done
        // ---- compact: rank -> tile-local position, '+' hits first, then '-'
        const uint32_t rp0 = (uint32_t)ex - lo_rank;                    // window-relative rank of this thread's first '+' hit
        const uint32_t rm0 = n_plus + (uint32_t)(ex >> 32) - lo_rank;  // same for '-'
        if (n_all <= CAP || by_strand) {
            // every entry of this round fits: no per-entry capacity test
            const bool with_plus = !by_strand || lo_rank == 0, with_minus = !by_strand || lo_rank != 0;
            uint32_t ap = list_lds + 2 * rp0, am = list_lds + 2 * rm0;
            if (G::OWNERS == G::BLOCK || tid < G::OWNERS) {  // (wave-uniform; the other waves own no words, hence list nothing)
#pragma unroll
                for (int k = 0; k < G::WPT; ++k) {
                    const uint32_t wbase = (uint32_t)(tid * G::WPT + k) * 64u;
                    if (with_plus) {
                        peel32((uint32_t)mp[k], wbase, ap);
                        peel32((uint32_t)(mp[k] >> 32), wbase + 32, ap);
                    }
                    if (with_minus) {
                        peel32((uint32_t)mm[k], wbase, am);
                        peel32((uint32_t)(mm[k] >> 32), wbase + 32, am);
                    }
                }
            }
        } else {
            // rare (a tile with more than G::LIST hits on ONE strand: poly-G and the like): windows of CAP ranks
            uint32_t rp = rp0, rm = rm0;
#pragma unroll
            for (int k = 0; k < G::WPT; ++k) {
                const uint32_t wbase = (uint32_t)(tid * G::WPT + k) * 64u;
                for (uint64_t m = mp[k]; m; m &= m - 1, ++rp)
                    if (rp < CAP) list[rp] = (uint16_t)(wbase + __builtin_ctzll(m));
                for (uint64_t m = mm[k]; m; m &= m - 1, ++rm)
                    if (rm < CAP) list[rm] = (uint16_t)(wbase + __builtin_ctzll(m));
            }
        }
        __syncthreads();
        // ---- one hit per lane: extract the 30-window, score, store
        const uint32_t n_round = hi_rank - lo_rank;
        struct Hit {
            uint32_t e, r, seed;
            double pre, score;
        };
        // what a row needs from LDS before anything can be computed: its list entry and the four 31-bit windows
        struct Fetched {
            uint32_t e, r, h, w, u, a;
        };
        auto fetch = [&](uint32_t k) -> Fetched {  // list entry k of this round
            Fetched f{list[k], lo_rank + k, 0, 0, 0, 0};
            // '+': long_sequence = T(s[i-l-5 : i+5])       (CROPSR.py:421)
            // '-': long_sequence = T(R(s[j-2 : j+l+8]))     (CROPSR.py:432)
            const uint32_t q = 64u + f.e - (f.r >= n_plus ? 2u : (uint32_t)(l + 5));
            if (l >= 20) {
                f.h = window31(sh[0], q);
                f.w = window31(sh[1], q);
                f.u = window31(sh[2], q);
                f.a = window31(sh[3], q);
            }
            return f;
        };
        auto score = [&](const Fetched &f) -> Hit {
            Hit hit{f.e, f.r, SEED_RAW_NONE, -1.0, -1.0};
            const bool minus = hit.r >= n_plus;
            // Python clamps the slice at len(s); the row is scored iff the result
            // has exactly 30 characters (CROPSR.py:458,466): for l = 20 a complete
            // window, for l > 20 a window cut to 30 by the end of the string, for
            // l < 20 never.
            if (l >= 20) {
                uint32_t h = f.h, w = f.w, u = f.u, a = f.a;
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
                if (SEEDS) {
                    // Off-target scan (crp_offtarget.hip): seed character k (k = 0 next to the PAM) is character k of the
                    // `sequence` column -- '+': s[i-1-k] = window bit l+4-k = bit 25-l+k after the reversal; '-': s[j+3+k]
                    // = window bit 5+k.  For l = 20 (the only length this variant is built for) both are bit 5+k.
                    static_assert(!SEEDS || PAM, "seed words are emitted by the l = 20 variant only");
                    if (((valid >> 5) & 0xfffu) == 0xfffu) hit.seed = (((h >> 5) & 0xfffu) << 12) | ((w >> 5) & 0xfffu);
                }
                if (complete) {
                    const uint32_t mG = h & w & valid, mC = h & ~w & valid;
                    const uint32_t mT = ~h & w & valid, mA = ~h & ~w & valid;
                    crp_score_masks<PAM>(mA, mT, mC, mG, exp_tab, score_tab, hit.pre, hit.score);
                }
            }
            return hit;
        };
        auto store = [&](const Hit &hit) {
            const uint32_t pos = tile_pos + hit.e;
            const int o4 = (int)(hit.r * 4u), o8 = (int)(hit.r * 8u);
            const bool minus = hit.r >= n_plus;
            // Two guarded regions, one per strand, NOT an if / else: the compiler would fold an if / else into one store
            // through a per-lane SELECTED descriptor, which no longer lives in SGPRs (a "waterfall" loop per store).
            // A 64-row chunk is of one strand except at the seam, so one of the regions is normally skipped (execz).
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (minus == (st == 1)) {
                    __builtin_amdgcn_raw_buffer_store_b32(pos, ts.pos[st], o4, 0, STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b64(f64_words(hit.score), ts.score[st], o8, 0, STORE_AUX);
                    if (PRE) __builtin_amdgcn_raw_buffer_store_b64(f64_words(hit.pre), ts.pre[st], o8, 0, STORE_AUX);
                    if (SEEDS) __builtin_amdgcn_raw_buffer_store_b32(hit.seed, ts.seed[st], o4, 0, STORE_AUX);
                    // (distinct tails keep the optimiser from sinking the two regions' stores into one)
                    if (st == 0) asm volatile("; '+' rows stored" ::: "memory");
                    else asm volatile("; '-' rows stored" ::: "memory");
                }
            }
        };
        // Rows k, k + 256, ... of one lane.  `before_first_store` runs once, after the first row is scored: from the
        // second row on a row is stored as soon as it is scored -- nothing is parked in registers.  (Issuing the next
        // row's LDS reads ahead of the current row's arithmetic was measured: +4 %, profiles/EXPERIMENTS.md.)
        auto run_rows = [&](uint32_t k, auto before_first_store) {
            const bool any = k < n_round;
            Hit first{};
            if (any) first = score(fetch(k));
            before_first_store(any);
            if (any) {
                store(first);
                for (k += G::BLOCK; k < n_round; k += G::BLOCK) store(score(fetch(k)));
            }
        };
        if (!CHAINED) {
            run_rows(tid, [](bool) {});
        } else {
            // Single pass.  The workgroup's slot (LDS, wave slots) is held until its LAST wave is done, and wave 0
            // also resolves the tile's prefix.  So wave 0 takes the chunks of 64 rows nobody would miss: waves 1 ... 7
            // own chunks 0 ... 6 (mod 8) and wave 0 chunk 7 -- when the row count is not a multiple of 512 it is
            // wave 0 that has one chunk less, not one more.
            const uint32_t k0 = (uint32_t)(tid & 63) | ((((uint32_t)tid >> 6) + (G::BLOCK / 64 - 1)) % (G::BLOCK / 64)) << 6;
            // the whole look-back BEFORE wave 0 scores anything (nothing of the scorer is live then)
            if (lo_rank == 0 && tid < 64) chain_resolve(ch);
            // The other waves score their first rows meanwhile and need the offsets only to STORE them: by then
            // (one scoring iteration, ~4 us, after wave 0 started looking back) the flag is normally up.  From
            // the second iteration on a row is stored as soon as it is scored -- nothing is parked in registers.
            run_rows(k0, [&](bool any) {
                // wave-uniform (the descriptors must stay in SGPRs): a wave with rows waits for the flag once per tile
                if (!resolved && __ballot(any)) {
                    while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ch.s_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0)
                        __builtin_amdgcn_s_sleep(1);
                    const uint64_t e = uniform64(*ch.s_excl);
                    tile_stores<PRE, SEEDS>(ts, out, e & 0xffffffffull, e >> 32, n_plus);
                    resolved = true;
                }
            });
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

// The same packing for a BATCH of small contigs that were copied to the device back to back (crp_arena_add_contigs_ascii):
// one wave per group of <= 64 words of ONE contig; `groups` says where the group's characters start in `text`, how many
// of them are real (the rest of its words is void: the contig's tail and its separator word) and which words it fills.
__global__ __launch_bounds__(BLOCK) void pack_groups_kernel(const uint8_t *__restrict__ text, const PackGroup *__restrict__ groups,
                                                             uint32_t n_groups, uint64_t *__restrict__ hi, uint64_t *__restrict__ lo,
                                                             uint64_t *__restrict__ up, uint64_t *__restrict__ ac)
{
    __shared__ uint8_t lut[256];
    __shared__ __attribute__((aligned(16))) uint8_t buf[BLOCK / 64][4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    lut[tid] = classify_char(tid);
    const uint32_t n_iters = (n_groups + BLOCK / 64 - 1) / (BLOCK / 64);
    for (uint32_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const uint32_t g = it * (BLOCK / 64) + wave;
        __syncthreads();  // lut ready / previous iteration's reads done
        PackGroup pg{0, 0, 0, 0, 0};
        if (g < n_groups) {
            pg = groups[g];
            // (reads up to 4096 bytes from the group's start: inside the device text buffer by construction, what lies
            // behind the group's own characters is masked below)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                *reinterpret_cast<uint4 *>(&buf[wave][k * 1024 + lane * 16]) =
                    *reinterpret_cast<const uint4 *>(text + pg.src + (uint32_t)k * 1024 + (uint32_t)lane * 16);
        }
        __syncthreads();
        if (g < n_groups) {
            uint64_t w_hi = 0, w_lo = 0, w_up = 0, w_ac = 0;
            for (uint32_t t = 0; t < pg.n_words; ++t) {
                const uint32_t idx = t * 64 + lane;
                const uint32_t nib = idx < pg.n_chars ? lut[buf[wave][idx]] : 0x3u;  // void past the end
                const uint64_t b0 = __ballot(nib & 1), b1 = __ballot(nib & 2);
                const uint64_t b2 = __ballot(nib & 4), b3 = __ballot(nib & 8);
                if ((uint32_t)lane == t) { w_hi = b0; w_lo = b1; w_up = b2; w_ac = b3; }
            }
            if ((uint32_t)lane < pg.n_words) {
                const uint64_t w = pg.dst_word + lane;
                hi[w] = w_hi;
                lo[w] = w_lo;
                up[w] = w_up;
                ac[w] = w_ac;
            }
        }
    }
}

// ------------------------------------------------------------ launch wrappers
template <class G>
static hipError_t launch_count_geo(hipStream_t s, const Planes &pl, uint64_t n_words_padded, int l, uint2 *tile_cnt, uint32_t n_tiles)
{
    if (l == 20)
        hipLaunchKernelGGL((count_kernel<G, 20>), dim3(n_tiles), dim3(G::BLOCK), 0, s, pl, n_words_padded, l, tile_cnt);
    else
        hipLaunchKernelGGL((count_kernel<G, 0>), dim3(n_tiles), dim3(G::BLOCK), 0, s, pl, n_words_padded, l, tile_cnt);
    return hipGetLastError();
}

#define CRP_BY_GEOMETRY(geo, CALL)                 \
    switch (geo) {                                 \
        case GEO_LARGE: return CALL(GeoLarge);     \
        case GEO_SMALL: return CALL(GeoSmall);     \
        default: return hipErrorInvalidValue;      \
    }

hipError_t launch_count(hipStream_t s, int geo, const Planes &pl, uint64_t n_words_padded, int l, uint2 *tile_cnt,
                        uint32_t n_tiles)
{
    if (n_tiles == 0) return hipSuccess;
#define CRP_CALL(G) launch_count_geo<G>(s, pl, n_words_padded, l, tile_cnt, n_tiles)
    CRP_BY_GEOMETRY(geo, CRP_CALL)
#undef CRP_CALL
}

// one workgroup per chunk of SCAN_CHUNK_TILES tiles
hipError_t launch_tile_scan(hipStream_t s, const uint2 *tile_cnt, uint32_t n_tiles, uint2 *tile_off, uint64_t *totals)
{
    if (n_tiles == 0) return hipSuccess;
    const uint32_t n_chunks = (n_tiles + SCAN_CHUNK_TILES - 1) / SCAN_CHUNK_TILES;
    hipLaunchKernelGGL(tile_scan_kernel, dim3(n_chunks), dim3(1024), 0, s, tile_cnt, n_tiles, tile_off, totals);
    return hipGetLastError();
}

// the emit kernel's variants: the tile geometry; guide length 20 (compile-time windows, chain-prefix tables) or any; with
// or without the pre-sigmoid column; with or without the off-target scan's seed words (l = 20 only)
template <class G, bool CHAINED>
static hipError_t launch_emit_variant(hipStream_t s, const Planes &pl, uint64_t n_words_padded, int l, const uint2 *tile_off,
                                      uint64_t *chain, uint64_t *chain_next, const HitTables &out, uint32_t mute_tile,
                                      uint32_t timeout_ticks)
{
    const uint32_t n_tiles = (uint32_t)(n_words_padded / G::WORDS);
    if (n_tiles == 0) return hipSuccess;
    const bool pre = out.pre_plus != nullptr, seeds = out.seed_plus != nullptr;
    if (seeds && l != 20) return hipErrorInvalidValue;
#define CRP_LAUNCH(LFIX, PRE, SEEDS)                                                                                       \
    hipLaunchKernelGGL((emit_kernel<G, CHAINED, LFIX, PRE, SEEDS>), dim3(n_tiles), dim3(G::BLOCK), 0, s, pl, n_words_padded, l, \
                       tile_off, chain, chain_next, out, mute_tile, timeout_ticks)
    if (l == 20) {
        if (seeds) {
            if (pre) CRP_LAUNCH(20, true, true);
            else CRP_LAUNCH(20, false, true);
        } else {
            if (pre) CRP_LAUNCH(20, true, false);
            else CRP_LAUNCH(20, false, false);
        }
    } else {
        if (pre) CRP_LAUNCH(0, true, false);
        else CRP_LAUNCH(0, false, false);
    }
#undef CRP_LAUNCH
    return hipGetLastError();
}

hipError_t launch_emit(hipStream_t s, int geo, const Planes &pl, uint64_t n_words_padded, int l, const uint2 *tile_off,
                       const HitTables &out)
{
#define CRP_CALL(G) launch_emit_variant<G, false>(s, pl, n_words_padded, l, tile_off, nullptr, nullptr, out, 0xffffffffu, 0u)
    CRP_BY_GEOMETRY(geo, CRP_CALL)
#undef CRP_CALL
}

size_t chain_bytes(uint32_t n_tiles) { return (CHAIN_HEADER_WORDS + (size_t)n_tiles) * sizeof(uint64_t); }

hipError_t launch_emit_chained(hipStream_t s, int geo, const Planes &pl, uint64_t n_words_padded, int l, uint64_t *chain,
                               uint64_t *chain_next, const HitTables &out, uint32_t mute_tile, uint32_t timeout_ticks)
{
#define CRP_CALL(G) launch_emit_variant<G, true>(s, pl, n_words_padded, l, nullptr, chain, chain_next, out, mute_tile, timeout_ticks)
    CRP_BY_GEOMETRY(geo, CRP_CALL)
#undef CRP_CALL
}

int tile_words(int geo) { return geo == GEO_SMALL ? GeoSmall::WORDS : GeoLarge::WORDS; }

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

hipError_t launch_pack_groups(hipStream_t s, const uint8_t *text, const PackGroup *groups, uint32_t n_groups, uint64_t *hi,
                              uint64_t *lo, uint64_t *up, uint64_t *ac)
{
    if (n_groups == 0) return hipSuccess;
    const uint32_t iters = (n_groups + BLOCK / 64 - 1) / (BLOCK / 64);
    hipLaunchKernelGGL(pack_groups_kernel, dim3(iters < 4096 ? iters : 4096), dim3(BLOCK), 0, s, text, groups, n_groups, hi, lo, up, ac);
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

// ------------------------------------------------------- composition of an arena
// how many characters are upper-case A/C/G/T (bit set in both the `up` and the `ac` plane; void positions have neither)
__global__ __launch_bounds__(BLOCK) void count_plain_kernel(const uint64_t *__restrict__ up, const uint64_t *__restrict__ ac,
                                                             uint64_t n_words, unsigned long long *__restrict__ out)
{
    uint64_t c = 0;
    for (uint64_t w = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * BLOCK)
        c += (uint64_t)__popcll(up[w] & ac[w]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

hipError_t launch_count_plain(hipStream_t s, const uint64_t *up, const uint64_t *ac, uint64_t n_words, uint64_t *out)
{
    if (n_words == 0) return hipSuccess;
    const uint64_t blocks = (n_words + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(count_plain_kernel, dim3((uint32_t)(blocks < 2048 ? blocks : 2048)), dim3(BLOCK), 0, s, up, ac, n_words,
                       reinterpret_cast<unsigned long long *>(out));
    return hipGetLastError();
}

uint8_t host_classify_char(uint32_t ch) { return classify_char(ch); }

}  // namespace crp
