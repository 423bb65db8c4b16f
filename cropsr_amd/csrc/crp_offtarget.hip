// crp_offtarget.hip -- genome-wide off-target seed scan (BASELINE.json configs[4]).
//
// The reference has no off-target step at all (its only alignment code is the dead bowtie2
// shell-out of prmrdsgn2.py:139-160), so the DEFINITION is this engine's own, stated on the
// reference's own strings (include/cropsr_hip.h):
//   site   a kept hit (CROPSR.py:419 / :430) whose `sequence` column (CROPSR.py:420 / :431) starts
//          with 12 bases -- after the scoring transform of CROPSR.py:458 -- next to the PAM
//   count  c_k(g), k = 0..3: the OTHER sites whose 12-base seed differs from g's in exactly k places
//
// Method (exact, no pairwise comparison):
//   1. ot_seed_kernel     one lane per kept hit: 12 characters next to the PAM out of the arena's
//      + partition        bit-planes (two 64-bit words per plane), oriented like the scoring string
//                         (crp_kernels.hip emit_rounds), Morton-coded to 24 bits; the sites are then
//                         partitioned by the high 12 seed bits and counted bucket by bucket in LDS:
//                         a histogram of the 4^12 seeds without a global atomic per site;
//   2. ot_ball_kernel x3  the Hamming-ball sums of that histogram for ALL seeds at once.  With
//                         F_p[x][d] = sum of hist[y] over the y that agree with x on positions >= p
//                         and differ in exactly d of the positions < p,
//                             F_{p+1}[x][d] = F_p[x][d] + S_p[x][d-1] - F_p[x][d-1],
//                             S_p[x][d] = sum over the 4 bases b at position p of F_p[x with b at p][d]
//                         -- a group of 4 entries per step.  Four positions (8 index bits) per
//                         pass, 4096-seed tiles through LDS: pass 0 walks contiguous tiles, passes 1
//                         and 2 walk 256-byte pieces at strides of 4 KiB and 1 MiB;
//   3. ot_lookup_kernel   one 16-byte read of the ball table per hit, minus the hit itself.
// Cost: two passes over the hit tables, 1.4 GB of table traffic whatever the genome, one 16-byte
// gather per hit: HBM-bound like the scan, no MFMA (sums of integers).
// Multi-GPU: every rank adds its own sites; the histogram is summed over the ranks by one RCCL
// all-reduce (64 MiB, crp_comm.cpp) between steps 1 and 2 -- the one bandwidth-heavy xGMI collective
// of the engine -- and every rank then solves and looks up its own hits.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "crp_internal.h"
#include "crp_roctx.h"

namespace crp {

constexpr uint32_t OT_SEEDS = 1u << (2 * CRP_OT_SEED_LEN);  // 4^12
constexpr uint32_t OT_NOT_A_SITE = 0xffffffffu;
constexpr uint32_t OT_NOT_OWNED = 0xfffffffeu;

// 12 consecutive arena positions starting at q, from one bit-plane: ONE unaligned 32-bit load at the byte the
// window starts in (12 + 7 bits; gfx950 serves unaligned global loads) instead of one or two aligned 64-bit words.
// (The bytes behind a window always exist: an arena ends in void padding words.)
struct __attribute__((packed)) UnalignedU32 {
    uint32_t v;
};
__device__ __forceinline__ uint32_t window12(const uint64_t *plane, uint64_t q)
{
    const uint32_t v = reinterpret_cast<const UnalignedU32 *>(reinterpret_cast<const uint8_t *>(plane) + (q >> 3))->v;
    return (v >> ((uint32_t)q & 7u)) & 0xfffu;
}

// bit k of x -> bit 2k
__device__ __forceinline__ uint32_t spread12(uint32_t x)
{
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// Seed of the hit at arena position p.  Seed character k (k = 0 next to the PAM) is character k of
// the `sequence` column after .replace('U','T').upper():
//   '+'  sequence = get_gRNA_sequence(s[i-l:i])  (CROPSR.py:420,128): s[i-1-k], upper-case bases
//        complemented, lower-case ones only reversed
//   '-'  sequence = get_gRNA_sequence(get_reverse_complement(s[j+3:j+3+l]))  (:431): s[j+3+k]
// A character is a base iff it is one of acgtACGT, U (== A) or Z (== C) -- `ac | up` of the planes,
// exactly the scorer's `valid` mask; anything else, or the end of the string (void), is not.
// The histogram is built WITHOUT one global atomic per site (52 M random single-dword atomics run at
// ~19 G/s on MI355X: 2.7 ms of a 4.2 ms step).  Seeds are partitioned by their high 12 bits instead:
//   ot_seed_kernel       seeds of the hits + per-workgroup counts of the 4096 buckets in LDS, added to
//                        global bucket totals with CONTIGUOUS atomics (64 consecutive counters per wave
//                        instruction: the fast path, one instruction per 64 buckets and workgroup)
//   ot_bucket_scan       exclusive scan of the 4096 totals (one workgroup)
//   ot_partition1_kernel / ot_partition_kernel
//                        two scatter passes (top 6 seed bits, then the next 6): every workgroup recounts its
//                        chunk, reserves a range in every bucket with one contiguous atomic sweep and scatters
//                        there -- whole codes first, the low 12 bits at the end (order inside a bucket does
//                        not matter); two levels so that every run a chunk writes is a few hundred bytes long
//   ot_bucket_hist       one workgroup per bucket: 4096-bin histogram of its entries in LDS, added to
//                        the bucket's slice of the global histogram with plain 16-byte accesses
// Chunks: workgroup w owns hits [w * OT_CHUNK, (w + 1) * OT_CHUNK) of a strand's table in both passes.
constexpr uint32_t OT_BUCKETS = 4096;
#ifndef CRP_OT_CHUNK
#define CRP_OT_CHUNK 16384
#endif
constexpr uint32_t OT_CHUNK = CRP_OT_CHUNK;

template <bool MINUS>
__device__ __forceinline__ uint32_t seed_of_windows(uint32_t h, uint32_t w, uint32_t u, uint32_t a, uint64_t p,
                                                    const uint64_t *__restrict__ own, uint32_t n_own)
{
    const bool valid = ((a | u) & 0xfffu) == 0xfffu;
    if (!MINUS) {
        w ^= u;  // complement = flip the low code bit, upper case only
        h = __brev(h) >> 20;
        w = __brev(w) >> 20;
    }
    if (!valid) return OT_NOT_A_SITE;
    if (n_own) {  // the range with the largest begin <= p
        uint32_t lo = 0, hi = n_own;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (own[2 * mid] <= p) lo = mid + 1;
            else hi = mid;
        }
        if (!(lo > 0 && p < own[2 * (lo - 1) + 1])) return OT_NOT_OWNED;
    }
    return (spread12(h) << 1) | spread12(w);
}

// A chunk's entries, FOUR consecutive ones per lane and trip: one 16-byte load where four dependent 4-byte loads stood --
// these passes are latency-bound (six waves per SIMD, every trip a load, an LDS atomic that returns a slot and a store that
// wait for each other), not bound by the LDS atomics themselves -- and the four atomics / stores of a trip are independent.
// src + lo is 16-byte aligned (chunks start at multiples of OT_CHUNK); the last chunk's ragged end goes one by one.
#ifndef CRP_OT_VEC4
#define CRP_OT_VEC4 1
#endif
template <class F>
__device__ __forceinline__ void for_chunk(const uint32_t *__restrict__ src, uint64_t lo, uint64_t hi, F f)
{
#if CRP_OT_VEC4
    const uint64_t body = lo + ((hi - lo) & ~(uint64_t)3);
    uint64_t t = lo + 4 * (uint64_t)threadIdx.x;
#if CRP_OT_VEC4 >= 2
    for (; t + 4 * BLOCK < body; t += 8 * BLOCK) {  // two loads in flight per lane
        const uint4 v = *reinterpret_cast<const uint4 *>(src + t);
        const uint4 w = *reinterpret_cast<const uint4 *>(src + t + 4 * BLOCK);
        f(t, v.x);
        f(t + 1, v.y);
        f(t + 2, v.z);
        f(t + 3, v.w);
        f(t + 4 * BLOCK, w.x);
        f(t + 4 * BLOCK + 1, w.y);
        f(t + 4 * BLOCK + 2, w.z);
        f(t + 4 * BLOCK + 3, w.w);
    }
#endif
    for (; t < body; t += 4 * BLOCK) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + t);
        f(t, v.x);
        f(t + 1, v.y);
        f(t + 2, v.z);
        f(t + 3, v.w);
    }
    for (uint64_t u = body + threadIdx.x; u < hi; u += BLOCK) f(u, src[u]);
#else
    for (uint64_t t = lo + threadIdx.x; t < hi; t += BLOCK) f(t, src[t]);
#endif
}

// (Staging each round's stretch of the planes through LDS -- coalesced copies, windows cut from LDS -- was measured
// slower than divergent loads: two extra barriers per round.  What helped is window12's single unaligned load.)
template <bool MINUS>
__global__ __launch_bounds__(BLOCK) void ot_seed_kernel(Planes pl, const uint32_t *__restrict__ pos, uint64_t n,
                                                         const uint64_t *__restrict__ own, uint32_t n_own,
                                                         uint32_t *__restrict__ seeds, uint32_t *__restrict__ bucket_total)
{
    __shared__ uint32_t cnt[OT_BUCKETS];
    for (uint32_t b = threadIdx.x; b < OT_BUCKETS; b += BLOCK) cnt[b] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * OT_CHUNK, hi = lo + OT_CHUNK < n ? lo + OT_CHUNK : n;
    for (uint64_t t = lo + threadIdx.x; t < hi; t += BLOCK) {
        const uint64_t p = pos[t];
        const uint64_t q = MINUS ? p + 3 : p - CRP_OT_SEED_LEN;
        const uint32_t code = seed_of_windows<MINUS>(window12(pl.plane[0], q), window12(pl.plane[1], q), window12(pl.plane[2], q),
                                                     window12(pl.plane[3], q), p, own, n_own);
        seeds[t] = code;
        if (code < OT_SEEDS) atomicAdd(&cnt[code >> 12], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < OT_BUCKETS; b += BLOCK)
        if (cnt[b]) atomicAdd(&bucket_total[b], cnt[b]);
}

// The same step when the scan itself delivered the seeds (crp_scan_score with CRP_SCAN_SEEDS, l = 20): the emit kernel
// wrote one RAW word per hit -- low code bits of the 12 seed characters in bits 11..0, high code bits in bits 23..12,
// already oriented, SEED_RAW_NONE where the 12 characters are not all bases -- because it has every hit's window in
// registers anyway.  This pass only interleaves the two halves (Morton code: character k at bits 2k, 2k+1), applies
// the ownership ranges and counts the buckets: 4 B in, 4 B out per hit (+ 4 B of position with ownership ranges),
// no second visit to the planes (ot_seed_kernel: four divergent loads per hit).
__global__ __launch_bounds__(BLOCK) void ot_seed_from_raw_kernel(const uint32_t *__restrict__ raw, const uint32_t *__restrict__ pos,
                                                                  uint64_t n, const uint64_t *__restrict__ own, uint32_t n_own,
                                                                  uint32_t *__restrict__ seeds, uint32_t *__restrict__ bucket_total)
{
    __shared__ uint32_t cnt[OT_BUCKETS];
    for (uint32_t b = threadIdx.x; b < OT_BUCKETS; b += BLOCK) cnt[b] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * OT_CHUNK, hi = lo + OT_CHUNK < n ? lo + OT_CHUNK : n;
    auto code_of = [&](uint64_t t, uint32_t r) -> uint32_t {
        uint32_t code = OT_NOT_A_SITE;
        if (r != SEED_RAW_NONE) {
            code = (spread12(r >> 12) << 1) | spread12(r & 0xfffu);
            if (n_own) {  // the range with the largest begin <= p
                const uint64_t p = pos[t];
                uint32_t a = 0, b = n_own;
                while (a < b) {
                    const uint32_t mid = (a + b) >> 1;
                    if (own[2 * mid] <= p) a = mid + 1;
                    else b = mid;
                }
                if (!(a > 0 && p < own[2 * (a - 1) + 1])) code = OT_NOT_OWNED;
            }
        }
        if (code < OT_SEEDS) atomicAdd(&cnt[code >> 12], 1u);
        return code;
    };
#if CRP_OT_VEC4
    const uint64_t body = lo + ((hi - lo) & ~(uint64_t)3);
    for (uint64_t t = lo + 4 * (uint64_t)threadIdx.x; t < body; t += 4 * BLOCK) {  // 16 bytes in, 16 bytes out per lane and trip
        const uint4 r = *reinterpret_cast<const uint4 *>(raw + t);
        uint4 c;
        c.x = code_of(t, r.x);
        c.y = code_of(t + 1, r.y);
        c.z = code_of(t + 2, r.z);
        c.w = code_of(t + 3, r.w);
        *reinterpret_cast<uint4 *>(seeds + t) = c;
    }
    for (uint64_t t = body + threadIdx.x; t < hi; t += BLOCK) seeds[t] = code_of(t, raw[t]);
#else
    for (uint64_t t = lo + threadIdx.x; t < hi; t += BLOCK) seeds[t] = code_of(t, raw[t]);
#endif
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < OT_BUCKETS; b += BLOCK)
        if (cnt[b]) atomicAdd(&bucket_total[b], cnt[b]);
}

// bucket_total[4096] -> cursor[4096] = exclusive prefix (the partition's write cursors), *n_sites = sum
__global__ __launch_bounds__(1024) void ot_bucket_scan_kernel(const uint32_t *__restrict__ bucket_total,
                                                               uint32_t *__restrict__ bucket_start, uint32_t *__restrict__ cursor,
                                                               uint32_t *__restrict__ cursor1,
                                                               unsigned long long *__restrict__ n_sites)
{
    __shared__ uint32_t part[1024];
    const int tid = threadIdx.x;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = bucket_total[4 * tid + k];
        sum += v[k];
    }
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele, 10 steps
        const uint32_t add = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    uint32_t ex = part[tid] - sum;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        bucket_start[4 * tid + k] = ex;
        cursor[4 * tid + k] = ex;
        if ((4 * tid + k) % 64 == 0) cursor1[(4 * tid + k) / 64] = ex;  // a super-bucket = 64 consecutive buckets
        ex += v[k];
    }
    if (tid == 1023) {
        bucket_start[OT_BUCKETS] = ex;
        *n_sites = ex;
    }
}

// Level 1: the sites of a chunk go to their SUPER-bucket (the seed's top 6 bits, 64 buckets of the final 4096) as
// whole 24-bit codes: a chunk of 16 384 hits writes runs of ~256 x 4 bytes, full lines -- where scattering straight
// into 4096 buckets wrote ~4 x 2 bytes per bucket and chunk (partial lines: 0.44 ms per strand on the bench genome).
__global__ __launch_bounds__(BLOCK) void ot_partition1_kernel(const uint32_t *__restrict__ seeds, uint64_t n,
                                                               uint32_t *__restrict__ cursor1, uint32_t *__restrict__ part1)
{
    __shared__ uint32_t cnt[64];
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * OT_CHUNK, hi = lo + OT_CHUNK < n ? lo + OT_CHUNK : n;
    for_chunk(seeds, lo, hi, [&](uint64_t, uint32_t code) {
        if (code < OT_SEEDS) atomicAdd(&cnt[code >> 18], 1u);
    });
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t c = cnt[threadIdx.x];
        cnt[threadIdx.x] = c ? atomicAdd(&cursor1[threadIdx.x], c) : 0;
    }
    __syncthreads();
    for_chunk(seeds, lo, hi, [&](uint64_t, uint32_t code) {
        if (code < OT_SEEDS) part1[atomicAdd(&cnt[code >> 18], 1u)] = code;
    });
}

// Level 2 (n_ptr: the number of entries, known on the device only): entries grouped by super-bucket -> their final
// bucket; a chunk now lies inside one or two super-buckets, so it writes runs of ~256 x 2 bytes.
__global__ __launch_bounds__(BLOCK) void ot_partition_kernel(const uint32_t *__restrict__ seeds, uint64_t n,
                                                              const unsigned long long *__restrict__ n_ptr,
                                                              uint32_t *__restrict__ cursor, uint16_t *__restrict__ part)
{
    __shared__ uint32_t cnt[OT_BUCKETS];  // first the chunk's count per bucket, then its next write index
    if (n_ptr) n = *n_ptr;
    if ((uint64_t)blockIdx.x * OT_CHUNK >= n) return;
    for (uint32_t b = threadIdx.x; b < OT_BUCKETS; b += BLOCK) cnt[b] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * OT_CHUNK, hi = lo + OT_CHUNK < n ? lo + OT_CHUNK : n;
    for_chunk(seeds, lo, hi, [&](uint64_t, uint32_t code) {
        if (code < OT_SEEDS) atomicAdd(&cnt[code >> 12], 1u);
    });
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < OT_BUCKETS; b += BLOCK) {
        const uint32_t c = cnt[b];
        cnt[b] = c ? atomicAdd(&cursor[b], c) : 0;  // this chunk's range in bucket b starts here
    }
    __syncthreads();
    for_chunk(seeds, lo, hi, [&](uint64_t, uint32_t code) {
        if (code < OT_SEEDS) part[atomicAdd(&cnt[code >> 12], 1u)] = (uint16_t)(code & 0xfffu);
    });
}

// ---- level 1 with the chunk STAGED: loaded once into registers (16-byte loads, all in flight), counted and ranked with LDS
// atomics out of the registers, sorted into LDS, and written out in slot order -- consecutive lanes write consecutive addresses
// of one run, so the scatter goes out in whole sectors instead of 4-byte pieces of 32-byte ones, and the codes are read from HBM
// once instead of twice: 0.120 -> 0.076 ms per strand.  (CRP_OT_STAGED=0: the unstaged kernels, for A/B builds.)
#ifndef CRP_OT_STAGED
#define CRP_OT_STAGED 1
#endif
constexpr uint32_t OT_STAGE = 8192;            // entries per workgroup: 32 KiB of LDS for the sorted chunk
constexpr int OT_STAGE_TRIPS = OT_STAGE / (4 * BLOCK);  // 16-byte loads per lane

// the chunk [lo, hi) of src into v[] (OT_NOT_A_SITE beyond hi); src + lo is 16-byte aligned
__device__ __forceinline__ void load_stage(const uint32_t *__restrict__ src, uint64_t lo, uint64_t hi, uint4 (&v)[OT_STAGE_TRIPS])
{
#pragma unroll
    for (int it = 0; it < OT_STAGE_TRIPS; ++it) {
        const uint64_t t = lo + 4 * ((uint64_t)it * BLOCK + threadIdx.x);
        if (t + 4 <= hi) {
            v[it] = *reinterpret_cast<const uint4 *>(src + t);
        } else {
            v[it].x = t < hi ? src[t] : OT_NOT_A_SITE;
            v[it].y = t + 1 < hi ? src[t + 1] : OT_NOT_A_SITE;
            v[it].z = t + 2 < hi ? src[t + 2] : OT_NOT_A_SITE;
            v[it].w = t + 3 < hi ? src[t + 3] : OT_NOT_A_SITE;
        }
    }
}

template <class F>
__device__ __forceinline__ void for_stage(const uint4 (&v)[OT_STAGE_TRIPS], F f)
{
#pragma unroll
    for (int it = 0; it < OT_STAGE_TRIPS; ++it) {
        f(v[it].x);
        f(v[it].y);
        f(v[it].z);
        f(v[it].w);
    }
}

// inclusive scan over the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

__global__ __launch_bounds__(BLOCK) void ot_partition1_staged_kernel(const uint32_t *__restrict__ seeds, uint64_t n,
                                                                      uint32_t *__restrict__ cursor1, uint32_t *__restrict__ part1)
{
    __shared__ uint32_t sorted[OT_STAGE];
    __shared__ uint32_t cnt[64], delta[64];
    __shared__ uint32_t n_valid;
    const uint64_t lo = (uint64_t)blockIdx.x * OT_STAGE, hi = lo + OT_STAGE < n ? lo + OT_STAGE : n;
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    uint4 v[OT_STAGE_TRIPS];
    load_stage(seeds, lo, hi, v);
    __syncthreads();
    for_stage(v, [&](uint32_t code) {
        if (code < OT_SEEDS) atomicAdd(&cnt[code >> 18], 1u);
    });
    __syncthreads();
    if (threadIdx.x < 64) {  // wave 0: where every digit's run starts in the chunk, and in the super-bucket it goes to
        const uint32_t c = cnt[threadIdx.x];
        const uint32_t incl = wave_inclusive_scan(c), ex = incl - c;
        const uint32_t g = c ? atomicAdd(&cursor1[threadIdx.x], c) : 0;
        delta[threadIdx.x] = g - ex;  // (mod 2^32)
        cnt[threadIdx.x] = ex;
        if (threadIdx.x == 63) n_valid = incl;
    }
    __syncthreads();
    for_stage(v, [&](uint32_t code) {
        if (code < OT_SEEDS) sorted[atomicAdd(&cnt[code >> 18], 1u)] = code;
    });
    __syncthreads();
    const uint32_t m = n_valid;
    for (uint32_t i = threadIdx.x; i < m; i += BLOCK) {
        const uint32_t code = sorted[i];
        part1[delta[code >> 18] + i] = code;
    }
}

// Level 2 with the chunk held in registers between its two passes: the codes are read from HBM once (0.103 -> 0.099 ms per
// strand).  Sorting it through LDS as level 1 does costs more than it saves here -- 4096 counters to scan and 64 KiB of LDS per
// workgroup: 0.126 ms (measured, profiles/EXPERIMENTS.md round 6).
constexpr int OT_CHUNK_TRIPS = OT_CHUNK / (4 * BLOCK);
__global__ __launch_bounds__(BLOCK) void ot_partition_reg_kernel(const uint32_t *__restrict__ seeds, uint64_t n,
                                                                  const unsigned long long *__restrict__ n_ptr,
                                                                  uint32_t *__restrict__ cursor, uint16_t *__restrict__ part)
{
    __shared__ uint32_t cnt[OT_BUCKETS];  // first the chunk's count per bucket, then its next write index
    if (n_ptr) n = *n_ptr;
    const uint64_t lo = (uint64_t)blockIdx.x * OT_CHUNK;
    if (lo >= n) return;
    const uint64_t hi = lo + OT_CHUNK < n ? lo + OT_CHUNK : n;
    for (uint32_t b = threadIdx.x; b < OT_BUCKETS; b += BLOCK) cnt[b] = 0;
    uint4 v[OT_CHUNK_TRIPS];
#pragma unroll
    for (int it = 0; it < OT_CHUNK_TRIPS; ++it) {
        const uint64_t t = lo + 4 * ((uint64_t)it * BLOCK + threadIdx.x);
        if (t + 4 <= hi) {
            v[it] = *reinterpret_cast<const uint4 *>(seeds + t);
        } else {
            v[it].x = t < hi ? seeds[t] : OT_NOT_A_SITE;
            v[it].y = t + 1 < hi ? seeds[t + 1] : OT_NOT_A_SITE;
            v[it].z = t + 2 < hi ? seeds[t + 2] : OT_NOT_A_SITE;
            v[it].w = t + 3 < hi ? seeds[t + 3] : OT_NOT_A_SITE;
        }
    }
    __syncthreads();
    auto each = [&](auto f) {
#pragma unroll
        for (int it = 0; it < OT_CHUNK_TRIPS; ++it) {
            f(v[it].x);
            f(v[it].y);
            f(v[it].z);
            f(v[it].w);
        }
    };
    each([&](uint32_t code) {
        if (code < OT_SEEDS) atomicAdd(&cnt[code >> 12], 1u);
    });
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < OT_BUCKETS; b += BLOCK) {
        const uint32_t c = cnt[b];
        cnt[b] = c ? atomicAdd(&cursor[b], c) : 0;  // this chunk's range in bucket b starts here
    }
    __syncthreads();
    each([&](uint32_t code) {
        if (code < OT_SEEDS) part[atomicAdd(&cnt[code >> 12], 1u)] = (uint16_t)(code & 0xfffu);
    });
}

__global__ __launch_bounds__(1024) void ot_bucket_hist_kernel(const uint16_t *__restrict__ part,
                                                               const uint32_t *__restrict__ bucket_start,
                                                               uint32_t *__restrict__ hist)
{
    __shared__ uint32_t bins[4096];
    const uint32_t b = blockIdx.x;
    const uint32_t lo = bucket_start[b], hi = bucket_start[b + 1];
    if (lo == hi) return;  // nothing to add to this slice
    for (uint32_t k = threadIdx.x; k < 4096; k += 1024) bins[k] = 0;
    __syncthreads();
#if CRP_OT_VEC4
    {   // eight 16-bit entries per lane and trip on the 16-byte-aligned body, the ragged ends one by one
        const uint32_t a0 = min(hi, (lo + 7u) & ~7u), a1 = a0 + ((hi - a0) & ~7u);
        for (uint32_t t = lo + threadIdx.x; t < a0; t += 1024) atomicAdd(&bins[part[t]], 1u);
        for (uint32_t t = a0 + 8 * threadIdx.x; t < a1; t += 8 * 1024) {
            const uint4 v = *reinterpret_cast<const uint4 *>(part + t);
            atomicAdd(&bins[v.x & 0xffffu], 1u);
            atomicAdd(&bins[v.x >> 16], 1u);
            atomicAdd(&bins[v.y & 0xffffu], 1u);
            atomicAdd(&bins[v.y >> 16], 1u);
            atomicAdd(&bins[v.z & 0xffffu], 1u);
            atomicAdd(&bins[v.z >> 16], 1u);
            atomicAdd(&bins[v.w & 0xffffu], 1u);
            atomicAdd(&bins[v.w >> 16], 1u);
        }
        for (uint32_t t = a1 + threadIdx.x; t < hi; t += 1024) atomicAdd(&bins[part[t]], 1u);
    }
#else
    for (uint32_t t = lo + threadIdx.x; t < hi; t += 1024) atomicAdd(&bins[part[t]], 1u);
#endif
    __syncthreads();
    uint4 *slice = reinterpret_cast<uint4 *>(hist + (size_t)b * 4096);
    uint4 v = slice[threadIdx.x];
    const uint4 a = reinterpret_cast<const uint4 *>(bins)[threadIdx.x];
    v.x += a.x;
    v.y += a.y;
    v.z += a.z;
    v.w += a.w;
    slice[threadIdx.x] = v;
}

// One pass of the ball recurrence over the four seed positions whose index bits are
// [FIELD_SHIFT, FIELD_SHIFT + 8).  A tile = all 256 values of that field x 16 consecutive values of
// the bits below it (FIRST: field at bit 0, a tile is 4096 consecutive seeds).  LDS holds the tile
// as f[d][m * 16 + r] (m: field value, r: the 16 neighbours; FIRST: f[d][seed & 4095]).
template <int FIELD_SHIFT, bool FIRST>
__global__ __launch_bounds__(1024) void ot_ball_kernel(const uint32_t *__restrict__ hist, uint4 *__restrict__ ball)
{
    constexpr int TILE = 4096;
    __shared__ uint32_t f[4][TILE];
    const int tid = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    // local bit layout: FIRST: the field is local bits 0..7; otherwise local bits 4..11 (r below it)
    constexpr int LOCAL_FIELD = FIRST ? 0 : 4;
    uint32_t gbase = 0;
    if constexpr (FIRST) {
        gbase = tile * TILE;
        // 4096 counts = 1024 x 16 bytes
        const uint4 v = reinterpret_cast<const uint4 *>(hist + gbase)[tid];
        reinterpret_cast<uint4 *>(f[0])[tid] = v;
#pragma unroll
        for (int d = 1; d < 4; ++d) reinterpret_cast<uint4 *>(f[d])[tid] = make_uint4(0, 0, 0, 0);
    } else {
        constexpr uint32_t N_LO = 1u << (FIELD_SHIFT - 4);  // tiles that share the bits above the field
        const uint32_t t_lo = tile % N_LO, t_hi = tile / N_LO;
        gbase = (t_hi << (FIELD_SHIFT + 8)) | (t_lo << 4);
#pragma unroll
        for (int it = 0; it < TILE / 1024; ++it) {
            const int e = tid + it * 1024;  // e = m * 16 + r: 16 lanes read 256 contiguous bytes
            const uint4 v = ball[gbase | ((uint32_t)(e >> 4) << FIELD_SHIFT) | (uint32_t)(e & 15)];
            f[0][e] = v.x;
            f[1][e] = v.y;
            f[2][e] = v.z;
            f[3][e] = v.w;
        }
    }
    __syncthreads();
    // four steps, one seed position each; 1024 groups of 4 entries per step = one per thread
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int low = LOCAL_FIELD + 2 * p;  // the group's members differ in local bits low, low + 1
        const uint32_t g = (uint32_t)tid;
        const uint32_t base = ((g >> low) << (low + 2)) | (g & ((1u << low) - 1u));
        uint32_t v[4][4], s[4] = {0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                v[b][d] = f[d][base + ((uint32_t)b << low)];
                s[d] += v[b][d];
            }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int d = 1; d < 4; ++d) f[d][base + ((uint32_t)b << low)] = v[b][d] + s[d - 1] - v[b][d - 1];
        __syncthreads();
    }
    if constexpr (FIRST) {
#pragma unroll
        for (int it = 0; it < TILE / 1024; ++it) {
            const int e = tid + it * 1024;
            ball[gbase + e] = make_uint4(f[0][e], f[1][e], f[2][e], f[3][e]);
        }
    } else {
#pragma unroll
        for (int it = 0; it < TILE / 1024; ++it) {
            const int e = tid + it * 1024;
            ball[gbase | ((uint32_t)(e >> 4) << FIELD_SHIFT) | (uint32_t)(e & 15)] = make_uint4(f[0][e], f[1][e], f[2][e], f[3][e]);
        }
    }
}

// ILP independent gathers in flight per lane.  Measured: 1, 2, 4, 8 all run at the same rate,
// and a table cut down to 4 MB (L2-resident) is only 2x faster -- the gather is bound by the ~6 cycles the vector
// memory path spends per divergent lane, neither by latency nor by HBM bandwidth.  (Non-temporal loads: 1.44 ms
// instead of 1.12 -- the caches do serve part of the gather.)
__global__ __launch_bounds__(BLOCK) void ot_lookup_kernel(const uint32_t *__restrict__ seeds, uint64_t n,
                                                           const uint4 *__restrict__ ball, uint4 *__restrict__ out)
{
    constexpr int ILP = 1;
    const uint64_t span = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t t0 = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; t0 < n; t0 += span * ILP) {
        uint32_t s[ILP];
        uint4 v[ILP];
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            const uint64_t t = t0 + k * span;
            s[k] = t < n ? seeds[t] : OT_NOT_A_SITE;
        }
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            v[k] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
            if (s[k] < OT_SEEDS) {
                v[k] = ball[s[k]];
                v[k].x -= 1;  // the hit itself is one of the sites at distance 0
            }
        }
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            const uint64_t t = t0 + k * span;
            if (t < n) out[t] = v[k];
        }
    }
}

static uint32_t grid_for(uint64_t n)
{
    const uint64_t b = (n + BLOCK - 1) / BLOCK;
    return (uint32_t)std::min<uint64_t>(b, 16384);
}

}  // namespace crp

extern "C" {

int crp_offtarget_reset(crp_ctx *ctx)
{
    if (!ctx) return CRP_ERR_INVALID;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_ot_hist) CRP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_ot_hist), (size_t)crp::OT_SEEDS * sizeof(uint32_t)));
    if (!ctx->d_ot_ball) CRP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_ot_ball), (size_t)crp::OT_SEEDS * sizeof(uint4)));
    CRP_HIP(ctx, hipMemsetAsync(ctx->d_ot_hist, 0, (size_t)crp::OT_SEEDS * sizeof(uint32_t), ctx->stream));
    ctx->ot_solved = false;
    ctx->ot_epoch += 1;
    return CRP_OK;
}

int crp_offtarget_add(crp_arena *a, int guide_len, const uint64_t *own_ranges, uint64_t n_ranges, uint64_t *n_sites)
{
    crp::Range roctx_range("crp: off-target seeds + histogram");
    if (!a || (n_ranges && !own_ranges) || n_ranges > 0x7fffffffu) return CRP_ERR_INVALID;
    if (guide_len < 1 || guide_len > 50) return CRP_ERR_UNSUPPORTED;
    crp_ctx *ctx = a->ctx;
    if (!a->have_hits || !ctx->d_ot_hist || ctx->ot_solved) return CRP_ERR_STATE;
    if (a->ot_epoch == ctx->ot_epoch) return CRP_ERR_STATE;  // already added since the last reset
    for (uint64_t r = 0; r < n_ranges; ++r) {
        if (own_ranges[2 * r] > own_ranges[2 * r + 1]) return CRP_ERR_INVALID;
        if (r && own_ranges[2 * r] < own_ranges[2 * r - 1]) return CRP_ERR_INVALID;  // ascending, disjoint
    }
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    for (int s = 0; s < 2; ++s) {
        uint64_t cap_seed = a->ot_cap[s], cap_cnt = a->ot_cap[s];
        int rc = crp::grow(ctx, reinterpret_cast<void **>(&a->d_ot_seed[s]), &cap_seed, a->n_hits[s], sizeof(uint32_t));
        if (rc == CRP_OK) rc = crp::grow(ctx, reinterpret_cast<void **>(&a->d_ot_cnt[s]), &cap_cnt, a->n_hits[s], sizeof(uint4));
        if (rc != CRP_OK) {
            a->ot_cap[s] = 0;
            return rc;
        }
        a->ot_cap[s] = std::min(cap_seed, cap_cnt);
    }
    if (n_ranges) {
        int rc = crp::grow(ctx, reinterpret_cast<void **>(&a->d_ot_own), &a->ot_own_cap, 2 * n_ranges, sizeof(uint64_t));
        if (rc != CRP_OK) return rc;
        CRP_HIP(ctx, hipMemcpyAsync(a->d_ot_own, own_ranges, 2 * n_ranges * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    }
    // scratch of the partition, for one strand's table at a time: every site's whole code (level 1) and its low
    // 12 bits (level 2); bucket totals | bucket starts (+ end) | level-2 cursors | 64 level-1 cursors
    const uint64_t n_max = std::max(a->n_hits[0], a->n_hits[1]);
    int rc0 = crp::grow(ctx, reinterpret_cast<void **>(&ctx->d_ot_part), &ctx->ot_part_cap, n_max, sizeof(uint16_t));
    if (rc0 == CRP_OK) rc0 = crp::grow(ctx, reinterpret_cast<void **>(&ctx->d_ot_part1), &ctx->ot_part1_cap, n_max, sizeof(uint32_t));
    if (rc0 != CRP_OK) return rc0;
    if (!ctx->d_ot_bucket) CRP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_ot_bucket), (3 * crp::OT_BUCKETS + 8 + 64) * sizeof(uint32_t)));
    uint32_t *d_total = ctx->d_ot_bucket, *d_start = d_total + crp::OT_BUCKETS, *d_cursor = d_start + crp::OT_BUCKETS + 4;
    uint32_t *d_cursor1 = d_cursor + crp::OT_BUCKETS + 4;
    crp::Planes pl{{a->d_plane[0], a->d_plane[1], a->d_plane[2], a->d_plane[3]}};
    unsigned long long *d_n = reinterpret_cast<unsigned long long *>(ctx->d_scalar);
    uint64_t sites = 0;
    crp::prof_begin(ctx, CRP_K_OT_SEED);
    if (guide_len < CRP_OT_SEED_LEN) {
        // `sequence` has fewer than 12 characters: no hit is a site
        for (int s = 0; s < 2; ++s)
            if (a->n_hits[s]) CRP_HIP(ctx, hipMemsetAsync(a->d_ot_seed[s], 0xff, a->n_hits[s] * sizeof(uint32_t), ctx->stream));
    } else {
        for (int s = 0; s < 2; ++s) {  // one strand's table at a time (they share the partition scratch)
            const uint64_t n = a->n_hits[s];
            if (!n) continue;
            const uint32_t chunks = (uint32_t)((n + crp::OT_CHUNK - 1) / crp::OT_CHUNK);
            CRP_HIP(ctx, hipMemsetAsync(d_total, 0, crp::OT_BUCKETS * sizeof(uint32_t), ctx->stream));
            if (a->have_raw && guide_len == 20)  // the scan wrote the seed words (CRP_SCAN_SEEDS)
                hipLaunchKernelGGL(crp::ot_seed_from_raw_kernel, dim3(chunks), dim3(crp::BLOCK), 0, ctx->stream, a->d_ot_raw[s],
                                   a->d_pos[s], n, a->d_ot_own, (uint32_t)n_ranges, a->d_ot_seed[s], d_total);
            else if (s == 0)
                hipLaunchKernelGGL(crp::ot_seed_kernel<false>, dim3(chunks), dim3(crp::BLOCK), 0, ctx->stream, pl, a->d_pos[0], n,
                                   a->d_ot_own, (uint32_t)n_ranges, a->d_ot_seed[0], d_total);
            else
                hipLaunchKernelGGL(crp::ot_seed_kernel<true>, dim3(chunks), dim3(crp::BLOCK), 0, ctx->stream, pl, a->d_pos[1], n,
                                   a->d_ot_own, (uint32_t)n_ranges, a->d_ot_seed[1], d_total);
            hipLaunchKernelGGL(crp::ot_bucket_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_total, d_start, d_cursor, d_cursor1,
                               d_n + s);
#if CRP_OT_STAGED
            const uint32_t stages = (uint32_t)((n + crp::OT_STAGE - 1) / crp::OT_STAGE);
            hipLaunchKernelGGL(crp::ot_partition1_staged_kernel, dim3(stages), dim3(crp::BLOCK), 0, ctx->stream, a->d_ot_seed[s], n, d_cursor1,
                               ctx->d_ot_part1);
            hipLaunchKernelGGL(crp::ot_partition_reg_kernel, dim3(chunks), dim3(crp::BLOCK), 0, ctx->stream, ctx->d_ot_part1, n,
                               reinterpret_cast<const unsigned long long *>(d_n + s), d_cursor, ctx->d_ot_part);
#else
            hipLaunchKernelGGL(crp::ot_partition1_kernel, dim3(chunks), dim3(crp::BLOCK), 0, ctx->stream, a->d_ot_seed[s], n, d_cursor1,
                               ctx->d_ot_part1);
            hipLaunchKernelGGL(crp::ot_partition_kernel, dim3(chunks), dim3(crp::BLOCK), 0, ctx->stream, ctx->d_ot_part1, n,
                               reinterpret_cast<const unsigned long long *>(d_n + s), d_cursor, ctx->d_ot_part);
#endif
            hipLaunchKernelGGL(crp::ot_bucket_hist_kernel, dim3(crp::OT_BUCKETS), dim3(1024), 0, ctx->stream, ctx->d_ot_part, d_start,
                               ctx->d_ot_hist);
            CRP_HIP(ctx, hipGetLastError());
        }
    }
    crp::prof_end(ctx, CRP_K_OT_SEED);
    CRP_HIP(ctx, hipMemcpyAsync(ctx->h_scalar, ctx->d_scalar, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // own_ranges may be freed by the caller; the site counts are read
    crp::prof_collect(ctx, CRP_K_OT_SEED);
    if (guide_len >= CRP_OT_SEED_LEN) sites = (a->n_hits[0] ? ctx->h_scalar[0] : 0) + (a->n_hits[1] ? ctx->h_scalar[1] : 0);
    if (n_sites) *n_sites = sites;
    a->ot_epoch = ctx->ot_epoch;
    return CRP_OK;
}

int crp_offtarget_reduce(crp_ctx *ctx)
{
    crp::Range roctx_range("crp: off-target all-reduce (RCCL)");
    if (!ctx) return CRP_ERR_INVALID;
    if (!ctx->d_ot_hist || ctx->ot_solved) return CRP_ERR_STATE;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    crp::prof_begin(ctx, CRP_K_OT_REDUCE);
    const int rc = crp::comm_allreduce_u32(ctx, ctx->d_ot_hist, crp::OT_SEEDS);
    crp::prof_end(ctx, CRP_K_OT_REDUCE);
    if (rc != CRP_OK) return rc;
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    crp::prof_collect(ctx, CRP_K_OT_REDUCE);
    return CRP_OK;
}

int crp_offtarget_hist_get(crp_ctx *ctx, uint32_t *hist)
{
    if (!ctx || !hist) return CRP_ERR_INVALID;
    if (!ctx->d_ot_hist) return CRP_ERR_STATE;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    {
        const int rc = crp::staged_d2h(ctx, hist, ctx->d_ot_hist, (size_t)crp::OT_SEEDS * sizeof(uint32_t));
        if (rc != CRP_OK) return rc;
    }
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CRP_OK;
}

int crp_offtarget_hist_set(crp_ctx *ctx, const uint32_t *hist)
{
    if (!ctx || !hist) return CRP_ERR_INVALID;
    if (!ctx->d_ot_hist || ctx->ot_solved) return CRP_ERR_STATE;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    {
        const int rc = crp::staged_h2d(ctx, ctx->d_ot_hist, hist, (size_t)crp::OT_SEEDS * sizeof(uint32_t));
        if (rc != CRP_OK) return rc;
    }
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CRP_OK;
}

int crp_offtarget_solve(crp_ctx *ctx)
{
    crp::Range roctx_range("crp: off-target ball sums");
    if (!ctx) return CRP_ERR_INVALID;
    if (!ctx->d_ot_hist) return CRP_ERR_STATE;
    if (ctx->ot_solved) return CRP_OK;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    constexpr uint32_t TILES = crp::OT_SEEDS / 4096;
    crp::prof_begin(ctx, CRP_K_OT_BALL);
    hipLaunchKernelGGL((crp::ot_ball_kernel<0, true>), dim3(TILES), dim3(1024), 0, ctx->stream, ctx->d_ot_hist, ctx->d_ot_ball);
    hipLaunchKernelGGL((crp::ot_ball_kernel<8, false>), dim3(TILES), dim3(1024), 0, ctx->stream, ctx->d_ot_hist, ctx->d_ot_ball);
    hipLaunchKernelGGL((crp::ot_ball_kernel<16, false>), dim3(TILES), dim3(1024), 0, ctx->stream, ctx->d_ot_hist, ctx->d_ot_ball);
    CRP_HIP(ctx, hipGetLastError());
    crp::prof_end(ctx, CRP_K_OT_BALL);
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    crp::prof_collect(ctx, CRP_K_OT_BALL);
    ctx->ot_solved = true;
    return CRP_OK;
}

int crp_offtarget_counts(crp_arena *a, uint32_t *counts_plus, uint32_t *counts_minus)
{
    crp::Range roctx_range("crp: off-target look-up");
    if (!a) return CRP_ERR_INVALID;
    crp_ctx *ctx = a->ctx;
    if (!ctx->ot_solved || !a->have_hits || a->ot_epoch != ctx->ot_epoch) return CRP_ERR_STATE;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    crp::prof_begin(ctx, CRP_K_OT_LOOKUP);
    for (int s = 0; s < 2; ++s)
        if (a->n_hits[s])
            hipLaunchKernelGGL(crp::ot_lookup_kernel, dim3(crp::grid_for(a->n_hits[s])), dim3(crp::BLOCK), 0, ctx->stream,
                               a->d_ot_seed[s], a->n_hits[s], ctx->d_ot_ball, a->d_ot_cnt[s]);
    CRP_HIP(ctx, hipGetLastError());
    crp::prof_end(ctx, CRP_K_OT_LOOKUP);
    uint32_t *host[2] = {counts_plus, counts_minus};
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    crp::prof_collect(ctx, CRP_K_OT_LOOKUP);
    for (int s = 0; s < 2; ++s)
        if (host[s] && a->n_hits[s]) {
            const int rc = crp::staged_d2h(ctx, host[s], a->d_ot_cnt[s], a->n_hits[s] * sizeof(uint4));
            if (rc != CRP_OK) return rc;
        }
    return CRP_OK;
}

int crp_offtarget_seeds(crp_arena *a, uint32_t *seeds_plus, uint32_t *seeds_minus)
{
    if (!a) return CRP_ERR_INVALID;
    crp_ctx *ctx = a->ctx;
    if (!a->have_hits || a->ot_epoch != ctx->ot_epoch || a->ot_epoch == 0) return CRP_ERR_STATE;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t *host[2] = {seeds_plus, seeds_minus};
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int s = 0; s < 2; ++s)
        if (host[s] && a->n_hits[s]) {
            const int rc = crp::staged_d2h(ctx, host[s], a->d_ot_seed[s], a->n_hits[s] * sizeof(uint32_t));
            if (rc != CRP_OK) return rc;
        }
    return CRP_OK;
}

}  // extern "C"
