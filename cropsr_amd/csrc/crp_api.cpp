// crp_api.cpp -- C ABI of libcropsr_hip.so (see include/cropsr_hip.h).
//
// Host side of the MI355X PAM-scan/score engine: owns the HIP stream, the
// device-resident arena (four bit-planes in HBM), the hit tables and the scan
// launches (one chained-scan kernel by default, count -> tile scan -> emit with
// CRP_OPT_TWO_PASS).  No CPU compute path exists here except crp_pack_ascii
// (host packing, the alternative to the on-GPU pack kernel).
#include "cropsr_hip.h"

#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "crp_internal.h"
#include "crp_roctx.h"

namespace {

// arena positions (and therefore per-strand hit counts) stay below 2^31: the chained-scan
// descriptors carry two 31-bit counts in one 64-bit word
constexpr uint64_t kMaxArenaWords = (1ull << 31) / 64 - 2 * crp::ARENA_ALIGN_WORDS;
constexpr uint64_t kUploadChunk = crp::STAGE_CHUNK;  // characters per H2D + pack round (multiple of 4096) = one pinned buffer

inline uint64_t round_up(uint64_t x, uint64_t m) { return (x + m - 1) / m * m; }

}  // namespace

namespace crp {

// ---- host-side copies between the caller's pages and the pinned staging buffers
// (Round 6 kept the copy threads in a process-wide pool for a while -- no thread start per 32 MiB chunk.  It bought nothing
// measurable: the staging copy reads the caller's pages at 41-43 GB/s whatever the thread count and is not what an upload
// waits for.  Threads per copy it is; profiles/EXPERIMENTS.md round 6.)
// memcpy spread over a few threads.  (Asking the kernel for the destination's pages up front -- MADV_POPULATE_WRITE -- was
// measured and dropped: numpy's large arrays are already advised for huge pages and the plain copy was faster there,
// 0.044 s against 0.055 s for the bench workload's tables; profiles/microbench/pcie_copy.hip.)
void parallel_copy(void *dst, const void *src, size_t n, int threads)
{
    const CopySeg seg{dst, src, n};
    parallel_copy_multi(&seg, 1, threads);
}

// several copies at once, the bytes of all of them dealt evenly to the threads (a slice's four table columns are 5-20 MB
// each: one after the other they would never occupy more than a few threads)
void parallel_copy_multi(const CopySeg *segs, int n_segs, int threads)
{
    size_t total = 0;
    for (int i = 0; i < n_segs; ++i) total += segs[i].bytes;
    if (!total) return;
    // (4 MiB per thread at least: a 32 MiB staging round still takes eight threads, a small genome's tables take one)
    constexpr size_t MIN_PER_THREAD = 4ull << 20;
    const int tt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, std::min(threads, 64)), total / MIN_PER_THREAD));
    const size_t per_t = ((total + tt - 1) / tt + 4095) & ~(size_t)4095;
    auto work = [=](int k) {  // piece k: the bytes [k * per_t, (k + 1) * per_t) of the concatenation
        size_t lo = per_t * (size_t)k, hi = std::min(total, lo + per_t), base = 0;
        for (int i = 0; i < n_segs && lo < hi; ++i) {
            const size_t end = base + segs[i].bytes;
            if (lo < end) {
                const size_t a = lo - base, b = std::min(hi, end) - base;
                std::memcpy(static_cast<uint8_t *>(segs[i].dst) + a, static_cast<const uint8_t *>(segs[i].src) + a, b - a);
                lo = base + b;
            }
            base = end;
        }
    };
    if (tt <= 1) {
        work(0);
        return;
    }
    // (nothing may throw across the C ABI: no allocation here, and a thread that cannot be started is replaced by doing its
    // piece on this one)
    constexpr int MAX_THREADS = 64;
    std::thread pool[MAX_THREADS];
    bool started[MAX_THREADS] = {};
    for (int k = 1; k < tt; ++k) {
        try {
            pool[k] = std::thread(work, k);
            started[k] = true;
        } catch (...) {
        }
    }
    work(0);
    for (int k = 1; k < tt; ++k) {
        if (started[k]) pool[k].join();
        else work(k);
    }
}

int staging_ready(crp_ctx *ctx)
{
    if (ctx->pin[0]) return CRP_OK;
    for (int b = 0; b < crp_ctx::N_PIN; ++b) {
        CRP_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->pin[b]), STAGE_CHUNK, hipHostMallocDefault));
        CRP_HIP(ctx, hipEventCreateWithFlags(&ctx->pin_done[b], hipEventDisableTiming));
        ctx->pin_busy[b] = false;
    }
    return CRP_OK;
}

static int pin_wait(crp_ctx *ctx, int b)
{
    if (ctx->pin_busy[b]) {
        CRP_HIP(ctx, hipEventSynchronize(ctx->pin_done[b]));
        ctx->pin_busy[b] = false;
    }
    return CRP_OK;
}

// host memory the GPU can reach by DMA (crp_host_alloc, hipHostMalloc, hipHostRegister): no staging copy needed
bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t attr;
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // (pageable memory: "invalid value", not an error of the caller's)
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

int staged_h2d(crp_ctx *ctx, void *d_dst, const void *src, size_t n)
{
    if (n >= STAGE_CHUNK / 4 && is_pinned_host(src)) {  // straight off the caller's pinned pages; the caller keeps them until the stream has passed
        CRP_HIP(ctx, hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, ctx->stream));
        CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return CRP_OK;
    }
    if (n < (64u << 10)) {  // tiny: the runtime's own path (it copies such a source into its staging area and returns)
        if (n) CRP_HIP(ctx, hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, ctx->stream));
        return CRP_OK;
    }
    // anything larger goes through OUR pinned buffers, never through the runtime's path for pageable memory: that one
    // registers the caller's pages with the driver, and freeing such memory later costs the process a quiesce and restore of
    // its GPU queues (staged_d2h below; profiles/EXPERIMENTS.md round 6)
    int rc = staging_ready(ctx);
    if (rc != CRP_OK) return rc;
    int b = 0;
    for (size_t off = 0; off < n; off += STAGE_CHUNK, b ^= 1) {
        const size_t len = std::min(STAGE_CHUNK, n - off);
        rc = pin_wait(ctx, b);  // the copy that last read this buffer has left
        if (rc != CRP_OK) return rc;
        parallel_copy(ctx->pin[b], static_cast<const uint8_t *>(src) + off, len, ctx->copy_threads);
        CRP_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t *>(d_dst) + off, ctx->pin[b], len, hipMemcpyHostToDevice, ctx->stream));
        CRP_HIP(ctx, hipEventRecord(ctx->pin_done[b], ctx->stream));
        ctx->pin_busy[b] = true;
    }
    return CRP_OK;
}

int staged_d2h(crp_ctx *ctx, void *dst, const void *d_src, size_t n)
{
    if (!n) return CRP_OK;
    if (is_pinned_host(dst)) {  // DMA straight into the caller's pinned pages
        CRP_HIP(ctx, hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, ctx->stream));
        CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return CRP_OK;
    }
    int rc = staging_ready(ctx);
    if (rc != CRP_OK) return rc;
    for (int b = 0; b < 2; ++b) {
        rc = pin_wait(ctx, b);
        if (rc != CRP_OK) return rc;
    }
    if (n <= STAGE_CHUNK) {
        // Small: one bounce through a staging buffer.  NEVER the runtime's own path for pageable memory: it registers the
        // caller's pages with the driver for the copy, and when the caller later frees that memory (numpy arrays of a few MB)
        // the driver quiesces and restores the process's GPU queues -- the next launch completed 17 ms late, every time
        // (bench.py --workload ecoli, profiles/EXPERIMENTS.md round 6).
        CRP_HIP(ctx, hipMemcpyAsync(ctx->pin[0], d_src, n, hipMemcpyDeviceToHost, ctx->stream));
        CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        parallel_copy(dst, ctx->pin[0], n, ctx->copy_threads);
        return CRP_OK;
    }
    // chunk k is copied out of its pinned buffer by the host threads (which also take the first-touch page faults of a
    // fresh destination, in parallel) while chunk k + 1 crosses the link into the other one
    size_t prev_off = 0, prev_len = 0;
    int b = 0, prev_b = 0;
    for (size_t off = 0; off < n; off += STAGE_CHUNK, b ^= 1) {
        const size_t len = std::min(STAGE_CHUNK, n - off);
        CRP_HIP(ctx, hipMemcpyAsync(ctx->pin[b], static_cast<const uint8_t *>(d_src) + off, len, hipMemcpyDeviceToHost, ctx->stream));
        CRP_HIP(ctx, hipEventRecord(ctx->pin_done[b], ctx->stream));
        if (prev_len) {
            CRP_HIP(ctx, hipEventSynchronize(ctx->pin_done[prev_b]));
            parallel_copy(static_cast<uint8_t *>(dst) + prev_off, ctx->pin[prev_b], prev_len, ctx->copy_threads);
        }
        prev_off = off;
        prev_len = len;
        prev_b = b;
    }
    if (prev_len) {
        CRP_HIP(ctx, hipEventSynchronize(ctx->pin_done[prev_b]));
        parallel_copy(static_cast<uint8_t *>(dst) + prev_off, ctx->pin[prev_b], prev_len, ctx->copy_threads);
    }
    return CRP_OK;
}

int grow(crp_ctx *ctx, void **p, uint64_t *cap, uint64_t need, size_t elem)
{
    if (*cap >= need && *p) return CRP_OK;
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const uint64_t n = std::max<uint64_t>(need + need / 16, 64);
    CRP_HIP(ctx, hipMalloc(p, n * elem));
    *cap = n;
    return CRP_OK;
}
}  // namespace crp
extern "C" {

int crp_abi_version(void) { return CRP_ABI_VERSION; }

const char *crp_strerror(int status)
{
    switch (status) {
        case CRP_OK: return "ok";
        case CRP_ERR_INVALID: return "invalid argument";
        case CRP_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
        case CRP_ERR_HIP: return "HIP runtime error";
        case CRP_ERR_NOMEM: return "out of memory";
        case CRP_ERR_STATE: return "call out of order";
        case CRP_ERR_CAPACITY: return "arena capacity exceeded";
        case CRP_ERR_UNSUPPORTED: return "unsupported parameter (guide length outside 0..50)";
        case CRP_ERR_IO: return "write to the output descriptor failed";
        case CRP_ERR_COMM: return "RCCL error";
        case CRP_ERR_PEER: return "abandoned on every rank: another rank reported an error before the exchange";
        default: return "unknown status";
    }
}

int crp_init(int device_id, crp_ctx **out)
{
    if (!out || device_id < 0) return CRP_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id >= n) return CRP_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return CRP_ERR_NO_DEVICE;
    crp_ctx *ctx = new (std::nothrow) crp_ctx();
    if (!ctx) return CRP_ERR_NOMEM;
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        delete ctx;  // the library carries gfx950 code objects only
        return CRP_ERR_NO_DEVICE;
    }
    {
        std::snprintf(ctx->name, sizeof ctx->name, "%s (%s)", prop.name, prop.gcnArchName);
        ctx->n_cu = prop.multiProcessorCount;
        ctx->hbm = prop.totalGlobalMem;
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return CRP_ERR_HIP;
    }
    for (auto &e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            crp_destroy(ctx);
            return CRP_ERR_HIP;
        }
    if (hipMalloc(reinterpret_cast<void **>(&ctx->d_scalar), 8 * sizeof(uint64_t)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&ctx->h_scalar), 8 * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) {
        crp_destroy(ctx);
        return CRP_ERR_NOMEM;
    }
    if (const char *e = std::getenv("CRP_TEST_MUTE_TILE")) ctx->mute_tile = (uint32_t)std::strtoul(e, nullptr, 10);
    {
        const unsigned hw = std::thread::hardware_concurrency();
        ctx->copy_threads = (int)std::max(1u, std::min(8u, hw ? hw / 2 : 4u));
        if (const char *e = std::getenv("CRP_COPY_THREADS")) ctx->copy_threads = std::max(1, std::min(64, std::atoi(e)));
    }
    *out = ctx;
    return CRP_OK;
}

int crp_destroy(crp_ctx *ctx)
{
    if (!ctx) return CRP_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    (void)hipFree(ctx->d_text);
    (void)hipFree(ctx->d_rows);
    (void)hipFree(ctx->d_rpre);
    (void)hipFree(ctx->d_rscore);
    (void)hipFree(ctx->d_scalar);
    if (ctx->h_scalar) (void)hipHostFree(ctx->h_scalar);
    for (int b = 0; b < crp_ctx::N_PIN; ++b) {
        if (ctx->pin[b]) (void)hipHostFree(ctx->pin[b]);
        if (ctx->pin_done[b]) (void)hipEventDestroy(ctx->pin_done[b]);
    }
    (void)hipFree(ctx->d_ot_hist);
    (void)hipFree(ctx->d_ot_ball);
    (void)hipFree(ctx->d_ot_part);
    (void)hipFree(ctx->d_ot_part1);
    (void)hipFree(ctx->d_ot_bucket);
    crp::comm_release(ctx);
    crp::stream_release(ctx);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CRP_OK;
}

const char *crp_last_error(const crp_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int crp_device_info(const crp_ctx *ctx, char *name, int name_cap, int *n_cu, uint64_t *hbm_bytes)
{
    if (!ctx) return CRP_ERR_INVALID;
    if (name && name_cap > 0) {
        std::strncpy(name, ctx->name, (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (n_cu) *n_cu = ctx->n_cu;
    if (hbm_bytes) *hbm_bytes = ctx->hbm;
    return CRP_OK;
}

// ---------------------------------------------------------------- host pack
uint64_t crp_arena_words_for(uint64_t len) { return (len + 63) / 64 + 1; }
uint64_t crp_arena_words_total(uint64_t sum) { return sum + 1; }
uint64_t crp_arena_max_words(void) { return kMaxArenaWords; }

static void pack_range(const uint8_t *text, uint64_t len, uint64_t w0, uint64_t w1, const uint8_t *lut,
                       uint64_t *hi, uint64_t *lo, uint64_t *up, uint64_t *ac)
{
    for (uint64_t w = w0; w < w1; ++w) {
        uint64_t b[4] = {0, 0, 0, 0};
        const uint64_t base = w * 64;
        const uint64_t n = std::min<uint64_t>(64, len - base);
        for (uint64_t k = 0; k < n; ++k) {
            const uint64_t nib = lut[text[base + k]];
            b[0] |= (nib & 1) << k;
            b[1] |= ((nib >> 1) & 1) << k;
            b[2] |= ((nib >> 2) & 1) << k;
            b[3] |= ((nib >> 3) & 1) << k;
        }
        if (n < 64) {  // void past the end: hi = lo = 1, up = ac = 0
            const uint64_t tail = ~0ull << n;
            b[0] |= tail;
            b[1] |= tail;
        }
        hi[w] = b[0];
        lo[w] = b[1];
        up[w] = b[2];
        ac[w] = b[3];
    }
}

int crp_pack_ascii(const uint8_t *text, uint64_t len, uint64_t *hi, uint64_t *lo, uint64_t *up, uint64_t *ac,
                   int n_threads)
{
    if ((len && !text) || !hi || !lo || !up || !ac) return len ? CRP_ERR_INVALID : CRP_OK;
    uint8_t lut[256];
    for (int c = 0; c < 256; ++c) lut[c] = crp::host_classify_char((uint32_t)c);
    const uint64_t n_words = (len + 63) / 64;
    if (n_threads <= 1 || n_words < 4096) {
        pack_range(text, len, 0, n_words, lut, hi, lo, up, ac);
        return CRP_OK;
    }
    // (no allocation, and a thread that cannot be started is replaced by packing its range here: nothing throws)
    constexpr int MAX_THREADS = 64;
    n_threads = std::min(n_threads, MAX_THREADS);
    std::thread pool[MAX_THREADS];
    const uint64_t per = (n_words + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        const uint64_t w0 = std::min<uint64_t>(n_words, per * t), w1 = std::min<uint64_t>(n_words, w0 + per);
        if (w0 >= w1) continue;
        try {
            pool[t] = std::thread(pack_range, text, len, w0, w1, lut, hi, lo, up, ac);
        } catch (...) {
            pack_range(text, len, w0, w1, lut, hi, lo, up, ac);
        }
    }
    for (int t = 0; t < n_threads; ++t)
        if (pool[t].joinable()) pool[t].join();
    return CRP_OK;
}

// -------------------------------------------------------------------- arena
int crp_arena_create(crp_ctx *ctx, uint64_t capacity_words, crp_arena **out)
{
    if (!ctx || !out || capacity_words < 1) return CRP_ERR_INVALID;
    *out = nullptr;
    if (capacity_words > kMaxArenaWords) return CRP_ERR_CAPACITY;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    {   // an arena means uploads: the staging buffers (2 x 32 MiB pinned, 32 MiB device) are set up here, once per context,
        // not inside the first upload
        const int rc = crp::staging_ready(ctx);
        if (rc != CRP_OK) return rc;
        if (!ctx->d_text) {
            CRP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_text), kUploadChunk));
            ctx->d_text_cap = kUploadChunk;
        }
    }
    crp_arena *a = new (std::nothrow) crp_arena();
    if (!a) return CRP_ERR_NOMEM;
    a->ctx = ctx;
    a->cap_words = capacity_words;
    a->padded_words = round_up(capacity_words + 1, crp::ARENA_ALIGN_WORDS);
    const size_t bytes = a->padded_words * sizeof(uint64_t);
    for (int p = 0; p < 4; ++p) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&a->d_plane[p]), bytes);
        if (e == hipSuccess) e = hipMemsetAsync(a->d_plane[p], p < 2 ? 0xFF : 0x00, bytes, ctx->stream);  // all void
        if (e != hipSuccess) {
            ctx->last_error = std::string("arena plane allocation: ") + hipGetErrorString(e);
            crp_arena_destroy(a);
            return CRP_ERR_NOMEM;
        }
    }
    *out = a;
    return CRP_OK;
}

int crp_arena_destroy(crp_arena *a)
{
    if (!a) return CRP_OK;
    (void)hipSetDevice(a->ctx->device);
    (void)hipStreamSynchronize(a->ctx->stream);
    crp::comm_forget_arena(a->ctx, a);
    for (int p = 0; p < 4; ++p) (void)hipFree(a->d_plane[p]);
    (void)hipFree(a->d_tile_cnt);
    (void)hipFree(a->d_tile_off);
    (void)hipFree(a->d_chain[0]);
    (void)hipFree(a->d_chain[1]);
    (void)hipFree(a->d_totals);
    if (a->h_totals) (void)hipHostFree(a->h_totals);
    for (int s = 0; s < 2; ++s) {
        (void)hipFree(a->d_pos[s]);
        (void)hipFree(a->d_score[s]);
        (void)hipFree(a->d_pre[s]);
        (void)hipFree(a->d_ot_raw[s]);
        (void)hipFree(a->d_ot_seed[s]);
        (void)hipFree(a->d_ot_cnt[s]);
    }
    (void)hipFree(a->d_ot_own);
    (void)hipFree(a->d_ann_points);
    (void)hipFree(a->d_ann_ids);
    (void)hipFree(a->d_ann_bucket);
    (void)hipFree(a->d_feat[0]);
    (void)hipFree(a->d_feat[1]);
    delete a;
    return CRP_OK;
}

static int arena_reserve(crp_arena *a, uint64_t len, uint64_t *first_word)
{
    if (a->sealed) return CRP_ERR_STATE;
    const uint64_t need = crp_arena_words_for(len);
    if (a->used_words + need > a->cap_words) return CRP_ERR_CAPACITY;
    *first_word = a->used_words;
    return CRP_OK;
}

int crp_arena_add_contig_ascii(crp_arena *a, const uint8_t *text, uint64_t len, uint64_t *arena_offset)
{
    crp::Range roctx_range("crp: H2D + pack");
    if (!a || (len && !text)) return CRP_ERR_INVALID;
    crp_ctx *ctx = a->ctx;
    uint64_t w_first = 0;
    int rc = arena_reserve(a, len, &w_first);
    if (rc != CRP_OK) return rc;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n_words = crp_arena_words_for(len);
    if (!ctx->d_text) {
        CRP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_text), kUploadChunk));
        ctx->d_text_cap = kUploadChunk;
    }
    rc = crp::staging_ready(ctx);
    if (rc != CRP_OK) return rc;
    // Upload and pack in rounds of kUploadChunk characters (word-aligned cuts), every round through one of the two
    // pinned buffers: the host threads copy round k + 1 out of the caller's pages while round k crosses the link and
    // is packed.  Copies and pack kernels are ordered on the stream, so ONE device text buffer serves all rounds and
    // all contigs, and nothing here waits for the GPU except for a pinned buffer that is still on the link: the call
    // returns when `text` has been read (crp_arena_seal waits for the uploads).  A small contig costs a memcpy and
    // two asynchronous calls, not a stream synchronisation.
    // (a string in pinned memory -- crp_host_alloc -- is read by DMA straight off its pages: no staging copy; the caller keeps
    // it until crp_arena_seal)
    const bool direct = len >= kUploadChunk / 4 && crp::is_pinned_host(text);
    uint64_t c0 = 0;
    do {
        const uint64_t c1 = std::min<uint64_t>(len, c0 + kUploadChunk);
        const bool last = c1 == len;
        const uint64_t w0 = c0 / 64, w1 = last ? n_words : c1 / 64;
        if (c1 > c0 && direct) {
            CRP_HIP(ctx, hipMemcpyAsync(ctx->d_text, text + c0, c1 - c0, hipMemcpyHostToDevice, ctx->stream));
        } else if (c1 > c0) {
            const int b = ctx->pin_next;
            ctx->pin_next = (ctx->pin_next + 1) % crp_ctx::N_PIN;
            if (ctx->pin_busy[b]) {
                CRP_HIP(ctx, hipEventSynchronize(ctx->pin_done[b]));
                ctx->pin_busy[b] = false;
            }
            crp::parallel_copy(ctx->pin[b], text + c0, c1 - c0, ctx->copy_threads);
            CRP_HIP(ctx, hipMemcpyAsync(ctx->d_text, ctx->pin[b], c1 - c0, hipMemcpyHostToDevice, ctx->stream));
            CRP_HIP(ctx, hipEventRecord(ctx->pin_done[b], ctx->stream));
            ctx->pin_busy[b] = true;
        }
        CRP_HIP(ctx, crp::launch_pack(ctx->stream, ctx->d_text, c1 - c0, w1 - w0, a->d_plane[0] + w_first + w0,
                                      a->d_plane[1] + w_first + w0, a->d_plane[2] + w_first + w0,
                                      a->d_plane[3] + w_first + w0));
        c0 = c1;
    } while (c0 < len);
    a->used_words += n_words;
    a->n_contigs += 1;
    a->n_chars += len;
    if (arena_offset) *arena_offset = w_first * 64;
    return CRP_OK;
}

// Several contigs in one call, in order.  Contigs of at least a quarter of a staging buffer go through
// crp_arena_add_contig_ascii one by one (they stream at the link's rate).  SMALL contigs -- an assembly's scaffolds:
// hundreds to hundreds of thousands of them -- are gathered into one pinned buffer, cross the link in ONE copy and are
// packed by ONE launch (pack_groups_kernel: one wave per 64 words of one contig), instead of a copy, a kernel and an
// event each: ~45 us per contig through the single-contig path (626 scaffolds of the bench genome: 30 ms of its
// 54 ms upload), ~1 us here.
int crp_arena_add_contigs_ascii(crp_arena *a, const uint8_t *const *texts, const uint64_t *lens, uint64_t n,
                                uint64_t *arena_offsets)
{
    crp::Range roctx_range("crp: H2D + pack (batch)");
    if (!a || (n && (!texts || !lens))) return CRP_ERR_INVALID;
    crp_ctx *ctx = a->ctx;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    constexpr uint64_t kSmall = crp::STAGE_CHUNK / 4;
    constexpr uint64_t kTextRoom = crp::STAGE_CHUNK - (1ull << 20);          // characters per batch; the group table
    constexpr uint32_t kMaxGroups = (1u << 20) / sizeof(crp::PackGroup) - 1;  // sits in the buffer's last MiB
    for (uint64_t i = 0; i < n; ++i)
        if (lens[i] && !texts[i]) return CRP_ERR_INVALID;
    uint64_t i = 0;
    while (i < n) {
        if (lens[i] >= kSmall) {
            const int rc = crp_arena_add_contig_ascii(a, texts[i], lens[i], arena_offsets ? arena_offsets + i : nullptr);
            if (rc != CRP_OK) return rc;
            ++i;
            continue;
        }
        // a batch of consecutive small contigs
        if (!ctx->d_text) {
            CRP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_text), kUploadChunk));
            ctx->d_text_cap = kUploadChunk;
        }
        int rc = crp::staging_ready(ctx);
        if (rc != CRP_OK) return rc;
        const int b = ctx->pin_next;
        ctx->pin_next = (ctx->pin_next + 1) % crp_ctx::N_PIN;
        if (ctx->pin_busy[b]) {
            CRP_HIP(ctx, hipEventSynchronize(ctx->pin_done[b]));
            ctx->pin_busy[b] = false;
        }
        // the group table is written where it travels from: the last MiB of the pinned buffer (no allocation in this call)
        crp::PackGroup *groups = reinterpret_cast<crp::PackGroup *>(ctx->pin[b] + kTextRoom);
        uint32_t n_table = 0;
        uint64_t cursor = 0;
        const uint64_t first = i;
        while (i < n && lens[i] < kSmall) {
            const uint64_t len = lens[i], n_words = crp_arena_words_for(len), n_groups = (n_words + 63) / 64;
            // (a group reads 4096 bytes from its start: keep that inside the text part of the buffer)
            if (cursor + n_groups * 4096 > kTextRoom || n_table + n_groups > kMaxGroups) break;
            uint64_t w_first = 0;
            rc = arena_reserve(a, len, &w_first);
            if (rc != CRP_OK) break;
            if (len) std::memcpy(ctx->pin[b] + cursor, texts[i], len);
            for (uint64_t g = 0; g < n_groups; ++g) {
                const uint64_t c0 = g * 4096;
                groups[n_table++] = crp::PackGroup{(uint32_t)(cursor + c0), (uint32_t)(len > c0 ? std::min<uint64_t>(4096, len - c0) : 0),
                                                   w_first + g * 64, (uint32_t)std::min<uint64_t>(64, n_words - g * 64), 0};
            }
            a->used_words += n_words;
            a->n_contigs += 1;
            a->n_chars += len;
            if (arena_offsets) arena_offsets[i] = w_first * 64;
            cursor += (len + 15) & ~(uint64_t)15;  // 16-byte loads in the kernel
            ++i;
        }
        if (i == first) return rc != CRP_OK ? rc : CRP_ERR_CAPACITY;  // not even one contig fitted (cannot happen for small ones)
        // one copy for the characters, one for the table (both from the pinned buffer), one launch
        CRP_HIP(ctx, hipMemcpyAsync(ctx->d_text, ctx->pin[b], cursor, hipMemcpyHostToDevice, ctx->stream));
        CRP_HIP(ctx, hipMemcpyAsync(ctx->d_text + kTextRoom, ctx->pin[b] + kTextRoom, n_table * sizeof(crp::PackGroup),
                                    hipMemcpyHostToDevice, ctx->stream));
        CRP_HIP(ctx, hipEventRecord(ctx->pin_done[b], ctx->stream));
        ctx->pin_busy[b] = true;
        CRP_HIP(ctx, crp::launch_pack_groups(ctx->stream, ctx->d_text, reinterpret_cast<const crp::PackGroup *>(ctx->d_text + kTextRoom),
                                             n_table, a->d_plane[0], a->d_plane[1], a->d_plane[2], a->d_plane[3]));
        if (rc != CRP_OK) return rc;  // (the arena ran out of room in the middle of the batch: what fitted is uploaded)
    }
    return CRP_OK;
}

int crp_arena_add_contig_packed(crp_arena *a, const uint64_t *hi, const uint64_t *lo, const uint64_t *up,
                                const uint64_t *ac, uint64_t len, uint64_t *arena_offset)
{
    crp::Range roctx_range("crp: H2D planes");
    if (!a) return CRP_ERR_INVALID;
    if (len && (!hi || !lo || !up || !ac)) return CRP_ERR_INVALID;
    crp_ctx *ctx = a->ctx;
    uint64_t w_first = 0;
    int rc = arena_reserve(a, len, &w_first);
    if (rc != CRP_OK) return rc;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t body = (len + 63) / 64;
    const uint64_t *src[4] = {hi, lo, up, ac};
    if (body)
        for (int p = 0; p < 4; ++p) {
            rc = crp::staged_h2d(ctx, a->d_plane[p] + w_first, src[p], body * sizeof(uint64_t));
            if (rc != CRP_OK) return rc;
        }
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // caller may free its planes on return
    // the separator word after the body is still void from crp_arena_create
    a->used_words += crp_arena_words_for(len);
    a->n_contigs += 1;
    a->n_chars += len;
    if (arena_offset) *arena_offset = w_first * 64;
    return CRP_OK;
}

// Word 3 of both chain headers = device address of the arena's pinned h_totals: the single-launch kernel
// writes its fail flag and totals there as well, and the host reads them after the stream synchronisation
// without a device-to-host copy in between.
static int chain_headers_point_at_host(crp_arena *a)
{
    crp_ctx *ctx = a->ctx;
    void *dev = nullptr;
    CRP_HIP(ctx, hipHostGetDevicePointer(&dev, a->h_totals, 0));
    a->h_totals[3] = reinterpret_cast<uint64_t>(dev);
    for (int b = 0; b < 2; ++b)
        CRP_HIP(ctx, hipMemcpyAsync(a->d_chain[b] + 3, &a->h_totals[3], sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    return CRP_OK;
}

// sync = false (crp_stream.cpp): nothing here needs the host to wait -- the scratch is zeroed and the headers written in
// stream order, before whatever scan is queued next
static int arena_seal_impl(crp_arena *a, bool sync)
{
    if (!a) return CRP_ERR_INVALID;
    if (a->sealed) return CRP_OK;
    crp_ctx *ctx = a->ctx;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    // Tile geometry (crp_kernels.h), fixed for the arena's life: by default from the arena's size against the GPU's -- an
    // arena that gives a CU one or two LARGE tiles at most is all latency and takes the SMALL shape (half the tile, half the
    // life), anything bigger the throughput shape
    const uint64_t large_tiles = round_up(a->used_words, crp::GeoLarge::WORDS) / crp::GeoLarge::WORDS;
    const uint64_t cus = (uint64_t)std::max(1, ctx->n_cu);
    a->geo = ctx->geometry > 0 ? ctx->geometry - 1 : 2 * large_tiles < 3 * cus ? crp::GEO_SMALL : crp::GEO_LARGE;
    const uint64_t tw = (uint64_t)crp::tile_words(a->geo);
    const uint64_t eff = round_up(a->used_words, tw);  // <= padded_words
    a->n_tiles = (uint32_t)(eff / tw);
    if (a->n_tiles > a->tile_cap) {  // (an arena that is reset and filled again keeps its scratch: crp::arena_reset)
        (void)hipFree(a->d_tile_cnt);
        (void)hipFree(a->d_tile_off);
        (void)hipFree(a->d_chain[0]);
        (void)hipFree(a->d_chain[1]);
        a->d_tile_cnt = a->d_tile_off = nullptr;
        a->d_chain[0] = a->d_chain[1] = nullptr;
        a->tile_cap = 0;
        // room for the finest geometry over the whole capacity, so that a refill never allocates
        const uint32_t cap = (uint32_t)std::max<uint64_t>(a->n_tiles, a->padded_words / (uint64_t)crp::tile_words(crp::GEO_SMALL) + 1);
        CRP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&a->d_tile_cnt), cap * sizeof(uint2)));
        CRP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&a->d_tile_off), cap * sizeof(uint2)));
        for (int b = 0; b < 2; ++b) CRP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&a->d_chain[b]), crp::chain_bytes(cap)));
        a->tile_cap = cap;
    }
    for (int b = 0; b < 2; ++b) CRP_HIP(ctx, hipMemsetAsync(a->d_chain[b], 0, crp::chain_bytes(a->n_tiles), ctx->stream));
    a->chain_cur = 0;
    if (!a->d_totals) CRP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&a->d_totals), 2 * sizeof(uint64_t)));
    if (!a->h_totals) CRP_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&a->h_totals), 4 * sizeof(uint64_t), hipHostMallocDefault));
    int rc = chain_headers_point_at_host(a);
    if (rc != CRP_OK) return rc;
    if (sync) CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    a->sealed = true;
    return CRP_OK;
}

int crp_arena_seal(crp_arena *a) { return arena_seal_impl(a, true); }

int crp_arena_tiles(const crp_arena *a, int *geometry, uint64_t *n_tiles, uint64_t *tile_words)
{
    if (!a) return CRP_ERR_INVALID;
    if (!a->sealed) return CRP_ERR_STATE;
    if (geometry) *geometry = a->geo + 1;
    if (n_tiles) *n_tiles = a->n_tiles;
    if (tile_words) *tile_words = (uint64_t)crp::tile_words(a->geo);
    return CRP_OK;
}

int crp_arena_stats(const crp_arena *a, uint64_t *n_contigs, uint64_t *n_chars, uint64_t *n_words)
{
    if (!a) return CRP_ERR_INVALID;
    if (n_contigs) *n_contigs = a->n_contigs;
    if (n_chars) *n_chars = a->n_chars;
    if (n_words) *n_words = a->used_words;
    return CRP_OK;
}

// --------------------------------------------------------------- scan+score
using crp::grow;
using crp::prof_begin;
using crp::prof_collect;
using crp::prof_end;

// scan flags of one crp_scan_score call
struct ScanWant {
    bool pre, seeds;
};

static int reserve_tables(crp_arena *a, const uint64_t n[2], ScanWant want)
{
    crp_ctx *ctx = a->ctx;
    for (int s = 0; s < 2; ++s) {
        uint64_t cap_pos = a->tab_cap[s], cap_score = a->tab_cap[s];
        int rc = grow(ctx, reinterpret_cast<void **>(&a->d_pos[s]), &cap_pos, n[s], sizeof(uint32_t));
        if (rc == CRP_OK) rc = grow(ctx, reinterpret_cast<void **>(&a->d_score[s]), &cap_score, n[s], sizeof(double));
        if (rc != CRP_OK) { a->tab_cap[s] = 0; return rc; }
        a->tab_cap[s] = std::min(cap_pos, cap_score);
        if (want.pre) {
            rc = grow(ctx, reinterpret_cast<void **>(&a->d_pre[s]), &a->pre_cap[s], n[s], sizeof(double));
            if (rc != CRP_OK) return rc;
        }
        if (want.seeds) {
            rc = grow(ctx, reinterpret_cast<void **>(&a->d_ot_raw[s]), &a->raw_cap[s], n[s], sizeof(uint32_t));
            if (rc != CRP_OK) return rc;
        }
    }
    return CRP_OK;
}

// every column the launch writes holds at least cap_* rows
static crp::HitTables table_args(const crp_arena *a, ScanWant want)
{
    uint64_t cap[2];
    for (int s = 0; s < 2; ++s) {
        cap[s] = a->tab_cap[s];
        if (want.pre) cap[s] = std::min(cap[s], a->pre_cap[s]);
        if (want.seeds) cap[s] = std::min(cap[s], a->raw_cap[s]);
    }
    return crp::HitTables{a->d_pos[0], a->d_score[0], want.pre ? a->d_pre[0] : nullptr,
                          a->d_pos[1], a->d_score[1], want.pre ? a->d_pre[1] : nullptr, cap[0], cap[1],
                          want.seeds ? a->d_ot_raw[0] : nullptr, want.seeds ? a->d_ot_raw[1] : nullptr};
}

static bool tables_exist(const crp_arena *a, ScanWant want)
{
    for (int s = 0; s < 2; ++s)
        if (!a->tab_cap[s] || (want.pre && !a->pre_cap[s]) || (want.seeds && !a->raw_cap[s])) return false;
    return true;
}

// count -> tile scan -> emit.  The table sizes must be known before the emit: once an arena has been
// scanned the tables exist and the emit is queued speculatively (stores are bounds-checked, totals
// compared afterwards); the very first scan of an arena waits for the totals instead.
static int scan_two_pass(crp_arena *a, const crp::Planes &pl, uint64_t eff_words, int guide_len, ScanWant want,
                         uint64_t n[2])
{
    crp_ctx *ctx = a->ctx;
    const uint32_t n_tiles = a->n_tiles;
    const bool speculative = tables_exist(a, want);
    prof_begin(ctx, 0);
    CRP_HIP(ctx, crp::launch_count(ctx->stream, a->geo, pl, eff_words, guide_len, a->d_tile_cnt, n_tiles));
    prof_end(ctx, 0);
    prof_begin(ctx, 1);
    CRP_HIP(ctx, crp::launch_tile_scan(ctx->stream, a->d_tile_cnt, n_tiles, a->d_tile_off, a->d_totals));
    prof_end(ctx, 1);
    if (!speculative) {
        CRP_HIP(ctx, hipMemcpyAsync(a->h_totals, a->d_totals, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        n[0] = a->h_totals[0];
        n[1] = a->h_totals[1];
        int rc = reserve_tables(a, n, want);
        if (rc != CRP_OK) return rc;
    }
    crp::HitTables out = table_args(a, want);
    prof_begin(ctx, 2);
    CRP_HIP(ctx, crp::launch_emit(ctx->stream, a->geo, pl, eff_words, guide_len, a->d_tile_off, out));
    prof_end(ctx, 2);
    CRP_HIP(ctx, hipMemcpyAsync(a->h_totals, a->d_totals, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int kind = 0; kind < 3; ++kind) prof_collect(ctx, kind);
    n[0] = a->h_totals[0];
    n[1] = a->h_totals[1];
    if (n[0] > out.cap_plus || n[1] > out.cap_minus) {
        // a speculative run met more hits than the tables hold (e.g. another guide length):
        // nothing was written out of bounds; size exactly and emit again
        int rc = reserve_tables(a, n, want);
        if (rc != CRP_OK) return rc;
        out = table_args(a, want);
        CRP_HIP(ctx, crp::launch_emit(ctx->stream, a->geo, pl, eff_words, guide_len, a->d_tile_off, out));
        CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CRP_OK;
}

// One launch: table offsets come from a chained scan inside the emit kernel.  The tables are sized from the
// previous scan of this arena (hit counts are a property of the sealed arena) or, the first time, from a density
// guess; a scan that finds more hits than fit writes nothing out of bounds and is repeated once with the exact
// sizes.  The launch and its collection are two steps so that a caller with several devices (crp_node.cpp) can
// queue the launch on every device before it waits for any of them.
static int single_launch(crp_arena *a, const crp::Planes &pl, uint64_t eff_words, int guide_len, ScanWant want,
                         const uint64_t rows[2], crp::HitTables *out)
{
    crp_ctx *ctx = a->ctx;
    int rc = reserve_tables(a, rows, want);
    if (rc != CRP_OK) return rc;
    *out = table_args(a, want);
    uint64_t *cur = a->d_chain[a->chain_cur], *next = a->d_chain[a->chain_cur ^ 1];
    a->h_totals[0] = a->h_totals[1] = a->h_totals[2] = 0;
    prof_begin(ctx, 2);
    CRP_HIP(ctx, crp::launch_emit_chained(ctx->stream, a->geo, pl, eff_words, guide_len, cur, next, *out, ctx->mute_tile,
                                          ctx->chain_timeout_ticks));
    prof_end(ctx, 2);
    return CRP_OK;
}

// Waits for single_launch's kernel.  CRP_OK with *too_small = true: the tables held fewer rows than n[] (nothing was
// written out of bounds).  CRP_ERR_STATE with *chain_failed = true: a look-back timed out (the caller then runs the
// three-launch sequence).
static int single_collect(crp_arena *a, const crp::HitTables &out, uint64_t n[2], bool *chain_failed, bool *too_small, bool kernel_done = false)
{
    crp_ctx *ctx = a->ctx;
    *chain_failed = *too_small = false;
    // header: fail << 32, total '+', total '-' -- the kernel also writes it to h_totals (pinned)
    // (kernel_done: the caller has waited for an event behind the launch -- crp_stream.cpp, whose stream already holds the
    // uploads of later slices and must not be waited for as a whole)
    if (!kernel_done) CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    a->chain_cur ^= 1;  // the kernel left the other buffer zeroed
    if (a->h_totals[0] >> 32) {  // fail flag: a look-back spin ran out; neither buffer can be trusted now
        for (int b = 0; b < 2; ++b)
            CRP_HIP(ctx, hipMemsetAsync(a->d_chain[b], 0, crp::chain_bytes(a->n_tiles), ctx->stream));
        int rc2 = chain_headers_point_at_host(a);
        if (rc2 != CRP_OK) return rc2;
        *chain_failed = true;
        return CRP_ERR_STATE;
    }
    prof_collect(ctx, 2);
    n[0] = a->h_totals[1];
    n[1] = a->h_totals[2];
    *too_small = n[0] > out.cap_plus || n[1] > out.cap_minus;
    return CRP_OK;
}

static ScanWant scan_want(int guide_len, int flags)
{
    // seed words come out of the l = 20 kernel only; for other lengths the off-target step derives them itself
    return ScanWant{(flags & CRP_SCAN_PRE) != 0, (flags & CRP_SCAN_SEEDS) != 0 && guide_len == 20};
}

}  // extern "C"

namespace crp {

// An arena emptied for another genome slice (crp_stream.cpp): same planes (all void again, in stream order), same scratch,
// same tables -- nothing is allocated or freed, so refilling costs a 4-plane memset and no synchronisation.  Whatever was
// queued for the arena must have completed or be ordered before on the context's stream.
int arena_reset(crp_arena *a)
{
    if (!a) return CRP_ERR_INVALID;
    crp_ctx *ctx = a->ctx;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = std::min<uint64_t>(a->padded_words, round_up(a->used_words + 1, crp::ARENA_ALIGN_WORDS)) * sizeof(uint64_t);
    for (int p = 0; p < 4; ++p) CRP_HIP(ctx, hipMemsetAsync(a->d_plane[p], p < 2 ? 0xFF : 0x00, bytes, ctx->stream));  // all void
    a->used_words = 1;
    a->n_contigs = a->n_chars = 0;
    a->sealed = false;
    a->have_hits = a->have_pre = a->have_raw = a->have_feat = a->have_track = false;
    a->n_hits[0] = a->n_hits[1] = 0;
    a->scan_pending = 0;
    a->ot_epoch = 0;
    return CRP_OK;
}

int arena_seal_async(crp_arena *a) { return arena_seal_impl(a, false); }

// the tables a first scan of `chars` characters would ask for (scan_begin's density guess), allocated ahead of it
int arena_reserve_tables(crp_arena *a, uint64_t chars, bool want_pre)
{
    if (!a) return CRP_ERR_INVALID;
    CRP_HIP(a->ctx, hipSetDevice(a->ctx->device));
    const uint64_t rows[2] = {chars / 8 + 1024, chars / 8 + 1024};
    return reserve_tables(a, rows, ScanWant{want_pre, false});
}

// First half of crp_scan_score: validates, and in the single-launch mode queues the kernel on the context's stream
// without waiting for it.  (Three-launch mode: nothing is queued; scan_finish runs the whole sequence.)
int scan_begin(crp_arena *a, int guide_len, int flags)
{
    if (!a || (flags & ~(CRP_SCAN_PRE | CRP_SCAN_SEEDS))) return CRP_ERR_INVALID;
    if (!a->sealed) return CRP_ERR_STATE;
    if (guide_len < 0 || guide_len > 50) return CRP_ERR_UNSUPPORTED;
    crp_ctx *ctx = a->ctx;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    a->have_hits = false;
    a->have_raw = false;
    a->have_feat = false;
    a->scan_pending = 0;
    a->pend_guide_len = guide_len;
    a->pend_flags = flags;
    if (ctx->two_pass || ctx->two_pass_latched) {
        a->scan_pending = 2;
        return CRP_OK;
    }
    const ScanWant want = scan_want(guide_len, flags);
    const uint64_t eff_words = (uint64_t)a->n_tiles * (uint64_t)crp::tile_words(a->geo);
    crp::Planes pl{{a->d_plane[0], a->d_plane[1], a->d_plane[2], a->d_plane[3]}};
    const uint64_t rows[2] = {std::max<uint64_t>(a->tab_cap[0], a->n_chars / 8 + 1024),
                              std::max<uint64_t>(a->tab_cap[1], a->n_chars / 8 + 1024)};
    const int rc = single_launch(a, pl, eff_words, guide_len, want, rows, &a->pend_out);
    if (rc != CRP_OK) return rc;
    a->scan_pending = 1;
    return CRP_OK;
}

int scan_finish(crp_arena *a, uint64_t *n_plus, uint64_t *n_minus, bool kernel_done)
{
    if (!a) return CRP_ERR_INVALID;
    if (!a->scan_pending) return CRP_ERR_STATE;
    crp_ctx *ctx = a->ctx;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    const int guide_len = a->pend_guide_len;
    const ScanWant want = scan_want(guide_len, a->pend_flags);
    const uint64_t eff_words = (uint64_t)a->n_tiles * (uint64_t)crp::tile_words(a->geo);
    crp::Planes pl{{a->d_plane[0], a->d_plane[1], a->d_plane[2], a->d_plane[3]}};
    uint64_t n[2] = {0, 0};
    int rc;
    if (a->scan_pending == 2) {
        rc = scan_two_pass(a, pl, eff_words, guide_len, want, n);
    } else {
        bool chain_failed = false, too_small = false;
        rc = single_collect(a, a->pend_out, n, &chain_failed, &too_small, kernel_done);
        if (rc == CRP_OK && too_small) {  // once more, with the exact sizes
            const uint64_t rows[2] = {n[0], n[1]};
            rc = single_launch(a, pl, eff_words, guide_len, want, rows, &a->pend_out);
            if (rc == CRP_OK) rc = single_collect(a, a->pend_out, n, &chain_failed, &too_small);
            if (rc == CRP_OK && too_small) {
                ctx->last_error = "emit kernel: tables still too small after resizing";
                rc = CRP_ERR_STATE;
            }
        }
        if (chain_failed) {
            // A workgroup waited longer than CRP_OPT_CHAIN_TIMEOUT_US for a predecessor's counts (a
            // shared or pre-empted GPU can do that; a wrong assumption about dispatch order would
            // too).  Nothing of that launch is used: this scan runs again as three launches.  The
            // next scan tries the single launch again; three failures in a row latch.
            ctx->chain_timeouts++;
            if (++ctx->timeout_streak >= 3) ctx->two_pass_latched = true;
            ctx->last_error = "single-launch scan: a chained look-back timed out; this scan was repeated with the "
                              "count / scan / emit sequence";
            rc = scan_two_pass(a, pl, eff_words, guide_len, want, n);
        } else if (rc == CRP_OK) {
            ctx->timeout_streak = 0;
        }
    }
    a->scan_pending = 0;
    if (rc != CRP_OK) return rc;
    a->n_hits[0] = n[0];
    a->n_hits[1] = n[1];
    a->have_hits = true;
    a->have_pre = want.pre;
    a->have_raw = want.seeds;
    if (n_plus) *n_plus = n[0];
    if (n_minus) *n_minus = n[1];
    return CRP_OK;
}

}  // namespace crp

extern "C" {

int crp_scan_score(crp_arena *a, int guide_len, int flags, uint64_t *n_plus, uint64_t *n_minus)
{
    crp::Range roctx_range("crp: scan + score");
    const int rc = crp::scan_begin(a, guide_len, flags);
    if (rc != CRP_OK) return rc;
    return crp::scan_finish(a, n_plus, n_minus);
}

int crp_fetch_hits(crp_arena *a, uint32_t *pos_plus, double *pre_plus, double *score_plus, uint32_t *pos_minus,
                   double *pre_minus, double *score_minus)
{
    crp::Range roctx_range("crp: D2H tables");
    if (!a) return CRP_ERR_INVALID;
    if (!a->have_hits) return CRP_ERR_STATE;
    if ((pre_plus || pre_minus) && !a->have_pre) return CRP_ERR_STATE;
    crp_ctx *ctx = a->ctx;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t *hp[2] = {pos_plus, pos_minus};
    double *hs[2] = {score_plus, score_minus};
    double *hr[2] = {pre_plus, pre_minus};
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int s = 0; s < 2; ++s) {
        const uint64_t n = a->n_hits[s];
        if (!n) continue;
        int rc = CRP_OK;
        if (hp[s]) rc = crp::staged_d2h(ctx, hp[s], a->d_pos[s], n * sizeof(uint32_t));
        if (rc == CRP_OK && hs[s]) rc = crp::staged_d2h(ctx, hs[s], a->d_score[s], n * sizeof(double));
        if (rc == CRP_OK && hr[s]) rc = crp::staged_d2h(ctx, hr[s], a->d_pre[s], n * sizeof(double));
        if (rc != CRP_OK) return rc;
    }
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CRP_OK;
}

int crp_hits_counts(const crp_arena *a, uint64_t *n_plus, uint64_t *n_minus)
{
    if (!a) return CRP_ERR_INVALID;
    if (!a->have_hits) return CRP_ERR_STATE;
    if (n_plus) *n_plus = a->n_hits[0];
    if (n_minus) *n_minus = a->n_hits[1];
    return CRP_OK;
}

int crp_hits_device(crp_arena *a, void **pos_plus, void **score_plus, void **pos_minus, void **score_minus)
{
    if (!a) return CRP_ERR_INVALID;
    if (!a->have_hits) return CRP_ERR_STATE;
    if (pos_plus) *pos_plus = a->d_pos[0];
    if (score_plus) *score_plus = a->d_score[0];
    if (pos_minus) *pos_minus = a->d_pos[1];
    if (score_minus) *score_minus = a->d_score[1];
    return CRP_OK;
}

// ------------------------------------------------------------------- seam 2
int crp_score_30mers(crp_ctx *ctx, const uint8_t *rows, uint64_t n, int order, double *pre, double *score)
{
    crp::Range roctx_range("crp: score 30-mers");
    if (!ctx || (n && (!rows || !score))) return CRP_ERR_INVALID;
    if (order < CRP_ORDER_BODY4 || order > CRP_ORDER_DOT1) return CRP_ERR_INVALID;
    if (n == 0) return CRP_OK;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->d_rows_cap < n) {
        CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_rows);
        (void)hipFree(ctx->d_rpre);
        (void)hipFree(ctx->d_rscore);
        ctx->d_rows = nullptr;
        ctx->d_rpre = ctx->d_rscore = nullptr;
        ctx->d_rows_cap = 0;
        const uint64_t cap = n + n / 8 + 64;
        CRP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_rows), cap * 30));
        CRP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_rpre), cap * sizeof(double)));
        CRP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_rscore), cap * sizeof(double)));
        ctx->d_rows_cap = cap;
    }
    int rc = crp::staged_h2d(ctx, ctx->d_rows, rows, n * 30);  // (the caller's rows and results are pageable: our staging, not the runtime's)
    if (rc != CRP_OK) return rc;
    CRP_HIP(ctx, crp::launch_score30(ctx->stream, ctx->d_rows, n, order, ctx->d_rpre, ctx->d_rscore));
    if (pre) rc = crp::staged_d2h(ctx, pre, ctx->d_rpre, n * sizeof(double));
    if (rc == CRP_OK) rc = crp::staged_d2h(ctx, score, ctx->d_rscore, n * sizeof(double));
    if (rc != CRP_OK) return rc;
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CRP_OK;
}

// -------------------------------------------------------------- measurement
int crp_profile_enable(crp_ctx *ctx, int on)
{
    if (!ctx) return CRP_ERR_INVALID;
    ctx->profiling = on < 0 ? 0 : (on > 2 ? 2 : on);
    return CRP_OK;
}

int crp_profile_read(crp_ctx *ctx, double ms[3], uint64_t launches[3], int reset)
{
    if (!ctx) return CRP_ERR_INVALID;
    for (int k = 0; k < 3; ++k) {
        if (ms) ms[k] = ctx->ms[k];
        if (launches) launches[k] = ctx->launches[k];
        if (reset) {
            ctx->ms[k] = 0;
            ctx->launches[k] = 0;
        }
    }
    return CRP_OK;
}

int crp_profile_read_kind(crp_ctx *ctx, int kind, double *ms, uint64_t *launches, int reset)
{
    if (!ctx || kind < 0 || kind >= CRP_K_KINDS) return CRP_ERR_INVALID;
    if (ms) *ms = ctx->ms[kind];
    if (launches) *launches = ctx->launches[kind];
    if (reset) {
        ctx->ms[kind] = 0;
        ctx->launches[kind] = 0;
    }
    return CRP_OK;
}

int crp_configure(crp_ctx *ctx, int option, int64_t value)
{
    if (!ctx) return CRP_ERR_INVALID;
    switch (option) {
        case CRP_OPT_TWO_PASS:
            ctx->two_pass = value != 0;
            if (!ctx->two_pass) {  // asking for the single launch again clears a latch
                ctx->two_pass_latched = false;
                ctx->timeout_streak = 0;
            }
            return CRP_OK;
        case CRP_OPT_CHAIN_TIMEOUT_US:
            if (value < 1 || value > 10000000) return CRP_ERR_INVALID;
            ctx->chain_timeout_ticks = (uint32_t)(value * 100);  // 100 MHz counter
            return CRP_OK;
        case CRP_OPT_TILE_GEOMETRY:
            if (value < 0 || value > crp::GEO_COUNT) return CRP_ERR_INVALID;
            ctx->geometry = (int)value;  // arenas sealed from now on
            return CRP_OK;
        default: return CRP_ERR_INVALID;
    }
}

int crp_query(const crp_ctx *ctx, int what, int64_t *value)
{
    if (!ctx || !value) return CRP_ERR_INVALID;
    switch (what) {
        case CRP_Q_CHAIN_TIMEOUTS: *value = (int64_t)ctx->chain_timeouts; return CRP_OK;
        case CRP_Q_TWO_PASS_ACTIVE: *value = (ctx->two_pass || ctx->two_pass_latched) ? 1 : 0; return CRP_OK;
        case CRP_Q_COMM_WORLD: *value = crp::comm_world(ctx); return CRP_OK;
        case CRP_Q_COMM_RANK: *value = crp::comm_rank(ctx); return CRP_OK;
        case CRP_Q_GATHER_BYTES: *value = (int64_t)crp::comm_gather_bytes(ctx); return CRP_OK;
        case CRP_Q_HBM_FREE:
        case CRP_Q_HBM_TOTAL: {
            size_t free_b = 0, total_b = 0;
            if (hipSetDevice(ctx->device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return CRP_ERR_HIP;
            *value = (int64_t)(what == CRP_Q_HBM_FREE ? free_b : total_b);
            return CRP_OK;
        }
        default: return CRP_ERR_INVALID;
    }
}

#ifndef CRP_BUILD_ID
#define CRP_BUILD_ID "unknown"
#endif
const char *crp_build_id(void) { return CRP_BUILD_ID; }

int crp_count_scored(crp_arena *a, uint64_t *n_scored)
{
    if (!a || !n_scored) return CRP_ERR_INVALID;
    if (!a->have_hits) return CRP_ERR_STATE;
    crp_ctx *ctx = a->ctx;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    CRP_HIP(ctx, hipMemsetAsync(ctx->d_scalar, 0, sizeof(uint64_t), ctx->stream));
    for (int s = 0; s < 2; ++s)
        CRP_HIP(ctx, crp::launch_count_scored(ctx->stream, a->d_score[s], a->n_hits[s], ctx->d_scalar));
    CRP_HIP(ctx, hipMemcpyAsync(ctx->h_scalar, ctx->d_scalar, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_scored = ctx->h_scalar[0];
    return CRP_OK;
}

int crp_arena_composition(crp_arena *a, uint64_t *n_plain, uint64_t *n_other)
{
    if (!a) return CRP_ERR_INVALID;
    if (!a->sealed) return CRP_ERR_STATE;
    crp_ctx *ctx = a->ctx;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    CRP_HIP(ctx, hipMemsetAsync(ctx->d_scalar, 0, sizeof(uint64_t), ctx->stream));
    CRP_HIP(ctx, crp::launch_count_plain(ctx->stream, a->d_plane[2], a->d_plane[3], a->used_words, ctx->d_scalar));
    CRP_HIP(ctx, hipMemcpyAsync(ctx->h_scalar, ctx->d_scalar, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_plain) *n_plain = ctx->h_scalar[0];
    if (n_other) *n_other = a->n_chars - ctx->h_scalar[0];
    return CRP_OK;
}

int crp_synchronize(crp_ctx *ctx)
{
    if (!ctx) return CRP_ERR_INVALID;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CRP_OK;
}

}  // extern "C"
