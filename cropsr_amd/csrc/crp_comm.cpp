// crp_comm.cpp -- the multi-GPU exchange of the path, on RCCL, inside the library.
//
// One process per GPU.  The scan itself needs no collective (contigs are independent,
// CROPSR.py:409); what crosses xGMI is
//   * crp_gather_hits      the final gatherv of the per-rank hit tables to a root.  RCCL has no
//                          gatherv primitive: ncclAllGather of the two counts, then
//                          ncclGroupStart / ncclSend (peers) | ncclRecv (root), one per column /
//                          ncclGroupEnd.  Every peer->root transfer rides its own point-to-point
//                          xGMI link, so the step is bounded by one link, not by a ring;
//   * crp_offtarget_reduce (crp_offtarget.hip) the all-reduce of the 64 MiB site histogram;
//   * crp_comm_barrier / crp_comm_allreduce_f64: bench fences and sums.
// librccl.so is loaded with dlopen on the first crp_comm_* call -- a single-GPU process never
// pays for it, and the library has no link-time dependency on RCCL.  The communicator's unique id is
// created here (rank 0) and carried to the other ranks by the host (cropsr_amd/rendezvous.py).
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "crp_internal.h"
#include "crp_rccl.h"
#include "crp_roctx.h"

namespace crp {

static Rccl g_rccl;
static std::once_flag g_rccl_once;

template <class F>
static bool bind(void *h, const char *name, F &fn)
{
    fn = reinterpret_cast<F>(dlsym(h, name));
    return fn != nullptr;
}

const Rccl *rccl()
{
    std::call_once(g_rccl_once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *h = nullptr;
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) {
            g_rccl.error = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found");
            return;
        }
        Rccl &r = g_rccl;
        const bool ok = bind(h, "ncclGetUniqueId", r.GetUniqueId) && bind(h, "ncclCommInitRank", r.CommInitRank) &&
                        bind(h, "ncclCommInitAll", r.CommInitAll) &&
                        bind(h, "ncclCommDestroy", r.CommDestroy) && bind(h, "ncclCommAbort", r.CommAbort) &&
                        bind(h, "ncclAllGather", r.AllGather) && bind(h, "ncclAllReduce", r.AllReduce) &&
                        bind(h, "ncclSend", r.Send) && bind(h, "ncclRecv", r.Recv) &&
                        bind(h, "ncclGroupStart", r.GroupStart) && bind(h, "ncclGroupEnd", r.GroupEnd) &&
                        bind(h, "ncclGetErrorString", r.GetErrorString);
        if (!ok) {
            g_rccl.error = "librccl.so lacks a required symbol";
            return;
        }
        r.handle = h;
    });
    return g_rccl.handle ? &g_rccl : nullptr;
}

const std::string &rccl_load_error() { return g_rccl.error; }

}  // namespace crp

using crp::rccl;
using crp::Rccl;

struct crp_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0;
    // scratch in HBM: counts (2 per rank) and small reductions
    uint64_t *d_counts = nullptr;  // 4 * world {n_plus, n_minus, status, position buckets} + world (second agreement round)
    double *d_small = nullptr;     // 64 doubles in, 64 out
    std::vector<uint64_t> counts;  // last gather: {n_plus, n_minus} per rank
    std::vector<uint64_t> raw;     // host copy of d_counts
    int test_fail = 0;             // test hook, environment CRP_TEST_GATHER_FAIL: 1 = the root's receive buffers
                                   // "do not fit", 2 = the last rank's arena "has no tables", 3 = the last rank sends
                                   // one row too few of its score column (a protocol error only a checking transport sees)
    // root's receive side of the last crp_gather_hits: column-wise, peers back to back in rank order
    uint32_t *d_gpos[2] = {nullptr, nullptr};
    double *d_gscore[2] = {nullptr, nullptr};
    uint4 *d_got[2] = {nullptr, nullptr};  // CRP_GATHER_OFFTARGET: per-hit counts
    uint32_t *d_gfeat[2] = {nullptr, nullptr};  // CRP_GATHER_FEATURES: per-hit label-set ids
    uint64_t gpos_cap[2] = {0, 0}, gscore_cap[2] = {0, 0}, got_cap[2] = {0, 0}, gfeat_cap[2] = {0, 0};
    // CRP_GATHER_POS16 (crp_gather.hip): a peer's packed positions and bucket starts; at the root the staging area they
    // arrive in, peer after peer (every peer's rows at a multiple of 8)
    uint16_t *d_lo16[2] = {nullptr, nullptr}, *d_glo16[2] = {nullptr, nullptr};
    uint32_t *d_bstart[2] = {nullptr, nullptr}, *d_gbstart[2] = {nullptr, nullptr};
    uint64_t lo16_cap[2] = {0, 0}, bstart_cap[2] = {0, 0}, glo16_cap[2] = {0, 0}, gbstart_cap[2] = {0, 0};
    uint64_t bytes_to_root = 0;  // of the last gather: what this rank sent (peer) or received (root)
    int gflags = 0;
    std::vector<uint64_t> goff[2];  // element offset of every rank's slice (root's own slice: unused)
    int groot = -1;
    crp_arena *garena = nullptr;    // root's own contribution stays in its arena
    bool have_gather = false;
};

#define CRP_NCCL(ctx, call)                                                                       \
    do {                                                                                          \
        ncclResult_t r__ = (call);                                                                \
        if (r__ != ncclSuccess) {                                                                 \
            (ctx)->last_error = std::string(#call) + ": " + rccl()->GetErrorString(r__);          \
            return CRP_ERR_COMM;                                                                  \
        }                                                                                         \
    } while (0)

namespace crp {

void comm_release(crp_ctx *ctx)
{
    crp_comm *c = ctx->comm;
    if (!c) return;
    if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
    (void)hipFree(c->d_counts);
    (void)hipFree(c->d_small);
    for (int s = 0; s < 2; ++s) {
        (void)hipFree(c->d_gpos[s]);
        (void)hipFree(c->d_gscore[s]);
        (void)hipFree(c->d_got[s]);
        (void)hipFree(c->d_gfeat[s]);
        (void)hipFree(c->d_lo16[s]);
        (void)hipFree(c->d_glo16[s]);
        (void)hipFree(c->d_bstart[s]);
        (void)hipFree(c->d_gbstart[s]);
    }
    delete c;
    ctx->comm = nullptr;
}

// crp_arena_destroy: the root's own rows of the last gather lived in this arena
void comm_forget_arena(crp_ctx *ctx, const crp_arena *a)
{
    if (ctx->comm && ctx->comm->garena == a) {
        ctx->comm->garena = nullptr;
        ctx->comm->have_gather = false;
    }
}

int comm_world(const crp_ctx *ctx) { return ctx->comm ? ctx->comm->world : 0; }
int comm_rank(const crp_ctx *ctx) { return ctx->comm ? ctx->comm->rank : 0; }
uint64_t comm_gather_bytes(const crp_ctx *ctx) { return ctx->comm ? ctx->comm->bytes_to_root : 0; }

// used by crp_offtarget.hip: in-place sum of n uint32 over all ranks (no-op without a communicator)
int comm_allreduce_u32(crp_ctx *ctx, uint32_t *d_buf, uint64_t n)
{
    crp_comm *c = ctx->comm;
    if (!c || c->world == 1) return CRP_OK;
    CRP_NCCL(ctx, rccl()->AllReduce(d_buf, d_buf, n, ncclUint32, ncclSum, c->comm, ctx->stream));
    return CRP_OK;
}

}  // namespace crp

extern "C" {

int crp_comm_unique_id(uint8_t id[CRP_COMM_ID_BYTES])
{
    static_assert(CRP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    if (!id) return CRP_ERR_INVALID;
    const Rccl *r = rccl();
    if (!r) return CRP_ERR_COMM;
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != ncclSuccess) return CRP_ERR_COMM;
    std::memcpy(id, u.internal, CRP_COMM_ID_BYTES);
    return CRP_OK;
}

int crp_comm_init(crp_ctx *ctx, const uint8_t id[CRP_COMM_ID_BYTES], int rank, int world)
{
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return CRP_ERR_INVALID;
    if (ctx->comm) return CRP_ERR_STATE;
    const Rccl *r = rccl();
    if (!r) {
        ctx->last_error = crp::rccl_load_error();
        return CRP_ERR_COMM;
    }
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    crp_comm *c = new (std::nothrow) crp_comm();
    if (!c) return CRP_ERR_NOMEM;
    c->rank = rank;
    c->world = world;
    ctx->comm = c;
    ncclUniqueId u;
    std::memcpy(u.internal, id, CRP_COMM_ID_BYTES);
    ncclResult_t st = r->CommInitRank(&c->comm, world, u, rank);
    if (st != ncclSuccess) {
        ctx->last_error = std::string("ncclCommInitRank: ") + r->GetErrorString(st);
        c->comm = nullptr;
        crp::comm_release(ctx);
        return CRP_ERR_COMM;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&c->d_counts), 5 * (size_t)world * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->d_small), 128 * sizeof(double));
    if (e != hipSuccess) {
        ctx->last_error = std::string("communicator scratch: ") + hipGetErrorString(e);
        crp::comm_release(ctx);
        return CRP_ERR_NOMEM;
    }
    c->counts.assign(2 * (size_t)world, 0);
    c->raw.assign(5 * (size_t)world, 0);
    if (const char *e = std::getenv("CRP_TEST_GATHER_FAIL")) c->test_fail = std::atoi(e);
    return CRP_OK;
}

int crp_comm_destroy(crp_ctx *ctx)
{
    if (!ctx) return CRP_ERR_INVALID;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    crp::comm_release(ctx);
    return CRP_OK;
}

int crp_comm_allreduce_f64(crp_ctx *ctx, double *values, int n, int op)
{
    if (!ctx || !values || n < 1 || n > 64 || (op != CRP_REDUCE_SUM && op != CRP_REDUCE_MAX)) return CRP_ERR_INVALID;
    crp_comm *c = ctx->comm;
    if (!c) return CRP_ERR_STATE;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    CRP_HIP(ctx, hipMemcpyAsync(c->d_small, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    CRP_NCCL(ctx, rccl()->AllReduce(c->d_small, c->d_small + 64, (size_t)n, ncclDouble, op == CRP_REDUCE_SUM ? ncclSum : ncclMax,
                                    c->comm, ctx->stream));
    CRP_HIP(ctx, hipMemcpyAsync(values, c->d_small + 64, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CRP_OK;
}

int crp_comm_barrier(crp_ctx *ctx)
{
    double one = 1.0;
    return crp_comm_allreduce_f64(ctx, &one, 1, CRP_REDUCE_SUM);
}

// Every rank's status word of one agreement round -> this rank's return value: its own error if it has
// one, CRP_ERR_PEER if only others do, CRP_OK if nobody does.  The same words are seen by every rank (they
// come out of an all-gather), so all ranks leave the call together and none enters the next collective.
static int agreed_status(crp_ctx *ctx, const crp_comm *c, const uint64_t *status, size_t stride, int own, const char *what)
{
    int first_bad = -1;
    for (int p = 0; p < c->world; ++p)
        if (status[(size_t)p * stride] != 0 && first_bad < 0) first_bad = p;
    if (first_bad < 0) return CRP_OK;
    if (own != CRP_OK) return own;
    ctx->last_error = std::string("crp_gather_hits abandoned on every rank: rank ") + std::to_string(first_bad) + " reported '" +
                      crp_strerror(-(int)status[(size_t)first_bad * stride]) + "' " + what;
    return CRP_ERR_PEER;
}

int crp_gather_hits(crp_ctx *ctx, crp_arena *a, int root, int flags, uint64_t *counts_all)
{
    crp::Range roctx_range("crp: gatherv (RCCL)");
    if (!ctx) return CRP_ERR_INVALID;
    crp_comm *c = ctx->comm;
    if (!c) return CRP_ERR_STATE;
    // (arguments that are the same on every rank by contract: a bad one fails everywhere alike)
    if (root < 0 || root >= c->world || (flags & ~(CRP_GATHER_OFFTARGET | CRP_GATHER_PRE | CRP_GATHER_FEATURES | CRP_GATHER_POS16)))
        return CRP_ERR_INVALID;
    const bool send_pre = (flags & CRP_GATHER_PRE) != 0;
    const bool with_ot = (flags & CRP_GATHER_OFFTARGET) != 0;
    const bool with_feat = (flags & CRP_GATHER_FEATURES) != 0;
    const bool pos16 = (flags & CRP_GATHER_POS16) != 0;
    // What can differ from rank to rank -- the state of this rank's arena, the root's allocation -- is never
    // answered with an early return: a rank that left here alone would leave its peers inside a collective
    // that cannot complete.  Each rank's status travels WITH its counts, and everyone acts on all of them.
    int local = CRP_OK;
    if (a && (a->ctx != ctx || !a->have_hits)) local = CRP_ERR_STATE;
    else if (send_pre && a && !a->have_pre) local = CRP_ERR_STATE;
    else if (with_ot && a && (!ctx->ot_solved || a->ot_epoch != ctx->ot_epoch)) local = CRP_ERR_STATE;
    else if (with_feat && a && !a->have_feat) local = CRP_ERR_STATE;
    if (c->test_fail == 2 && c->rank == c->world - 1) local = CRP_ERR_STATE;  // test hook (CRP_TEST_GATHER_FAIL)
    const Rccl *r = rccl();
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    c->have_gather = false;
    const bool ok_local = local == CRP_OK;
    const uint32_t my_buckets = ok_local && a ? crp::pos16_buckets_for(a->padded_words) : 0;
    const uint64_t mine[4] = {ok_local && a ? a->n_hits[0] : 0, ok_local && a ? a->n_hits[1] : 0, (uint64_t)(-local), my_buckets};
    const size_t W = (size_t)c->world;
    // 1. everyone learns every rank's two counts and its status
    CRP_HIP(ctx, hipMemcpyAsync(c->d_counts + 4 * c->rank, mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
    CRP_NCCL(ctx, r->AllGather(c->d_counts + 4 * c->rank, c->d_counts, 4, ncclUint64, c->comm, ctx->stream));
    CRP_HIP(ctx, hipMemcpyAsync(c->raw.data(), c->d_counts, 4 * W * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> buckets(W);
    for (size_t p = 0; p < W; ++p) {
        c->counts[2 * p] = c->raw[4 * p];
        c->counts[2 * p + 1] = c->raw[4 * p + 1];
        buckets[p] = (uint32_t)c->raw[4 * p + 3];
    }
    if (counts_all) std::memcpy(counts_all, c->counts.data(), 2 * W * sizeof(uint64_t));
    int rc = agreed_status(ctx, c, c->raw.data() + 2, 4, local, "before the exchange (its arena has no tables to send)");
    if (rc != CRP_OK) return rc;
    // 2. the root sizes its receive buffers; whether that worked is agreed on as well (one more word per rank)
    int alloc = CRP_OK;
    std::vector<uint64_t> soff[2], boff[2];  // CRP_GATHER_POS16: every peer's place in the root's staging area
    if (c->rank == root) {
        for (int s = 0; s < 2 && alloc == CRP_OK; ++s) {
            c->goff[s].assign(W, 0);
            soff[s].assign(W, 0);
            boff[s].assign(W, 0);
            uint64_t total = 0, lo_total = 0, b_total = 0;
            for (int p = 0; p < c->world; ++p) {
                c->goff[s][(size_t)p] = total;
                soff[s][(size_t)p] = lo_total;
                boff[s][(size_t)p] = b_total;
                if (p == root) continue;
                total += c->counts[2 * (size_t)p + s];
                lo_total += (c->counts[2 * (size_t)p + s] + 7) & ~(uint64_t)7;
                b_total += buckets[(size_t)p];
            }
            if (pos16 && lo_total) {
                alloc = crp::grow(ctx, reinterpret_cast<void **>(&c->d_glo16[s]), &c->glo16_cap[s], lo_total, sizeof(uint16_t));
                if (alloc == CRP_OK)
                    alloc = crp::grow(ctx, reinterpret_cast<void **>(&c->d_gbstart[s]), &c->gbstart_cap[s], b_total, sizeof(uint32_t));
                if (alloc != CRP_OK) break;
            }
            alloc = crp::grow(ctx, reinterpret_cast<void **>(&c->d_gpos[s]), &c->gpos_cap[s], total, sizeof(uint32_t));
            if (alloc == CRP_OK)
                alloc = crp::grow(ctx, reinterpret_cast<void **>(&c->d_gscore[s]), &c->gscore_cap[s], total, sizeof(double));
            if (alloc == CRP_OK && with_ot)
                alloc = crp::grow(ctx, reinterpret_cast<void **>(&c->d_got[s]), &c->got_cap[s], total, sizeof(uint4));
            if (alloc == CRP_OK && with_feat)
                alloc = crp::grow(ctx, reinterpret_cast<void **>(&c->d_gfeat[s]), &c->gfeat_cap[s], total, sizeof(uint32_t));
        }
    } else if (pos16 && a) {  // a peer's packed positions
        for (int s = 0; s < 2 && alloc == CRP_OK; ++s) {
            alloc = crp::grow(ctx, reinterpret_cast<void **>(&c->d_lo16[s]), &c->lo16_cap[s], (mine[s] + 7) & ~(uint64_t)7, sizeof(uint16_t));
            if (alloc == CRP_OK)
                alloc = crp::grow(ctx, reinterpret_cast<void **>(&c->d_bstart[s]), &c->bstart_cap[s], my_buckets, sizeof(uint32_t));
        }
    }
    if (c->rank == root) {
        if (c->test_fail == 1) {  // test hook: as if the receive buffers did not fit
            alloc = CRP_ERR_NOMEM;
            ctx->last_error = "gatherv receive buffers: out of memory (injected by CRP_TEST_GATHER_FAIL)";
        }
    }
    const uint64_t word = (uint64_t)(-alloc);
    uint64_t *d_status = c->d_counts + 4 * W;
    CRP_HIP(ctx, hipMemcpyAsync(d_status + c->rank, &word, sizeof word, hipMemcpyHostToDevice, ctx->stream));
    CRP_NCCL(ctx, r->AllGather(d_status + c->rank, d_status, 1, ncclUint64, c->comm, ctx->stream));
    CRP_HIP(ctx, hipMemcpyAsync(c->raw.data(), d_status, W * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rc = agreed_status(ctx, c, c->raw.data(), 1, alloc, "while sizing the root's receive buffers");
    if (rc != CRP_OK) return rc;
    // 3. the tables: grouped point-to-point, peers -> root
    crp::prof_begin(ctx, CRP_K_GATHER);
    uint64_t moved = 0;
    if (pos16 && c->rank != root && a)  // (on the stream the sends are queued on: packed before they read)
        for (int s = 0; s < 2; ++s) {
            CRP_HIP(ctx, crp::launch_pos16_buckets(ctx->stream, a->d_pos[s], mine[s], 0, mine[s], c->d_bstart[s], my_buckets));
            CRP_HIP(ctx, crp::launch_pos16_pack(ctx->stream, a->d_pos[s], mine[s], c->d_lo16[s]));
        }
    CRP_NCCL(ctx, r->GroupStart());
    ncclResult_t st = ncclSuccess;
    if (c->rank == root) {
        for (int p = 0; p < c->world && st == ncclSuccess; ++p) {
            if (p == root) continue;
            for (int s = 0; s < 2 && st == ncclSuccess; ++s) {
                const uint64_t n = c->counts[2 * (size_t)p + s];
                if (!n) continue;
                moved += n * ((pos16 ? 2 : 4) + 8 + (with_ot ? 16 : 0) + (with_feat ? 4 : 0)) + (pos16 ? 4ull * buckets[(size_t)p] : 0);
                if (pos16) {
                    st = r->Recv(c->d_glo16[s] + soff[s][(size_t)p], 2 * n, ncclUint8, p, c->comm, ctx->stream);
                    if (st == ncclSuccess)
                        st = r->Recv(c->d_gbstart[s] + boff[s][(size_t)p], buckets[(size_t)p], ncclUint32, p, c->comm, ctx->stream);
                } else {
                    st = r->Recv(c->d_gpos[s] + c->goff[s][(size_t)p], n, ncclUint32, p, c->comm, ctx->stream);
                }
                if (st == ncclSuccess)
                    st = r->Recv(c->d_gscore[s] + c->goff[s][(size_t)p], n, ncclDouble, p, c->comm, ctx->stream);
                if (st == ncclSuccess && with_ot)
                    st = r->Recv(c->d_got[s] + c->goff[s][(size_t)p], 4 * n, ncclUint32, p, c->comm, ctx->stream);
                if (st == ncclSuccess && with_feat)
                    st = r->Recv(c->d_gfeat[s] + c->goff[s][(size_t)p], n, ncclUint32, p, c->comm, ctx->stream);
            }
        }
    } else {
        for (int s = 0; s < 2 && st == ncclSuccess; ++s) {
            if (!mine[s]) continue;
            moved += mine[s] * ((pos16 ? 2 : 4) + 8 + (with_ot ? 16 : 0) + (with_feat ? 4 : 0)) + (pos16 ? 4ull * my_buckets : 0);
            if (pos16) {
                st = r->Send(c->d_lo16[s], 2 * mine[s], ncclUint8, root, c->comm, ctx->stream);
                if (st == ncclSuccess) st = r->Send(c->d_bstart[s], my_buckets, ncclUint32, root, c->comm, ctx->stream);
            } else {
                st = r->Send(a->d_pos[s], mine[s], ncclUint32, root, c->comm, ctx->stream);
            }
            if (st == ncclSuccess)
                st = r->Send(send_pre ? a->d_pre[s] : a->d_score[s], mine[s] - (c->test_fail == 3 && c->rank == c->world - 1 ? 1 : 0), ncclDouble,
                             root, c->comm, ctx->stream);
            if (st == ncclSuccess && with_ot) st = r->Send(a->d_ot_cnt[s], 4 * mine[s], ncclUint32, root, c->comm, ctx->stream);
            if (st == ncclSuccess && with_feat) st = r->Send(a->d_feat[s], mine[s], ncclUint32, root, c->comm, ctx->stream);
        }
    }
    const ncclResult_t st_end = r->GroupEnd();
    if (pos16 && c->rank == root && st == ncclSuccess && st_end == ncclSuccess)  // packed -> the root's position tables
        for (int p = 0; p < c->world; ++p)
            for (int s = 0; s < 2 && p != root; ++s) {
                const uint64_t n = c->counts[2 * (size_t)p + s];
                CRP_HIP(ctx, crp::launch_pos16_expand(ctx->stream, c->d_glo16[s] + soff[s][(size_t)p], n, c->d_gbstart[s] + boff[s][(size_t)p],
                                                      buckets[(size_t)p], crp::PieceMap{nullptr, nullptr, 0}, c->d_gpos[s] + c->goff[s][(size_t)p]));
            }
    crp::prof_end(ctx, CRP_K_GATHER);
    if (st != ncclSuccess || st_end != ncclSuccess) {
        // not agreed on: the communicator is in an unknown state and the peers may be inside the exchange --
        // the caller must take the whole run down (cli.EngineResident.gather: Group.abort)
        ctx->last_error = std::string("gatherv send/recv: ") + r->GetErrorString(st != ncclSuccess ? st : st_end);
        return CRP_ERR_COMM;
    }
    // the sends read the arena's tables: they must have left before the caller may scan or destroy it
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    crp::prof_collect(ctx, CRP_K_GATHER);
    c->bytes_to_root = moved;
    c->groot = root;
    c->gflags = flags;
    c->garena = a;
    c->have_gather = true;
    return CRP_OK;
}

int crp_gathered_fetch(crp_ctx *ctx, int rank, uint32_t *pos_plus, double *score_plus, uint32_t *ot_plus,
                       uint32_t *pos_minus, double *score_minus, uint32_t *ot_minus)
{
    if (!ctx) return CRP_ERR_INVALID;
    crp_comm *c = ctx->comm;
    if (!c || !c->have_gather || c->rank != c->groot) return CRP_ERR_STATE;
    if (rank < 0 || rank >= c->world) return CRP_ERR_INVALID;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t *hp[2] = {pos_plus, pos_minus};
    double *hs[2] = {score_plus, score_minus};
    uint32_t *ho[2] = {ot_plus, ot_minus};
    if ((ot_plus || ot_minus) && !(c->gflags & CRP_GATHER_OFFTARGET)) return CRP_ERR_STATE;
    for (int s = 0; s < 2; ++s) {
        const uint64_t n = c->counts[2 * (size_t)rank + s];
        if (!n) continue;
        const uint32_t *dp;
        const double *ds;
        const uint4 *dt = nullptr;
        if (rank == c->groot) {  // root's own rows never moved
            if (!c->garena || !c->garena->have_hits) return CRP_ERR_STATE;
            dp = c->garena->d_pos[s];
            ds = (c->gflags & CRP_GATHER_PRE) ? c->garena->d_pre[s] : c->garena->d_score[s];
            dt = c->garena->d_ot_cnt[s];
        } else {
            dp = c->d_gpos[s] + c->goff[s][(size_t)rank];
            ds = c->d_gscore[s] + c->goff[s][(size_t)rank];
            if (c->gflags & CRP_GATHER_OFFTARGET) dt = c->d_got[s] + c->goff[s][(size_t)rank];
        }
        int rc = CRP_OK;
        if (ho[s]) rc = crp::staged_d2h(ctx, ho[s], dt, n * sizeof(uint4));
        if (rc == CRP_OK && hp[s]) rc = crp::staged_d2h(ctx, hp[s], dp, n * sizeof(uint32_t));
        if (rc == CRP_OK && hs[s]) rc = crp::staged_d2h(ctx, hs[s], ds, n * sizeof(double));
        if (rc != CRP_OK) return rc;
    }
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CRP_OK;
}

int crp_gathered_fetch_features(crp_ctx *ctx, int rank, uint32_t *feat_plus, uint32_t *feat_minus)
{
    if (!ctx) return CRP_ERR_INVALID;
    crp_comm *c = ctx->comm;
    if (!c || !c->have_gather || c->rank != c->groot || !(c->gflags & CRP_GATHER_FEATURES)) return CRP_ERR_STATE;
    if (rank < 0 || rank >= c->world) return CRP_ERR_INVALID;
    CRP_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t *hf[2] = {feat_plus, feat_minus};
    for (int s = 0; s < 2; ++s) {
        const uint64_t n = c->counts[2 * (size_t)rank + s];
        if (!n || !hf[s]) continue;
        const uint32_t *df;
        if (rank == c->groot) {  // root's own rows never moved
            if (!c->garena || !c->garena->have_feat) return CRP_ERR_STATE;
            df = c->garena->d_feat[s];
        } else {
            df = c->d_gfeat[s] + c->goff[s][(size_t)rank];
        }
        const int rc = crp::staged_d2h(ctx, hf[s], df, n * sizeof(uint32_t));
        if (rc != CRP_OK) return rc;
    }
    CRP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CRP_OK;
}

}  // extern "C"
