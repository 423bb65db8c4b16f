// fake_rccl.cpp -- a LOOP-BACK TEST DOUBLE for librccl.so.1.  TEST INFRASTRUCTURE: never shipped, never linked; built by
// tests/fake_rccl.py into a directory that a child test process puts first on its LD_LIBRARY_PATH, so that the product's
// own dlopen("librccl.so.1") (cropsr_amd/csrc/crp_comm.cpp) finds this file instead of ROCm's.  The product is untouched.
//
// Why: the path's one exchange (the gatherv of the hit tables, CROPSR.py:409's loop spread over GPUs) and the off-target
// histogram all-reduce are written against RCCL, and the one-GPU boxes this build is developed on cannot run RCCL between
// two ranks (RCCL refuses two ranks on one device).  This double accepts duplicate devices and moves the bytes itself, so the
// N > 1 branches of crp_node.cpp (ncclCommInitAll, one grouped send/recv over N communicators, grouped all-reduce) and of
// crp_comm.cpp (ncclCommInitRank, all-gather + grouped send/recv between processes) execute -- and it is STRICTER than the
// real library: it checks the protocol.
//
//   in one process   (ncclCommInitAll)   a group's sends and receives are matched per (source, destination) pair in posting
//                                        order; a send without a receive, a receive without a send or a byte-count mismatch
//                                        fails ncclGroupEnd with ncclInvalidUsage (the real library would hang or corrupt);
//                                        matched pairs become hipMemcpy(Peer)Async on the receiver's stream behind an event
//                                        on the sender's, and the sender's stream waits for the copy (send completion).
//                                        Point-to-point outside a group is refused (on one host thread it deadlocks in RCCL).
//                                        AllReduce / AllGather must be posted on every rank of the communicator clique,
//                                        same count / type / operator, inside one group.
//   across processes (ncclCommInitRank)  a mailbox of files under $FAKE_RCCL_SHM_DIR (default /dev/shm) keyed by the unique
//                                        id; data bounces through the host.  The same checks, made by the receiver.
//
// Knobs (environment): FAKE_RCCL_HANG=init  communicator creation never returns (bounded by FAKE_RCCL_HANG_MAX_S, 120);
//                      FAKE_RCCL_HANG=group every stream of a group is held by a kernel that spins until ncclCommAbort (or the
//                                           same bound) before its copies run: "a collective that never completes";
//                      FAKE_RCCL_TIMEOUT_S  how long a receiver / a bootstrap waits for its peer processes (60);
//                      FAKE_RCCL_STATS_FILE every communicator appends one JSON line when it is destroyed.
// fake_rccl_stats() hands the same counters to a test in the same process.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

struct World;

struct ncclComm {
    World *world = nullptr;
    int rank = 0, device = 0;
    hipEvent_t ev_send = nullptr, ev_recv = nullptr;
    // across processes: message numbers per peer, collective number, files this rank published
    std::vector<uint64_t> send_seq, recv_seq;
    uint64_t coll_seq = 0;
    std::string prev_coll_file;
    bool live = true;
};

struct World {
    bool local = true;
    int n = 0;
    std::vector<ncclComm *> comms;  // local cliques: every rank
    int alive = 0;
    // "never completes": a flag in pinned host memory the spin kernel watches
    uint32_t *h_release = nullptr;
    // local reductions
    int red_device = 0;
    hipStream_t red_stream = nullptr;
    hipEvent_t red_done = nullptr;
    void *tmp[2] = {nullptr, nullptr};
    size_t tmp_cap = 0;
    std::string prefix;  // across processes
};

namespace {

enum Kind { SEND, RECV, ALLREDUCE, ALLGATHER };

struct Op {
    Kind kind;
    ncclComm *comm;
    const void *send;
    void *recv;
    size_t count;
    ncclDataType_t type;
    ncclRedOp_t op;
    int peer;
    hipStream_t stream;
};

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local std::string t_detail;  // what the last ncclInvalidUsage / ncclSystemError of this thread was about

std::mutex g_mutex;
std::atomic<uint64_t> g_pairs{0}, g_p2p_bytes{0}, g_colls{0}, g_groups{0}, g_mismatches{0}, g_inits{0}, g_hangs{0}, g_aborts{0};

size_t type_size(ncclDataType_t t)
{
    switch (t) {
        case ncclInt8: case ncclUint8: case ncclFloat8e4m3: case ncclFloat8e5m2: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

double env_seconds(const char *name, double dflt)
{
    const char *e = std::getenv(name);
    return e && *e ? std::atof(e) : dflt;
}

bool hang_mode(const char *what)
{
    const char *e = std::getenv("FAKE_RCCL_HANG");
    return e && !std::strcmp(e, what);
}

ncclResult_t fail(ncclResult_t rc, const std::string &detail)
{
    t_detail = detail;
    if (rc == ncclInvalidUsage) g_mismatches++;
    if (std::getenv("FAKE_RCCL_VERBOSE")) std::fprintf(stderr, "fake_rccl: %s\n", detail.c_str());
    return rc;
}

#define F_HIP(call)                                                                                       \
    do {                                                                                                  \
        hipError_t e__ = (call);                                                                          \
        if (e__ != hipSuccess) return fail(ncclUnhandledCudaError, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

struct DeviceGuard {  // like the real library, leave the caller's current device alone
    int saved = -1;
    DeviceGuard() { (void)hipGetDevice(&saved); }
    ~DeviceGuard()
    {
        if (saved >= 0) (void)hipSetDevice(saved);
    }
};

// ---- device side: element-wise reduction (local cliques) and the spin that makes a group "never complete"
template <class T>
__global__ void reduce_kernel(T *acc, const T *in, size_t n, int op)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T a = acc[i], b = in[i];
        acc[i] = op == ncclSum ? (T)(a + b) : op == ncclProd ? (T)(a * b) : op == ncclMax ? (a > b ? a : b) : (a < b ? a : b);
    }
}

// one wave, asleep most of the time; EVERY wave reaches its exit: the host sets *release (ncclCommAbort, ncclCommDestroy) or
// the wall-clock bound runs out
__global__ void hold_kernel(const volatile uint32_t *release, unsigned long long max_ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (!*release && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(127);
}

template <class T>
hipError_t launch_reduce(hipStream_t s, void *acc, const void *in, size_t n, ncclRedOp_t op)
{
    const unsigned blocks = (unsigned)std::min<size_t>(4096, (n + 255) / 256);
    hipLaunchKernelGGL(reduce_kernel<T>, dim3(blocks ? blocks : 1), dim3(256), 0, s, (T *)acc, (const T *)in, n, (int)op);
    return hipGetLastError();
}

hipError_t device_reduce(hipStream_t s, void *acc, const void *in, size_t n, ncclDataType_t t, ncclRedOp_t op)
{
    switch (t) {
        case ncclInt32: return launch_reduce<int32_t>(s, acc, in, n, op);
        case ncclUint32: return launch_reduce<uint32_t>(s, acc, in, n, op);
        case ncclInt64: return launch_reduce<int64_t>(s, acc, in, n, op);
        case ncclUint64: return launch_reduce<uint64_t>(s, acc, in, n, op);
        case ncclFloat32: return launch_reduce<float>(s, acc, in, n, op);
        case ncclFloat64: return launch_reduce<double>(s, acc, in, n, op);
        default: return hipErrorInvalidValue;
    }
}

template <class T>
void host_reduce_t(void *acc, const void *in, size_t n, ncclRedOp_t op)
{
    T *a = (T *)acc;
    const T *b = (const T *)in;
    for (size_t i = 0; i < n; ++i) a[i] = op == ncclSum ? (T)(a[i] + b[i]) : op == ncclProd ? (T)(a[i] * b[i]) : op == ncclMax ? (a[i] > b[i] ? a[i] : b[i]) : (a[i] < b[i] ? a[i] : b[i]);
}

bool host_reduce(void *acc, const void *in, size_t n, ncclDataType_t t, ncclRedOp_t op)
{
    switch (t) {
        case ncclInt32: host_reduce_t<int32_t>(acc, in, n, op); return true;
        case ncclUint32: host_reduce_t<uint32_t>(acc, in, n, op); return true;
        case ncclInt64: host_reduce_t<int64_t>(acc, in, n, op); return true;
        case ncclUint64: host_reduce_t<uint64_t>(acc, in, n, op); return true;
        case ncclFloat32: host_reduce_t<float>(acc, in, n, op); return true;
        case ncclFloat64: host_reduce_t<double>(acc, in, n, op); return true;
        default: return false;
    }
}

bool reducible(ncclDataType_t t, ncclRedOp_t op)
{
    return (op == ncclSum || op == ncclProd || op == ncclMax || op == ncclMin) &&
           (t == ncclInt32 || t == ncclUint32 || t == ncclInt64 || t == ncclUint64 || t == ncclFloat32 || t == ncclFloat64);
}

// ------------------------------------------------------------------ one process: the clique of ncclCommInitAll
ncclResult_t hold_streams(World *w, const std::vector<Op> &ops)
{
    // FAKE_RCCL_HANG=group: every stream of this group first runs the spin; whatever the group enqueues stays behind it
    std::vector<std::pair<int, hipStream_t>> seen;
    const unsigned long long ticks = (unsigned long long)(env_seconds("FAKE_RCCL_HANG_MAX_S", 120) * 1e8);
    for (const Op &o : ops) {
        if (o.comm->world != w) continue;
        bool dup = false;
        for (auto &p : seen) dup |= p.first == o.comm->device && p.second == o.stream;
        if (dup) continue;
        seen.emplace_back(o.comm->device, o.stream);
        F_HIP(hipSetDevice(o.comm->device));
        hipLaunchKernelGGL(hold_kernel, dim3(1), dim3(1), 0, o.stream, w->h_release, ticks);
        F_HIP(hipGetLastError());
    }
    g_hangs++;
    return ncclSuccess;
}

ncclResult_t copy_between(ncclComm *from, const void *src, ncclComm *to, void *dst, size_t bytes, hipStream_t on)
{
    if (!bytes || src == dst) return ncclSuccess;
    F_HIP(hipSetDevice(to->device));
    if (from->device == to->device) F_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, on));
    else F_HIP(hipMemcpyPeerAsync(dst, to->device, src, from->device, bytes, on));
    return ncclSuccess;
}

ncclResult_t run_local(World *w, const std::vector<Op> &all)
{
    std::vector<const Op *> ops;
    for (const Op &o : all)
        if (o.comm->world == w) ops.push_back(&o);
    if (ops.empty()) return ncclSuccess;
    // ---- check first, move nothing unless the whole group is consistent
    std::map<std::pair<int, int>, std::vector<const Op *>> sends, recvs;  // (source, destination) -> posting order
    std::vector<std::vector<const Op *>> colls((size_t)w->n);
    for (const Op *o : ops) {
        if (o->kind == SEND) sends[{o->comm->rank, o->peer}].push_back(o);
        else if (o->kind == RECV) recvs[{o->peer, o->comm->rank}].push_back(o);
        else colls[(size_t)o->comm->rank].push_back(o);
    }
    for (auto &kv : sends) {
        auto it = recvs.find(kv.first);
        const size_t nr = it == recvs.end() ? 0 : it->second.size();
        if (nr != kv.second.size())
            return fail(ncclInvalidUsage, "rank " + std::to_string(kv.first.first) + " posted " + std::to_string(kv.second.size()) +
                                              " send(s) to rank " + std::to_string(kv.first.second) + " which posted " + std::to_string(nr) +
                                              " receive(s) from it in this group");
        for (size_t k = 0; k < nr; ++k) {
            const size_t sb = kv.second[k]->count * type_size(kv.second[k]->type), rb = it->second[k]->count * type_size(it->second[k]->type);
            if (sb != rb)
                return fail(ncclInvalidUsage, "send #" + std::to_string(k) + " of rank " + std::to_string(kv.first.first) + " to rank " +
                                                  std::to_string(kv.first.second) + " has " + std::to_string(sb) + " bytes, the matching receive expects " +
                                                  std::to_string(rb));
        }
    }
    for (auto &kv : recvs)
        if (!sends.count(kv.first))
            return fail(ncclInvalidUsage, "rank " + std::to_string(kv.first.second) + " posted " + std::to_string(kv.second.size()) +
                                              " receive(s) from rank " + std::to_string(kv.first.first) + " which sends nothing in this group");
    size_t n_coll = 0;
    for (int r = 0; r < w->n; ++r) n_coll = std::max(n_coll, colls[(size_t)r].size());
    for (int r = 0; r < w->n && n_coll; ++r) {
        if (colls[(size_t)r].size() != n_coll)
            return fail(ncclInvalidUsage, "rank " + std::to_string(r) + " posted " + std::to_string(colls[(size_t)r].size()) + " collective(s) in a group in which another rank posted " +
                                              std::to_string(n_coll) + ": every rank of the clique must take part");
        for (size_t k = 0; k < n_coll; ++k) {
            const Op *a = colls[0][k], *b = colls[(size_t)r][k];
            if (a->kind != b->kind || a->count != b->count || a->type != b->type || (a->kind == ALLREDUCE && a->op != b->op))
                return fail(ncclInvalidUsage, "collective #" + std::to_string(k) + " differs between rank 0 and rank " + std::to_string(r) + " (kind / count / type / operator)");
            if (a->kind == ALLREDUCE && !reducible(a->type, a->op)) return fail(ncclInvalidArgument, "all-reduce: type / operator not supported by the test double");
        }
    }
    if (hang_mode("group")) {
        ncclResult_t rc = hold_streams(w, all);
        if (rc != ncclSuccess) return rc;
    }
    // ---- point to point: copies on the receiver's stream behind the sender's work; the sender's stream waits for them
    std::vector<char> sent((size_t)w->n, 0), received((size_t)w->n, 0);
    for (auto &kv : sends) sent[(size_t)kv.first.first] = 1;
    auto record_send_events = [&](const std::vector<const Op *> &list) -> ncclResult_t {
        std::vector<char> done((size_t)w->n, 0);
        for (const Op *o : list) {
            ncclComm *c = o->comm;
            if (done[(size_t)c->rank]) continue;
            done[(size_t)c->rank] = 1;
            F_HIP(hipSetDevice(c->device));
            F_HIP(hipEventRecord(c->ev_send, o->stream));
        }
        return ncclSuccess;
    };
    {
        std::vector<const Op *> senders;
        for (auto &kv : sends) senders.push_back(kv.second[0]);
        ncclResult_t rc = record_send_events(senders);
        if (rc != ncclSuccess) return rc;
    }
    std::map<int, hipStream_t> recv_stream, send_stream;
    for (auto &kv : sends) {
        const auto &sl = kv.second;
        const auto &rl = recvs[kv.first];
        for (size_t k = 0; k < sl.size(); ++k) {
            ncclComm *from = sl[k]->comm, *to = rl[k]->comm;
            F_HIP(hipSetDevice(to->device));
            F_HIP(hipStreamWaitEvent(rl[k]->stream, from->ev_send, 0));
            ncclResult_t rc = copy_between(from, sl[k]->send, to, rl[k]->recv, sl[k]->count * type_size(sl[k]->type), rl[k]->stream);
            if (rc != ncclSuccess) return rc;
            received[(size_t)to->rank] = 1;
            recv_stream[to->rank] = rl[k]->stream;
            send_stream[from->rank] = sl[k]->stream;
            g_pairs++;
            g_p2p_bytes += sl[k]->count * type_size(sl[k]->type);
        }
    }
    for (auto &kv : recv_stream) {
        ncclComm *c = w->comms[(size_t)kv.first];
        F_HIP(hipSetDevice(c->device));
        F_HIP(hipEventRecord(c->ev_recv, kv.second));
    }
    for (auto &kv : sends) {  // a send is complete when its bytes have left
        ncclComm *from = w->comms[(size_t)kv.first.first], *to = w->comms[(size_t)kv.first.second];
        F_HIP(hipSetDevice(from->device));
        F_HIP(hipStreamWaitEvent(send_stream[from->rank], to->ev_recv, 0));
    }
    // ---- collectives, one after the other
    for (size_t k = 0; k < n_coll; ++k) {
        const Op *first = colls[0][k];
        const size_t bytes = first->count * type_size(first->type);
        std::vector<const Op *> row;
        for (int r = 0; r < w->n; ++r) row.push_back(colls[(size_t)r][k]);
        ncclResult_t rc = record_send_events(row);
        if (rc != ncclSuccess) return rc;
        if (first->kind == ALLGATHER) {
            for (int d = 0; d < w->n; ++d) {
                const Op *od = row[(size_t)d];
                F_HIP(hipSetDevice(od->comm->device));
                for (int r = 0; r < w->n; ++r) {
                    F_HIP(hipStreamWaitEvent(od->stream, row[(size_t)r]->comm->ev_send, 0));
                    rc = copy_between(row[(size_t)r]->comm, row[(size_t)r]->send, od->comm, (char *)od->recv + (size_t)r * bytes, bytes, od->stream);
                    if (rc != ncclSuccess) return rc;
                }
                F_HIP(hipEventRecord(od->comm->ev_recv, od->stream));
            }
            for (int r = 0; r < w->n; ++r) {  // nobody's send buffer is free before everybody has read it
                F_HIP(hipSetDevice(row[(size_t)r]->comm->device));
                for (int d = 0; d < w->n; ++d) F_HIP(hipStreamWaitEvent(row[(size_t)r]->stream, row[(size_t)d]->comm->ev_recv, 0));
            }
        } else {
            F_HIP(hipSetDevice(w->red_device));
            if (w->tmp_cap < bytes) {
                F_HIP(hipStreamSynchronize(w->red_stream));
                for (int j = 0; j < 2; ++j) {
                    (void)hipFree(w->tmp[j]);
                    w->tmp[j] = nullptr;
                    F_HIP(hipMalloc(&w->tmp[j], bytes));
                }
                w->tmp_cap = bytes;
            }
            ncclComm root_like;  // (the scratch lives on red_device)
            root_like.device = w->red_device;
            for (int r = 0; r < w->n; ++r) {
                F_HIP(hipStreamWaitEvent(w->red_stream, row[(size_t)r]->comm->ev_send, 0));
                rc = copy_between(row[(size_t)r]->comm, row[(size_t)r]->send, &root_like, w->tmp[r ? 1 : 0], bytes, w->red_stream);
                if (rc != ncclSuccess) return rc;
                if (r) F_HIP(device_reduce(w->red_stream, w->tmp[0], w->tmp[1], first->count, first->type, first->op));
            }
            F_HIP(hipEventRecord(w->red_done, w->red_stream));
            for (int r = 0; r < w->n; ++r) {
                const Op *o = row[(size_t)r];
                F_HIP(hipSetDevice(o->comm->device));
                F_HIP(hipStreamWaitEvent(o->stream, w->red_done, 0));
                rc = copy_between(&root_like, w->tmp[0], o->comm, o->recv, bytes, o->stream);
                if (rc != ncclSuccess) return rc;
                F_HIP(hipEventRecord(o->comm->ev_recv, o->stream));
            }
            F_HIP(hipSetDevice(w->red_device));
            for (int r = 0; r < w->n; ++r) F_HIP(hipStreamWaitEvent(w->red_stream, row[(size_t)r]->comm->ev_recv, 0));  // the scratch is free again
        }
        g_colls++;
    }
    return ncclSuccess;
}

// ------------------------------------------------------------------ across processes: a mailbox of files
struct Header {
    uint64_t magic, kind, count, type, op, bytes;
};
constexpr uint64_t MAGIC = 0x46414b455243434cull;  // "FAKERCCL"

std::string shm_dir()
{
    const char *e = std::getenv("FAKE_RCCL_SHM_DIR");
    return e && *e ? e : "/dev/shm";
}

bool exists(const std::string &p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0;
}

// header + payload written under a temporary name, then renamed: a reader never sees half a message
ncclResult_t publish(const std::string &name, const Header &h, const void *d_src, hipStream_t stream)
{
    const std::string tmp = name + ".tmp" + std::to_string((long)getpid());
    const int fd = ::open(tmp.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd < 0) return fail(ncclSystemError, "open " + tmp + ": " + std::strerror(errno));
    const size_t total = sizeof(Header) + h.bytes;
    if (::ftruncate(fd, (off_t)total) != 0) {
        ::close(fd);
        return fail(ncclSystemError, "ftruncate " + tmp);
    }
    char *m = (char *)::mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) return fail(ncclSystemError, "mmap " + tmp);
    std::memcpy(m, &h, sizeof h);
    hipError_t e = hipSuccess;
    if (h.bytes) e = hipMemcpyAsync(m + sizeof h, d_src, h.bytes, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    ::munmap(m, total);
    if (e != hipSuccess) return fail(ncclUnhandledCudaError, std::string("D2H into the mailbox: ") + hipGetErrorString(e));
    if (::rename(tmp.c_str(), name.c_str()) != 0) return fail(ncclSystemError, "rename " + tmp);
    return ncclSuccess;
}

// waits for `name`, maps it (the caller unmaps)
ncclResult_t await(const std::string &name, char **map, size_t *total, const char *what)
{
    const double limit = env_seconds("FAKE_RCCL_TIMEOUT_S", 60);
    const auto t0 = std::chrono::steady_clock::now();
    while (!exists(name)) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
            return fail(ncclRemoteError, std::string("no ") + what + " after " + std::to_string((int)limit) + " s: " + name);
        ::usleep(200);
    }
    const int fd = ::open(name.c_str(), O_RDONLY);
    if (fd < 0) return fail(ncclSystemError, "open " + name);
    struct stat st;
    if (::fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(Header)) {
        ::close(fd);
        return fail(ncclSystemError, "short mailbox file " + name);
    }
    *total = (size_t)st.st_size;
    *map = (char *)::mmap(nullptr, *total, PROT_READ, MAP_SHARED, fd, 0);
    ::close(fd);
    if (*map == MAP_FAILED) return fail(ncclSystemError, "mmap " + name);
    return ncclSuccess;
}

std::string p2p_name(const World *w, int src, int dst, uint64_t seq)
{
    return w->prefix + ".p2p." + std::to_string(src) + "." + std::to_string(dst) + "." + std::to_string(seq);
}

std::string coll_name(const World *w, uint64_t seq, int rank) { return w->prefix + ".coll." + std::to_string(seq) + "." + std::to_string(rank); }

ncclResult_t run_remote(World *w, const std::vector<Op> &all)
{
    // every send first (a mailbox never blocks the sender), then the receives, then the collectives in posting order
    for (const Op &o : all) {
        if (o.comm->world != w || o.kind != SEND) continue;
        ncclComm *c = o.comm;
        DeviceGuard g;
        F_HIP(hipSetDevice(c->device));
        Header h{MAGIC, SEND, o.count, (uint64_t)o.type, 0, o.count * type_size(o.type)};
        ncclResult_t rc = publish(p2p_name(w, c->rank, o.peer, c->send_seq[(size_t)o.peer]++), h, o.send, o.stream);
        if (rc != ncclSuccess) return rc;
    }
    for (const Op &o : all) {
        if (o.comm->world != w || o.kind != RECV) continue;
        ncclComm *c = o.comm;
        F_HIP(hipSetDevice(c->device));
        const std::string name = p2p_name(w, o.peer, c->rank, c->recv_seq[(size_t)o.peer]++);
        char *m = nullptr;
        size_t total = 0;
        ncclResult_t rc = await(name, &m, &total, "message from the peer");
        if (rc != ncclSuccess) return rc;
        Header h;
        std::memcpy(&h, m, sizeof h);
        const size_t want = o.count * type_size(o.type);
        hipError_t e = hipSuccess;
        if (h.magic != MAGIC || h.bytes != want || total != sizeof h + h.bytes) {
            ::munmap(m, total);
            ::unlink(name.c_str());
            return fail(ncclInvalidUsage, "message #" + std::to_string(c->recv_seq[(size_t)o.peer] - 1) + " from rank " + std::to_string(o.peer) + " to rank " +
                                              std::to_string(c->rank) + " has " + std::to_string((unsigned long long)h.bytes) + " bytes, the receive expects " + std::to_string(want));
        }
        if (want) e = hipMemcpyAsync(o.recv, m + sizeof h, want, hipMemcpyHostToDevice, o.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(o.stream);
        ::munmap(m, total);
        ::unlink(name.c_str());
        if (e != hipSuccess) return fail(ncclUnhandledCudaError, std::string("H2D out of the mailbox: ") + hipGetErrorString(e));
        g_pairs++;
        g_p2p_bytes += want;
    }
    for (const Op &o : all) {
        if (o.comm->world != w || (o.kind != ALLREDUCE && o.kind != ALLGATHER)) continue;
        ncclComm *c = o.comm;
        F_HIP(hipSetDevice(c->device));
        if (o.kind == ALLREDUCE && !reducible(o.type, o.op)) return fail(ncclInvalidArgument, "all-reduce: type / operator not supported by the test double");
        const size_t bytes = o.count * type_size(o.type);
        const uint64_t seq = c->coll_seq++;
        Header h{MAGIC, (uint64_t)o.kind, o.count, (uint64_t)o.type, (uint64_t)o.op, bytes};
        const std::string mine = coll_name(w, seq, c->rank);
        ncclResult_t rc = publish(mine, h, o.send, o.stream);
        if (rc != ncclSuccess) return rc;
        std::vector<char> out(o.kind == ALLGATHER ? bytes * (size_t)w->n : bytes);
        for (int r = 0; r < w->n; ++r) {
            char *m = nullptr;
            size_t total = 0;
            rc = await(coll_name(w, seq, r), &m, &total, "contribution to a collective");
            if (rc != ncclSuccess) return rc;
            Header hr;
            std::memcpy(&hr, m, sizeof hr);
            if (hr.magic != MAGIC || hr.kind != h.kind || hr.count != h.count || hr.type != h.type || hr.op != h.op || total != sizeof hr + bytes) {
                ::munmap(m, total);
                return fail(ncclInvalidUsage, "collective #" + std::to_string((unsigned long long)seq) + ": rank " + std::to_string(r) + " and rank " + std::to_string(c->rank) +
                                                  " disagree on kind / count / type / operator");
            }
            if (o.kind == ALLGATHER) std::memcpy(out.data() + (size_t)r * bytes, m + sizeof hr, bytes);
            else if (r == 0) std::memcpy(out.data(), m + sizeof hr, bytes);
            else host_reduce(out.data(), m + sizeof hr, o.count, o.type, o.op);
            ::munmap(m, total);
        }
        if (!out.empty()) {
            F_HIP(hipMemcpyAsync(o.recv, out.data(), out.size(), hipMemcpyHostToDevice, o.stream));
            F_HIP(hipStreamSynchronize(o.stream));
        }
        // every rank has published collective #seq, so every rank has finished reading #seq - 1: my file of it can go
        if (!c->prev_coll_file.empty()) ::unlink(c->prev_coll_file.c_str());
        c->prev_coll_file = mine;
        g_colls++;
    }
    return ncclSuccess;
}

ncclResult_t flush()
{
    std::vector<Op> ops;
    ops.swap(t_ops);
    if (ops.empty()) return ncclSuccess;
    g_groups++;
    DeviceGuard g;
    std::vector<World *> worlds;
    for (const Op &o : ops) {
        bool seen = false;
        for (World *w : worlds) seen |= w == o.comm->world;
        if (!seen) worlds.push_back(o.comm->world);
    }
    for (World *w : worlds) {
        ncclResult_t rc;
        if (w->local) {
            std::lock_guard<std::mutex> lk(g_mutex);
            rc = run_local(w, ops);
        } else {
            rc = run_remote(w, ops);
        }
        if (rc != ncclSuccess) return rc;
    }
    return ncclSuccess;
}

ncclResult_t post(const Op &o)
{
    if (!o.comm || !o.comm->live) return fail(ncclInvalidArgument, "operation on a destroyed or null communicator");
    if (!type_size(o.type)) return fail(ncclInvalidArgument, "unknown data type");
    if ((o.kind == SEND || o.kind == RECV) && (o.peer < 0 || o.peer >= o.comm->world->n)) return fail(ncclInvalidArgument, "peer out of range");
    if ((o.kind == SEND || o.kind == RECV) && t_depth == 0 && o.comm->world->local && o.comm->world->n > 1)
        return fail(ncclInvalidUsage, "point-to-point call outside ncclGroupStart/End on a clique driven by one process: with the real library this blocks for ever");
    t_ops.push_back(o);
    return t_depth == 0 ? flush() : ncclSuccess;
}

void write_stats(const ncclComm *c)
{
    const char *path = std::getenv("FAKE_RCCL_STATS_FILE");
    if (!path || !*path) return;
    char line[512];
    const int n = std::snprintf(line, sizeof line,
                                "{\"pid\": %ld, \"rank\": %d, \"world\": %d, \"in_process\": %s, \"pairs\": %llu, \"p2p_bytes\": %llu, \"collectives\": %llu, "
                                "\"groups\": %llu, \"mismatches\": %llu, \"inits\": %llu, \"hangs\": %llu, \"aborts\": %llu}\n",
                                (long)getpid(), c->rank, c->world->n, c->world->local ? "true" : "false", (unsigned long long)g_pairs.load(),
                                (unsigned long long)g_p2p_bytes.load(), (unsigned long long)g_colls.load(), (unsigned long long)g_groups.load(),
                                (unsigned long long)g_mismatches.load(), (unsigned long long)g_inits.load(), (unsigned long long)g_hangs.load(),
                                (unsigned long long)g_aborts.load());
    const int fd = ::open(path, O_CREAT | O_WRONLY | O_APPEND, 0600);
    if (fd >= 0) {
        (void)!::write(fd, line, (size_t)n);
        ::close(fd);
    }
}

ncclResult_t hang_in_init()
{
    const double limit = env_seconds("FAKE_RCCL_HANG_MAX_S", 120);
    const auto t0 = std::chrono::steady_clock::now();
    g_hangs++;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < limit) ::usleep(100000);
    return fail(ncclSystemError, "FAKE_RCCL_HANG=init: the bootstrap was held for FAKE_RCCL_HANG_MAX_S");
}

ncclResult_t comm_events(ncclComm *c)
{
    F_HIP(hipSetDevice(c->device));
    F_HIP(hipEventCreateWithFlags(&c->ev_send, hipEventDisableTiming));
    F_HIP(hipEventCreateWithFlags(&c->ev_recv, hipEventDisableTiming));
    return ncclSuccess;
}

void release_world(World *w)
{
    if (w->h_release) __atomic_store_n(w->h_release, 1u, __ATOMIC_SEQ_CST);
}

}  // namespace

extern "C" {

const char *ncclGetErrorString(ncclResult_t r)
{
    static thread_local std::string text;
    const char *base = r == ncclSuccess ? "no error" : r == ncclUnhandledCudaError ? "unhandled hip error" : r == ncclSystemError ? "unhandled system error" :
                       r == ncclInternalError ? "internal error" : r == ncclInvalidArgument ? "invalid argument" : r == ncclInvalidUsage ? "invalid usage" :
                       r == ncclRemoteError ? "remote process exited or there was a network error" : "unknown result code";
    text = base;
    if (r != ncclSuccess && !t_detail.empty()) text += " [fake_rccl: " + t_detail + "]";
    return text.c_str();
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return fail(ncclInvalidArgument, "null unique id");
    std::memset(id->internal, 0, sizeof id->internal);
    const int fd = ::open("/dev/urandom", O_RDONLY);
    if (fd < 0 || ::read(fd, id->internal, 16) != 16) {
        if (fd >= 0) ::close(fd);
        return fail(ncclSystemError, "/dev/urandom");
    }
    ::close(fd);
    std::memcpy(id->internal + 16, "fake_rccl", 9);
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    if (!comms || ndev < 1) return fail(ncclInvalidArgument, "ncclCommInitAll arguments");
    if (hang_mode("init")) return hang_in_init();
    DeviceGuard g;
    World *w = new World();  // (worlds are never freed: a test double's few bytes)
    w->local = true;
    w->n = ndev;
    w->alive = ndev;
    for (int k = 0; k < ndev; ++k) {
        ncclComm *c = new ncclComm();
        c->world = w;
        c->rank = k;
        c->device = devlist ? devlist[k] : k;  // duplicates welcome: that is what this double is for
        w->comms.push_back(c);
        ncclResult_t rc = comm_events(c);
        if (rc != ncclSuccess) return rc;
    }
    w->red_device = w->comms[0]->device;
    F_HIP(hipSetDevice(w->red_device));
    F_HIP(hipStreamCreateWithFlags(&w->red_stream, hipStreamNonBlocking));
    F_HIP(hipEventCreateWithFlags(&w->red_done, hipEventDisableTiming));
    F_HIP(hipHostMalloc(reinterpret_cast<void **>(&w->h_release), sizeof(uint32_t), hipHostMallocDefault));
    *w->h_release = 0;
    for (int k = 0; k < ndev; ++k) comms[k] = w->comms[(size_t)k];
    g_inits++;
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(ncclInvalidArgument, "ncclCommInitRank arguments");
    if (hang_mode("init")) return hang_in_init();
    World *w = new World();
    w->local = false;
    w->n = nranks;
    w->alive = 1;
    char hex[33];
    for (int k = 0; k < 16; ++k) std::snprintf(hex + 2 * k, 3, "%02x", (unsigned)(uint8_t)id.internal[k]);
    w->prefix = shm_dir() + "/fakerccl-" + hex;
    ncclComm *c = new ncclComm();
    c->world = w;
    c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) return fail(ncclUnhandledCudaError, "hipGetDevice");
    c->send_seq.assign((size_t)nranks, 0);
    c->recv_seq.assign((size_t)nranks, 0);
    // bootstrap: everybody checks in, nobody leaves before everybody has
    const std::string mine = w->prefix + ".join." + std::to_string(rank);
    const int fd = ::open(mine.c_str(), O_CREAT | O_WRONLY, 0600);
    if (fd < 0) return fail(ncclSystemError, "open " + mine + ": " + std::strerror(errno));
    ::close(fd);
    const double limit = env_seconds("FAKE_RCCL_TIMEOUT_S", 60);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < nranks; ++r)
        while (!exists(w->prefix + ".join." + std::to_string(r))) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
                return fail(ncclRemoteError, "rank " + std::to_string(r) + " never joined the communicator");
            ::usleep(500);
        }
    *comm = c;
    g_inits++;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    release_world(c->world);
    g_aborts++;
    return ncclCommDestroy(c);
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c || !c->live) return ncclSuccess;
    DeviceGuard g;
    write_stats(c);
    c->live = false;
    World *w = c->world;
    release_world(w);  // whatever a "never completes" group still holds is let go
    if (w->local) {
        std::lock_guard<std::mutex> lk(g_mutex);
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        if (c->ev_send) (void)hipEventDestroy(c->ev_send);
        if (c->ev_recv) (void)hipEventDestroy(c->ev_recv);
        if (--w->alive == 0) {
            (void)hipSetDevice(w->red_device);
            (void)hipStreamSynchronize(w->red_stream);
            (void)hipStreamDestroy(w->red_stream);
            (void)hipEventDestroy(w->red_done);
            (void)hipFree(w->tmp[0]);
            (void)hipFree(w->tmp[1]);
            (void)hipHostFree(w->h_release);
            w->h_release = nullptr;
        }
    }
    // (across processes nothing is unlinked here: a peer may still be reading this rank's last contribution, and destroying
    // a communicator is no barrier; the test removes its FAKE_RCCL_SHM_DIR as a whole afterwards)
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    ++t_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (t_depth == 0) return fail(ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart");
    if (--t_depth) return ncclSuccess;
    return flush();
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(Op{SEND, comm, buf, nullptr, count, type, ncclSum, peer, stream});
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(Op{RECV, comm, nullptr, buf, count, type, ncclSum, peer, stream});
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    return post(Op{ALLREDUCE, comm, send, recv, count, type, op, -1, stream});
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t type, ncclComm_t comm, hipStream_t stream)
{
    return post(Op{ALLGATHER, comm, send, recv, sendcount, type, ncclSum, -1, stream});
}

// for a test in the same process: {matched pairs, p2p bytes, collectives, groups, mismatches, inits, hangs, aborts}
void fake_rccl_stats(uint64_t out[8])
{
    const uint64_t v[8] = {g_pairs.load(), g_p2p_bytes.load(), g_colls.load(), g_groups.load(), g_mismatches.load(), g_inits.load(), g_hangs.load(), g_aborts.load()};
    std::memcpy(out, v, sizeof v);
}

const char *fake_rccl_last_detail() { return t_detail.c_str(); }

}  // extern "C"
