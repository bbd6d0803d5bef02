// rccl_comm.hip -- mpg_comm on a native RCCL communicator (mpg_rccl_* of include/mpgadget_hip.h).
//
// The reference's force path is collective over MPI_COMM_WORLD: the export / import of tree-walk queries (treewalk.c:586-655:
// MPI_Alltoall of counts, then MPI_Isend / MPI_Irecv per peer) and the PM mesh exchanges (petapm.c:751, 815, 869: MPI_Alltoallv).  On a
// node of MI355X those exchanges belong on RCCL over xGMI, device memory to device memory, as stream-ordered work: this file implements
// the three collectives csrc/dist.hip asks its caller for
//     alltoallv  = ncclGroupStart + one ncclSend / ncclRecv per peer with data + ncclGroupEnd  (xGMI is point to point: every pair of
//                  GPUs has its own link, so the grouped sends of one exchange run on all links at once)
//     allreduce  = ncclAllReduce in place
//     alltoall_i64 = the same grouped send / recv on a small device buffer (counts; host arrays at the interface)
// on ONE HIP stream - the engine's, handed over through mpg_comm.bind_stream by mpg_dist_create.  With device pointers the callbacks
// enqueue and return: no host synchronisation, no Python frame and no host staging between two collectives of a force step.  Host
// pointers (the small tree / count arrays of the domain decomposition) are staged through a device scratch buffer and the call blocks.
//
// The library does not link librccl: it is opened at run time (dlopen of librccl.so.1, which inside a PyTorch process resolves to the
// RCCL PyTorch has already loaded, elsewhere to /opt/rocm/lib through the library's RUNPATH), so libmpgadget_hip.so still loads on a box
// without RCCL and a caller with MPI only (shim/mpg_mpi_comm.c) pays nothing.  Bootstrap: rank 0 calls mpg_rccl_get_unique_id and the
// CALLER distributes the 128 bytes (MPI_Bcast in shim/mpg_rccl_mpi.c, torch.distributed in bench.py, shared memory in the C test).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "engine_internal.h"

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;           // optional: a failed callback leaves peers with enqueued work
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi &api()
{
    static RcclApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("MPG_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for(const char *n : names) {
            if(!n || !*n)
                continue;
            a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if(a.handle)
                break;
            a.error = dlerror();
        }
        if(!a.handle)
            return;
        bool ok = true;
        auto sym = [&](const char *s) {
            void *p = dlsym(a.handle, s);
            if(!p) {
                ok = false;
                a.error = std::string("librccl lacks ") + s;
            }
            return p;
        };
        a.GetVersion = (decltype(a.GetVersion))sym("ncclGetVersion");
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
        a.CommUserRank = (decltype(a.CommUserRank))sym("ncclCommUserRank");
        a.CommAbort = (decltype(a.CommAbort))dlsym(a.handle, "ncclCommAbort");
        a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
        a.Send = (decltype(a.Send))sym("ncclSend");
        a.Recv = (decltype(a.Recv))sym("ncclRecv");
        a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        if(!ok) {
            dlclose(a.handle);
            a.handle = nullptr;
        }
    });
    return a;
}

} // namespace

struct mpg_rccl {
    ncclComm_t comm = nullptr;
    int me = 0, nt = 1, device = 0;
    hipStream_t stream = nullptr; // the engine's (bind_stream); until then an own stream, and every call blocks
    hipStream_t own_stream = nullptr;
    bool bound = false;
    bool self_through_rccl = false; // MPG_RCCL_SELF=1: the rank's own block goes through ncclSend / ncclRecv too (a one-rank test then exercises them)
    size_t piece = (size_t)1 << 30; // no single ncclSend / ncclRecv carries more than this many bytes (MPG_RCCL_PIECE)
    mpg::DevBuf<char> scratch;
    std::string error;
    int64_t calls[3] = {0, 0, 0};   // allreduce, alltoall_i64, alltoallv
    int64_t bytes_sent = 0;
    bool dead = false;              // a collective failed half-way: the communicator was aborted (fail_fatal)
};

namespace {

#define NCCL_TRY(r, call)                                                                              \
    do {                                                                                               \
        const ncclResult_t res_ = (call);                                                              \
        if(res_ != ncclSuccess) {                                                                      \
            (r)->error = std::string(#call) + ": " + (api().GetErrorString ? api().GetErrorString(res_) : "RCCL error"); \
            return 1;                                                                                  \
        }                                                                                              \
    } while(0)
#define HIP_TRY(r, call)                                                              \
    do {                                                                              \
        const hipError_t e_ = (call);                                                 \
        if(e_ != hipSuccess) {                                                        \
            (r)->error = std::string(#call) + ": " + hipGetErrorString(e_);           \
            return 1;                                                                 \
        }                                                                             \
    } while(0)

// one block of `bytes` to / from `peer`, in pieces of at most r->piece bytes.  Always as bytes (ncclChar): a send and its matching receive
// must agree on datatype and count, and the two sides of a block know nothing of each other's alignment; for point-to-point transfers
// the datatype only scales the count.
// A callback that fails after some ranks have enqueued their side of a collective cannot be repaired locally: the peers' matching
// operations would wait for ever.  The communicator is aborted (ncclCommAbort: pending operations of every rank of it end with an
// error) and every later call on it fails at once - a failed collective is fatal for the job, as an MPI error is in the reference.
int fail_fatal(mpg_rccl *r)
{
    if(r->comm && api().CommAbort) {
        api().CommAbort(r->comm);
        r->comm = nullptr;
    }
    r->dead = true;
    return 1;
}

int send_block(mpg_rccl *r, const char *p, size_t bytes, int peer)
{
    for(size_t o = 0; o < bytes; o += r->piece)
        NCCL_TRY(r, api().Send(p + o, std::min(r->piece, bytes - o), ncclChar, peer, r->comm, r->stream));
    return 0;
}
int recv_block(mpg_rccl *r, char *p, size_t bytes, int peer)
{
    for(size_t o = 0; o < bytes; o += r->piece)
        NCCL_TRY(r, api().Recv(p + o, std::min(r->piece, bytes - o), ncclChar, peer, r->comm, r->stream));
    return 0;
}

// the exchange itself, device pointers, enqueued on r->stream
int a2av_device(mpg_rccl *r, const char *send, const int64_t *sb, const int64_t *sd, char *recv, const int64_t *rb, const int64_t *rd)
{
    const int nt = r->nt, me = r->me;
    if(r->dead) {
        r->error = "the communicator was aborted after a failed collective";
        return 1;
    }
    if(!r->self_through_rccl) {
        if(sb[me] != rb[me]) {
            r->error = "alltoallv: the rank's own block has different send and receive sizes";
            return 1;
        }
        if(sb[me] > 0)
            HIP_TRY(r, hipMemcpyAsync(recv + rd[me], send + sd[me], (size_t)sb[me], hipMemcpyDeviceToDevice, r->stream));
    }
    bool any = false;
    for(int p = 0; p < nt; p++)
        if((p != me || r->self_through_rccl) && (sb[p] > 0 || rb[p] > 0))
            any = true;
    if(!any)
        return 0;
    NCCL_TRY(r, api().GroupStart());
    // peers in a rotated order (rank + k): every rank starts on a different link
    for(int k = 0; k < nt; k++) {
        const int to = (me + k) % nt, from = (me - k + nt) % nt;
        if((to != me || r->self_through_rccl) && sb[to] > 0)
            if(send_block(r, send + sd[to], (size_t)sb[to], to)) {
                api().GroupEnd();
                return fail_fatal(r);
            }
        if((from != me || r->self_through_rccl) && rb[from] > 0)
            if(recv_block(r, recv + rd[from], (size_t)rb[from], from)) {
                api().GroupEnd();
                return fail_fatal(r);
            }
    }
    NCCL_TRY(r, api().GroupEnd());
    return 0;
}

int finish(mpg_rccl *r, bool block)
{
    if(block || !r->bound)
        HIP_TRY(r, hipStreamSynchronize(r->stream));
    return 0;
}

int cb_bind_stream(void *ctx, void *hip_stream)
{
    mpg_rccl *r = (mpg_rccl *)ctx;
    hipStream_t next = (hipStream_t)hip_stream;
    if(next != r->stream) {
        // collectives still pending on the stream used so far must precede anything enqueued on the new one (RCCL orders the
        // operations of one communicator by their stream: two streams need an edge between them)
        HIP_TRY(r, hipSetDevice(r->device));
        hipEvent_t ev;
        HIP_TRY(r, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, r->stream);
        if(e == hipSuccess)
            e = hipStreamWaitEvent(next, ev, 0);
        (void)hipEventDestroy(ev);
        if(e != hipSuccess) {
            r->error = std::string("bind_stream: ") + hipGetErrorString(e);
            return 1;
        }
    }
    r->stream = next;
    r->bound = true;
    return 0;
}

int cb_allreduce(void *ctx, void *buf, int64_t count, int dtype, int op, int on_device)
{
    mpg_rccl *r = (mpg_rccl *)ctx;
    r->calls[0]++;
    if(count <= 0)
        return 0;
    if(r->dead) {
        r->error = "the communicator was aborted after a failed collective";
        return 1;
    }
    HIP_TRY(r, hipSetDevice(r->device));
    const ncclDataType_t ty = dtype ? ncclInt64 : ncclDouble;
    const ncclRedOp_t red = op ? ncclMax : ncclSum;
    if(on_device) {
        NCCL_TRY(r, api().AllReduce(buf, buf, (size_t)count, ty, red, r->comm, r->stream));
        return finish(r, false);
    }
    const size_t bytes = (size_t)count * 8;
    r->scratch.reserve(bytes + 64);
    HIP_TRY(r, hipMemcpyAsync(r->scratch.p, buf, bytes, hipMemcpyHostToDevice, r->stream));
    NCCL_TRY(r, api().AllReduce(r->scratch.p, r->scratch.p, (size_t)count, ty, red, r->comm, r->stream));
    HIP_TRY(r, hipMemcpyAsync(buf, r->scratch.p, bytes, hipMemcpyDeviceToHost, r->stream));
    return finish(r, true);
}

int cb_alltoallv(void *ctx, const void *send, const int64_t *sb, const int64_t *sd, void *recv, const int64_t *rb, const int64_t *rd, int on_device)
{
    mpg_rccl *r = (mpg_rccl *)ctx;
    r->calls[2]++;
    if(r->dead) { // (every callback fails at once on an aborted communicator; cb_alltoall_i64 and the self-test come through here)
        r->error = "the communicator was aborted after a failed collective";
        return 1;
    }
    HIP_TRY(r, hipSetDevice(r->device));
    for(int p = 0; p < r->nt; p++)
        if(p != r->me)
            r->bytes_sent += sb[p];
    if(on_device) {
        if(a2av_device(r, (const char *)send, sb, sd, (char *)recv, rb, rd))
            return 1;
        return finish(r, false);
    }
    // host buffers: packed copies on the device, exchanged there
    std::vector<int64_t> psd((size_t)r->nt), prd((size_t)r->nt);
    int64_t stot = 0, rtot = 0;
    for(int p = 0; p < r->nt; p++) {
        psd[p] = stot;
        stot += (sb[p] + 7) / 8 * 8;
    }
    for(int p = 0; p < r->nt; p++) {
        prd[p] = stot + rtot;
        rtot += (rb[p] + 7) / 8 * 8;
    }
    r->scratch.reserve((size_t)(stot + rtot) + 64);
    for(int p = 0; p < r->nt; p++)
        if(sb[p] > 0)
            HIP_TRY(r, hipMemcpyAsync(r->scratch.p + psd[p], (const char *)send + sd[p], (size_t)sb[p], hipMemcpyHostToDevice, r->stream));
    if(a2av_device(r, r->scratch.p, sb, psd.data(), r->scratch.p, rb, prd.data()))
        return 1;
    for(int p = 0; p < r->nt; p++)
        if(rb[p] > 0)
            HIP_TRY(r, hipMemcpyAsync((char *)recv + rd[p], r->scratch.p + prd[p], (size_t)rb[p], hipMemcpyDeviceToHost, r->stream));
    return finish(r, true);
}

int cb_alltoall_i64(void *ctx, const int64_t *send, int64_t *recv)
{
    mpg_rccl *r = (mpg_rccl *)ctx;
    r->calls[1]++;
    r->calls[2]--; // (counted below)
    std::vector<int64_t> b((size_t)r->nt, 8), dsp((size_t)r->nt);
    for(int p = 0; p < r->nt; p++)
        dsp[p] = 8 * p;
    return cb_alltoallv(ctx, send, b.data(), dsp.data(), recv, b.data(), dsp.data(), 0);
}

} // namespace

extern "C" {

int mpg_rccl_available(void)
{
    return api().handle != nullptr;
}

int mpg_rccl_get_unique_id(void *id128)
{
    API_BEGIN
    MPG_CHECK(id128, "null argument");
    MPG_CHECK(api().handle, "RCCL is not available: " + api().error);
    static_assert(sizeof(ncclUniqueId) == MPG_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    const ncclResult_t res = api().GetUniqueId(&id);
    MPG_CHECK(res == ncclSuccess, std::string("ncclGetUniqueId: ") + api().GetErrorString(res));
    memcpy(id128, &id, sizeof(id));
    API_END
}

int mpg_rccl_create(mpg_rccl **out, int ThisTask, int NTask, const void *id128, int device)
{
    API_BEGIN
    MPG_CHECK(out && id128, "null argument");
    MPG_CHECK(NTask >= 1 && NTask <= 64 && ThisTask >= 0 && ThisTask < NTask, "mpg_rccl_create: bad ThisTask / NTask");
    MPG_CHECK(api().handle, "RCCL is not available: " + api().error);
    MPG_HIP(hipSetDevice(device));
    mpg_rccl *r = new mpg_rccl();
    r->me = ThisTask;
    r->nt = NTask;
    r->device = device;
    if(const char *e = getenv("MPG_RCCL_SELF"))
        r->self_through_rccl = atoi(e) != 0;
    if(const char *e = getenv("MPG_RCCL_PIECE"))
        if(atoll(e) >= 8)
            r->piece = (size_t)atoll(e) / 8 * 8;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    const ncclResult_t res = api().CommInitRank(&r->comm, NTask, id, ThisTask);
    if(res != ncclSuccess) {
        const std::string msg = std::string("ncclCommInitRank: ") + api().GetErrorString(res);
        delete r;
        throw mpg::Error(msg);
    }
    const hipError_t se = hipStreamCreateWithFlags(&r->own_stream, hipStreamNonBlocking);
    if(se != hipSuccess) {
        api().CommDestroy(r->comm);
        delete r;
        throw mpg::Error(std::string("mpg_rccl_create: hipStreamCreateWithFlags: ") + hipGetErrorString(se));
    }
    r->stream = r->own_stream;
    *out = r;
    API_END
}

int mpg_rccl_comm(mpg_rccl *r, mpg_comm *out)
{
    API_BEGIN
    MPG_CHECK(r && out, "null argument");
    memset(out, 0, sizeof(*out));
    out->ctx = r;
    out->ThisTask = r->me;
    out->NTask = r->nt;
    out->device_buffers = 1;
    out->allreduce = cb_allreduce;
    out->alltoall_i64 = cb_alltoall_i64;
    out->alltoallv = cb_alltoallv;
    out->bind_stream = cb_bind_stream;
    API_END
}

int mpg_rccl_comm_info(mpg_rccl *r, int *nranks, int *rank, int *device)
{
    API_BEGIN
    MPG_CHECK(r, "null argument");
    MPG_CHECK(r->comm, "the communicator was aborted after a failed collective");
    // what RCCL itself says the communicator spans (ncclCommCount / ncclCommUserRank), not what the caller passed to mpg_rccl_create
    int n = -1, me = -1;
    ncclResult_t res = api().CommCount(r->comm, &n);
    MPG_CHECK(res == ncclSuccess, std::string("ncclCommCount: ") + api().GetErrorString(res));
    res = api().CommUserRank(r->comm, &me);
    MPG_CHECK(res == ncclSuccess, std::string("ncclCommUserRank: ") + api().GetErrorString(res));
    if(nranks)
        *nranks = n;
    if(rank)
        *rank = me;
    if(device)
        *device = r->device;
    API_END
}

const char *mpg_rccl_last_error(mpg_rccl *r)
{
    return r ? r->error.c_str() : "";
}

int mpg_rccl_stats(mpg_rccl *r, int64_t *calls3, int64_t *bytes_sent, int *version)
{
    API_BEGIN
    MPG_CHECK(r, "null argument");
    if(calls3)
        for(int i = 0; i < 3; i++)
            calls3[i] = r->calls[i];
    if(bytes_sent)
        *bytes_sent = r->bytes_sent;
    if(version) {
        *version = 0;
        api().GetVersion(version);
    }
    API_END
}

/* Every collective once on a known pattern (collective over the ranks): rank r sends (r, p, k) words to peer p, all-reduces a sum and
 * a maximum, exchanges counts; returns non-zero and mpg_last_error() if any byte differs.  `bytes_per_peer` sizes the alltoallv (0: 1 MiB);
 * a caller that doubts the transport at large sizes passes more (the 1 GiB threshold at which torch's all_to_all_single misbehaved on
 * this ROCm is covered by MPG_RCCL_PIECE = 2^30 per ncclSend). */
int mpg_rccl_selftest(mpg_rccl *r, int64_t bytes_per_peer)
{
    API_BEGIN
    MPG_CHECK(r, "null argument");
    MPG_HIP(hipSetDevice(r->device));
    const int nt = r->nt, me = r->me;
    const int64_t words = (bytes_per_peer > 0 ? bytes_per_peer : (1 << 20)) / 8;
    // counts
    std::vector<int64_t> sc((size_t)nt), rc((size_t)nt, -1);
    for(int p = 0; p < nt; p++)
        sc[p] = 1000 * me + p;
    MPG_CHECK(cb_alltoall_i64(r, sc.data(), rc.data()) == 0, "selftest alltoall_i64: " + r->error);
    for(int p = 0; p < nt; p++)
        MPG_CHECK(rc[p] == 1000 * p + me, "selftest alltoall_i64: wrong value received");
    // all-reduce, host and device forms
    double hs[2] = {(double)(me + 1), 0.5};
    MPG_CHECK(cb_allreduce(r, hs, 2, 0, 0, 0) == 0, "selftest allreduce: " + r->error);
    MPG_CHECK(hs[0] == 0.5 * nt * (nt + 1) && hs[1] == 0.5 * nt, "selftest allreduce (sum): wrong value");
    int64_t hm = 7 * me;
    MPG_CHECK(cb_allreduce(r, &hm, 1, 1, 1, 0) == 0, "selftest allreduce: " + r->error);
    MPG_CHECK(hm == 7 * (nt - 1), "selftest allreduce (max): wrong value");
    // alltoallv on device buffers: block for peer p holds words (me << 40) | (p << 32) | k, sizes differ per pair
    std::vector<int64_t> sb((size_t)nt), sd((size_t)nt), rb((size_t)nt), rd((size_t)nt);
    int64_t stot = 0, rtot = 0;
    for(int p = 0; p < nt; p++) {
        sb[p] = 8 * (words - (me + 2 * p) % 5);
        rb[p] = 8 * (words - (p + 2 * me) % 5);
        sd[p] = stot;
        rd[p] = rtot;
        stot += sb[p];
        rtot += rb[p];
    }
    std::vector<uint64_t> hsend((size_t)stot / 8), hrecv((size_t)rtot / 8, 0);
    for(int p = 0; p < nt; p++)
        for(int64_t k = 0; k < sb[p] / 8; k++)
            hsend[sd[p] / 8 + k] = ((uint64_t)me << 40) | ((uint64_t)p << 32) | (uint64_t)(k & 0xffffffff);
    mpg::DevBuf<char> ds, dr;
    ds.reserve((size_t)stot + 8);
    dr.reserve((size_t)rtot + 8);
    MPG_HIP(hipMemcpyAsync(ds.p, hsend.data(), (size_t)stot, hipMemcpyHostToDevice, r->stream));
    MPG_HIP(hipMemsetAsync(dr.p, 0xff, (size_t)rtot, r->stream));
    MPG_CHECK(cb_alltoallv(r, ds.p, sb.data(), sd.data(), dr.p, rb.data(), rd.data(), 1) == 0, "selftest alltoallv: " + r->error);
    MPG_HIP(hipMemcpyAsync(hrecv.data(), dr.p, (size_t)rtot, hipMemcpyDeviceToHost, r->stream));
    MPG_HIP(hipStreamSynchronize(r->stream));
    for(int p = 0; p < nt; p++)
        for(int64_t k = 0; k < rb[p] / 8; k++)
            MPG_CHECK(hrecv[rd[p] / 8 + k] == (((uint64_t)p << 40) | ((uint64_t)me << 32) | (uint64_t)(k & 0xffffffff)),
                      "selftest alltoallv: wrong data received from rank " + std::to_string(p) + " at word " + std::to_string(k));
    API_END
}

void mpg_rccl_destroy(mpg_rccl *r)
{
    if(!r)
        return;
    (void)hipSetDevice(r->device);
    //  a communicator that died half-way through a group - and could not be aborted because this RCCL exports no
    // ncclCommAbort - still has operations enqueued whose peers will never arrive: neither wait for the stream nor ask RCCL to tear the
    // communicator down (both can block for ever); the process is on its way out anyway
    if(!r->dead) {
        if(r->stream)
            (void)hipStreamSynchronize(r->stream);
        if(r->comm && api().CommDestroy)
            api().CommDestroy(r->comm);
    }
    if(r->own_stream)
        (void)hipStreamDestroy(r->own_stream);
    delete r;
}

} // extern "C"
