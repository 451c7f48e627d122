// Gradient all-reduce over RCCL's C API: the collective of the data-parallel reducer
// (pasero_amd/ddp.py) that replaces `torch.nn.parallel.DistributedDataParallel` (pasero/training.py:243-250), without
// the round trip through torch.distributed's Python / c10d layers per bucket.
//
// The RCCL symbols are resolved at run time from the library the process already has (PyTorch-ROCm ships and loads its
// own librccl; linking a second copy would put two RCCL instances in one process): pk_comm_open(path) = dlopen + dlsym.
// One communicator per process (one process per GPU).  Three schedules of  buf <- mean over ranks (buf), in place, on
// the given stream:
//   0  ncclAllReduce(ncclAvg)                                        RCCL picks its algorithm
//   1  ncclReduceScatter(ncclAvg) into this rank's shard + ncclAllGather
//   2  DIRECT: every rank sends shard j of its buffer to rank j and receives the n - 1 peer copies of its own shard
//      (one grouped send/recv: all 7 xGMI links of a GPU carry count/n elements at once instead of a ring's 2 (n-1)/n
//      of the bucket over one link per hop), sums them in rank order with pk's own kernel (deterministic, fp32
//      accumulation), then sends its reduced shard to every peer (second grouped exchange).  Needs `scratch` of the
//      bucket's size.
// ddp.py times the three at construction on the real topology, checks each against torch.distributed's all-reduce on
// random data, and keeps the fastest correct one (falling back to torch.distributed if none is).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstring>
#include "common.h"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;  // NCCL_UNIQUE_ID_BYTES (rccl.h:40-43)
enum { NCCL_SUM = 0, NCCL_AVG = 4, NCCL_F16 = 6, NCCL_F32 = 7, NCCL_BF16 = 9 };  // rccl.h:447-468

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = -1;
} g;

#define PK_NCCL(call, what)                                                                          \
    do {                                                                                             \
        int r_ = (call);                                                                             \
        if (r_ != 0) {                                                                               \
            pk_set_error("%s: %s", what, g.GetErrorString ? g.GetErrorString(r_) : "rccl error");    \
            return r_ > 0 ? -r_ - 1000 : r_;                                                         \
        }                                                                                            \
    } while (0)

// inside ncclGroupStart .. ncclGroupEnd: close the group before reporting (an open group would swallow the next call)
#define PK_NCCL_G(call, what)                                                                        \
    do {                                                                                             \
        int r_ = (call);                                                                             \
        if (r_ != 0) {                                                                               \
            pk_set_error("%s: %s", what, g.GetErrorString ? g.GetErrorString(r_) : "rccl error");    \
            g.GroupEnd();                                                                            \
            return r_ > 0 ? -r_ - 1000 : r_;                                                         \
        }                                                                                            \
    } while (0)

int nccl_type(int dtype) { return dtype == PK_F32 ? NCCL_F32 : dtype == PK_BF16 ? NCCL_BF16 : NCCL_F16; }
size_t elem_size(int dtype) { return dtype == PK_F32 ? 4 : 2; }

// own[i] = (sum over r of parts[r][i]) / n, r = 0 .. n-1 in order, fp32 accumulation; parts[r] = scratch + r * shard for
// r != rank and the rank's own shard of the bucket for r == rank
template <typename T>
__global__ __launch_bounds__(256) void shard_mean_kernel(T* __restrict__ own, const T* __restrict__ scratch,
                                                         long long shard, int n, int rank) {
    constexpr int V = 16 / sizeof(T);
    const float inv = 1.f / (float)n;
    const long long nv = shard / V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int r = 0; r < n; ++r) {
            const T* src = (r == rank ? own : scratch + (long long)r * shard) + i * V;
            Vec16<T> v = load16<T>(src);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += v.get(e);
        }
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < V; ++e) o.set(e, acc[e] * inv);
        store16<T>(own + i * V, o);
    }
}

}  // namespace

extern "C" int pk_comm_open(const char* librccl_path) {
    if (g.handle) return 0;
    PK_CHECK_ARG(librccl_path, "pk_comm_open: null path");
    void* h = dlopen(librccl_path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { pk_set_error("pk_comm_open: dlopen(%s): %s", librccl_path, dlerror()); return -1; }
#define PK_SYM(field, name)                                                                          \
    *(void**)(&g.field) = dlsym(h, name);                                                            \
    if (!g.field) { pk_set_error("pk_comm_open: %s has no symbol %s", librccl_path, name); return -1; }
    PK_SYM(GetUniqueId, "ncclGetUniqueId") PK_SYM(CommInitRank, "ncclCommInitRank") PK_SYM(CommDestroy, "ncclCommDestroy")
    PK_SYM(AllReduce, "ncclAllReduce") PK_SYM(ReduceScatter, "ncclReduceScatter") PK_SYM(AllGather, "ncclAllGather")
    PK_SYM(Send, "ncclSend") PK_SYM(Recv, "ncclRecv") PK_SYM(GroupStart, "ncclGroupStart") PK_SYM(GroupEnd, "ncclGroupEnd")
    PK_SYM(GetErrorString, "ncclGetErrorString")
#undef PK_SYM
    g.handle = h;
    return 0;
}

extern "C" int pk_comm_unique_id(void* out, int nbytes) {
    PK_CHECK_ARG(g.handle, "pk_comm_unique_id: pk_comm_open first");
    PK_CHECK_ARG(out && nbytes >= (int)sizeof(ncclUniqueId), "pk_comm_unique_id: need %d bytes", (int)sizeof(ncclUniqueId));
    ncclUniqueId id;
    PK_NCCL(g.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out, &id, sizeof(id));
    return 0;
}

// collective over all ranks: every rank passes rank 0's id; the communicator lives on the CURRENT device
extern "C" int pk_comm_init(const void* id_bytes, int nranks, int rank) {
    PK_CHECK_ARG(g.handle, "pk_comm_init: pk_comm_open first");
    PK_CHECK_ARG(!g.comm, "pk_comm_init: communicator already initialised");
    PK_CHECK_ARG(id_bytes && nranks >= 1 && rank >= 0 && rank < nranks, "pk_comm_init: bad arguments");
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    PK_NCCL(g.CommInitRank(&g.comm, nranks, id, rank), "ncclCommInitRank");
    g.nranks = nranks;
    g.rank = rank;
    return 0;
}

extern "C" int pk_comm_destroy(void) {
    if (g.comm) {
        PK_NCCL(g.CommDestroy(g.comm), "ncclCommDestroy");
        g.comm = nullptr;
        g.nranks = 0;
        g.rank = -1;
    }
    return 0;
}

extern "C" int pk_comm_size(void) { return g.comm ? g.nranks : 0; }

// Where the direct schedule (2) of pk_comm_all_reduce_mean puts things, in ELEMENTS — pure host arithmetic, no GPU and no
// communicator needed (tests/test_ddp_cpu.py replays it for 2, 4 and 8 ranks).  Exchange 1: this rank sends
// buf[send_off[p], + shard) to every peer p and receives p's copy of its own shard into scratch[recv_off[p], + shard);
// the reduction reads part r from scratch + recv_off[r] (r != rank) or from buf + own_off (r == rank), r = 0..n-1 in
// order, and writes buf[own_off, + shard).  Exchange 2: sends that shard to every peer, receives p's reduced shard into
// buf[send_off[p], + shard).
extern "C" int pk_comm_direct_plan(long long count, int nranks, int rank, long long* shard, long long* own_off,
                                   long long* send_off, long long* recv_off) {
    PK_CHECK_ARG(nranks >= 1 && rank >= 0 && rank < nranks && count >= 0, "pk_comm_direct_plan: bad arguments");
    PK_CHECK_ARG(count % ((long long)nranks * 8) == 0, "pk_comm_direct_plan: count %lld is not a multiple of %d", count,
                 nranks * 8);
    PK_CHECK_ARG(shard && own_off && send_off && recv_off, "pk_comm_direct_plan: null output");
    const long long sh = count / nranks;
    *shard = sh;
    *own_off = (long long)rank * sh;
    for (int p = 0; p < nranks; ++p) { send_off[p] = (long long)p * sh; recv_off[p] = (long long)p * sh; }
    return 0;
}

// buf[0 .. count) <- mean over the ranks, in place, enqueued on `stream`.  schedules 1 and 2: count % (nranks * 8) == 0
// (whole 16-byte chunks per shard); schedule 2: `scratch` of count elements.
extern "C" int pk_comm_all_reduce_mean(void* buf, long long count, int dtype, int schedule, void* scratch, void* stream) {
    PK_CHECK_ARG(g.comm, "pk_comm_all_reduce_mean: no communicator");
    PK_CHECK_ARG(buf && count >= 0, "pk_comm_all_reduce_mean: bad buffer");
    PK_CHECK_ARG(dtype == PK_F32 || dtype == PK_BF16 || dtype == PK_F16, "pk_comm_all_reduce_mean: dtype %d", dtype);
    if (count == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int n = g.nranks, t = nccl_type(dtype);
    if (schedule == 0) {
        PK_NCCL(g.AllReduce(buf, buf, (size_t)count, t, NCCL_AVG, g.comm, s), "ncclAllReduce");
        return 0;
    }
    PK_CHECK_ARG(n <= 64, "pk_comm_all_reduce_mean: %d ranks", n);
    long long shard, own_off, send_off[64], recv_off[64];
    if (int rc = pk_comm_direct_plan(count, n, g.rank, &shard, &own_off, send_off, recv_off)) return rc;
    const size_t esz = elem_size(dtype);
    char* own = (char*)buf + (size_t)own_off * esz;
    if (schedule == 1) {
        PK_NCCL(g.ReduceScatter(buf, own, (size_t)shard, t, NCCL_AVG, g.comm, s), "ncclReduceScatter");
        PK_NCCL(g.AllGather(own, buf, (size_t)shard, t, g.comm, s), "ncclAllGather");
        return 0;
    }
    PK_CHECK_ARG(schedule == 2, "pk_comm_all_reduce_mean: schedule %d", schedule);
    PK_CHECK_ARG(scratch || n == 1, "pk_comm_all_reduce_mean: the direct schedule needs scratch");
    if (n > 1) {
        PK_NCCL(g.GroupStart(), "ncclGroupStart");
        for (int peer = 0; peer < n; ++peer) {
            if (peer == g.rank) continue;
            PK_NCCL_G(g.Send((char*)buf + (size_t)send_off[peer] * esz, (size_t)shard, t, peer, g.comm, s), "ncclSend");
            PK_NCCL_G(g.Recv((char*)scratch + (size_t)recv_off[peer] * esz, (size_t)shard, t, peer, g.comm, s), "ncclRecv");
        }
        PK_NCCL(g.GroupEnd(), "ncclGroupEnd");
    }
    const int blocks = (int)std::min<long long>(1024, (shard / (16 / (long long)esz) + 255) / 256);
    if (dtype == PK_F32)
        hipLaunchKernelGGL((shard_mean_kernel<float>), dim3(blocks), dim3(256), 0, s, (float*)own, (const float*)scratch, shard, n, g.rank);
    else if (dtype == PK_BF16)
        hipLaunchKernelGGL((shard_mean_kernel<bf16>), dim3(blocks), dim3(256), 0, s, (bf16*)own, (const bf16*)scratch, shard, n, g.rank);
    else
        hipLaunchKernelGGL((shard_mean_kernel<f16>), dim3(blocks), dim3(256), 0, s, (f16*)own, (const f16*)scratch, shard, n, g.rank);
    PK_LAUNCH_CHECK();
    if (n > 1) {
        PK_NCCL(g.GroupStart(), "ncclGroupStart");
        for (int peer = 0; peer < n; ++peer) {
            if (peer == g.rank) continue;
            PK_NCCL_G(g.Send(own, (size_t)shard, t, peer, g.comm, s), "ncclSend");
            PK_NCCL_G(g.Recv((char*)buf + (size_t)send_off[peer] * esz, (size_t)shard, t, peer, g.comm, s), "ncclRecv");
        }
        PK_NCCL(g.GroupEnd(), "ncclGroupEnd");
    }
    return 0;
}
