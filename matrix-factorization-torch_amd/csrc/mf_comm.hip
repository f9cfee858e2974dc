// mf_comm.hip -- the exchange steps of the sharded hot path, issued from the C side ON THE COMPUTE STREAM.
//
// The reference has no collective at all.  The sharded design of distributed.py needs three
// latency-bound exchanges per training step (ids out, rows back, row gradients out) and one per retrieval call.
// Through torch.distributed every one of them costs two cross-stream joins (ProcessGroupNCCL runs its own
// stream): 20-30 us each on this part, +10 % on a 1 ms step before a byte crosses xGMI.  Here RCCL is
// called directly -- grouped ncclSend / ncclRecv pairs: a direct all-to-all over the point-to-point xGMI links, the
// right shape for <= 10 MB messages (never a ring) -- on the caller's stream, so an exchange is just another node
// between two kernels.  librccl is opened lazily (dlopen): a single-GPU process never loads it.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "mf_common.h"

namespace {

typedef struct { char internal[128]; } RcclUniqueId;          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* RcclComm;
enum { RCCL_CHAR = 0 };                                        // ncclChar / ncclInt8

struct RcclApi {
    int (*GetUniqueId)(RcclUniqueId*);
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int);
    int (*CommDestroy)(RcclComm);
    int (*GroupStart)();
    int (*GroupEnd)();
    int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t);
    int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t);
    int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t);
    const char* (*GetErrorString)(int);
    const char* source = "not loaded";
    bool ok = false;
};

RcclApi& api() {
    static RcclApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        // torch has usually mapped ITS librccl already (the bootstrap process group): take that very copy -- two RCCL
        // runtimes in one process would each own a set of proxy threads and IPC handles.  First by soname without
        // loading (RTLD_NOLOAD), then whatever image already exports the symbols (torch may link RCCL into another of
        // its libraries), and only in a process that has no RCCL at all a fresh dlopen.
        // (RTLD_DEFAULT is a NULL handle on glibc: "found" is tracked apart from the handle -- ADVICE r3)
        void* h = nullptr;
        bool found = false;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (h) { found = true; a.source = "shared: already mapped"; break; }
        }
        if (!found && dlsym(RTLD_DEFAULT, "ncclCommInitRank") && dlsym(RTLD_DEFAULT, "ncclSend")) {
            h = RTLD_DEFAULT;
            found = true;
            a.source = "shared: process symbols";
        }
        if (!found) {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) { found = true; a.source = "own: dlopen"; }
        }
        if (!found) return;
        bool all = true;
        auto sym = [&](const char* n) { void* p = dlsym(h, n); all = all && p; return p; };
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
        a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
        a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        a.ok = all;
    });
    return a;
}

struct MfComm {
    RcclComm comm;
    int world, rank;
};

int rccl_fail(const char* what, int rc) {
    return mf_set_error(MF_ELAUNCH, "%s: RCCL error %d (%s)", what, rc, api().GetErrorString ? api().GetErrorString(rc) : "?");
}

}  // namespace

extern "C" int mf_comm_unique_id(void* out128) {
    if (!out128) return mf_set_error(MF_EINVAL, "mf_comm_unique_id: bad argument");
    if (!api().ok) return mf_set_error(MF_ENOTSUP, "mf_comm_unique_id: librccl.so.1 could not be opened");
    RcclUniqueId id;
    const int rc = api().GetUniqueId(&id);
    if (rc) return rccl_fail("mf_comm_unique_id", rc);
    memcpy(out128, id.internal, 128);
    return MF_OK;
}

extern "C" int mf_comm_create(int world, int rank, const void* id128, void** out_comm) {
    if (world <= 0 || rank < 0 || rank >= world || !id128 || !out_comm) return mf_set_error(MF_EINVAL, "mf_comm_create: bad argument");
    if (!api().ok) return mf_set_error(MF_ENOTSUP, "mf_comm_create: librccl.so.1 could not be opened");
    RcclUniqueId id;
    memcpy(id.internal, id128, 128);
    RcclComm c = nullptr;
    const int rc = api().CommInitRank(&c, world, id, rank);       // collective: every rank of the job calls it
    if (rc) return rccl_fail("mf_comm_create", rc);
    *out_comm = new MfComm{c, world, rank};
    return MF_OK;
}

extern "C" int mf_comm_destroy(void* comm) {
    if (!comm) return MF_OK;
    MfComm* c = static_cast<MfComm*>(comm);
    const int rc = api().CommDestroy(c->comm);
    delete c;
    return rc ? rccl_fail("mf_comm_destroy", rc) : MF_OK;
}

extern "C" int mf_comm_world(void* comm) { return comm ? static_cast<MfComm*>(comm)->world : 0; }

// where the RCCL entry points came from ("shared: ..." = the copy torch had already mapped; "own: dlopen"; "not loaded")
extern "C" const char* mf_comm_source(void) { return api().ok ? api().source : "not loaded"; }

// Direct all-to-all of row blocks: rank r receives recv_rows_host[p] rows of row_bytes bytes from every peer p (laid out in
// peer order in `recv`) and sends send_rows_host[p] rows to it (peer order in `send`).  The counts live on the HOST (RCCL
// needs them there); no synchronisation, everything is enqueued on `stream`.
extern "C" int mf_comm_all_to_all_rows(void* comm, const void* send, const int64_t* send_rows_host, void* recv,
                                       const int64_t* recv_rows_host, int64_t row_bytes, mf_stream_t stream) {
    if (!comm || !send_rows_host || !recv_rows_host || row_bytes <= 0) return mf_set_error(MF_EINVAL, "mf_comm_all_to_all_rows: bad argument");
    MfComm* c = static_cast<MfComm*>(comm);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = api().GroupStart();
    if (rc) return rccl_fail("mf_comm_all_to_all_rows", rc);
    int64_t so = 0, ro = 0;
    for (int p = 0; p < c->world && !rc; ++p) {
        const int64_t sb = send_rows_host[p] * row_bytes, rb = recv_rows_host[p] * row_bytes;
        if (sb < 0 || rb < 0) { rc = -1; break; }
        if (p == c->rank) {
            // the block a rank keeps for itself never enters RCCL: a plain device copy on the same stream (a generic
            // RCCL kernel for a self send / recv pair costs 20 .. 100 us; at one rank that is the whole exchange)
            if (sb != rb) { rc = -1; break; }
            if (sb > 0 && hipMemcpyAsync(static_cast<char*>(recv) + ro, static_cast<const char*>(send) + so, (size_t)sb,
                                         hipMemcpyDeviceToDevice, s) != hipSuccess) { rc = -1; break; }
        } else {
            if (sb > 0) rc = api().Send(static_cast<const char*>(send) + so, (size_t)sb, RCCL_CHAR, p, c->comm, s);
            if (!rc && rb > 0) rc = api().Recv(static_cast<char*>(recv) + ro, (size_t)rb, RCCL_CHAR, p, c->comm, s);
        }
        so += sb;
        ro += rb;
    }
    const int rc2 = api().GroupEnd();
    if (rc || rc2) return rccl_fail("mf_comm_all_to_all_rows", rc ? rc : rc2);
    return MF_OK;
}

// every rank contributes `bytes` bytes; `recv` holds world * bytes in rank order
extern "C" int mf_comm_all_gather(void* comm, const void* send, void* recv, int64_t bytes, mf_stream_t stream) {
    if (!comm || !send || !recv || bytes <= 0) return mf_set_error(MF_EINVAL, "mf_comm_all_gather: bad argument");
    MfComm* c = static_cast<MfComm*>(comm);
    if (c->world == 1) {
        if (send != recv && hipMemcpyAsync(recv, send, (size_t)bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)) != hipSuccess)
            return mf_set_error(MF_ELAUNCH, "mf_comm_all_gather: copy failed");
        return MF_OK;
    }
    const int rc = api().AllGather(send, recv, (size_t)bytes, RCCL_CHAR, c->comm, static_cast<hipStream_t>(stream));
    return rc ? rccl_fail("mf_comm_all_gather", rc) : MF_OK;
}

// ------------------------------------------------------------- shard initialisation ----
// Row r of a (virtual) table of `std`-scaled normal variates, as a pure function of (seed, global row, column): a shard is
// filled ON ITS OWN DEVICE (no host copy of the whole table -- 102 GB for 100 M x 256) and the values do not depend on the
// number of ranks.  Counter-based: SplitMix64 of the element's index, two 32-bit uniforms, Box-Muller (spec restated in
// oracle/embed.py init_rows; tolerance 1e-6: log / cos differ in the last bits between libm and the device).
__global__ __launch_bounds__(256) void init_rows_kernel(float* __restrict__ table, int64_t n_local, int d, int64_t row_start,
                                                        int64_t row_stride, unsigned long long seed, float std) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_local * d) return;
    const int64_t lr = t / d;
    const int c = (int)(t % d);
    const unsigned long long row = (unsigned long long)(row_start + lr * row_stride);
    const unsigned long long z = mf_splitmix64(seed + (row * (unsigned long long)d + (unsigned long long)c) * 0x9E3779B97F4A7C15ull);
    const float u1 = ((float)(unsigned)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);        // (0, 1): 24 bits
    const float u2 = ((float)(unsigned)((z >> 16) & 0xFFFFFFu) + 0.5f) * (1.0f / 16777216.0f);
    table[t] = std * sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

extern "C" int mf_init_rows(float* table, int64_t n_local, int d, int64_t row_start, int64_t row_stride, uint64_t seed,
                            float std, mf_stream_t stream) {
    if (!table || n_local < 0 || d <= 0 || row_stride <= 0 || row_start < 0) return mf_set_error(MF_EINVAL, "mf_init_rows: bad argument");
    if (n_local == 0) return MF_OK;
    init_rows_kernel<<<dim3((unsigned)((n_local * d + 255) / 256)), 256, 0, static_cast<hipStream_t>(stream)>>>(
        table, n_local, d, row_start, row_stride, seed, std);
    return mf_check_launch("mf_init_rows");
}
