// gather.hip -- the one exchange step of the multi-GPU path (BASELINE north_star: "RCCL over xGMI carries only
// the tiny done/reward gather").  One process per GPU; each process owns one engine = one contiguous shard of the env batch.
// Per step every rank contributes its packed {reward:i32, done:u8, lives:u8, pad:u16} records (TBX_BUF_PACKED, 8 B/env) to
// one ncclAllGather.  It replaces the pipes of pickled (ob, rew, done, info) tuples between the reference's N worker
// processes and the learner (baselines/baselines/common/vec_env/subproc_vec_env.py:63-74); frames are NOT gathered.
//
// RCCL is resolved with dlopen at tbx_gather_init, so a single-GPU user needs no librccl, and there is no PyTorch anywhere.
// The collective runs on an engine-owned communication stream: it is ordered after the step that produced the records and
// the next step is ordered after it, but the rasteriser the caller queues in between overlaps with it (64 KiB - 512 KiB per
// rank, latency-bound over xGMI).

#include "tbx_common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

struct GatherState {
    void* dl = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0, width = 0;      // width = records per rank in the gathered layout (>= N of every rank)
    hipStream_t stream = nullptr;             // communication stream
    hipEvent_t ready = nullptr, done = nullptr;
    bool pending = false;                     // a gather has been queued and nothing waited for it yet
    uint64_t* send = nullptr;                 // [width] padded copy of the local records when width != N
    uint64_t* out = nullptr;                  // [nranks][width] engine-owned result (TBX_BUF_GATHERED)
    double* scalar = nullptr;                 // device scalar for the max-reduction
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};

namespace {

const char* const RCCL_NAMES[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};

void* open_rccl(std::string& err)
{
    for (const char* name : RCCL_NAMES)
        if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) return h;
    err = std::string("librccl.so not found (") + dlerror() + ")";
    return nullptr;
}

template <class F>
bool sym(void* dl, const char* name, F& f, std::string& err)
{
    f = reinterpret_cast<F>(dlsym(dl, name));
    if (!f) err = std::string("librccl: missing symbol ") + name;
    return f != nullptr;
}

bool load_symbols(GatherState& g, std::string& err)
{
    return sym(g.dl, "ncclGetUniqueId", g.get_unique_id, err) && sym(g.dl, "ncclCommInitRank", g.comm_init_rank, err) &&
           sym(g.dl, "ncclCommDestroy", g.comm_destroy, err) && sym(g.dl, "ncclAllGather", g.all_gather, err) &&
           sym(g.dl, "ncclAllReduce", g.all_reduce, err) && sym(g.dl, "ncclGetErrorString", g.error_string, err);
}

#define GHIP(call)                                                                                        \
    do {                                                                                                  \
        hipError_t _e = (call);                                                                           \
        if (_e != hipSuccess) return e->fail(TBX_E_NO_DEVICE, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define GNCCL(call)                                                                                       \
    do {                                                                                                  \
        ncclResult_t _r = (call);                                                                         \
        if (_r != ncclSuccess) return e->fail(TBX_E_NO_DEVICE, std::string(#call) + ": " + g.error_string(_r)); \
    } while (0)

__global__ void pad_records_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, int n, int width)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < width) dst[i] = i < n ? src[i] : 0ull;
}

}  // namespace

void tbx_gather_free(tbx_engine* e)
{
    GatherState* g = e->gather;
    if (!g) return;
    if (g->stream) hipStreamSynchronize(g->stream);
    if (g->comm && g->comm_destroy) g->comm_destroy(g->comm);
    if (g->ready) hipEventDestroy(g->ready);
    if (g->done) hipEventDestroy(g->done);
    if (g->stream) hipStreamDestroy(g->stream);
    hipFree(g->send); hipFree(g->out); hipFree(g->scalar);
    if (g->dl) dlclose(g->dl);
    delete g;
    e->gather = nullptr;
}

// steps overwrite the records a queued gather still has to read
hipError_t tbx_gather_before_step(tbx_engine* e, hipStream_t s)
{
    GatherState* g = e->gather;
    if (!g || !g->pending) return hipSuccess;
    g->pending = false;
    return hipStreamWaitEvent(s, g->done, 0);
}

int tbx_gather_buffer(tbx_engine* e, void** out_ptr, size_t* out_bytes)
{
    if (!e->gather) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called");
    *out_ptr = e->gather->out;
    if (out_bytes) *out_bytes = sizeof(uint64_t) * (size_t)e->gather->nranks * (size_t)e->gather->width;
    return TBX_OK;
}

extern "C" {

int tbx_gather_unique_id(void* id_out, size_t id_bytes)
{
    std::string err;
    if (!id_out || id_bytes != TBX_GATHER_ID_BYTES) { tbx_set_create_error("id buffer must be TBX_GATHER_ID_BYTES long"); return TBX_E_INVALID; }
    static_assert(TBX_GATHER_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    GatherState g;
    g.dl = open_rccl(err);
    if (!g.dl) { tbx_set_create_error(err); return TBX_E_UNSUPPORTED; }
    if (!load_symbols(g, err)) { dlclose(g.dl); tbx_set_create_error(err); return TBX_E_UNSUPPORTED; }
    ncclUniqueId id;
    const ncclResult_t r = g.get_unique_id(&id);
    if (r != ncclSuccess) { tbx_set_create_error(std::string("ncclGetUniqueId: ") + g.error_string(r)); return TBX_E_NO_DEVICE; }
    memcpy(id_out, &id, sizeof id);
    // the library stays loaded (the bootstrap listener of the id lives in it)
    return TBX_OK;
}

int tbx_gather_init(tbx_engine* e, int nranks, int rank, int records_per_rank, const void* id, size_t id_bytes)
{
    if (!e) return TBX_E_INVALID;
    if (nranks < 1 || rank < 0 || rank >= nranks) return e->fail(TBX_E_INVALID, "gather: rank / nranks out of range");
    if (records_per_rank < e->n) return e->fail(TBX_E_INVALID, "gather: records_per_rank must be >= the engine's env count");
    if (!id || id_bytes != TBX_GATHER_ID_BYTES) return e->fail(TBX_E_INVALID, "gather: id must be TBX_GATHER_ID_BYTES long");
    GHIP(hipSetDevice(e->device));
    GHIP(tbx_serve_stop(e));
    tbx_gather_free(e);
    e->gather = new GatherState();
    GatherState& g = *e->gather;
    std::string err;
    g.dl = open_rccl(err);
    if (!g.dl || !load_symbols(g, err)) { tbx_gather_free(e); return e->fail(TBX_E_UNSUPPORTED, err); }
    g.nranks = nranks; g.rank = rank; g.width = records_per_rank;
    GHIP(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    GHIP(hipEventCreateWithFlags(&g.ready, hipEventDisableTiming));
    GHIP(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
    GHIP(hipMalloc((void**)&g.out, sizeof(uint64_t) * (size_t)nranks * (size_t)g.width));
    GHIP(hipMemset(g.out, 0, sizeof(uint64_t) * (size_t)nranks * (size_t)g.width));
    GHIP(hipMalloc((void**)&g.scalar, sizeof(double)));
    if (g.width != e->n) GHIP(hipMalloc((void**)&g.send, sizeof(uint64_t) * (size_t)g.width));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    GNCCL(g.comm_init_rank(&g.comm, nranks, uid, rank));
    return TBX_OK;
}

int tbx_gather(tbx_engine* e, uint64_t* out_dev, void* stream)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called");
    GatherState& g = *e->gather;
    GHIP(hipSetDevice(e->device));
    // after everything queued through this handle so far (the step that wrote the records) ...
    hipStream_t prev = e->has_last ? e->last_stream : (hipStream_t)stream;
    GHIP(hipEventRecord(g.ready, prev));
    GHIP(hipStreamWaitEvent(g.stream, g.ready, 0));
    const uint64_t* send = e->packed;
    if (g.send) {
        hipLaunchKernelGGL(pad_records_kernel, dim3((g.width + 255) / 256), dim3(256), 0, g.stream, e->packed, g.send, e->n, g.width);
        GHIP(hipGetLastError());
        send = g.send;
    }
    GNCCL(g.all_gather(send, out_dev ? out_dev : g.out, (size_t)g.width, ncclUint64, g.comm, g.stream));
    // ... and before the next step (tbx_gather_before_step); what the caller queues next on its own stream overlaps
    GHIP(hipEventRecord(g.done, g.stream));
    g.pending = true;
    return TBX_OK;
}

int tbx_gather_wait(tbx_engine* e, void* stream)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called");
    GatherState& g = *e->gather;
    GHIP(hipSetDevice(e->device));
    GHIP(hipStreamWaitEvent((hipStream_t)stream, g.done, 0));
    return TBX_OK;
}

int tbx_gather_host(tbx_engine* e, uint64_t* out_host)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called");
    if (!out_host) return e->fail(TBX_E_INVALID, "output pointer is NULL");
    GatherState& g = *e->gather;
    GHIP(hipSetDevice(e->device));
    GHIP(hipMemcpyAsync(out_host, g.out, sizeof(uint64_t) * (size_t)g.nranks * (size_t)g.width, hipMemcpyDeviceToHost, g.stream));
    GHIP(hipStreamSynchronize(g.stream));
    return TBX_OK;
}

int tbx_gather_reduce_max(tbx_engine* e, double* inout_host)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called");
    if (!inout_host) return e->fail(TBX_E_INVALID, "value pointer is NULL");
    GatherState& g = *e->gather;
    GHIP(hipSetDevice(e->device));
    GHIP(hipMemcpyAsync(g.scalar, inout_host, sizeof(double), hipMemcpyHostToDevice, g.stream));
    GNCCL(g.all_reduce(g.scalar, g.scalar, 1, ncclFloat64, ncclMax, g.comm, g.stream));
    GHIP(hipMemcpyAsync(inout_host, g.scalar, sizeof(double), hipMemcpyDeviceToHost, g.stream));
    GHIP(hipStreamSynchronize(g.stream));
    return TBX_OK;
}

}  // extern "C"
