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
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <cstring>

// The handful of RCCL declarations this file needs, stated locally: the library is found with dlopen at run time and its
// header is not a build dependency.  Values are NCCL's stable public ABI (nccl.h / rccl.h: ncclUint64 = 5, ncclFloat64 = 8,
// ncclMax = 2, NCCL_UNIQUE_ID_BYTES = 128); where the header is present the build checks them.
typedef struct tbxNcclComm* tbx_nccl_comm_t;
typedef struct { char internal[TBX_GATHER_ID_BYTES]; } tbx_nccl_id_t;
enum { TBX_NCCL_SUCCESS = 0, TBX_NCCL_UINT64 = 5, TBX_NCCL_FLOAT64 = 8, TBX_NCCL_MAX = 2 };
typedef int (*tbx_nccl_get_unique_id_fn)(tbx_nccl_id_t*);
typedef int (*tbx_nccl_comm_init_rank_fn)(tbx_nccl_comm_t*, int, tbx_nccl_id_t, int);
typedef int (*tbx_nccl_comm_destroy_fn)(tbx_nccl_comm_t);
typedef int (*tbx_nccl_comm_count_fn)(tbx_nccl_comm_t, int*);
typedef int (*tbx_nccl_all_gather_fn)(const void*, void*, size_t, int, tbx_nccl_comm_t, hipStream_t);
typedef int (*tbx_nccl_all_reduce_fn)(const void*, void*, size_t, int, int, tbx_nccl_comm_t, hipStream_t);
typedef const char* (*tbx_nccl_error_string_fn)(int);
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert(TBX_GATHER_ID_BYTES == NCCL_UNIQUE_ID_BYTES && sizeof(tbx_nccl_id_t) == sizeof(ncclUniqueId), "unique id size");
static_assert((int)ncclSuccess == TBX_NCCL_SUCCESS && (int)ncclUint64 == TBX_NCCL_UINT64 && (int)ncclFloat64 == TBX_NCCL_FLOAT64 &&
              (int)ncclMax == TBX_NCCL_MAX, "RCCL enum values");
#endif

// ---- host transport.  Segment layout: seq[3][64] call counters (records / scalar / attach), two parity buffers of
// nranks * slot bytes for the records, two parity buffers of nranks doubles.  A collective: write the own slot of parity
// (call & 1), publish the call number, wait until every rank has published it, read all slots.  Two parities are enough: a rank
// can be at most one call ahead of the slowest one (it waits for everybody before it returns).
struct HostWorld {
    struct Head { volatile uint64_t seq[3][64]; };
    void* map = nullptr;
    size_t len = 0, slot = 0;
    int nranks = 1, rank = 0;
    uint64_t calls[2] = {0, 0};
    char name[64] = {0};
    uint64_t* stage_out = nullptr;    // pinned: this rank's records on their way out
    uint64_t* gathered = nullptr;     // pinned: [nranks][slot / 8] result of the last collective

    Head* head() const { return reinterpret_cast<Head*>(map); }
    char* records(int parity) const { return reinterpret_cast<char*>(map) + sizeof(Head) + (size_t)parity * nranks * slot; }
    double* scalars(int parity) const { return reinterpret_cast<double*>(reinterpret_cast<char*>(map) + sizeof(Head) + 2 * (size_t)nranks * slot) + (size_t)parity * nranks; }

    // false: a rank did not arrive within `seconds`
    bool wait_all(int which, uint64_t call, double seconds = 120.0) const
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < nranks; r++) {
            unsigned spins = 0;
            while (__atomic_load_n(&head()->seq[which][r], __ATOMIC_ACQUIRE) < call) {
                sched_yield();
                if ((++spins & 0xFFFu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
            }
        }
        return true;
    }
    bool exchange_records(const void* mine, void* all)
    {
        const uint64_t call = ++calls[0];
        char* buf = records((int)(call & 1u));
        memcpy(buf + (size_t)rank * slot, mine, slot);
        __atomic_store_n(&head()->seq[0][rank], call, __ATOMIC_RELEASE);
        if (!wait_all(0, call)) return false;
        memcpy(all, buf, (size_t)nranks * slot);
        return true;
    }
    bool max_scalar(double* v)
    {
        const uint64_t call = ++calls[1];
        double* buf = scalars((int)(call & 1u));
        buf[rank] = *v;
        __atomic_store_n(&head()->seq[1][rank], call, __ATOMIC_RELEASE);
        if (!wait_all(1, call)) return false;
        for (int r = 0; r < nranks; r++)
            if (buf[r] > *v) *v = buf[r];
        return true;
    }
};

struct GatherState {
    void* dl = nullptr;
    tbx_nccl_comm_t comm = nullptr;
    int nranks = 1, rank = 0, width = 0;      // width = records per rank in the gathered layout (>= N of every rank)
    int comm_count = 0;                       // what ncclCommCount said after ncclCommInitRank
    std::string lib_path;                     // the librccl that dlopen found
    hipStream_t stream = nullptr;             // communication stream
    hipEvent_t ready = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr};  // behind the last gather that read output set p (tbx_engine::outs) ...
    hipStream_t done_on[2] = {nullptr, nullptr};   // ... which ran on this stream
    bool pending[2] = {false, false};         // ... and no step has waited for it yet
    int last_par = 0;
    bool any = false;                         // a gather has been queued at all
    uint64_t* send = nullptr;                 // [width] padded copy of the local records when width != N
    // TBX_OPT_GATHER_EVERY = K > 1: the step kernels write their records straight into slot `fill` of ring[ring_par]
    // ([K][width], zero beyond the engine's envs), tbx_gather only counts, and the K-th call sends the whole ring with ONE
    // collective; the next K steps fill the other ring meanwhile
    int every = 1;
    int fill = 0;                             // steps whose records sit in the current ring and have not been sent
    int ring_par = 0;
    bool advance = false;                     // the next step has to move to the next slot first (tbx_gather_before_step)
    uint64_t* ring[2] = {nullptr, nullptr};
    uint64_t* out = nullptr;                  // [nranks][width] engine-owned result (TBX_BUF_GATHERED)
    double* scalar = nullptr;                 // device scalar for the max-reduction
    // TBX_OPT_GATHER_TRANSPORT = 1: no RCCL -- the ranks of ONE node exchange the records through a POSIX shared-memory segment
    // named after the id ("a host-staged gather is the fallback if RCCL is missing"); see HostWorld below
    HostWorld* host = nullptr;
    tbx_nccl_get_unique_id_fn get_unique_id = nullptr;
    tbx_nccl_comm_init_rank_fn comm_init_rank = nullptr;
    tbx_nccl_comm_destroy_fn comm_destroy = nullptr;
    tbx_nccl_comm_count_fn comm_count_fn = nullptr;
    tbx_nccl_all_gather_fn all_gather = nullptr;
    tbx_nccl_all_reduce_fn all_reduce = nullptr;
    tbx_nccl_error_string_fn error_string = nullptr;
};

namespace {

const char* const RCCL_NAMES[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};

void* open_rccl(std::string& err, std::string* path = nullptr)
{
    for (const char* name : RCCL_NAMES)
        if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
            if (path) {
                *path = name;
                Dl_info info;
                if (void* f = dlsym(h, "ncclAllGather"))
                    if (dladdr(f, &info) && info.dli_fname) *path = info.dli_fname;
            }
            return h;
        }
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
           sym(g.dl, "ncclCommDestroy", g.comm_destroy, err) && sym(g.dl, "ncclCommCount", g.comm_count_fn, err) &&
           sym(g.dl, "ncclAllGather", g.all_gather, err) &&
           sym(g.dl, "ncclAllReduce", g.all_reduce, err) && sym(g.dl, "ncclGetErrorString", g.error_string, err);
}

#define GHIP(call)                                                                                        \
    do {                                                                                                  \
        hipError_t _e = (call);                                                                           \
        if (_e != hipSuccess) return e->fail(TBX_E_NO_DEVICE, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define GNCCL(call)                                                                                       \
    do {                                                                                                  \
        int _r = (call);                                                                                  \
        if (_r != TBX_NCCL_SUCCESS) return e->fail(TBX_E_NO_DEVICE, std::string(#call) + ": " + g.error_string(_r)); \
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
    if (g->host) {
        if (g->host->map) munmap(g->host->map, g->host->len);
        if (g->host->rank == 0 && g->host->name[0]) shm_unlink(g->host->name);
        if (g->host->stage_out) hipHostFree(g->host->stage_out);
        if (g->host->gathered) hipHostFree(g->host->gathered);
        delete g->host;
    }
    if (g->comm && g->comm_destroy) g->comm_destroy(g->comm);
    if (g->ready) hipEventDestroy(g->ready);
    for (int k = 0; k < 2; k++)
        if (g->done[k]) hipEventDestroy(g->done[k]);
    if (g->stream) hipStreamDestroy(g->stream);
    hipFree(g->send); hipFree(g->out); hipFree(g->scalar);
    if (g->ring[0]) {                         // the step kernels go back to the engine's own record array
        hipFree(g->ring[0]); hipFree(g->ring[1]);
        e->packed = e->outs[e->out_par].packed;
        if (e->ops) e->ops->rebind_outputs(e);
    }
    e->gather_ring = false;
    e->gather_wants_step_event = false;
    if (g->dl) dlclose(g->dl);
    delete g;
    e->gather = nullptr;
}

// a step overwrites the records of the output set it writes (tbx_engine::out_par at the time of this call): a queued gather
// that still has to read that set goes first.  Ring mode: the step moves on to the next slot of the ring (the address only --
// no stream operation), and only when it opens a ring that was sent 2 K steps ago does it wait for that collective.
hipError_t tbx_gather_before_step(tbx_engine* e, hipStream_t s)
{
    GatherState* g = e->gather;
    if (!g) return hipSuccess;
    if (g->every > 1) {
        if (!g->advance) return hipSuccess;                 // a step without a tbx_gather since the last one rewrites its slot
        g->advance = false;
        e->packed = g->ring[g->ring_par] + (size_t)g->fill * (size_t)g->width;
        e->ops->rebind_outputs(e);
        e->gather_wants_step_event = g->fill == g->every - 1;   // the step that completes the ring is the one the collective waits for
        const int p = g->ring_par;
        if (g->fill != 0 || !g->pending[p]) return hipSuccess;
        g->pending[p] = false;
        if (g->done_on[p] == s) return hipSuccess;
        return hipStreamWaitEvent(s, g->done[p], 0);
    }
    const int p = e->out_par;
    if (!g->pending[p]) return hipSuccess;
    g->pending[p] = false;
    if (g->done_on[p] == s) return hipSuccess;          // the same stream: already in order
    return hipStreamWaitEvent(s, g->done[p], 0);
}

int tbx_gather_buffer(tbx_engine* e, void** out_ptr, size_t* out_bytes)
{
    if (!e->gather) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called");
    *out_ptr = e->gather->out;
    if (out_bytes) *out_bytes = sizeof(uint64_t) * (size_t)e->gather->nranks * (size_t)e->gather->every * (size_t)e->gather->width;
    return TBX_OK;
}

extern "C" {

int tbx_gather_unique_id(void* id_out, size_t id_bytes)
{
    std::string err;
    if (!id_out || id_bytes != TBX_GATHER_ID_BYTES) { tbx_set_create_error("id buffer must be TBX_GATHER_ID_BYTES long"); return TBX_E_INVALID; }
    GatherState g;
    g.dl = open_rccl(err);
    if (!g.dl) {
        // no librccl on this machine: an id that names a host-transport world (TBX_OPT_GATHER_TRANSPORT = 1) and nothing else
        int fd = open("/dev/urandom", O_RDONLY);
        const bool ok = fd >= 0 && read(fd, id_out, TBX_GATHER_ID_BYTES) == (ssize_t)TBX_GATHER_ID_BYTES;
        if (fd >= 0) close(fd);
        if (!ok) { tbx_set_create_error(err + "; and /dev/urandom is not readable"); return TBX_E_UNSUPPORTED; }
        return TBX_OK;
    }
    if (!load_symbols(g, err)) { dlclose(g.dl); tbx_set_create_error(err); return TBX_E_UNSUPPORTED; }
    tbx_nccl_id_t id;
    const int r = g.get_unique_id(&id);
    if (r != TBX_NCCL_SUCCESS) { tbx_set_create_error(std::string("ncclGetUniqueId: ") + g.error_string(r)); return TBX_E_NO_DEVICE; }
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
    const bool host_transport = e->opt[TBX_OPT_GATHER_TRANSPORT] == 1;
    if (host_transport) {
        if (nranks > 64) { tbx_gather_free(e); return e->fail(TBX_E_UNSUPPORTED, "gather: the host transport handles at most 64 ranks"); }
        g.lib_path = "host: POSIX shared memory, staged through page-locked buffers (no RCCL)";
    } else {
        g.dl = open_rccl(err, &g.lib_path);
        if (!g.dl || !load_symbols(g, err)) { tbx_gather_free(e); return e->fail(TBX_E_UNSUPPORTED, err); }
    }
    g.nranks = nranks; g.rank = rank; g.width = records_per_rank;
    g.every = e->opt[TBX_OPT_GATHER_EVERY] > 1 ? e->opt[TBX_OPT_GATHER_EVERY] : 1;
#ifdef TBX_DIAG
    if (getenv("TBX_GATHER_PRIORITY") && atoi(getenv("TBX_GATHER_PRIORITY"))) {      // measurement builds: the communication stream in the high-priority pool
        int lo = 0, hi = 0;
        GHIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        GHIP(hipStreamCreateWithPriority(&g.stream, hipStreamNonBlocking, hi));
    } else
#endif
    GHIP(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    GHIP(hipEventCreateWithFlags(&g.ready, hipEventDisableTiming));
    GHIP(hipEventCreateWithFlags(&g.done[0], hipEventDisableTiming));
    GHIP(hipEventCreateWithFlags(&g.done[1], hipEventDisableTiming));
    const size_t out_records = (size_t)nranks * (size_t)g.every * (size_t)g.width;
    GHIP(hipMalloc((void**)&g.out, sizeof(uint64_t) * out_records));
    GHIP(hipMemset(g.out, 0, sizeof(uint64_t) * out_records));
    GHIP(hipMalloc((void**)&g.scalar, sizeof(double)));
    if (g.every > 1) {
        // the engine must be idle while the step kernels' record pointer moves into the ring
        GHIP(hipDeviceSynchronize());
        e->has_last = false;
        e->pipe.active = false;
        const size_t ring_bytes = sizeof(uint64_t) * (size_t)g.every * (size_t)g.width;
        for (int k = 0; k < 2; k++) {
            GHIP(hipMalloc((void**)&g.ring[k], ring_bytes));
            GHIP(hipMemset(g.ring[k], 0, ring_bytes));      // slots are `width` wide: what lies beyond the engine's envs stays 0
        }
        // slot 0 starts as a copy of the records of the step before (TBX_BUF_PACKED keeps reading what it read)
        GHIP(hipMemcpy(g.ring[0], e->packed, sizeof(uint64_t) * (size_t)e->n, hipMemcpyDeviceToDevice));
        e->packed = g.ring[0];
        e->ops->rebind_outputs(e);
        e->gather_ring = true;
        e->gather_ring_every = g.every;
        e->gather_ring_width = g.width;
        e->gather_wants_step_event = g.every == 1;
    } else {
        e->gather_wants_step_event = true;
        if (g.width != e->n) GHIP(hipMalloc((void**)&g.send, sizeof(uint64_t) * (size_t)g.width));
    }
    if (host_transport) {
        HostWorld* h = g.host = new HostWorld();
        h->nranks = nranks; h->rank = rank;
        h->slot = sizeof(uint64_t) * (size_t)g.every * (size_t)g.width;
        h->len = sizeof(HostWorld::Head) + 2 * (size_t)nranks * h->slot + 2 * (size_t)nranks * sizeof(double);
        uint64_t fnv = 1469598103934665603ull;                              // the segment's name: a hash of the 128 id bytes
        for (size_t i = 0; i < TBX_GATHER_ID_BYTES; i++) fnv = (fnv ^ ((const uint8_t*)id)[i]) * 1099511628211ull;
        snprintf(h->name, sizeof h->name, "/tbx_hg_%016llx", (unsigned long long)fnv);
        if (hipHostMalloc((void**)&h->stage_out, h->slot, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void**)&h->gathered, (size_t)nranks * h->slot, hipHostMallocDefault) != hipSuccess) {
            h->name[0] = 0;                                     // (no segment has been made under this name yet)
            tbx_gather_free(e);
            return e->fail(TBX_E_NOMEM, "gather: page-locked staging buffers of the host transport");
        }
        memset(h->stage_out, 0, h->slot);
        memset(h->gathered, 0, (size_t)nranks * h->slot);
        // every failure from here on takes the half-made world down with it (tbx_gather_free: staging buffers, the mapping, the
        // segment's name), so that a later tbx_gather finds no communicator instead of a HostWorld without a map
        auto bail = [&](const char* msg) { tbx_gather_free(e); return e->fail(TBX_E_NO_DEVICE, msg); };
        const int fd = shm_open(h->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0) { h->name[0] = 0; return bail("gather: shm_open failed for the host transport"); }
        if (ftruncate(fd, (off_t)h->len) != 0) { close(fd); return bail("gather: ftruncate failed"); }
        void* m = mmap(nullptr, h->len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);     // a fresh segment reads as zeros
        close(fd);
        if (m == MAP_FAILED) return bail("gather: mmap failed");
        h->map = m;
        // collective like ncclCommInitRank: returns once every rank has attached
        __atomic_store_n(&h->head()->seq[2][rank], (uint64_t)1, __ATOMIC_RELEASE);
        if (!h->wait_all(2, 1)) return bail("gather: not every rank attached to the host transport's segment");
        g.comm_count = nranks;
        return TBX_OK;
    }
    tbx_nccl_id_t uid;
    memcpy(&uid, id, sizeof uid);
    int nr = g.comm_init_rank(&g.comm, nranks, uid, rank);
    if (nr == TBX_NCCL_SUCCESS) nr = g.comm_count_fn(g.comm, &g.comm_count);
    if (nr != TBX_NCCL_SUCCESS) {
        const std::string msg = std::string("ncclCommInitRank / ncclCommCount: ") + g.error_string(nr);
        tbx_gather_free(e);
        return e->fail(TBX_E_NO_DEVICE, msg);
    }
    if (g.comm_count != nranks) { tbx_gather_free(e); return e->fail(TBX_E_NO_DEVICE, "gather: the communicator does not span the ranks asked for"); }
    return TBX_OK;
}

// the host transport's collective: records device -> page-locked memory on the communication stream, the host WAITS for them
// (so the step that wrote them has finished), exchange through the segment, gathered block -> device.  The caller's thread is
// blocked for the length of the step + the exchange: a fallback and a dress rehearsal, not the product's data path.
static int host_collective(tbx_engine* e, GatherState& g, const uint64_t* send_dev, uint64_t* out_dev, hipStream_t gs)
{
    HostWorld& h = *g.host;
    GHIP(hipMemcpyAsync(h.stage_out, send_dev, h.slot, hipMemcpyDeviceToHost, gs));
    GHIP(hipStreamSynchronize(gs));
    if (!h.exchange_records(h.stage_out, h.gathered)) return e->fail(TBX_E_NO_DEVICE, "gather: a rank did not reach the collective (host transport)");
    GHIP(hipMemcpyAsync(out_dev, h.gathered, (size_t)h.nranks * h.slot, hipMemcpyHostToDevice, gs));
    return TBX_OK;
}

// the communication stream behind the step(s) whose records the collective reads.  Overlapped fused launches (TbxPipe::fused): the
// completion events of both lanes -- the last launch and the one before it, which may still be painting but whose step blocks
// the last launch waited for; pipelined mode: the step stream's event; otherwise the tail of the handle.
static hipError_t wait_for_steps(tbx_engine* e, hipStream_t gs)
{
    const TbxPipe& p = e->pipe;
    if (p.active && p.fused) {
        for (int k = 0; k < 2; k++)
            if (p.launch_rec[k]) {
                hipError_t r = hipStreamWaitEvent(gs, p.launch_ev[k], 0);
                if (r != hipSuccess) return r;
            }
        return hipSuccess;
    }
    if (p.active && p.step_outstanding) return hipStreamWaitEvent(gs, p.step_ev, 0);
    return tbx_wait_tail(e, gs, true);
}

int tbx_gather(tbx_engine* e, uint64_t* out_dev, void* stream)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather || (!e->gather->comm && !(e->gather->host && e->gather->host->map))) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called (or failed)");
    GatherState& g = *e->gather;
    GHIP(hipSetDevice(e->device));
    // After the step that wrote the records, on the communication stream, so that the rasteriser the caller queues next overlaps
    // with it.  Pipelined mode: behind the step stream's event (the caller's stream, which calls of other kinds order themselves
    // behind, also waits for the previous frame's rasteriser).  Running the collective on the step's own internal stream instead
    // was measured (scripts/pipeline_sweep.py with PS_GATHER=1, one-rank communicator): 0.26 ms per step at 4 096 envs
    // against 0.10 -- the lanes then wait for each other's collectives.
    (void)stream;
    hipStream_t gs = g.stream;
    if (g.every > 1) {
        // ring mode: the records of the last step already sit in slot `fill` of the current ring
        g.advance = true;
        if (++g.fill < g.every) return TBX_OK;              // nothing is queued: K - 1 of K calls cost no stream operation at all
        const int p = g.ring_par;
        GHIP(wait_for_steps(e, gs));
        if (g.any && g.done_on[g.last_par] != gs) GHIP(hipStreamWaitEvent(gs, g.done[g.last_par], 0));
        if (g.host) {
            int rc = host_collective(e, g, g.ring[p], out_dev ? out_dev : g.out, gs);
            if (rc) return rc;
        } else
            GNCCL(g.all_gather(g.ring[p], out_dev ? out_dev : g.out, (size_t)g.every * (size_t)g.width, TBX_NCCL_UINT64, g.comm, gs));
        g.last_par = p;
        GHIP(hipEventRecord(g.done[p], gs));
        g.done_on[p] = gs;
        g.pending[p] = true;                                // the step that re-opens this ring, 2 K steps from now, waits for it
        g.any = true;
        g.fill = 0;
        g.ring_par = p ^ 1;
        return TBX_OK;
    }
    GHIP(wait_for_steps(e, gs));
    if (g.any && g.done_on[g.last_par] != gs) GHIP(hipStreamWaitEvent(gs, g.done[g.last_par], 0));
    const uint64_t* send = e->packed;
    if (g.send) {
        hipLaunchKernelGGL(pad_records_kernel, dim3((g.width + 255) / 256), dim3(256), 0, gs, e->packed, g.send, e->n, g.width);
        GHIP(hipGetLastError());
        send = g.send;
    }
    if (g.host) {
        int rc = host_collective(e, g, send, out_dev ? out_dev : g.out, gs);
        if (rc) return rc;
    } else
        GNCCL(g.all_gather(send, out_dev ? out_dev : g.out, (size_t)g.width, TBX_NCCL_UINT64, g.comm, gs));
    // ... and before the next step that rewrites these records (tbx_gather_before_step)
    g.last_par = e->out_par;
    GHIP(hipEventRecord(g.done[g.last_par], gs));
    g.done_on[g.last_par] = gs;
    g.pending[g.last_par] = true;
    g.any = true;
    return TBX_OK;
}

int tbx_gather_wait(tbx_engine* e, void* stream)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called");
    GatherState& g = *e->gather;
    GHIP(hipSetDevice(e->device));
    GHIP(hipStreamWaitEvent((hipStream_t)stream, g.done[g.last_par], 0));
    return TBX_OK;
}

int tbx_gather_host(tbx_engine* e, uint64_t* out_host)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called");
    if (!out_host) return e->fail(TBX_E_INVALID, "output pointer is NULL");
    GatherState& g = *e->gather;
    GHIP(hipSetDevice(e->device));
    if (g.any) GHIP(hipStreamWaitEvent(g.stream, g.done[g.last_par], 0));
    GHIP(hipMemcpyAsync(out_host, g.out, sizeof(uint64_t) * (size_t)g.nranks * (size_t)g.every * (size_t)g.width, hipMemcpyDeviceToHost, g.stream));
    GHIP(hipStreamSynchronize(g.stream));
    return TBX_OK;
}

int tbx_gather_reduce_max(tbx_engine* e, double* inout_host)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called");
    if (!inout_host) return e->fail(TBX_E_INVALID, "value pointer is NULL");
    GatherState& g = *e->gather;
    if (!g.comm && !(g.host && g.host->map)) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called (or failed)");
    GHIP(hipSetDevice(e->device));
    if (g.host) {
        if (!g.host->max_scalar(inout_host)) return e->fail(TBX_E_NO_DEVICE, "gather: a rank did not reach the reduction (host transport)");
        return TBX_OK;
    }
    if (g.any) GHIP(hipStreamWaitEvent(g.stream, g.done[g.last_par], 0));     // one collective of a communicator at a time
    GHIP(hipMemcpyAsync(g.scalar, inout_host, sizeof(double), hipMemcpyHostToDevice, g.stream));
    GNCCL(g.all_reduce(g.scalar, g.scalar, 1, TBX_NCCL_FLOAT64, TBX_NCCL_MAX, g.comm, g.stream));
    GHIP(hipMemcpyAsync(inout_host, g.scalar, sizeof(double), hipMemcpyDeviceToHost, g.stream));
    GHIP(hipStreamSynchronize(g.stream));
    return TBX_OK;
}

/* how many ranks the communicator spans as RCCL itself reports it (ncclCommCount), and the library that was loaded */
int tbx_gather_nranks(tbx_engine* e)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather || (!e->gather->comm && !e->gather->host)) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called");
    return e->gather->comm_count;
}

const char* tbx_gather_library(tbx_engine* e)
{
    if (!e || !e->gather) return "";
    return e->gather->lib_path.c_str();
}

int tbx_gather_every(tbx_engine* e)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called");
    return e->gather->every;
}

int tbx_gather_fill(tbx_engine* e)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather) return e->fail(TBX_E_INVALID, "tbx_gather_init has not been called");
    return e->gather->fill;
}

}  // extern "C"

// ---- a whole ring filled by ONE launch (tbx_rollout_synthetic, engine.hip)

int tbx_gather_ring_open(tbx_engine* e, hipStream_t s, int k, uint64_t** base, size_t* stride)
{
    GatherState* g = e->gather;
    if (!g || g->every < 2) return e->fail(TBX_E_INVALID, "no K-step record ring is in force");
    if (g->every != k) return e->fail(TBX_E_INVALID, "tbx_rollout_synthetic: the chunk must be as long as the gather's record ring (TBX_OPT_GATHER_EVERY)");
    if (g->fill != 0) return e->fail(TBX_E_INVALID, "tbx_rollout_synthetic: the record ring is partly filled (finish it with single steps + tbx_gather)");
    const int p = g->ring_par;
    g->advance = false;
    *base = g->ring[p];
    *stride = (size_t)g->width;
    if (g->pending[p]) {                                    // the collective that read this ring, 2 K steps ago
        g->pending[p] = false;
        if (g->done_on[p] != s) GHIP(hipStreamWaitEvent(s, g->done[p], 0));
    }
    return TBX_OK;
}

int tbx_gather_ring_filled(tbx_engine* e, hipEvent_t filled_ev)
{
    GatherState& g = *e->gather;
    hipStream_t gs = g.stream;
    const int p = g.ring_par;
    GHIP(hipStreamWaitEvent(gs, filled_ev, 0));             // the step launch alone: the chunk's rasterisers run beside the collective
    if (g.any && g.done_on[g.last_par] != gs) GHIP(hipStreamWaitEvent(gs, g.done[g.last_par], 0));
    if (g.host) {
        int rc = host_collective(e, g, g.ring[p], g.out, gs);
        if (rc) return rc;
    } else
        GNCCL(g.all_gather(g.ring[p], g.out, (size_t)g.every * (size_t)g.width, TBX_NCCL_UINT64, g.comm, gs));
    g.last_par = p;
    GHIP(hipEventRecord(g.done[p], gs));
    g.done_on[p] = gs;
    g.pending[p] = true;
    g.any = true;
    g.fill = 0;
    g.advance = true;                                       // a single step that follows opens slot 0 of the other ring
    g.ring_par = p ^ 1;
    e->gather_wants_step_event = false;
    return TBX_OK;
}
