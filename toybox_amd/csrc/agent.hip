// agent.hip -- the agent-side wrapper stack fused on the device: frame-skip with reward
// sum, max over the last two frames, area resize of the gray frame, reward clipping and the rolling frame stack
// (baselines/baselines/common/atari_wrappers.py:193-244, common/vec_env/vec_frame_stack.py:17-30), with the VecEnv
// auto-reset.  Generic over the games: it drives the game's own step / new_game / gray render launches and adds
// three small kernels of its own.  Only the stacked observation (out_h x out_w x stack bytes per env, 28 KB at
// 84x84x4) is ever meant to leave the chip; the two full-resolution gray frames are scratch in HBM.

#include "tbx_common.hpp"
#include "agent_device.hpp"

#include <cstdlib>
#include <cstring>
#include <vector>

struct AgentState {
    tbx_agent_config_t cfg{};
    int H = 0, W = 0;
    uint8_t *gray_a = nullptr, *gray_b = nullptr, *obs = nullptr, *fin = nullptr, *done_out = nullptr;
    int32_t* racc = nullptr;
    float* reward_out = nullptr;
    AgentTaps *ty = nullptr, *tx = nullptr;
    // reset-time wrappers + episode monitor
    uint8_t *kind = nullptr, *ep_done = nullptr;
    int32_t *ep_ret = nullptr, *ep_len = nullptr, *ep_index = nullptr, *prev_lives = nullptr, *ep_len_out = nullptr;
    float* ep_ret_out = nullptr;
    int32_t* reset_list = nullptr;   // [N] envs the monitor kernel flagged for a reset
    int32_t* reset_count = nullptr;  // [2] their number, double-buffered by step parity
    int parity = 0;
    bool force_generic = false;   // TBX_AGENT_GENERIC=1: always go through full-resolution gray frames
};

namespace {

constexpr int MAX_TAPS = 8;

// after the `skip` frames: Monitor bookkeeping, EpisodicLifeEnv's done rule, clipped reward, and what kind of reset the
// env needs (0 none, 1 life lost, 2 game over).  simple: no in-kernel reset follows, so a finished episode's counters
// are cleared here.
__global__ void agent_monitor_kernel(const int32_t* racc, const uint8_t* fin, const int32_t* lives, int32_t* ep_ret, int32_t* ep_len,
                                     int32_t* ep_index, int32_t* prev_lives, uint8_t* kind, uint8_t* ep_done, float* ep_ret_out,
                                     int32_t* ep_len_out, float* reward_out, uint8_t* done_out, int32_t* list, int32_t* count,
                                     int32_t* next_count, int episodic, int clip, int simple, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *next_count = 0;                         // the other step parity's counter: its readers ran a step ago
    if (i >= n) return;
    const int r = racc[i];
    int er = ep_ret[i] + r, el = ep_len[i] + 1;
    const bool real = fin[i] != 0;
    const int l = lives[i];
    const bool life_lost = episodic && !real && l < prev_lives[i] && l > 0;
    kind[i] = real ? 2 : life_lost ? 1 : 0;
    if (real || life_lost) list[atomicAdd(count, 1)] = i;  // order is irrelevant: the reset of one env touches nothing else
    done_out[i] = (real || life_lost) ? 1 : 0;
    reward_out[i] = clip ? (float)((r > 0) - (r < 0)) : (float)r;
    ep_done[i] = real ? 1 : 0;
    if (real) {
        ep_ret_out[i] = (float)er; ep_len_out[i] = el;
        if (simple) { er = 0; el = 0; ep_index[i] += 1; }
    }
    ep_ret[i] = er; ep_len[i] = el;
    prev_lives[i] = l;
}

__global__ void agent_fill_u8_kernel(uint8_t* p, uint8_t v, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// One wave per env, one pass over the SOURCE rows: each row (max of the two frames unless the env was just reset)
// is staged in LDS once, every lane takes the horizontal area sums of its (up to two) output columns with two
// v_dot4_u32_u8, and adds them with the row's vertical weights into the current / next output-row accumulators
// (a source row overlaps at most two output rows because out_h <= H).  A finished output row is normalised with a
// multiply-shift reciprocal (exact for sums < 2^25) and rolled into the env's frame stack.
template <int S>
__global__ __launch_bounds__(TBX_BLOCK) void agent_warp_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                               const uint8_t* __restrict__ fin, const AgentTaps* __restrict__ tx,
                                                               uint8_t* __restrict__ obs, int H, int W, int oh, int ow,
                                                               uint64_t magic, int reset_mode, int n)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[TBX_WAVES_PER_BLOCK][352];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int env = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + wave);
    if (env >= n) return;
    uint8_t* row = lds_all[wave];
    const bool fresh = reset_mode || fin[env];          // the observation is the (warped) reset frame alone
    const uint8_t* fa = A + (size_t)env * H * W;
    const uint8_t* fb = B + (size_t)env * H * W;
    uint8_t* o = obs + (size_t)env * oh * ow * S;
    const int words = W >> 2;
    const uint32_t half = (uint32_t)(H * W) / 2u;
    const ColTaps c0 = load_col(tx, lane, ow), c1 = load_col(tx, lane + 64, ow);
    const bool on0 = lane < ow, on1 = lane + 64 < ow;
    if (lane < 8) reinterpret_cast<uint32_t*>(row)[80 + lane] = 0u;   // padding read by the 12-byte windows

    // The kernel is bound by the latency of re-reading the two gray frames, so source rows are fetched PF at a time, one
    // group ahead of the group being reduced, and the frame-stack word of the NEXT output row is fetched while the
    // current one accumulates (S == 4; other depths roll byte-wise in stack_push).
    constexpr int PF = 4;
    uint32_t cur_a[PF][2], cur_b[PF][2], nxt_a[PF][2], nxt_b[PF][2];
    auto fetch = [&](int sy0, uint32_t (&va)[PF][2], uint32_t (&vb)[PF][2]) {
#pragma unroll
        for (int p = 0; p < PF; p++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int w4 = lane + 64 * q, sy = sy0 + p;
                va[p][q] = 0u; vb[p][q] = 0u;
                if (sy < H && w4 < words) {
                    vb[p][q] = *reinterpret_cast<const uint32_t*>(fb + (size_t)sy * W + 4 * w4);
                    if (!fresh) va[p][q] = *reinterpret_cast<const uint32_t*>(fa + (size_t)sy * W + 4 * w4);
                }
            }
    };
    uint32_t old[2] = {0u, 0u};                          // stack words of the output row being accumulated
    auto fetch_old = [&](int oy) {
        if (S == 4 && !fresh && oy < oh) {
            if (on0) old[0] = *reinterpret_cast<const uint32_t*>(o + ((size_t)oy * ow + lane) * 4);
            if (on1) old[1] = *reinterpret_cast<const uint32_t*>(o + ((size_t)oy * ow + lane + 64) * 4);
        }
    };
    fetch(0, cur_a, cur_b);
    fetch_old(0);

    uint32_t acc0[2] = {0, 0}, acc1[2] = {0, 0};        // [column slot]: current / next output row
    for (int sy0 = 0; sy0 < H; sy0 += PF) {
        fetch(sy0 + PF, nxt_a, nxt_b);
#pragma unroll
        for (int p = 0; p < PF; p++) {
            const int sy = sy0 + p;
            if (sy >= H) break;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int w4 = lane + 64 * q;
                if (w4 < words) reinterpret_cast<uint32_t*>(row)[w4] = fresh ? cur_b[p][q] : bytemax4(cur_a[p][q], cur_b[p][q]);
            }
            __builtin_amdgcn_wave_barrier();
            // vertical split of this source row (extent oh in refined units) over output rows oy and oy+1 (extent H each)
            const int oy = (sy * oh) / H;
            const int top = (oy + 1) * H;
            const int w_cur = min((sy + 1) * oh, top) - sy * oh, w_next = oh - w_cur;
            const uint32_t h0 = on0 ? hsum(row, c0) : 0u, h1 = on1 ? hsum(row, c1) : 0u;
            acc0[0] += (uint32_t)w_cur * h0; acc0[1] += (uint32_t)w_cur * h1;
            acc1[0] += (uint32_t)w_next * h0; acc1[1] += (uint32_t)w_next * h1;
            __builtin_amdgcn_wave_barrier();
            if ((sy + 1) * oh >= top) {                  // output row oy is complete
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int ox = lane + 64 * q;
                    if (q == 0 ? on0 : on1) {
                        // (sum + area/2) / area by multiply-shift
                        const uint32_t val = (uint32_t)(((uint64_t)(acc0[q] + half) * magic) >> 42);
                        uint8_t* px = o + ((size_t)oy * ow + ox) * S;
                        if (S == 4) *reinterpret_cast<uint32_t*>(px) = ((fresh ? 0u : old[q]) >> 8) | (val << 24);
                        else stack_push<S>(px, val, fresh);
                    }
                    acc0[q] = acc1[q];
                    acc1[q] = 0;
                }
                fetch_old(oy + 1);
            }
        }
#pragma unroll
        for (int p = 0; p < PF; p++)
#pragma unroll
            for (int q = 0; q < 2; q++) { cur_a[p][q] = nxt_a[p][q]; cur_b[p][q] = nxt_b[p][q]; }
    }
}

int agent_fail(tbx_engine* e, const char* what, hipError_t err)
{
    return e->fail(TBX_E_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(err));
}

#define AHIP(call)                                           \
    do {                                                     \
        hipError_t _e = (call);                              \
        if (_e != hipSuccess) return agent_fail(e, #call, _e); \
    } while (0)

// overlap of output cell o (extent `src` in 1/out units) with the source pixels (extent `out` each)
std::vector<AgentTaps> make_taps(int src, int out)
{
    std::vector<AgentTaps> t((size_t)out);
    for (int o = 0; o < out; o++) {
        const long lo = (long)o * src, hi = (long)(o + 1) * src;
        AgentTaps a;
        memset(&a, 0, sizeof a);
        a.start = (int)(lo / out);
        for (long s = a.start; s * out < hi && a.n < MAX_TAPS; s++) {
            const long l = s * out > lo ? s * out : lo, h = (s + 1) * out < hi ? (s + 1) * out : hi;
            a.w[a.n++] = (uint8_t)(h - l);
        }
        t[(size_t)o] = a;
    }
    return t;
}

int launch_warp(tbx_engine* e, int reset_mode, hipStream_t s)
{
    AgentState& a = *e->agent;
    const dim3 grid((e->n + TBX_WAVES_PER_BLOCK - 1) / TBX_WAVES_PER_BLOCK), block(TBX_BLOCK);
    const uint8_t* A = a.cfg.skip >= 2 ? a.gray_a : a.gray_b;
    const uint64_t magic = (1ull << 42) / (uint64_t)(a.H * a.W) + 1ull;   // exact for numerators < 2^42 / area >= 2^25
    // fresh = the env was reset during this agent step (game over, or a lost life in episodic-life mode)
#define WARP(S) hipLaunchKernelGGL(agent_warp_kernel<S>, grid, block, 0, s, A, a.gray_b, a.done_out, a.tx, a.obs, \
                                   a.H, a.W, a.cfg.out_h, a.cfg.out_w, magic, reset_mode, e->n)
    switch (a.cfg.stack) {
    case 1: WARP(1); break;
    case 2: WARP(2); break;
    case 3: WARP(3); break;
    default: WARP(4); break;
    }
#undef WARP
    AHIP(hipGetLastError());
    return TBX_OK;
}

AgentWarpArgs warp_args(tbx_engine* e, int reset_mode)
{
    AgentState& a = *e->agent;
    AgentWarpArgs w;
    w.fin = a.done_out; w.tx = a.tx; w.obs = a.obs;
    w.H = a.H; w.W = a.W; w.oh = a.cfg.out_h; w.ow = a.cfg.out_w; w.stack = a.cfg.stack;
    w.reset_mode = reset_mode;
    w.two_frames = a.cfg.skip >= 2;
    w.magic = (1ull << 42) / (uint64_t)(a.H * a.W) + 1ull;
    return w;
}

bool needs_reset_kernel(const AgentState& a) { return a.cfg.episodic_life || a.cfg.fire_reset || a.cfg.noop_max > 0; }

AgentResetArgs reset_args(tbx_engine* e)
{
    AgentState& a = *e->agent;
    AgentResetArgs r;
    r.kind = a.kind;
    r.list = nullptr; r.count = nullptr;
    r.skip = a.cfg.skip; r.episodic_life = a.cfg.episodic_life; r.fire_reset = a.cfg.fire_reset; r.noop_max = a.cfg.noop_max;
    r.noop_seed = a.cfg.noop_seed; r.env_offset = a.cfg.env_offset;
    // action #1 and #2 of the game's (sorted) legal action set: FIRE and the next one (atari_wrappers.py:146-149)
    r.fire_buttons = tbx_ale_buttons(tbx_legal_action(e->game, 1));
    r.third_buttons = tbx_ale_buttons(tbx_legal_action(e->game, 2));
    r.ep_ret = a.ep_ret; r.ep_len = a.ep_len; r.ep_index = a.ep_index; r.prev_lives = a.prev_lives;
    r.ep_done = a.ep_done; r.ep_ret_out = a.ep_ret_out; r.ep_len_out = a.ep_len_out;
    return r;
}

// frame B (and the observation) once the sub-frames and the resets are done
int observe(tbx_engine* e, int reset_mode, hipStream_t s)
{
    AgentState& a = *e->agent;
    if (e->ops->agent_fused() && !a.force_generic) {
        int rc = e->ops->agent_snapshot(e, 1, s);
        if (rc) return rc;
        return e->ops->agent_warp(e, warp_args(e, reset_mode), s);
    }
    int rc = e->ops->render(e, a.gray_b, 1, 0, e->n, s);
    if (rc) return rc;
    return launch_warp(e, reset_mode, s);
}

// the whole agent step, asynchronous on `s`
int agent_step_async(tbx_engine* e, ActionSource src, hipStream_t s)
{
    AgentState& a = *e->agent;
    const int n = e->n, tb = 256, gb = (n + tb - 1) / tb;
    const bool fused = e->ops->agent_fused() && !a.force_generic;
    const bool in_kernel_reset = needs_reset_kernel(a);
    // the step kernels sum reward / latch done themselves (tbx_accumulate), so the skip loop is one launch per frame
    src.acc_reward = a.racc; src.acc_done = a.fin;
    if (e->ops->multi_frame_step() && fused) {
        // the whole skip loop in one launch: state stays in registers, frame A's snapshot is stored on the way
        src.acc_first = 1;
        src.frames = a.cfg.skip;
        src.snapshot_after = a.cfg.skip >= 2 ? a.cfg.skip - 1 : 0;
        int rc = e->ops->step(e, src, 0, s);
        if (rc) return rc;
    } else {
        src.frames = 1;
        src.snapshot_after = 0;
        for (int i = 0; i < a.cfg.skip; i++) {
            src.acc_first = i == 0;
            int rc = e->ops->step(e, src, 0, s);
            if (rc) return rc;
            if (i == a.cfg.skip - 2) {
                rc = fused ? e->ops->agent_snapshot(e, 0, s) : e->ops->render(e, a.gray_a, 1, 0, n, s);
                if (rc) return rc;
            }
        }
    }
    hipLaunchKernelGGL(agent_monitor_kernel, dim3(gb), dim3(tb), 0, s, a.racc, a.fin, e->lives_out, a.ep_ret, a.ep_len, a.ep_index,
                       a.prev_lives, a.kind, a.ep_done, a.ep_ret_out, a.ep_len_out, a.reward_out, a.done_out,
                       a.reset_list, a.reset_count + a.parity, a.reset_count + (a.parity ^ 1),
                       a.cfg.episodic_life, a.cfg.clip_reward, in_kernel_reset ? 0 : 1, n);
    AHIP(hipGetLastError());
    // VecEnv auto-reset: plain new game of the finished envs, or the reset-time wrappers run in-kernel
    AgentResetArgs ra = reset_args(e);
    ra.list = a.reset_list; ra.count = a.reset_count + a.parity;
    a.parity ^= 1;
    int rc = in_kernel_reset ? e->ops->agent_reset_envs(e, ra, s) : e->ops->new_game(e, a.fin, s);
    if (rc) return rc;
    return observe(e, 0, s);
}

}  // namespace

void tbx_agent_free(tbx_engine* e)
{
    if (!e->agent) return;
    AgentState* a = e->agent;
    hipFree(a->gray_a); hipFree(a->gray_b); hipFree(a->obs); hipFree(a->fin); hipFree(a->done_out);
    hipFree(a->racc); hipFree(a->reward_out); hipFree(a->ty); hipFree(a->tx);
    hipFree(a->kind); hipFree(a->ep_done); hipFree(a->ep_ret); hipFree(a->ep_len); hipFree(a->ep_index);
    hipFree(a->prev_lives); hipFree(a->ep_len_out); hipFree(a->ep_ret_out); hipFree(a->reset_list); hipFree(a->reset_count);
    delete a;
    e->agent = nullptr;
}

int tbx_agent_buffer(tbx_engine* e, int which, void** out_ptr, size_t* out_bytes)
{
    if (!e->agent) return e->fail(TBX_E_INVALID, "tbx_agent_init has not been called");
    AgentState& a = *e->agent;
    const size_t N = (size_t)e->n;
    void* p = nullptr;
    size_t b = 0;
    switch (which) {
    case TBX_BUF_AGENT_OBS: p = a.obs; b = N * a.cfg.out_h * a.cfg.out_w * a.cfg.stack; break;
    case TBX_BUF_AGENT_REWARD: p = a.reward_out; b = N * sizeof(float); break;
    case TBX_BUF_AGENT_DONE: p = a.done_out; b = N; break;
    case TBX_BUF_AGENT_EP_DONE: p = a.ep_done; b = N; break;
    case TBX_BUF_AGENT_EP_RETURN: p = a.ep_ret_out; b = N * sizeof(float); break;
    case TBX_BUF_AGENT_EP_LENGTH: p = a.ep_len_out; b = N * sizeof(int32_t); break;
    default: return e->fail(TBX_E_INVALID, "unknown buffer id");
    }
    *out_ptr = p;
    if (out_bytes) *out_bytes = b;
    return TBX_OK;
}

extern "C" {

int tbx_agent_init(tbx_engine* e, const tbx_agent_config_t* cfg)
{
    if (!e) return TBX_E_INVALID;
    if (!cfg) return e->fail(TBX_E_INVALID, "agent config is NULL");
    const int H = e->ops->height(), W = e->ops->width();
    if (cfg->skip < 1 || cfg->skip > 64 || cfg->stack < 1 || cfg->stack > 4 || cfg->out_h < 1 || cfg->out_w < 1 ||
        cfg->out_h > H || cfg->out_w > W || cfg->out_w > 128 || cfg->out_h * cfg->out_w > AGENT_MAX_OUT_PX || cfg->noop_max < 0 ||
        cfg->noop_max > 1000)
        return e->fail(TBX_E_INVALID, "agent config out of range (skip 1..64, stack 1..4, 1 <= out <= frame, out_w <= 128, out_h*out_w <= 7056, noop_max 0..1000)");
    if ((H + cfg->out_h - 1) / cfg->out_h + 1 > MAX_TAPS || (W + cfg->out_w - 1) / cfg->out_w + 1 > MAX_TAPS)
        return e->fail(TBX_E_UNSUPPORTED, "agent: the resize ratio needs more than 8 taps per axis");
    AHIP(hipSetDevice(e->device));
    tbx_agent_free(e);
    AgentState* a = new AgentState();
    e->agent = a;
    a->cfg = *cfg;
    a->H = H; a->W = W;
    if (const char* v = getenv("TBX_AGENT_GENERIC")) a->force_generic = atoi(v) != 0;
    const size_t N = (size_t)e->n;
    AHIP(hipMalloc((void**)&a->gray_a, N * H * W));
    AHIP(hipMalloc((void**)&a->gray_b, N * H * W));
    AHIP(hipMalloc((void**)&a->obs, N * cfg->out_h * cfg->out_w * cfg->stack));
    AHIP(hipMalloc((void**)&a->fin, N));
    AHIP(hipMalloc((void**)&a->done_out, N));
    AHIP(hipMalloc((void**)&a->racc, N * sizeof(int32_t)));
    AHIP(hipMalloc((void**)&a->reward_out, N * sizeof(float)));
    AHIP(hipMalloc((void**)&a->kind, N));
    AHIP(hipMalloc((void**)&a->ep_done, N));
    AHIP(hipMalloc((void**)&a->ep_ret, N * sizeof(int32_t)));
    AHIP(hipMalloc((void**)&a->ep_len, N * sizeof(int32_t)));
    AHIP(hipMalloc((void**)&a->ep_index, N * sizeof(int32_t)));
    AHIP(hipMalloc((void**)&a->prev_lives, N * sizeof(int32_t)));
    AHIP(hipMalloc((void**)&a->ep_len_out, N * sizeof(int32_t)));
    AHIP(hipMalloc((void**)&a->ep_ret_out, N * sizeof(float)));
    AHIP(hipMalloc((void**)&a->reset_list, N * sizeof(int32_t)));
    AHIP(hipMalloc((void**)&a->reset_count, 2 * sizeof(int32_t)));
    AHIP(hipMemset(a->reset_count, 0, 2 * sizeof(int32_t)));
    const std::vector<AgentTaps> ty = make_taps(H, cfg->out_h), tx = make_taps(W, cfg->out_w);
    AHIP(hipMalloc((void**)&a->ty, ty.size() * sizeof(AgentTaps)));
    AHIP(hipMalloc((void**)&a->tx, tx.size() * sizeof(AgentTaps)));
    AHIP(hipMemcpy(a->ty, ty.data(), ty.size() * sizeof(AgentTaps), hipMemcpyHostToDevice));
    AHIP(hipMemcpy(a->tx, tx.data(), tx.size() * sizeof(AgentTaps), hipMemcpyHostToDevice));
    AHIP(hipMemset(a->obs, 0, N * cfg->out_h * cfg->out_w * cfg->stack));
    AHIP(hipMemset(a->fin, 0, N));
    AHIP(hipMemset(a->done_out, 0, N));
    AHIP(hipMemset(a->racc, 0, N * sizeof(int32_t)));
    AHIP(hipMemset(a->reward_out, 0, N * sizeof(float)));
    AHIP(hipMemset(a->kind, 0, N));
    AHIP(hipMemset(a->ep_done, 0, N));
    AHIP(hipMemset(a->ep_ret, 0, N * sizeof(int32_t)));
    AHIP(hipMemset(a->ep_len, 0, N * sizeof(int32_t)));
    AHIP(hipMemset(a->ep_index, 0, N * sizeof(int32_t)));
    AHIP(hipMemset(a->prev_lives, 0, N * sizeof(int32_t)));
    AHIP(hipMemset(a->ep_len_out, 0, N * sizeof(int32_t)));
    AHIP(hipMemset(a->ep_ret_out, 0, N * sizeof(float)));
    return TBX_OK;
}

int tbx_agent_reset(tbx_engine* e, uint8_t* obs_host)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent) return e->fail(TBX_E_INVALID, "tbx_agent_init has not been called");
    AgentState& a = *e->agent;
    AHIP(hipSetDevice(e->device));
    AHIP(tbx_use_stream(e, e->stream));
    const int n = e->n, tb = 256, gb = (n + tb - 1) / tb;
    const size_t N = (size_t)n;
    AHIP(hipMemsetAsync(a.ep_ret, 0, N * sizeof(int32_t), e->stream));
    AHIP(hipMemsetAsync(a.ep_len, 0, N * sizeof(int32_t), e->stream));
    AHIP(hipMemsetAsync(a.ep_done, 0, N, e->stream));
    int rc;
    if (needs_reset_kernel(a)) {
        // env.reset() of the whole wrapper stack == the game-over reset path for every env
        hipLaunchKernelGGL(agent_fill_u8_kernel, dim3(gb), dim3(tb), 0, e->stream, a.kind, (uint8_t)2, n);
        rc = e->ops->agent_reset_envs(e, reset_args(e), e->stream);
        if (rc) return rc;
        AHIP(hipMemsetAsync(a.ep_done, 0, N, e->stream));
    } else {
        rc = e->ops->new_game(e, nullptr, e->stream);
        if (rc) return rc;
    }
    rc = observe(e, 1, e->stream);
    if (rc) return rc;
    if (obs_host)
        AHIP(hipMemcpyAsync(obs_host, a.obs, N * a.cfg.out_h * a.cfg.out_w * a.cfg.stack, hipMemcpyDeviceToHost, e->stream));
    AHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_agent_episodes(tbx_engine* e, uint8_t* ep_done_host, float* ep_return_host, int32_t* ep_length_host)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent) return e->fail(TBX_E_INVALID, "tbx_agent_init has not been called");
    AgentState& a = *e->agent;
    AHIP(hipSetDevice(e->device));
    AHIP(tbx_use_stream(e, e->stream));
    const size_t N = (size_t)e->n;
    if (ep_done_host) AHIP(hipMemcpyAsync(ep_done_host, a.ep_done, N, hipMemcpyDeviceToHost, e->stream));
    if (ep_return_host) AHIP(hipMemcpyAsync(ep_return_host, a.ep_ret_out, N * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (ep_length_host) AHIP(hipMemcpyAsync(ep_length_host, a.ep_len_out, N * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    AHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_agent_step_device(tbx_engine* e, const int32_t* actions_dev, void* stream)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent) return e->fail(TBX_E_INVALID, "tbx_agent_init has not been called");
    if (!actions_dev) return e->fail(TBX_E_INVALID, "actions pointer is NULL");
    AHIP(hipSetDevice(e->device));
    AHIP(tbx_use_stream(e, (hipStream_t)stream));
    AHIP(tbx_gather_before_step(e, (hipStream_t)stream));
    ActionSource src{};
    src.actions = actions_dev;
    src.single_env = -1;
    return agent_step_async(e, src, (hipStream_t)stream);
}

int tbx_agent_step_synthetic(tbx_engine* e, uint64_t action_seed, uint64_t t, uint64_t env_offset, void* stream)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent) return e->fail(TBX_E_INVALID, "tbx_agent_init has not been called");
    AHIP(hipSetDevice(e->device));
    AHIP(tbx_use_stream(e, (hipStream_t)stream));
    AHIP(tbx_gather_before_step(e, (hipStream_t)stream));
    ActionSource src{};
    src.seed = action_seed; src.t = t; src.env_offset = env_offset;
    src.single_env = -1;
    return agent_step_async(e, src, (hipStream_t)stream);
}

int tbx_agent_step(tbx_engine* e, const int32_t* actions_host, float* reward_host, uint8_t* done_host, uint8_t* obs_host)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent) return e->fail(TBX_E_INVALID, "tbx_agent_init has not been called");
    if (!actions_host) return e->fail(TBX_E_INVALID, "actions pointer is NULL");
    AgentState& a = *e->agent;
    AHIP(hipSetDevice(e->device));
    AHIP(tbx_use_stream(e, e->stream));
    AHIP(tbx_gather_before_step(e, e->stream));
    const size_t N = (size_t)e->n;
    AHIP(hipMemcpyAsync(e->actions, actions_host, N * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    ActionSource src{};
    src.actions = e->actions;
    src.single_env = -1;
    int rc = agent_step_async(e, src, e->stream);
    if (rc) return rc;
    if (reward_host) AHIP(hipMemcpyAsync(reward_host, a.reward_out, N * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (done_host) AHIP(hipMemcpyAsync(done_host, a.done_out, N, hipMemcpyDeviceToHost, e->stream));
    if (obs_host) AHIP(hipMemcpyAsync(obs_host, a.obs, N * a.cfg.out_h * a.cfg.out_w * a.cfg.stack, hipMemcpyDeviceToHost, e->stream));
    AHIP(hipStreamSynchronize(e->stream));
    // an illegal action id is reported like tbx_step does
    uint32_t f = 0;
    AHIP(hipMemcpy(&f, e->err_flag, sizeof f, hipMemcpyDeviceToHost));
    if (f) {
        AHIP(hipMemset(e->err_flag, 0, sizeof f));
        return e->fail(TBX_E_ACTION, "an illegal ALE action id was passed (treated as NOOP)");
    }
    return TBX_OK;
}

}  // extern "C"
