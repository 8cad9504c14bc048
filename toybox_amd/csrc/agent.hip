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
    uint8_t *gray_a = nullptr, *gray_b = nullptr;   // generic path only: full-resolution gray frames of the two buffer slots
    uint8_t *obs = nullptr, *fin = nullptr, *done_out = nullptr;
    uint8_t* plane = nullptr;                       // [N][out_h][out_w] newest plane alone (cfg.new_plane = 1), else nullptr
    // cfg.new_plane = 2: no rolled stack (obs == nullptr) but a ring of planes [stack][N][out_h][out_w]; slot `head` holds the newest
    uint8_t* ring = nullptr;
    int head = 0;
    // host delivery (tbx_agent_step_begin / _end): an agent step whose outputs are on their way to the caller's host buffers
    bool host_pending = false;
    int32_t* host_actions = nullptr;                // pinned [N]: the caller's actions, copied before _begin returns
    // the small outputs travel as ONE block (six separate copies cost the stream 10-15 us each, a tenth of the 8 192-env step):
    // [reward f32 N | ep_return f32 N | ep_length i32 N | error word | done u8 N | ep_done u8 N], gathered by a kernel
    uint32_t* io_dev = nullptr;
    uint32_t* io_host = nullptr;                    // pinned mirror; tbx_agent_step_end hands its parts to the caller's arrays
    tbx_agent_host_out_t host_out{};
    // the observation of a big batch goes out in CHUNKS of envs: chunk c's planes (stacks) travel on the copy stream while the
    // observation kernel of chunk c + 1 runs (the link carries 7 KB per env and step: at 8 192 envs the copy takes 1.05 ms, the
    // SpaceInvaders observation kernel 0.26 ms -- all of it would otherwise sit in front of the copy)
    hipStream_t copy_stream = nullptr;
    hipEvent_t chunk_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t copies_done = nullptr;
    int32_t* racc = nullptr;
    float* reward_out = nullptr;
    AgentTaps *ty = nullptr, *tx = nullptr;
    // per-env state of the wrapper classes: Monitor (return, length, needs_reset), EpisodicLifeEnv (lives, was_real_done),
    // the episode counter behind the no-op count, MaxAndSkipEnv's buffer validity; and this step's outputs
    uint8_t *kind = nullptr, *ep_done = nullptr, *was_real_done = nullptr, *needs_reset = nullptr;
    uint8_t *mode = nullptr, *buf_valid = nullptr, *exec_flag = nullptr;
    int32_t *ep_ret = nullptr, *ep_len = nullptr, *ep_index = nullptr, *prev_lives = nullptr, *ep_len_out = nullptr;
    int32_t* noop_override = nullptr;   // [N] or nullptr
    float* ep_ret_out = nullptr;
    int32_t* reset_list = nullptr;   // [N] envs the monitor kernel flagged for a reset
    int32_t* reset_count = nullptr;  // [2] their number, double-buffered by step parity
    int parity = 0;
    bool force_generic = false;   // TBX_AGENT_GENERIC=1: observations through full-resolution gray frames
};

namespace {

constexpr int MAX_TAPS = 8;

// after MaxAndSkipEnv.step: Monitor.step (bench/monitor.py:51-76), EpisodicLifeEnv.step (atari_wrappers.py:166-178),
// ClipRewardEnv, and DummyVecEnv's decision to reset (dummy_vec_env.py:51-52): kind = 1 for envs that report done.
// simple: no in-kernel reset procedure follows (custom-brick Breakout): a finished game is restarted by a plain new game and
// Monitor.reset's bookkeeping happens here.
__global__ void agent_monitor_kernel(const int32_t* racc, const uint8_t* fin, const int32_t* lives, int32_t* ep_ret, int32_t* ep_len,
                                     int32_t* ep_index, int32_t* prev_lives, uint8_t* was_real_done, uint8_t* needs_reset, uint8_t* kind,
                                     uint8_t* mode, uint8_t* ep_done, float* ep_ret_out, int32_t* ep_len_out, float* reward_out,
                                     uint8_t* done_out, int32_t* list, int32_t* count, int32_t* next_count, uint32_t* err_flag,
                                     int episodic, int clip, int simple, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *next_count = 0;                         // the other step parity's counter: its readers ran a step ago
    if (i >= n) return;
    const int r = racc[i];
    const bool real = fin[i] != 0;
    bool emitted = false;
    int er = ep_ret[i], el = ep_len[i];
    if (needs_reset[i]) atomicOr(err_flag, 2u);          // Monitor raises "Tried to step environment that needs reset"
    else {
        er += r; el += 1;
        if (real) { emitted = true; needs_reset[i] = 1; }
    }
    bool done = real;
    const int l = lives[i];
    if (episodic) {
        was_real_done[i] = real ? 1 : 0;
        if (l < prev_lives[i] && l > 0) done = true;
        prev_lives[i] = l;
    }
    kind[i] = done ? 1 : 0;
    mode[i] = (simple && done) ? 1 : 0;                  // a plain new game: the observation is its raw frame
    if (done) list[atomicAdd(count, 1)] = i;             // order is irrelevant: the reset of one env touches nothing else
    done_out[i] = done ? 1 : 0;
    reward_out[i] = clip ? (float)((r > 0) - (r < 0)) : (float)r;
    ep_done[i] = emitted ? 1 : 0;
    if (emitted) { ep_ret_out[i] = (float)er; ep_len_out[i] = el; }
    if (simple && real) { er = 0; el = 0; ep_index[i] += 1; needs_reset[i] = 0; }
    ep_ret[i] = er; ep_len[i] = el;
}

// the per-env outputs of an agent step as one block for the host (layout: AgentState::io_dev)
__global__ void agent_pack_outputs_kernel(const float* reward, const float* ep_ret, const int32_t* ep_len, const uint8_t* done, const uint8_t* ep_done,
                                          const uint32_t* err_flag, uint32_t* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) out[3 * (size_t)n] = *err_flag;
    if (i >= n) return;
    out[i] = __float_as_uint(reward[i]);
    out[(size_t)n + i] = __float_as_uint(ep_ret[i]);
    out[2 * (size_t)n + i] = (uint32_t)ep_len[i];
    uint8_t* b = reinterpret_cast<uint8_t*>(out + 3 * (size_t)n + 1);
    b[i] = done[i];
    b[(size_t)n + i] = ep_done[i];
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
__global__ __launch_bounds__(TBX_BLOCK) void agent_warp_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B, AgentWarpArgs wa, int n)
{
    const AgentTaps* __restrict__ tx = wa.tx;
    uint8_t* __restrict__ obs = wa.obs;
    const int H = wa.H, W = wa.W, oh = wa.oh, ow = wa.ow;
    const uint64_t magic = wa.magic;
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[TBX_WAVES_PER_BLOCK][352];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int env = wave_uniform(wa.first + blockIdx.x * TBX_WAVES_PER_BLOCK + wave);
    if (env >= wa.end) return;
    uint8_t* row = lds_all[wave];
    // B holds slot B, or the live frame where the env's observation is a raw frame; A holds slot A
    const ObsSel sel = agent_obs_sel(wa, env);
    const bool use_a = !sel.none && (sel.two || sel.single == 1), use_b = !sel.none && (sel.two || sel.single != 1);
    const int fresh = sel.zero;                         // VecFrameStack: older slots become zero (FrameStack: the new frame)
    const uint8_t* fa = A + (size_t)env * H * W;
    const uint8_t* fb = B + (size_t)env * H * W;
    uint8_t* o = S != 0 ? obs + (size_t)env * oh * ow * S : nullptr;
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
                    if (use_b) vb[p][q] = *reinterpret_cast<const uint32_t*>(fb + (size_t)sy * W + 4 * w4);
                    if (use_a) va[p][q] = *reinterpret_cast<const uint32_t*>(fa + (size_t)sy * W + 4 * w4);
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
                if (w4 < words) reinterpret_cast<uint32_t*>(row)[w4] = bytemax4(cur_a[p][q], cur_b[p][q]);   // an unused slot reads 0
            }
            __builtin_amdgcn_wave_barrier();
            // vertical split of this source row (extent oh in refined units) over output rows oy and oy+1 (extent H each)
            const int oy = (sy * oh) / H;
            const int top = (oy + 1) * H;
            const int w_cur = min((sy + 1) * oh, top) - sy * oh, w_next = oh - w_cur;
            const uint32_t h0 = on0 ? hsum(row, c0) : 0u, h1 = on1 ? hsum(row, c1) : 0u;
            acc0[0] += __umul24((uint32_t)w_cur, h0); acc0[1] += __umul24((uint32_t)w_cur, h1);     // 24-bit operands (weights <= out_h, sums <= 255 W): full-rate v_mad_u32_u24, not v_mad_u64_u32
            acc1[0] += __umul24((uint32_t)w_next, h0); acc1[1] += __umul24((uint32_t)w_next, h1);
            __builtin_amdgcn_wave_barrier();
            if ((sy + 1) * oh >= top) {                  // output row oy is complete
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int ox = lane + 64 * q;
                    if (q == 0 ? on0 : on1) {
                        // (sum + area/2) / area by multiply-shift
                        const uint32_t val = (uint32_t)(((uint64_t)(acc0[q] + half) * magic) >> 42);
                        const size_t at = ((size_t)env * oh + oy) * ow + ox;
                        if constexpr (S != 0) {
                            uint8_t* px = o + ((size_t)oy * ow + ox) * S;
                            if (S == 4) *reinterpret_cast<uint32_t*>(px) = (stack_old_word(fresh ? 0u : old[q], val, fresh) >> 8) | (val << 24);
                            else stack_push<S>(px, val, fresh);
                        } else if (fresh) {                  // the plane ring (S == 0): a stack that starts afresh rewrites the other slots too
                            for (int k = 0; k + 1 < wa.stack; k++) ring_older(wa, k)[at] = fresh == 2 ? (uint8_t)val : (uint8_t)0;
                        }
                        if (wa.plane) wa.plane[at] = (uint8_t)val;
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

size_t ring_stride(const AgentState& a, int n) { return (size_t)n * a.cfg.out_h * a.cfg.out_w; }

// where the observation kernels put the newest plane alone: the plane buffer, the ring's head slot, or nowhere
uint8_t* newest_plane(const AgentState& a, int n) { return a.ring ? a.ring + (size_t)a.head * ring_stride(a, n) : a.plane; }

int no_stack(tbx_engine* e)
{
    return e->fail(TBX_E_INVALID, "no rolled stack on the device with tbx_agent_config_t::new_plane = 2: the planes are in TBX_BUF_AGENT_RING (newest: TBX_BUF_AGENT_PLANE)");
}

AgentWarpArgs warp_args(tbx_engine* e, int reset_mode)
{
    AgentState& a = *e->agent;
    AgentWarpArgs w;
    w.zero = a.done_out; w.mode = a.mode; w.valid = a.buf_valid; w.tx = a.tx; w.obs = a.obs; w.plane = newest_plane(a, e->n);
    uint8_t* older[3];
    for (int k = 0; k < 3; k++)
        older[k] = (a.ring && k + 1 < a.cfg.stack) ? a.ring + (size_t)((a.head + 1 + k) % a.cfg.stack) * ring_stride(a, e->n) : nullptr;
    w.older0 = older[0]; w.older1 = older[1]; w.older2 = older[2];
    w.H = a.H; w.W = a.W; w.oh = a.cfg.out_h; w.ow = a.cfg.out_w; w.stack = a.cfg.stack;
    w.first = 0; w.end = e->n;
    w.reset_mode = reset_mode;
    w.fill_repeat = a.cfg.stack_fill != 0;
    w.magic = (1ull << 42) / (uint64_t)(a.H * a.W) + 1ull;   // exact for numerators < 2^42 / area >= 2^25
#ifdef TBX_DIAG
    w.diag = getenv("TBX_AGENT_DIAG") ? atoi(getenv("TBX_AGENT_DIAG")) : 0;
#endif
    return w;
}

int launch_warp(tbx_engine* e, int reset_mode, hipStream_t s)
{
    AgentState& a = *e->agent;
    const dim3 grid((e->n + TBX_WAVES_PER_BLOCK - 1) / TBX_WAVES_PER_BLOCK), block(TBX_BLOCK);
    const AgentWarpArgs w = warp_args(e, reset_mode);
#define WARP(S) hipLaunchKernelGGL(agent_warp_kernel<S>, grid, block, 0, s, a.gray_a, a.gray_b, w, e->n)
    switch (a.obs ? a.cfg.stack : 0) {
    case 0: WARP(0); break;                              // the plane ring (new_plane = 2), any depth
    case 1: WARP(1); break;
    case 2: WARP(2); break;
    case 3: WARP(3); break;
    default: WARP(4); break;
    }
#undef WARP
    AHIP(hipGetLastError());
    return TBX_OK;
}

bool wants_wrappers(const AgentState& a) { return a.cfg.episodic_life || a.cfg.fire_reset || a.cfg.noop_max > 0 || a.noop_override; }

AgentResetArgs reset_args(tbx_engine* e)
{
    AgentState& a = *e->agent;
    AgentResetArgs r;
    r.kind = a.kind;
    r.list = nullptr; r.count = nullptr;
    r.skip = a.cfg.skip; r.episodic_life = a.cfg.episodic_life; r.fire_reset = a.cfg.fire_reset; r.noop_max = a.cfg.noop_max;
    r.noop_seed = a.cfg.noop_seed; r.env_offset = a.cfg.env_offset;
    r.noop_override = a.noop_override;
    // action #1 and #2 of the game's (sorted) legal action set: FIRE and the next one (atari_wrappers.py:146-149)
    r.fire_buttons = tbx_ale_buttons(tbx_legal_action(e->game, 1));
    r.third_buttons = tbx_ale_buttons(tbx_legal_action(e->game, 2));
    r.ep_ret = a.ep_ret; r.ep_len = a.ep_len; r.ep_index = a.ep_index; r.prev_lives = a.prev_lives;
    r.was_real_done = a.was_real_done; r.needs_reset = a.needs_reset;
    r.ep_done = a.ep_done; r.ep_ret_out = a.ep_ret_out; r.ep_len_out = a.ep_len_out;
    r.mode = a.mode; r.buf_valid = a.buf_valid; r.err_flag = e->err_flag;
    return r;
}

// the fused observation kernels can be launched for a range of envs (observe_chunk); the generic path cannot
bool observe_in_chunks(tbx_engine* e) { return e->ops->agent_fused() && !e->agent->force_generic; }

int observe_chunk(tbx_engine* e, int first, int end, hipStream_t s)
{
    AgentWarpArgs w = warp_args(e, 0);
    w.first = first; w.end = end;
    return e->ops->agent_warp(e, w, s);
}

// the observation, once the frames and the resets of the agent step are done
int observe(tbx_engine* e, int reset_mode, hipStream_t s)
{
    AgentState& a = *e->agent;
    if (e->ops->agent_fused() && !a.force_generic) return e->ops->agent_warp(e, warp_args(e, reset_mode), s);
    // generic path: the two buffer slots (and the live frame where a reset returned one) as full-resolution gray frames
    int rc = e->ops->render_from(e, 1, nullptr, a.gray_a, 1, s);
    if (rc) return rc;
    rc = e->ops->render_from(e, 2, a.mode, a.gray_b, 1, s);
    if (rc) return rc;
    return launch_warp(e, reset_mode, s);
}

// the whole agent step, asynchronous on `s` (with_observation = false: everything but the observation kernel, which the caller
// launches itself -- tbx_agent_step_begin does, chunk by chunk)
int agent_step_async(tbx_engine* e, ActionSource src, hipStream_t s, bool with_observation = true)
{
    AgentState& a = *e->agent;
    const int n = e->n, tb = 256, gb = (n + tb - 1) / tb;
    const bool in_kernel_reset = e->ops->agent_reset_supported();
    if (!in_kernel_reset && wants_wrappers(a))
        return e->fail(TBX_E_UNSUPPORTED, "agent: episodic-life / fire-reset / no-op-reset are not available for this engine state");
    // MaxAndSkipEnv.step: the step kernels sum reward / latch done themselves (tbx_accumulate) and write the buffer slots
    src.acc_reward = a.racc; src.acc_done = a.fin;
    src.snap_a_after = a.cfg.skip >= 2 ? a.cfg.skip - 1 : 0;
    src.snap_b_after = a.cfg.skip;
    src.buf_valid = a.buf_valid;
    if (e->ops->multi_frame_step()) {
        // the whole skip loop in one launch: state stays in registers, the slots are stored on the way
        src.frames = a.cfg.skip; src.frame0 = 0; src.exec_flag = nullptr;
        int rc = e->ops->step(e, src, 0, s);
        if (rc) return rc;
    } else {
        src.frames = 1;
        src.exec_flag = a.exec_flag;
        for (int i = 0; i < a.cfg.skip; i++) {
            src.frame0 = i;
            int rc = e->ops->step(e, src, 0, s);
            if (rc) return rc;
            if (i + 1 == src.snap_a_after) rc = e->ops->agent_snapshot(e, 0, a.exec_flag, a.buf_valid, s);
            if (rc) return rc;
            if (i + 1 == src.snap_b_after) rc = e->ops->agent_snapshot(e, 1, a.exec_flag, a.buf_valid, s);
            if (rc) return rc;
        }
    }
    hipLaunchKernelGGL(agent_monitor_kernel, dim3(gb), dim3(tb), 0, s, a.racc, a.fin, e->lives_out, a.ep_ret, a.ep_len, a.ep_index,
                       a.prev_lives, a.was_real_done, a.needs_reset, a.kind, a.mode, a.ep_done, a.ep_ret_out, a.ep_len_out,
                       a.reward_out, a.done_out, a.reset_list, a.reset_count + a.parity, a.reset_count + (a.parity ^ 1), e->err_flag,
                       a.cfg.episodic_life, a.cfg.clip_reward, in_kernel_reset ? 0 : 1, n);
    AHIP(hipGetLastError());
    // DummyVecEnv: obs = env.reset() for the envs that reported done -- the whole reset path of the stack in-kernel, or (engines
    // without one) a plain new game
    AgentResetArgs ra = reset_args(e);
    ra.list = a.reset_list; ra.count = a.reset_count + a.parity;
    a.parity ^= 1;
    int rc = in_kernel_reset ? e->ops->agent_reset_envs(e, ra, s) : e->ops->new_game(e, a.fin, s);
    if (rc) return rc;
    if (a.ring) a.head = (a.head + 1) % a.cfg.stack;     // the slot of the oldest plane takes the new one
    return with_observation ? observe(e, 0, s) : TBX_OK;
}

// the error word of the device: bit 0 illegal action id, bit 1 a step on an env whose Monitor needed a reset
int report_flags(tbx_engine* e)
{
    uint32_t f = 0;
    AHIP(hipMemcpy(&f, e->err_flag, sizeof f, hipMemcpyDeviceToHost));
    if (f) {
        AHIP(hipMemset(e->err_flag, 0, sizeof f));
        if (f & 2u)
            return e->fail(TBX_E_NEEDS_RESET, "an env was stepped after its game ended inside EpisodicLifeEnv's no-op step (bench.Monitor raises here)");
        return e->fail(TBX_E_ACTION, "an illegal ALE action id was passed (treated as NOOP)");
    }
    return TBX_OK;
}

}  // namespace

void tbx_agent_free(tbx_engine* e)
{
    if (!e->agent) return;
    AgentState* a = e->agent;
    if (a->host_pending) {                                     // copies into the caller's buffers are still in flight
        hipStreamSynchronize(e->stream);
        if (e->pending_kind == 2) e->pending_kind = 0;
    }
    hipFree(a->plane); hipFree(a->ring); hipHostFree(a->host_actions); hipFree(a->io_dev); hipHostFree(a->io_host);
    if (a->copy_stream) { hipStreamSynchronize(a->copy_stream); hipStreamDestroy(a->copy_stream); }
    for (hipEvent_t ev : a->chunk_ev) if (ev) hipEventDestroy(ev);
    if (a->copies_done) hipEventDestroy(a->copies_done);
    hipFree(a->gray_a); hipFree(a->gray_b); hipFree(a->obs); hipFree(a->fin); hipFree(a->done_out);
    hipFree(a->racc); hipFree(a->reward_out); hipFree(a->ty); hipFree(a->tx);
    hipFree(a->was_real_done); hipFree(a->needs_reset); hipFree(a->mode); hipFree(a->buf_valid); hipFree(a->exec_flag);
    hipFree(a->noop_override);
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
    case TBX_BUF_AGENT_OBS:
        if (!a.obs) return no_stack(e);
        p = a.obs; b = N * a.cfg.out_h * a.cfg.out_w * a.cfg.stack; break;
    case TBX_BUF_AGENT_PLANE:
        if (!newest_plane(a, e->n)) return e->fail(TBX_E_INVALID, "TBX_BUF_AGENT_PLANE needs tbx_agent_config_t::new_plane = 1 or 2");
        p = newest_plane(a, e->n); b = N * a.cfg.out_h * a.cfg.out_w; break;
    case TBX_BUF_AGENT_RING:
        if (!a.ring) return e->fail(TBX_E_INVALID, "TBX_BUF_AGENT_RING needs tbx_agent_config_t::new_plane = 2");
        p = a.ring; b = N * a.cfg.out_h * a.cfg.out_w * a.cfg.stack; break;
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
        cfg->noop_max > 1000 || cfg->stack_fill < 0 || cfg->stack_fill > 1 || cfg->new_plane < 0 || cfg->new_plane > 2)
        return e->fail(TBX_E_INVALID, "agent config out of range (skip 1..64, stack 1..4, 1 <= out <= frame, out_w <= 128, out_h*out_w <= 7056, noop_max 0..1000, stack_fill 0..1, new_plane 0..2)");
    if ((H + cfg->out_h - 1) / cfg->out_h + 1 > MAX_TAPS || (W + cfg->out_w - 1) / cfg->out_w + 1 > MAX_TAPS)
        return e->fail(TBX_E_UNSUPPORTED, "agent: the resize ratio needs more than 8 taps per axis");
    AHIP(hipSetDevice(e->device));
    // nothing of this handle may be in flight while the agent layer's buffers are freed and made anew: not a queued kernel,
    // not the resident step kernel of tbx_step1 (tbx_use_stream stops it)
    AHIP(tbx_use_stream(e, e->stream));
    AHIP(hipStreamSynchronize(e->stream));
    tbx_agent_free(e);
    AgentState* a = new AgentState();
    e->agent = a;
    a->cfg = *cfg;
    a->H = H; a->W = W;
    a->force_generic = e->opt[TBX_OPT_AGENT_GENERIC] != 0;
    const size_t N = (size_t)e->n;
    if (a->force_generic || !e->ops->agent_fused()) {
        AHIP(hipMalloc((void**)&a->gray_a, N * H * W));
        AHIP(hipMalloc((void**)&a->gray_b, N * H * W));
    }
    AHIP(hipMalloc((void**)&a->was_real_done, N));
    AHIP(hipMalloc((void**)&a->needs_reset, N));
    AHIP(hipMalloc((void**)&a->mode, N));
    AHIP(hipMalloc((void**)&a->buf_valid, N));
    AHIP(hipMalloc((void**)&a->exec_flag, N));
    AHIP(hipMemset(a->was_real_done, 1, N));         // EpisodicLifeEnv.__init__: was_real_done = True
    AHIP(hipMemset(a->needs_reset, 0, N));
    AHIP(hipMemset(a->mode, 0, N));
    AHIP(hipMemset(a->buf_valid, 0, N));             // MaxAndSkipEnv.__init__: _obs_buffer = np.zeros
    AHIP(hipMemset(a->exec_flag, 0, N));
    {
        int rc = e->ops->agent_prepare(e);
        if (rc) return rc;
    }
    if (cfg->new_plane == 2) {
        AHIP(hipMalloc((void**)&a->ring, N * cfg->out_h * cfg->out_w * cfg->stack));
        AHIP(hipMemset(a->ring, 0, N * cfg->out_h * cfg->out_w * cfg->stack));
    } else {
        AHIP(hipMalloc((void**)&a->obs, N * cfg->out_h * cfg->out_w * cfg->stack));
        AHIP(hipMemset(a->obs, 0, N * cfg->out_h * cfg->out_w * cfg->stack));
    }
    if (cfg->new_plane == 1) {
        AHIP(hipMalloc((void**)&a->plane, N * cfg->out_h * cfg->out_w));
        AHIP(hipMemset(a->plane, 0, N * cfg->out_h * cfg->out_w));
    }
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

int tbx_agent_set_noops(tbx_engine* e, const int32_t* counts_host)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent) return e->fail(TBX_E_INVALID, "tbx_agent_init has not been called");
    AgentState& a = *e->agent;
    AHIP(hipSetDevice(e->device));
    AHIP(tbx_use_stream(e, e->stream));
    AHIP(hipStreamSynchronize(e->stream));
    if (!counts_host) {
        hipFree(a.noop_override);
        a.noop_override = nullptr;
        return TBX_OK;
    }
    if (!e->ops->agent_reset_supported())
        return e->fail(TBX_E_UNSUPPORTED, "agent: no-op resets are not available for this engine state");
    if (!a.noop_override) AHIP(hipMalloc((void**)&a.noop_override, (size_t)e->n * sizeof(int32_t)));
    AHIP(hipMemcpy(a.noop_override, counts_host, (size_t)e->n * sizeof(int32_t), hipMemcpyHostToDevice));
    return TBX_OK;
}

int tbx_agent_reset(tbx_engine* e, uint8_t* obs_host)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent) return e->fail(TBX_E_INVALID, "tbx_agent_init has not been called");
    AgentState& a = *e->agent;
    if (obs_host && !a.obs) return no_stack(e);
    AHIP(hipSetDevice(e->device));
    AHIP(tbx_use_stream(e, e->stream));
    const int n = e->n, tb = 256, gb = (n + tb - 1) / tb;
    const size_t N = (size_t)n;
    int rc;
    if (e->ops->agent_reset_supported()) {
        // reset() of every env's wrapper stack (AgentResetProc::run)
        hipLaunchKernelGGL(agent_fill_u8_kernel, dim3(gb), dim3(tb), 0, e->stream, a.kind, (uint8_t)1, n);
        rc = e->ops->agent_reset_envs(e, reset_args(e), e->stream);
        if (rc) return rc;
    } else {
        if (wants_wrappers(a))
            return e->fail(TBX_E_UNSUPPORTED, "agent: episodic-life / fire-reset / no-op-reset are not available for this engine state");
        // Monitor.reset + a plain new game; the observation is its raw frame
        AHIP(hipMemsetAsync(a.ep_ret, 0, N * sizeof(int32_t), e->stream));
        AHIP(hipMemsetAsync(a.ep_len, 0, N * sizeof(int32_t), e->stream));
        AHIP(hipMemsetAsync(a.needs_reset, 0, N, e->stream));
        hipLaunchKernelGGL(agent_fill_u8_kernel, dim3(gb), dim3(tb), 0, e->stream, a.mode, (uint8_t)1, n);
        rc = e->ops->new_game(e, nullptr, e->stream);
        if (rc) return rc;
    }
    AHIP(hipMemsetAsync(a.ep_done, 0, N, e->stream));   // records of games that ended inside the reset procedure are not reported here
    if (a.ring) a.head = (a.head + 1) % a.cfg.stack;
    rc = observe(e, 1, e->stream);
    if (rc) return rc;
    if (obs_host)
        AHIP(hipMemcpyAsync(obs_host, a.obs, N * a.cfg.out_h * a.cfg.out_w * a.cfg.stack, hipMemcpyDeviceToHost, e->stream));
    AHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_agent_ring_head(tbx_engine* e, int32_t* out_head)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent) return e->fail(TBX_E_INVALID, "tbx_agent_init has not been called");
    if (!e->agent->ring) return e->fail(TBX_E_INVALID, "tbx_agent_ring_head needs tbx_agent_config_t::new_plane = 2");
    if (!out_head) return e->fail(TBX_E_INVALID, "output pointer is NULL");
    *out_head = e->agent->head;
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

// the copies of an agent step's outputs into the caller's host buffers, queued on `s`
static int agent_queue_outputs(tbx_engine* e, const tbx_agent_host_out_t& out, hipStream_t s)
{
    AgentState& a = *e->agent;
    const size_t N = (size_t)e->n, px = (size_t)a.cfg.out_h * a.cfg.out_w;
    if (out.plane && !newest_plane(a, e->n)) return e->fail(TBX_E_INVALID, "the newest plane needs tbx_agent_config_t::new_plane = 1 or 2");
    if (out.obs && !a.obs) return no_stack(e);
    if (out.reward) AHIP(hipMemcpyAsync(out.reward, a.reward_out, N * sizeof(float), hipMemcpyDeviceToHost, s));
    if (out.done) AHIP(hipMemcpyAsync(out.done, a.done_out, N, hipMemcpyDeviceToHost, s));
    if (out.ep_done) AHIP(hipMemcpyAsync(out.ep_done, a.ep_done, N, hipMemcpyDeviceToHost, s));
    if (out.ep_return) AHIP(hipMemcpyAsync(out.ep_return, a.ep_ret_out, N * sizeof(float), hipMemcpyDeviceToHost, s));
    if (out.ep_length) AHIP(hipMemcpyAsync(out.ep_length, a.ep_len_out, N * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (out.plane) AHIP(hipMemcpyAsync(out.plane, newest_plane(a, e->n), N * px, hipMemcpyDeviceToHost, s));
    if (out.obs) AHIP(hipMemcpyAsync(out.obs, a.obs, N * px * a.cfg.stack, hipMemcpyDeviceToHost, s));
    return TBX_OK;
}

int tbx_agent_step_begin(tbx_engine* e, const int32_t* actions_host, const tbx_agent_host_out_t* out)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent) return e->fail(TBX_E_INVALID, "tbx_agent_init has not been called");
    if (!actions_host || !out) return e->fail(TBX_E_INVALID, "actions / output descriptor is NULL");
    AgentState& a = *e->agent;
    if (e->pending_kind) return e->fail(TBX_E_INVALID, "tbx_agent_step_begin: the previous step has not been ended (tbx_agent_step_end / tbx_step_end)");
    e->ended_early_kind = 0;
    if (out->plane && !newest_plane(a, e->n)) return e->fail(TBX_E_INVALID, "the newest plane needs tbx_agent_config_t::new_plane = 1 or 2");
    if (out->obs && !a.obs) return no_stack(e);
    AHIP(hipSetDevice(e->device));
    AHIP(tbx_use_stream(e, e->stream));
    AHIP(tbx_gather_before_step(e, e->stream));
    const size_t N = (size_t)e->n, px = (size_t)a.cfg.out_h * a.cfg.out_w;
    const size_t io_bytes = (3 * N + 1) * sizeof(uint32_t) + 2 * N;
    if (!a.host_actions) {
        AHIP(hipHostMalloc((void**)&a.host_actions, N * sizeof(int32_t), hipHostMallocDefault));
        AHIP(hipMalloc((void**)&a.io_dev, io_bytes));
        AHIP(hipHostMalloc((void**)&a.io_host, io_bytes, hipHostMallocDefault));
    }
    memcpy(a.host_actions, actions_host, N * sizeof(int32_t));
    AHIP(hipMemcpyAsync(e->actions, a.host_actions, N * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    ActionSource src{};
    src.actions = e->actions;
    src.single_env = -1;
    // Batches of 2 048 envs and more (fused observation kernels): four chunks, the copy of chunk c beside the kernel of chunk c + 1.
    const int chunks = (observe_in_chunks(e) && e->n >= 2048 && (out->plane || out->obs)) ? 4 : 1;
    int rc = agent_step_async(e, src, e->stream, chunks == 1);
    if (rc) return rc;
    if (chunks == 1) {
        // the observation first (the long copy starts as soon as the observation kernel has finished), the small block behind it
        if (out->plane) AHIP(hipMemcpyAsync(out->plane, newest_plane(a, e->n), N * px, hipMemcpyDeviceToHost, e->stream));
        if (out->obs) AHIP(hipMemcpyAsync(out->obs, a.obs, N * px * a.cfg.stack, hipMemcpyDeviceToHost, e->stream));
    } else {
        if (!a.copy_stream) {
            AHIP(hipStreamCreateWithFlags(&a.copy_stream, hipStreamNonBlocking));
            for (int c = 0; c < 8; c++) AHIP(hipEventCreateWithFlags(&a.chunk_ev[c], hipEventDisableTiming));
            AHIP(hipEventCreateWithFlags(&a.copies_done, hipEventDisableTiming));
        }
        for (int c = 0; c < chunks; c++) {
            const int first = (int)((long)e->n * c / chunks), end = (int)((long)e->n * (c + 1) / chunks);
            rc = observe_chunk(e, first, end, e->stream);
            if (rc) return rc;
            AHIP(hipEventRecord(a.chunk_ev[c], e->stream));
            AHIP(hipStreamWaitEvent(a.copy_stream, a.chunk_ev[c], 0));
            const size_t off = (size_t)first * px, cnt = (size_t)(end - first) * px;
            if (out->plane) AHIP(hipMemcpyAsync(out->plane + off, newest_plane(a, e->n) + off, cnt, hipMemcpyDeviceToHost, a.copy_stream));
            if (out->obs) AHIP(hipMemcpyAsync(out->obs + off * a.cfg.stack, a.obs + off * a.cfg.stack, cnt * a.cfg.stack, hipMemcpyDeviceToHost, a.copy_stream));
        }
        AHIP(hipEventRecord(a.copies_done, a.copy_stream));
        AHIP(hipStreamWaitEvent(e->stream, a.copies_done, 0));     // the engine's stream is the tail everything else orders itself behind
    }
    hipLaunchKernelGGL(agent_pack_outputs_kernel, dim3((e->n + 255) / 256), dim3(256), 0, e->stream, a.reward_out, a.ep_ret_out, a.ep_len_out,
                       a.done_out, a.ep_done, e->err_flag, a.io_dev, e->n);
    AHIP(hipGetLastError());
    AHIP(hipMemcpyAsync(a.io_host, a.io_dev, io_bytes, hipMemcpyDeviceToHost, e->stream));
    a.host_out = *out;
    a.host_pending = true;
    e->pending_kind = 2;
    return TBX_OK;
}

int tbx_agent_step_end(tbx_engine* e)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent) return e->fail(TBX_E_INVALID, "tbx_agent_init has not been called");
    if (!e->agent->host_pending) {
        if (e->ended_early_kind == 2) {        // another call on the handle ended the step (tbx_finish_pending): outputs delivered
            e->ended_early_kind = 0;
            return e->ended_early_rc ? e->fail(e->ended_early_rc, e->ended_early_msg) : TBX_OK;
        }
        return e->fail(TBX_E_INVALID, "tbx_agent_step_end without tbx_agent_step_begin");
    }
    return tbx_agent_deliver(e);
}

}  // extern "C"

// the waiting half of tbx_agent_step_end (also reached through tbx_finish_pending when another call ends the step)
int tbx_agent_deliver(tbx_engine* e)
{
    AgentState& a = *e->agent;
    AHIP(hipSetDevice(e->device));
    AHIP(hipStreamSynchronize(e->stream));
    a.host_pending = false;
    e->pending_kind = 0;
    const size_t N = (size_t)e->n;
    const tbx_agent_host_out_t& o = a.host_out;
    const uint32_t* io = a.io_host;
    const uint8_t* bytes = reinterpret_cast<const uint8_t*>(io + 3 * N + 1);
    if (o.reward) memcpy(o.reward, io, N * sizeof(float));
    if (o.ep_return) memcpy(o.ep_return, io + N, N * sizeof(float));
    if (o.ep_length) memcpy(o.ep_length, io + 2 * N, N * sizeof(int32_t));
    if (o.done) memcpy(o.done, bytes, N);
    if (o.ep_done) memcpy(o.ep_done, bytes + N, N);
    const uint32_t f = io[3 * N];
    if (f) {
        AHIP(hipMemsetAsync(e->err_flag, 0, sizeof f, e->stream));
        if (f & 2u)
            return e->fail(TBX_E_NEEDS_RESET, "an env was stepped after its game ended inside EpisodicLifeEnv's no-op step (bench.Monitor raises here)");
        return e->fail(TBX_E_ACTION, "an illegal ALE action id was passed (treated as NOOP)");
    }
    return TBX_OK;
}

extern "C" {

int tbx_agent_fetch(tbx_engine* e, const tbx_agent_host_out_t* out)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent) return e->fail(TBX_E_INVALID, "tbx_agent_init has not been called");
    if (!out) return e->fail(TBX_E_INVALID, "output descriptor is NULL");
    AHIP(hipSetDevice(e->device));
    AHIP(tbx_use_stream(e, e->stream));
    int rc = agent_queue_outputs(e, *out, e->stream);
    if (rc) return rc;
    AHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_agent_step(tbx_engine* e, const int32_t* actions_host, float* reward_host, uint8_t* done_host, uint8_t* obs_host)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent) return e->fail(TBX_E_INVALID, "tbx_agent_init has not been called");
    if (!actions_host) return e->fail(TBX_E_INVALID, "actions pointer is NULL");
    AgentState& a = *e->agent;
    if (obs_host && !a.obs) return no_stack(e);
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
    // an illegal action id is reported like tbx_step does; so is a step that bench.Monitor would have refused
    return report_flags(e);
}

}  // extern "C"
