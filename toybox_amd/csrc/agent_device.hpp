// agent_device.hpp -- device helpers shared by the generic warp kernel (agent.hip) and the per-game fused variants.
#pragma once

#include "tbx_common.hpp"

struct AgentTaps {          // area-resize taps of one output row / column
    int32_t start, n;
    uint8_t w[8];           // overlap lengths in units of 1/out of a source pixel; their sum is the source extent
};


// what a game's fused observation kernel needs from the agent layer
struct AgentWarpArgs {
    const uint8_t* fin;        // [N] game ended during this agent step (observation = reset frame alone)
    const int32_t* racc;       // [N] summed reward
    const AgentTaps* tx;       // [out_w] column taps
    uint8_t* obs;              // [N][out_h][out_w][stack]
    float* reward_out;         // [N]
    uint8_t* done_out;         // [N]
    int H, W, oh, ow, stack, clip, reset_mode;
    int two_frames;            // skip >= 2: the observation is max(frame A, frame B); otherwise frame B alone
    uint64_t magic;            // floor(2^42 / (H*W)) + 1
};

__device__ __forceinline__ uint32_t bytemax4(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t x = (a >> (8 * k)) & 255u, y = (b >> (8 * k)) & 255u;
        r |= (x > y ? x : y) << (8 * k);
    }
    return r;
}

// this lane's taps for one output column: first source pixel and 8 byte weights (zero padded)
struct ColTaps { int start; uint32_t wlo, whi; };

__device__ __forceinline__ ColTaps load_col(const AgentTaps* tx, int ox, int ow)
{
    const AgentTaps t = tx[ox < ow ? ox : 0];
    ColTaps c;
    c.start = t.start;
    c.wlo = (uint32_t)t.w[0] | ((uint32_t)t.w[1] << 8) | ((uint32_t)t.w[2] << 16) | ((uint32_t)t.w[3] << 24);
    c.whi = (uint32_t)t.w[4] | ((uint32_t)t.w[5] << 8) | ((uint32_t)t.w[6] << 16) | ((uint32_t)t.w[7] << 24);
    return c;
}

// horizontal area sum of one staged source row for one output column: 8 bytes from `start`, dotted with the weights
__device__ __forceinline__ uint32_t hsum(const uint8_t* row, const ColTaps& c)
{
    const uint32_t* w = reinterpret_cast<const uint32_t*>(row + (c.start & ~3));
    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
    const uint32_t sh = (uint32_t)(c.start & 3);
    const uint32_t b0 = __builtin_amdgcn_alignbyte(d1, d0, sh), b1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
    return __builtin_amdgcn_udot4(b0, c.wlo, __builtin_amdgcn_udot4(b1, c.whi, 0u, false), false);
}


// rolls `val` into the frame stack at px (S bytes per pixel, newest last); fresh: older slots become 0
template <int S>
__device__ __forceinline__ void stack_push(uint8_t* px, uint32_t val, bool fresh)
{
    if (S == 4) {
        const uint32_t old = fresh ? 0u : *reinterpret_cast<uint32_t*>(px);
        *reinterpret_cast<uint32_t*>(px) = (old >> 8) | (val << 24);
    } else {
#pragma unroll
        for (int c = 0; c + 1 < S; c++) px[c] = fresh ? (uint8_t)0 : px[c + 1];
        px[S - 1] = (uint8_t)val;
    }
}

// The observation of one env is first collected in LDS (one byte per output pixel) and committed to the frame stack in
// one sweep, so that the read-modify-write of the stack runs with many loads in flight instead of one dependent
// load -> store per output row.  out_h * out_w <= AGENT_MAX_OUT_PX.
constexpr int AGENT_MAX_OUT_PX = 84 * 84;

template <int S>
__device__ __forceinline__ void stack_commit(const uint8_t* vals, uint8_t* o, int n_px, int lane, bool fresh)
{
    __builtin_amdgcn_wave_barrier();
    if (S == 4) {
        uint32_t* o4 = reinterpret_cast<uint32_t*>(o);
        int i = lane;
        for (; i + 7 * 64 < n_px; i += 8 * 64) {
            uint32_t old[8];
#pragma unroll
            for (int k = 0; k < 8; k++) old[k] = fresh ? 0u : o4[i + 64 * k];
#pragma unroll
            for (int k = 0; k < 8; k++) o4[i + 64 * k] = (old[k] >> 8) | ((uint32_t)vals[i + 64 * k] << 24);
        }
        for (; i < n_px; i += 64) {
            const uint32_t old = fresh ? 0u : o4[i];
            o4[i] = (old >> 8) | ((uint32_t)vals[i] << 24);
        }
    } else {
        for (int i = lane; i < n_px; i += 64) stack_push<S>(o + (size_t)i * S, vals[i], fresh);
    }
}

// ------------------------------------------------------------------ reset-time wrappers + episode monitor (8f rank 2)

// what a game's agent-reset kernel needs (see tbx_agent_config_t)
struct AgentResetArgs {
    const uint8_t* kind;        // [N] 0 = nothing to do, 1 = a life was lost (episodic life), 2 = game over
    int skip, episodic_life, fire_reset, noop_max;
    uint64_t noop_seed, env_offset;
    uint32_t fire_buttons, third_buttons;   // buttons of action #1 and action #2 of the game's action set
    int32_t *ep_ret, *ep_len, *ep_index, *prev_lives;   // [N] monitor / episodic-life state
    uint8_t* ep_done;           // [N] episode records of this agent step
    float* ep_ret_out;
    int32_t* ep_len_out;
};

struct AgentMonitor {
    int32_t ep_ret, ep_len, ep_index, prev_lives;
    bool emitted;
    int32_t out_ret, out_len;
};

// The reset path of the wrapper stack NoopResetEnv -> MaxAndSkipEnv -> Monitor -> EpisodicLifeEnv -> FireResetEnv
// (baselines/baselines/common/atari_wrappers.py:108-191, bench/monitor.py:51-76), run in-kernel for one env.
// Env provides step(buttons), new_game(), lives(), score(); every call is wave-uniform for wave-per-env games.
template <class Env>
struct AgentResetProc {
    Env& env;
    const AgentResetArgs& r;
    AgentMonitor& m;
    uint64_t env_global;
    bool was_real_done;

    __device__ __forceinline__ void inner_real_reset()      // Monitor.reset + NoopResetEnv.reset
    {
        m.ep_ret = 0; m.ep_len = 0; m.ep_index += 1;
        env.new_game();
        if (r.noop_max > 0) {
            const int k = 1 + (int)(tbx_splitmix64(r.noop_seed ^ (env_global << 32) ^ (uint64_t)(uint32_t)m.ep_index) % (uint64_t)r.noop_max);
            for (int j = 0; j < k; j++) {
                env.step(0u);
                if (env.lives() <= 0) env.new_game();
            }
        }
    }
    __device__ __forceinline__ bool mstep(uint32_t buttons)  // MaxAndSkipEnv.step under Monitor
    {
        int rsum = 0;
        bool done = false;
        for (int i = 0; i < r.skip && !done; i++) {
            const int s0 = env.score();
            env.step(buttons);
            const int dsc = env.score() - s0;
            rsum += dsc > 0 ? dsc : 0;
            if (env.lives() <= 0) done = true;
        }
        m.ep_ret += rsum; m.ep_len += 1;
        if (done) { m.emitted = true; m.out_ret = m.ep_ret; m.out_len = m.ep_len; }
        return done;
    }
    __device__ __forceinline__ void elife_reset()            // EpisodicLifeEnv.reset
    {
        if (was_real_done || !r.episodic_life) inner_real_reset();
        else if (mstep(0u)) inner_real_reset();              // no-op step to advance from the lost-life state
                                                             // (a game that ends inside it starts over: own rule,
                                                             // the wrapper stack would raise on its next step)
        m.prev_lives = env.lives();
    }
    __device__ __forceinline__ bool elife_step(uint32_t buttons)   // EpisodicLifeEnv.step
    {
        bool d = mstep(buttons);
        was_real_done = d;
        const int l = env.lives();
        if (r.episodic_life && l < m.prev_lives && l > 0) d = true;
        m.prev_lives = l;
        return d;
    }
    __device__ __forceinline__ void run(int kind)            // FireResetEnv.reset (or the reset below it)
    {
        was_real_done = kind == 2;
        elife_reset();
        if (r.fire_reset) {
            if (elife_step(r.fire_buttons)) elife_reset();
            if (elife_step(r.third_buttons)) elife_reset();
        }
    }
};
