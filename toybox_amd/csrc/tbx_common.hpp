// tbx_common.hpp -- shared pieces of the gfx950 engine: RNG, action tables, wave helpers,
// the engine object and the per-game operations table.  gfx950 (CDNA4, wave64) only.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string>

#include "../../include/toybox_amd.h"

#define TBX_WAVE 64
#define TBX_WAVES_PER_BLOCK 4
#define TBX_BLOCK (TBX_WAVE * TBX_WAVES_PER_BLOCK)

#define TBX_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t _e = (call);                                                            \
        if (_e != hipSuccess) {                                                            \
            return e->fail(TBX_E_NO_DEVICE, std::string(#call) + ": " + hipGetErrorString(_e)); \
        }                                                                                  \
    } while (0)

// ------------------------------------------------------------------ device helpers

// xoroshiro128+ (55,14,36); pinned by tests/golden/rng_kat.json
struct Rng {
    uint64_t s0, s1;
    __device__ __forceinline__ uint64_t next()
    {
        uint64_t r = s0 + s1;
        uint64_t t = s1 ^ s0;
        s0 = ((s0 << 55) | (s0 >> 9)) ^ t ^ (t << 14);
        s1 = (t << 36) | (t >> 28);
        return r;
    }
    __device__ __forceinline__ Rng child()
    {
        Rng c;
        c.s0 = next();
        c.s1 = next();
        return c;
    }
    // uniform in [0,n): widening multiply + rejection zone (rand's UniformInt::sample_single)
    __device__ __forceinline__ uint64_t range(uint64_t n)
    {
        if (n <= 1) return 0;
        uint64_t zone = (n << __clzll((long long)n)) - 1;
        for (;;) {
            uint64_t v = next();
            uint64_t lo = v * n;
            if (lo <= zone) return __umul64hi(v, n);
        }
    }
};

__host__ __device__ __forceinline__ void tbx_seed_state(uint32_t seed, uint64_t& s0, uint64_t& s1)
{
    s0 = 0x193a6754a8a7d469ULL ^ (uint64_t)seed;
    s1 = 0x97830e05113ba7bbULL;
}

__host__ __device__ __forceinline__ uint64_t tbx_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

// ALE action id -> button mask (envs/atari/constants.py:16-35); 0xFF = illegal id
__host__ __device__ __forceinline__ uint32_t tbx_ale_buttons(int a)
{
    // packed table, one byte per action
    const uint8_t L = TBX_BTN_LEFT, R = TBX_BTN_RIGHT, U = TBX_BTN_UP, D = TBX_BTN_DOWN, F = TBX_BTN_BUTTON1;
    switch (a) {
    case 0: return 0;
    case 1: return F;
    case 2: return U;
    case 3: return R;
    case 4: return L;
    case 5: return D;
    case 6: return U | R;
    case 7: return U | L;
    case 8: return D | R;
    case 9: return D | L;
    case 10: return U | F;
    case 11: return R | F;
    case 12: return L | F;
    case 13: return D | F;
    case 14: return U | R | F;
    case 15: return U | L | F;
    case 16: return D | R | F;
    case 17: return D | L | F;
    default: return 0xFFu;
    }
}

__host__ __device__ __forceinline__ int tbx_legal_count(int game)
{
    return game == TBX_GAME_BREAKOUT ? 4 : game == TBX_GAME_GRIDWORLD ? 5 : 6;
}
__host__ __device__ __forceinline__ int tbx_legal_action(int game, int i)
{
    // Breakout [0,1,3,4]; Amidar [0..5]; SpaceInvaders [0,1,3,4,11,12]; GridWorld [0,2,3,4,5]
    if (game == TBX_GAME_BREAKOUT) return i == 0 ? 0 : i == 1 ? 1 : i == 2 ? 3 : 4;
    if (game == TBX_GAME_GRIDWORLD) return i == 0 ? 0 : i + 1;
    if (game == TBX_GAME_AMIDAR) return i;
    return i == 0 ? 0 : i == 1 ? 1 : i == 2 ? 3 : i == 3 ? 4 : i == 4 ? 11 : 12;
}

// how the step kernels obtain their action
struct ActionSource {
    const int32_t* actions;   // device array, or nullptr for synthetic
    uint64_t seed, t, env_offset;
    int single_env;           // >= 0: only this env steps, with buttons `single_buttons`
    uint32_t single_buttons;
    // agent layer (MaxAndSkipEnv.step, atari_wrappers.py:201-216): sum the rewards of the agent step's frames into
    // acc_reward[env] and latch acc_done[env]; an env whose game has ended does not run the remaining frames
    int32_t* acc_reward;
    uint8_t* acc_done;
    // one launch runs `frames` frames of the same action with the state held in registers (0 means 1; games with
    // GameOps::multi_frame_step run the whole action repeat in ONE launch); frame0 = frames of this agent step that earlier
    // launches already ran
    int frames;
    int frame0;
    // MaxAndSkipEnv._obs_buffer, kept as two persistent state snapshots per env ("slot A" / "slot B", what the rasteriser
    // needs to repaint the frame): written after snap_a_after / snap_b_after frames of the agent step (0 = never), i.e. by
    // frame skip-2 and frame skip-1, and only by envs that get that far.  buf_valid[env] bit 0 / 1: the slot has been written
    // since construction (until then it is the zero frame of np.zeros).
    int snap_a_after, snap_b_after;
    uint8_t* buf_valid;
    // single-frame launches (games without multi_frame_step): 1 if the env ran this frame, for the snapshot kernel that follows
    uint8_t* exec_flag;
};

// this env's game ended in an earlier launch of the same agent step: MaxAndSkipEnv has left its loop
__device__ __forceinline__ bool tbx_agent_env_finished(const ActionSource& src, int env)
{
    return src.acc_done && src.frame0 > 0 && src.acc_done[env] != 0;
}

// frame: index of this frame inside the launch
__device__ __forceinline__ void tbx_accumulate(const ActionSource& src, int env, int32_t rew, bool is_done, int frame = 0)
{
    if (!src.acc_reward) return;
    const bool first = src.frame0 + frame == 0;
    src.acc_reward[env] = (first ? 0 : src.acc_reward[env]) + rew;
    src.acc_done[env] = is_done ? 1 : 0;
}

// which buffer slots frame number `done_frames` (1-based count of frames run in this agent step) writes: bit 0 = A, bit 1 = B
__device__ __forceinline__ uint32_t tbx_snap_slots(const ActionSource& src, int frame)
{
    const int g = src.frame0 + frame + 1;
    return (g == src.snap_a_after ? 1u : 0u) | (g == src.snap_b_after ? 2u : 0u);
}

__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t wave_uniform64(uint64_t v)
{
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
}

// value of lane `src` for a WAVE-UNIFORM src: v_readlane_b32 (a few cycles, result in an SGPR) instead of ds_bpermute_b32
__device__ __forceinline__ int bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ uint32_t bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }

__device__ __forceinline__ uint32_t gray_of(uint32_t rgba)
{
    uint32_t r = rgba & 255u, g = (rgba >> 8) & 255u, b = (rgba >> 16) & 255u;
    return (77u * r + 150u * g + 29u * b + 128u) >> 8;
}

__host__ __device__ __forceinline__ uint32_t pack_color(tbx_color_t c)
{
    return (uint32_t)c.r | ((uint32_t)c.g << 8) | ((uint32_t)c.b << 16) | ((uint32_t)c.a << 24);
}
__host__ __device__ __forceinline__ tbx_color_t unpack_color(uint32_t v)
{
    tbx_color_t c;
    c.r = (uint8_t)v; c.g = (uint8_t)(v >> 8); c.b = (uint8_t)(v >> 16); c.a = (uint8_t)(v >> 24);
    return c;
}

// double -> pixel coordinate: clamp, then truncate toward zero (matches the oracle's f2i)
__device__ __forceinline__ int f2i(double v)
{
    if (!(v > -1.0e6)) v = -1.0e6;
    if (v > 1.0e6) v = 1.0e6;
    return (int)v;
}

// ------------------------------------------------------------------ resident single-env step ("server" kernel)
//
// The reference's whole user-facing surface is single-env: Toybox.apply_ale_action + get_score / get_lives / game_over, one FFI
// round trip per frame (test/benchmark.py:50-56).  On a GPU that loop is pure latency: a launch, a host-device copy each
// way and a stream synchronisation are ~30 us.  For a one-env engine tbx_step1 therefore talks to a RESIDENT kernel
// instead: one wave that waits on a mailbox in host-coherent pinned memory, runs the game's ordinary step body for env 0
// and posts the outputs back -- two PCIe hops per frame, no launch, no copy, no synchronisation.  The wave leaves by
// itself after TBX_SERVE_IDLE_TICKS without a request (or when told to), and every other entry point of the handle stops
// it first, so nothing else ever runs beside it on the env's state.
struct TbxServeCtl {
    // host -> device: ONE 64-bit word, so that a single PCIe read per poll brings the whole request:
    //   bits 0..31 request number, 32..47 ALE action id (int16), 48..51 TBX_STEP_* flags, 52..53 frame wanted (0 none, 1 gray,
    //   2 RGB, 3 RGBA: the wave rasterises the env into `frame_dev` after the step), 63 "leave now"
    uint64_t req;
    uint64_t frame_dev;      // device address of the engine's mapped pinned frame buffer (written once before the launch)
    uint64_t _pad0[6];
    // device -> host (its own cache line): outputs, then the request number they belong to (written last)
    int32_t reward, lives, score;
    uint32_t done_err;       // bit 0 done, bit 1 illegal action id, bit 2 the frame was asked for but this kernel cannot paint it
    uint32_t ack_seq;
    uint32_t exited;
    uint32_t _pad1[10];
};
constexpr uint32_t TBX_SERVE_FRAME_SHIFT = 4;   // within the 8 flag bits of the request word
constexpr unsigned long long TBX_SERVE_IDLE_TICKS = 5000000ull;   // 50 ms of the 100 MHz s_memrealtime clock
constexpr uint64_t TBX_SERVE_STOP = 1ull << 63;

__host__ __device__ __forceinline__ uint64_t tbx_serve_word(uint32_t seq, int action, uint32_t flags)
{
    // ids outside int16 are all illegal anyway: clamp them onto one illegal id
    const int a = action < -32768 || action > 32767 ? 32767 : action;
    return (uint64_t)seq | ((uint64_t)(uint16_t)(int16_t)a << 32) | ((uint64_t)(flags & 0xFFu) << 48);
}

// The resident kernel is ONE block of TBX_SERVE_WAVES waves.  Wave 0 waits for requests and steps: step(src, flags) runs one
// frame of env 0 on it (every lane of wave 0 calls it; outputs land in out_* [0]).  When the request wants the picture, ALL
// waves paint: render(channels, frame, part, split) rasterises units part, part + split, ... of env 0 into `frame` (host
// memory, mapped) and returns false if this game's kernel cannot -- a lone wave needs ~45 us for a frame (one dependent
// instruction stream), eight need ~8.  The other waves sleep at the block barrier in between and take no issue slots.
constexpr int TBX_SERVE_WAVES = 8;

template <class StepFn, class RenderFn>
__device__ __forceinline__ void tbx_serve_loop(TbxServeCtl* ctl, int lane, StepFn step, RenderFn render, const int32_t* out_reward, const uint8_t* out_done,
                                               const int32_t* out_lives, const int32_t* out_score, uint32_t* err_flag)
{
    __shared__ uint32_t cmd[2];              // [0]: 0 = nothing to paint, 1 / 3 / 4 = paint that many channels, ~0u = leave; [1]: "cannot paint"
    const int wave = wave_uniform((int)(threadIdx.x >> 6));
    uint32_t last = __hip_atomic_load(&ctl->ack_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint8_t* const frame = reinterpret_cast<uint8_t*>(__hip_atomic_load(&ctl->frame_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    for (;;) {
        uint32_t seq = last, err = 0;
        if (wave == 0) {
            bool leave = false;
            uint32_t hi = 0;
            for (;;) {
                uint64_t w = last;
                bool idle_out = false;
                if (lane == 0) {
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    unsigned polls = 0;
                    for (;;) {
                        w = __hip_atomic_load(&ctl->req, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                        if ((uint32_t)w != last || (w & TBX_SERVE_STOP)) break;
                        if ((++polls & 63u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > TBX_SERVE_IDLE_TICKS) { idle_out = true; break; }
                    }
                }
                seq = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w);
                hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w >> 32));
                idle_out = __builtin_amdgcn_readfirstlane((int)idle_out) != 0;
                if (seq != last) break;                          // a request (one that raced with the stop bit is served first)
                if ((hi >> 31) || idle_out) { leave = true; break; }   // told to leave, or idle for too long
            }
            if (leave) {
                if (lane == 0) cmd[0] = ~0u;
            } else {
                const int action = (int)(int16_t)(hi & 0xFFFFu);
                const uint32_t flags = (hi >> 16) & 0xFFu;
                ActionSource src{};
                uint32_t buttons = tbx_ale_buttons(action);
                if (buttons == 0xFFu) { buttons = 0; err = 2; }  // illegal id: NOOP + TBX_E_ACTION, as in the batch kernels
                src.single_env = 0;
                src.single_buttons = buttons;
                step(src, flags & 0x0Fu);
                const uint32_t want = (flags >> TBX_SERVE_FRAME_SHIFT) & 3u;
                if (want) __threadfence();                       // what the step stored, for the waves that paint
                if (lane == 0) { cmd[0] = want == 0 ? 0u : want == 1 ? 1u : want == 2 ? 3u : 4u; cmd[1] = 0u; }
            }
        }
        __syncthreads();
        const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cmd[0]);
        if (c == ~0u) {
            if (wave == 0 && lane == 0) __hip_atomic_store(&ctl->exited, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        if (c) {
            if (!render((int)c, frame, wave, TBX_SERVE_WAVES) && lane == 0) cmd[1] = 1u;
            __threadfence_system();                              // this wave's part of the frame is in host memory ...
        }
        __syncthreads();                                         // ... and so is everybody's, before the acknowledgement
        if (wave == 0) {
            if (c && __builtin_amdgcn_readfirstlane((int)cmd[1])) err |= 4u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            if (lane == 0) {
                ctl->reward = out_reward[0]; ctl->lives = out_lives[0]; ctl->score = out_score[0];
                ctl->done_err = (out_done[0] ? 1u : 0u) | err;
                __hip_atomic_store(&ctl->ack_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            last = seq;
        }
        __syncthreads();                                         // cmd[] may be rewritten
    }
}

// ------------------------------------------------------------------ host side

struct GameOps;

// the outputs of a batch step (TBX_BUF_REWARD / DONE / LIVES / SCORE / PACKED); two sets exist once the pipelined mode is on
struct TbxStepOut {
    int32_t* reward = nullptr;
    uint8_t* done = nullptr;
    int32_t* lives = nullptr;
    int32_t* score = nullptr;
    uint64_t* packed = nullptr;
};

// Pipelined mode (TBX_OPT_PIPELINE, engine.hip): random-rollout steps and batch renders of engines whose rasteriser reads
// step-written records run on internal streams, over double-buffered records, step outputs and frames.
struct TbxPipe {
    hipStream_t lane[2] = {nullptr, nullptr};        // value 2: lane[0] is the step stream; value 3: step N and render N on lane[parity]
    hipEvent_t step_ev = nullptr;                    // behind the last pipelined step ...
    hipStream_t step_on = nullptr;                   // ... which ran on this stream ...
    hipStream_t step_user = nullptr;                 // ... and which this caller's stream has been made to wait for (compared, never used)
    hipEvent_t render_ev[2] = {nullptr, nullptr};    // behind the last render that READ records buffer p ...
    hipStream_t render_on[2] = {nullptr, nullptr};   // ... which ran on this stream
    bool render_pending[2] = {false, false};
    // reader fences on the caller's stream: everything the caller had queued there when the step (render) call that
    // superseded buffer p's content was made -- the next writer of buffer p waits for it
    hipEvent_t user_step_ev[2] = {nullptr, nullptr}, user_frame_ev[2] = {nullptr, nullptr};
    bool user_step_rec[2] = {false, false}, user_frame_rec[2] = {false, false};
    bool active = false;          // the last call through the handle was a pipelined step or render
    bool prepared = false;        // every resource above exists (set last by pipe_prepare)
    bool step_outstanding = false;
    int live_reader = -1;         // >= 0: the render behind render_ev[live_reader] read LIVE state (records were not valid): the next step waits for it
    int frame_par = -1;           // frame buffer the last overlapped render wrote (-1: none since the pipeline was entered)
    uint8_t* frame[2] = {nullptr, nullptr};
    size_t frame_bytes[2] = {0, 0};
    // Overlapped fused launches (TBX_OPT_FUSED_OVERLAP; engine.hip, fused_overlapped): consecutive tbx_render_step_synthetic
    // launches alternate between the two lanes, the two output sets and the two frame buffers, and launch N+1 is ordered behind
    // the STEP BLOCKS of launch N only -- a device counter they bump when their state, records and outputs are written, waited
    // for by a one-wave kernel in front of launch N+1 (tbx_ticket_wait_kernel) -- not behind its rasteriser blocks.
    bool fused = false;                              // the calls since the pipeline was entered are such launches (pipe_enter joins before the kind of call changes)
    // Rollout chunks (tbx_rollout_synthetic; engine.hip, rollout_chunked): k frames per call.  ONE step launch on the step lane
    // writes the k render records, the k step records and the state; the rasteriser launches (k, alternating between the two lanes, or
    // one over the chunk's k x N frames on lane 0) depend on that step launch alone -- and the step launch of chunk c+1 runs beside them, a whole chunk ahead of its own
    // rasterisers: no launch waits for a step that runs beside a rasteriser (what kept overlapped fused launches unstable).
    bool rollout = false;                            // the calls since the pipeline was entered are rollout chunks
    hipStream_t step_lane = nullptr;
    hipEvent_t chunk_step_ev[2] = {nullptr, nullptr};        // behind the step launch of the last chunk of parity q
    hipEvent_t chunk_raster_ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [q][lane]: behind that chunk's last rasteriser launch on the lane
    bool chunk_step_rec[2] = {false, false}, chunk_raster_rec[2][2] = {{false, false}, {false, false}};
    bool chunk_user_waits[2] = {false, false};       // the caller's stream has been made to wait for chunk q (lazy join)
    uint8_t* chunk_frames[2] = {nullptr, nullptr};   // [k][N][H][W][C] frames of the last chunk of parity q
    size_t chunk_frame_bytes[2] = {0, 0};
    uint64_t* chunk_packed[2] = {nullptr, nullptr};  // [k][N] step records of that chunk when no gather ring takes them
    size_t chunk_packed_bytes[2] = {0, 0};
    int chunk_cur = 0, chunk_k = 0, chunk_channels = 0;      // what TBX_BUF_ROLLOUT_* name: parity, frames and channels of the last chunk
    uint64_t* chunk_packed_base = nullptr;           // ... and where its step records lie (a ring of the gather, or chunk_packed)
    size_t chunk_packed_stride = 0;
    hipEvent_t launch_ev[2] = {nullptr, nullptr};    // completion event of the last fused launch on lane k (it rides on the launch)
    bool launch_rec[2] = {false, false};
    unsigned long long* arrive = nullptr;            // device [2]: step blocks of overlapped launches that have finished, ever; launches
                                                     // whose RELEASE block has started (the block `lead` blocks before the end of the grid)
    unsigned long long arrive_want = 0;              // host: what the launches issued so far add up to
    unsigned long long release_want = 0;
    bool prev_overlapped = false;                    // the call before this one was an overlapped launch (else stream order holds)
    // The caller's stream joins LAZILY in this mode: it is made to wait for the launch that wrote a result when the caller asks
    // for the result's address (tbx_device_buffer -- required after every call, the addresses alternate) or makes a call of any
    // other kind.  A loop that only rolls queues nothing on the caller's stream: with a record gather on the device, a wait and
    // a fence per call on that stream turned a gain of 4-25 % into a loss of 10-100 % (profiles/r06_experiments.txt).
    bool user_waits[2] = {false, false};             // the caller's stream has been made to wait for launch_ev[k]
    bool reader_seen = false;                        // an address was handed out since the last call: the next call fences the caller's stream
};

// what an overlapped fused launch gets from the engine: the counter its step blocks bump, the event that rides on the launch as
// its completion event; step_blocks comes back (how many arrivals the launch adds)
// measurement builds only (make DIAG=1; env TBX_OVERLAP_DIAG read in engine.hip; results are WRONG or racy with any bit set): parts of
// the overlapped launch switched off one at a time -- 1 the first build's step half: plain loads and stores behind a per-wave L2
// invalidate and write-back (+ 2: no write-back) instead of agent-scope loads and stores, 4 plain record load, 8 both launches on ONE lane (the machinery without any overlap), 16 no fence / wait on the caller's
// stream, 32 one frame buffer, 64 no wait kernel, 128 no completion event on the launch
#ifdef TBX_DIAG
#define OVL_DIAG(mask, bit) (((mask) & (bit)) != 0)
#else
#define OVL_DIAG(mask, bit) false
#endif

struct TbxOverlapLaunch {
    unsigned long long* arrive;     // [0] bumped by every step block when its stores are out, [1] by the release block when it starts
    hipEvent_t done;
    int lead;                       // TBX_OPT_FUSED_OVERLAP_LEAD (0: the engine's choice)
    int step_blocks;
    int diag;                       // OVL_DIAG mask (0 in product builds)
};

struct tbx_engine {
    int game = -1, n = 0, device = 0;
    mutable std::string err;
    hipStream_t stream = nullptr;   // engine-owned stream used by the host-pointer entry points
    // cross-stream ordering of everything queued through this handle (tbx_use_stream): the stream the last call used.  It may
    // be the caller's, which therefore has to outlive the next call on the handle (tbx_sync forgets it).
    hipStream_t last_stream = nullptr;
    bool has_last = false;
    hipEvent_t order_ev = nullptr;
    bool step_carries_order_ev = false;        // order_ev is the completion event of the last launch on last_stream (a batch step)
    int opt[TBX_OPT_COUNT] = {0, 0, 0, 0, 1, 1, 0, 0, 0, 0};
    bool gather_ring = false;                  // a K-step record ring is in force (TBX_OPT_GATHER_EVERY > 1 at tbx_gather_init): no pipelined mode
    int gather_ring_every = 0, gather_ring_width = 0;   // ... its K and its row width in records (tbx_rollout_synthetic)
    bool gather_wants_step_event = false;      // the next batch step is one a collective will wait for: its launch carries the ordering event
    TbxPipe pipe;
    // common device buffers (SoA over envs)
    uint64_t* sim_rng = nullptr;    // [2][N] simulator RNG
    int32_t* prev_score = nullptr;  // [N]
    TbxStepOut outs[2];             // [1] is allocated when the pipelined mode is switched on
    int out_par = 0;
    int32_t* reward = nullptr;      // [N]   == outs[out_par].* : the outputs of the most recently issued step
    uint8_t* done = nullptr;        // [N]
    int32_t* lives_out = nullptr;   // [N]
    int32_t* score_out = nullptr;   // [N]
    uint64_t* packed = nullptr;     // [N]
    int32_t* actions = nullptr;     // [N] staging for host actions
    uint8_t* mask = nullptr;        // [N] staging for new_game masks
    uint32_t* err_flag = nullptr;   // device word: bit0 = illegal action seen
    int32_t* scal = nullptr;        // [3][N] scratch of tbx_get_scalars
    int32_t* scal_host = nullptr;   // pinned mirror of it
    uint8_t* one_frame = nullptr;   // H*W*4 scratch of tbx_render_env
    // host-pointer step path: one device block [reward | lives | score | err | done] gathered by a kernel and ONE copy
    // into pinned host memory (five pageable copies cost ~100 us per call); actions go up through the pinned block too
    int32_t* io_dev = nullptr;      // 3N + 1 dwords + N bytes
    int32_t* io_host = nullptr;     // pinned mirror (+ N action dwords in front)
    bool host_pending = false;      // a tbx_step_begin whose outputs are on their way (tbx_step_end takes them)
    // "Any other call on the handle between _begin and _end ends the step first" (toybox_amd.h): which kind of step is pending
    // (0 none, 1 tbx_step_begin, 2 tbx_agent_step_begin), and -- once another entry point has ended it through
    // tbx_finish_pending -- the result its own "_end" call still has to report
    int pending_kind = 0;
    int ended_early_kind = 0;
    int ended_early_rc = 0;
    std::string ended_early_msg;
    tbx_step_host_out_t host_out{}; // where they go
    uint8_t* frame_own = nullptr;   // engine-owned frame buffer (lazy)
    size_t frame_own_bytes = 0;
    uint8_t* frame = nullptr;       // what TBX_BUF_FRAME reports: frame_own, or in pipelined mode the buffer the last render wrote
    size_t frame_bytes = 0;
    double* edit_args = nullptr;    // [N][n_args] per-env arguments of tbx_edit / tbx_reduce (host-pointer forms)
    size_t edit_args_bytes = 0;
    double* reduce_out = nullptr;   // [N][width] result staging of tbx_reduce
    size_t reduce_out_bytes = 0;
    void* staging = nullptr;        // device POD staging for get/set state
    size_t staging_bytes = 0;
    GameOps* ops = nullptr;
    struct AgentState* agent = nullptr;   // fused agent-side preprocessing (agent.hip), lazily created
    struct GatherState* gather = nullptr; // multi-GPU record gather over RCCL (gather.hip), created by tbx_gather_init
    // resident single-env step kernel (tbx_step1 on one-env engines)
    TbxServeCtl* serve_ctl = nullptr;     // host-coherent pinned mailbox (host address)
    TbxServeCtl* serve_ctl_dev = nullptr; // its device address
    uint8_t* serve_frame = nullptr;       // mapped pinned frame buffer the resident kernel rasterises into (H * W * 4 bytes, host address)
    uint8_t* serve_frame_dev = nullptr;   // its device address
    hipStream_t serve_stream = nullptr;
    bool serve_running = false;           // a server kernel has been launched and not yet been seen to exit
    uint32_t serve_seq = 0;

    int fail(int code, const std::string& msg) const
    {
        err = msg;
        return code;
    }
};

hipError_t tbx_serve_stop(tbx_engine* e);   // engine.hip
hipError_t tbx_finish_pending(tbx_engine* e);   // engine.hip: ends a step that is between "_begin" and "_end" (outputs delivered, result kept)
int tbx_agent_deliver(tbx_engine* e);       // agent.hip: the waiting half of tbx_agent_step_end

// Stream `s` waits for everything queued so far on the stream the previous call used.  That stream may be the caller's: the
// handle is kept until the next call or tbx_sync (toybox_amd.h: a stream named in a call must stay alive that long -- the
// runtime does not survive an event record on a destroyed stream, so a stale handle cannot be detected here).
// after_step_only: the caller (the gather) needs nothing but the last batch step; when that step's launch carried the ordering
// event as its completion event (TBX_LAUNCH_STEP below) no event has to be recorded behind it.
inline hipError_t tbx_wait_tail(tbx_engine* e, hipStream_t s, bool after_step_only = false)
{
    if (!e->has_last || e->last_stream == s) return hipSuccess;
    if (!e->order_ev) {
        hipError_t r = hipEventCreateWithFlags(&e->order_ev, hipEventDisableTiming);
        if (r != hipSuccess) return r;
    }
    if (!(after_step_only && e->step_carries_order_ev)) {
        hipError_t r = hipEventRecord(e->order_ev, e->last_stream);
        if (r != hipSuccess) return r;
    }
    return hipStreamWaitEvent(s, e->order_ev, 0);
}

// The launch of a whole-batch step kernel.  With a per-step gather initialised the ordering event rides on the launch as its
// completion event (hipExtLaunchKernelGGL's stopEvent) instead of being recorded behind it: measured on this runtime
// (scripts/ubench/evgap.hip), an event record between two kernels of a stream that another stream waits for delays the second
// kernel by 5.7 us, the completion-event form by 2.4 (two kernels with nothing between them: 1.0).
inline hipEvent_t tbx_step_order_event(tbx_engine* e)
{
    if (!e->gather || !e->gather_wants_step_event) return nullptr;    // (ring mode: only the step that completes the ring)
    if (!e->order_ev && hipEventCreateWithFlags(&e->order_ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e->order_ev;
}
#define TBX_LAUNCH_STEP(e, s, KERNEL, GRID, BLOCK, ...)                                                             \
    do {                                                                                                            \
        hipEvent_t tail_ev_ = tbx_step_order_event(e);                                                              \
        if (tail_ev_) {                                                                                             \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, s, nullptr, tail_ev_, 0, __VA_ARGS__);                    \
            (e)->step_carries_order_ev = true;                                                                      \
        } else                                                                                                      \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, s, __VA_ARGS__);                                             \
    } while (0)

// Every entry point that queues work names the stream it is about to use.  When that differs from the stream the previous
// entry point used (the "_device" forms run on the caller's stream -- including the NULL stream, which does not order itself
// against the engine's non-blocking stream -- the host-pointer forms on the engine's own), the new stream first waits for an
// event recorded on the old one, so calls on one handle take effect in program order whatever streams they name.  (Pipelined
// calls make the caller's stream wait for their internal work and leave it as `last_stream`, so this also joins the pipeline.)
inline hipError_t tbx_use_stream(tbx_engine* e, hipStream_t s)
{
    if (e->pending_kind) {                     // a step between "_begin" and "_end": this call ends it first
        hipError_t r = tbx_finish_pending(e);
        if (r != hipSuccess) return r;
    }
    if (e->serve_running) {                    // nothing else runs beside the resident step kernel
        hipError_t r = tbx_serve_stop(e);
        if (r != hipSuccess) return r;
    }
    hipError_t r = tbx_wait_tail(e, s);
    if (r != hipSuccess) return r;
    if (e->pipe.active) {
        // leaving the pipelined mode: a render that ran on another caller stream than the one the last pipelined call named
        // (mode 2, or out_dev given: render on U1, then a step naming U2) is not behind last_stream -- join it here
        TbxPipe& p = e->pipe;
        for (int k = 0; k < 2; k++)
            if (p.render_pending[k] && p.render_on[k] && p.render_on[k] != s) {
                r = hipStreamWaitEvent(s, p.render_ev[k], 0);
                if (r != hipSuccess) return r;
            }
        // ... and overlapped fused launches, which the caller's stream joins lazily (TbxPipe::user_waits): both lanes
        if (p.fused)
            for (int k = 0; k < 2; k++)
                if (p.launch_rec[k] && p.lane[k] != s) {
                    r = hipStreamWaitEvent(s, p.launch_ev[k], 0);
                    if (r != hipSuccess) return r;
                }
        // ... and rollout chunks: the step lane and the rasterisers of both parities
        if (p.rollout)
            for (int q = 0; q < 2; q++) {
                if (p.chunk_step_rec[q]) {
                    r = hipStreamWaitEvent(s, p.chunk_step_ev[q], 0);
                    if (r != hipSuccess) return r;
                }
                for (int l = 0; l < 2; l++)
                    if (p.chunk_raster_rec[q][l]) {
                        r = hipStreamWaitEvent(s, p.chunk_raster_ev[q][l], 0);
                        if (r != hipSuccess) return r;
                    }
            }
    }
    e->step_carries_order_ev = false;          // whatever this call queues moves the tail
    e->pipe.active = false;
    e->last_stream = s;
    e->has_last = true;
    return hipSuccess;
}

// arguments of a batched intervention (tbx_edit / tbx_reduce): the same row for every env, or one row per env in HBM
struct TbxEditArgs {
    double v[TBX_EDIT_MAX_ARGS];
    int n;
    const double* per_env;        // device [N][n] or nullptr
    __device__ __forceinline__ double get(int env, int i) const { return i >= n ? 0.0 : per_env ? per_env[(size_t)env * n + i] : v[i]; }
    // an unsigned 32-bit argument (masks, seeds, counters: integers below 2^32 are exact in binary64)
    __device__ __forceinline__ uint32_t getu(int env, int i) const
    {
        const double x = get(env, i);
        return x >= 4294967295.0 ? 0xFFFFFFFFu : x > 0.0 ? (uint32_t)x : 0u;
    }
    __device__ __forceinline__ int geti(int env, int i) const
    {
        double x = get(env, i);
        if (!(x > -2.0e9)) x = -2.0e9;
        if (x > 2.0e9) x = 2.0e9;
        return (int)x;
    }
};

// per-game operations; all launches are asynchronous on `s`
struct GameOps {
    virtual ~GameOps() {}
    virtual int init(tbx_engine* e, const void* cfg, size_t cfg_size) = 0;
    virtual void destroy(tbx_engine* e) = 0;
    virtual int height() const = 0;
    virtual int width() const = 0;
    virtual size_t state_size() const = 0;
    virtual size_t config_size() const = 0;
    virtual int get_config(tbx_engine* e, void* pod) = 0;
    virtual int set_config(tbx_engine* e, const void* pod) = 0;
    virtual int new_game(tbx_engine* e, const uint8_t* mask_dev, hipStream_t s) = 0;
    virtual int step(tbx_engine* e, const ActionSource& src, uint32_t flags, hipStream_t s) = 0;
    virtual int render(tbx_engine* e, uint8_t* out_dev, int channels, int first_env, int n_envs, hipStream_t s) = 0;
    // pack envs [env, env+count) -> e->staging (device, count records), unpack host records -> those envs
    virtual int pack_state(tbx_engine* e, int env, int count, hipStream_t s) = 0;
    virtual int unpack_state(tbx_engine* e, int env, int count, const void* pod_host, hipStream_t s) = 0;
    virtual int scalars(tbx_engine* e, int32_t* score_dev, int32_t* lives_dev, int32_t* level_dev, hipStream_t s) = 0;
    // launch the resident single-env step kernel for env 0 on `s` (tbx_serve_loop); optional
    virtual int serve(tbx_engine*, TbxServeCtl* /*ctl_dev*/, hipStream_t) { return TBX_E_UNSUPPORTED; }
    virtual bool serve_paints() const { return false; }        // the resident kernel can rasterise env 0 on request
    // ---- agent layer (agent.hip).  The two-frame buffer of MaxAndSkipEnv lives with the game as two snapshot slots.
    virtual int agent_prepare(tbx_engine*) { return TBX_OK; }  // allocate the slots
    virtual bool multi_frame_step() const { return false; }   // step() honours ActionSource::frames and writes the slots itself
    // single-frame launches: copy the live state of the envs with exec_flag set into slot 0 (A) / 1 (B), set their valid bit
    virtual int agent_snapshot(tbx_engine*, int /*slot*/, const uint8_t* /*exec_flag*/, uint8_t* /*buf_valid*/, hipStream_t) { return TBX_E_UNSUPPORTED; }
    // fused observation kernels (no full-resolution frame leaves the chip)
    virtual bool agent_fused() const { return false; }
    virtual int agent_warp(tbx_engine*, const struct AgentWarpArgs&, hipStream_t) { return TBX_E_UNSUPPORTED; }
    // pipelined mode: the rasteriser reads records the step kernel writes, and there are two buffers of them -- a batch step
    // may then run while the previous frame is still being rasterised.  pipeline_ok(): this engine can (canonical state
    // layout, thread-per-env step); records_parity(): the buffer a render launched now reads; step_ahead(): one frame of
    // every env on stream s, records into the OTHER buffer, which becomes the current one.  The step outputs go wherever
    // tbx_engine::reward / done / ... point at the time of the launch (rebind_outputs() after the engine moved them).
    virtual bool pipeline_ok() const { return false; }
    // TBX_OPT_PIPELINE = 1, the engine's choice: the mode (0, 2 or 3) for a batch of n envs, with or without a per-step gather.
    // Large batches: stream order for every game since the rasterisers stagger their first waves (raster.hpp) -- what values 2
    // and 3 bought Breakout and SpaceInvaders there was a rasteriser launch that did not start against an idle memory system.
    // Small batches without a gather: overlapped launches (value 3) where they measure faster (BrkOps, SiOps); a gather adds
    // cross-queue dependencies that cost more than the overlap gains.
    virtual int pipeline_auto(int /*n*/, bool /*gather*/) const { return 0; }
    virtual int records_parity() const { return 0; }
    virtual bool records_valid() const { return true; }        // false: the next render starts from live state (prep kernel / state-reading rasteriser)
    virtual int step_ahead(tbx_engine*, const ActionSource&, uint32_t, hipStream_t) { return TBX_E_UNSUPPORTED; }
    virtual void rebind_outputs(tbx_engine*) {}
    // tbx_render_step_synthetic: the rasteriser of the current frame and the batch step to the next one as ONE launch on s
    // (engines whose rasteriser reads step-written records); render_step_fused() false: the engine runs render(), then step()
    virtual bool render_step_fused(int /*channels*/) const { return false; }
    // ov != nullptr: an overlapped launch (TbxPipe::fused) -- its step blocks bump ov->arrive when their stores are visible
    // device-wide, ov->done rides on the launch as its completion event, ov->step_blocks is filled in
    virtual int render_step(tbx_engine*, uint8_t* /*out_dev*/, int /*channels*/, const ActionSource&, uint32_t /*flags*/, hipStream_t,
                            TbxOverlapLaunch* /*ov*/ = nullptr) { return TBX_E_UNSUPPORTED; }
    // TBX_OPT_FUSED_OVERLAP = 0, the engine's choice: overlap consecutive fused launches for a batch of n envs?
    // (gather_kind: 0 no record gather, 1 one collective per step, 2 a K-step ring)
    virtual bool fused_overlap_auto(int /*n*/, int /*gather_kind*/) const { return false; }
    // tbx_rollout_synthetic as chunks (TbxPipe::rollout).  rollout_ok(): this engine can right now (canonical state layout, RGB /
    // RGBA); rollout_auto(): the engine's choice for n envs; rollout_step(): frames t .. t + k - 1 of every env in ONE launch on s --
    // render record j (the state BEFORE frame j) into the chunk's record buffer of parity q, step record j into packed + j * stride,
    // the last frame's outputs into tbx_engine::reward / ...; rollout_render(): the rasteriser of record j of parity q into out.
    virtual bool rollout_ok(int /*channels*/) const { return false; }
    virtual bool rollout_auto(int /*n*/, int /*gather_kind*/) const { return false; }
    virtual int rollout_step(tbx_engine*, const ActionSource&, uint32_t /*flags*/, int /*k*/, int /*q*/, uint64_t* /*packed*/, size_t /*stride*/, hipStream_t) { return TBX_E_UNSUPPORTED; }
    virtual int rollout_render(tbx_engine*, uint8_t* /*out*/, int /*channels*/, int /*q*/, int /*j*/, hipStream_t) { return TBX_E_UNSUPPORTED; }
    // rollout_render_span(): records j0 .. j0 + count - 1 of parity q in ONE rasteriser launch (they lie one behind the other, and so do
    // their frames: count x n "envs" to the rasteriser; behind_rasteriser: another rasteriser launch is still running in front of it on
    // s, so its first waves start against draining stores -- raster.hpp); rollout_span_auto(): the engine's choice between one such
    // launch per chunk on one internal stream and a launch per frame on two
    virtual bool rollout_span_ok() const { return false; }
    virtual bool rollout_span_auto(int /*n*/, int /*gather_kind*/) const { return false; }
    virtual int rollout_render_span(tbx_engine*, uint8_t* /*out*/, int /*channels*/, int /*q*/, int /*j0*/, int /*count*/, bool /*behind_rasteriser*/, hipStream_t) { return TBX_E_UNSUPPORTED; }
    // batched interventions (include/toybox_amd.h, tbx_edit / tbx_reduce): one kernel over the selected envs
    virtual int edit(tbx_engine* e, int /*op*/, const TbxEditArgs&, const uint8_t* /*mask_dev*/, hipStream_t) { return e->fail(TBX_E_INVALID, "this game has no such edit"); }
    virtual int reduce(tbx_engine* e, int /*query*/, const TbxEditArgs&, double* /*out_dev*/, int /*width*/, hipStream_t) { return e->fail(TBX_E_INVALID, "this game has no such query"); }
    // an engine option changed (tbx_set_option): pick it up
    virtual void options_changed(tbx_engine*) {}
    // generic path: full-resolution gray frames of slot A (source 1), slot B (2) or the live state (0); envs whose
    // pick_live byte is non-zero are painted from the live state instead
    virtual int render_from(tbx_engine*, int /*source*/, const uint8_t* /*pick_live*/, uint8_t* /*out_dev*/, int /*channels*/, hipStream_t) { return TBX_E_UNSUPPORTED; }
    // the reset path of the wrapper stack, run in-kernel for the envs flagged in AgentResetArgs::kind
    virtual bool agent_reset_supported() const { return false; }
    virtual int agent_reset_envs(tbx_engine*, const struct AgentResetArgs&, hipStream_t) { return TBX_E_UNSUPPORTED; }
};

void tbx_agent_free(tbx_engine* e);
void tbx_gather_free(tbx_engine* e);
hipError_t tbx_gather_before_step(tbx_engine* e, hipStream_t s);   // a step must not overwrite records a queued gather still reads
void tbx_set_out_parity(tbx_engine* e, int p);                    // engine.hip: which TbxStepOut set the next step writes
void tbx_set_create_error(const std::string& msg);                 // text behind tbx_last_error(NULL)
int tbx_gather_buffer(tbx_engine* e, void** out_ptr, size_t* out_bytes);
// gather.hip, for tbx_rollout_synthetic over a K-step ring: the ring the next K steps fill (stream s is made to wait for the collective
// that last read it) and, once ONE launch has filled it, the collective behind that launch's event
int tbx_gather_ring_open(tbx_engine* e, hipStream_t s, int k, uint64_t** base, size_t* stride);
int tbx_gather_ring_filled(tbx_engine* e, hipEvent_t filled_ev);
GameOps* tbx_make_breakout_ops();
GameOps* tbx_make_si_ops();
GameOps* tbx_make_amidar_ops();
GameOps* tbx_make_gridworld_ops();
