// agent_device.hpp -- device helpers shared by the generic warp kernel (agent.hip) and the per-game fused variants.
#pragma once

#include "tbx_common.hpp"
#include "raster.hpp"

struct AgentTaps {          // area-resize taps of one output row / column
    int32_t start, n;
    uint8_t w[8];           // overlap lengths in units of 1/out of a source pixel; their sum is the source extent
};


// what a game's observation kernel needs from the agent layer
struct AgentWarpArgs {
    const uint8_t* zero;       // [N] the env reported done in this agent step: VecFrameStack zeroes its older slots
    const uint8_t* mode;       // [N] 1: the observation is the raw frame of the live state (what a reset without FireResetEnv
                               //     returns after a new game); 0: the max over the two-frame buffer (MaxAndSkipEnv.step)
    const uint8_t* valid;      // [N] bit 0 / 1: buffer slot A / B has been written since construction (else a zero frame)
    const AgentTaps* tx;       // [out_w] column taps
    uint8_t* obs;              // [N][out_h][out_w][stack]
    uint8_t* plane;            // [N][out_h][out_w] the newest plane alone (tbx_agent_config_t::new_plane), or nullptr
    uint8_t *older0, *older1, *older2;   // ring mode (new_plane = 2; obs == nullptr, plane = the ring slot that takes the newest plane): the ring's
                               // stack - 1 other slots, [N][out_h][out_w] each -- written only for an env whose stack starts afresh
    int H, W, oh, ow, stack;
    int first, end;            // the envs this launch makes observations for: [first, end) (the whole batch, or one chunk of it when the
                               // host-delivery path overlaps the copy of a chunk's planes with the next chunk's kernel)
    int reset_mode;            // venv.reset(): every stack starts from zeros
    int fill_repeat;           // tbx_agent_config_t::stack_fill: a fresh stack holds the new frame in every slot, not zeros
    uint64_t magic;            // floor(2^42 / (H*W)) + 1
#ifdef TBX_DIAG
    int diag;                  // measurement builds only (make DIAG=1, scripts/agent_diag.sh; observations are WRONG with any bit set):
                               // 1 no stack commit, 2 no scanline loop, 4 every scanline skipped, 8 never a second painter, 16 no fast rows
#endif
};
#ifdef TBX_DIAG
#define AGENT_DIAG(a, bit) (((a).diag & (bit)) != 0)
#else
#define AGENT_DIAG(a, bit) false
#endif

// which frames make up this env's observation (wave-uniform)
struct ObsSel {
    int zero;     // 0: roll the stack; 1: zero the older slots (VecFrameStack); 2: fill them with the new frame (FrameStack.reset)
    bool none;    // both buffer slots still hold np.zeros: the observation is black
    bool two;     // max(slot A, slot B)
    int single;   // !two: the one source -- 0 live state, 1 slot A, 2 slot B
};

__device__ __forceinline__ ObsSel agent_obs_sel(const AgentWarpArgs& a, int env)
{
    ObsSel s;
    const uint32_t mode = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.mode[env]);
    const uint32_t valid = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.valid[env]);
    s.zero = (a.reset_mode || __builtin_amdgcn_readfirstlane((int)a.zero[env]) != 0) ? (a.fill_repeat ? 2 : 1) : 0;
    const bool raw = (mode & 1u) != 0;
    s.none = !raw && (valid & 3u) == 0u;
    s.two = !raw && (valid & 3u) == 3u;
    s.single = raw ? 0 : (valid & 2u) ? 2 : 1;
    return s;
}

// per-byte max of two packed dwords: even and odd bytes as two packed-u16 maxima (v_pk_max_u16)
__device__ __forceinline__ uint32_t bytemax4(uint32_t a, uint32_t b)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const uint32_t m = 0x00FF00FFu;
    const uint32_t ae = a & m, be = b & m, ao = (a >> 8) & m, bo = (b >> 8) & m;
    const u16x2 e = __builtin_elementwise_max(__builtin_bit_cast(u16x2, ae), __builtin_bit_cast(u16x2, be));
    const u16x2 o = __builtin_elementwise_max(__builtin_bit_cast(u16x2, ao), __builtin_bit_cast(u16x2, bo));
    return __builtin_bit_cast(uint32_t, e) | (__builtin_bit_cast(uint32_t, o) << 8);
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


// what the stack word of a pixel is taken to have held before the new value is rolled in (S == 4); fresh as ObsSel::zero
__device__ __forceinline__ uint32_t stack_old_word(uint32_t stored, uint32_t val, int fresh)
{
    return fresh == 0 ? stored : fresh == 2 ? val * 0x01010101u : 0u;
}
// rolls `val` into the frame stack at px (S bytes per pixel, newest last); fresh: 1 = older slots become 0, 2 = become val
template <int S>
__device__ __forceinline__ void stack_push(uint8_t* px, uint32_t val, int fresh)
{
    if (S == 4) {
        const uint32_t old = stack_old_word(fresh ? 0u : *reinterpret_cast<uint32_t*>(px), val, fresh);
        *reinterpret_cast<uint32_t*>(px) = (old >> 8) | (val << 24);
    } else {
#pragma unroll
        for (int c = 0; c + 1 < S; c++) px[c] = fresh == 2 ? (uint8_t)val : fresh ? (uint8_t)0 : px[c + 1];
        px[S - 1] = (uint8_t)val;
    }
}

// The observation of one env is first collected in LDS (one byte per output pixel) and committed to the frame stack in
// one sweep, so that the read-modify-write of the stack runs with many loads in flight instead of one dependent
// load -> store per output row.  out_h * out_w <= AGENT_MAX_OUT_PX.
constexpr int AGENT_MAX_OUT_PX = 84 * 84;

// plane_env: where the env's newest plane goes as well (dense, the host-delivery form), or nullptr.
//
// Depth 4 and a whole number of pixel quads (84 x 84 is one): a lane takes FOUR consecutive pixels per turn -- one 32-bit LDS
// read of the new bytes, one 16-byte load and one 16-byte store of their four stack words (1 KiB per instruction, like the
// frame stores), and the roll of a word, (old >> 8) | (new << 24), is ONE v_perm_b32 that picks bytes 1, 2, 3 of the old
// word and byte k of the LDS dword.  Round 5: the byte-at-a-time form before it cost 1 021 VALU and 418 memory instructions
// per wave of si_agent_warp_kernel<4> (scripts/asm_attrib.py: 9 % of the kernel's VALU, a third of its memory instructions).
template <int S>
__device__ __forceinline__ void stack_commit(const uint8_t* vals, uint8_t* o, int n_px, int lane, int fresh, uint8_t* plane_env)
{
    __builtin_amdgcn_wave_barrier();
    if (S == 4 && (n_px & 3) == 0) {
        const int nq = n_px >> 2;
        const uint32_t* v4 = reinterpret_cast<const uint32_t*>(vals);
        uint4* oq = reinterpret_cast<uint4*>(o);                         // n_px * 4 bytes per env: 16-byte aligned
        uint32_t* p4 = reinterpret_cast<uint32_t*>(plane_env);
        // byte selectors of v_perm_b32 over {new dword (bytes 4-7), old word (bytes 0-3)}; 0x0C selects a zero byte
        const uint32_t keep = fresh == 0 ? 0x00030201u : fresh == 2 ? 0u : 0x000C0C0Cu;
        const uint32_t rep = fresh == 2 ? 0x01010101u : 0x01000000u;    // where the new byte's selector goes
        const uint32_t s0 = keep + 4u * rep, s1 = keep + 5u * rep, s2 = keep + 6u * rep, s3 = keep + 7u * rep;
        constexpr int B = 4;                                             // quads in flight per lane
        int i = lane;
        for (; i + (B - 1) * 64 < nq; i += B * 64) {
            uint4 old[B];
            uint32_t w[B];
#pragma unroll
            for (int k = 0; k < B; k++) {
                w[k] = v4[i + 64 * k];
                if (!fresh) old[k] = oq[i + 64 * k];
                else old[k] = make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int k = 0; k < B; k++) {
                oq[i + 64 * k] = make_uint4(__builtin_amdgcn_perm(w[k], old[k].x, s0), __builtin_amdgcn_perm(w[k], old[k].y, s1),
                                            __builtin_amdgcn_perm(w[k], old[k].z, s2), __builtin_amdgcn_perm(w[k], old[k].w, s3));
                if (plane_env) p4[i + 64 * k] = w[k];
            }
        }
        for (; i < nq; i += 64) {
            const uint32_t w = v4[i];
            const uint4 old = fresh ? make_uint4(0u, 0u, 0u, 0u) : oq[i];
            oq[i] = make_uint4(__builtin_amdgcn_perm(w, old.x, s0), __builtin_amdgcn_perm(w, old.y, s1), __builtin_amdgcn_perm(w, old.z, s2),
                               __builtin_amdgcn_perm(w, old.w, s3));
            if (plane_env) p4[i] = w;
        }
        return;
    }
    if (plane_env) {                                   // (wave-uniform)
        if ((n_px & 3) == 0) {
            const uint32_t* v4 = reinterpret_cast<const uint32_t*>(vals);
            uint32_t* p4 = reinterpret_cast<uint32_t*>(plane_env);
            for (int i = lane; i < (n_px >> 2); i += 64) p4[i] = v4[i];
        } else {
            for (int i = lane; i < n_px; i += 64) plane_env[i] = vals[i];
        }
    }
    if (S == 4) {
        uint32_t* o4 = reinterpret_cast<uint32_t*>(o);
        if (fresh) {                                   // (wave-uniform) nothing of the old stack survives: stores only
            const uint32_t spread = fresh == 2 ? 0x01010101u : 0x01000000u;      // FrameStack.reset: every slot / VecFrameStack: the last
            for (int i = lane; i < n_px; i += 64) o4[i] = (uint32_t)vals[i] * spread;
            return;
        }
        for (int i = lane; i < n_px; i += 64) o4[i] = (o4[i] >> 8) | ((uint32_t)vals[i] << 24);
    } else {
        for (int i = lane; i < n_px; i += 64) stack_push<S>(o + (size_t)i * S, vals[i], fresh);
    }
}

// Ring mode (tbx_agent_config_t::new_plane = 2): no rolled stack on the device.  The newest plane goes into the ring slot a.plane
// points at -- 7 KB per env and agent step instead of the roll's 21 KB read + 28 KB written -- and only an env whose stack starts
// afresh (VecFrameStack zeroes a finished env's older frames, FrameStack.reset repeats the observation) touches the other slots.
// (named fields, not an array: an array in the by-value kernel argument is copied to scratch when it is indexed)
__device__ __forceinline__ uint8_t* ring_older(const AgentWarpArgs& a, int k) { return k == 0 ? a.older0 : k == 1 ? a.older1 : a.older2; }

__device__ __forceinline__ void ring_commit(const uint8_t* vals, const AgentWarpArgs& a, int env, int n_px, int lane, int fresh)
{
    __builtin_amdgcn_wave_barrier();
    const size_t off = (size_t)env * n_px;
    const int others = a.stack - 1;                        // (wave-uniform)
    if ((n_px & 15) == 0) {                                // 84 x 84 = 441 x 16: 1 KiB per store instruction
        const uint4* v = reinterpret_cast<const uint4*>(vals);
        uint4* p = reinterpret_cast<uint4*>(a.plane + off);
        const int nq = n_px >> 4;
        for (int i = lane; i < nq; i += 64) p[i] = v[i];
        if (fresh) {                                       // (wave-uniform)
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            for (int k = 0; k < others; k++) {
                uint4* q = reinterpret_cast<uint4*>(ring_older(a, k) + off);
                if (fresh == 2) for (int i = lane; i < nq; i += 64) q[i] = v[i];
                else for (int i = lane; i < nq; i += 64) q[i] = z;
            }
        }
        return;
    }
    for (int i = lane; i < n_px; i += 64) a.plane[off + i] = vals[i];
    if (fresh) {
        for (int k = 0; k < others; k++) {
            uint8_t* q = ring_older(a, k) + off;
            for (int i = lane; i < n_px; i += 64) q[i] = fresh == 2 ? vals[i] : (uint8_t)0;
        }
    }
}

// the finished observation plane of one env (LDS) -> the device's frame stack of depth S (and the plane alone), or -- the S == 0
// instantiation of the observation kernels, one per game whatever the depth -- the plane ring.  (A run-time choice between the
// two inside one kernel cost the depth-4 stack kernels of GridWorld and Amidar 2.8 % and 1.2 %: register allocation.)
template <int S>
__device__ __forceinline__ void observation_commit(const uint8_t* vals, const AgentWarpArgs& a, int env, int lane, int fresh)
{
    const int n_px = a.oh * a.ow;
    if constexpr (S == 0) ring_commit(vals, a, env, n_px, lane, fresh);
    else stack_commit<S>(vals, a.obs + (size_t)env * n_px * S, n_px, lane, fresh, a.plane ? a.plane + (size_t)env * n_px : nullptr);
}

// ------------------------------------------------------------------ reset-time wrappers + episode monitor (8f rank 2)

// what a game's agent-reset kernel needs (see tbx_agent_config_t)
struct AgentResetArgs {
    const uint8_t* kind;        // [N] 0 = nothing to do, else the env reported done and DummyVecEnv calls reset()
    // the flagged envs as a compact list (built by the monitor kernel) so that heavy wave-per-env reset kernels launch a
    // small persistent grid instead of N waves that exit at once; nullptr: every env is flagged (tbx_agent_reset)
    const int32_t* list;
    const int32_t* count;
    int skip, episodic_life, fire_reset, noop_max;
    uint64_t noop_seed, env_offset;
    const int32_t* noop_override;            // [N] or nullptr: NoopResetEnv.override_num_noops (> 0 overrides)
    uint32_t fire_buttons, third_buttons;   // buttons of action #1 and action #2 of the game's action set
    int32_t *ep_ret, *ep_len, *ep_index, *prev_lives;   // [N] Monitor.rewards (sum, count), episode counter, EpisodicLifeEnv.lives
    uint8_t *was_real_done, *needs_reset;   // [N] EpisodicLifeEnv.was_real_done, Monitor.needs_reset
    uint8_t* ep_done;           // [N] episode records of this agent step
    float* ep_ret_out;
    int32_t* ep_len_out;
    uint8_t *mode, *buf_valid;  // [N] see AgentWarpArgs
    uint32_t* err_flag;         // bit 1: an env was stepped although Monitor.needs_reset (TBX_E_NEEDS_RESET)
};

struct AgentMonitor {
    int32_t ep_ret, ep_len, ep_index, prev_lives;
    bool was_real_done, needs_reset;
    bool emitted, stale;
    int32_t out_ret, out_len;
};

__device__ __forceinline__ AgentMonitor agent_monitor_load(const AgentResetArgs& r, int env)
{
    AgentMonitor m;
    m.ep_ret = r.ep_ret[env]; m.ep_len = r.ep_len[env]; m.ep_index = r.ep_index[env]; m.prev_lives = r.prev_lives[env];
    m.was_real_done = r.was_real_done[env] != 0; m.needs_reset = r.needs_reset[env] != 0;
    m.emitted = false; m.stale = false; m.out_ret = 0; m.out_len = 0;
    return m;
}

// (one lane / thread per env calls this)
__device__ __forceinline__ void agent_monitor_store(const AgentResetArgs& r, int env, const AgentMonitor& m, uint32_t valid, bool obs_raw)
{
    r.ep_ret[env] = m.ep_ret; r.ep_len[env] = m.ep_len; r.ep_index[env] = m.ep_index; r.prev_lives[env] = m.prev_lives;
    r.was_real_done[env] = m.was_real_done ? 1 : 0; r.needs_reset[env] = m.needs_reset ? 1 : 0;
    if (m.emitted) { r.ep_done[env] = 1; r.ep_ret_out[env] = (float)m.out_ret; r.ep_len_out[env] = m.out_len; }
    if (m.stale) atomicOr(r.err_flag, 2u);
    r.buf_valid[env] = (uint8_t)valid;
    r.mode[env] = obs_raw ? 1 : 0;
}

// reset() of one env's wrapper stack, class by class (baselines/baselines/common/atari_wrappers.py, bench/monitor.py):
//   FireResetEnv.reset :144-152 -> EpisodicLifeEnv.reset / .step :166-191 -> Monitor.reset / .step (monitor.py:36-76)
//   -> MaxAndSkipEnv.step :201-216 / .reset :218-219 -> NoopResetEnv.reset :117-132 -> ToyboxBaseEnv.step / .reset
// run in-kernel for one env.  Env provides step(buttons), new_game(), lives(), score(), snapshot(slot); every call is
// wave-uniform for wave-per-env games.  `prev` is ToyboxBaseEnv.score, `valid` the written slots of the frame buffer,
// `obs_raw` what the last wrapper call returned: a raw frame of the live state (true) or the buffer max (false).
template <class Env>
struct AgentResetProc {
    Env& env;
    const AgentResetArgs& r;
    AgentMonitor& m;
    uint64_t env_global;
    int32_t prev;
    uint32_t valid;
    int32_t noop_override;
    bool obs_raw;

    __device__ __forceinline__ bool base_step(uint32_t buttons, int& reward)   // ToyboxBaseEnv.step
    {
        env.step(buttons);
        const int sc = env.score();
        reward = sc - prev > 0 ? sc - prev : 0;
        prev = sc;
        return env.lives() <= 0;
    }
    __device__ __forceinline__ void base_reset()                               // ToyboxBaseEnv.reset
    {
        env.new_game();
        prev = env.score();
    }
    __device__ __forceinline__ void noop_reset()                               // NoopResetEnv.reset
    {
        base_reset();
        obs_raw = true;
        int k = 0;
        if (noop_override > 0) k = noop_override;
        else if (r.noop_max > 0)
            k = 1 + (int)(tbx_splitmix64(r.noop_seed ^ (env_global << 32) ^ (uint64_t)(uint32_t)m.ep_index) % (uint64_t)r.noop_max);
        for (int j = 0; j < k; j++) {
            int rew;
            if (base_step(0u, rew)) base_reset();
        }
    }
    __device__ __forceinline__ bool skip_step(uint32_t buttons, int& total)    // MaxAndSkipEnv.step
    {
        bool done = false;
        total = 0;
        for (int i = 0; i < r.skip; i++) {
            int rew;
            done = base_step(buttons, rew);
            if (i == r.skip - 2) { env.snapshot(0); valid |= 1u; }
            if (i == r.skip - 1) { env.snapshot(1); valid |= 2u; }
            total += rew;
            if (done) break;
        }
        obs_raw = false;
        return done;
    }
    __device__ __forceinline__ bool monitor_step(uint32_t buttons)            // Monitor.step
    {
        const bool stale = m.needs_reset;
        int total;
        const bool done = skip_step(buttons, total);
        if (stale) { m.stale = true; return done; }                            // Monitor raises here: reported, and carried on
        m.ep_ret += total; m.ep_len += 1;
        if (done) { m.needs_reset = true; m.emitted = true; m.out_ret = m.ep_ret; m.out_len = m.ep_len; }
        return done;
    }
    __device__ __forceinline__ void monitor_reset()                            // Monitor.reset (allow_early_resets)
    {
        m.ep_ret = 0; m.ep_len = 0; m.needs_reset = false; m.ep_index += 1;
        noop_reset();
    }
    __device__ __forceinline__ bool episodic_step(uint32_t buttons)           // EpisodicLifeEnv.step
    {
        bool d = monitor_step(buttons);
        if (!r.episodic_life) return d;
        m.was_real_done = d;
        const int l = env.lives();
        if (l < m.prev_lives && l > 0) d = true;
        m.prev_lives = l;
        return d;
    }
    __device__ __forceinline__ void episodic_reset()                           // EpisodicLifeEnv.reset
    {
        if (!r.episodic_life) { monitor_reset(); return; }
        if (m.was_real_done) monitor_reset();
        else monitor_step(0u);                   // no-op step to advance from the lost-life state; its `done` is ignored
        m.prev_lives = env.lives();
    }
    __device__ __forceinline__ void run()                                      // FireResetEnv.reset (or the reset below it)
    {
        episodic_reset();
        if (r.fire_reset) {
            if (episodic_step(r.fire_buttons)) episodic_reset();
            const bool d = episodic_step(r.third_buttons);
            if (d) episodic_reset();
            obs_raw = false;                     // the observation is the one step(2) returned, whatever followed
        }
    }
};


// ------------------------------------------------------------------ fused observation from two painters
//
// max(frame A, frame B) -> area warp -> frame stack for games whose rasteriser is a "painter" (a struct that is set up once
// per frame from the SoA state and then composes any scanline as packed gray dwords), without the two full-resolution
// gray frames ever reaching HBM.  One wave per env holds both painters: A = a snapshot of the state after frame skip-2,
// B = the live state.  A painter sorts what it draws into NCLS classes (enemies, shields, lasers, HUD ...) and records
// each class's scanlines as a 256-bit mask in LDS; the game supplies diff_classes(A, B), the classes whose entities are
// not identical in the two states.  A scanline is then
//   * skipped when it is blank in B and holds no differing class (its horizontal sums are a per-column constant),
//   * painted from B alone when no differing class touches it (frame A shows the same pixels there),
//   * painted from both and maxed otherwise
// -- consecutive frames differ in a few moving objects, so most scanlines take the first two forms.
// (A two-wave variant, one painter per wave with a block barrier per chunk of scanlines, measured 18 % slower: fewer envs
// in flight and barrier stalls outweighed the lower register count.)
//
// Painter P: static W, H, NG (4-pixel groups per lane), NCLS, NLDS (>= NCLS: mask slots of LDS it uses); type Dev; setup(dev, env, lane, cls) with cls = this wave's
// [NCLS][8] dwords of LDS (the painter fills them and keeps their union in busy[4]); row_dwords(y, v[NG]); blank_dword()
// (the packed value of a scanline outside busy); rep[4]: wave-uniform mask of the scanlines that paint exactly as the one above them;
// static diff_classes(const P& a, const P& b) -> wave-uniform bit mask.
template <class P>
struct AgentFusedLds {
    static constexpr int ROWB = ((P::W + 32 + 15) / 16) * 16;
    uint8_t row[ROWB] __attribute__((aligned(16)));
    uint8_t vals[AGENT_MAX_OUT_PX] __attribute__((aligned(16)));   // the new observation plane, committed to the stack at the end
    uint32_t cls[2][P::NLDS][8];   // per painter: NCLS class masks (+ painter-private scratch masks)
    uint32_t masks[2][8];       // need / need_a
};

template <int S, class P>
__device__ __forceinline__ void agent_fused_wave(P& pa, P& pb, const typename P::Dev& dLive, const typename P::Dev& dA, const typename P::Dev& dB,
                                                 const AgentWarpArgs& a, int env, int lane, AgentFusedLds<P>& L)
{
    constexpr int W = P::W, H = P::H, NG = P::NG, NCLS = P::NCLS;
    static_assert(W % 4 == 0 && W / 4 <= 64 * NG && H <= 256, "painter geometry");
    const ObsSel sel = agent_obs_sel(a, env);
    if (sel.none) {                                                    // max over two zero frames
        for (int i = lane; i < a.oh * a.ow; i += 64) L.vals[i] = 0;
        observation_commit<S>(L.vals, a, env, lane, sel.zero);
        return;
    }
    const bool two = sel.two && !AGENT_DIAG(a, 8);
    uint8_t* row = L.row;
    // painter B is the one that is always set up: slot B, or the single source of a one-frame observation
    pb.setup(sel.single == 0 ? dLive : sel.single == 1 ? dA : dB, env, lane, &L.cls[1][0][0]);
    uint64_t need[4], need_a[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
    for (int k = 0; k < 4; k++) need[k] = pb.busy[k];
    if (two) {
        pa.setup(dA, env, lane, &L.cls[0][0][0]);
        const uint32_t diff = P::diff_classes(pa, pb);
        if (lane < 8) {
            uint32_t na = 0u;
            for (int c = 0; c < NCLS; c++)
                if ((diff >> c) & 1u) na |= L.cls[0][c][lane] | L.cls[1][c][lane];
            L.masks[1][lane] = na;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t lo = __builtin_amdgcn_readfirstlane(L.masks[1][2 * k]), hi = __builtin_amdgcn_readfirstlane(L.masks[1][2 * k + 1]);
            need_a[k] = (uint64_t)lo | ((uint64_t)hi << 32);
            need[k] |= need_a[k];
        }
    }
    const uint32_t half = (uint32_t)(H * W) / 2u;
    const ColTaps c0 = load_col(a.tx, lane, a.ow), c1 = load_col(a.tx, lane + 64, a.ow);
    const bool on0 = lane < a.ow, on1 = lane + 64 < a.ow;
    const uint32_t blank = pb.blank_dword();
    // scanlines whose horizontal sums the painter can give without painting (P::FAST_ROWS; SpaceInvaders: scanlines that hold
    // nothing but enemies): those of B's that frame A cannot change
    uint64_t fast[4] = {0ull, 0ull, 0ull, 0ull};
    if (P::FAST_ROWS && !AGENT_DIAG(a, 16)) {
        if (lane < 8) L.masks[0][lane] = P::fast_row_word(&L.cls[1][0][0], lane) & ~(two ? L.masks[1][lane] : 0u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t lo = __builtin_amdgcn_readfirstlane(L.masks[0][2 * k]), hi = __builtin_amdgcn_readfirstlane(L.masks[0][2 * k + 1]);
            fast[k] = (uint64_t)lo | ((uint64_t)hi << 32);
        }
        pb.fast_init(c0, c1, on0, on1);
    }
    for (int i = lane; i < (AgentFusedLds<P>::ROWB - W) / 4; i += 64) reinterpret_cast<uint32_t*>(row + W)[i] = 0u;   // window padding
    uint32_t hb0, hb1;                                                 // horizontal sums of a blank scanline
    {
        for (int i = lane; i < W / 4; i += 64) reinterpret_cast<uint32_t*>(row)[i] = blank;
        __builtin_amdgcn_wave_barrier();
        hb0 = on0 ? hsum(row, c0) : 0u;
        hb1 = on1 ? hsum(row, c1) : 0u;
        __builtin_amdgcn_wave_barrier();
    }
    // Two forms of the reduction over the scanlines, chosen by the painter (P::SPARSE_ROWS):
    if constexpr (P::SPARSE_ROWS) {
        // The reduction walks the ACTIVE scanlines only (round 5).  A blank scanline has the same horizontal sums hb in every column
        // (blank byte x W), an output row that meets nothing but blank scanlines is the blank byte itself, and the vertical weights
        // of an output row add up to H: so the plane starts out as the blank byte everywhere and an output row's accumulator as
        // hb x H, and only the scanlines of `need` / `fast` are visited -- each adds its DEVIATION w x (h - hb) (signed 24-bit
        // multiply-adds) to the one or two output rows it overlaps.  Same integers as summing every scanline: Sum w h =
        // hb (H - Sum_active w) + Sum_active w h.  The loop over all H scanlines before it spent 21 % of SpaceInvaders'
        // observation kernel on scanlines that contribute nothing (scripts/agent_diag.sh: 545 of 2 565 us with every row skipped);
    // the agent step at 65 536 envs went 3.085 -> 2.845 ms on one box (with the 16-byte stack commit and the 24-bit multiplies).
        {
            const int nq = (a.oh * a.ow + 3) >> 2;                          // (the plane buffer is a whole number of dwords)
            for (int i = lane; i < nq; i += 64) reinterpret_cast<uint32_t*>(L.vals)[i] = blank;
            __builtin_amdgcn_wave_barrier();
        }
        // source row sy covers [sy*oh, (sy+1)*oh) and output row oy covers [top - H, top) in refined units
        int oy = 0, top = H;
        int32_t dev0[2] = {0, 0}, dev1[2] = {0, 0};                         // [column slot]: deviations of output rows oy / oy + 1
        bool dirty0 = false, dirty1 = false;
        const uint32_t base0 = hb0 * (uint32_t)H + half, base1 = hb1 * (uint32_t)H + half;
        auto flush = [&](int row, const int32_t (&dv)[2]) {
            if (on0) L.vals[row * a.ow + lane] = (uint8_t)(((uint64_t)(base0 + (uint32_t)dv[0]) * a.magic) >> 42);
            if (on1) L.vals[row * a.ow + lane + 64] = (uint8_t)(((uint64_t)(base1 + (uint32_t)dv[1]) * a.magic) >> 42);
        };
        int prev_kind = 0, last_painted = -2;                              // kind (1 + B painted + 2 * A painted) and number of the last composed scanline
        uint32_t hl0 = 0u, hl1 = 0u;                                        // its sums
    #pragma unroll 1
        for (int wi = 0; wi < (AGENT_DIAG(a, 2) ? 0 : (H + 63) / 64); wi++) {
            const uint64_t nw = AGENT_DIAG(a, 4) ? 0ull : sel4(wi, need[0], need[1], need[2], need[3]);
            const uint64_t bw = sel4(wi, pb.busy[0], pb.busy[1], pb.busy[2], pb.busy[3]);
            const uint64_t aw = sel4(wi, need_a[0], need_a[1], need_a[2], need_a[3]);
            const uint64_t rb = sel4(wi, pb.rep[0], pb.rep[1], pb.rep[2], pb.rep[3]), ra = two ? sel4(wi, pa.rep[0], pa.rep[1], pa.rep[2], pa.rep[3]) : 0ull;
            const uint64_t fw = P::FAST_ROWS && !AGENT_DIAG(a, 4) ? sel4(wi, fast[0], fast[1], fast[2], fast[3]) : 0ull;
            uint64_t act = nw | fw;
            if (64 * wi + 64 > H) act &= (1ull << (H - 64 * wi)) - 1ull;     // (H is not a multiple of 64 anywhere: no shift by 64)
    #pragma unroll 1
            while (act) {
                const int bit = (int)__builtin_ctzll(act);
                act &= act - 1ull;
                const int sy = 64 * wi + bit;
                uint32_t h0, h1;
                if (P::FAST_ROWS && ((fw >> bit) & 1ull) && pb.fast_ready(sy)) {
                    h0 = hb0; h1 = hb1;
                    pb.fast_sums(sy, c0, c1, on0, on1, h0, h1);
                } else if ((nw >> bit) & 1ull) {
                    const bool b_on = (bw >> bit) & 1ull, a_on = (aw >> bit) & 1ull;
                    const int kind = 1 + (b_on ? 1 : 0) + (a_on ? 2 : 0);
                    // a scanline that paints exactly like the one above it (same tile / cell / glyph row, same objects) has
                    // the same horizontal sums: neither painted nor reduced again
                    const bool reuse = last_painted == sy - 1 && prev_kind == kind && (!b_on || ((rb >> bit) & 1ull)) && (!a_on || ((ra >> bit) & 1ull));
                    if (!reuse) {
                        uint32_t v[NG];
                        if (b_on) pb.row_dwords(sy, v);
                        else {
    #pragma unroll
                            for (int g = 0; g < NG; g++) v[g] = blank;
                        }
                        if (a_on) {                                        // only then can frame A show different pixels
                            uint32_t va[NG];
                            pa.row_dwords(sy, va);
    #pragma unroll
                            for (int g = 0; g < NG; g++) v[g] = bytemax4(v[g], va[g]);
                        }
    #pragma unroll
                        for (int g = 0; g < NG; g++)
                            if (lane + 64 * g < W / 4) reinterpret_cast<uint32_t*>(row)[lane + 64 * g] = v[g];
                        __builtin_amdgcn_wave_barrier();
                        hl0 = on0 ? hsum(row, c0) : 0u;
                        hl1 = on1 ? hsum(row, c1) : 0u;
                        __builtin_amdgcn_wave_barrier();
                    }
                    h0 = hl0; h1 = hl1;
                    prev_kind = kind;
                    last_painted = sy;
                } else {
                    continue;                                              // a fast row whose sums are not ready and that nobody needs painted: blank
                }
                // the output row this scanline starts in (rows in between met nothing but blank scanlines: they stay as they are)
                const int pos = sy * a.oh;
                while (pos >= top) {
                    if (dirty0) flush(oy, dev0);
                    dev0[0] = dev1[0]; dev0[1] = dev1[1]; dirty0 = dirty1;
                    dev1[0] = 0; dev1[1] = 0; dirty1 = false;
                    oy += 1;
                    top += H;
                }
                const int w_cur = min(pos + a.oh, top) - pos, w_next = a.oh - w_cur;
                const int32_t e0 = (int32_t)h0 - (int32_t)hb0, e1 = (int32_t)h1 - (int32_t)hb1;      // |e| <= 255 W < 2^23, weights <= out_h
                dev0[0] += __mul24(w_cur, e0); dev0[1] += __mul24(w_cur, e1);
                dirty0 = true;
                if (w_next > 0) {
                    dev1[0] += __mul24(w_next, e0); dev1[1] += __mul24(w_next, e1);
                    dirty1 = true;
                }
            }
        }
        if (dirty0) flush(oy, dev0);
        if (dirty1 && oy + 1 < a.oh) flush(oy + 1, dev1);
        __builtin_amdgcn_wave_barrier();
    } else {
    // The scanlines where the horizontal sums CHANGE, and only those (Amidar, GridWorld: the board fills the frame, but a tile /
    // cell row paints the same scanline several times over).  A scanline keeps the sums of the one above it when it paints
    // exactly like it -- same painters on, and each painter says so (rep masks) -- or when both are blank; the rest are found
    // with mask arithmetic per 64-scanline word (the previous word's last bit carried in), and the run since the previous such
    // scanline adds h x (its overlap with each output row): one turn per change plus one per finished output row instead of
    // one per source scanline.  Same integers as summing every scanline.  (Round 5; the form before it walked all H scanlines
    // bit by bit, and the sparse walk above it visits every ACTIVE scanline, which for these two games is nearly all of them.)
        static_assert(!P::FAST_ROWS, "fast rows belong to the sparse form");
        uint32_t acc[2] = {0, 0};                                          // [column slot]: the output row being filled
        int oy = 0, top = H, pos = 0;                                      // source row sy covers [sy*oh, (sy+1)*oh), output row oy [oy*H, (oy+1)*H)
        uint32_t h0 = hb0, h1 = hb1;                                       // sums of the current run
        auto run_to = [&](int end) {                                       // [pos, end) in refined units has the sums h0 / h1   (wave-uniform)
            while (top <= end) {
                const uint32_t w = (uint32_t)(top - pos);                  // <= H < 2^8, sums <= 255 W: 24-bit operands, full-rate v_mad_u32_u24
                const uint32_t s0 = acc[0] + __umul24(w, h0), s1 = acc[1] + __umul24(w, h1);
                if (on0) L.vals[oy * a.ow + lane] = (uint8_t)(((uint64_t)(s0 + half) * a.magic) >> 42);
                if (on1) L.vals[oy * a.ow + lane + 64] = (uint8_t)(((uint64_t)(s1 + half) * a.magic) >> 42);
                acc[0] = 0; acc[1] = 0;
                pos = top; top += H; oy += 1;
            }
            const uint32_t w = (uint32_t)(end - pos);
            acc[0] += __umul24(w, h0); acc[1] += __umul24(w, h1);
            pos = end;
        };
        uint64_t cn = 0ull, cb = 0ull, ca = 0ull;                          // the previous word's last scanline: composed / B on / A on
    #pragma unroll 1
        for (int wi = 0; wi < (AGENT_DIAG(a, 2) ? 0 : (H + 63) / 64); wi++) {
            const uint64_t nw = AGENT_DIAG(a, 4) ? 0ull : sel4(wi, need[0], need[1], need[2], need[3]);
            const uint64_t bw = nw & sel4(wi, pb.busy[0], pb.busy[1], pb.busy[2], pb.busy[3]);
            const uint64_t aw = nw & sel4(wi, need_a[0], need_a[1], need_a[2], need_a[3]);
            const uint64_t rb = sel4(wi, pb.rep[0], pb.rep[1], pb.rep[2], pb.rep[3]), ra = two ? sel4(wi, pa.rep[0], pa.rep[1], pa.rep[2], pa.rep[3]) : 0ull;
            const uint64_t pn = (nw << 1) | cn, pbw = (bw << 1) | cb, paw = (aw << 1) | ca;     // the same three of the scanline above
            cn = nw >> 63; cb = bw >> 63; ca = aw >> 63;
            // paints exactly like the scanline above: both composed, the same painters on, each of them repeating
            const uint64_t reuse = nw & pn & ~(bw ^ pbw) & ~(aw ^ paw) & (~bw | rb) & (~aw | ra);
            const uint64_t compose = nw & ~reuse;
            uint64_t ev = compose | (~nw & pn);                            // ... or the first blank scanline after a composed one
            if (64 * wi + 64 > H) ev &= (1ull << (H - 64 * wi)) - 1ull;     // (H is not a multiple of 64 anywhere: no shift by 64)
    #pragma unroll 1
            for (; ev; ev &= ev - 1ull) {
                const int bit = (int)__builtin_ctzll(ev), sy = 64 * wi + bit;
                run_to(sy * a.oh);
                if ((compose >> bit) & 1ull) {
                    const bool b_on = (bw >> bit) & 1ull, a_on = (aw >> bit) & 1ull;
                    uint32_t v[NG];
                    if (b_on) pb.row_dwords(sy, v);
                    else {
    #pragma unroll
                        for (int g = 0; g < NG; g++) v[g] = blank;
                    }
                    if (a_on) {                                            // only then can frame A show different pixels
                        uint32_t va[NG];
                        pa.row_dwords(sy, va);
    #pragma unroll
                        for (int g = 0; g < NG; g++) v[g] = bytemax4(v[g], va[g]);
                    }
    #pragma unroll
                    for (int g = 0; g < NG; g++)
                        if (lane + 64 * g < W / 4) reinterpret_cast<uint32_t*>(row)[lane + 64 * g] = v[g];
                    __builtin_amdgcn_wave_barrier();
                    h0 = on0 ? hsum(row, c0) : 0u;
                    h1 = on1 ? hsum(row, c1) : 0u;
                    __builtin_amdgcn_wave_barrier();
                } else {
                    h0 = hb0; h1 = hb1;
                }
            }
        }
        run_to(H * a.oh);
    }
    // the read-modify-write of the frame stack in one sweep with many loads in flight (a dependent load -> store per
    // output row, even fetched a row ahead, left this kernel waiting on HBM latency 84 times per env)
    if (!AGENT_DIAG(a, 1)) observation_commit<S>(L.vals, a, env, lane, sel.zero);
}
