// gridworld.hip -- GridWorld on gfx950: thread-per-env step, wave-per-env rasteriser.
//
// The reference holds this game's env class (toybox/envs/atari/gridworld.py:8-13) and two golden dumps
// (toybox/interventions/defaults/gridworld_config_default.json, gridworld_state_default.json); the rules are in the
// absent ctoybox core, so movement, reward bookkeeping and the picture are this repo's specification (SPEC.md
// "GridWorld"), restated independently in the CPU checker.
//
// HBM layout: nine int32 scalars as SoA [field][N]; the tile table env-major [N][16][3 words]; the board env-major
// [N][32*32] bytes.  A step touches the scalars (coalesced), one board byte and one tile record, so it is one thread
// per env; the rasteriser is one wavefront per env with 8-scanline LDS units like the other games.
#include "tbx_common.hpp"
#include "raster.hpp"
#include "agent_device.hpp"

#include <cstdlib>
#include <cstring>

namespace {

enum { G_SCORE, G_OVER, G_PX, G_PY, G_BECOMES, G_W, G_H, G_NT, G_PCOL, GF };
constexpr int GD = TBX_GW_MAX_DIM, GT = TBX_GW_MAX_TILES, CELLS = GD * GD;

struct GwDev {
    int n;
    int32_t* sc;        // [GF][N]
    uint32_t* tiles;    // [N][GT][3]: colour, reward, goal | walkable << 8
    uint8_t* grid;      // [N][CELLS]
    const tbx_gridworld_config_t* cfg;   // device copy of the engine's config
    uint64_t* sim_rng;
    int32_t *prev_score, *reward, *lives_out, *score_out;
    uint8_t* done;
    uint64_t* packed;
    uint32_t* err_flag;
};

struct GwT { int32_t score, over, px, py, becomes, w, h, nt; };

__device__ __forceinline__ void gw_load(const GwDev& d, int env, GwT& s)
{
    const size_t N = (size_t)d.n;
    s.score = d.sc[G_SCORE * N + env]; s.over = d.sc[G_OVER * N + env];
    s.px = d.sc[G_PX * N + env]; s.py = d.sc[G_PY * N + env];
    s.becomes = d.sc[G_BECOMES * N + env];
    s.w = d.sc[G_W * N + env]; s.h = d.sc[G_H * N + env]; s.nt = d.sc[G_NT * N + env];
}
__device__ __forceinline__ void gw_store(const GwDev& d, int env, const GwT& s)
{
    const size_t N = (size_t)d.n;
    d.sc[G_SCORE * N + env] = s.score; d.sc[G_OVER * N + env] = s.over;
    d.sc[G_PX * N + env] = s.px; d.sc[G_PY * N + env] = s.py;
    d.sc[G_BECOMES * N + env] = s.becomes;
    d.sc[G_W * N + env] = s.w; d.sc[G_H * N + env] = s.h; d.sc[G_NT * N + env] = s.nt;
}

__host__ __device__ __forceinline__ uint32_t tile_flags(const tbx_gw_tile_t& t) { return (t.goal ? 1u : 0u) | (t.walkable ? 256u : 0u); }

// the board and tile table of the config, copied by `lanes` cooperating lanes (1 for the thread-per-env callers)
__device__ __forceinline__ void gw_copy_board(const GwDev& d, int env, int lane, int lanes)
{
    const tbx_gridworld_config_t& c = *d.cfg;
    uint32_t* tiles = d.tiles + (size_t)env * GT * 3;
    for (int t = lane; t < GT; t += lanes) {
        tiles[t * 3 + 0] = pack_color(c.tiles[t].color);
        tiles[t * 3 + 1] = (uint32_t)c.tiles[t].reward;
        tiles[t * 3 + 2] = tile_flags(c.tiles[t]);
    }
    const uint32_t* src = reinterpret_cast<const uint32_t*>(c.grid);   // 4-byte aligned in the record
    uint32_t* dst = reinterpret_cast<uint32_t*>(d.grid + (size_t)env * CELLS);
    for (int i = lane; i < CELLS / 4; i += lanes) dst[i] = src[i];
}

__device__ __forceinline__ void gw_new_scalars(const GwDev& d, GwT& s)
{
    const tbx_gridworld_config_t& c = *d.cfg;
    s.score = 0; s.over = 0;
    s.px = c.player_start_x; s.py = c.player_start_y;
    s.becomes = c.reward_becomes;
    s.w = c.width; s.h = c.height; s.nt = c.n_tiles;
}

// one frame of one env: at most one cell in the held direction (up, down, left, right in that priority)
__device__ __forceinline__ void gw_step(const GwDev& d, int env, GwT& s, uint32_t buttons)
{
    if (s.over) return;
    int dx = 0, dy = 0;
    if (buttons & TBX_BTN_UP) dy = -1;
    else if (buttons & TBX_BTN_DOWN) dy = 1;
    else if (buttons & TBX_BTN_LEFT) dx = -1;
    else if (buttons & TBX_BTN_RIGHT) dx = 1;
    else return;
    const int nx = s.px + dx, ny = s.py + dy;
    if (nx < 0 || ny < 0 || nx >= s.w || ny >= s.h || nx >= GD || ny >= GD) return;
    uint8_t* cell = d.grid + (size_t)env * CELLS + ny * GD + nx;
    const int id = *cell;
    if (id >= s.nt || id >= GT) return;
    const uint32_t* t = d.tiles + ((size_t)env * GT + id) * 3;
    const uint32_t fl = t[2];
    if (!(fl & 256u)) return;
    const int32_t reward = (int32_t)t[1];
    s.px = nx; s.py = ny;
    s.score += reward;
    if (reward != 0) *cell = (uint8_t)s.becomes;
    if (fl & 1u) s.over = 1;
}

__global__ __launch_bounds__(TBX_BLOCK) void gw_new_game_kernel(GwDev d, const uint8_t* mask)
{
    const int lane = threadIdx.x & 63;
    const int env = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (env >= d.n) return;
    if (mask && !mask[env]) return;
    gw_copy_board(d, env, lane, 64);
    if (lane == 0) {
        GwT s;
        gw_new_scalars(d, s);
        gw_store(d, env, s);
        d.sc[G_PCOL * (size_t)d.n + env] = (int32_t)pack_color(d.cfg->player_color);
        d.prev_score[env] = 0;
    }
}

// one frame of one env on one thread
__device__ __forceinline__ void gw_step_body(const GwDev& d, const ActionSource& src, uint32_t flags, int env)
{
    if (src.exec_flag) src.exec_flag[env] = tbx_agent_env_finished(src, env) ? 0 : 1;
    if (tbx_agent_env_finished(src, env)) return;     // MaxAndSkipEnv left its loop when this env's game ended
    uint32_t buttons;
    if (src.single_env >= 0) {
        buttons = src.single_buttons;
    } else {
        int a;
        if (src.actions) a = src.actions[env];
        else {
            const uint64_t h = tbx_splitmix64(src.seed ^ ((src.env_offset + (uint64_t)env) << 32) ^ src.t);
            a = tbx_legal_action(TBX_GAME_GRIDWORLD, (int)(h % 5ull));
        }
        buttons = tbx_ale_buttons(a);
        if (buttons == 0xFFu) { buttons = 0; atomicOr(d.err_flag, 1u); }
    }
    GwT s;
    gw_load(d, env, s);
    gw_step(d, env, s, buttons);
    int32_t rew = s.score - d.prev_score[env];
    if (rew < 0) rew = 0;
    const int32_t out_lives = s.over ? 0 : 1, out_score = s.score;
    const bool is_done = out_lives <= 0;
    int32_t prev = s.score;
    // auto-reset: the wave copies the 1 KB board of each of its finished envs together, one env at a time
    const bool resets = is_done && (flags & TBX_STEP_AUTO_RESET);
    const uint64_t act = __ballot(true);                 // the last wave of a launch may be partly empty
    const int n_act = __popcll(act), rank = __popcll(act & ((1ull << (threadIdx.x & 63)) - 1ull));
    for (uint64_t m = __ballot(resets); m; m &= m - 1) {
        const int env_r = __builtin_amdgcn_readlane(env, (int)__builtin_ctzll(m));
        gw_copy_board(d, env_r, rank, n_act);
    }
    if (resets) {
        gw_new_scalars(d, s);
        d.sc[G_PCOL * (size_t)d.n + env] = (int32_t)pack_color(d.cfg->player_color);
        prev = 0;
    }
    gw_store(d, env, s);
    d.prev_score[env] = prev;
    d.reward[env] = rew;
    d.done[env] = is_done ? 1 : 0;
    d.lives_out[env] = out_lives;
    d.score_out[env] = out_score;
    d.packed[env] = (uint64_t)(uint32_t)rew | ((uint64_t)(is_done ? 1u : 0u) << 32) | ((uint64_t)(uint32_t)out_lives << 40);
    tbx_accumulate(src, env, rew, is_done);
}


__global__ __launch_bounds__(128) void gw_step_kernel(GwDev d, ActionSource src, uint32_t flags, int first_env, int count)
{
    const int rel = blockIdx.x * blockDim.x + threadIdx.x;
    if (rel >= count) return;
    gw_step_body(d, src, flags, first_env + rel);
}

// reset-time wrappers of the agent layer (agent_device.hpp, AgentResetProc), thread per flagged env
// the dynamic state of one env (scalars, tile table, board) copied live -> slot by `lanes` cooperating lanes
__device__ __forceinline__ void gw_copy_env(const GwDev& dst, const GwDev& src, int env, int lane, int lanes)
{
    const size_t N = (size_t)src.n;
    static_assert(CELLS % 16 == 0 && (GT * 3 * 4) % 16 == 0, "16-byte copies");
    // 16 bytes per lane: the 1 KB board is ONE load and one store per wave, the tile table a second pair (all loads first)
    const uint4* gs = reinterpret_cast<const uint4*>(src.grid + (size_t)env * CELLS);
    uint4* gd = reinterpret_cast<uint4*>(dst.grid + (size_t)env * CELLS);
    const uint4* ts = reinterpret_cast<const uint4*>(src.tiles + (size_t)env * GT * 3);
    uint4* td = reinterpret_cast<uint4*>(dst.tiles + (size_t)env * GT * 3);
    if (lanes == 64 && CELLS / 16 == 64) {
        const uint4 g = gs[lane];
        uint4 t = make_uint4(0u, 0u, 0u, 0u);
        int32_t sc = 0;
        if (lane < GT * 3 / 4) t = ts[lane];
        if (lane < GF) sc = src.sc[(size_t)lane * N + env];
        gd[lane] = g;
        if (lane < GT * 3 / 4) td[lane] = t;
        if (lane < GF) dst.sc[(size_t)lane * N + env] = sc;
        return;
    }
    for (int f = lane; f < GF; f += lanes) dst.sc[(size_t)f * N + env] = src.sc[(size_t)f * N + env];
    for (int i = lane; i < GT * 3 / 4; i += lanes) td[i] = ts[i];
    for (int i = lane; i < CELLS / 16; i += lanes) gd[i] = gs[i];
}

// agent layer, single-frame launches: the envs that ran the frame copy their state into a buffer slot
__global__ __launch_bounds__(TBX_BLOCK) void gw_snapshot_kernel(GwDev dst, GwDev src, const uint8_t* exec_flag, uint8_t* buf_valid, int bit)
{
    const int lane = threadIdx.x & 63;
    const int env = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (env >= src.n) return;
    if (!wave_uniform((int)exec_flag[env])) return;
    gw_copy_env(dst, src, env, lane, 64);
    if (lane == 0) buf_valid[env] |= (uint8_t)bit;
}

// One WAVE per flagged env: every lane runs the (scalar) procedure redundantly -- same loads, same values, same stores -- so
// that the 1 KB board copies of new_game() and of the frame-buffer snapshots are done by 64 lanes instead of one (a single
// thread copying them was 166 us per agent step at 65 536 envs)
struct GwAgentEnv {
    const GwDev& d;
    const GwDev& slot_a;
    const GwDev& slot_b;
    int env;
    int lane;
    GwT& s;
    __device__ __forceinline__ void snapshot(int slot)
    {
        const GwDev& dst = slot ? slot_b : slot_a;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the cell gw_step() wrote is read by another lane below
        gw_copy_env(dst, d, env, lane, 64);  // board, tile table, player colour; the scalars held in registers follow
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // lane 0's scalars land after the copied ones
        if (lane == 0) gw_store(dst, env, s);
    }
    __device__ __forceinline__ void step(uint32_t buttons) { gw_step(d, env, s, buttons); }
    __device__ __forceinline__ void new_game()
    {
        gw_copy_board(d, env, lane, 64);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // gw_step() reads cells other lanes wrote
        gw_new_scalars(d, s);
        if (lane == 0) d.sc[G_PCOL * (size_t)d.n + env] = (int32_t)pack_color(d.cfg->player_color);
    }
    __device__ __forceinline__ int lives() const { return s.over ? 0 : 1; }
    __device__ __forceinline__ int score() const { return s.score; }
};

__global__ __launch_bounds__(TBX_BLOCK) void gw_agent_reset_kernel(GwDev d, GwDev slot_a, GwDev slot_b, AgentResetArgs r)
{
    const int lane = threadIdx.x & 63;
    // a persistent grid walks the compact list of flagged envs (or every env when there is no list)
    const int wave_id = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6)), n_waves = gridDim.x * TBX_WAVES_PER_BLOCK;
    const int total = r.list ? wave_uniform(*r.count) : d.n;
    for (int it = wave_id; it < total; it += n_waves) {
        const int env = r.list ? wave_uniform(r.list[it]) : it;
        if (wave_uniform((int)r.kind[env]) == 0) continue;
        GwT s;
        gw_load(d, env, s);
        AgentMonitor m = agent_monitor_load(r, env);
        GwAgentEnv ops{d, slot_a, slot_b, env, lane, s};
        AgentResetProc<GwAgentEnv> proc{ops, r, m, r.env_offset + (uint64_t)env, d.prev_score[env], (uint32_t)r.buf_valid[env],
                                        r.noop_override ? r.noop_override[env] : 0, false};
        proc.run();
        if (lane == 0) {
            gw_store(d, env, s);
            d.prev_score[env] = proc.prev;
            agent_monitor_store(r, env, m, proc.valid, proc.obs_raw);
        }
    }
}

// ------------------------------------------------------------------ render

constexpr int GW_UNIT_ROWS = 8;

// Everything one wave needs to paint scanlines of one env: lane l -> pixels 4l..4l+3 (lanes 0..39); lane t < 16 also
// keeps tile t's colour for the board lookups (ds_bpermute).  A scanline only changes when it enters the next cell row,
// so the four colours are rebuilt once per cell row.
template <int C>
struct GwPainter {
    typedef GwDev Dev;
    static constexpr int W = TBX_GW_W, H = TBX_GW_H, NG = 1;
    static constexpr bool FAST_ROWS = false;      // (agent_fused_wave: no scanline class with sums known without painting)
    static constexpr bool SPARSE_ROWS = false;    // (agent_fused_wave: nearly every scanline is busy but most repeat the one above -- the walk over the CHANGES)
    static __device__ __forceinline__ uint32_t fast_row_word(const uint32_t*, int) { return 0u; }
    __device__ __forceinline__ void fast_init(const ColTaps&, const ColTaps&, bool, bool) {}
    __device__ __forceinline__ bool fast_ready(int) const { return false; }
    __device__ __forceinline__ void fast_sums(int, const ColTaps&, const ColTaps&, bool, bool, uint32_t&, uint32_t&) const {}
    enum { CLS_BOARD, CLS_PLAYER, NCLS, NLDS = NCLS };
    int lane, gw, gh, tw, th, px, py;
    uint32_t pcol, black, tcol;
    const uint8_t* g;
    bool active;
    int cx[4];
    uint4 cells;                    // this lane's 16 board bytes (for diff_classes)
    uint32_t tile_rec[3];           // lane t < 16: tile t's record
    uint64_t busy[4];
    uint64_t rep[4];                // scanlines in the same cell row as the one above them
    mutable int last_cy;
    mutable uint32_t c[4];

    __device__ __forceinline__ void setup(const GwDev& d, int env, int lane_, uint32_t* cls)
    {
        lane = lane_;
        const size_t N = (size_t)d.n;
        gw = wave_uniform(d.sc[G_W * N + env]); gh = wave_uniform(d.sc[G_H * N + env]);
        gw = gw < 1 ? 1 : gw > GD ? GD : gw;
        gh = gh < 1 ? 1 : gh > GD ? GD : gh;
        tw = W / gw; th = H / gh;
        px = wave_uniform(d.sc[G_PX * N + env]); py = wave_uniform(d.sc[G_PY * N + env]);
        const int nt = wave_uniform(d.sc[G_NT * N + env]);
        pcol = pix_of<C>((uint32_t)wave_uniform(d.sc[G_PCOL * N + env]));
        black = pix_of<C>(0xFF000000u);
        const uint32_t* trec = d.tiles + ((size_t)env * GT + (lane < GT ? lane : 0)) * 3;
        tile_rec[0] = trec[0]; tile_rec[1] = trec[1]; tile_rec[2] = trec[2];
        tcol = (lane < GT && lane < nt) ? pix_of<C>(tile_rec[0]) : black;
        g = d.grid + (size_t)env * CELLS;
        cells = reinterpret_cast<const uint4*>(g)[lane];
        active = lane < W / 4;
#pragma unroll
        for (int k = 0; k < 4; k++) cx[k] = (lane * 4 + k) / tw;
        last_cy = -1;
#pragma unroll
        for (int k = 0; k < 4; k++) c[k] = black;
        // scanline masks per class (all wave-uniform): the board band and the player's cell row
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint64_t wb = row_range_bits(0, (long)gh * th, k);
                const bool p_in = px >= 0 && px < gw && py >= 0 && py < gh;
                const uint64_t wp = p_in ? row_range_bits((long)py * th, (long)py * th + th, k) : 0ull;
                cls[CLS_BOARD * 8 + 2 * k] = (uint32_t)wb; cls[CLS_BOARD * 8 + 2 * k + 1] = (uint32_t)(wb >> 32);
                cls[CLS_PLAYER * 8 + 2 * k] = (uint32_t)wp; cls[CLS_PLAYER * 8 + 2 * k + 1] = (uint32_t)(wp >> 32);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) busy[k] = row_range_bits(0, (long)gh * th, k);
        {   // every scanline of the board band except the first of each cell row
            uint64_t firsts[4] = {0ull, 0ull, 0ull, 0ull};
            for (int r = 0; r < gh; r++)
#pragma unroll
                for (int k = 0; k < 4; k++) firsts[k] |= row_range_bits((long)r * th, (long)r * th + 1, k);
#pragma unroll
            for (int k = 0; k < 4; k++) rep[k] = busy[k] & ~firsts[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    static __device__ __forceinline__ uint32_t diff_classes(const GwPainter& a, const GwPainter& b)
    {
        uint32_t m = 0u;
        const bool cell_diff = a.cells.x != b.cells.x || a.cells.y != b.cells.y || a.cells.z != b.cells.z || a.cells.w != b.cells.w;
        const bool tile_diff = a.lane < GT && (a.tile_rec[0] != b.tile_rec[0] || a.tcol != b.tcol);
        if (a.gw != b.gw || a.gh != b.gh || __ballot(cell_diff || tile_diff)) m |= 1u << CLS_BOARD;
        if (a.px != b.px || a.py != b.py || a.pcol != b.pcol) m |= 1u << CLS_PLAYER;
        return m;
    }

    __device__ __forceinline__ void paint_row(int y, uint32_t (&out)[4]) const
    {
        const int cy = y / th;
        if (cy != last_cy) {
            last_cy = cy;
            // the cell row's sixteen board bytes out of lane cy's registers (setup loaded them): a load here would sit inside
            // the unit loop, and its `s_waitcnt vmcnt(0)` waits for the wave's frame stores as well
            static_assert(GD == 32, "a grid row is 32 bytes: lanes 2 cy and 2 cy + 1 of `cells`");
            const int rl = wave_uniform(2 * (cy < GD ? cy : GD - 1));
            uint32_t row[8];
            row[0] = (uint32_t)__builtin_amdgcn_readlane((int)cells.x, rl); row[1] = (uint32_t)__builtin_amdgcn_readlane((int)cells.y, rl);
            row[2] = (uint32_t)__builtin_amdgcn_readlane((int)cells.z, rl); row[3] = (uint32_t)__builtin_amdgcn_readlane((int)cells.w, rl);
            row[4] = (uint32_t)__builtin_amdgcn_readlane((int)cells.x, rl + 1); row[5] = (uint32_t)__builtin_amdgcn_readlane((int)cells.y, rl + 1);
            row[6] = (uint32_t)__builtin_amdgcn_readlane((int)cells.z, rl + 1); row[7] = (uint32_t)__builtin_amdgcn_readlane((int)cells.w, rl + 1);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool inside = active && cx[k] < gw && cy < gh;
                uint32_t word = row[0];
#pragma unroll
                for (int q = 1; q < 8; q++) word = (cx[k] >> 2) == q ? row[q] : word;
                const int id = inside ? (int)((word >> (8 * (cx[k] & 3))) & 255u) : 255;
                const uint32_t tc = __shfl(tcol, id & 15);
                c[k] = !inside ? black : (cx[k] == px && cy == py) ? pcol : id < GT ? tc : black;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) out[k] = c[k];
    }
};

struct GwGrayPainter : GwPainter<1> {
    static __device__ __forceinline__ uint32_t diff_classes(const GwGrayPainter& a, const GwGrayPainter& b) { return GwPainter<1>::diff_classes(a, b); }
    __device__ __forceinline__ uint32_t blank_dword() const { return black * 0x01010101u; }
    __device__ __forceinline__ void row_dwords(int y, uint32_t (&v)[1]) const
    {
        uint32_t p[4];
        paint_row(y, p);
        v[0] = p[0] | (p[1] << 8) | (p[2] << 16) | (p[3] << 24);
    }
};

// units part, part + split, ... of one env's frame from a painter that has been set up, on one wave
template <int C>
__device__ __forceinline__ void gw_paint_units(const GwPainter<C>& p, uint8_t* __restrict__ dst, int env, int lane,
                                               const RowStager<C, TBX_GW_W, GW_UNIT_ROWS>& st, int part, int split)
{
    constexpr int UNITS = TBX_GW_H / GW_UNIT_ROWS;
    using Stager = RowStager<C, TBX_GW_W, GW_UNIT_ROWS>;
    for (int u = part; u < UNITS; u += split) {
        const int unit = split > 1 ? u : (u + env) % UNITS;  // one wave per frame: rotate the start so waves do not march in lockstep
        for (int r = 0; r < GW_UNIT_ROWS; r++) {
            uint32_t px[4];
            p.paint_row(unit * GW_UNIT_ROWS + r, px);
            if (p.active) st.put4p(r, lane, px[0], px[1], px[2], px[3]);
        }
        st.flush(dst + (size_t)unit * Stager::UNIT_BYTES, lane);
    }
}

template <int C, bool ALT>
__global__ __launch_bounds__(TBX_BLOCK) void gw_render_kernel(GwDev d, uint8_t* out, int first_env, int count, int split, GwDev d_alt,
                                                              const uint8_t* __restrict__ pick_alt)
{
    constexpr int W = TBX_GW_W, H = TBX_GW_H;
    using Stager = RowStager<C, W, GW_UNIT_ROWS>;
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[TBX_WAVES_PER_BLOCK * Stager::UNIT_BYTES];
    __shared__ uint32_t lds_mask[TBX_WAVES_PER_BLOCK][GwPainter<C>::NCLS * 8];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wid = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + wave);   // `split` waves share a frame (see breakout.hip)
    const int rel = wid / split, part = wid - rel * split;
    if (rel >= count) return;
    const int env = first_env + rel;
    Stager st{lds_all + wave * Stager::UNIT_BYTES};
    GwPainter<C> p;
    // (agent layer, generic path: flagged envs are painted from d_alt)
    GwDev src = d;                                             // by VALUE: a select between references to kernel arguments puts both into scratch
    if (ALT && pick_alt && wave_uniform((int)pick_alt[env])) src = d_alt;   // (ALT: the agent layer's generic path only)
    p.setup(src, env, lane, lds_mask[wave]);
    gw_paint_units<C>(p, out + (size_t)rel * H * W * C, env, lane, st, part, split);
}

// resident single-env form (tbx_serve_loop, tbx_common.hpp): one wave, env 0; steps on request and, when the request asks for
// it, rasterises the env straight into the engine's mapped pinned frame buffer
template <int C>
__device__ __forceinline__ void gw_serve_paint(const GwDev& d, uint8_t* frame, int lane, uint8_t* lds, uint32_t* cls, int part, int split)
{
    const RowStager<C, TBX_GW_W, GW_UNIT_ROWS> st{lds};
    GwPainter<C> p;
    p.setup(d, 0, lane, cls);
    gw_paint_units<C>(p, frame, 0, lane, st, part, split);
}

__global__ __launch_bounds__(64 * TBX_SERVE_WAVES) void gw_serve_kernel(GwDev d, TbxServeCtl* ctl)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[TBX_SERVE_WAVES][RowStager<4, TBX_GW_W, GW_UNIT_ROWS>::UNIT_BYTES];
    __shared__ uint32_t cls[TBX_SERVE_WAVES][GwPainter<1>::NCLS * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    tbx_serve_loop(ctl, lane, [&](const ActionSource& src, uint32_t flags) { if (lane == 0) gw_step_body(d, src, flags, 0); },
                   [&](int channels, uint8_t* frame, int part, int split) {
                       switch (channels) {
                       case 1: gw_serve_paint<1>(d, frame, lane, lds[wave], cls[wave], part, split); break;
                       case 3: gw_serve_paint<3>(d, frame, lane, lds[wave], cls[wave], part, split); break;
                       default: gw_serve_paint<4>(d, frame, lane, lds[wave], cls[wave], part, split); break;
                       }
                       return true;
                   },
                   d.reward, d.done, d.lives_out, d.score_out, d.err_flag);
}

// fused agent observation: agent_fused_wave (agent_device.hpp) with two GwGrayPainters per wave
template <int S>
// Held to FIVE waves per SIMD (the LDS of a block allows five): with the newest-plane output and the 16-byte stack commit of round 5
// the depth-4 instantiation asked for 99-101 VGPRs -- four waves -- and the agent step at 65 536 envs lost 3-7 % against round 4;
// at 96 VGPRs it spills 16-52 bytes per lane and runs 3.5 % AHEAD of round 4 (same box: Amidar 1.848 / 1.987 / 1.911 ms pinned /
// unpinned / round 4, GridWorld 1.053 / 1.156 / 1.091).
#ifndef GW_AGENT_WAVES
#define GW_AGENT_WAVES 5
#endif
__global__ __launch_bounds__(TBX_BLOCK) __attribute__((amdgpu_waves_per_eu(GW_AGENT_WAVES))) void gw_agent_warp_kernel(GwDev dLive, GwDev dA, GwDev dB, AgentWarpArgs a, int n)
{
    __shared__ AgentFusedLds<GwGrayPainter> lds[TBX_WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int env = wave_uniform(a.first + blockIdx.x * TBX_WAVES_PER_BLOCK + wave);
    if (env >= a.end) return;
    GwGrayPainter pa, pb;
    agent_fused_wave<S, GwGrayPainter>(pa, pb, dLive, dA, dB, a, env, lane, lds[wave]);
}

// ------------------------------------------------------------------ state records

__global__ void gw_pack_kernel(GwDev d, int env0, tbx_gridworld_state_t* out)
{
    const int env = env0 + blockIdx.x, lane = threadIdx.x;
    const size_t N = (size_t)d.n;
    tbx_gridworld_state_t& o = out[blockIdx.x];
    if (lane == 0) {
        o.score = d.sc[G_SCORE * N + env]; o.game_over = d.sc[G_OVER * N + env];
        o.player_x = d.sc[G_PX * N + env]; o.player_y = d.sc[G_PY * N + env];
        o.reward_becomes = d.sc[G_BECOMES * N + env];
        o.width = d.sc[G_W * N + env]; o.height = d.sc[G_H * N + env]; o.n_tiles = d.sc[G_NT * N + env];
        o.player_color = unpack_color((uint32_t)d.sc[G_PCOL * N + env]);
    }
    if (lane < GT) {
        const uint32_t* t = d.tiles + ((size_t)env * GT + lane) * 3;
        tbx_gw_tile_t tile;
        tile.color = unpack_color(t[0]);
        tile.reward = (int32_t)t[1];
        tile.goal = (t[2] & 1u) ? 1 : 0; tile.walkable = (t[2] & 256u) ? 1 : 0;
        tile._pad[0] = tile._pad[1] = 0;
        o.tiles[lane] = tile;
    }
    for (int i = lane; i < CELLS; i += 64) o.grid[i] = d.grid[(size_t)env * CELLS + i];
}

__global__ void gw_unpack_kernel(GwDev d, int env0, const tbx_gridworld_state_t* in)
{
    const int env = env0 + blockIdx.x, lane = threadIdx.x;
    const size_t N = (size_t)d.n;
    const tbx_gridworld_state_t& o = in[blockIdx.x];
    if (lane == 0) {
        d.sc[G_SCORE * N + env] = o.score; d.sc[G_OVER * N + env] = o.game_over ? 1 : 0;
        d.sc[G_PX * N + env] = o.player_x; d.sc[G_PY * N + env] = o.player_y;
        d.sc[G_BECOMES * N + env] = o.reward_becomes;
        d.sc[G_W * N + env] = o.width; d.sc[G_H * N + env] = o.height; d.sc[G_NT * N + env] = o.n_tiles;
        d.sc[G_PCOL * N + env] = (int32_t)pack_color(o.player_color);
        // prev_score stays: like ToyboxBaseEnv.score (envs/atari/base.py:136-142) it follows steps, not state writes
    }
    if (lane < GT) {
        uint32_t* t = d.tiles + ((size_t)env * GT + lane) * 3;
        t[0] = pack_color(o.tiles[lane].color);
        t[1] = (uint32_t)o.tiles[lane].reward;
        t[2] = tile_flags(o.tiles[lane]);
    }
    for (int i = lane; i < CELLS; i += 64) d.grid[(size_t)env * CELLS + i] = o.grid[i];
}

__global__ void gw_scalars_kernel(GwDev d, int32_t* score, int32_t* lives, int32_t* level)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    const size_t N = (size_t)d.n;
    score[i] = d.sc[G_SCORE * N + i];
    lives[i] = d.sc[G_OVER * N + i] ? 0 : 1;
    level[i] = 1;
}

int check_dims(tbx_engine* e, int w, int h, int nt)
{
    if (w < 1 || w > GD || h < 1 || h > GD) return e->fail(TBX_E_UNSUPPORTED, "gridworld: game_size must be 1..32 x 1..32");
    if (nt < 1 || nt > GT) return e->fail(TBX_E_UNSUPPORTED, "gridworld: 1..16 tiles");
    return TBX_OK;
}

struct GridWorldOps : GameOps {
    GwDev d{};
    tbx_gridworld_config_t cfg{};
    tbx_gridworld_config_t* cfg_dev = nullptr;

    int height() const override { return TBX_GW_H; }
    int width() const override { return TBX_GW_W; }
    size_t state_size() const override { return sizeof(tbx_gridworld_state_t); }
    size_t config_size() const override { return sizeof(tbx_gridworld_config_t); }

    int load_cfg(tbx_engine* e, const tbx_gridworld_config_t& k)
    {
        int rc = check_dims(e, k.width, k.height, k.n_tiles);
        if (rc) return rc;
        cfg = k;
        TBX_HIP(hipMemcpy(cfg_dev, &cfg, sizeof cfg, hipMemcpyHostToDevice));
        return TBX_OK;
    }

    int init(tbx_engine* e, const void* cfg_pod, size_t cfg_size) override
    {
        if (!cfg_pod || cfg_size != sizeof(tbx_gridworld_config_t)) return e->fail(TBX_E_INVALID, "gridworld: config size mismatch");
        const size_t N = (size_t)e->n;
        TBX_HIP(hipMalloc((void**)&cfg_dev, sizeof(tbx_gridworld_config_t)));
        tbx_gridworld_config_t k;
        memcpy(&k, cfg_pod, sizeof k);
        int rc = load_cfg(e, k);
        if (rc) return rc;
        d.n = e->n;
        d.cfg = cfg_dev;
        d.sim_rng = e->sim_rng; d.prev_score = e->prev_score; d.reward = e->reward; d.done = e->done;
        d.lives_out = e->lives_out; d.score_out = e->score_out; d.packed = e->packed; d.err_flag = e->err_flag;
        TBX_HIP(hipMalloc((void**)&d.sc, (size_t)GF * N * sizeof(int32_t)));
        TBX_HIP(hipMalloc((void**)&d.tiles, N * GT * 3 * sizeof(uint32_t)));
        TBX_HIP(hipMalloc((void**)&d.grid, N * CELLS));
        return TBX_OK;
    }

    void rebind_outputs(tbx_engine* e) override
    {
        d.reward = e->reward; d.done = e->done; d.lives_out = e->lives_out; d.score_out = e->score_out; d.packed = e->packed;
    }

    void destroy(tbx_engine*) override
    {
        hipFree(d.sc); hipFree(d.tiles); hipFree(d.grid); hipFree(cfg_dev);
        hipFree(dA.sc); hipFree(dA.tiles); hipFree(dA.grid);
        hipFree(dB.sc); hipFree(dB.tiles); hipFree(dB.grid);
    }

    int get_config(tbx_engine*, void* pod) override { memcpy(pod, &cfg, sizeof cfg); return TBX_OK; }
    int set_config(tbx_engine* e, const void* pod) override
    {
        tbx_gridworld_config_t k;
        memcpy(&k, pod, sizeof k);
        TBX_HIP(hipStreamSynchronize(e->stream));     // kernels in flight still read the old table
        return load_cfg(e, k);
    }

    static dim3 wave_grid(int count) { return dim3((count + TBX_WAVES_PER_BLOCK - 1) / TBX_WAVES_PER_BLOCK); }

    int new_game(tbx_engine* e, const uint8_t* mask_dev, hipStream_t s) override
    {
        hipLaunchKernelGGL(gw_new_game_kernel, wave_grid(e->n), dim3(TBX_BLOCK), 0, s, d, mask_dev);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int step(tbx_engine* e, const ActionSource& src, uint32_t flags, hipStream_t s) override
    {
        int first = 0, count = e->n;
        if (src.single_env >= 0) { first = src.single_env; count = 1; }
        hipLaunchKernelGGL(gw_step_kernel, dim3((count + 127) / 128), dim3(128), 0, s, d, src, flags, first, count);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    bool serve_paints() const override { return true; }
    int serve(tbx_engine* e, TbxServeCtl* ctl_dev, hipStream_t s) override
    {
        hipLaunchKernelGGL(gw_serve_kernel, dim3(1), dim3(64 * TBX_SERVE_WAVES), 0, s, d, ctl_dev);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    // ---- agent layer: MaxAndSkipEnv's two-frame buffer is two snapshots of the dynamic state per env
    GwDev dA{}, dB{};
    bool agent_fused() const override { return true; }
    bool agent_reset_supported() const override { return true; }

    int alloc_slot(tbx_engine* e, GwDev& x)
    {
        if (x.sc) { x.cfg = d.cfg; return TBX_OK; }
        const size_t N = (size_t)e->n;
        x = d;
        x.sc = nullptr; x.tiles = nullptr; x.grid = nullptr;
        TBX_HIP(hipMalloc((void**)&x.sc, (size_t)GF * N * sizeof(int32_t)));
        TBX_HIP(hipMalloc((void**)&x.tiles, N * GT * 3 * sizeof(uint32_t)));
        TBX_HIP(hipMalloc((void**)&x.grid, N * CELLS));
        return TBX_OK;
    }

    int agent_prepare(tbx_engine* e) override
    {
        int rc = alloc_slot(e, dA);
        if (rc) return rc;
        return alloc_slot(e, dB);
    }

    int agent_snapshot(tbx_engine* e, int slot, const uint8_t* exec_flag, uint8_t* buf_valid, hipStream_t s) override
    {
        dA.cfg = dB.cfg = d.cfg;
        hipLaunchKernelGGL(gw_snapshot_kernel, wave_grid(e->n), dim3(TBX_BLOCK), 0, s, slot ? dB : dA, d, exec_flag, buf_valid, slot ? 2 : 1);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int agent_warp(tbx_engine* e, const AgentWarpArgs& a, hipStream_t s) override
    {
        dA.cfg = dB.cfg = d.cfg;
        const dim3 grid = wave_grid(a.end - a.first), block(TBX_BLOCK);
        switch (a.obs ? a.stack : 0) {
        case 0: hipLaunchKernelGGL(gw_agent_warp_kernel<0>, grid, block, 0, s, d, dA, dB, a, e->n); break;      // the plane ring (new_plane = 2), any depth
        case 1: hipLaunchKernelGGL(gw_agent_warp_kernel<1>, grid, block, 0, s, d, dA, dB, a, e->n); break;
        case 2: hipLaunchKernelGGL(gw_agent_warp_kernel<2>, grid, block, 0, s, d, dA, dB, a, e->n); break;
        case 3: hipLaunchKernelGGL(gw_agent_warp_kernel<3>, grid, block, 0, s, d, dA, dB, a, e->n); break;
        default: hipLaunchKernelGGL(gw_agent_warp_kernel<4>, grid, block, 0, s, d, dA, dB, a, e->n); break;
        }
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int agent_reset_envs(tbx_engine* e, const AgentResetArgs& r, hipStream_t s) override
    {
        dA.cfg = dB.cfg = d.cfg;
        const unsigned blocks = (unsigned)((e->n + TBX_WAVES_PER_BLOCK - 1) / TBX_WAVES_PER_BLOCK);
        hipLaunchKernelGGL(gw_agent_reset_kernel, dim3(r.list ? std::min(blocks, 1024u) : blocks), dim3(TBX_BLOCK), 0, s, d, dA, dB, r);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int render_from(tbx_engine* e, int source, const uint8_t* pick_live, uint8_t* out_dev, int channels, hipStream_t s) override
    {
        dA.cfg = dB.cfg = d.cfg;
        const GwDev& src = source == 1 ? dA : source == 2 ? dB : d;
        return render_impl(e, src, d, source ? pick_live : nullptr, out_dev, channels, 0, e->n, s);
    }

    int render(tbx_engine* e, uint8_t* out_dev, int channels, int first_env, int n_envs, hipStream_t s) override
    {
        return render_impl(e, d, d, nullptr, out_dev, channels, first_env, n_envs, s);
    }

    int render_impl(tbx_engine* e, const GwDev& src, const GwDev& alt, const uint8_t* pick_alt, uint8_t* out_dev, int channels, int first_env,
                    int n_envs, hipStream_t s)
    {
        const int split_env = e->opt[TBX_OPT_RENDER_SPLIT];
        // waves per RGB frame (sixteen 8-row units), measured per output buffer with scripts/ubench/rate_addr (eight buffers, one
        // box, [step ; render] at 65 536 envs): 2 / 3 / 4 / 5 / 6 / 7 -> 0.76-0.80 / 0.74-0.81 / 0.72-0.78 / 0.695-0.716 / 0.76-0.775 /
        // 0.845-0.86 ms -- five is 8 % faster than two and its rate no longer depends on the buffer; 16 384 envs 0.189-0.201
        // against 0.195-0.208, 32 768 equal, 4 096 envs 0.047 against 0.0456: five from 16 384 envs, two below; gray and RGBA one
        const int split = split_env > 0 ? split_env : channels == 3 ? (n_envs >= 16384 ? 5 : 2) : 1;
        switch (channels) {
        case 1: if (pick_alt) hipLaunchKernelGGL((gw_render_kernel<1, true>), wave_grid(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, split, alt, pick_alt); else hipLaunchKernelGGL((gw_render_kernel<1, false>), wave_grid(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, split, alt, pick_alt); break;
        case 3: if (pick_alt) hipLaunchKernelGGL((gw_render_kernel<3, true>), wave_grid(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, split, alt, pick_alt); else hipLaunchKernelGGL((gw_render_kernel<3, false>), wave_grid(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, split, alt, pick_alt); break;
        case 4: if (pick_alt) hipLaunchKernelGGL((gw_render_kernel<4, true>), wave_grid(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, split, alt, pick_alt); else hipLaunchKernelGGL((gw_render_kernel<4, false>), wave_grid(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, split, alt, pick_alt); break;
        default: return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
        }
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int pack_state(tbx_engine* e, int env, int count, hipStream_t s) override
    {
        hipLaunchKernelGGL(gw_pack_kernel, dim3(count), dim3(64), 0, s, d, env, (tbx_gridworld_state_t*)e->staging);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int unpack_state(tbx_engine* e, int env, int count, const void* pod_host, hipStream_t s) override
    {
        const auto* sts = (const tbx_gridworld_state_t*)pod_host;
        for (int i = 0; i < count; i++) {
            int rc = check_dims(e, sts[i].width, sts[i].height, sts[i].n_tiles);
            if (rc) return rc;
        }
        TBX_HIP(hipMemcpyAsync(e->staging, pod_host, sizeof(tbx_gridworld_state_t) * (size_t)count, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(gw_unpack_kernel, dim3(count), dim3(64), 0, s, d, env, (const tbx_gridworld_state_t*)e->staging);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int scalars(tbx_engine* e, int32_t* score_dev, int32_t* lives_dev, int32_t* level_dev, hipStream_t s) override
    {
        hipLaunchKernelGGL(gw_scalars_kernel, dim3((e->n + 255) / 256), dim3(256), 0, s, d, score_dev, lives_dev, level_dev);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }
};

}  // namespace

GameOps* tbx_make_gridworld_ops() { return new GridWorldOps(); }
