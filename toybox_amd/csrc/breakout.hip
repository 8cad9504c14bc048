// breakout.hip -- Breakout on gfx950: one 64-lane wavefront per env.
//
// Replaces the per-env Rust transition + rasteriser behind ctoybox.Toybox.apply_ale_action /
// get_state (call sites: /root/reference/toybox/envs/atari/base.py:126,109).  Semantics are the
// ones stated in SPEC.md "Breakout" and restated independently (scalar C) by the CPU checker under oracle/;
// the two are compared bit for bit by tests/test_gpu_parity.py.
//
// Layout in HBM: struct-of-arrays over the env batch for every scalar field (field f of env e at
// base_f[e]); brick liveness as 4 x uint64 bit-words per env ([4][N]).  Brick geometry / colour /
// points are canonical functions of the brick index and the config unless an intervention wrote a
// non-canonical brick, in which case the engine switches to per-env brick tables ([N][field][256],
// brick index fastest so that the 64 lanes of an env's wave read them coalesced).
//
// Lane roles: every lane carries the (wave-uniform) scalar state; lane l owns bricks l, l+64,
// l+128, l+192 for the collision test (ballot -> lowest index) and pixels 4l..4l+3 of each
// scanline in the rasteriser.
//
// Arithmetic is IEEE binary64 with -ffp-contract=off: + - * / sqrt ceil fabs only, all trig is
// host-evaluated into the config tables, so results equal the CPU oracle's bit for bit.

#include "tbx_common.hpp"
#include "raster.hpp"
#include "agent_device.hpp"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int MAXB = TBX_BRK_MAX_BALLS;
constexpr int MAXK = TBX_BRK_MAX_BRICKS / 64;   // brick slots per lane

// per-env custom brick table: [field][256]
struct BrkCustom {
    double x[TBX_BRK_MAX_BRICKS], y[TBX_BRK_MAX_BRICKS], w[TBX_BRK_MAX_BRICKS], h[TBX_BRK_MAX_BRICKS];
    int32_t points[TBX_BRK_MAX_BRICKS], depth[TBX_BRK_MAX_BRICKS], row[TBX_BRK_MAX_BRICKS], col[TBX_BRK_MAX_BRICKS];
    uint32_t color[TBX_BRK_MAX_BRICKS];
    uint8_t destructible[TBX_BRK_MAX_BRICKS];
};

// device view of the SoA state (passed by value as kernel argument)
struct BrkDev {
    int n;
    // engine-common
    uint64_t* sim_rng;      // [2][N]
    int32_t* prev_score;
    int32_t* reward;
    uint8_t* done;
    int32_t* lives_out;
    int32_t* score_out;
    uint64_t* packed;
    uint32_t* err_flag;
    // game state
    uint64_t* rng;          // [2][N]
    int32_t* score;
    int32_t* lives;
    int32_t* level;
    int32_t* flags;         // bit0 is_dead, bit1 reset
    double* paddle;         // [7][N]: x y vx vy width speed ball_radius
    int32_t* n_balls;
    double* balls;          // [16][N]: x[4] y[4] vx[4] vy[4]
    int32_t* n_bricks;
    uint64_t* alive;        // [4][N]
    BrkCustom* custom;      // [N] or nullptr
};

// config as the kernels see it (kernel argument, scalar loads)
struct BrkCfg {
    int32_t start_lives, n_rows, ball_speed_row_depth, n_starts, segments;
    int32_t row_scores[TBX_BRK_MAX_ROWS];
    uint32_t row_colors[TBX_BRK_MAX_ROWS];
    double speed_slow, speed_fast;
    double start_x[TBX_BRK_MAX_STARTS], start_y[TBX_BRK_MAX_STARTS];
    double start_dx[TBX_BRK_MAX_STARTS], start_dy[TBX_BRK_MAX_STARTS];
    double pad_dx[TBX_BRK_MAX_SEGMENTS], pad_dy[TBX_BRK_MAX_SEGMENTS];
    uint32_t bg, frame, paddle, ball;
};

// wave-uniform register copy of one env
struct BrkRegs {
    Rng rng;
    int32_t score, lives, level, flags;
    double px, py, pvx, pvy, pw, pspeed, radius;
    int32_t n_balls, n_bricks;
    double bx[MAXB], by[MAXB], bvx[MAXB], bvy[MAXB];
    uint32_t mybits;   // per lane: bit k = brick (lane + 64k) alive
};

__device__ __forceinline__ void brk_load(const BrkDev& d, int env, int lane, BrkRegs& s)
{
    const size_t N = (size_t)d.n;
    s.rng.s0 = d.rng[env];
    s.rng.s1 = d.rng[N + env];
    s.score = d.score[env];
    s.lives = d.lives[env];
    s.level = d.level[env];
    s.flags = d.flags[env];
    s.px = d.paddle[0 * N + env];
    s.py = d.paddle[1 * N + env];
    s.pvx = d.paddle[2 * N + env];
    s.pvy = d.paddle[3 * N + env];
    s.pw = d.paddle[4 * N + env];
    s.pspeed = d.paddle[5 * N + env];
    s.radius = d.paddle[6 * N + env];
    s.n_balls = d.n_balls[env];
    s.n_bricks = d.n_bricks[env];
#pragma unroll
    for (int b = 0; b < MAXB; b++) {
        s.bx[b] = d.balls[(size_t)(0 * MAXB + b) * N + env];
        s.by[b] = d.balls[(size_t)(1 * MAXB + b) * N + env];
        s.bvx[b] = d.balls[(size_t)(2 * MAXB + b) * N + env];
        s.bvy[b] = d.balls[(size_t)(3 * MAXB + b) * N + env];
    }
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < MAXK; k++) {
        uint64_t w = d.alive[(size_t)k * N + env];
        bits |= (uint32_t)((w >> lane) & 1ull) << k;
    }
    s.mybits = bits;
}

__device__ __forceinline__ void brk_store(const BrkDev& d, int env, int lane, const BrkRegs& s)
{
    const size_t N = (size_t)d.n;
    uint64_t words[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; k++) words[k] = __ballot((s.mybits >> k) & 1u);
    if (lane == 0) {
        d.rng[env] = s.rng.s0;
        d.rng[N + env] = s.rng.s1;
        d.score[env] = s.score;
        d.lives[env] = s.lives;
        d.level[env] = s.level;
        d.flags[env] = s.flags;
        d.paddle[0 * N + env] = s.px;
        d.paddle[1 * N + env] = s.py;
        d.paddle[2 * N + env] = s.pvx;
        d.paddle[3 * N + env] = s.pvy;
        d.paddle[4 * N + env] = s.pw;
        d.paddle[5 * N + env] = s.pspeed;
        d.paddle[6 * N + env] = s.radius;
        d.n_balls[env] = s.n_balls;
        d.n_bricks[env] = s.n_bricks;
#pragma unroll
        for (int b = 0; b < MAXB; b++) {
            d.balls[(size_t)(0 * MAXB + b) * N + env] = s.bx[b];
            d.balls[(size_t)(1 * MAXB + b) * N + env] = s.by[b];
            d.balls[(size_t)(2 * MAXB + b) * N + env] = s.bvx[b];
            d.balls[(size_t)(3 * MAXB + b) * N + env] = s.bvy[b];
        }
#pragma unroll
        for (int k = 0; k < MAXK; k++) d.alive[(size_t)k * N + env] = words[k];
    }
}

__device__ __forceinline__ void brk_start_ball(const BrkCfg& c, BrkRegs& s)
{
    uint64_t i = s.rng.range((uint64_t)c.n_starts);
    int k = s.n_balls;
    if (k >= MAXB) return;
    double x = c.start_x[i], y = c.start_y[i];
    double vx = c.speed_slow * c.start_dx[i], vy = c.speed_slow * c.start_dy[i];
#pragma unroll
    for (int b = 0; b < MAXB; b++)
        if (b == k) { s.bx[b] = x; s.by[b] = y; s.bvx[b] = vx; s.bvy[b] = vy; }
    s.n_balls = k + 1;
}

// canonical brick j of an n_rows wall
__device__ __forceinline__ void brk_canon(int j, int rows, int& row, int& col, double& x, double& y)
{
    col = j / rows;
    row = j - col * rows;
    x = TBX_BRK_LEFT + TBX_BRK_BRICK_W * (double)col;
    y = TBX_BRK_BRICK_Y0 + TBX_BRK_BRICK_H * (double)row;
}

template <bool CUSTOM>
__device__ __forceinline__ void brk_new_game(const BrkDev& d, const BrkCfg& c, int env, int lane, Rng& sim, BrkRegs& s)
{
    s.rng = sim.child();
    s.score = 0;
    s.lives = c.start_lives;
    s.level = 1;
    s.flags = 3;
    s.px = 120.0; s.py = 143.0; s.pvx = 0.0; s.pvy = 0.0;
    s.pw = 24.0; s.pspeed = 4.0; s.radius = 2.0;
    s.n_bricks = TBX_BRK_COLS * c.n_rows;
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < MAXK; k++) {
        int j = lane + 64 * k;
        if (j < s.n_bricks) bits |= 1u << k;
        if (CUSTOM) {
            BrkCustom& t = d.custom[env];
            if (j < s.n_bricks) {
                int row, col; double x, y;
                brk_canon(j, c.n_rows, row, col, x, y);
                t.x[j] = x; t.y[j] = y; t.w[j] = TBX_BRK_BRICK_W; t.h[j] = TBX_BRK_BRICK_H;
                t.points[j] = c.row_scores[row]; t.depth[j] = c.n_rows - 1 - row;
                t.row[j] = row; t.col[j] = col; t.color[j] = c.row_colors[row]; t.destructible[j] = 1;
            } else {
                t.x[j] = t.y[j] = t.w[j] = t.h[j] = 0.0;
                t.points[j] = t.depth[j] = t.row[j] = t.col[j] = 0; t.color[j] = 0; t.destructible[j] = 0;
            }
        }
    }
    s.mybits = bits;
    s.n_balls = 0;
#pragma unroll
    for (int b = 0; b < MAXB; b++) { s.bx[b] = 0.0; s.by[b] = 0.0; s.bvx[b] = 0.0; s.bvy[b] = 0.0; }
    brk_start_ball(c, s);
}

// What the rasteriser needs of one env, in one 64-byte record (one scalar 64-byte load per frame
// instead of ~30 SoA field reads): written by brk_render_prep_kernel, thread-per-env over the SoA
// state (coalesced), which also does the binary64 -> pixel conversions and clips every rectangle
// to the screen so that it packs into one dword: x0 | x1 << 8 | y0 << 16 | y1 << 24 (x1, y1 exclusive).
struct alignas(64) BrkRenderRec {
    uint64_t alive[MAXK];
    uint32_t paddle;
    uint32_t ball[MAXB];   // empty rect (0) for absent balls
    uint32_t hud;          // 4-bit digits: score 10^4..10^0 (bits 0..19), lives (20..23), level (24..27)
    int32_t n_bricks;
    uint32_t _pad;
};
static_assert(sizeof(BrkRenderRec) == 64, "render record is one 64-byte line");

// 4-bit digits: score 10^4..10^0 (bits 0..19), lives (20..23), level (24..27)
__device__ __forceinline__ uint32_t brk_hud_word(int sc, int lv, int le)
{
    if (sc < 0) sc = 0;
    sc %= 100000;
    lv = lv < 0 ? 0 : lv > 9 ? 9 : lv;
    if (le < 0) le = 0;
    le %= 10;
    uint32_t hud = 0;
    int div = 10000;
#pragma unroll
    for (int g = 0; g < 5; g++) { hud |= (uint32_t)((sc / div) % 10) << (4 * g); div /= 10; }
    return hud | ((uint32_t)lv << 20) | ((uint32_t)le << 24);
}

__device__ __forceinline__ uint32_t pack_rect(int x0, int y0, int w, int h)
{
    int x1 = x0 + w, y1 = y0 + h;
    x0 = min(max(x0, 0), TBX_BRK_W); x1 = min(max(x1, x0), TBX_BRK_W);
    y0 = min(max(y0, 0), TBX_BRK_H); y1 = min(max(y1, y0), TBX_BRK_H);
    return (uint32_t)x0 | ((uint32_t)x1 << 8) | ((uint32_t)y0 << 16) | ((uint32_t)y1 << 24);
}


// ------------------------------------------------------------------ new game

template <bool CUSTOM>
__global__ __launch_bounds__(TBX_BLOCK) void brk_new_game_kernel(BrkDev d, BrkCfg c, const uint8_t* mask)
{
    const int lane = threadIdx.x & 63;
    const int env = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (env >= d.n) return;
    if (mask && !mask[env]) return;
    const size_t N = (size_t)d.n;
    Rng sim;
    sim.s0 = d.sim_rng[env];
    sim.s1 = d.sim_rng[N + env];
    BrkRegs s;
    brk_new_game<CUSTOM>(d, c, env, lane, sim, s);
    brk_store(d, env, lane, s);
    if (lane == 0) {
        d.sim_rng[env] = sim.s0;
        d.sim_rng[N + env] = sim.s1;
        d.prev_score[env] = s.score;
    }
}

// ------------------------------------------------------------------ step

// one frame of one env on one wave
template <bool CUSTOM>
__device__ __forceinline__ void brk_step_body(const BrkDev& d, const BrkCfg& c, const ActionSource& src, uint32_t flags, int env, int lane)
{
    const size_t N = (size_t)d.n;
    if (src.exec_flag && lane == 0) src.exec_flag[env] = tbx_agent_env_finished(src, env) ? 0 : 1;
    if (tbx_agent_env_finished(src, env)) return;     // MaxAndSkipEnv left its loop when this env's game ended

    // ---- action -> buttons
    uint32_t buttons;
    if (src.single_env >= 0) {
        buttons = src.single_buttons;
    } else {
        int a;
        if (src.actions) a = src.actions[env];
        else {
            uint64_t h = tbx_splitmix64(src.seed ^ ((src.env_offset + (uint64_t)env) << 32) ^ src.t);
            a = tbx_legal_action(TBX_GAME_BREAKOUT, (int)(h % 4ull));
        }
        buttons = tbx_ale_buttons(a);
        if (buttons == 0xFFu) {
            buttons = 0;
            if (lane == 0) atomicOr(d.err_flag, 1u);
        }
    }

    BrkRegs s;
    brk_load(d, env, lane, s);

    // ---- lane-owned brick rectangles
    double kx[MAXK], ky[MAXK], kw[MAXK], kh[MAXK];
    const int nk = (s.n_bricks + 63) >> 6;
#pragma unroll
    for (int k = 0; k < MAXK; k++) {
        int j = lane + 64 * k;
        kx[k] = ky[k] = kw[k] = kh[k] = 0.0;
        if (k < nk && j < s.n_bricks) {
            if (CUSTOM) {
                const BrkCustom& t = d.custom[env];
                kx[k] = t.x[j]; ky[k] = t.y[j]; kw[k] = t.w[j]; kh[k] = t.h[j];
            } else {
                int row, col;
                brk_canon(j, c.n_rows, row, col, kx[k], ky[k]);
                kw[k] = TBX_BRK_BRICK_W; kh[k] = TBX_BRK_BRICK_H;
            }
        }
    }

    // 1. paddle intent
    if (buttons & TBX_BTN_LEFT) s.pvx = -s.pspeed;
    else if (buttons & TBX_BTN_RIGHT) s.pvx = s.pspeed;
    else s.pvx = 0.0;
    s.pvy = 0.0;

    // 2. launch
    if ((s.flags & 1) && (buttons & TBX_BTN_BUTTON1)) s.flags = 0;
    const bool launched = !(s.flags & 1);

    // 3. slices
    const double r = s.radius;
    int nsl = 1;
    if (launched && r > 0.0) {
        double vmax = 0.0;
#pragma unroll
        for (int b = 0; b < MAXB; b++)
            if (b < s.n_balls) {
                double m = sqrt(s.bvx[b] * s.bvx[b] + s.bvy[b] * s.bvy[b]);
                if (m > vmax) vmax = m;
            }
        double q = ceil(vmax / r);
        if (q > 16.0) q = 16.0;
        if (q >= 1.0) nsl = (int)q;
    }
    nsl = wave_uniform(nsl);
    const double dt = 1.0 / (double)nsl;
    const double half = s.pw * 0.5;
    bool gone[MAXB] = {false, false, false, false};

    for (int sl = 0; sl < nsl; sl++) {
        s.px = s.px + s.pvx * dt;
        if (s.px - half < TBX_BRK_LEFT) s.px = TBX_BRK_LEFT + half;
        else if (s.px + half > TBX_BRK_RIGHT) s.px = TBX_BRK_RIGHT - half;
        if (!launched) continue;
        const double pl = s.px - half, pr = s.px + half;
#pragma unroll
        for (int b = 0; b < MAXB; b++) {
            if (b >= s.n_balls || gone[b]) continue;
            double x = s.bx[b] + s.bvx[b] * dt;
            double y = s.by[b] + s.bvy[b] * dt;
            double vx = s.bvx[b], vy = s.bvy[b];
            if (x - r < TBX_BRK_LEFT) vx = fabs(vx);
            if (x + r > TBX_BRK_RIGHT) vx = -fabs(vx);
            if (y - r < TBX_BRK_TOP) vy = fabs(vy);
            if (vy > 0.0 && y + r >= s.py && y - r <= s.py + TBX_BRK_PADDLE_H && x + r >= pl && x - r <= pr) {
                const int S = c.segments;
                double t = (x - pl) / s.pw;
                if (t < 0.0) t = 0.0;
                if (t > 1.0) t = 1.0;
                int seg = (int)(t * (double)S);
                if (seg > S - 1) seg = S - 1;
                seg = wave_uniform(seg);
                double sp = sqrt(vx * vx + vy * vy);
                vx = sp * c.pad_dx[seg];
                vy = sp * c.pad_dy[seg];
            }
            // bricks: every lane tests its own; lowest index wins
            int hit = -1;
#pragma unroll
            for (int k = 0; k < MAXK; k++) {
                if (k < nk && hit < 0) {
                    bool o = ((s.mybits >> k) & 1u) && x + r > kx[k] && x - r < kx[k] + kw[k] &&
                             y + r > ky[k] && y - r < ky[k] + kh[k];
                    uint64_t m = __ballot(o);
                    if (m) hit = 64 * k + (int)__builtin_ctzll(m);
                }
            }
            if (hit >= 0) {
                double hx, hy, hw, hh;
                int points, depth, destr;
                if (CUSTOM) {
                    const BrkCustom& t = d.custom[env];
                    hx = t.x[hit]; hy = t.y[hit]; hw = t.w[hit]; hh = t.h[hit];
                    points = t.points[hit]; depth = t.depth[hit]; destr = t.destructible[hit];
                } else {
                    int row, col;
                    brk_canon(hit, c.n_rows, row, col, hx, hy);
                    hw = TBX_BRK_BRICK_W; hh = TBX_BRK_BRICK_H;
                    points = c.row_scores[row]; depth = c.n_rows - 1 - row; destr = 1;
                }
                bool cx_in = (x >= hx && x <= hx + hw);
                bool cy_in = (y >= hy && y <= hy + hh);
                if (cy_in && !cx_in) vx = (x < hx) ? -fabs(vx) : fabs(vx);
                else vy = (y < hy + hh * 0.5) ? -fabs(vy) : fabs(vy);
                if (destr) {
                    if (lane == (hit & 63)) s.mybits &= ~(1u << (hit >> 6));
                    s.score += points;
                }
                if (depth >= c.ball_speed_row_depth) {
                    double m = sqrt(vx * vx + vy * vy);
                    if (m < c.speed_fast && m > 0.0) {
                        double f = c.speed_fast / m;
                        vx = vx * f; vy = vy * f;
                    }
                }
            }
            if (y - r > TBX_BRK_BOTTOM) gone[b] = true;
            s.bx[b] = x; s.by[b] = y; s.bvx[b] = vx; s.bvy[b] = vy;
        }
    }

    // 4. compaction of lost balls, life lost
    if (launched) {
        int kdst = 0;
#pragma unroll
        for (int b = 0; b < MAXB; b++) {
            if (b < s.n_balls && !gone[b]) {
                double x = s.bx[b], y = s.by[b], vx = s.bvx[b], vy = s.bvy[b];
#pragma unroll
                for (int kk = 0; kk < MAXB; kk++)
                    if (kk == kdst) { s.bx[kk] = x; s.by[kk] = y; s.bvx[kk] = vx; s.bvy[kk] = vy; }
                kdst++;
            }
        }
#pragma unroll
        for (int b = 0; b < MAXB; b++)
            if (b >= kdst) { s.bx[b] = 0.0; s.by[b] = 0.0; s.bvx[b] = 0.0; s.bvy[b] = 0.0; }
        s.n_balls = kdst;
        if (kdst == 0) {
            s.lives -= 1;
            s.flags = 3;
            brk_start_ball(c, s);
        }
    }

    // 5. wall cleared -> next level
    {
        int n_d = 0, n_alive = 0;
        uint32_t dbits = 0;
#pragma unroll
        for (int k = 0; k < MAXK; k++) {
            int j = lane + 64 * k;
            bool valid = k < nk && j < s.n_bricks;
            bool destr = valid && (CUSTOM ? d.custom[env].destructible[j] != 0 : true);
            if (destr) dbits |= 1u << k;
            n_d += __popcll(__ballot(destr));
            n_alive += __popcll(__ballot(destr && ((s.mybits >> k) & 1u)));
        }
        if (n_d > 0 && n_alive == 0) {
            s.level += 1;
            s.mybits |= dbits;
        }
    }

    // ---- outputs, auto-reset
    int32_t rew = s.score - d.prev_score[env];
    if (rew < 0) rew = 0;
    const int32_t out_lives = s.lives, out_score = s.score;
    const bool is_done = s.lives <= 0;
    int32_t prev = s.score;
    if (is_done && (flags & TBX_STEP_AUTO_RESET)) {
        Rng sim;
        sim.s0 = d.sim_rng[env];
        sim.s1 = d.sim_rng[N + env];
        brk_new_game<CUSTOM>(d, c, env, lane, sim, s);
        if (lane == 0) { d.sim_rng[env] = sim.s0; d.sim_rng[N + env] = sim.s1; }
        prev = s.score;
    }
    brk_store(d, env, lane, s);
    if (lane == 0) {
        d.prev_score[env] = prev;
        d.reward[env] = rew;
        d.done[env] = is_done ? 1 : 0;
        d.lives_out[env] = out_lives;
        d.score_out[env] = out_score;
        uint32_t lv = out_lives < 0 ? 0u : out_lives > 255 ? 255u : (uint32_t)out_lives;
        d.packed[env] = (uint64_t)(uint32_t)rew | ((uint64_t)(is_done ? 1u : 0u) << 32) | ((uint64_t)lv << 40);
        tbx_accumulate(src, env, rew, is_done);
    }
}


template <bool CUSTOM>
__global__ __launch_bounds__(TBX_BLOCK) void brk_step_kernel(BrkDev d, BrkCfg c, ActionSource src, uint32_t flags, int first_env, int count)
{
    const int lane = threadIdx.x & 63;
    const int rel = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (rel >= count) return;
    brk_step_body<CUSTOM>(d, c, src, flags, first_env + rel, lane);
}

// ------------------------------------------------------------------ step, thread per env (canonical wall)
//
// With the canonical brick wall the transition needs no lane-parallel scan: the bricks a ball can
// touch are a handful of grid cells around it, so one THREAD steps one env and the SoA state is read
// and written perfectly coalesced (64 envs per wave).  Measured on MI355X at 65 536 envs this kernel
// replaces the 117 us wave-per-env step (SURVEY.md section 7 left this choice to measurement); the
// wave-per-env kernel above stays for engines with intervention-written (non-canonical) bricks and for
// single-env calls.  It also emits the rasteriser's 64-byte record, saving the separate prep launch.

struct BrkT {
    Rng rng;
    int32_t score, lives, level, flags;
    double px, py, pvx, pvy, pw, pspeed, radius;
    int32_t n_balls, n_bricks;
    double bx[MAXB], by[MAXB], bvx[MAXB], bvy[MAXB];
    uint64_t alive[MAXK];
};

__device__ __forceinline__ bool t_alive(const BrkT& s, int j)
{
    const uint64_t w = sel4(j >> 6, s.alive[0], s.alive[1], s.alive[2], s.alive[3]);   // values, not lvalues (scratch)
    return (w >> (j & 63)) & 1ull;
}

__device__ __forceinline__ void t_kill(BrkT& s, int j)
{
    const uint64_t m = ~(1ull << (j & 63));
    const int w = j >> 6;
#pragma unroll
    for (int k = 0; k < MAXK; k++) s.alive[k] &= k == w ? m : ~0ull;   // every word written: no select of addresses
}

__device__ __forceinline__ void t_fill_wall(BrkT& s)
{
#pragma unroll
    for (int k = 0; k < MAXK; k++) {
        const int n = s.n_bricks - 64 * k;
        s.alive[k] = n >= 64 ? ~0ull : n > 0 ? ((1ull << n) - 1ull) : 0ull;
    }
}

__device__ __forceinline__ void t_start_ball(const BrkCfg& c, BrkT& s)
{
    const uint64_t i = s.rng.range((uint64_t)c.n_starts);
    const int k = s.n_balls;
    if (k >= MAXB) return;
    const double x = c.start_x[i], y = c.start_y[i];
    const double vx = c.speed_slow * c.start_dx[i], vy = c.speed_slow * c.start_dy[i];
#pragma unroll
    for (int b = 0; b < MAXB; b++)
{   // value selects: an `if (b == k) a[b] = v` chain is folded into a[k] = v, which sends the arrays to scratch
            s.bx[b] = b == k ? x : s.bx[b]; s.by[b] = b == k ? y : s.by[b];
            s.bvx[b] = b == k ? vx : s.bvx[b]; s.bvy[b] = b == k ? vy : s.bvy[b];
        }
    s.n_balls = k + 1;
}

__device__ __forceinline__ void t_new_game(const BrkCfg& c, Rng& sim, BrkT& s)
{
    s.rng = sim.child();
    s.score = 0; s.lives = c.start_lives; s.level = 1; s.flags = 3;
    s.px = 120.0; s.py = 143.0; s.pvx = 0.0; s.pvy = 0.0; s.pw = 24.0; s.pspeed = 4.0; s.radius = 2.0;
    s.n_bricks = TBX_BRK_COLS * c.n_rows;
    t_fill_wall(s);
    s.n_balls = 0;
#pragma unroll
    for (int b = 0; b < MAXB; b++) { s.bx[b] = 0.0; s.by[b] = 0.0; s.bvx[b] = 0.0; s.bvy[b] = 0.0; }
    t_start_ball(c, s);
}

// COH = the step half of an OVERLAPPED fused launch (engine.hip, fused_overlapped): the launch before it may still be running on
// the other stream, on other XCDs with L2s of their own, and the launch after it starts before this one ends.  Every load of
// step-written memory and every store then carries agent scope (global_load / global_store ... sc1: past the XCD's L2 to the
// memory side), so that publishing a block's results needs no L2 write-back and reading the previous step's needs no L2
// invalidate -- one `buffer_wbl2` per step wave cost the 8 192-env launch 7 % and the 65 536-env one 7 % (profiles/r06_experiments.txt).
template <bool COH, class T>
__device__ __forceinline__ T t_ld(const T* p)
{
    if (!COH) return *p;
    if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const uint8_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <bool COH, class T, class V>
__device__ __forceinline__ void t_st(T* p, V value)
{
    const T v = (T)value;
    if (!COH) { *p = v; return; }
    if constexpr (sizeof(T) == 8) __hip_atomic_store(reinterpret_cast<uint64_t*>(p), __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (sizeof(T) == 4) __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(reinterpret_cast<uint8_t*>(p), __builtin_bit_cast(uint8_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool COH = false>
__device__ __forceinline__ void t_load(const BrkDev& d, int env, BrkT& s)
{
    const size_t N = (size_t)d.n;
    s.rng.s0 = t_ld<COH>(&d.rng[env]); s.rng.s1 = t_ld<COH>(&d.rng[N + env]);
    s.score = t_ld<COH>(&d.score[env]); s.lives = t_ld<COH>(&d.lives[env]); s.level = t_ld<COH>(&d.level[env]); s.flags = t_ld<COH>(&d.flags[env]);
    s.px = t_ld<COH>(&d.paddle[0 * N + env]); s.py = t_ld<COH>(&d.paddle[1 * N + env]); s.pvx = t_ld<COH>(&d.paddle[2 * N + env]); s.pvy = t_ld<COH>(&d.paddle[3 * N + env]);
    s.pw = t_ld<COH>(&d.paddle[4 * N + env]); s.pspeed = t_ld<COH>(&d.paddle[5 * N + env]); s.radius = t_ld<COH>(&d.paddle[6 * N + env]);
    s.n_balls = t_ld<COH>(&d.n_balls[env]); s.n_bricks = t_ld<COH>(&d.n_bricks[env]);
#pragma unroll
    for (int b = 0; b < MAXB; b++) {
        s.bx[b] = t_ld<COH>(&d.balls[(size_t)(0 * MAXB + b) * N + env]); s.by[b] = t_ld<COH>(&d.balls[(size_t)(1 * MAXB + b) * N + env]);
        s.bvx[b] = t_ld<COH>(&d.balls[(size_t)(2 * MAXB + b) * N + env]); s.bvy[b] = t_ld<COH>(&d.balls[(size_t)(3 * MAXB + b) * N + env]);
    }
#pragma unroll
    for (int k = 0; k < MAXK; k++) s.alive[k] = t_ld<COH>(&d.alive[(size_t)k * N + env]);
}

template <bool COH = false>
__device__ __forceinline__ void t_store(const BrkDev& d, int env, const BrkT& s)
{
    const size_t N = (size_t)d.n;
    t_st<COH>(&d.rng[env], s.rng.s0); t_st<COH>(&d.rng[N + env], s.rng.s1);
    t_st<COH>(&d.score[env], s.score); t_st<COH>(&d.lives[env], s.lives); t_st<COH>(&d.level[env], s.level); t_st<COH>(&d.flags[env], s.flags);
    t_st<COH>(&d.paddle[0 * N + env], s.px); t_st<COH>(&d.paddle[1 * N + env], s.py); t_st<COH>(&d.paddle[2 * N + env], s.pvx); t_st<COH>(&d.paddle[3 * N + env], s.pvy);
    t_st<COH>(&d.paddle[4 * N + env], s.pw); t_st<COH>(&d.paddle[5 * N + env], s.pspeed); t_st<COH>(&d.paddle[6 * N + env], s.radius);
    t_st<COH>(&d.n_balls[env], s.n_balls); t_st<COH>(&d.n_bricks[env], s.n_bricks);
#pragma unroll
    for (int b = 0; b < MAXB; b++) {
        t_st<COH>(&d.balls[(size_t)(0 * MAXB + b) * N + env], s.bx[b]); t_st<COH>(&d.balls[(size_t)(1 * MAXB + b) * N + env], s.by[b]);
        t_st<COH>(&d.balls[(size_t)(2 * MAXB + b) * N + env], s.bvx[b]); t_st<COH>(&d.balls[(size_t)(3 * MAXB + b) * N + env], s.bvy[b]);
    }
#pragma unroll
    for (int k = 0; k < MAXK; k++) t_st<COH>(&d.alive[(size_t)k * N + env], s.alive[k]);
}

// the rasteriser's record of one env
__device__ __forceinline__ BrkRenderRec t_record(const BrkT& s)
{
    BrkRenderRec rec;
#pragma unroll
    for (int k = 0; k < MAXK; k++) rec.alive[k] = s.alive[k];
    rec.paddle = pack_rect(f2i(s.px - s.pw * 0.5), f2i(s.py), f2i(s.pw), 3);
    const int ball_s = f2i(s.radius * 2.0);
#pragma unroll
    for (int b = 0; b < MAXB; b++)
        rec.ball[b] = b < s.n_balls ? pack_rect(f2i(s.bx[b] - s.radius), f2i(s.by[b] - s.radius), ball_s, ball_s) : 0u;
    rec.n_bricks = s.n_bricks;
    rec.hud = brk_hud_word(s.score, s.lives, s.level);
    rec._pad = 0;
    return rec;
}

// one frame of one env (the transition of SPEC.md "Breakout" for the canonical wall)
__device__ __forceinline__ void brk_t_step(const BrkCfg& c, BrkT& s, uint32_t buttons)
{
    const int rows = c.n_rows;
    // 1. paddle intent
    if (buttons & TBX_BTN_LEFT) s.pvx = -s.pspeed;
    else if (buttons & TBX_BTN_RIGHT) s.pvx = s.pspeed;
    else s.pvx = 0.0;
    s.pvy = 0.0;
    // 2. launch
    if ((s.flags & 1) && (buttons & TBX_BTN_BUTTON1)) s.flags = 0;
    const bool launched = !(s.flags & 1);
    // 3. slices
    const double r = s.radius;
    int nsl = 1;
    if (launched && r > 0.0) {
        double vmax = 0.0;
#pragma unroll
        for (int b = 0; b < MAXB; b++)
            if (b < s.n_balls) {
                const double m = sqrt(s.bvx[b] * s.bvx[b] + s.bvy[b] * s.bvy[b]);
                if (m > vmax) vmax = m;
            }
        double q = ceil(vmax / r);
        if (q > 16.0) q = 16.0;
        if (q >= 1.0) nsl = (int)q;
    }
    const double dt = 1.0 / (double)nsl;
    const double half = s.pw * 0.5;
    bool gone[MAXB] = {false, false, false, false};

    for (int sl = 0; sl < nsl; sl++) {
        s.px = s.px + s.pvx * dt;
        if (s.px - half < TBX_BRK_LEFT) s.px = TBX_BRK_LEFT + half;
        else if (s.px + half > TBX_BRK_RIGHT) s.px = TBX_BRK_RIGHT - half;
        if (!launched) continue;
        const double pl = s.px - half, pr = s.px + half;
#pragma unroll
        for (int b = 0; b < MAXB; b++) {
            if (b >= s.n_balls || gone[b]) continue;
            double x = s.bx[b] + s.bvx[b] * dt;
            double y = s.by[b] + s.bvy[b] * dt;
            double vx = s.bvx[b], vy = s.bvy[b];
            if (x - r < TBX_BRK_LEFT) vx = fabs(vx);
            if (x + r > TBX_BRK_RIGHT) vx = -fabs(vx);
            if (y - r < TBX_BRK_TOP) vy = fabs(vy);
            if (vy > 0.0 && y + r >= s.py && y - r <= s.py + TBX_BRK_PADDLE_H && x + r >= pl && x - r <= pr) {
                const int S = c.segments;
                double t = (x - pl) / s.pw;
                if (t < 0.0) t = 0.0;
                if (t > 1.0) t = 1.0;
                int seg = (int)(t * (double)S);
                if (seg > S - 1) seg = S - 1;
                const double sp = sqrt(vx * vx + vy * vy);
                vx = sp * c.pad_dx[seg];
                vy = sp * c.pad_dy[seg];
            }
            // bricks: only grid cells around the ball can overlap it.  The cell range is estimated with one
            // cell of margin each side (rounding-safe); the exact overlap predicate of the wave kernel decides,
            // scanning in ascending brick index (col-major), so the lowest index wins.
            int hit = -1;
            {
                const double fx0 = floor((x - r - TBX_BRK_LEFT) / TBX_BRK_BRICK_W) - 1.0, fx1 = floor((x + r - TBX_BRK_LEFT) / TBX_BRK_BRICK_W) + 1.0;
                const double fy0 = floor((y - r - TBX_BRK_BRICK_Y0) / TBX_BRK_BRICK_H) - 1.0, fy1 = floor((y + r - TBX_BRK_BRICK_Y0) / TBX_BRK_BRICK_H) + 1.0;
                const int c0 = (int)fmin(fmax(fx0, 0.0), (double)(TBX_BRK_COLS - 1)), c1 = (int)fmin(fmax(fx1, -1.0), (double)(TBX_BRK_COLS - 1));
                const int r0 = (int)fmin(fmax(fy0, 0.0), (double)(rows - 1)), r1 = (int)fmin(fmax(fy1, -1.0), (double)(rows - 1));
                for (int cc = c0; cc <= c1 && hit < 0; cc++) {
                    const double kx = TBX_BRK_LEFT + TBX_BRK_BRICK_W * (double)cc;
                    if (!(x + r > kx && x - r < kx + TBX_BRK_BRICK_W)) continue;
                    for (int rr = r0; rr <= r1; rr++) {
                        const double ky = TBX_BRK_BRICK_Y0 + TBX_BRK_BRICK_H * (double)rr;
                        const int j = cc * rows + rr;
                        if (t_alive(s, j) && y + r > ky && y - r < ky + TBX_BRK_BRICK_H) { hit = j; break; }
                    }
                }
            }
            if (hit >= 0) {
                int row, col;
                double hx, hy;
                brk_canon(hit, rows, row, col, hx, hy);
                const double hw = TBX_BRK_BRICK_W, hh = TBX_BRK_BRICK_H;
                const bool cx_in = (x >= hx && x <= hx + hw);
                const bool cy_in = (y >= hy && y <= hy + hh);
                if (cy_in && !cx_in) vx = (x < hx) ? -fabs(vx) : fabs(vx);
                else vy = (y < hy + hh * 0.5) ? -fabs(vy) : fabs(vy);
                t_kill(s, hit);
                s.score += c.row_scores[row];
                if (rows - 1 - row >= c.ball_speed_row_depth) {
                    const double m = sqrt(vx * vx + vy * vy);
                    if (m < c.speed_fast && m > 0.0) {
                        const double f = c.speed_fast / m;
                        vx = vx * f; vy = vy * f;
                    }
                }
            }
            if (y - r > TBX_BRK_BOTTOM) gone[b] = true;
            s.bx[b] = x; s.by[b] = y; s.bvx[b] = vx; s.bvy[b] = vy;
        }
    }
    // 4. compaction, life lost
    if (launched) {
        int kdst = 0;
#pragma unroll
        for (int b = 0; b < MAXB; b++) {
            if (b < s.n_balls && !gone[b]) {
                const double x = s.bx[b], y = s.by[b], vx = s.bvx[b], vy = s.bvy[b];
#pragma unroll
                for (int kk = 0; kk < MAXB; kk++)
{
                        s.bx[kk] = kk == kdst ? x : s.bx[kk]; s.by[kk] = kk == kdst ? y : s.by[kk];
                        s.bvx[kk] = kk == kdst ? vx : s.bvx[kk]; s.bvy[kk] = kk == kdst ? vy : s.bvy[kk];
                    }
                kdst++;
            }
        }
#pragma unroll
        for (int b = 0; b < MAXB; b++)
            if (b >= kdst) { s.bx[b] = 0.0; s.by[b] = 0.0; s.bvx[b] = 0.0; s.bvy[b] = 0.0; }
        s.n_balls = kdst;
        if (kdst == 0) { s.lives -= 1; s.flags = 3; t_start_ball(c, s); }
    }
    // 5. wall cleared
    if (s.n_bricks > 0 && (s.alive[0] | s.alive[1] | s.alive[2] | s.alive[3]) == 0ull) { s.level += 1; t_fill_wall(s); }

}

// one frame (or the agent layer's whole action repeat) of one env on one THREAD
// AGENT: the agent layer's action repeat with MaxAndSkipEnv's bookkeeping and frame-buffer records; the batch protocol's
// instantiation carries neither (fewer registers, less spill code)
template <bool AGENT, bool COH = false>
__device__ __forceinline__ void brk_step_tpe_body(const BrkDev& d, const BrkCfg& c, const ActionSource& src, uint32_t flags, BrkRenderRec* recs,
                                                  BrkRenderRec* recs_a, BrkRenderRec* recs_b, int env)
{
    static_assert(!(AGENT && COH), "the agent layer's action repeat is never an overlapped launch");
    const size_t N = (size_t)d.n;
    if (AGENT) {
        if (src.exec_flag) src.exec_flag[env] = tbx_agent_env_finished(src, env) ? 0 : 1;
        if (tbx_agent_env_finished(src, env)) return;     // MaxAndSkipEnv left its loop when this env's game ended
    }

    uint32_t buttons;
    if (src.single_env >= 0) buttons = src.single_buttons;
    else {
        int a;
        if (src.actions) a = src.actions[env];
        else {
            const uint64_t h = tbx_splitmix64(src.seed ^ ((src.env_offset + (uint64_t)env) << 32) ^ src.t);
            a = tbx_legal_action(TBX_GAME_BREAKOUT, (int)(h % 4ull));
        }
        buttons = tbx_ale_buttons(a);
        if (buttons == 0xFFu) { buttons = 0; atomicOr(d.err_flag, 1u); }
    }

    BrkT s;
    t_load<COH>(d, env, s);
    int32_t prev = t_ld<COH>(&d.prev_score[env]);
    const int frames = AGENT && src.frames > 1 ? src.frames : 1;
    int32_t rew = 0, out_lives = 0, out_score = 0;
    bool is_done = false;
    for (int fr = 0; fr < frames; fr++) {                  // > 1: the agent layer's action repeat, state stays in registers
        brk_t_step(c, s, buttons);
        rew = s.score - prev;
        if (rew < 0) rew = 0;
        out_lives = s.lives; out_score = s.score;
        is_done = s.lives <= 0;
        prev = s.score;
        if (!AGENT && is_done && (flags & TBX_STEP_AUTO_RESET)) {   // (the agent layer resets through its own procedure)
            Rng sim;
            sim.s0 = t_ld<COH>(&d.sim_rng[env]); sim.s1 = t_ld<COH>(&d.sim_rng[N + env]);
            t_new_game(c, sim, s);
            t_st<COH>(&d.sim_rng[env], sim.s0); t_st<COH>(&d.sim_rng[N + env], sim.s1);
            prev = s.score;
        }
        if (AGENT) {
            tbx_accumulate(src, env, rew, is_done, fr);
            if (src.buf_valid) {                             // MaxAndSkipEnv's frame buffer: slot A after frame skip-2, B after skip-1
                const uint32_t slots = tbx_snap_slots(src, fr);
                if (slots) {
                    const BrkRenderRec rec = t_record(s);
                    if (slots & 1u) recs_a[env] = rec;
                    if (slots & 2u) recs_b[env] = rec;
                    src.buf_valid[env] |= (uint8_t)slots;
                }
                if (is_done) break;                          // ... and its loop ends with the game
            }
        }
    }
    t_store<COH>(d, env, s);
    t_st<COH>(&d.prev_score[env], prev);
    t_st<COH>(&d.reward[env], rew);
    t_st<COH>(&d.done[env], is_done ? 1 : 0);
    t_st<COH>(&d.lives_out[env], out_lives);
    t_st<COH>(&d.score_out[env], out_score);
    const uint32_t lv8 = out_lives < 0 ? 0u : out_lives > 255 ? 255u : (uint32_t)out_lives;
    t_st<COH>(&d.packed[env], (uint64_t)(uint32_t)rew | ((uint64_t)(is_done ? 1u : 0u) << 32) | ((uint64_t)lv8 << 40));
    if (COH) {                                               // the 64-byte record as eight agent-scope stores
        const BrkRenderRec rec = t_record(s);
        static_assert(sizeof(BrkRenderRec) == 64, "render record size");
        uint64_t w[8];
        __builtin_memcpy(w, &rec, sizeof rec);
#pragma unroll
        for (int i = 0; i < 8; i++) t_st<true>(reinterpret_cast<uint64_t*>(&recs[env]) + i, w[i]);
    } else
        recs[env] = t_record(s);
}


// tbx_rollout_synthetic, chunk form: frames t .. t + k - 1 of one env on one thread with the state in registers -- ONE load and ONE
// store of the state per chunk.  Before frame j the rasteriser's record of the state goes to recs[j][env] (frame j of the chunk
// shows the state BEFORE step t + j, as tbx_render_step_synthetic's frame does), after it the step's 8-byte record to
// packed[j * stride + env]; the last frame's outputs land in the engine's output arrays.  Actions by the synthetic rule with
// the frame's own t; auto-reset inside the loop like the single-frame kernel.
__global__ __launch_bounds__(128) void brk_rollout_step_kernel(BrkDev d, const BrkCfg* __restrict__ cp, ActionSource src, uint32_t flags, int k,
                                                               BrkRenderRec* __restrict__ recs, uint64_t* __restrict__ packed, size_t stride)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.n) return;
    const BrkCfg& c = *cp;
    const size_t N = (size_t)d.n;
    BrkT s;
    t_load(d, env, s);
    int32_t prev = d.prev_score[env];
    int32_t rew = 0, out_lives = 0, out_score = 0;
    bool is_done = false;
    for (int j = 0; j < k; j++) {
        recs[(size_t)j * N + env] = t_record(s);
        const uint64_t h = tbx_splitmix64(src.seed ^ ((src.env_offset + (uint64_t)env) << 32) ^ (src.t + (uint64_t)j));
        const uint32_t buttons = tbx_ale_buttons(tbx_legal_action(TBX_GAME_BREAKOUT, (int)(h % 4ull)));
        brk_t_step(c, s, buttons);
        rew = s.score - prev;
        if (rew < 0) rew = 0;
        out_lives = s.lives; out_score = s.score;
        is_done = s.lives <= 0;
        prev = s.score;
        if (is_done && (flags & TBX_STEP_AUTO_RESET)) {
            Rng sim;
            sim.s0 = d.sim_rng[env]; sim.s1 = d.sim_rng[N + env];
            t_new_game(c, sim, s);
            d.sim_rng[env] = sim.s0; d.sim_rng[N + env] = sim.s1;
            prev = s.score;
        }
        const uint32_t lv8 = out_lives < 0 ? 0u : out_lives > 255 ? 255u : (uint32_t)out_lives;
        packed[(size_t)j * stride + env] = (uint64_t)(uint32_t)rew | ((uint64_t)(is_done ? 1u : 0u) << 32) | ((uint64_t)lv8 << 40);
    }
    t_store(d, env, s);
    d.prev_score[env] = prev;
    d.reward[env] = rew;
    d.done[env] = is_done ? 1 : 0;
    d.lives_out[env] = out_lives;
    d.score_out[env] = out_score;
}

template <bool AGENT>
__global__ __launch_bounds__(128) void brk_step_tpe_kernel(BrkDev d, const BrkCfg* __restrict__ cp, ActionSource src, uint32_t flags, BrkRenderRec* recs,
                                                           BrkRenderRec* recs_a, BrkRenderRec* recs_b)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.n) return;
    brk_step_tpe_body<AGENT>(d, *cp, src, flags, recs, recs_a, recs_b, env);   // tables are indexed per thread: read from memory, not from kernel arguments
}

// reset-time wrappers of the agent layer for the envs flagged in r.kind (agent_device.hpp, AgentResetProc)
struct BrkTEnv {
    const BrkCfg& c;
    BrkT& s;
    Rng& sim;
    BrkRenderRec* slot_a;
    BrkRenderRec* slot_b;
    __device__ __forceinline__ void snapshot(int slot) { *(slot ? slot_b : slot_a) = t_record(s); }
    __device__ __forceinline__ void step(uint32_t buttons) { brk_t_step(c, s, buttons); }
    __device__ __forceinline__ void new_game() { t_new_game(c, sim, s); }
    __device__ __forceinline__ int lives() const { return s.lives; }
    __device__ __forceinline__ int score() const { return s.score; }
};

__global__ __launch_bounds__(128) void brk_agent_reset_kernel(BrkDev d, const BrkCfg* __restrict__ cp, AgentResetArgs r, BrkRenderRec* recs,
                                                              BrkRenderRec* recs_a, BrkRenderRec* recs_b)
{
    const BrkCfg& c = *cp;
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.n) return;
    if (r.kind[env] == 0) return;
    const size_t N = (size_t)d.n;
    BrkT s;
    t_load(d, env, s);
    Rng sim;
    sim.s0 = d.sim_rng[env]; sim.s1 = d.sim_rng[N + env];
    AgentMonitor m = agent_monitor_load(r, env);
    BrkTEnv env_ops{c, s, sim, recs_a + env, recs_b + env};
    AgentResetProc<BrkTEnv> proc{env_ops, r, m, r.env_offset + (uint64_t)env, d.prev_score[env], (uint32_t)r.buf_valid[env],
                                 r.noop_override ? r.noop_override[env] : 0, false};
    proc.run();
    t_store(d, env, s);
    d.sim_rng[env] = sim.s0; d.sim_rng[N + env] = sim.s1;
    d.prev_score[env] = proc.prev;
    agent_monitor_store(r, env, m, proc.valid, proc.obs_raw);
    recs[env] = t_record(s);
}

// ------------------------------------------------------------------ render

// 8 scanlines = 5 760 B (RGB) = 45 x 128 B: units whose size is not a multiple of 128 B (5, 10, 20 rows) measured 15-40 %
// slower, 16 rows no better
constexpr int BRK_UNIT_ROWS = 8;    // scanlines per work item, staged in LDS (160 = 20 units)

// the rasteriser's record of one env from the SoA state
__device__ __forceinline__ BrkRenderRec brk_record_of(const BrkDev& d, int env)
{
    const size_t N = (size_t)d.n;
    BrkRenderRec r;
#pragma unroll
    for (int k = 0; k < MAXK; k++) r.alive[k] = d.alive[(size_t)k * N + env];
    const double radius = d.paddle[6 * N + env];
    const double pw = d.paddle[4 * N + env];
    r.paddle = pack_rect(f2i(d.paddle[0 * N + env] - pw * 0.5), f2i(d.paddle[1 * N + env]), f2i(pw), 3);
    const int ball_s = f2i(radius * 2.0);
    const int n_balls = d.n_balls[env];
#pragma unroll
    for (int b = 0; b < MAXB; b++) {
        const int bx = f2i(d.balls[(size_t)(0 * MAXB + b) * N + env] - radius);
        const int by = f2i(d.balls[(size_t)(1 * MAXB + b) * N + env] - radius);
        r.ball[b] = b < n_balls ? pack_rect(bx, by, ball_s, ball_s) : 0u;
    }
    r.n_bricks = d.n_bricks[env];
    r.hud = brk_hud_word(d.score[env], d.lives[env], d.level[env]);
    r._pad = 0;
    return r;
}

// exec_flag (agent layer, single-frame launches): only the envs that ran the frame write their record, and set `bit` of buf_valid
__global__ __launch_bounds__(256) void brk_render_prep_kernel(BrkDev d, BrkRenderRec* recs, int first_env, int count,
                                                              const uint8_t* exec_flag = nullptr, uint8_t* buf_valid = nullptr, int bit = 0)
{
    const int rel = blockIdx.x * blockDim.x + threadIdx.x;
    if (rel >= count) return;
    const int env = first_env + rel;
    if (exec_flag) {
        if (!exec_flag[env]) return;
        buf_valid[env] |= (uint8_t)bit;
    }
    recs[env] = brk_record_of(d, env);
}

// paints the clipped rect `rc` (pack_rect) into this lane's 4 pixels of scanline y
__device__ __forceinline__ void overlay_rect(uint32_t (&px)[4], int x0, int y, uint32_t rc, uint32_t col)
{
    const int ry0 = (int)((rc >> 16) & 255u), ry1 = (int)(rc >> 24);
    if (y >= ry0 && y < ry1) {
        const int rx0 = (int)(rc & 255u), rx1 = (int)((rc >> 8) & 255u);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int x = x0 + i;
            if (x >= rx0 && x < rx1) px[i] = col;
        }
    }
}

__device__ __forceinline__ void overlay4(uint32_t (&px)[4], int x0, int rx0, int rw, uint32_t col)
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = x0 + i;
        if (x >= rx0 && x < rx0 + rw) px[i] = col;
    }
}

// sets bits y0..y1-1 (clipped to 0..159) of a 160-bit row mask
__device__ __forceinline__ void brk_mark_rows(uint64_t (&m)[3], int y0, int y1)
{
#pragma unroll
    for (int w = 0; w < 3; w++) {
        const int lo = max(y0 - 64 * w, 0), hi = min(y1 - 64 * w, 64);
        if (hi > lo) m[w] |= (hi - lo >= 64 ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << lo;
    }
}

__device__ __forceinline__ void brk_overlay_rows(const BrkRenderRec& rec, uint64_t (&m)[3])
{
#pragma unroll
    for (int k = 0; k < 1 + MAXB; k++) {
        const uint32_t rc = k == 0 ? rec.paddle : rec.ball[k - 1];
        if ((rc & 255u) < ((rc >> 8) & 255u)) brk_mark_rows(m, (int)((rc >> 16) & 255u), (int)(rc >> 24));
    }
}

// colours and wall geometry the rasteriser needs from the config
struct BrkPalette {
    uint32_t bg, frame, paddle, ball;
    int32_t rows;
    uint32_t row_colors[TBX_BRK_MAX_ROWS];
};

// The rasteriser.  A wave loads the env's 64-byte record (scalar loads), composes each scanline per lane as 4 packed
// pixels (lane l -> pixels 4l..4l+3) from wave-uniform row classes (HUD / top bar / side walls / brick band / paddle /
// balls), stages BRK_UNIT_ROWS scanlines in its LDS slice and flushes them as 1 KiB-per-instruction stores.  `split`
// waves share a frame: wave `part` takes units part, part + split, ...  RGB launches use split = 10 (two units per
// wave, one from the busy upper half of the screen and one from the lower): 6.05-6.25 TB/s, against 5.4-5.7 TB/s for
// one wave per frame (units then visited in an env-rotated order so that co-resident waves do not march in lockstep)
// and for every other split except 9..12; 4.6 TB/s at one unit per wave.  Other measured alternatives
// (scripts/ubench/): a persistent grid over address-ordered units reaches 6.1 TB/s as bare stores but 4.7-4.9 TB/s with
// record loads and LDS staging; one-shot address-ordered waves of 1, 2, 4 or 10 CONSECUTIVE units 5.3-5.5 TB/s
// (hipMemset on the same boxes: 6.3-6.5 TB/s).
// a render record travels through the rasteriser as ONE VGPR (lane i < 16 holds dword i: a single 64-byte request)
// coherent: the record may have been written by a launch that is still running on another stream (overlapped fused launches) --
// its step blocks fenced it out to memory, and this load goes past whatever older copy an L2 of this XCD still holds
__device__ __forceinline__ uint32_t brk_rec_load_lanes(const BrkRenderRec* __restrict__ r, int lane, bool coherent = false)
{
    if (coherent) return lane < 16 ? __hip_atomic_load(reinterpret_cast<const uint32_t*>(r) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    return lane < 16 ? reinterpret_cast<const uint32_t*>(r)[lane] : 0u;
}
__device__ __forceinline__ BrkRenderRec brk_rec_from_lanes(uint32_t rv)
{
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)__builtin_amdgcn_readlane((int)rv, i);
    BrkRenderRec rec;
#pragma unroll
    for (int i = 0; i < MAXK; i++) rec.alive[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    rec.paddle = w[8];
#pragma unroll
    for (int i = 0; i < MAXB; i++) rec.ball[i] = w[9 + i];
    rec.hud = w[13]; rec.n_bricks = (int32_t)w[14]; rec._pad = w[15];
    return rec;
}
static_assert(MAXK == 4 && MAXB == 4, "record layout: alive[4] u64, paddle, ball[4], hud, n_bricks, pad");

// what a lane needs for every scanline whatever the env: finished palette values, its side-wall pattern, its HUD glyph slots.
// Built by the kernel BEFORE it looks at its work item (so the record's scalar loads overlap with it).
template <int C>
struct BrkLaneTables {
    int x0;
    bool active;
    uint32_t c_bg, c_frame, c_paddle, c_ball;
    uint32_t side[4];
    uint32_t hud_sel[4];
    __device__ __forceinline__ BrkLaneTables(const BrkPalette& pal, int lane)
    {
        constexpr int W = TBX_BRK_W;
        x0 = lane * 4;
        active = x0 < W;
        // palette through pix_of<C>() once; the scanline loop only moves finished pixel values
        c_bg = pix_of<C>(pal.bg); c_frame = pix_of<C>(pal.frame); c_paddle = pix_of<C>(pal.paddle); c_ball = pix_of<C>(pal.ball);
        // per-lane base pattern of a side-wall row
#pragma unroll
        for (int i = 0; i < 4; i++) side[i] = (x0 + i < 12 || x0 + i >= 228) ? c_frame : c_bg;
        // per-lane HUD slots: which glyph (0..6, 7 = none) and which glyph column covers pixel x0+i
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int hud_x0[7] = {36, 44, 52, 60, 68, 148, 196};
            uint32_t sel = 7u << 2;
#pragma unroll
            for (int g = 0; g < 7; g++) {
                const int dx = x0 + i - hud_x0[g];
                if (dx >= 0 && dx < 6) sel = ((uint32_t)g << 2) | (uint32_t)(dx >> 1);
            }
            hud_sel[i] = sel;
        }
    }
};

// One frame's units part, part + split, ... of one env from its record, on one wave (st: the wave's LDS slice): the body of
// brk_render_kernel, also what the resident single-env kernel calls after a step (tbx_serve_loop).  frame_out: the env's frame.
// how brk_paint_units gets at the record: carried as one VGPR and turned into SGPRs per unit (RGB / RGBA launches), or held in
// SGPRs for the whole frame (gray launches, where the former form measured 25 % slower: 0.83 against 0.66 ms at 65 536 envs)
struct BrkRecLanes {
    uint32_t rv;
    __device__ __forceinline__ BrkRenderRec get() const { return brk_rec_from_lanes(rv); }
    // called once before the unit loop: one lane of the record is read there, so the wait for the record's load stands in front
    // of the loop.  Left to the first use inside the loop, the compiler's wait sits at the loop header as `s_waitcnt vmcnt(0)`,
    // and every later unit of the wave then starts by waiting for the previous unit's stores
    __device__ __forceinline__ void arrive() const { const int x = __builtin_amdgcn_readlane((int)rv, 15); asm volatile("" :: "s"(x)); }
};
struct BrkRecHeld { const BrkRenderRec& r; __device__ __forceinline__ const BrkRenderRec& get() const { return r; } __device__ __forceinline__ void arrive() const {} };

template <int C, bool CUSTOM, class RecSrc>
__device__ __forceinline__ void brk_paint_units(const RecSrc src, const BrkCustom* __restrict__ custom, const BrkPalette& pal, const BrkLaneTables<C>& t,
                                                uint8_t* __restrict__ frame_out, int env, int lane, const RowStager<C, TBX_BRK_W, BRK_UNIT_ROWS>& st,
                                                int part, int split)
{
    constexpr int W = TBX_BRK_W, H = TBX_BRK_H;
    constexpr int NUNITS = H / BRK_UNIT_ROWS;
    const int x0 = t.x0;
    const bool active = t.active;
    const int rows = pal.rows;
    const uint32_t c_bg = t.c_bg, c_frame = t.c_frame, c_paddle = t.c_paddle, c_ball = t.c_ball;
    const uint32_t (&side)[4] = t.side;
    const uint32_t (&hud_sel)[4] = t.hud_sel;

    src.arrive();
    for (int q = part; q < NUNITS; q += split) {
        // the record, wave-uniform, for the length of one unit: sixteen v_readlane out of the VGPR that carries it (rv: lane i =
        // dword i) -- sixteen SGPRs that are not held across the frame loop of the kernel
        const BrkRenderRec rec = src.get();
        const int u = (int)(((uint32_t)env * 7u + (uint32_t)q) % (uint32_t)NUNITS);
        uint8_t* dst = frame_out + (size_t)u * BRK_UNIT_ROWS * W * C;
        const int y_first = u * BRK_UNIT_ROWS;

        // HUD column bits of this lane's pixels (only the first units of a frame show the HUD)
        uint32_t hud[4] = {0, 0, 0, 0};
        if (y_first < 12) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t g = hud_sel[i] >> 2;
                const uint32_t digit = (rec.hud >> (4 * g)) & 15u;    // g == 7 reads the unused top nibble
                const uint32_t glyph = g < 7 ? tbx_digit_glyph(digit) : 0u;
                hud[i] = (glyph >> (hud_sel[i] & 3u)) & 0x1249u;      // bit 3*r = lit in glyph row r
            }
        }
        uint32_t brick4[4] = {0, 0, 0, 0};
        int brick_row_cached = -1;
        // scanlines of this unit crossed by the paddle or a ball
        uint64_t ov[3] = {0ull, 0ull, 0ull};
        brk_overlay_rows(rec, ov);
        const uint64_t ov4[4] = {ov[0], ov[1], ov[2], 0ull};
        const uint32_t ov_chunk = row_mask_chunk<BRK_UNIT_ROWS>(ov4, y_first);

#pragma unroll 1
        for (int r = 0; r < BRK_UNIT_ROWS; r++) {
            const int y = y_first + r;
            uint32_t px[4];
            if (y < TBX_BRK_WALL_Y0) {
#pragma unroll
                for (int i = 0; i < 4; i++) px[i] = c_bg;
            } else if (y < TBX_BRK_WALL_Y0 + 12) {
#pragma unroll
                for (int i = 0; i < 4; i++) px[i] = c_frame;
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) px[i] = side[i];
            }
            // bricks
            if (!CUSTOM) {
                const int by = y - 43;
                if (by >= 0 && by < 4 * rows) {
                    const int row = by >> 2;
                    if (row != brick_row_cached) {
                        brick_row_cached = row;
                        const uint32_t rc = pix_of<C>(pal.row_colors[row]);
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            brick4[i] = side[i];
                            const int bxp = x0 + i - 12;
                            if (bxp >= 0 && bxp < 216) {
                                const int j = (bxp / 12) * rows + row;   // < 256 (18*14 = 252)
                                const uint64_t w = sel4(j >> 6, rec.alive[0], rec.alive[1], rec.alive[2], rec.alive[3]);
                                if ((w >> (j & 63)) & 1ull) brick4[i] = rc;
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) px[i] = brick4[i];
                }
            } else {
                const BrkCustom& t = custom[env];
                const int n_bricks = rec.n_bricks;
                const int nk = (n_bricks + 63) >> 6;
                for (int kk = 0; kk < nk; kk++) {
                    const int j = lane + 64 * kk;
                    bool on = false;
                    int rx0 = 0, rw = 0; uint32_t rc = 0;
                    if (j < n_bricks && ((rec.alive[kk] >> lane) & 1ull)) {
                        const int ry0 = f2i(t.y[j]), rh = f2i(t.h[j]);
                        on = y >= ry0 && y < ry0 + rh;
                        rx0 = f2i(t.x[j]); rw = f2i(t.w[j]); rc = pix_of<C>(t.color[j]);
                    }
                    uint64_t m = __ballot(on);
                    while (m) {   // ascending brick index == the oracle's paint order
                        const int src = (int)__builtin_ctzll(m);
                        m &= m - 1;
                        overlay4(px, x0, bcast(rx0, src), bcast(rw, src), bcast(rc, src));
                    }
                }
            }
            // paddle, balls
            if ((ov_chunk >> r) & 1u) {
                overlay_rect(px, x0, y, rec.paddle, c_paddle);
#pragma unroll
                for (int b = 0; b < MAXB; b++) overlay_rect(px, x0, y, rec.ball[b], c_ball);
            }
            // HUD
            if (y >= 2 && y < 12) {
                const int gr = ((y - 2) >> 1) * 3;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if ((hud[i] >> gr) & 1u) px[i] = c_frame;
            }
            if (active) st.put4p(r, lane, px[0], px[1], px[2], px[3]);
        }
        st.flush(dst, lane);
    }
}

template <int C, bool CUSTOM, bool ALT, bool COH = false>
__device__ __forceinline__ void brk_render_body(const BrkRenderRec* __restrict__ recs, const BrkCustom* __restrict__ custom,
                                                               BrkPalette pal, uint8_t* __restrict__ out, int first_env, int count, int split,
                                                               const BrkRenderRec* __restrict__ recs_alt, const uint8_t* __restrict__ pick_alt,
                                                               const int block)
{
    constexpr int W = TBX_BRK_W, H = TBX_BRK_H;
    using Stager = RowStager<C, W, BRK_UNIT_ROWS>;
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[TBX_WAVES_PER_BLOCK * Stager::UNIT_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    Stager st{lds_all + wave * Stager::UNIT_BYTES};
    const BrkLaneTables<C> tables(pal, lane);
    // `split` waves share a frame, wave `part` taking units part, part + split, ...
    const bool stagger = C == 3 && !((split >> 16) & 1);  // (bit 16 of the argument: one of the two parts of a big launch)
    const bool coherent = COH && !OVL_DIAG(split >> 17, 1);   // (COH: an overlapped fused launch -- brk_rec_load_lanes; bit 17: DIAG, plain load)
    split &= 0xFFFF;
    const int wid = wave_uniform(block * TBX_WAVES_PER_BLOCK + wave);
    const int rel = wid / split, part = wid - rel * split;
    if (rel >= count) return;
    // (agent layer, generic path: flagged envs paint recs_alt -- the ARRAY is selected; a select between two loaded records turns
    // scalar loads into vector loads)
    const BrkRenderRec* __restrict__ rsrc = (ALT && pick_alt && wave_uniform((int)pick_alt[first_env + rel])) ? recs_alt : recs;
    uint8_t* frame = out + (size_t)rel * H * W * C;
    if (C == 1) {          // gray: the record by scalar loads, held in SGPRs for the frame
        const BrkRenderRec rec = rsrc[first_env + rel];
        brk_paint_units<C, CUSTOM>(BrkRecHeld{rec}, custom, pal, tables, frame, first_env + rel, lane, st, part, split);
        return;
    }
    // RGB / RGBA: the record arrives as ONE 64-byte vector load (lane i < 16 = dword i) and is turned into SGPRs unit by unit,
    // sixteen v_readlane each time: 53 VGPRs instead of 87 and 19 spilled SGPRs instead of 64 (same-box A/B of the two forms,
    // render-only loops: 1.214-1.222 ms against 1.236-1.244).  The "two rates" of round 2 -- the same loop of [step ; render]
    // at 1.20 ms on one box and 1.35 on the next -- were NOT about this load (a kernel that re-reads every record between the
    // step and the rasteriser changes nothing; writing the records to a buffer the rasteriser does not read removes the slow
    // rate, and so does a rasteriser launch that follows another one): see launch_render (the two-part launch) and raster.hpp.
    const BrkRecLanes rl{brk_rec_load_lanes(&rsrc[first_env + rel], lane, coherent)};
    if (stagger) tbx_stagger_first_waves(wid);          // (raster.hpp: mid-size launches; the record's load is in flight meanwhile)
    brk_paint_units<C, CUSTOM>(rl, custom, pal, tables, frame, first_env + rel, lane, st, part, split);
}

// Two launch forms of the same body.  RGB and RGBA frames stream fastest with exactly FIVE waves per SIMD (measured at 65 536 envs,
// builds with amdgpu_waves_per_eu 3 / 4 / 5 / 6: 1.75 / 1.40 / 1.22 / 1.37 ms -- the record carried as one VGPR leaves room for
// seven, and the launch is slower with more writers in flight, not faster); gray frames, a third of the bytes, take what the
// registers allow (0.66 against 0.85 ms when held to five).
template <int C, bool CUSTOM, bool ALT>
__global__ __launch_bounds__(TBX_BLOCK) __attribute__((amdgpu_waves_per_eu(5, 5))) void brk_render_kernel_w5(
    const BrkRenderRec* __restrict__ recs, const BrkCustom* __restrict__ custom, BrkPalette pal, uint8_t* __restrict__ out, int first_env, int count, int split,
    const BrkRenderRec* __restrict__ recs_alt = nullptr, const uint8_t* __restrict__ pick_alt = nullptr)
{
    brk_render_body<C, CUSTOM, ALT>(recs, custom, pal, out, first_env, count, split, recs_alt, pick_alt, (int)blockIdx.x);
}

// The fused rollout launch (tbx_render_step_synthetic): the rasteriser of frame t and, in the SAME launch, the batch step that
// produces frame t + 1.  The first `step_blocks` blocks step 256 envs each (one thread per env, brk_step_tpe_body) into the OTHER
// records buffer and the step outputs; every later block is a rasteriser block of the launch above, reading the records the
// previous launch's step left.  Nothing the two halves touch overlaps: the painter reads `recs` only, the step reads and writes
// state, outputs and `recs_next`.  Why one launch: at the per-GPU share of a strong-scaled batch (8 192 envs) the step kernel
// is 8 us of pure latency on 128 waves plus a kernel boundary in front of a 150 us rasteriser, and a rasteriser that starts
// behind a short kernel starts in lockstep (raster.hpp) -- here the step's waves hide in the launch's ramp-up, consecutive
// launches follow each other like render-only loops, and neither the staggered first waves nor the two-part launch is needed.
// The kernel is held to five waves per SIMD like the rasteriser it contains, so the step half (190 VGPRs on its own) is
// compiled to the rasteriser's register budget and spills; it runs on 0.6 % of the launch's waves beside the ramp-up.
// OVL: the overlapped form (arrive != nullptr).  Its own instantiation: with the counter, the release block and the coherent record
// load as run-time branches of ONE kernel the stream-order launch of 65 536 envs lost 2.7 % (1.229 against 1.197 ms, same box,
// either order) -- the record's load no longer stayed in flight behind the set-up.
template <int C, bool OVL>
__global__ __launch_bounds__(TBX_BLOCK) __attribute__((amdgpu_waves_per_eu(5, 5))) void brk_render_step_kernel_w5(
    const BrkRenderRec* __restrict__ recs, BrkPalette pal, uint8_t* __restrict__ out, int count, int split, BrkDev d, const BrkCfg* __restrict__ cp,
    ActionSource src, uint32_t flags, BrkRenderRec* __restrict__ recs_next, int step_blocks, unsigned long long* arrive, int release_block, int diag)
{
    if (!OVL) {
        if ((int)blockIdx.x < step_blocks) {
            const int env = (int)blockIdx.x * TBX_BLOCK + (int)threadIdx.x;
            if (env < d.n) brk_step_tpe_body<false, false>(d, *cp, src, flags, recs_next, nullptr, nullptr, env);
            return;
        }
        brk_render_body<C, false, false>(recs, nullptr, pal, out, 0, count, split | (1 << 16), nullptr, nullptr, (int)blockIdx.x - step_blocks);
        return;
    }
    // arrive != nullptr: an overlapped launch (engine.hip, fused_overlapped).  The launch before this one may still be painting on
    // the other lane; what it STEPPED (state, the records read below) was fenced out before its step blocks bumped the counter
    // the engine's wait kernel saw.  The step blocks' waves start by dropping what their caches hold from before that (one
    // invalidate per wave of a few hundred waves; as the first instruction of EVERY wave of the launch it made the launch four times
    // slower, profiles/r06_experiments.txt); the rasteriser blocks read one record each, with a load that bypasses those caches.
    if ((int)blockIdx.x < step_blocks) {
        const int env = (int)blockIdx.x * TBX_BLOCK + (int)threadIdx.x;
        if (OVL_DIAG(diag, 1)) {                                // (DIAG: plain loads and stores behind a per-wave invalidate / write-back)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (env < d.n) brk_step_tpe_body<false, false>(d, *cp, src, flags, recs_next, nullptr, nullptr, env);
            if (!OVL_DIAG(diag, 2)) __threadfence();
        } else {
            if (env < d.n) brk_step_tpe_body<false, true>(d, *cp, src, flags, recs_next, nullptr, nullptr, env);
            // every store above went past the L2 (sc1); once this wave's have been acknowledged they are visible device-wide
            __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0), nothing else (gfx9 encoding)
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(arrive, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // blocks start in index order: once this one runs, the launch has only `lead` blocks left to hand out -- the next launch may come
    if ((int)blockIdx.x == release_block && threadIdx.x == 0) __hip_atomic_fetch_add(arrive + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    brk_render_body<C, false, false, true>(recs, nullptr, pal, out, 0, count, split | (1 << 16) | (OVL_DIAG(diag, 4) ? 1 << 17 : 0), nullptr, nullptr, (int)blockIdx.x - step_blocks);
}

template <int C, bool CUSTOM, bool ALT>
__global__ __launch_bounds__(TBX_BLOCK) void brk_render_kernel(const BrkRenderRec* __restrict__ recs, const BrkCustom* __restrict__ custom, BrkPalette pal,
                                                               uint8_t* __restrict__ out, int first_env, int count, int split,
                                                               const BrkRenderRec* __restrict__ recs_alt = nullptr, const uint8_t* __restrict__ pick_alt = nullptr)
{
    brk_render_body<C, CUSTOM, ALT>(recs, custom, pal, out, first_env, count, split, recs_alt, pick_alt, (int)blockIdx.x);
}

// ------------------------------------------------------------------ resident single-env form (tbx_serve_loop, tbx_common.hpp)
//
// One wave, env 0: steps on request and, when the request asks for it, rasterises the env straight into the engine's mapped
// pinned frame buffer -- the whole of ToyboxBaseEnv.step (apply_ale_action + get_state, envs/atari/base.py:126,109) without a
// launch, a copy or a synchronisation.
template <bool CUSTOM>
__device__ __forceinline__ bool brk_serve_paint(const BrkDev& d, const BrkRenderRec* recs, const BrkPalette& pal, int channels, uint8_t* frame, int lane, uint8_t* lds,
                                                int part, int split)
{
    const BrkRecLanes rec{brk_rec_load_lanes(&recs[0], lane)};
    switch (channels) {
    case 1: brk_paint_units<1, CUSTOM>(rec, d.custom, pal, BrkLaneTables<1>(pal, lane), frame, 0, lane, RowStager<1, TBX_BRK_W, BRK_UNIT_ROWS>{lds}, part, split); break;
    case 3: brk_paint_units<3, CUSTOM>(rec, d.custom, pal, BrkLaneTables<3>(pal, lane), frame, 0, lane, RowStager<3, TBX_BRK_W, BRK_UNIT_ROWS>{lds}, part, split); break;
    default: brk_paint_units<4, CUSTOM>(rec, d.custom, pal, BrkLaneTables<4>(pal, lane), frame, 0, lane, RowStager<4, TBX_BRK_W, BRK_UNIT_ROWS>{lds}, part, split); break;
    }
    return true;
}

template <bool CUSTOM>
__global__ __launch_bounds__(64 * TBX_SERVE_WAVES) void brk_serve_kernel(BrkDev d, BrkCfg c, BrkRenderRec* recs, BrkPalette pal, TbxServeCtl* ctl)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[TBX_SERVE_WAVES][RowStager<4, TBX_BRK_W, BRK_UNIT_ROWS>::UNIT_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    tbx_serve_loop(ctl, lane,
                   [&](const ActionSource& src, uint32_t flags) {
                       brk_step_body<CUSTOM>(d, c, src, flags, 0, lane);
                       __threadfence();
                       if (lane == 0) recs[0] = brk_record_of(d, 0);        // the wave-per-env step leaves no record
                   },
                   [&](int channels, uint8_t* frame, int part, int split) { return brk_serve_paint<CUSTOM>(d, recs, pal, channels, frame, lane, lds[wave], part, split); },
                   d.reward, d.done, d.lives_out, d.score_out, d.err_flag);
}

// thread-per-env step on lane 0 of wave 0 (canonical wall): it writes env 0's record itself
__global__ __launch_bounds__(64 * TBX_SERVE_WAVES) void brk_serve_tpe_kernel(BrkDev d, const BrkCfg* __restrict__ cp, BrkRenderRec* recs, BrkPalette pal, TbxServeCtl* ctl)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[TBX_SERVE_WAVES][RowStager<4, TBX_BRK_W, BRK_UNIT_ROWS>::UNIT_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    tbx_serve_loop(ctl, lane, [&](const ActionSource& src, uint32_t flags) { if (lane == 0) brk_step_tpe_body<false>(d, *cp, src, flags, recs, nullptr, nullptr, 0); },
                   [&](int channels, uint8_t* frame, int part, int split) { return brk_serve_paint<false>(d, recs, pal, channels, frame, lane, lds[wave], part, split); },
                   d.reward, d.done, d.lives_out, d.score_out, d.err_flag);
}

// ------------------------------------------------------------------ fused agent observation
//
// max(frame A, frame B) -> gray -> area warp -> frame stack, straight from the two 64-byte render records: the
// full-resolution frames are never materialised.  One wave per env walks the 160 source scanlines; a scanline is
// composed as one packed-gray dword per lane (4 pixels) for each record, max'd bytewise, staged in LDS and reduced
// with the column taps (agent_device.hpp).  A scanline that neither starts a new row class (host-built boundary
// mask: HUD glyph rows, top bar, side walls, each brick row) nor holds a paddle / ball of either record (per-env
// overlay masks built from the packed rects) equals the previous one, so its horizontal sums are reused: ~130 of
// the 160 lines of a Breakout frame cost a handful of scalar instructions.

struct BrkGrayPal {
    uint32_t bg, frame, paddle, ball;   // gray byte values
    int32_t rows;
    uint32_t row[TBX_BRK_MAX_ROWS];
    uint64_t boundary[3];               // bit y: scanline y may differ from y-1 even without moving objects
};

struct BrkLineCache { int row; uint32_t dw; };

// one scanline of one record as 4 gray bytes
// has_overlay: the paddle or a ball of this record crosses scanline y (bit of brk_overlay_rows)
__device__ __forceinline__ uint32_t brk_gray_line(const BrkRenderRec& rec, const BrkGrayPal& pal, int y, int x0, uint32_t side_dw,
                                                  const uint32_t (&hud)[4], BrkLineCache& bc, bool has_overlay)
{
    uint32_t d;
    if (y < TBX_BRK_WALL_Y0) d = pal.bg * 0x01010101u;
    else if (y < TBX_BRK_WALL_Y0 + 12) d = pal.frame * 0x01010101u;
    else d = side_dw;
    const int by = y - 43;
    if (by >= 0 && by < 4 * pal.rows) {
        const int row = by >> 2;
        if (row != bc.row) {
            bc.row = row;
            uint32_t b = side_dw;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int bxp = x0 + i - 12;
                if (bxp >= 0 && bxp < 216) {
                    const int j = (bxp / 12) * pal.rows + row;
                    const uint64_t w = sel4(j >> 6, rec.alive[0], rec.alive[1], rec.alive[2], rec.alive[3]);
                    if ((w >> (j & 63)) & 1ull) b = (b & ~(0xFFu << (8 * i))) | (pal.row[row] << (8 * i));
                }
            }
            bc.dw = b;
        }
        d = bc.dw;
    }
    // paddle, balls: packed clipped rects of the record
#pragma unroll
    for (int k = 0; k < 1 + MAXB; k++) {
        if (!has_overlay) break;
        const uint32_t rc = k == 0 ? rec.paddle : rec.ball[k - 1];
        const int ry0 = (int)((rc >> 16) & 255u), ry1 = (int)(rc >> 24);
        if (y >= ry0 && y < ry1) {
            const int rx0 = (int)(rc & 255u), rx1 = (int)((rc >> 8) & 255u);
            const uint32_t col = (k == 0 ? pal.paddle : pal.ball) * 0x01010101u;
            uint32_t m = 0;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (x0 + i >= rx0 && x0 + i < rx1) m |= 0xFFu << (8 * i);
            d = (d & ~m) | (col & m);
        }
    }
    if (y >= 2 && y < 12) {
        const int gr = ((y - 2) >> 1) * 3;
        uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if ((hud[i] >> gr) & 1u) m |= 0xFFu << (8 * i);
        d = (d & ~m) | ((pal.frame * 0x01010101u) & m);
    }
    return d;
}

// (five waves per SIMD is what the 29 KB of LDS per block allow; the depth-4 instantiation came out at 97 VGPRs -- four waves -- and ran
// 10 % slower than before the run-based walk; held to five it stays under 96)
template <int S>
__global__ __launch_bounds__(TBX_BLOCK) __attribute__((amdgpu_waves_per_eu(5))) void brk_agent_warp_kernel(const BrkRenderRec* __restrict__ recsLive, const BrkRenderRec* __restrict__ recsA,
                                                                   const BrkRenderRec* __restrict__ recsB, BrkGrayPal pal, AgentWarpArgs a, int n)
{
    constexpr int W = TBX_BRK_W, H = TBX_BRK_H;
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[TBX_WAVES_PER_BLOCK][352];
    __shared__ __attribute__((aligned(16))) uint8_t vals_all[TBX_WAVES_PER_BLOCK][AGENT_MAX_OUT_PX];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int env = wave_uniform(a.first + blockIdx.x * TBX_WAVES_PER_BLOCK + wave);
    if (env >= a.end) return;
    uint8_t* row = lds_all[wave];
    uint8_t* vals = vals_all[wave];
    const ObsSel sel = agent_obs_sel(a, env);
    if (sel.none) {                                        // max over two zero frames
        for (int i = lane; i < a.oh * a.ow; i += 64) vals[i] = 0;
        observation_commit<S>(vals, a, env, lane, sel.zero);
        return;
    }
    const bool fresh = !sel.two;                           // one frame alone: record B is its source, record A is not composed
    // the SOURCE ARRAY is selected (wave-uniform pointer), then one scalar load: a select between loaded records would
    // move them into vector registers
    const BrkRenderRec* __restrict__ srcB = sel.single == 0 ? recsLive : sel.single == 1 ? recsA : recsB;
    const BrkRenderRec recA = recsA[env];
    const BrkRenderRec recB = srcB[env];
    const int x0 = lane * 4;
    const bool active = x0 < W;
    const uint32_t half = (uint32_t)(H * W) / 2u;
    const ColTaps c0 = load_col(a.tx, lane, a.ow), c1 = load_col(a.tx, lane + 64, a.ow);
    const bool on0 = lane < a.ow, on1 = lane + 64 < a.ow;
    if (lane < 8) reinterpret_cast<uint32_t*>(row)[80 + lane] = 0u;
    if (lane >= 60 && lane < 64) reinterpret_cast<uint32_t*>(row)[lane] = 0u;   // words 60..79 are never pixels
    if (lane < 16) reinterpret_cast<uint32_t*>(row)[64 + lane] = 0u;

    uint32_t side_dw = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) side_dw |= ((x0 + i < 12 || x0 + i >= 228) ? pal.frame : pal.bg) << (8 * i);
    // HUD column bits of this lane's pixels for both records: bit 3*r = lit in glyph row r
    uint32_t hudA[4], hudB[4];
    {
        const int hud_x0[7] = {36, 44, 52, 60, 68, 148, 196};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t sel = 7u << 2;
#pragma unroll
            for (int g = 0; g < 7; g++) {
                const int dx = x0 + i - hud_x0[g];
                if (dx >= 0 && dx < 6) sel = ((uint32_t)g << 2) | (uint32_t)(dx >> 1);
            }
            const uint32_t g = sel >> 2;
            const uint32_t ga = g < 7 ? tbx_digit_glyph((recA.hud >> (4 * g)) & 15u) : 0u;
            const uint32_t gb = g < 7 ? tbx_digit_glyph((recB.hud >> (4 * g)) & 15u) : 0u;
            hudA[i] = (ga >> (sel & 3u)) & 0x1249u;
            hudB[i] = (gb >> (sel & 3u)) & 0x1249u;
        }
    }

    BrkLineCache bcA{-1, 0}, bcB{-1, 0};
    uint64_t need[3] = {pal.boundary[0], pal.boundary[1], pal.boundary[2]};   // scanlines that must be composed
    uint64_t ovA[3] = {0, 0, 0}, ovB[3] = {0, 0, 0}, diff_a[3] = {0, 0, 0};
    {
        brk_overlay_rows(recB, ovB);
        if (!fresh) brk_overlay_rows(recA, ovA);
        uint64_t ov[3];
#pragma unroll
        for (int w = 0; w < 3; w++) { ov[w] = ovA[w] | ovB[w]; need[w] |= ov[w]; diff_a[w] = fresh ? 0ull : ov[w]; }
        // frame A can only show other pixels than frame B where a paddle / ball of either frame lies, in the brick band if a
        // brick went, and in the HUD if a digit changed: everywhere else the composed line of B is the max already
        if (!fresh) {
            if (recA.alive[0] != recB.alive[0] || recA.alive[1] != recB.alive[1] || recA.alive[2] != recB.alive[2] ||
                recA.alive[3] != recB.alive[3] || recA.n_bricks != recB.n_bricks)
                brk_mark_rows(diff_a, 43, 43 + 4 * pal.rows);
            if (recA.hud != recB.hud) brk_mark_rows(diff_a, 2, 12);
        }
        // the line after a run of overlay lines differs from it as well
        need[2] |= (ov[2] << 1) | (ov[1] >> 63);
        need[1] |= (ov[1] << 1) | (ov[0] >> 63);
        need[0] |= (ov[0] << 1) | 1ull;
    }
    uint32_t h0 = 0, h1 = 0;
    uint32_t acc[2] = {0, 0};
    // source row sy covers [sy*oh, (sy+1)*oh), output row oy covers [oy*H, (oy+1)*H) in refined units.  Only the COMPOSED
    // scanlines are visited (find-first-set over the need mask; scanline 0 always is one): the run of lines since the previous
    // composed one shares its horizontal sums h, so it adds h x (its overlap with each output row) -- one turn per composed line
    // plus one per finished output row instead of one per source scanline (round 5; the walk over all 160 lines was ~130 turns of
    // bookkeeping and four multiply-adds for lines that change nothing)
    int oy = 0, top = H, pos = 0;
    auto run_to = [&](int end) {                           // the refined interval [pos, end) has the sums h0 / h1   (wave-uniform)
        while (top <= end) {
            const uint32_t w = (uint32_t)(top - pos);      // <= H: 24-bit operands (sums <= 255 W): full-rate v_mad_u32_u24
            const uint32_t s0 = acc[0] + __umul24(w, h0), s1 = acc[1] + __umul24(w, h1);
            if (on0) vals[oy * a.ow + lane] = (uint8_t)(((uint64_t)(s0 + half) * a.magic) >> 42);
            if (on1) vals[oy * a.ow + lane + 64] = (uint8_t)(((uint64_t)(s1 + half) * a.magic) >> 42);
            acc[0] = 0; acc[1] = 0;
            pos = top; top += H; oy += 1;
        }
        const uint32_t w = (uint32_t)(end - pos);
        acc[0] += __umul24(w, h0); acc[1] += __umul24(w, h1);
        pos = end;
    };
#pragma unroll 1
    for (int wi = 0; wi < 3; wi++) {
        uint64_t nw = sel4(wi, need[0], need[1], need[2], 0ull);
        const uint64_t oa = sel4(wi, ovA[0], ovA[1], ovA[2], 0ull), ob = sel4(wi, ovB[0], ovB[1], ovB[2], 0ull);
        const uint64_t da = sel4(wi, diff_a[0], diff_a[1], diff_a[2], 0ull);
        if (wi == 2) nw &= (1ull << (H - 128)) - 1ull;      // (bits past the last scanline: the shifted overlay mask may set one)
#pragma unroll 1
        for (; nw; nw &= nw - 1ull) {
            const int b = (int)__builtin_ctzll(nw), sy = 64 * wi + b;
            run_to(sy * a.oh);
            const uint32_t dB = brk_gray_line(recB, pal, sy, x0, side_dw, hudB, bcB, (ob >> b) & 1ull);
            uint32_t v = dB;
            if ((da >> b) & 1ull) v = bytemax4(brk_gray_line(recA, pal, sy, x0, side_dw, hudA, bcA, (oa >> b) & 1ull), dB);   // diff_a is 0 when fresh
            if (active) reinterpret_cast<uint32_t*>(row)[lane] = v;
            __builtin_amdgcn_wave_barrier();
            h0 = on0 ? hsum(row, c0) : 0u;
            h1 = on1 ? hsum(row, c1) : 0u;
            __builtin_amdgcn_wave_barrier();
        }
    }
    run_to(H * a.oh);
    observation_commit<S>(vals, a, env, lane, sel.zero);
}

// ------------------------------------------------------------------ state pack / unpack, scalars

template <bool CUSTOM>
__global__ void brk_pack_kernel(BrkDev d, BrkCfg c, int env, tbx_breakout_state_t* out)
{
    env += blockIdx.x;      // one block per env of the requested range
    out += blockIdx.x;
    const int lane = threadIdx.x & 63;
    BrkRegs s;
    brk_load(d, env, lane, s);
    if (lane == 0) {
        out->rand[0] = s.rng.s0; out->rand[1] = s.rng.s1;
        out->score = s.score; out->lives = s.lives; out->level = s.level;
        out->is_dead = (s.flags & 1) ? 1 : 0; out->reset = (s.flags & 2) ? 1 : 0;
        out->_pad0[0] = out->_pad0[1] = 0;
        out->paddle_x = s.px; out->paddle_y = s.py; out->paddle_vx = s.pvx; out->paddle_vy = s.pvy;
        out->paddle_width = s.pw; out->paddle_speed = s.pspeed; out->ball_radius = s.radius;
        out->n_balls = s.n_balls; out->n_bricks = s.n_bricks;
        for (int b = 0; b < MAXB; b++) {
            out->ball_x[b] = s.bx[b]; out->ball_y[b] = s.by[b]; out->ball_vx[b] = s.bvx[b]; out->ball_vy[b] = s.bvy[b];
        }
    }
    for (int k = 0; k < MAXK; k++) {
        const int j = lane + 64 * k;
        tbx_brick_t b;
        memset(&b, 0, sizeof b);
        if (j < s.n_bricks) {
            if (CUSTOM) {
                const BrkCustom& t = d.custom[env];
                b.x = t.x[j]; b.y = t.y[j]; b.w = t.w[j]; b.h = t.h[j];
                b.points = t.points[j]; b.depth = t.depth[j]; b.row = t.row[j]; b.col = t.col[j];
                b.color = unpack_color(t.color[j]); b.destructible = t.destructible[j];
            } else {
                int row, col;
                brk_canon(j, c.n_rows, row, col, b.x, b.y);
                b.w = TBX_BRK_BRICK_W; b.h = TBX_BRK_BRICK_H;
                b.points = c.row_scores[row]; b.depth = c.n_rows - 1 - row; b.row = row; b.col = col;
                b.color = unpack_color(c.row_colors[row]); b.destructible = 1;
            }
            b.alive = (s.mybits >> k) & 1u;
        }
        out->bricks[j] = b;
    }
}

template <bool CUSTOM>
__global__ void brk_unpack_kernel(BrkDev d, int env, const tbx_breakout_state_t* in)
{
    env += blockIdx.x;
    in += blockIdx.x;
    const int lane = threadIdx.x & 63;
    BrkRegs s;
    s.rng.s0 = in->rand[0]; s.rng.s1 = in->rand[1];
    s.score = in->score; s.lives = in->lives; s.level = in->level;
    s.flags = (in->is_dead ? 1 : 0) | (in->reset ? 2 : 0);
    s.px = in->paddle_x; s.py = in->paddle_y; s.pvx = in->paddle_vx; s.pvy = in->paddle_vy;
    s.pw = in->paddle_width; s.pspeed = in->paddle_speed; s.radius = in->ball_radius;
    s.n_balls = in->n_balls; s.n_bricks = in->n_bricks;
    for (int b = 0; b < MAXB; b++) {
        const bool v = b < s.n_balls;
        s.bx[b] = v ? in->ball_x[b] : 0.0; s.by[b] = v ? in->ball_y[b] : 0.0;
        s.bvx[b] = v ? in->ball_vx[b] : 0.0; s.bvy[b] = v ? in->ball_vy[b] : 0.0;
    }
    uint32_t bits = 0;
    for (int k = 0; k < MAXK; k++) {
        const int j = lane + 64 * k;
        const tbx_brick_t& b = in->bricks[j];
        if (j < s.n_bricks && b.alive) bits |= 1u << k;
        if (CUSTOM) {
            BrkCustom& t = d.custom[env];
            const bool v = j < s.n_bricks;
            t.x[j] = v ? b.x : 0.0; t.y[j] = v ? b.y : 0.0; t.w[j] = v ? b.w : 0.0; t.h[j] = v ? b.h : 0.0;
            t.points[j] = v ? b.points : 0; t.depth[j] = v ? b.depth : 0; t.row[j] = v ? b.row : 0; t.col[j] = v ? b.col : 0;
            t.color[j] = v ? pack_color(b.color) : 0u; t.destructible[j] = v ? (b.destructible ? 1 : 0) : 0;
        }
    }
    s.mybits = bits;
    brk_store(d, env, lane, s);
}

// fills every env's custom table with the canonical wall (when the engine switches to custom mode)
__global__ __launch_bounds__(TBX_BLOCK) void brk_fill_custom_kernel(BrkDev d, BrkCfg c)
{
    const int lane = threadIdx.x & 63;
    const int env = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (env >= d.n) return;
    const int n_bricks = d.n_bricks[env];
    BrkCustom& t = d.custom[env];
    for (int k = 0; k < MAXK; k++) {
        const int j = lane + 64 * k;
        if (j < n_bricks) {
            int row, col; double x, y;
            brk_canon(j, c.n_rows, row, col, x, y);
            t.x[j] = x; t.y[j] = y; t.w[j] = TBX_BRK_BRICK_W; t.h[j] = TBX_BRK_BRICK_H;
            t.points[j] = c.row_scores[row]; t.depth[j] = c.n_rows - 1 - row;
            t.row[j] = row; t.col[j] = col; t.color[j] = c.row_colors[row]; t.destructible[j] = 1;
        } else {
            t.x[j] = t.y[j] = t.w[j] = t.h[j] = 0.0;
            t.points[j] = t.depth[j] = t.row[j] = t.col[j] = 0; t.color[j] = 0; t.destructible[j] = 0;
        }
    }
}

// ------------------------------------------------------------------ batched interventions (tbx_edit / tbx_reduce)
//
// The helper methods of the reference's BreakoutIntervention (toybox/interventions/breakout.py:303-429) over the whole batch:
// one thread per env on the struct-of-arrays state.  Brick j's column / row are the canonical col = j / rows, row = j % rows,
// or the per-env table's once an intervention has written non-canonical bricks.

template <bool CUSTOM>
__device__ __forceinline__ void brk_brick_rc(const BrkDev& d, int env, int j, int rows, int& row, int& col)
{
    if (CUSTOM) { row = d.custom[env].row[j]; col = d.custom[env].col[j]; }
    else { col = j / rows; row = j - col * rows; }
}

template <bool CUSTOM>
__global__ __launch_bounds__(256) void brk_edit_kernel(BrkDev d, int rows, int op, TbxEditArgs a, const uint8_t* __restrict__ mask)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.n || (mask && !mask[env])) return;
    const size_t N = (size_t)d.n;
    switch (op) {
    case TBX_EDIT_SET_LIVES: d.lives[env] = a.geti(env, 0); break;
    case TBX_EDIT_SET_SCORE: d.score[env] = a.geti(env, 0); break;      // (like tbx_set_state: the next step's reward sees the jump)
    case TBX_EDIT_SET_LEVEL: d.level[env] = a.geti(env, 0); break;
    case TBX_EDIT_BRK_COLUMN_ALIVE: case TBX_EDIT_BRK_ROW_ALIVE: case TBX_EDIT_BRK_ALL_ALIVE: case TBX_EDIT_BRK_BRICK_ALIVE: {
        const int nb = d.n_bricks[env];
        const int key = a.geti(env, 0);
        const bool on = a.geti(env, op == TBX_EDIT_BRK_ALL_ALIVE ? 0 : 1) != 0;
        for (int k = 0; k < MAXK; k++) {
            uint64_t w = d.alive[(size_t)k * N + env], sel = 0;
            for (int b = 0; b < 64; b++) {
                const int j = 64 * k + b;
                if (j >= nb) break;
                int row, col;
                brk_brick_rc<CUSTOM>(d, env, j, rows, row, col);
                const bool hit = op == TBX_EDIT_BRK_ALL_ALIVE || (op == TBX_EDIT_BRK_COLUMN_ALIVE && col == key) ||
                                 (op == TBX_EDIT_BRK_ROW_ALIVE && row == key) || (op == TBX_EDIT_BRK_BRICK_ALIVE && j == key);
                if (hit) sel |= 1ull << b;
            }
            w = on ? (w | sel) : (w & ~sel);
            d.alive[(size_t)k * N + env] = w;
        }
        break;
    }
    case TBX_EDIT_BRK_PADDLE:
        d.paddle[0 * N + env] = a.get(env, 0);
        if (a.n >= 2) d.paddle[1 * N + env] = a.get(env, 1);
        break;
    case TBX_EDIT_BRK_BALL: {
        const int b = a.geti(env, 0);
        if (b >= 0 && b < MAXB && b < d.n_balls[env]) {
            d.balls[(size_t)(0 * MAXB + b) * N + env] = a.get(env, 1); d.balls[(size_t)(1 * MAXB + b) * N + env] = a.get(env, 2);
            d.balls[(size_t)(2 * MAXB + b) * N + env] = a.get(env, 3); d.balls[(size_t)(3 * MAXB + b) * N + env] = a.get(env, 4);
        }
        break;
    }
    default: break;
    }
}

template <bool CUSTOM>
__global__ __launch_bounds__(256) void brk_reduce_kernel(BrkDev d, int rows, int query, TbxEditArgs a, double* __restrict__ out, int width)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.n) return;
    const size_t N = (size_t)d.n;
    double* o = out + (size_t)env * width;
    const int nb = d.n_bricks[env];
    uint64_t al[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; k++) al[k] = d.alive[(size_t)k * N + env];
    auto alive = [&](int j) { return (int)((sel4(j >> 6, al[0], al[1], al[2], al[3]) >> (j & 63)) & 1ull); };
    switch (query) {
    case TBX_QUERY_BRK_BRICKS_REMAINING: {
        int c = 0;
        for (int j = 0; j < nb; j++) c += alive(j);
        o[0] = c;
        break;
    }
    case TBX_QUERY_BRK_NUM_BRICKS: o[0] = nb; break;
    case TBX_QUERY_BRK_COLUMN: case TBX_QUERY_BRK_ROW: {
        const int key = a.geti(env, 0);
        int m = 0;
        for (int j = 0; j < nb && m < width; j++) {
            int row, col;
            brk_brick_rc<CUSTOM>(d, env, j, rows, row, col);
            if ((query == TBX_QUERY_BRK_COLUMN ? col : row) == key) o[m++] = alive(j);
        }
        for (; m < width; m++) o[m] = -1.0;
        break;
    }
    case TBX_QUERY_BRK_IS_CHANNEL: case TBX_QUERY_BRK_CHANNEL_COUNT: case TBX_QUERY_BRK_FIND_CHANNEL: {
        // per column: bricks, alive bricks
        const int ncols = query == TBX_QUERY_BRK_IS_CHANNEL ? 0 : (rows > 0 ? nb / rows : 0);
        const int want = a.geti(env, 0);
        int count = 0, first = -1, is = 0;
        const int c0 = query == TBX_QUERY_BRK_IS_CHANNEL ? want : 0, c1 = query == TBX_QUERY_BRK_IS_CHANNEL ? want + 1 : ncols;
        for (int cc = c0; cc < c1; cc++) {
            int bricks = 0, live = 0;
            for (int j = 0; j < nb; j++) {
                int row, col;
                brk_brick_rc<CUSTOM>(d, env, j, rows, row, col);
                if (col == cc) { bricks++; live += alive(j); }
            }
            if (bricks > 0 && live == 0) { count++; if (first < 0) first = cc; is = 1; }
        }
        o[0] = query == TBX_QUERY_BRK_IS_CHANNEL ? is : query == TBX_QUERY_BRK_CHANNEL_COUNT ? count : first;
        break;
    }
    case TBX_QUERY_BRK_FIND_BRICK: {
        const int want = a.geti(env, 0);
        int first = -1;
        for (int j = 0; j < nb && first < 0; j++)
            if (((a.getu(env, 1 + (j >> 5)) >> (j & 31)) & 1u) && (want < 0 || alive(j) == (want != 0))) first = j;
        o[0] = first;
        break;
    }
    case TBX_QUERY_BRK_PADDLE:
        for (int i = 0; i < 4; i++) o[i] = d.paddle[(size_t)i * N + env];
        break;
    case TBX_QUERY_BRK_BALLS: {
        const int n = d.n_balls[env];
        o[0] = n;
        for (int f = 0; f < 4; f++)
            for (int b = 0; b < MAXB; b++) o[1 + f * MAXB + b] = b < n ? d.balls[(size_t)(f * MAXB + b) * N + env] : -1.0;
        break;
    }
    default: break;
    }
}

__global__ void brk_scalars_kernel(BrkDev d, int32_t* score, int32_t* lives, int32_t* level)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    if (score) score[i] = d.score[i];
    if (lives) lives[i] = d.lives[i];
    if (level) level[i] = d.level[i];
}

// ------------------------------------------------------------------ host ops

struct BreakoutOps : GameOps {
    BrkDev d{};
    BrkCfg c{};
    tbx_breakout_config_t cfg{};
    bool custom = false;
    BrkRenderRec* recs = nullptr;   // [N] rasteriser input records (the CURRENT of two buffers)
    BrkRenderRec* recs_other = nullptr;   // the other one: a step that runs ahead of the previous frame's rasteriser writes
    int recs_par = 0;               // here and the two swap (GameOps::step_ahead)
    BrkRenderRec* recs_third = nullptr;   // fused launches rotate through three (render_step): a launch overlapped on the other lane must not
                                          // rewrite what this launch's rasteriser blocks still read
    BrkRenderRec* recs_chunk[2] = {nullptr, nullptr};   // [k][N] records of a rollout chunk of parity q (tbx_rollout_synthetic), made on first use
    int recs_chunk_k[2] = {0, 0};
    bool recs_valid = false;        // records reflect the current state of every env
    BrkCfg* cfg_dev = nullptr;      // device copy of `c` for kernels that index the tables per thread

    static void default_config(tbx_breakout_config_t* k);

    int height() const override { return TBX_BRK_H; }
    int width() const override { return TBX_BRK_W; }
    size_t state_size() const override { return sizeof(tbx_breakout_state_t); }
    size_t config_size() const override { return sizeof(tbx_breakout_config_t); }

    int load_cfg(tbx_engine* e, const tbx_breakout_config_t& k)
    {
        if (k.n_rows < 1 || k.n_rows > TBX_BRK_MAX_ROWS) return e->fail(TBX_E_UNSUPPORTED, "breakout: n_rows must be 1..14");
        if (k.n_starts < 1 || k.n_starts > TBX_BRK_MAX_STARTS) return e->fail(TBX_E_UNSUPPORTED, "breakout: 1..8 ball_start_positions");
        if (k.paddle_discrete_segments < 1 || k.paddle_discrete_segments > TBX_BRK_MAX_SEGMENTS)
            return e->fail(TBX_E_UNSUPPORTED, "breakout: paddle_discrete_segments must be 1..16 (continuous bounce needs device trig)");
        cfg = k;
        c.start_lives = k.start_lives; c.n_rows = k.n_rows; c.ball_speed_row_depth = k.ball_speed_row_depth;
        c.n_starts = k.n_starts; c.segments = k.paddle_discrete_segments;
        for (int i = 0; i < TBX_BRK_MAX_ROWS; i++) { c.row_scores[i] = k.row_scores[i]; c.row_colors[i] = pack_color(k.row_colors[i]); }
        c.speed_slow = k.ball_speed_slow; c.speed_fast = k.ball_speed_fast;
        for (int i = 0; i < TBX_BRK_MAX_STARTS; i++) {
            c.start_x[i] = k.start_x[i]; c.start_y[i] = k.start_y[i];
            c.start_dx[i] = k.start_dir_x[i]; c.start_dy[i] = k.start_dir_y[i];
        }
        for (int i = 0; i < TBX_BRK_MAX_SEGMENTS; i++) { c.pad_dx[i] = k.paddle_dir_x[i]; c.pad_dy[i] = k.paddle_dir_y[i]; }
        c.bg = pack_color(k.bg_color); c.frame = pack_color(k.frame_color);
        c.paddle = pack_color(k.paddle_color); c.ball = pack_color(k.ball_color);
        if (!cfg_dev) TBX_HIP(hipMalloc((void**)&cfg_dev, sizeof(BrkCfg)));
        TBX_HIP(hipMemcpy(cfg_dev, &c, sizeof(BrkCfg), hipMemcpyHostToDevice));
        return TBX_OK;
    }

    template <typename T>
    static hipError_t dalloc(T** p, size_t count) { return hipMalloc((void**)p, count * sizeof(T)); }

    int init(tbx_engine* e, const void* cfg_pod, size_t cfg_size) override
    {
        tbx_breakout_config_t k;
        if (cfg_pod) {
            if (cfg_size != sizeof k) return e->fail(TBX_E_INVALID, "breakout: config size mismatch");
            memcpy(&k, cfg_pod, sizeof k);
        } else {
            return e->fail(TBX_E_INVALID, "breakout: a config record is required");
        }
        int rc = load_cfg(e, k);
        if (rc) return rc;
        const size_t N = (size_t)e->n;
        options_changed(e);
        d.n = e->n;
        d.sim_rng = e->sim_rng; d.prev_score = e->prev_score; d.reward = e->reward; d.done = e->done;
        d.lives_out = e->lives_out; d.score_out = e->score_out; d.packed = e->packed; d.err_flag = e->err_flag;
        TBX_HIP(dalloc(&d.rng, 2 * N));
        TBX_HIP(dalloc(&d.score, N));
        TBX_HIP(dalloc(&d.lives, N));
        TBX_HIP(dalloc(&d.level, N));
        TBX_HIP(dalloc(&d.flags, N));
        TBX_HIP(dalloc(&d.paddle, 7 * N));
        TBX_HIP(dalloc(&d.n_balls, N));
        TBX_HIP(dalloc(&d.balls, 16 * N));
        TBX_HIP(dalloc(&d.n_bricks, N));
        TBX_HIP(dalloc(&d.alive, 4 * N));
        TBX_HIP(dalloc(&recs, N));
        TBX_HIP(dalloc(&recs_other, N));
        TBX_HIP(dalloc(&recs_third, N));
        d.custom = nullptr;
        return TBX_OK;
    }

    void destroy(tbx_engine*) override
    {
        hipFree(d.rng); hipFree(d.score); hipFree(d.lives); hipFree(d.level); hipFree(d.flags);
        hipFree(d.paddle); hipFree(d.n_balls); hipFree(d.balls); hipFree(d.n_bricks); hipFree(d.alive);
        if (d.custom) hipFree(d.custom);
        hipFree(recs);
        hipFree(recs_other);
        hipFree(recs_third);
        hipFree(recs_chunk[0]); hipFree(recs_chunk[1]);
        hipFree(recsA);
        hipFree(recsB);
        hipFree(cfg_dev);
    }

    int get_config(tbx_engine*, void* pod) override { memcpy(pod, &cfg, sizeof cfg); return TBX_OK; }
    int set_config(tbx_engine* e, const void* pod) override
    {
        tbx_breakout_config_t k;
        memcpy(&k, pod, sizeof k);
        return load_cfg(e, k);
    }

    static dim3 grid_for(int count) { return dim3((count + TBX_WAVES_PER_BLOCK - 1) / TBX_WAVES_PER_BLOCK); }

    int new_game(tbx_engine* e, const uint8_t* mask_dev, hipStream_t s) override
    {
        if (custom) hipLaunchKernelGGL(brk_new_game_kernel<true>, grid_for(e->n), dim3(TBX_BLOCK), 0, s, d, c, mask_dev);
        else hipLaunchKernelGGL(brk_new_game_kernel<false>, grid_for(e->n), dim3(TBX_BLOCK), 0, s, d, c, mask_dev);
        TBX_HIP(hipGetLastError());
        recs_valid = false;
        return TBX_OK;
    }

    int step(tbx_engine* e, const ActionSource& src, uint32_t flags, hipStream_t s) override
    {
        if (!custom && src.single_env < 0 && use_tpe) {
            if (src.acc_reward || src.buf_valid || src.exec_flag || src.frames > 1) {    // an agent step's frames (never auto-reset)
                if (flags & TBX_STEP_AUTO_RESET) return e->fail(TBX_E_INVALID, "an agent step cannot auto-reset");
                hipLaunchKernelGGL(brk_step_tpe_kernel<true>, dim3((e->n + 127) / 128), dim3(128), 0, s, d, cfg_dev, src, flags, recs, recsA, recsB);
            } else
                TBX_LAUNCH_STEP(e, s, (brk_step_tpe_kernel<false>), dim3((e->n + 127) / 128), dim3(128), d, cfg_dev, src, flags, recs, recsA, recsB);
            TBX_HIP(hipGetLastError());
            recs_valid = true;
            return TBX_OK;
        }
        int first = 0, count = e->n;
        if (src.single_env >= 0) { first = src.single_env; count = 1; }
        if (custom) hipLaunchKernelGGL(brk_step_kernel<true>, grid_for(count), dim3(TBX_BLOCK), 0, s, d, c, src, flags, first, count);
        else hipLaunchKernelGGL(brk_step_kernel<false>, grid_for(count), dim3(TBX_BLOCK), 0, s, d, c, src, flags, first, count);
        TBX_HIP(hipGetLastError());
        recs_valid = false;
        return TBX_OK;
    }

    // the rasteriser reads nothing but the records, and there are two buffers of them: a batch step of the canonical wall
    // can run while the previous frame is still being painted (engine.hip, pipelined mode)
    bool pipeline_ok() const override { return !custom && use_tpe && recs_other != nullptr; }
    // scripts/pipeline_sweep.py, stream order against value 3, ms per step without a gather: 1 024 envs 0.0315 / 0.038, 2 048
    // 0.0491 / 0.0507, 4 096 0.0985 / 0.0868, 8 192 0.168 / 0.164, 12 288 0.243 / 0.240; with one at 8 192: 0.175 / 0.230
    int pipeline_auto(int n, bool gather) const override { return (!gather && n >= 4096 && n < 16384) ? 3 : 0; }
    void rebind_outputs(tbx_engine* e) override
    {
        d.reward = e->reward; d.done = e->done; d.lives_out = e->lives_out; d.score_out = e->score_out; d.packed = e->packed;
    }
    void options_changed(tbx_engine* e) override
    {
        use_tpe = e->opt[TBX_OPT_STEP_FORM] != 2;     // thread per env unless the wave-per-env kernel is asked for
        split_opt = e->opt[TBX_OPT_RENDER_SPLIT];
    }
    int records_parity() const override { return recs_par; }
    bool records_valid() const override { return recs_valid; }
    int step_ahead(tbx_engine* e, const ActionSource& src, uint32_t flags, hipStream_t s) override
    {
        hipLaunchKernelGGL(brk_step_tpe_kernel<false>, dim3((e->n + 127) / 128), dim3(128), 0, s, d, cfg_dev, src, flags, recs_other, recsA, recsB);
        TBX_HIP(hipGetLastError());
        std::swap(recs, recs_other);
        recs_par ^= 1;
        recs_valid = true;
        return TBX_OK;
    }

    // tbx_render_step_synthetic: frame t and the step to frame t + 1 in one launch (brk_render_step_kernel_w5)
    bool render_step_fused(int channels) const override { return pipeline_ok() && channels >= 3; }   // (gray frames stream fastest at more than five waves per SIMD)
    // Same-box interleaved A/B of TBX_OPT_FUSED_OVERLAP 2 / 1 (scripts/overlap_diag.py with OD_LIB=product; profiles/r06_experiments.txt),
    // ms per step stream order / overlapped, no gather | K = 4 ring | a collective per step: 4 096 envs 0.0861 / 0.0781 | 0.0880 /
    // 0.0786 | 0.1076 / 0.0798; 8 192: 0.1626 / 0.1571 | 0.1644 / 0.1576 | 0.1840 / 0.1590 (lead 1 024; released as early as possible
    // 0.1640 / 0.1518 with the ring); 16 384: 0.3161 / 0.2979 with the ring; 32 768: 0.6169 / 0.6109 | 0.6194 / 0.5933; 65 536:
    // 1.2244 / 1.2634 | 1.2272 / 1.2734 -- two 1.2 ms launches side by side lose, so the headline batch stays in stream order.
    // The engine's choice by what else is on the device (same files; the ticket is sensitive to when the step blocks get their
    // memory requests through beside the other lane's rasteriser): a collective per step -- which stream order cannot overlap at
    // all -- up to 32 768 envs; a K-step ring up to 4 096 (8 192: 0.1593 / 0.1649 in one run, 0.1644 / 0.1576 in another); no
    // gather up to 8 192 (0.1572 / 0.1479)
    // ... in single-engine probes.  In bench.py's own arms and in processes that hold several engines the same comparisons came out
    // between -7 % and +30 % at 8 192 envs and above (r06_experiments.txt, "what did not reproduce"): the engine's choice is the
    // one case where every run agreed, 4 096 envs and below without a record gather (8-13 % faster)
    bool fused_overlap_auto(int n, int gather_kind) const override { return gather_kind == 0 && n <= 4096; }
    // the next launch is released as soon as this one's step blocks are through (lead = the whole grid): leads of 128 ... 8 192
    // blocks were 1-6 points behind at every size up to 16 384 envs (r06_overlap_lead.txt)
    static constexpr int FUSED_LEAD_BLOCKS = 1 << 20;
    int render_step(tbx_engine* e, uint8_t* out_dev, int channels, const ActionSource& src, uint32_t flags, hipStream_t s, TbxOverlapLaunch* ov) override
    {
        if (!recs_valid) {                                     // the painter reads records: bring them up to the state first
            hipLaunchKernelGGL(brk_render_prep_kernel, dim3((e->n + 255) / 256), dim3(256), 0, s, d, recs, 0, e->n);
            TBX_HIP(hipGetLastError());
            recs_valid = true;
        }
        const BrkPalette pal = palette();
        const int split = split_opt > 0 ? split_opt : channels == 3 ? 10 : e->n <= 8192 ? 4 : e->n <= 32768 ? 2 : 1;
        const int step_blocks = (e->n + TBX_BLOCK - 1) / TBX_BLOCK;
        const dim3 grid(grid_for(e->n * split).x + (unsigned)step_blocks), block(TBX_BLOCK);
        unsigned long long* const no_counter = nullptr;
        if (ov) {                                              // overlapped: the lane's completion event rides on the launch
            ov->step_blocks = step_blocks;
            // the next launch is released when the block `lead` blocks before the end of this grid starts (never a step block).
            // The engine's choice: see fused_overlap_auto
            const int lead = ov->lead > 0 ? ov->lead : FUSED_LEAD_BLOCKS;
            const int release_block = std::max(step_blocks, (int)grid.x - lead);
            hipEvent_t done = OVL_DIAG(ov->diag, 128) ? nullptr : ov->done;
            switch (channels) {
            case 3: hipExtLaunchKernelGGL((brk_render_step_kernel_w5<3, true>), grid, block, 0, s, nullptr, done, 0, recs, pal, out_dev, e->n, split, d, cfg_dev, src, flags, recs_other, step_blocks, ov->arrive, release_block, ov->diag); break;
            case 4: hipExtLaunchKernelGGL((brk_render_step_kernel_w5<4, true>), grid, block, 0, s, nullptr, done, 0, recs, pal, out_dev, e->n, split, d, cfg_dev, src, flags, recs_other, step_blocks, ov->arrive, release_block, ov->diag); break;
            default: return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
            }
        } else
            switch (channels) {
            case 3: TBX_LAUNCH_STEP(e, s, (brk_render_step_kernel_w5<3, false>), grid, block, recs, pal, out_dev, e->n, split, d, cfg_dev, src, flags, recs_other, step_blocks, no_counter, 0, 0); break;
            case 4: TBX_LAUNCH_STEP(e, s, (brk_render_step_kernel_w5<4, false>), grid, block, recs, pal, out_dev, e->n, split, d, cfg_dev, src, flags, recs_other, step_blocks, no_counter, 0, 0); break;
            default: return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
            }
        TBX_HIP(hipGetLastError());
        // THREE buffers in rotation: the launch after this one reads what this one's step wrote and writes the third, so that a
        // launch on the other lane never rewrites records this launch's rasteriser blocks may still be reading
        BrkRenderRec* const was_read = recs;
        recs = recs_other;
        recs_other = recs_third;
        recs_third = was_read;
        recs_par ^= 1;
        return TBX_OK;
    }

    // ---- rollout chunks (engine.hip, rollout_chunked)
    bool rollout_ok(int channels) const override { return pipeline_ok() && channels >= 3; }
    // same-box A/B against the loop of single calls in stream order (scripts/rollout_ab.py, k = 4, ms per step, with the K = 4 record
    // ring | without a gather; profiles/r06_experiments.txt): 4 096 envs 0.0854 -> 0.0747 | 0.0833 -> 0.0748 (0.79 of 8 TB/s); 8 192:
    // 0.1593 -> 0.1559 | 0.1572 -> 0.1536; 16 384: 0.3067 -> 0.3137 | 0.3060 -> 0.3134; 32 768 and 65 536: 4-10 % slower -- two
    // rasteriser launches side by side gain what a launch loses to ramp-up and tail and lose a little everywhere else
    // -- and at 8 192 the sign changes with the BOX and with what else the process has done: an engine that runs nothing but chunks in
    // a process of its own 0.152-0.156 ms on five boxes (0.96-0.98 of linear for the 1/8 batch) and 0.170-0.173 on two (all ten runs
    // of one box), against 0.160-0.161 in stream order everywhere; as the second engine of the process that ran the 65 536-env batch
    // 0.162-0.165 (r06_experiments.txt 1g, 4, 5).  An 8-GPU run is as fast as its slowest rank: 4 096 envs and below.
    // With a record ring on the device the 4 096-env chunks were -13 %, -2 %, +3 % and +11 % on four boxes (without a gather: -10 % on
    // every one).  All of that is the form with a rasteriser launch per frame on two lanes; its spread is a lottery of where the buffers
    // lie (r06_experiments item 6), the form with ONE rasteriser launch per chunk (rollout_render_span) has none and gains 7.5 % at
    // 4 096 envs with the ring, 4 % at 8 192 (6 % with K = 8, 6.5 % with 16), 2 % at 16 384, 1 % at 32 768, 0.3-0.8 % at 65 536 -- and loses
    // 45 % at 1 024 with the ring (165-175 us per chunk where the launches add up to less: the host queues a chunk in 28 us and the clock
    // is the same, but the step launch takes 60-105 us there against 32-48 without the ring and starts 70 us after the one before it
    // ended -- r06_chunk_timelines_3.txt; cause not established; 2 048: -3 %).  The engine's choice:
    // chunks up to 32 768 envs, under a ring from 2 048; per-frame launches only where they never lost (no gather, 4 096 envs and
    // below: -15 % at 1 024, -7 % at 2 048, -10 % at 4 096).
    bool rollout_auto(int n, int gather_kind) const override { return n <= 32768 && (gather_kind == 0 || n >= 2048); }
    int rollout_step(tbx_engine* e, const ActionSource& src, uint32_t flags, int k, int q, uint64_t* packed, size_t stride, hipStream_t s) override
    {
        if (recs_chunk_k[q] < k) {                             // (the caller has made sure nothing reads the old buffer any more)
            TBX_HIP(hipStreamSynchronize(s));
            hipFree(recs_chunk[q]);
            recs_chunk[q] = nullptr;
            recs_chunk_k[q] = 0;
            TBX_HIP(hipMalloc((void**)&recs_chunk[q], sizeof(BrkRenderRec) * (size_t)k * (size_t)e->n));
            recs_chunk_k[q] = k;
        }
        hipLaunchKernelGGL(brk_rollout_step_kernel, dim3((e->n + 127) / 128), dim3(128), 0, s, d, cfg_dev, src, flags, k, recs_chunk[q], packed, stride);
        TBX_HIP(hipGetLastError());
        recs_valid = false;                                    // the single-frame records no longer show the state
        return TBX_OK;
    }
    int rollout_render(tbx_engine* e, uint8_t* out, int channels, int q, int j, hipStream_t s) override
    {
        const BrkRenderRec* rr = recs_chunk[q] + (size_t)j * (size_t)e->n;
        if (channels == 3) launch_render<3>(out, 0, e->n, s, rr);
        else launch_render<4>(out, 0, e->n, s, rr);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    bool rollout_span_ok() const override { return true; }
    // one launch per chunk on one lane / a launch per frame on two (scripts/box_probe.py, k = 4, same process, ms per step; r06_experiments
    // item 6): K = 4 ring 4 096 envs 0.0792 / 0.0817-0.0843, 8 192: 0.1535 / 0.151-0.164 BY PLACEMENT, 16 384: 0.302 / 0.295-0.354; without a
    // gather 4 096: 0.0789 / 0.0748, 8 192: 0.1527 / 0.1533-0.1543
    bool rollout_span_auto(int n, int gather_kind) const override { return gather_kind != 0 || n > 4096; }
    int rollout_render_span(tbx_engine* e, uint8_t* out, int channels, int q, int j0, int count, bool behind_rasteriser, hipStream_t s) override
    {
        // behind another rasteriser launch: one part, no staggered first waves (what those two are for -- first waves that start
        // together into an idle memory system -- does not happen there); 8 192 envs x 4 frames 0.1533-0.1536 ms per step against
        // 0.1553 with the two-part launch, 4 096: 0.0792 / 0.0813, 16 384: 0.3022 / 0.3035 (three processes each, r06_experiments item 6)
        const BrkRenderRec* rr = recs_chunk[q] + (size_t)j0 * (size_t)e->n;
        if (channels == 3) launch_render<3>(out, 0, count * e->n, s, rr, nullptr, nullptr, behind_rasteriser);
        else launch_render<4>(out, 0, count * e->n, s, rr, nullptr, nullptr, behind_rasteriser);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    BrkPalette palette() const
    {
        BrkPalette pal;
        pal.bg = c.bg; pal.frame = c.frame; pal.paddle = c.paddle; pal.ball = c.ball; pal.rows = c.n_rows;
        for (int i = 0; i < TBX_BRK_MAX_ROWS; i++) pal.row_colors[i] = c.row_colors[i];
        return pal;
    }

    bool serve_paints() const override { return true; }
    int serve(tbx_engine* e, TbxServeCtl* ctl_dev, hipStream_t s) override
    {
        const BrkPalette pal = palette();
        if (custom) hipLaunchKernelGGL(brk_serve_kernel<true>, dim3(1), dim3(64 * TBX_SERVE_WAVES), 0, s, d, c, recs, pal, ctl_dev);
        else if (use_tpe) hipLaunchKernelGGL(brk_serve_tpe_kernel, dim3(1), dim3(64 * TBX_SERVE_WAVES), 0, s, d, cfg_dev, recs, pal, ctl_dev);
        else hipLaunchKernelGGL(brk_serve_kernel<false>, dim3(1), dim3(64 * TBX_SERVE_WAVES), 0, s, d, c, recs, pal, ctl_dev);
        TBX_HIP(hipGetLastError());
        recs_valid = false;         // (the thread-per-env form keeps env 0's record current, but nothing here relies on it)
        return TBX_OK;
    }

    bool use_tpe = true;            // thread-per-env step for the canonical wall (TBX_OPT_STEP_FORM = 2 keeps the wave kernel)
    int split_opt = 0;              // TBX_OPT_RENDER_SPLIT

    template <int C>
    void launch_render(uint8_t* out, int first, int count, hipStream_t s, const BrkRenderRec* src_recs = nullptr,
                       const BrkRenderRec* alt = nullptr, const uint8_t* pick_alt = nullptr, bool overlapped = false)
    {
        const BrkPalette pal = palette();
        if (!recs_valid && (!src_recs || alt)) {           // the live records are read
            hipLaunchKernelGGL(brk_render_prep_kernel, dim3((count + 255) / 256), dim3(256), 0, s, d, recs, first, count);
            if (first == 0 && count == d.n) recs_valid = true;
        }
        const BrkRenderRec* rr = src_recs ? src_recs : recs;
        // ten waves per frame, each doing unit p and unit p + 10 (one from the busy upper half of the screen, one from the
        // lower): measured 6.05-6.25 TB/s against 5.4-5.7 for one wave per frame and for every other split from 1 to 20
        // except 9..12 (scripts/ab_render.py over TBX_OPT_RENDER_SPLIT); also what keeps small batches from under-filling the chip
        const int split = split_opt > 0 ? split_opt : C == 3 ? 10 : count <= 8192 ? 4 : count <= 32768 ? 2 : 1;   // gray / RGBA: no such effect
        // A big RGB launch goes out in TWO parts, the first 1 024 envs (two generations of waves) and then the rest.  A launch
        // whose first waves all start together into an idle memory system -- behind a step kernel -- keeps them in lockstep,
        // and the frame stores then cost 0-15 % more depending on where the frame buffer lies (the "two rate states" of rounds
        // 2-3; profiles/HISTORY.md, raster.hpp).  The second part starts against the draining stores of the first and its waves
        // are spread by that, like those of a launch that follows another rasteriser launch.  Measured per output buffer
        // (scripts/ubench/rate_addr, [step ; render], three processes x eight buffers each, ms per step): 65 536 envs
        // 1.208-1.214 in all 24 against 1.211-1.236 with the staggered first waves of raster.hpp and 1.19-1.38 with neither;
        // 32 768: 0.615-0.621 / 0.618-0.692; 16 384: 0.319-0.320 / 0.317-0.345; 8 192: 0.171-0.173 / 0.165-0.181.  First
        // parts of 256 or 512 envs leave some buffers slow, 2 048 costs 3 us more.  Render-only loops pay 8 us per launch for
        // it (1.195 against 1.187 ms).
        // Launches of 16 384 .. 32 767 blocks (6 554 .. 13 107 envs) keep the staggered first waves instead: there the second
        // kernel boundary costs as much as it saves (scripts/pipeline_sweep.py, 8 192 envs: 0.172 / 0.169-0.173 ms per step two
        // parts / stagger, with a per-step gather 0.179 / 0.175).
        // overlapped = true: one part, no stagger -- a launch that starts behind another rasteriser launch on its stream (a rollout chunk's
        // span launch, rollout_render_span).  (Tried for the per-frame rasteriser launches of a chunk on two lanes, which start side by
        // side: 8 192 envs 0.193 against 0.156 ms per step, 16 384: 0.363 against 0.314; those keep the launch forms of the stream-order loop)
        const bool two_parts = !overlapped && C == 3 && grid_for(count * split).x >= 32768u;
        const int split_arg = split | (two_parts || overlapped ? 1 << 16 : 0);          // bit 16: no stagger (this is one of two parts)
        auto launch_part = [&](int f0, int n) {                           // envs first + f0 .. first + f0 + n - 1 into their frames
            uint8_t* o = out + (size_t)f0 * TBX_BRK_H * TBX_BRK_W * C;
            const dim3 grid = grid_for(n * split), block(TBX_BLOCK);
#define BRK_LAUNCH(KERNEL, CUSTOM_, ALT_) hipLaunchKernelGGL((KERNEL<C, CUSTOM_, ALT_>), grid, block, 0, s, rr, d.custom, pal, o, first + f0, n, split_arg, alt, pick_alt)
            if (C >= 3) {          // (pick_alt: the agent layer's generic path, per-env choice between two record arrays)
                if (pick_alt) { if (custom) BRK_LAUNCH(brk_render_kernel_w5, true, true); else BRK_LAUNCH(brk_render_kernel_w5, false, true); }
                else { if (custom) BRK_LAUNCH(brk_render_kernel_w5, true, false); else BRK_LAUNCH(brk_render_kernel_w5, false, false); }
            } else {
                if (pick_alt) { if (custom) BRK_LAUNCH(brk_render_kernel, true, true); else BRK_LAUNCH(brk_render_kernel, false, true); }
                else { if (custom) BRK_LAUNCH(brk_render_kernel, true, false); else BRK_LAUNCH(brk_render_kernel, false, false); }
            }
#undef BRK_LAUNCH
        };
        constexpr int HEAD_ENVS = 1024;
        if (two_parts) { launch_part(0, HEAD_ENVS); launch_part(HEAD_ENVS, count - HEAD_ENVS); }
        else launch_part(0, count);
    }

    int render(tbx_engine* e, uint8_t* out_dev, int channels, int first_env, int n_envs, hipStream_t s) override
    {
        switch (channels) {
        case 1: launch_render<1>(out_dev, first_env, n_envs, s); break;
        case 3: launch_render<3>(out_dev, first_env, n_envs, s); break;
        case 4: launch_render<4>(out_dev, first_env, n_envs, s); break;
        default: return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
        }
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    // ---- agent layer: MaxAndSkipEnv's two-frame buffer is two 64-byte render records per env
    BrkRenderRec* recsA = nullptr;
    BrkRenderRec* recsB = nullptr;

    bool agent_fused() const override { return !custom; }
    bool multi_frame_step() const override { return !custom && use_tpe; }
    bool agent_reset_supported() const override { return !custom; }

    int agent_prepare(tbx_engine* e) override
    {
        if (!recsA) TBX_HIP(hipMalloc((void**)&recsA, sizeof(BrkRenderRec) * (size_t)e->n));
        if (!recsB) TBX_HIP(hipMalloc((void**)&recsB, sizeof(BrkRenderRec) * (size_t)e->n));
        TBX_HIP(hipMemset(recsA, 0, sizeof(BrkRenderRec) * (size_t)e->n));
        TBX_HIP(hipMemset(recsB, 0, sizeof(BrkRenderRec) * (size_t)e->n));
        return TBX_OK;
    }

    int agent_snapshot(tbx_engine* e, int slot, const uint8_t* exec_flag, uint8_t* buf_valid, hipStream_t s) override
    {
        hipLaunchKernelGGL(brk_render_prep_kernel, dim3((e->n + 255) / 256), dim3(256), 0, s, d, slot ? recsB : recsA, 0, e->n, exec_flag,
                           buf_valid, slot ? 2 : 1);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int render_from(tbx_engine* e, int source, const uint8_t* pick_live, uint8_t* out_dev, int channels, hipStream_t s) override
    {
        const BrkRenderRec* src_recs = source == 1 ? recsA : source == 2 ? recsB : nullptr;
        const BrkRenderRec* alt = (src_recs && pick_live) ? recs : nullptr;
        switch (channels) {
        case 1: launch_render<1>(out_dev, 0, e->n, s, src_recs, alt, alt ? pick_live : nullptr); break;
        case 3: launch_render<3>(out_dev, 0, e->n, s, src_recs, alt, alt ? pick_live : nullptr); break;
        case 4: launch_render<4>(out_dev, 0, e->n, s, src_recs, alt, alt ? pick_live : nullptr); break;
        default: return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
        }
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int agent_reset_envs(tbx_engine* e, const AgentResetArgs& r, hipStream_t s) override
    {
        if (custom) return e->fail(TBX_E_UNSUPPORTED, "breakout: episodic-life / fire-reset / no-op-reset need the canonical brick wall");
        hipLaunchKernelGGL(brk_agent_reset_kernel, dim3((e->n + 127) / 128), dim3(128), 0, s, d, cfg_dev, r, recs, recsA, recsB);
        TBX_HIP(hipGetLastError());
        // every other env's live record is still current if it was; the flagged envs' records were just rewritten
        return TBX_OK;
    }

    static uint32_t host_gray(uint32_t rgba)
    {
        const uint32_t r = rgba & 255u, g = (rgba >> 8) & 255u, b = (rgba >> 16) & 255u;
        return (77u * r + 150u * g + 29u * b + 128u) >> 8;
    }

    int agent_warp(tbx_engine* e, const AgentWarpArgs& a, hipStream_t s) override
    {
        BrkGrayPal pal;
        pal.bg = host_gray(c.bg); pal.frame = host_gray(c.frame); pal.paddle = host_gray(c.paddle); pal.ball = host_gray(c.ball);
        pal.rows = c.n_rows;
        for (int i = 0; i < TBX_BRK_MAX_ROWS; i++) pal.row[i] = host_gray(c.row_colors[i]);
        pal.boundary[0] = pal.boundary[1] = pal.boundary[2] = 0;
        auto mark = [&](int y) { if (y >= 0 && y < TBX_BRK_H) pal.boundary[y >> 6] |= 1ull << (y & 63); };
        mark(0);
        for (int y = 2; y <= 12; y += 2) mark(y);           // HUD glyph rows are 2 px tall, HUD ends at 12
        mark(TBX_BRK_WALL_Y0); mark(TBX_BRK_WALL_Y0 + 12);  // top bar
        for (int r = 0; r <= c.n_rows; r++) mark(43 + 4 * r); // each brick row and the line after the wall
        if (!recs_valid) {                                   // envs whose observation is the raw live frame read `recs`
            hipLaunchKernelGGL(brk_render_prep_kernel, dim3((e->n + 255) / 256), dim3(256), 0, s, d, recs, 0, e->n);
            TBX_HIP(hipGetLastError());
            recs_valid = true;
        }
        const dim3 grid = grid_for(a.end - a.first), block(TBX_BLOCK);
        switch (a.obs ? a.stack : 0) {
        case 0: hipLaunchKernelGGL(brk_agent_warp_kernel<0>, grid, block, 0, s, recs, recsA, recsB, pal, a, e->n); break;      // the plane ring (new_plane = 2), any depth
        case 1: hipLaunchKernelGGL(brk_agent_warp_kernel<1>, grid, block, 0, s, recs, recsA, recsB, pal, a, e->n); break;
        case 2: hipLaunchKernelGGL(brk_agent_warp_kernel<2>, grid, block, 0, s, recs, recsA, recsB, pal, a, e->n); break;
        case 3: hipLaunchKernelGGL(brk_agent_warp_kernel<3>, grid, block, 0, s, recs, recsA, recsB, pal, a, e->n); break;
        default: hipLaunchKernelGGL(brk_agent_warp_kernel<4>, grid, block, 0, s, recs, recsA, recsB, pal, a, e->n); break;
        }
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int pack_state(tbx_engine* e, int env, int count, hipStream_t s) override
    {
        auto* out = (tbx_breakout_state_t*)e->staging;
        if (custom) hipLaunchKernelGGL(brk_pack_kernel<true>, dim3(count), dim3(64), 0, s, d, c, env, out);
        else hipLaunchKernelGGL(brk_pack_kernel<false>, dim3(count), dim3(64), 0, s, d, c, env, out);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    bool is_canonical(const tbx_breakout_state_t& st) const
    {
        if (st.n_bricks != TBX_BRK_COLS * cfg.n_rows) return false;
        for (int j = 0; j < st.n_bricks; j++) {
            const tbx_brick_t& b = st.bricks[j];
            const int col = j / cfg.n_rows, row = j % cfg.n_rows;
            if (b.x != TBX_BRK_LEFT + TBX_BRK_BRICK_W * col || b.y != TBX_BRK_BRICK_Y0 + TBX_BRK_BRICK_H * row ||
                b.w != TBX_BRK_BRICK_W || b.h != TBX_BRK_BRICK_H || b.points != cfg.row_scores[row] ||
                b.depth != cfg.n_rows - 1 - row || b.row != row || b.col != col || !b.destructible ||
                pack_color(b.color) != pack_color(cfg.row_colors[row]))
                return false;
        }
        return true;
    }

    int enable_custom(tbx_engine* e, hipStream_t s)
    {
        if (custom) return TBX_OK;
        TBX_HIP(hipMalloc((void**)&d.custom, sizeof(BrkCustom) * (size_t)e->n));
        hipLaunchKernelGGL(brk_fill_custom_kernel, grid_for(e->n), dim3(TBX_BLOCK), 0, s, d, c);
        TBX_HIP(hipGetLastError());
        custom = true;
        return TBX_OK;
    }

    int unpack_state(tbx_engine* e, int env, int count, const void* pod_host, hipStream_t s) override
    {
        const auto* sts = (const tbx_breakout_state_t*)pod_host;
        bool all_canonical = true;
        for (int i = 0; i < count; i++) {
            const auto& st = sts[i];
            if (st.n_balls < 0 || st.n_balls > TBX_BRK_MAX_BALLS)
                return e->fail(TBX_E_UNSUPPORTED, "breakout: the device engine holds at most 4 balls per env");
            if (st.n_bricks < 0 || st.n_bricks > TBX_BRK_MAX_BRICKS)
                return e->fail(TBX_E_UNSUPPORTED, "breakout: the device engine holds at most 256 bricks per env");
            if (!custom && all_canonical && !is_canonical(st)) all_canonical = false;
        }
        if (!custom && !all_canonical) {
            int rc = enable_custom(e, s);
            if (rc) return rc;
        }
        TBX_HIP(hipMemcpyAsync(e->staging, pod_host, sizeof(tbx_breakout_state_t) * (size_t)count, hipMemcpyHostToDevice, s));
        auto* in = (const tbx_breakout_state_t*)e->staging;
        if (custom) hipLaunchKernelGGL(brk_unpack_kernel<true>, dim3(count), dim3(64), 0, s, d, env, in);
        else hipLaunchKernelGGL(brk_unpack_kernel<false>, dim3(count), dim3(64), 0, s, d, env, in);
        TBX_HIP(hipGetLastError());
        recs_valid = false;
        return TBX_OK;
    }

    int edit(tbx_engine* e, int op, const TbxEditArgs& a, const uint8_t* mask_dev, hipStream_t s) override
    {
        switch (op) {
        case TBX_EDIT_SET_LIVES: case TBX_EDIT_SET_SCORE: case TBX_EDIT_SET_LEVEL: case TBX_EDIT_BRK_COLUMN_ALIVE: case TBX_EDIT_BRK_ROW_ALIVE:
        case TBX_EDIT_BRK_ALL_ALIVE: case TBX_EDIT_BRK_BRICK_ALIVE: case TBX_EDIT_BRK_PADDLE: case TBX_EDIT_BRK_BALL: break;
        default: return e->fail(TBX_E_INVALID, "breakout: unknown edit");
        }
        const dim3 grid((e->n + 255) / 256), block(256);
        if (custom) hipLaunchKernelGGL(brk_edit_kernel<true>, grid, block, 0, s, d, c.n_rows, op, a, mask_dev);
        else hipLaunchKernelGGL(brk_edit_kernel<false>, grid, block, 0, s, d, c.n_rows, op, a, mask_dev);
        TBX_HIP(hipGetLastError());
        recs_valid = false;
        return TBX_OK;
    }

    int reduce(tbx_engine* e, int query, const TbxEditArgs& a, double* out_dev, int width, hipStream_t s) override
    {
        const dim3 grid((e->n + 255) / 256), block(256);
        if (custom) hipLaunchKernelGGL(brk_reduce_kernel<true>, grid, block, 0, s, d, c.n_rows, query, a, out_dev, width);
        else hipLaunchKernelGGL(brk_reduce_kernel<false>, grid, block, 0, s, d, c.n_rows, query, a, out_dev, width);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int scalars(tbx_engine* e, int32_t* score_dev, int32_t* lives_dev, int32_t* level_dev, hipStream_t s) override
    {
        hipLaunchKernelGGL(brk_scalars_kernel, dim3((e->n + 255) / 256), dim3(256), 0, s, d, score_dev, lives_dev, level_dev);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }
};

}  // namespace

GameOps* tbx_make_breakout_ops() { return new BreakoutOps(); }
