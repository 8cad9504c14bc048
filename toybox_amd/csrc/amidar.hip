// amidar.hip -- Amidar on gfx950: one 64-lane wavefront per env.
//
// Replaces the per-env Rust transition + rasteriser behind ctoybox.Toybox.apply_ale_action /
// get_state (call sites: /root/reference/toybox/envs/atari/base.py:126,109) for the game the
// reference registers as AmidarToyboxNoFrameskip-v4 (toybox/__init__.py:14-18).  Rules: SPEC.md
// "Amidar"; independently restated in scalar C by the CPU checker under oracle/ and compared bit
// for bit by tests/test_gpu_parity.py.  Integer arithmetic only.
//
// Layout in HBM (env-major tables, so a wave reads its env coalesced):
//   scalars [field][N] int32 (struct-of-arrays over envs)
//   tiles   [N][32] uint64   lane = board row, 2 bits per tile (32 tiles)
//   boxes   [N][2][64] uint32 lane = box (packed corners, flags)
//   movers  [N][NMF][16] int32 lane = mover slot (0..7 enemies, 8 the player), field-major
//   mh      [6][9][N]   int32 mirror of the movers' per-frame fields (x, y, speed, step, caught), struct-of-arrays over envs
//
// The maze chase is control-flow heavy and serial per mover (movement protocols, RNG draws in
// enemy order), so the wave runs the movers one after the other with wave-uniform control flow and
// uses its lanes for the data-parallel parts: the 31 board rows (segment painting, level-complete
// test), the <= 64 boxes (perimeter check), and the scanline pixels in the rasteriser.

#include "tbx_common.hpp"
#include "raster.hpp"
#include "agent_device.hpp"
#include "../../include/toybox_amd_spec.h"

#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>

namespace {

constexpr int BW = TBX_AMI_BOARD_W, BH = TBX_AMI_BOARD_H;
constexpr int PLAYER_SLOT = TBX_AMI_MAX_ENEMIES;   // lane 8

enum AmiField {
    A_SCORE, A_LIVES, A_LEVEL, A_JUMPS, A_JUMP_TIMER, A_CHASE_TIMER, A_N_ENEMIES, A_N_BOXES, A_N_CHASE,
    A_CJ0, A_CJ1, A_CJ2, A_CJ3, A_CJ4, A_CJ5, A_CJ6, A_CJ7, ANF
};
enum MoverField {
    M_X, M_Y, M_SPEED, M_STEP_TX, M_STEP_TY, M_NHIST, M_CAUGHT,
    M_KIND, M_NEXT, M_ROUTE, M_START_TX, M_START_TY, M_VERT, M_HORIZ, M_SVERT, M_SHORIZ, M_SDIR, M_DIR, M_VISION,
    M_SEEN_TX, M_SEEN_TY, M_HIST0, NMF = M_HIST0 + TBX_AMI_MAX_HISTORY
};

// batch-wide tables derived from the config on the host
struct AmiTables {
    uint64_t board_rows[32];
    uint32_t box_geom[TBX_AMI_MAX_BOXES];     // tl_tx | tl_ty << 8 | br_tx << 16 | br_ty << 24
    uint32_t box_flags[TBX_AMI_MAX_BOXES];    // bit0 painted, bit1 triggers_chase, bit2 valid
    int32_t n_boxes, n_chase, chase_j[TBX_AMI_MAX_CHASE_J];
    int32_t n_enemies;
    int32_t ai[TBX_AMI_MAX_ENEMIES][14];      // tbx_amidar_ai_t fields in declaration order
    int32_t player_start_tx, player_start_ty, player_hist0;   // player_hist0 < 0: empty history
    int32_t start_lives, start_jumps, jump_time, chase_time, box_bonus, chase_score_bonus;
    uint32_t bg, player, unpainted, painted, enemy, inner;
};

struct AmiDev {
    int n;
    uint64_t* sim_rng; int32_t* prev_score; int32_t* reward; uint8_t* done; int32_t* lives_out; int32_t* score_out;
    uint64_t* packed; uint32_t* err_flag;
    uint64_t* rng;       // [2][N]
    int32_t* sc;         // [ANF][N]
    uint64_t* tiles;     // [N][32]
    uint32_t* boxes;     // [N][2][64]
    int32_t* movers;     // [N][NMF][16]
    int32_t* mh;         // [NMH][MSLOTS][N] MIRROR of the movers' per-frame fields (position, step, speed, caught) as
                         // struct-of-arrays over envs: the thread-per-env step reads them coalesced from here; every writer
                         // (ami_store, the thread form's ms) updates both copies, wave-per-env kernels read the table
    const AmiTables* tab;
};

// the movers' per-frame ("hot") fields and their row in AmiDev::mh; -1 for the fields that stay in the env-major table
constexpr int NMH = 6, MSLOTS = TBX_AMI_MAX_ENEMIES + 1;
__host__ __device__ constexpr int ami_hot_row(int field)
{
    return field == M_X ? 0 : field == M_Y ? 1 : field == M_SPEED ? 2 : field == M_STEP_TX ? 3 : field == M_STEP_TY ? 4 : field == M_CAUGHT ? 5 : -1;
}

__constant__ int AMI_ROUTES[TBX_AMI_N_ROUTES][TBX_AMI_ROUTE_LEN] = TBX_AMI_ROUTES;

struct AmiRegs {
    Rng rng;
    int32_t f[ANF];
    uint64_t trow;        // lane = board row
    uint32_t bgeom, bflags;   // lane = box
    int32_t mv[NMF];      // lane = mover slot
};

__device__ __forceinline__ void ami_load(const AmiDev& d, int env, int lane, AmiRegs& s)
{
    const size_t N = (size_t)d.n;
    s.rng.s0 = d.rng[env];
    s.rng.s1 = d.rng[N + env];
    // the env's scalars with ONE load instruction (lane i fetches field i) and a v_readlane per field actually used
    static_assert(ANF <= 64, "one lane per scalar field");
    const int32_t fv = lane < ANF ? d.sc[(size_t)lane * N + env] : 0;
#pragma unroll
    for (int i = 0; i < ANF; i++) s.f[i] = __builtin_amdgcn_readlane(fv, i);
    s.trow = d.tiles[(size_t)env * 32 + (lane & 31)];
    s.bgeom = d.boxes[(size_t)env * 128 + lane];
    s.bflags = d.boxes[(size_t)env * 128 + 64 + lane];
    const int32_t* m = d.movers + (size_t)env * NMF * 16;
    const int slot = lane & 15;
#pragma unroll
    for (int i = 0; i < NMF; i++) s.mv[i] = m[i * 16 + slot];
}

__device__ __forceinline__ void ami_store(const AmiDev& d, int env, int lane, const AmiRegs& s)
{
    const size_t N = (size_t)d.n;
    if (lane == 0) {
        d.rng[env] = s.rng.s0;
        d.rng[N + env] = s.rng.s1;
    }
    {   // lane i stores field i: one store instruction for the scalars
        int32_t fv = 0;
#pragma unroll
        for (int i = 0; i < ANF; i++) fv = lane == i ? s.f[i] : fv;
        if (lane < ANF) d.sc[(size_t)lane * N + env] = fv;
    }
    if (lane < 32) d.tiles[(size_t)env * 32 + lane] = s.trow;
    d.boxes[(size_t)env * 128 + lane] = s.bgeom;
    d.boxes[(size_t)env * 128 + 64 + lane] = s.bflags;
    if (lane < 16) {
        int32_t* m = d.movers + (size_t)env * NMF * 16;
#pragma unroll
        for (int i = 0; i < NMF; i++) {
            m[i * 16 + lane] = s.mv[i];
            const int h = ami_hot_row(i);              // ... and the struct-of-arrays mirror of the per-frame fields
            if (h >= 0 && lane < MSLOTS) d.mh[((size_t)h * MSLOTS + lane) * N + env] = s.mv[i];
        }
    }
}

// ------------------------------------------------------------------ board helpers (wave-uniform arguments)

__device__ __forceinline__ uint64_t row_of(const AmiRegs& s, int ty)
{
    const uint32_t lo = bcast((uint32_t)s.trow, ty), hi = bcast((uint32_t)(s.trow >> 32), ty);   // ty is wave-uniform everywhere
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

__device__ __forceinline__ int tile_at(const AmiRegs& s, int tx, int ty)
{
    if (tx < 0 || ty < 0 || tx >= BW || ty >= BH) return TBX_TILE_EMPTY;
    return (int)((row_of(s, ty) >> (2 * tx)) & 3ull);
}

__device__ __forceinline__ bool walkable(const AmiRegs& s, int tx, int ty) { return tile_at(s, tx, ty) != TBX_TILE_EMPTY; }

__device__ __forceinline__ bool is_junction(const AmiRegs& s, int tx, int ty)
{
    if (!walkable(s, tx, ty)) return false;
    const bool h = walkable(s, tx - 1, ty) || walkable(s, tx + 1, ty);
    const bool v = walkable(s, tx, ty - 1) || walkable(s, tx, ty + 1);
    return h && v;
}

__device__ __forceinline__ void dir_delta(int dir, int& dx, int& dy)
{
    dx = dir == TBX_DIR_LEFT ? -1 : dir == TBX_DIR_RIGHT ? 1 : 0;
    dy = dir == TBX_DIR_UP ? -1 : dir == TBX_DIR_DOWN ? 1 : 0;
}

__device__ __forceinline__ bool can_go(const AmiRegs& s, int tx, int ty, int dir)
{
    int dx, dy;
    dir_delta(dir, dx, dy);
    return walkable(s, tx + dx, ty + dy);
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------ mover slots

// slot is wave-uniform (movers are processed one after the other by the whole wave): v_readlane, not ds_bpermute
__device__ __forceinline__ int mget(const AmiRegs& s, int field, int slot) { return __builtin_amdgcn_readlane(s.mv[field], slot); }
__device__ __forceinline__ void mset(AmiRegs& s, int lane, int field, int slot, int v)
{
    if (lane == slot) s.mv[field] = v;
}

__device__ __forceinline__ void reset_mover(AmiRegs& s, int lane, int slot, int tx, int ty)
{
    if (lane == slot) {
        s.mv[M_X] = tx * TBX_AMI_TILE_WX; s.mv[M_Y] = ty * TBX_AMI_TILE_WY;
        s.mv[M_STEP_TX] = -1; s.mv[M_STEP_TY] = -1;
        s.mv[M_NHIST] = 0;
#pragma unroll
        for (int i = 0; i < TBX_AMI_MAX_HISTORY; i++) s.mv[M_HIST0 + i] = 0;
        s.mv[M_CAUGHT] = 0;
    }
}

__device__ __forceinline__ void reset_enemy(AmiRegs& s, int lane, int slot)
{
    const int kind = mget(s, M_KIND, slot);
    int tx, ty;
    if (kind == TBX_AI_LOOKUP) {
        const int r = mget(s, M_ROUTE, slot);
        const int id = (r >= 0 && r < TBX_AMI_N_ROUTES) ? AMI_ROUTES[r][0] : 0;
        tx = id % BW; ty = id / BW;
    } else { tx = mget(s, M_START_TX, slot); ty = mget(s, M_START_TY, slot); }
    if (lane == slot) {
        if (kind == TBX_AI_LOOKUP) s.mv[M_NEXT] = 0;
        s.mv[M_VERT] = s.mv[M_SVERT]; s.mv[M_HORIZ] = s.mv[M_SHORIZ]; s.mv[M_DIR] = s.mv[M_SDIR];
        s.mv[M_SEEN_TX] = -1; s.mv[M_SEEN_TY] = -1;
    }
    reset_mover(s, lane, slot, tx, ty);
}

__device__ __forceinline__ void reset_player(const AmiTables& t, AmiRegs& s, int lane)
{
    reset_mover(s, lane, PLAYER_SLOT, t.player_start_tx, t.player_start_ty);
    // the first junction below the start tile, found on the CURRENT board like the checker does
    int found = -1;
    for (int ty = t.player_start_ty + 1; ty < BH; ty++)
        if (is_junction(s, t.player_start_tx, ty)) { found = ty * BW + t.player_start_tx; break; }
    if (found >= 0 && lane == PLAYER_SLOT) { s.mv[M_HIST0] = found; s.mv[M_NHIST] = 1; }
}

__device__ __forceinline__ void reset_board(const AmiTables& t, AmiRegs& s, int lane)
{
    s.trow = lane < BH ? t.board_rows[lane] : 0ull;
    s.f[A_N_CHASE] = t.n_chase;
#pragma unroll
    for (int k = 0; k < TBX_AMI_MAX_CHASE_J; k++) s.f[A_CJ0 + k] = k < t.n_chase ? t.chase_j[k] : 0;
    s.f[A_N_BOXES] = t.n_boxes;
    s.bgeom = t.box_geom[lane];
    s.bflags = t.box_flags[lane];
}

__device__ __forceinline__ void reset_positions(const AmiTables& t, AmiRegs& s, int lane)
{
    reset_player(t, s, lane);
    for (int i = 0; i < s.f[A_N_ENEMIES]; i++) reset_enemy(s, lane, i);
    s.f[A_JUMP_TIMER] = 0;
    s.f[A_CHASE_TIMER] = 0;
}

__device__ __forceinline__ void ami_new_game(const AmiTables& t, int lane, Rng& sim, AmiRegs& s)
{
    s.rng = sim.child();
#pragma unroll
    for (int i = 0; i < ANF; i++) s.f[i] = 0;
    s.f[A_LIVES] = t.start_lives;
    s.f[A_LEVEL] = 1;
    s.f[A_JUMPS] = t.start_jumps;
    reset_board(t, s, lane);
#pragma unroll
    for (int i = 0; i < NMF; i++) s.mv[i] = 0;
    s.f[A_N_ENEMIES] = t.n_enemies;
    if (lane == PLAYER_SLOT) {
        s.mv[M_SPEED] = TBX_AMI_SPEED; s.mv[M_KIND] = TBX_AI_PLAYER; s.mv[M_SEEN_TX] = -1; s.mv[M_SEEN_TY] = -1;
    }
    if (lane < t.n_enemies) {
        s.mv[M_SPEED] = TBX_AMI_SPEED;
#pragma unroll
        for (int k = 0; k < 14; k++) s.mv[M_KIND + k] = t.ai[lane][k];
    }
    reset_player(t, s, lane);
    for (int i = 0; i < t.n_enemies; i++) reset_enemy(s, lane, i);
}

// ------------------------------------------------------------------ movement

// advance slot toward its step; true when the target tile was reached this frame
__device__ __forceinline__ bool advance(AmiRegs& s, int lane, int slot)
{
    const int stx = mget(s, M_STEP_TX, slot);
    if (stx < 0) return false;
    const int sty = mget(s, M_STEP_TY, slot);
    int x = mget(s, M_X, slot), y = mget(s, M_Y, slot), sp = mget(s, M_SPEED, slot);
    const int gx = stx * TBX_AMI_TILE_WX, gy = sty * TBX_AMI_TILE_WY;
    if (sp < 0) sp = 0;
    if (x < gx) { x += sp; if (x > gx) x = gx; }
    else if (x > gx) { x -= sp; if (x < gx) x = gx; }
    else if (y < gy) { y += sp; if (y > gy) y = gy; }
    else if (y > gy) { y -= sp; if (y < gy) y = gy; }
    mset(s, lane, M_X, slot, x);
    mset(s, lane, M_Y, slot, y);
    if (x == gx && y == gy) {
        mset(s, lane, M_STEP_TX, slot, -1);
        mset(s, lane, M_STEP_TY, slot, -1);
        return true;
    }
    return false;
}

__device__ __forceinline__ void set_step(AmiRegs& s, int lane, int slot, int tx, int ty, int dir)
{
    int dx, dy;
    dir_delta(dir, dx, dy);
    mset(s, lane, M_STEP_TX, slot, tx + dx);
    mset(s, lane, M_STEP_TY, slot, ty + dy);
}

__device__ __forceinline__ void push_history(AmiRegs& s, int lane, int slot, int id)
{
    int n = mget(s, M_NHIST, slot);
    if (n >= TBX_AMI_MAX_HISTORY) {
        if (lane == slot) {
#pragma unroll
            for (int i = 1; i < TBX_AMI_MAX_HISTORY; i++) s.mv[M_HIST0 + i - 1] = s.mv[M_HIST0 + i];
        }
        n = TBX_AMI_MAX_HISTORY - 1;
    }
    if (lane == slot) {
#pragma unroll
        for (int i = 0; i < TBX_AMI_MAX_HISTORY; i++)     // selects of values: an `if (i == n) a[i] = id` chain is folded
            s.mv[M_HIST0 + i] = i == n ? id : s.mv[M_HIST0 + i];   // into a[n] = id, which sends the array to scratch
        s.mv[M_NHIST] = n + 1;
    }
}

__device__ __forceinline__ int last_history(const AmiRegs& s, int slot, int n)
{
    int v = 0;
#pragma unroll
    for (int i = 0; i < TBX_AMI_MAX_HISTORY; i++) {
        const int h = mget(s, M_HIST0 + i, slot);
        if (i == n - 1) v = h;
    }
    return v;
}

// lane = box: perimeter test against the rows broadcast one by one
__device__ __forceinline__ void check_boxes(const AmiTables& t, AmiRegs& s, int lane)
{
    const int tl_tx = s.bgeom & 255, tl_ty = (s.bgeom >> 8) & 255, br_tx = (s.bgeom >> 16) & 255, br_ty = (s.bgeom >> 24) & 255;
    const bool cand = lane < s.f[A_N_BOXES] && !(s.bflags & 1u);
    bool ok = cand && tl_tx < BW && br_tx < BW && tl_tx <= br_tx;
    // 2-bit fields tl_tx..br_tx all == PAINTED (binary 10)
    uint64_t span = 0, want = 0;
    if (ok) {
        const int nb = 2 * (br_tx - tl_tx + 1);
        span = (nb >= 64 ? ~0ull : ((1ull << nb) - 1ull)) << (2 * tl_tx);
        want = 0xAAAAAAAAAAAAAAAAull & span;
    }
    for (int y = 0; y < BH; y++) {
        const uint64_t row = row_of(s, y);
        if (ok && y >= tl_ty && y <= br_ty) {
            if (y == tl_ty || y == br_ty) { if ((row & span) != want) ok = false; }
            else if (((row >> (2 * tl_tx)) & 3ull) != TBX_TILE_PAINTED || ((row >> (2 * br_tx)) & 3ull) != TBX_TILE_PAINTED) ok = false;
        }
    }
    if (ok && (tl_ty >= BH || br_ty >= BH || tl_ty > br_ty)) ok = false;
    const uint64_t newly = __ballot(ok);
    if (!newly) return;
    if (ok) s.bflags |= 1u;
    s.f[A_SCORE] += t.box_bonus * __popcll(newly);
    const bool trig = lane < s.f[A_N_BOXES] && (s.bflags & 2u);
    if (__ballot(ok && trig) && !__ballot(trig && !(s.bflags & 1u))) s.f[A_CHASE_TIMER] = t.chase_time;
}

__device__ __forceinline__ void player_arrived(const AmiTables& t, AmiRegs& s, int lane)
{
    const int px = mget(s, M_X, PLAYER_SLOT), py = mget(s, M_Y, PLAYER_SLOT);
    const int tx = px / TBX_AMI_TILE_WX, ty = py / TBX_AMI_TILE_WY;
    if (!is_junction(s, tx, ty)) return;
    const int id = ty * BW + tx;
    int newly = 0;
    const int nh = mget(s, M_NHIST, PLAYER_SLOT);
    if (nh > 0) {
        const int prev = last_history(s, PLAYER_SLOT, nh);
        const int qx = prev % BW, qy = prev / BW;
        if (prev != id && prev >= 0 && prev < BW * BH && (qx == tx || qy == ty)) {
            const int x0 = qx < tx ? qx : tx, x1 = qx < tx ? tx : qx, y0 = qy < ty ? qy : ty, y1 = qy < ty ? ty : qy;
            const int nb = 2 * (x1 - x0 + 1);
            const uint64_t span = (nb >= 64 ? ~0ull : ((1ull << nb) - 1ull)) << (2 * x0);
            const bool mine = lane >= y0 && lane <= y1;
            // a tile is walkable when its 2-bit tag is non-zero
            const uint64_t nz = (s.trow | (s.trow >> 1)) & 0x5555555555555555ull;
            const bool blocked = mine && (nz & span) != (0x5555555555555555ull & span);
            if (!__ballot(blocked)) {
                // count tiles not yet PAINTED (tag != 10b), then paint
                const uint64_t painted = (s.trow >> 1) & ~s.trow & 0x5555555555555555ull;
                const int cnt = mine ? __popcll((0x5555555555555555ull & span) & ~painted) : 0;
                newly = wave_sum(cnt);
                if (mine) s.trow = (s.trow & ~span) | (0xAAAAAAAAAAAAAAAAull & span);
            }
        }
    }
    push_history(s, lane, PLAYER_SLOT, id);
    if (newly > 0) {
        s.f[A_SCORE] += newly;
        check_boxes(t, s, lane);
        const bool left = lane < BH && (s.trow & 0x5555555555555555ull) != 0;   // tags 01 / 11 remain
        if (!__ballot(left)) {
            s.f[A_LEVEL] += 1;
            reset_board(t, s, lane);
            reset_positions(t, s, lane);
            s.f[A_JUMPS] = t.start_jumps;
        }
    }
}

__device__ __forceinline__ int first_open(const AmiRegs& s, int tx, int ty, int avoid)
{
    for (int dd = 0; dd < 4; dd++)
        if (dd != avoid && can_go(s, tx, ty, dd)) return dd;
    return (avoid >= 0 && can_go(s, tx, ty, avoid)) ? avoid : -1;
}

__device__ __forceinline__ void enemy_decide(AmiRegs& s, int lane, int slot)
{
    const int x = mget(s, M_X, slot), y = mget(s, M_Y, slot);
    const int tx = x / TBX_AMI_TILE_WX, ty = y / TBX_AMI_TILE_WY;
    const int kind = mget(s, M_KIND, slot);
    int dir = -1;
    if (kind == TBX_AI_LOOKUP) {
        const int r = mget(s, M_ROUTE, slot);
        if (r < 0 || r >= TBX_AMI_N_ROUTES) return;
        int len = 0;
        while (len < TBX_AMI_ROUTE_LEN && AMI_ROUTES[r][len] >= 0) len++;
        int next = mget(s, M_NEXT, slot);
        if (next < 0 || next >= len) next = 0;
        if (AMI_ROUTES[r][next] == ty * BW + tx) next = (next + 1) % len;
        mset(s, lane, M_NEXT, slot, next);
        const int gx = AMI_ROUTES[r][next] % BW, gy = AMI_ROUTES[r][next] / BW;
        if (gx > tx && can_go(s, tx, ty, TBX_DIR_RIGHT)) dir = TBX_DIR_RIGHT;
        else if (gx < tx && can_go(s, tx, ty, TBX_DIR_LEFT)) dir = TBX_DIR_LEFT;
        else if (gy > ty && can_go(s, tx, ty, TBX_DIR_DOWN)) dir = TBX_DIR_DOWN;
        else if (gy < ty && can_go(s, tx, ty, TBX_DIR_UP)) dir = TBX_DIR_UP;
    } else if (kind == TBX_AI_PERIMETER) {
        if (ty == 0 && tx < BW - 1 && can_go(s, tx, ty, TBX_DIR_RIGHT)) dir = TBX_DIR_RIGHT;
        else if (tx == BW - 1 && ty < BH - 1 && can_go(s, tx, ty, TBX_DIR_DOWN)) dir = TBX_DIR_DOWN;
        else if (ty == BH - 1 && tx > 0 && can_go(s, tx, ty, TBX_DIR_LEFT)) dir = TBX_DIR_LEFT;
        else if (tx == 0 && ty > 0 && can_go(s, tx, ty, TBX_DIR_UP)) dir = TBX_DIR_UP;
        else {
            const int order[4] = {TBX_DIR_UP, TBX_DIR_LEFT, TBX_DIR_DOWN, TBX_DIR_RIGHT};
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (dir < 0 && can_go(s, tx, ty, order[k])) dir = order[k];
        }
    } else if (kind == TBX_AI_AMIDAR) {
        int vert = mget(s, M_VERT, slot) & 1;
        int horiz = 2 | (mget(s, M_HORIZ, slot) & 1);
        bool came_vertically = true;
        const int nh = mget(s, M_NHIST, slot);
        if (nh > 0) came_vertically = (last_history(s, slot, nh) % BW) == tx;
        const bool can_h = can_go(s, tx, ty, horiz), can_v = can_go(s, tx, ty, vert);
        if (came_vertically && can_h) dir = horiz;
        else if (can_v) dir = vert;
        else if (can_h) dir = horiz;
        else {
            vert ^= 1; horiz ^= 1;
            if (can_go(s, tx, ty, vert)) dir = vert;
            else if (can_go(s, tx, ty, horiz)) dir = horiz;
        }
        mset(s, lane, M_VERT, slot, vert);
        mset(s, lane, M_HORIZ, slot, horiz);
    } else if (kind == TBX_AI_TARGET_PLAYER) {
        const int ptx = mget(s, M_X, PLAYER_SLOT) / TBX_AMI_TILE_WX, pty = mget(s, M_Y, PLAYER_SLOT) / TBX_AMI_TILE_WY;
        const int ddx = ptx - tx, ddy = pty - ty;
        const int adx = ddx < 0 ? -ddx : ddx, ady = ddy < 0 ? -ddy : ddy;
        int cur = mget(s, M_DIR, slot) & 3;
        if (adx + ady <= mget(s, M_VISION, slot)) {
            mset(s, lane, M_SEEN_TX, slot, ptx);
            mset(s, lane, M_SEEN_TY, slot, pty);
            const int hd = ddx > 0 ? TBX_DIR_RIGHT : TBX_DIR_LEFT, vd = ddy > 0 ? TBX_DIR_DOWN : TBX_DIR_UP;
            const int first = adx >= ady ? hd : vd, second = adx >= ady ? vd : hd;
            const int fz = adx >= ady ? adx : ady, sz = adx >= ady ? ady : adx;
            if (fz > 0 && can_go(s, tx, ty, first)) dir = first;
            else if (sz > 0 && can_go(s, tx, ty, second)) dir = second;
        } else {
            mset(s, lane, M_SEEN_TX, slot, -1);
            mset(s, lane, M_SEEN_TY, slot, -1);
        }
        if (dir < 0) dir = can_go(s, tx, ty, cur) ? cur : first_open(s, tx, ty, cur ^ 1);
        if (dir >= 0) cur = dir;
        mset(s, lane, M_DIR, slot, cur);
    } else if (kind == TBX_AI_RANDOM) {
        int cur = mget(s, M_DIR, slot) & 3;
        int opts[4], n = 0;
#pragma unroll
        for (int dd = 0; dd < 4; dd++)
            if (dd != (cur ^ 1) && can_go(s, tx, ty, dd)) {
#pragma unroll
                for (int k = 0; k < 4; k++) opts[k] = k == n ? dd : opts[k];   // value selects (see push_history)
                n++;
            }
        if (n == 0) dir = can_go(s, tx, ty, cur ^ 1) ? (cur ^ 1) : -1;
        else {
            const int k = (int)s.rng.range((uint64_t)n);
            dir = (int)sel4(k, (uint64_t)opts[0], (uint64_t)opts[1], (uint64_t)opts[2], (uint64_t)opts[3]);
        }
        if (dir >= 0) cur = dir;
        mset(s, lane, M_DIR, slot, cur);
    } else {
        return;
    }
    if (dir >= 0) set_step(s, lane, slot, tx, ty, dir);
}

__device__ __forceinline__ bool at_tile(const AmiRegs& s, int slot)
{
    return mget(s, M_X, slot) % TBX_AMI_TILE_WX == 0 && mget(s, M_Y, slot) % TBX_AMI_TILE_WY == 0;
}

__device__ __forceinline__ void ami_step(const AmiTables& t, int lane, uint32_t buttons, AmiRegs& s)
{
    int32_t* f = s.f;
    // 1. timers
    if (f[A_JUMP_TIMER] > 0) f[A_JUMP_TIMER] -= 1;
    if (f[A_CHASE_TIMER] > 0) {
        f[A_CHASE_TIMER] -= 1;
        if (f[A_CHASE_TIMER] == 0)
            for (int i = 0; i < f[A_N_ENEMIES]; i++)
                if (mget(s, M_CAUGHT, i)) reset_enemy(s, lane, i);
    }
    // 2. jump
    if ((buttons & TBX_BTN_BUTTON1) && f[A_JUMPS] > 0 && f[A_JUMP_TIMER] == 0) { f[A_JUMPS] -= 1; f[A_JUMP_TIMER] = t.jump_time; }

    // 3. player
    if (mget(s, M_STEP_TX, PLAYER_SLOT) < 0 && at_tile(s, PLAYER_SLOT)) {
        const int tx = mget(s, M_X, PLAYER_SLOT) / TBX_AMI_TILE_WX, ty = mget(s, M_Y, PLAYER_SLOT) / TBX_AMI_TILE_WY;
        const int dir = (buttons & TBX_BTN_UP) ? TBX_DIR_UP : (buttons & TBX_BTN_DOWN) ? TBX_DIR_DOWN :
                        (buttons & TBX_BTN_LEFT) ? TBX_DIR_LEFT : (buttons & TBX_BTN_RIGHT) ? TBX_DIR_RIGHT : -1;
        if (dir >= 0 && can_go(s, tx, ty, dir)) set_step(s, lane, PLAYER_SLOT, tx, ty, dir);
    }
    const int level_before = f[A_LEVEL];
    if (advance(s, lane, PLAYER_SLOT)) player_arrived(t, s, lane);
    if (f[A_LEVEL] != level_before) return;

    // 4. enemies, in index order
    for (int i = 0; i < f[A_N_ENEMIES]; i++) {
        if (mget(s, M_CAUGHT, i)) continue;
        if (mget(s, M_STEP_TX, i) < 0 && at_tile(s, i)) enemy_decide(s, lane, i);
        if (advance(s, lane, i)) {
            if (mget(s, M_KIND, i) != TBX_AI_LOOKUP) {
                const int id = (mget(s, M_Y, i) / TBX_AMI_TILE_WY) * BW + mget(s, M_X, i) / TBX_AMI_TILE_WX;
                mset(s, lane, M_NHIST, i, 0);
                push_history(s, lane, i, id);
            }
        }
    }

    // 5. collisions
    const int px = mget(s, M_X, PLAYER_SLOT), py = mget(s, M_Y, PLAYER_SLOT);
    for (int i = 0; i < f[A_N_ENEMIES]; i++) {
        if (mget(s, M_CAUGHT, i)) continue;
        int dx = mget(s, M_X, i) - px, dy = mget(s, M_Y, i) - py;
        if (dx < 0) dx = -dx;
        if (dy < 0) dy = -dy;
        if (dx >= TBX_AMI_HIT_DX || dy >= TBX_AMI_HIT_DY) continue;
        if (f[A_JUMP_TIMER] > 0) continue;
        if (f[A_CHASE_TIMER] > 0) { mset(s, lane, M_CAUGHT, i, 1); f[A_SCORE] += t.chase_score_bonus; continue; }
        f[A_LIVES] -= 1;
        reset_positions(t, s, lane);
        break;
    }
}

// ------------------------------------------------------------------ step, thread per env
//
// The maze chase is scalar and branchy: in the wave-per-env form above one lane works and 63 idle through ~900 serial
// instructions per frame, and 65 536 such waves queue three deep on every SIMD (145 us per frame of the batch).  Here ONE
// THREAD steps one env, so the whole batch is 1 024 waves, one per SIMD, all running at once.  The rules are restated
// statement by statement over the same arrays in HBM:
//   * the env's scalars (struct-of-arrays over envs) are read coalesced into registers;
//   * the 31 board rows of the wave's 64 envs are staged in LDS with one contiguous 16 KB read (env-major table) and looked
//     up there by (row, column) -- the only indexed structure the rules touch all the time; written back only if some env
//     of the wave painted;
//   * the movers and boxes stay in their env-major tables and are read / written field by field (uncoalesced, but each
//     64-byte line serves the same env's next accesses out of L2).
// Lanes diverge where envs differ (a mover reaches a tile in one env and not in its neighbour); every branch is short.
namespace tpe {

constexpr int ROW_STRIDE = 33;   // 64-bit words per env in LDS: 32 rows + 1 pad (bank spread)

struct Env {
    const AmiDev& d;
    const AmiTables& t;
    int env;
    uint64_t* rows;      // LDS: this env's board rows
    int32_t* mv;         // HBM: this env's mover table [NMF][16] (AI parameters, history)
    uint32_t* bx;        // HBM: this env's boxes, geometry [64] then flags [64]
    Rng rng;
    int32_t f[A_CJ0];    // the scalars the rules read every frame (the chase-junction list behind them stays in memory)
    bool dirty;          // a board row changed

    __device__ __forceinline__ int mg(int field, int slot) const
    {
        const int h = ami_hot_row(field);
        return h >= 0 ? d.mh[((size_t)h * MSLOTS + slot) * (size_t)d.n + env] : mv[field * 16 + slot];
    }
    __device__ __forceinline__ void ms(int field, int slot, int v)
    {
        const int h = ami_hot_row(field);
        if (h >= 0) d.mh[((size_t)h * MSLOTS + slot) * (size_t)d.n + env] = v;
        mv[field * 16 + slot] = v;
    }
    __device__ __forceinline__ int tile_at(int tx, int ty) const
    {
        if (tx < 0 || ty < 0 || tx >= BW || ty >= BH) return TBX_TILE_EMPTY;
        return (int)((rows[ty] >> (2 * tx)) & 3ull);
    }
    __device__ __forceinline__ bool walkable(int tx, int ty) const { return tile_at(tx, ty) != TBX_TILE_EMPTY; }
    __device__ __forceinline__ bool is_junction(int tx, int ty) const
    {
        if (!walkable(tx, ty)) return false;
        const bool h = walkable(tx - 1, ty) || walkable(tx + 1, ty);
        const bool v = walkable(tx, ty - 1) || walkable(tx, ty + 1);
        return h && v;
    }
    __device__ __forceinline__ bool can_go(int tx, int ty, int dir) const
    {
        int dx, dy;
        dir_delta(dir, dx, dy);
        return walkable(tx + dx, ty + dy);
    }

    __device__ __forceinline__ void reset_mover(int slot, int tx, int ty)
    {
        ms(M_X, slot, tx * TBX_AMI_TILE_WX); ms(M_Y, slot, ty * TBX_AMI_TILE_WY);
        ms(M_STEP_TX, slot, -1); ms(M_STEP_TY, slot, -1);
        ms(M_NHIST, slot, 0);
        for (int i = 0; i < TBX_AMI_MAX_HISTORY; i++) ms(M_HIST0 + i, slot, 0);
        ms(M_CAUGHT, slot, 0);
    }
    __device__ __forceinline__ void reset_enemy(int slot)
    {
        const int kind = mg(M_KIND, slot);
        int tx, ty;
        if (kind == TBX_AI_LOOKUP) {
            const int r = mg(M_ROUTE, slot);
            const int id = (r >= 0 && r < TBX_AMI_N_ROUTES) ? AMI_ROUTES[r][0] : 0;
            tx = id % BW; ty = id / BW;
            ms(M_NEXT, slot, 0);
        } else { tx = mg(M_START_TX, slot); ty = mg(M_START_TY, slot); }
        ms(M_VERT, slot, mg(M_SVERT, slot)); ms(M_HORIZ, slot, mg(M_SHORIZ, slot)); ms(M_DIR, slot, mg(M_SDIR, slot));
        ms(M_SEEN_TX, slot, -1); ms(M_SEEN_TY, slot, -1);
        reset_mover(slot, tx, ty);
    }
    __device__ __forceinline__ void reset_player()
    {
        reset_mover(PLAYER_SLOT, t.player_start_tx, t.player_start_ty);
        for (int ty = t.player_start_ty + 1; ty < BH; ty++)
            if (is_junction(t.player_start_tx, ty)) { ms(M_HIST0, PLAYER_SLOT, ty * BW + t.player_start_tx); ms(M_NHIST, PLAYER_SLOT, 1); break; }
    }
    __device__ __forceinline__ void reset_board()
    {
        for (int y = 0; y < 32; y++) rows[y] = y < BH ? t.board_rows[y] : 0ull;
        dirty = true;
        f[A_N_CHASE] = t.n_chase;
        // (the chase-junction list is not part of the per-frame register set: it goes straight to its arrays)
        for (int k = 0; k < TBX_AMI_MAX_CHASE_J; k++) d.sc[(size_t)(A_CJ0 + k) * (size_t)d.n + env] = k < t.n_chase ? t.chase_j[k] : 0;
        f[A_N_BOXES] = t.n_boxes;
        for (int i = 0; i < 64; i++) { bx[i] = t.box_geom[i]; bx[64 + i] = t.box_flags[i]; }
    }
    __device__ __forceinline__ void reset_positions()
    {
        reset_player();
        for (int i = 0; i < f[A_N_ENEMIES]; i++) reset_enemy(i);
        f[A_JUMP_TIMER] = 0;
        f[A_CHASE_TIMER] = 0;
    }
    __device__ __forceinline__ void new_game(Rng& sim)
    {
        rng = sim.child();
        for (int i = 0; i < A_CJ0; i++) f[i] = 0;
        f[A_LIVES] = t.start_lives;
        f[A_LEVEL] = 1;
        f[A_JUMPS] = t.start_jumps;
        reset_board();
        for (int i = 0; i < NMF * 16; i++) mv[i] = 0;
        for (int h = 0; h < NMH * MSLOTS; h++) d.mh[(size_t)h * (size_t)d.n + env] = 0;
        f[A_N_ENEMIES] = t.n_enemies;
        ms(M_SPEED, PLAYER_SLOT, TBX_AMI_SPEED); ms(M_KIND, PLAYER_SLOT, TBX_AI_PLAYER);
        ms(M_SEEN_TX, PLAYER_SLOT, -1); ms(M_SEEN_TY, PLAYER_SLOT, -1);
        for (int e = 0; e < t.n_enemies; e++) {
            ms(M_SPEED, e, TBX_AMI_SPEED);
            for (int k = 0; k < 14; k++) ms(M_KIND + k, e, t.ai[e][k]);
        }
        reset_player();
        for (int i = 0; i < t.n_enemies; i++) reset_enemy(i);
    }

    __device__ __forceinline__ bool at_tile(int slot) const { return mg(M_X, slot) % TBX_AMI_TILE_WX == 0 && mg(M_Y, slot) % TBX_AMI_TILE_WY == 0; }
    __device__ __forceinline__ bool advance(int slot)
    {
        const int stx = mg(M_STEP_TX, slot);
        if (stx < 0) return false;
        const int sty = mg(M_STEP_TY, slot);
        int x = mg(M_X, slot), y = mg(M_Y, slot), sp = mg(M_SPEED, slot);
        const int gx = stx * TBX_AMI_TILE_WX, gy = sty * TBX_AMI_TILE_WY;
        if (sp < 0) sp = 0;
        if (x < gx) { x += sp; if (x > gx) x = gx; }
        else if (x > gx) { x -= sp; if (x < gx) x = gx; }
        else if (y < gy) { y += sp; if (y > gy) y = gy; }
        else if (y > gy) { y -= sp; if (y < gy) y = gy; }
        ms(M_X, slot, x); ms(M_Y, slot, y);
        if (x == gx && y == gy) { ms(M_STEP_TX, slot, -1); ms(M_STEP_TY, slot, -1); return true; }
        return false;
    }
    __device__ __forceinline__ void set_step(int slot, int tx, int ty, int dir)
    {
        int dx, dy;
        dir_delta(dir, dx, dy);
        ms(M_STEP_TX, slot, tx + dx); ms(M_STEP_TY, slot, ty + dy);
    }
    __device__ __forceinline__ void push_history(int slot, int id)
    {
        int n = mg(M_NHIST, slot);
        if (n >= TBX_AMI_MAX_HISTORY) {
            for (int i = 1; i < TBX_AMI_MAX_HISTORY; i++) ms(M_HIST0 + i - 1, slot, mg(M_HIST0 + i, slot));
            n = TBX_AMI_MAX_HISTORY - 1;
        }
        if (n >= 0) ms(M_HIST0 + n, slot, id);          // (a negative count -- hand-written states -- matches no slot, as in the wave form)
        ms(M_NHIST, slot, n + 1);
    }
    // M_NHIST outside 1..16 (hand-written states): like the wave form, no slot matches and the value is 0
    __device__ __forceinline__ int last_history(int slot, int n) const { return (n >= 1 && n <= TBX_AMI_MAX_HISTORY) ? mg(M_HIST0 + n - 1, slot) : 0; }

    __device__ __forceinline__ void check_boxes()
    {
        const int nb = f[A_N_BOXES];
        int n_new = 0;
        bool trig_new = false;
        for (int i = 0; i < nb && i < 64; i++) {
            const uint32_t fl = bx[64 + i];
            if (fl & 1u) continue;
            const uint32_t g = bx[i];
            const int tl_tx = g & 255, tl_ty = (g >> 8) & 255, br_tx = (g >> 16) & 255, br_ty = (g >> 24) & 255;
            if (!(tl_tx < BW && br_tx < BW && tl_tx <= br_tx) || tl_ty >= BH || br_ty >= BH || tl_ty > br_ty) continue;
            const int nbits = 2 * (br_tx - tl_tx + 1);
            const uint64_t span = (nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull)) << (2 * tl_tx);
            const uint64_t want = 0xAAAAAAAAAAAAAAAAull & span;
            bool ok = (rows[tl_ty] & span) == want && (rows[br_ty] & span) == want;
            for (int y = tl_ty + 1; y < br_ty && ok; y++) {
                const uint64_t row = rows[y];
                if (((row >> (2 * tl_tx)) & 3ull) != TBX_TILE_PAINTED || ((row >> (2 * br_tx)) & 3ull) != TBX_TILE_PAINTED) ok = false;
            }
            if (!ok) continue;
            bx[64 + i] = fl | 1u;
            n_new += 1;
            if (fl & 2u) trig_new = true;
        }
        if (!n_new) return;
        f[A_SCORE] += t.box_bonus * n_new;
        if (trig_new) {
            bool all = true;
            for (int k = 0; k < nb && k < 64; k++) {
                const uint32_t fl = bx[64 + k];
                if ((fl & 2u) && !(fl & 1u)) all = false;
            }
            if (all) f[A_CHASE_TIMER] = t.chase_time;
        }
    }

    __device__ __forceinline__ void player_arrived()
    {
        const int tx = mg(M_X, PLAYER_SLOT) / TBX_AMI_TILE_WX, ty = mg(M_Y, PLAYER_SLOT) / TBX_AMI_TILE_WY;
        if (!is_junction(tx, ty)) return;
        const int id = ty * BW + tx;
        int newly = 0;
        const int nh = mg(M_NHIST, PLAYER_SLOT);
        if (nh > 0) {
            const int prev = last_history(PLAYER_SLOT, nh);
            const int qx = prev % BW, qy = prev / BW;
            if (prev != id && prev >= 0 && prev < BW * BH && (qx == tx || qy == ty)) {
                const int x0 = qx < tx ? qx : tx, x1 = qx < tx ? tx : qx, y0 = qy < ty ? qy : ty, y1 = qy < ty ? ty : qy;
                const int nbits = 2 * (x1 - x0 + 1);
                const uint64_t span = (nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull)) << (2 * x0);
                const uint64_t ones = 0x5555555555555555ull & span;
                bool clear = true;
                for (int y = y0; y <= y1; y++) {
                    const uint64_t row = rows[y];
                    if ((((row | (row >> 1)) & 0x5555555555555555ull) & span) != ones) clear = false;   // a 2-bit tag of 0 = not walkable
                }
                if (clear) {
                    for (int y = y0; y <= y1; y++) {
                        const uint64_t row = rows[y];
                        const uint64_t painted = (row >> 1) & ~row & 0x5555555555555555ull;
                        newly += __popcll(ones & ~painted);
                        rows[y] = (row & ~span) | (0xAAAAAAAAAAAAAAAAull & span);
                    }
                    dirty = true;
                }
            }
        }
        push_history(PLAYER_SLOT, id);
        if (newly > 0) {
            f[A_SCORE] += newly;
            check_boxes();
            bool left = false;
            for (int y = 0; y < BH; y++) if (rows[y] & 0x5555555555555555ull) left = true;   // tags 01 / 11 remain
            if (!left) {
                f[A_LEVEL] += 1;
                reset_board();
                reset_positions();
                f[A_JUMPS] = t.start_jumps;
            }
        }
    }

    __device__ __forceinline__ int first_open(int tx, int ty, int avoid) const
    {
        for (int dd = 0; dd < 4; dd++)
            if (dd != avoid && can_go(tx, ty, dd)) return dd;
        return (avoid >= 0 && can_go(tx, ty, avoid)) ? avoid : -1;
    }

    __device__ __forceinline__ void enemy_decide(int slot)
    {
        const int tx = mg(M_X, slot) / TBX_AMI_TILE_WX, ty = mg(M_Y, slot) / TBX_AMI_TILE_WY;
        const int kind = mg(M_KIND, slot);
        int dir = -1;
        if (kind == TBX_AI_LOOKUP) {
            const int r = mg(M_ROUTE, slot);
            if (r < 0 || r >= TBX_AMI_N_ROUTES) return;
            int len = 0;
            while (len < TBX_AMI_ROUTE_LEN && AMI_ROUTES[r][len] >= 0) len++;
            int next = mg(M_NEXT, slot);
            if (next < 0 || next >= len) next = 0;
            if (AMI_ROUTES[r][next] == ty * BW + tx) next = (next + 1) % len;
            ms(M_NEXT, slot, next);
            const int gx = AMI_ROUTES[r][next] % BW, gy = AMI_ROUTES[r][next] / BW;
            if (gx > tx && can_go(tx, ty, TBX_DIR_RIGHT)) dir = TBX_DIR_RIGHT;
            else if (gx < tx && can_go(tx, ty, TBX_DIR_LEFT)) dir = TBX_DIR_LEFT;
            else if (gy > ty && can_go(tx, ty, TBX_DIR_DOWN)) dir = TBX_DIR_DOWN;
            else if (gy < ty && can_go(tx, ty, TBX_DIR_UP)) dir = TBX_DIR_UP;
        } else if (kind == TBX_AI_PERIMETER) {
            if (ty == 0 && tx < BW - 1 && can_go(tx, ty, TBX_DIR_RIGHT)) dir = TBX_DIR_RIGHT;
            else if (tx == BW - 1 && ty < BH - 1 && can_go(tx, ty, TBX_DIR_DOWN)) dir = TBX_DIR_DOWN;
            else if (ty == BH - 1 && tx > 0 && can_go(tx, ty, TBX_DIR_LEFT)) dir = TBX_DIR_LEFT;
            else if (tx == 0 && ty > 0 && can_go(tx, ty, TBX_DIR_UP)) dir = TBX_DIR_UP;
            else {
                if (can_go(tx, ty, TBX_DIR_UP)) dir = TBX_DIR_UP;
                else if (can_go(tx, ty, TBX_DIR_LEFT)) dir = TBX_DIR_LEFT;
                else if (can_go(tx, ty, TBX_DIR_DOWN)) dir = TBX_DIR_DOWN;
                else if (can_go(tx, ty, TBX_DIR_RIGHT)) dir = TBX_DIR_RIGHT;
            }
        } else if (kind == TBX_AI_AMIDAR) {
            int vert = mg(M_VERT, slot) & 1;
            int horiz = 2 | (mg(M_HORIZ, slot) & 1);
            bool came_vertically = true;
            const int nh = mg(M_NHIST, slot);
            if (nh > 0) came_vertically = (last_history(slot, nh) % BW) == tx;
            const bool can_h = can_go(tx, ty, horiz), can_v = can_go(tx, ty, vert);
            if (came_vertically && can_h) dir = horiz;
            else if (can_v) dir = vert;
            else if (can_h) dir = horiz;
            else {
                vert ^= 1; horiz ^= 1;
                if (can_go(tx, ty, vert)) dir = vert;
                else if (can_go(tx, ty, horiz)) dir = horiz;
            }
            ms(M_VERT, slot, vert);
            ms(M_HORIZ, slot, horiz);
        } else if (kind == TBX_AI_TARGET_PLAYER) {
            const int ptx = mg(M_X, PLAYER_SLOT) / TBX_AMI_TILE_WX, pty = mg(M_Y, PLAYER_SLOT) / TBX_AMI_TILE_WY;
            const int ddx = ptx - tx, ddy = pty - ty;
            const int adx = ddx < 0 ? -ddx : ddx, ady = ddy < 0 ? -ddy : ddy;
            int cur = mg(M_DIR, slot) & 3;
            if (adx + ady <= mg(M_VISION, slot)) {
                ms(M_SEEN_TX, slot, ptx); ms(M_SEEN_TY, slot, pty);
                const int hd = ddx > 0 ? TBX_DIR_RIGHT : TBX_DIR_LEFT, vd = ddy > 0 ? TBX_DIR_DOWN : TBX_DIR_UP;
                const int first = adx >= ady ? hd : vd, second = adx >= ady ? vd : hd;
                const int fz = adx >= ady ? adx : ady, sz = adx >= ady ? ady : adx;
                if (fz > 0 && can_go(tx, ty, first)) dir = first;
                else if (sz > 0 && can_go(tx, ty, second)) dir = second;
            } else { ms(M_SEEN_TX, slot, -1); ms(M_SEEN_TY, slot, -1); }
            if (dir < 0) dir = can_go(tx, ty, cur) ? cur : first_open(tx, ty, cur ^ 1);
            if (dir >= 0) cur = dir;
            ms(M_DIR, slot, cur);
        } else if (kind == TBX_AI_RANDOM) {
            int cur = mg(M_DIR, slot) & 3;
            int o0 = 0, o1 = 0, o2 = 0, o3 = 0, n = 0;     // (no indexed array: it would live in scratch)
            for (int dd = 0; dd < 4; dd++)
                if (dd != (cur ^ 1) && can_go(tx, ty, dd)) {
                    if (n == 0) o0 = dd; else if (n == 1) o1 = dd; else if (n == 2) o2 = dd; else o3 = dd;
                    n++;
                }
            if (n == 0) dir = can_go(tx, ty, cur ^ 1) ? (cur ^ 1) : -1;
            else {
                const int k = (int)rng.range((uint64_t)n);
                dir = k == 0 ? o0 : k == 1 ? o1 : k == 2 ? o2 : o3;
            }
            if (dir >= 0) cur = dir;
            ms(M_DIR, slot, cur);
        } else {
            return;
        }
        if (dir >= 0) set_step(slot, tx, ty, dir);
    }

    // a mover's per-frame fields, fetched with independent loads (one latency) and kept in registers while it moves
    struct Hot { int x, y, stx, sty, caught, sp; };
    __device__ __forceinline__ Hot load_hot(int slot) const
    {
        Hot h;
        h.x = mg(M_X, slot); h.y = mg(M_Y, slot); h.stx = mg(M_STEP_TX, slot); h.sty = mg(M_STEP_TY, slot);
        h.caught = mg(M_CAUGHT, slot); h.sp = mg(M_SPEED, slot);
        return h;
    }
    // advance toward the step; true when the target tile was reached this frame (fields written back by the caller)
    static __device__ __forceinline__ bool advance_hot(Hot& h)
    {
        if (h.stx < 0) return false;
        const int gx = h.stx * TBX_AMI_TILE_WX, gy = h.sty * TBX_AMI_TILE_WY;
        const int sp = h.sp < 0 ? 0 : h.sp;
        if (h.x < gx) { h.x += sp; if (h.x > gx) h.x = gx; }
        else if (h.x > gx) { h.x -= sp; if (h.x < gx) h.x = gx; }
        else if (h.y < gy) { h.y += sp; if (h.y > gy) h.y = gy; }
        else if (h.y > gy) { h.y -= sp; if (h.y < gy) h.y = gy; }
        if (h.x == gx && h.y == gy) { h.stx = -1; h.sty = -1; return true; }
        return false;
    }

    __device__ __forceinline__ void step(uint32_t buttons)
    {
        // 1. timers
        if (f[A_JUMP_TIMER] > 0) f[A_JUMP_TIMER] -= 1;
        if (f[A_CHASE_TIMER] > 0) {
            f[A_CHASE_TIMER] -= 1;
            if (f[A_CHASE_TIMER] == 0)
                for (int i = 0; i < f[A_N_ENEMIES]; i++)
                    if (mg(M_CAUGHT, i)) reset_enemy(i);
        }
        // 2. jump
        if ((buttons & TBX_BTN_BUTTON1) && f[A_JUMPS] > 0 && f[A_JUMP_TIMER] == 0) { f[A_JUMPS] -= 1; f[A_JUMP_TIMER] = t.jump_time; }
        // every mover's per-frame fields with independent loads up front: the walk below is a chain of short dependent steps,
        // and with one wave per SIMD each round trip to L2 in that chain would be paid in full
        const int ne = f[A_N_ENEMIES];
        Hot p = load_hot(PLAYER_SLOT);
        Hot hs[TBX_AMI_MAX_ENEMIES];
#pragma unroll
        for (int i = 0; i < TBX_AMI_MAX_ENEMIES; i++) {
            hs[i] = Hot{0, 0, -1, -1, 1, 0};
            if (i < ne) hs[i] = load_hot(i);
        }
        // 3. player
        if (p.stx < 0 && p.x % TBX_AMI_TILE_WX == 0 && p.y % TBX_AMI_TILE_WY == 0) {
            const int tx = p.x / TBX_AMI_TILE_WX, ty = p.y / TBX_AMI_TILE_WY;
            const int dir = (buttons & TBX_BTN_UP) ? TBX_DIR_UP : (buttons & TBX_BTN_DOWN) ? TBX_DIR_DOWN :
                            (buttons & TBX_BTN_LEFT) ? TBX_DIR_LEFT : (buttons & TBX_BTN_RIGHT) ? TBX_DIR_RIGHT : -1;
            if (dir >= 0 && can_go(tx, ty, dir)) {
                int dx, dy;
                dir_delta(dir, dx, dy);
                p.stx = tx + dx; p.sty = ty + dy;
            }
        }
        const int level_before = f[A_LEVEL];
        const bool p_arrived = advance_hot(p);
        ms(M_X, PLAYER_SLOT, p.x); ms(M_Y, PLAYER_SLOT, p.y); ms(M_STEP_TX, PLAYER_SLOT, p.stx); ms(M_STEP_TY, PLAYER_SLOT, p.sty);
        if (p_arrived) player_arrived();
        if (f[A_LEVEL] != level_before) return;
        // 4. enemies, in index order (unrolled: slot numbers are constants, the positions stay in registers for pass 5)
#pragma unroll
        for (int i = 0; i < TBX_AMI_MAX_ENEMIES; i++) {
            if (i < ne && !hs[i].caught) {
                Hot& h = hs[i];
                // (the protocol's own fields live in the env-major table: only the lanes that decide / arrive touch it)
                if (h.stx < 0 && h.x % TBX_AMI_TILE_WX == 0 && h.y % TBX_AMI_TILE_WY == 0) {
                    if (mg(M_KIND, i) == TBX_AI_LOOKUP) {
                        // EnemyLookupAI, the default protocol, with the step kept in registers
                        const int tx = h.x / TBX_AMI_TILE_WX, ty = h.y / TBX_AMI_TILE_WY, r = mg(M_ROUTE, i);
                        if (r >= 0 && r < TBX_AMI_N_ROUTES) {
                            int len = 0;
                            while (len < TBX_AMI_ROUTE_LEN && AMI_ROUTES[r][len] >= 0) len++;
                            int next = mg(M_NEXT, i);
                            if (next < 0 || next >= len) next = 0;
                            if (AMI_ROUTES[r][next] == ty * BW + tx) next = (next + 1) % len;
                            ms(M_NEXT, i, next);
                            const int gx = AMI_ROUTES[r][next] % BW, gy = AMI_ROUTES[r][next] / BW;
                            int dir = -1;
                            if (gx > tx && can_go(tx, ty, TBX_DIR_RIGHT)) dir = TBX_DIR_RIGHT;
                            else if (gx < tx && can_go(tx, ty, TBX_DIR_LEFT)) dir = TBX_DIR_LEFT;
                            else if (gy > ty && can_go(tx, ty, TBX_DIR_DOWN)) dir = TBX_DIR_DOWN;
                            else if (gy < ty && can_go(tx, ty, TBX_DIR_UP)) dir = TBX_DIR_UP;
                            if (dir >= 0) {
                                int dx, dy;
                                dir_delta(dir, dx, dy);
                                h.stx = tx + dx; h.sty = ty + dy;
                                ms(M_STEP_TX, i, h.stx); ms(M_STEP_TY, i, h.sty);
                            }
                        }
                    } else {
                        enemy_decide(i);                    // writes the step (and the protocol's own fields) to the table
                        h.stx = mg(M_STEP_TX, i); h.sty = mg(M_STEP_TY, i);
                    }
                }
                const bool moved = h.stx >= 0;
                const bool arrived = advance_hot(h);
                if (moved) { ms(M_X, i, h.x); ms(M_Y, i, h.y); }
                if (arrived) {
                    ms(M_STEP_TX, i, -1); ms(M_STEP_TY, i, -1);
                    if (mg(M_KIND, i) != TBX_AI_LOOKUP) {
                        const int id = (h.y / TBX_AMI_TILE_WY) * BW + h.x / TBX_AMI_TILE_WX;
                        ms(M_NHIST, i, 0);
                        push_history(i, id);
                    }
                }
            }
        }
        // 5. collisions
        bool hit = false;
#pragma unroll
        for (int i = 0; i < TBX_AMI_MAX_ENEMIES; i++) {
            if (i < ne && !hit && !hs[i].caught) {
                int dx = hs[i].x - p.x, dy = hs[i].y - p.y;
                if (dx < 0) dx = -dx;
                if (dy < 0) dy = -dy;
                if (dx < TBX_AMI_HIT_DX && dy < TBX_AMI_HIT_DY && f[A_JUMP_TIMER] <= 0) {
                    if (f[A_CHASE_TIMER] > 0) { ms(M_CAUGHT, i, 1); f[A_SCORE] += t.chase_score_bonus; }
                    else { f[A_LIVES] -= 1; reset_positions(); hit = true; }
                }
            }
        }
    }
};

}  // namespace tpe

// the batch protocol (tbx_step / tbx_step_device / tbx_step_synthetic) and the agent layer's action repeat (`frames` frames of
// one action with the reward summed, MaxAndSkipEnv's two buffer slots written on the way), one wave of 64 envs per block
// AGENT: the agent layer's action repeat and frame-buffer slots; the batch protocol's instantiation carries neither
template <bool AGENT>
__global__ __launch_bounds__(64) void ami_step_tpe_kernel(AmiDev d, AmiDev slot_a, AmiDev slot_b, ActionSource src, uint32_t flags)
{
    __shared__ uint64_t lds_rows[64 * tpe::ROW_STRIDE];
    const int lane = threadIdx.x;
    const int env0 = blockIdx.x * 64;
    const int env = env0 + lane;
    const size_t N = (size_t)d.n;
    const int n_here = min(64, d.n - env0);
    // the board rows of this wave's envs: one contiguous read of the env-major table
    for (int it = 0; it < 32; it++) {
        const int g = it * 64 + lane;                   // (env_rel, row) = (g / 32, g % 32)
        if ((g >> 5) < n_here) lds_rows[(g >> 5) * tpe::ROW_STRIDE + (g & 31)] = d.tiles[(size_t)env0 * 32 + g];
    }
    __syncthreads();
    const bool agent = AGENT && src.acc_reward != nullptr;
    // an env whose game ended in an earlier launch of the same agent step sits this one out (MaxAndSkipEnv left its loop)
    bool active = env < d.n && !(AGENT && env < d.n && tbx_agent_env_finished(src, env));
    if (AGENT && src.exec_flag && env < d.n) src.exec_flag[env] = active ? 1 : 0;
    uint32_t buttons = 0;
    if (active) {
        int a;
        if (src.actions) a = src.actions[env];
        else {
            const uint64_t h = tbx_splitmix64(src.seed ^ ((src.env_offset + (uint64_t)env) << 32) ^ src.t);
            a = tbx_legal_action(TBX_GAME_AMIDAR, (int)(h % 6ull));
        }
        buttons = tbx_ale_buttons(a);
        if (buttons == 0xFFu) { buttons = 0; atomicOr(d.err_flag, 1u); }
    }
    const int ec = env < d.n ? env : 0;                 // (idle lanes of the last block point at a valid env and never step)
    tpe::Env e{d, *d.tab, ec, lds_rows + lane * tpe::ROW_STRIDE, d.movers + (size_t)ec * NMF * 16, d.boxes + (size_t)ec * 128, Rng{}, {}, false};
    int32_t prev = 0, rew = 0, out_lives = 0, out_score = 0;
    bool is_done = false;
    const bool loaded = active;
    if (active) {
        e.rng.s0 = d.rng[env]; e.rng.s1 = d.rng[N + env];
#pragma unroll
        for (int i = 0; i < A_CJ0; i++) e.f[i] = d.sc[(size_t)i * N + env];
        prev = d.prev_score[env];
    }
    const int frames = AGENT && src.frames > 1 ? src.frames : 1;
    for (int fr = 0; fr < frames; fr++) {
        uint32_t slots = 0;
        if (active) {
            e.step(buttons);
            rew = e.f[A_SCORE] - prev;
            if (rew < 0) rew = 0;
            out_lives = e.f[A_LIVES]; out_score = e.f[A_SCORE];
            is_done = out_lives <= 0;
            prev = out_score;
            if (!AGENT && is_done && (flags & TBX_STEP_AUTO_RESET)) {   // (the agent layer resets through its own procedure)
                Rng sim;
                sim.s0 = d.sim_rng[env]; sim.s1 = d.sim_rng[N + env];
                e.new_game(sim);
                d.sim_rng[env] = sim.s0; d.sim_rng[N + env] = sim.s1;
                prev = e.f[A_SCORE];
            }
            if (AGENT) tbx_accumulate(src, env, rew, is_done, fr);
            if (AGENT && src.buf_valid) slots = tbx_snap_slots(src, fr);
        }
        // MaxAndSkipEnv's frame buffer: the envs that ran frame skip-2 / skip-1 copy what the rasteriser reads of their
        // state into slot A / B -- scalars by the env's own thread (coalesced), board rows, boxes and the movers' position
        // rows by the whole wave, one env at a time
        if (agent && __syncthreads_or(slots != 0)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // box flags written by check_boxes are read below by other lanes
            __syncthreads();
            for (int which = 0; which < 2; which++) {
                const bool mine = (slots >> which) & 1u;
                const AmiDev& dst = which ? slot_b : slot_a;
                if (mine) {
#pragma unroll
                    for (int i = 0; i < A_CJ0; i++) dst.sc[(size_t)i * N + env] = e.f[i];
                    src.buf_valid[env] |= (uint8_t)(1u << which);
                }
                // flat element loops over the wave's 64 envs (independent loads, so several are in flight at once; one env
                // after the other was a chain of 64 load -> store round trips per slot)
                const uint64_t want = __ballot(mine);
#pragma unroll 4
                for (int it = 0; it < 32; it++) {                      // board rows: 64 envs x 32 rows of 8 bytes, from LDS
                    const int g = it * 64 + lane, r = g >> 5;
                    if ((want >> r) & 1ull) dst.tiles[(size_t)env0 * 32 + g] = lds_rows[r * tpe::ROW_STRIDE + (g & 31)];
                }
                // (16 bytes per lane and eight loads in flight: at one wave per SIMD every dependent round trip is paid in full)
                const uint4* bsrc = reinterpret_cast<const uint4*>(d.boxes + (size_t)env0 * 128);
                uint4* bdst = reinterpret_cast<uint4*>(dst.boxes + (size_t)env0 * 128);
#pragma unroll 8
                for (int it = 0; it < 32; it++) {                      // boxes: 64 envs x 128 dwords = 64 x 32 quads
                    const int g = it * 64 + lane, r = g >> 5;
                    if ((want >> r) & 1ull) bdst[g] = bsrc[g];
                }
                // the movers' positions and caught flags: the three table rows the painter reads (the table is current: the
                // thread form writes it along with its struct-of-arrays mirror)
#pragma unroll 4
                for (int it = 0; it < 12; it++) {                      // 64 envs x 3 rows x 4 quads of slots
                    const int g = it * 64 + lane, r = g / 12, w = g - r * 12;
                    const int fld = w < 4 ? M_X : w < 8 ? M_Y : M_CAUGHT;
                    const size_t at = (size_t)(env0 + r) * NMF * 16 + (size_t)fld * 16 + (size_t)(w & 3) * 4;
                    if ((want >> r) & 1ull) *reinterpret_cast<uint4*>(dst.movers + at) = *reinterpret_cast<const uint4*>(d.movers + at);
                }
            }
        }
        if (agent && is_done) active = false;            // ... and its loop ends with the game
    }
    if (loaded) {
        d.rng[env] = e.rng.s0; d.rng[N + env] = e.rng.s1;
#pragma unroll
        for (int i = 0; i < A_CJ0; i++) d.sc[(size_t)i * N + env] = e.f[i];
        d.prev_score[env] = prev;
        d.reward[env] = rew;
        d.done[env] = is_done ? 1 : 0;
        d.lives_out[env] = out_lives;
        d.score_out[env] = out_score;
        const uint32_t lv = out_lives < 0 ? 0u : out_lives > 255 ? 255u : (uint32_t)out_lives;
        d.packed[env] = (uint64_t)(uint32_t)rew | ((uint64_t)(is_done ? 1u : 0u) << 32) | ((uint64_t)lv << 40);
    }
    // only the envs that changed their board write it back: 32 lanes store the env's rows as one 256-byte run
    __syncthreads();
    for (uint64_t m = __ballot(loaded && e.dirty); m; m &= m - 1) {
        const int r = (int)__builtin_ctzll(m);
        if (lane < 32) d.tiles[(size_t)(env0 + r) * 32 + lane] = lds_rows[r * tpe::ROW_STRIDE + lane];
    }
}

// ------------------------------------------------------------------ kernels

__global__ __launch_bounds__(TBX_BLOCK) void ami_new_game_kernel(AmiDev d, const uint8_t* mask)
{
    const int lane = threadIdx.x & 63;
    const int env = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (env >= d.n) return;
    if (mask && !mask[env]) return;
    const size_t N = (size_t)d.n;
    Rng sim;
    sim.s0 = d.sim_rng[env];
    sim.s1 = d.sim_rng[N + env];
    AmiRegs s;
    ami_new_game(*d.tab, lane, sim, s);
    ami_store(d, env, lane, s);
    if (lane == 0) {
        d.sim_rng[env] = sim.s0;
        d.sim_rng[N + env] = sim.s1;
        d.prev_score[env] = s.f[A_SCORE];
    }
}

// one frame (or the agent layer's whole action repeat) of one env on one wave
// AGENT: the agent layer's whole action repeat with MaxAndSkipEnv's bookkeeping; the batch protocol's instantiation carries
// neither the slot structs nor the frame loop and needs fewer registers (the kernel is bound by latency x occupancy)
template <bool AGENT>
__device__ __forceinline__ void ami_step_body(const AmiDev& d, const AmiDev& slot_a, const AmiDev& slot_b, const ActionSource& src, uint32_t flags, int env, int lane)
{
    const size_t N = (size_t)d.n;
    if (AGENT) {
        if (src.exec_flag && lane == 0) src.exec_flag[env] = tbx_agent_env_finished(src, env) ? 0 : 1;
        if (tbx_agent_env_finished(src, env)) return;     // MaxAndSkipEnv left its loop when this env's game ended
    }

    uint32_t buttons;
    if (src.single_env >= 0) {
        buttons = src.single_buttons;
    } else {
        int a;
        if (src.actions) a = src.actions[env];
        else {
            uint64_t h = tbx_splitmix64(src.seed ^ ((src.env_offset + (uint64_t)env) << 32) ^ src.t);
            a = tbx_legal_action(TBX_GAME_AMIDAR, (int)(h % 6ull));
        }
        buttons = tbx_ale_buttons(a);
        if (buttons == 0xFFu) {
            buttons = 0;
            if (lane == 0) atomicOr(d.err_flag, 1u);
        }
    }

    AmiRegs s;
    ami_load(d, env, lane, s);
    int32_t prev = d.prev_score[env];
    const int frames = AGENT && src.frames > 1 ? src.frames : 1;
    int32_t rew = 0, out_lives = 0, out_score = 0;
    bool is_done = false;
    for (int fr = 0; fr < frames; fr++) {                  // > 1: the agent layer's action repeat, state stays in registers
        ami_step(*d.tab, lane, buttons, s);
        rew = s.f[A_SCORE] - prev;
        if (rew < 0) rew = 0;
        out_lives = s.f[A_LIVES]; out_score = s.f[A_SCORE];
        is_done = out_lives <= 0;
        prev = out_score;
        if (!AGENT && is_done && (flags & TBX_STEP_AUTO_RESET)) {   // (the agent layer resets through its own procedure)
            Rng sim;
            sim.s0 = d.sim_rng[env];
            sim.s1 = d.sim_rng[N + env];
            ami_new_game(*d.tab, lane, sim, s);
            if (lane == 0) { d.sim_rng[env] = sim.s0; d.sim_rng[N + env] = sim.s1; }
            prev = s.f[A_SCORE];
        }
        if (AGENT) {
            if (lane == 0) tbx_accumulate(src, env, rew, is_done, fr);
            if (src.buf_valid) {                             // MaxAndSkipEnv's frame buffer: slot A after frame skip-2, B after skip-1
                const uint32_t slots = tbx_snap_slots(src, fr);
                if (slots & 1u) ami_store(slot_a, env, lane, s);
                if (slots & 2u) ami_store(slot_b, env, lane, s);
                if (slots && lane == 0) src.buf_valid[env] |= (uint8_t)slots;
                if (is_done) break;                          // ... and its loop ends with the game
            }
        }
    }
    ami_store(d, env, lane, s);
    if (lane == 0) {
        d.prev_score[env] = prev;
        d.reward[env] = rew;
        d.done[env] = is_done ? 1 : 0;
        d.lives_out[env] = out_lives;
        d.score_out[env] = out_score;
        uint32_t lv = out_lives < 0 ? 0u : out_lives > 255 ? 255u : (uint32_t)out_lives;
        d.packed[env] = (uint64_t)(uint32_t)rew | ((uint64_t)(is_done ? 1u : 0u) << 32) | ((uint64_t)lv << 40);
    }
}


__global__ __launch_bounds__(TBX_BLOCK) void ami_step_kernel(AmiDev d, ActionSource src, uint32_t flags, int first_env, int count)
{
    const int lane = threadIdx.x & 63;
    const int rel = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (rel >= count) return;
    ami_step_body<false>(d, d, d, src, flags, first_env + rel, lane);
}

__global__ __launch_bounds__(TBX_BLOCK) void ami_agent_step_kernel(AmiDev d, AmiDev slot_a, AmiDev slot_b, ActionSource src, uint32_t flags, int first_env, int count)
{
    const int lane = threadIdx.x & 63;
    const int rel = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (rel >= count) return;
    ami_step_body<true>(d, slot_a, slot_b, src, flags, first_env + rel, lane);
}

// reset-time wrappers of the agent layer for the envs flagged in r.kind (agent_device.hpp, AgentResetProc)
struct AmiAgentEnv {
    const AmiTables& c;
    int lane;
    AmiRegs& s;
    Rng& sim;
    const AmiDev& slot_a;
    const AmiDev& slot_b;
    int env;
    __device__ __forceinline__ void snapshot(int slot) { ami_store(slot ? slot_b : slot_a, env, lane, s); }
    __device__ __forceinline__ void step(uint32_t buttons) { ami_step(c, lane, buttons, s); }
    __device__ __forceinline__ void new_game() { ami_new_game(c, lane, sim, s); }
    __device__ __forceinline__ int lives() const { return wave_uniform(s.f[A_LIVES]); }
    __device__ __forceinline__ int score() const { return wave_uniform(s.f[A_SCORE]); }
};

__global__ __launch_bounds__(TBX_BLOCK) void ami_agent_reset_kernel(AmiDev d, AmiDev slot_a, AmiDev slot_b, AgentResetArgs r)
{
    const int lane = threadIdx.x & 63;
    // a persistent grid walks the compact list of flagged envs (or every env when there is no list)
    const int wave_id = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6)), n_waves = gridDim.x * TBX_WAVES_PER_BLOCK;
    const int total = r.list ? wave_uniform(*r.count) : d.n;
    for (int it = wave_id; it < total; it += n_waves) {
        const int env = r.list ? wave_uniform(r.list[it]) : it;
        if (wave_uniform((int)r.kind[env]) == 0) continue;
        const size_t N = (size_t)d.n;
        AmiRegs s;
        ami_load(d, env, lane, s);
        Rng sim;
        sim.s0 = d.sim_rng[env]; sim.s1 = d.sim_rng[N + env];
        AgentMonitor m = agent_monitor_load(r, env);
        AmiAgentEnv ops{*d.tab, lane, s, sim, slot_a, slot_b, env};
        AgentResetProc<AmiAgentEnv> proc{ops, r, m, r.env_offset + (uint64_t)env, wave_uniform(d.prev_score[env]),
                                         (uint32_t)wave_uniform((int)r.buf_valid[env]), r.noop_override ? wave_uniform(r.noop_override[env]) : 0, false};
        proc.run();
        ami_store(d, env, lane, s);
        if (lane == 0) {
            d.sim_rng[env] = sim.s0; d.sim_rng[N + env] = sim.s1;
            d.prev_score[env] = proc.prev;
            agent_monitor_store(r, env, m, proc.valid, proc.obs_raw);
        }
    }
}

// ------------------------------------------------------------------ render

#ifndef AMI_UNIT_ROWS_V
#define AMI_UNIT_ROWS_V 10
#endif
constexpr int AMI_UNIT_ROWS = AMI_UNIT_ROWS_V;   // 250 = 25 units; 10 x 480 B (RGB) of LDS per wave (other values: the last unit is shorter)
// Round 5 measured the unit SIZE as part of the store-address question (4 800-byte units start 64 bytes off a 128-byte line every
// other time; as a pure store stream 128-byte aligned units gain 9-11 %): builds with 12 rows (5 760 B, Breakout's unit) and 8 rows
// (3 840 B), parity-green for every split factor, RGB launch at 65 536 envs on one box, best split of each: 1.365 ms (10 rows, nine
// waves per frame) / 1.510 (12 rows, eleven) / 1.444 (8 rows, ten) -- ten rows are two tile rows of the board, and that outweighs the
// addresses.  scripts/ami_units_probe.sh, profiles/r05_experiments.txt.

__device__ __forceinline__ int world_to_px(int v)
{
    return v >= 0 ? v / TBX_AMI_WORLD_SCALE : -((-v + TBX_AMI_WORLD_SCALE - 1) / TBX_AMI_WORLD_SCALE);
}

// ---- what the painter derives from the state, as functions of the state (shared by its set-up and by the record writer)

// lane = board row: tiles strictly inside a painted box
__device__ __forceinline__ uint32_t ami_inner_of(const AmiRegs& s, int lane)
{
    uint32_t inner = 0;
    // only the PAINTED boxes (lane b holds box b): a ballot and a loop over its set bits, ascending like the loop over all boxes
    // it replaces -- a random agent's game has none or a few of the board's boxes painted, and nine waves per frame run this
    uint64_t painted = __ballot(lane < s.f[A_N_BOXES] && (s.bflags & 1u));
    while (painted) {
        const int b = (int)__builtin_ctzll(painted);
        painted &= painted - 1;
        const uint32_t g = bcast(s.bgeom, b);
        const int tl_tx = g & 255, tl_ty = (g >> 8) & 255, br_tx = (g >> 16) & 255, br_ty = (g >> 24) & 255;
        if (lane > tl_ty && lane < br_ty && br_tx - tl_tx >= 2) {
            const int lo = tl_tx + 1, hi = br_tx - 1;   // inclusive
            if (lo < 32) {
                const int h2 = hi > 31 ? 31 : hi;
                if (h2 >= lo) inner |= (h2 - lo + 1 >= 32 ? ~0u : ((1u << (h2 - lo + 1)) - 1u)) << lo;
            }
        }
    }
    return inner;
}

// lane = mover slot
__device__ __forceinline__ bool ami_mover_shown(const AmiRegs& s, int lane) { return lane == PLAYER_SLOT || (lane < s.f[A_N_ENEMIES] && !s.mv[M_CAUGHT]); }
__device__ __forceinline__ int ami_mover_px(int world, int origin) { return origin + world_to_px(world) - 1; }

// 4-bit digits: score 10^4..10^0 (bits 0..19), lives (20..23), jumps (24..27), level (28..31)
__device__ __forceinline__ uint32_t ami_hud_word(const int32_t* f)
{
    int sc = f[A_SCORE];
    if (sc < 0) sc = 0;
    sc %= 100000;
    int lv = f[A_LIVES];
    lv = lv < 0 ? 0 : lv > 9 ? 9 : lv;
    int jp = f[A_JUMPS];
    jp = jp < 0 ? 0 : jp > 9 ? 9 : jp;
    int le = f[A_LEVEL];
    if (le < 0) le = 0;
    le %= 10;
    uint32_t hud = 0;
    int div = 10000;
#pragma unroll
    for (int q = 0; q < 5; q++) { hud |= (uint32_t)((sc / div) % 10) << (4 * q); div /= 10; }
    return hud | ((uint32_t)lv << 20) | ((uint32_t)jp << 24) | ((uint32_t)le << 28);
}

// Everything one wave needs to paint scanlines of one env; lane l makes pixels 4l..4l+3 of each scanline (160 px = 40
// lanes).  In the board band a lane's 4 pixels are exactly one tile (tile = 4x5 px, board origin x = 16).  Built once
// per frame by setup(); paint_row() then composes one scanline (board, movers in index order then the player, HUD).
template <int C>
struct AmiPainter {
    typedef AmiDev Dev;
    static constexpr int W = TBX_AMI_W, H = TBX_AMI_H, NG = 1;
    static constexpr bool FAST_ROWS = false;      // (agent_fused_wave: no scanline class with sums known without painting)
    static constexpr bool SPARSE_ROWS = false;    // (agent_fused_wave: nearly every scanline is busy but most repeat the one above -- the walk over the CHANGES)
    static __device__ __forceinline__ uint32_t fast_row_word(const uint32_t*, int) { return 0u; }
    __device__ __forceinline__ void fast_init(const ColTaps&, const ColTaps&, bool, bool) {}
    __device__ __forceinline__ bool fast_ready(int) const { return false; }
    __device__ __forceinline__ void fast_sums(int, const ColTaps&, const ColTaps&, bool, bool, uint32_t&, uint32_t&) const {}
    enum { CLS_BOARD, CLS_MOVER, CLS_HUD, NCLS, SLOT_EDGE = NCLS, NLDS };
    static constexpr int BOARD_Y1 = TBX_AMI_BOARD_OY + BH * TBX_AMI_TILE_PH;
    AmiRegs s;
    int lane, x0, tx;
    bool active, in_board_x, m_on;
    uint32_t inner;                         // lane = board row: tiles strictly inside a painted box
    int m_x0, m_y0;                         // lane = mover slot: screen rect origin
    uint32_t hud[4];                        // bit 3*r = lit in glyph row r
    uint32_t c_bg, c_inner, c_painted, c_unpainted, c_enemy, c_player;
    uint64_t mv_rows[4];                    // scanlines crossed by a mover (wave-uniform)
    uint64_t busy[4];                       // scanlines that are not plain background
    uint64_t rep[4];                        // scanlines that paint exactly as the one above (same tile / glyph row, no mover edge)
    mutable int ty_cached;
    mutable uint32_t board_col;             // this lane's tile colour in tile row ty_cached

    // cls: [NCLS][8] dwords of LDS private to this wave
    __device__ __forceinline__ void setup(const AmiDev& d, int env, int lane_, uint32_t* cls)
    {
        lane = lane_;
        const AmiTables& t = *d.tab;
        ami_load(d, env, lane, s);
        inner = ami_inner_of(s, lane);
        m_on = ami_mover_shown(s, lane);
        m_x0 = ami_mover_px(s.mv[M_X], TBX_AMI_BOARD_OX); m_y0 = ami_mover_px(s.mv[M_Y], TBX_AMI_BOARD_OY);
        finish_setup(t, ami_hud_word(s.f), cls);
    }

    // hudw: 4-bit digits, score 10^4..10^0 (bits 0..19), lives, jumps, level
    __device__ __forceinline__ void finish_setup(const AmiTables& t, uint32_t hudw, uint32_t* cls)
    {
        x0 = lane * 4;
        active = x0 < W;
        tx = lane - TBX_AMI_BOARD_OX / 4;
        in_board_x = tx >= 0 && tx < BW;
#pragma unroll
        for (int i = 0; i < 4; i++) hud[i] = 0;
        {
            const int hud_x0[8] = {20, 28, 36, 44, 52, 84, 108, 132};
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t glyph = tbx_digit_glyph((hudw >> (4 * q)) & 15u);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int dx = x0 + i - hud_x0[q];
                    if (dx >= 0 && dx < 6) hud[i] = (glyph >> (dx >> 1)) & 0x1249u;
                }
            }
        }
        // palette through pix_of<C>() once; the scanline loop only moves finished pixel values
        c_bg = pix_of<C>(t.bg); c_inner = pix_of<C>(t.inner); c_painted = pix_of<C>(t.painted);
        c_unpainted = pix_of<C>(t.unpainted); c_enemy = pix_of<C>(t.enemy); c_player = pix_of<C>(t.player);
        ty_cached = -1;
        board_col = c_bg;

        // scanline masks per class in LDS: every mover lane ORs its rows; board band and HUD rows are fixed
        for (int i = lane; i < NLDS * 8; i += 64) cls[i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t w = m_on ? row_range_bits(m_y0, (long)m_y0 + TBX_AMI_MOVER_H, k) : 0ull;
            if ((uint32_t)w) atomicOr(&cls[CLS_MOVER * 8 + 2 * k], (uint32_t)w);
            if ((uint32_t)(w >> 32)) atomicOr(&cls[CLS_MOVER * 8 + 2 * k + 1], (uint32_t)(w >> 32));
            // the scanlines where a mover starts, and the first one below it
            const uint64_t we = m_on ? (row_range_bits(m_y0, (long)m_y0 + 1, k) | row_range_bits((long)m_y0 + TBX_AMI_MOVER_H, (long)m_y0 + TBX_AMI_MOVER_H + 1, k)) : 0ull;
            if ((uint32_t)we) atomicOr(&cls[SLOT_EDGE * 8 + 2 * k], (uint32_t)we);
            if ((uint32_t)(we >> 32)) atomicOr(&cls[SLOT_EDGE * 8 + 2 * k + 1], (uint32_t)(we >> 32));
            if (lane == 0) {
                const uint64_t wb = row_range_bits(TBX_AMI_BOARD_OY, BOARD_Y1, k), wh = row_range_bits(TBX_AMI_HUD_Y, TBX_AMI_HUD_Y + 10, k);
                cls[CLS_BOARD * 8 + 2 * k] = (uint32_t)wb; cls[CLS_BOARD * 8 + 2 * k + 1] = (uint32_t)(wb >> 32);
                cls[CLS_HUD * 8 + 2 * k] = (uint32_t)wh; cls[CLS_HUD * 8 + 2 * k + 1] = (uint32_t)(wh >> 32);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t mlo = __builtin_amdgcn_readfirstlane(cls[CLS_MOVER * 8 + 2 * k]), mhi = __builtin_amdgcn_readfirstlane(cls[CLS_MOVER * 8 + 2 * k + 1]);
            mv_rows[k] = (uint64_t)mlo | ((uint64_t)mhi << 32);
            busy[k] = mv_rows[k] | row_range_bits(TBX_AMI_BOARD_OY, BOARD_Y1, k) | row_range_bits(TBX_AMI_HUD_Y, TBX_AMI_HUD_Y + 10, k);
            const uint32_t elo = __builtin_amdgcn_readfirstlane(cls[SLOT_EDGE * 8 + 2 * k]), ehi = __builtin_amdgcn_readfirstlane(cls[SLOT_EDGE * 8 + 2 * k + 1]);
            uint64_t firsts = (uint64_t)elo | ((uint64_t)ehi << 32);     // + the first scanline of every tile row / glyph row
            for (int ty = 0; ty < BH; ty++) firsts |= row_range_bits(TBX_AMI_BOARD_OY + ty * TBX_AMI_TILE_PH, TBX_AMI_BOARD_OY + ty * TBX_AMI_TILE_PH + 1, k);
            for (int gr = 0; gr <= 5; gr++) firsts |= row_range_bits(TBX_AMI_HUD_Y + 2 * gr, TBX_AMI_HUD_Y + 2 * gr + 1, k);
            firsts |= row_range_bits(BOARD_Y1, BOARD_Y1 + 1, k);       // the first scanline below the board band
            rep[k] = busy[k] & ~firsts;
        }
        __builtin_amdgcn_wave_barrier();
    }

    // the classes whose entities differ between two states of one env (wave-uniform bit mask)
    static __device__ __forceinline__ uint32_t diff_classes(const AmiPainter& a, const AmiPainter& b)
    {
        uint32_t m = 0u;
        if (__ballot(a.s.trow != b.s.trow || a.inner != b.inner)) m |= 1u << CLS_BOARD;
        if (__ballot(a.m_on != b.m_on || (b.m_on && (a.m_x0 != b.m_x0 || a.m_y0 != b.m_y0)))) m |= 1u << CLS_MOVER;
        if (__ballot(a.hud[0] != b.hud[0] || a.hud[1] != b.hud[1] || a.hud[2] != b.hud[2] || a.hud[3] != b.hud[3])) m |= 1u << CLS_HUD;
        return (uint32_t)wave_uniform((int)m);
    }

    // one scanline as finished pixel values; mover_row: a mover crosses it (bit of mv_rows)
    __device__ __forceinline__ void paint_row(int y, bool mover_row, uint32_t (&px)[4]) const
    {
#pragma unroll
        for (int i = 0; i < 4; i++) px[i] = c_bg;
        const int by = y - TBX_AMI_BOARD_OY;
        if (by >= 0 && by < BH * TBX_AMI_TILE_PH) {
            const int ty = by / TBX_AMI_TILE_PH;
            if (ty != ty_cached) {                       // five scanlines share a tile row
                ty_cached = ty;
                const uint64_t row = row_of(s, ty);
                const uint32_t inn = bcast(inner, ty);
                board_col = c_bg;
                if (in_board_x) {
                    const int tag = (int)((row >> (2 * tx)) & 3ull);
                    if (tag == TBX_TILE_EMPTY) { if ((inn >> tx) & 1u) board_col = c_inner; }
                    else board_col = tag == TBX_TILE_PAINTED ? c_painted : c_unpainted;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) px[i] = board_col;
        }
        // movers: enemies in index order, then the player
        if (mover_row) {
            uint64_t m = __ballot(m_on && y >= m_y0 && y < m_y0 + TBX_AMI_MOVER_H);
            const bool player_here = (m >> PLAYER_SLOT) & 1;
            m &= (1ull << PLAYER_SLOT) - 1;
            while (m) {
                const int src = (int)__builtin_ctzll(m);
                m &= m - 1;
                const int sx = bcast(m_x0, src);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (x0 + i >= sx && x0 + i < sx + TBX_AMI_MOVER_W) px[i] = c_enemy;
            }
            if (player_here) {
                const int sx = bcast(m_x0, PLAYER_SLOT);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (x0 + i >= sx && x0 + i < sx + TBX_AMI_MOVER_W) px[i] = c_player;
            }
        }
        if (y >= TBX_AMI_HUD_Y && y < TBX_AMI_HUD_Y + 10) {
            const int gr = ((y - TBX_AMI_HUD_Y) >> 1) * 3;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if ((hud[i] >> gr) & 1u) px[i] = c_player;
        }
    }
};

struct AmiGrayPainter : AmiPainter<1> {
    static __device__ __forceinline__ uint32_t diff_classes(const AmiGrayPainter& a, const AmiGrayPainter& b) { return AmiPainter<1>::diff_classes(a, b); }
    // the background colour comes from the config table, so the blank value is per engine, not a constant
    __device__ __forceinline__ uint32_t blank_dword() const { return c_bg * 0x01010101u; }
    __device__ __forceinline__ void row_dwords(int y, uint32_t (&v)[1]) const
    {
        const int wi = y >> 6;
        const uint64_t mw = sel4(wi, mv_rows[0], mv_rows[1], mv_rows[2], mv_rows[3]);
        uint32_t px[4];
        paint_row(y, (mw >> (y & 63)) & 1ull, px);
        v[0] = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
    }
};

// units part, part + split, ... of one env's frame from a painter that has been set up, on one wave: the body of
// ami_render_kernel and of the resident single-env kernel's paint request.  Background-only units are stored directly.
template <int C>
__device__ __forceinline__ void ami_paint_units(const AmiPainter<C>& p, uint8_t* __restrict__ frame, int env, int lane,
                                                const RowStager<C, TBX_AMI_W, AMI_UNIT_ROWS>& st, int part, int split)
{
    constexpr int H = TBX_AMI_H;
    using Stager = RowStager<C, TBX_AMI_W, AMI_UNIT_ROWS>;
    constexpr int NUNITS = (H + AMI_UNIT_ROWS - 1) / AMI_UNIT_ROWS;
    const int u0 = split > 1 ? 0 : (int)(((uint32_t)env * 7u) % (uint32_t)NUNITS);
    for (int k = part; k < NUNITS; k += split) {
        int u = u0 + k;
        if (u >= NUNITS) u -= NUNITS;
        const int y_first = u * AMI_UNIT_ROWS;
        const uint32_t mv_chunk = row_mask_chunk<AMI_UNIT_ROWS>(p.mv_rows, y_first);
        const uint32_t busy_chunk = row_mask_chunk<AMI_UNIT_ROWS>(p.busy, y_first);
        const int rows_here = H % AMI_UNIT_ROWS == 0 ? AMI_UNIT_ROWS : min(AMI_UNIT_ROWS, H - y_first);
        if (busy_chunk == 0 && C != 4 && rows_here == AMI_UNIT_ROWS) {   // background only: no staging (RGBA: staged is faster)
            Stager::fill_unit(frame + (size_t)u * Stager::UNIT_BYTES, lane, p.c_bg);
            continue;
        }
#pragma unroll 1
        for (int r = 0; r < rows_here; r++) {
            uint32_t px[4];
            p.paint_row(y_first + r, (mv_chunk >> r) & 1u, px);
            if (p.active) st.put4p(r, lane, px[0], px[1], px[2], px[3]);
        }
        st.flush(frame + (size_t)u * Stager::UNIT_BYTES, lane, rows_here);
    }
}

// One wave rasterises one env; AMI_UNIT_ROWS scanlines are staged in LDS and flushed as 16-byte stores, background-only
// units are stored directly.
// RGB launches run fastest with SIX waves per SIMD (the kernel's registers and LDS allow eight): builds held to 4 / 5 / 6 / 7 / 8
// measured 1.68 / 1.47 / 1.33 / 1.52-1.62 / 1.53-1.59 ms per launch at 65 536 envs (scripts/ubench/rate_addr).  How many waves
// the frame stores want in flight depends on how much of its life a wave spends storing: with the painter's set-up as it was
// before ami_inner_of looped over the painted boxes only, eight was best (1.37 ms), and every cut in set-up work made the
// eight-wave launch SLOWER -- more of the waves in flight are then storing at any time, and the frames' write locality in HBM
// goes (the same effect that puts Breakout's RGB launches at five, breakout.hip).  Gray and RGBA launches: as the registers allow.
template <int C, bool ALT>
__device__ __forceinline__ void ami_render_body(const AmiDev& d, uint8_t* out, int first_env, int count, int split, const AmiDev& d_alt,
                                                const uint8_t* __restrict__ pick_alt)
{
    constexpr int W = TBX_AMI_W, H = TBX_AMI_H;
    using Stager = RowStager<C, W, AMI_UNIT_ROWS>;
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[TBX_WAVES_PER_BLOCK * Stager::UNIT_BYTES];
    __shared__ uint32_t lds_mask[TBX_WAVES_PER_BLOCK][AmiPainter<C>::NLDS * 8];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wid = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + wave);   // `split` waves share a frame (see breakout.hip)
    const int rel = wid / split, part = wid - rel * split;
    if (rel >= count) return;
    const int env = first_env + rel;
    Stager st{lds_all + wave * Stager::UNIT_BYTES};
    AmiPainter<C> p;
    // (agent layer, generic path: flagged envs are painted from d_alt)
    AmiDev src = d;                                             // by VALUE: a select between references to kernel arguments puts both into scratch
    if (ALT && pick_alt && wave_uniform((int)pick_alt[env])) src = d_alt;   // (ALT: the agent layer's generic path only)
    p.setup(src, env, lane, lds_mask[wave]);

    ami_paint_units<C>(p, out + (size_t)rel * H * W * C, env, lane, st, part, split);
}
template <int C, bool ALT>
__global__ __launch_bounds__(TBX_BLOCK) __attribute__((amdgpu_waves_per_eu(6, 6))) void ami_render_kernel_w6(AmiDev d, uint8_t* out, int first_env, int count,
                                                                                                         int split, AmiDev d_alt, const uint8_t* __restrict__ pick_alt)
{
    ami_render_body<C, ALT>(d, out, first_env, count, split, d_alt, pick_alt);
}
template <int C, bool ALT>
__global__ __launch_bounds__(TBX_BLOCK) void ami_render_kernel(AmiDev d, uint8_t* out, int first_env, int count, int split, AmiDev d_alt,
                                                               const uint8_t* __restrict__ pick_alt)
{
    ami_render_body<C, ALT>(d, out, first_env, count, split, d_alt, pick_alt);
}

// ------------------------------------------------------------------ resident single-env form (tbx_serve_loop, tbx_common.hpp)
//
// One wave, env 0: steps on request and, when the request asks for it, rasterises the env straight into the engine's mapped
// pinned frame buffer (ToyboxBaseEnv.step = apply_ale_action + get_state without a launch, a copy or a synchronisation).
template <int C>
__device__ __forceinline__ void ami_serve_paint(const AmiDev& d, uint8_t* frame, int lane, uint8_t* lds, uint32_t* cls, int part, int split)
{
    const RowStager<C, TBX_AMI_W, AMI_UNIT_ROWS> st{lds};
    AmiPainter<C> p;
    p.setup(d, 0, lane, cls);
    ami_paint_units<C>(p, frame, 0, lane, st, part, split);
}

__global__ __launch_bounds__(64 * TBX_SERVE_WAVES) void ami_serve_kernel(AmiDev d, TbxServeCtl* ctl)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[TBX_SERVE_WAVES][RowStager<4, TBX_AMI_W, AMI_UNIT_ROWS>::UNIT_BYTES];
    __shared__ uint32_t cls[TBX_SERVE_WAVES][AmiPainter<1>::NLDS * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    tbx_serve_loop(ctl, lane, [&](const ActionSource& src, uint32_t flags) { ami_step_body<false>(d, d, d, src, flags, 0, lane); },
                   [&](int channels, uint8_t* frame, int part, int split) {
                       switch (channels) {
                       case 1: ami_serve_paint<1>(d, frame, lane, lds[wave], cls[wave], part, split); break;
                       case 3: ami_serve_paint<3>(d, frame, lane, lds[wave], cls[wave], part, split); break;
                       default: ami_serve_paint<4>(d, frame, lane, lds[wave], cls[wave], part, split); break;
                       }
                       return true;
                   },
                   d.reward, d.done, d.lives_out, d.score_out, d.err_flag);
}

// ------------------------------------------------------------------ fused agent observation
//
// max(frame A, frame B) -> gray -> area warp -> frame stack without the two full-resolution gray frames ever reaching
// HBM: agent_fused_wave (agent_device.hpp) with two AmiGrayPainters in one wave per env.
template <int S>
// Held to FIVE waves per SIMD (the LDS of a block allows five): with the newest-plane output and the 16-byte stack commit of round 5
// the depth-4 instantiation asked for 99-101 VGPRs -- four waves -- and the agent step at 65 536 envs lost 3-7 % against round 4;
// at 96 VGPRs it spills 16-52 bytes per lane and runs 3.5 % AHEAD of round 4 (same box: Amidar 1.848 / 1.987 / 1.911 ms pinned /
// unpinned / round 4, GridWorld 1.053 / 1.156 / 1.091).
__global__ __launch_bounds__(TBX_BLOCK) __attribute__((amdgpu_waves_per_eu(5))) void ami_agent_warp_kernel(AmiDev dLive, AmiDev dA, AmiDev dB, AgentWarpArgs a, int n)
{
    __shared__ AgentFusedLds<AmiGrayPainter> lds[TBX_WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int env = wave_uniform(a.first + blockIdx.x * TBX_WAVES_PER_BLOCK + wave);
    if (env >= a.end) return;
    AmiGrayPainter pa, pb;
    agent_fused_wave<S, AmiGrayPainter>(pa, pb, dLive, dA, dB, a, env, lane, lds[wave]);
}

// ------------------------------------------------------------------ state pack / unpack, scalars

__device__ __forceinline__ void mover_out(const AmiRegs& s, tbx_amidar_mover_t& m)
{
    m.x = s.mv[M_X]; m.y = s.mv[M_Y]; m.speed = s.mv[M_SPEED]; m.step_tx = s.mv[M_STEP_TX]; m.step_ty = s.mv[M_STEP_TY];
    m.n_history = s.mv[M_NHIST];
    for (int i = 0; i < TBX_AMI_MAX_HISTORY; i++) m.history[i] = s.mv[M_HIST0 + i];
    m.caught = s.mv[M_CAUGHT];
    int32_t* a = &m.ai.kind;
    for (int k = 0; k < 14; k++) a[k] = s.mv[M_KIND + k];
}

__global__ void ami_pack_kernel(AmiDev d, int env, tbx_amidar_state_t* out)
{
    env += blockIdx.x;      // one block per env of the requested range
    out += blockIdx.x;
    const int lane = threadIdx.x & 63;
    AmiRegs s;
    ami_load(d, env, lane, s);
    const int32_t* f = s.f;
    if (lane == 0) {
        out->rand[0] = s.rng.s0; out->rand[1] = s.rng.s1;
        out->score = f[A_SCORE]; out->lives = f[A_LIVES]; out->level = f[A_LEVEL];
        out->jumps = f[A_JUMPS]; out->jump_timer = f[A_JUMP_TIMER]; out->chase_timer = f[A_CHASE_TIMER];
        out->n_enemies = f[A_N_ENEMIES]; out->n_boxes = f[A_N_BOXES]; out->n_chase_junctions = f[A_N_CHASE];
        for (int k = 0; k < TBX_AMI_MAX_CHASE_J; k++) out->chase_junctions[k] = f[A_CJ0 + k];
    }
    if (lane == PLAYER_SLOT) mover_out(s, out->player);
    if (lane < TBX_AMI_MAX_ENEMIES) {
        if (lane < f[A_N_ENEMIES]) mover_out(s, out->enemies[lane]);
        else memset(&out->enemies[lane], 0, sizeof(tbx_amidar_mover_t));
    }
    {
        tbx_amidar_box_t b;
        memset(&b, 0, sizeof b);
        if (lane < f[A_N_BOXES]) {
            b.tl_tx = s.bgeom & 255; b.tl_ty = (s.bgeom >> 8) & 255; b.br_tx = (s.bgeom >> 16) & 255; b.br_ty = (s.bgeom >> 24) & 255;
            b.painted = s.bflags & 1u; b.triggers_chase = (s.bflags >> 1) & 1u;
        }
        out->boxes[lane] = b;
    }
    if (lane < BH)
        for (int x = 0; x < BW; x++) out->tiles[lane][x] = (uint8_t)((s.trow >> (2 * x)) & 3ull);
}

__device__ __forceinline__ void mover_in(const tbx_amidar_mover_t& m, AmiRegs& s)
{
    s.mv[M_X] = m.x; s.mv[M_Y] = m.y; s.mv[M_SPEED] = m.speed; s.mv[M_STEP_TX] = m.step_tx; s.mv[M_STEP_TY] = m.step_ty;
    s.mv[M_NHIST] = m.n_history;
    for (int i = 0; i < TBX_AMI_MAX_HISTORY; i++) s.mv[M_HIST0 + i] = m.history[i];
    s.mv[M_CAUGHT] = m.caught;
    const int32_t* a = &m.ai.kind;
    for (int k = 0; k < 14; k++) s.mv[M_KIND + k] = a[k];
}

__global__ void ami_unpack_kernel(AmiDev d, int env, const tbx_amidar_state_t* in)
{
    env += blockIdx.x;
    in += blockIdx.x;
    const int lane = threadIdx.x & 63;
    AmiRegs s;
    int32_t* f = s.f;
    s.rng.s0 = in->rand[0]; s.rng.s1 = in->rand[1];
    f[A_SCORE] = in->score; f[A_LIVES] = in->lives; f[A_LEVEL] = in->level;
    f[A_JUMPS] = in->jumps; f[A_JUMP_TIMER] = in->jump_timer; f[A_CHASE_TIMER] = in->chase_timer;
    f[A_N_ENEMIES] = in->n_enemies; f[A_N_BOXES] = in->n_boxes; f[A_N_CHASE] = in->n_chase_junctions;
    for (int k = 0; k < TBX_AMI_MAX_CHASE_J; k++) f[A_CJ0 + k] = in->chase_junctions[k];
    for (int i = 0; i < NMF; i++) s.mv[i] = 0;
    const int slot = lane & 15;
    if (slot == PLAYER_SLOT) mover_in(in->player, s);
    else if (slot < in->n_enemies) mover_in(in->enemies[slot], s);
    s.bgeom = 0; s.bflags = 0;
    if (lane < in->n_boxes) {
        const tbx_amidar_box_t& b = in->boxes[lane];
        s.bgeom = (uint32_t)(b.tl_tx & 255) | ((uint32_t)(b.tl_ty & 255) << 8) | ((uint32_t)(b.br_tx & 255) << 16) | ((uint32_t)(b.br_ty & 255) << 24);
        s.bflags = (b.painted ? 1u : 0u) | (b.triggers_chase ? 2u : 0u) | 4u;
    }
    s.trow = 0;
    if (lane < BH)
        for (int x = 0; x < BW; x++) s.trow |= (uint64_t)(in->tiles[lane][x] & 3) << (2 * x);
    ami_store(d, env, lane, s);
}

// ------------------------------------------------------------------ batched interventions (tbx_edit / tbx_reduce)
//
// AmidarIntervention's helper methods (toybox/interventions/amidar.py:360-615) over the batch: one thread per env.  Mover fields
// are written to the env-major table AND to the struct-of-arrays mirror of the per-frame fields (AmiDev::mh), like every writer.

__device__ __forceinline__ int ami_floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

__device__ __forceinline__ void ami_mover_write(const AmiDev& d, int env, int field, int slot, int v)
{
    d.movers[((size_t)env * NMF + field) * 16 + slot] = v;
    const int h = ami_hot_row(field);
    if (h >= 0) d.mh[((size_t)h * MSLOTS + slot) * (size_t)d.n + env] = v;
}
__device__ __forceinline__ int ami_mover_read(const AmiDev& d, int env, int field, int slot) { return d.movers[((size_t)env * NMF + field) * 16 + slot]; }

__device__ __forceinline__ int ami_tile_tag(const AmiDev& d, int env, int tx, int ty)
{
    if (tx < 0 || ty < 0 || tx >= BW || ty >= BH) return -1;
    return (int)((d.tiles[(size_t)env * 32 + ty] >> (2 * tx)) & 3ull);
}

// The counter-RNG forms of the reference's `random`-driven helpers (toybox_amd.h).  Candidates of a tile draw: tiles row by row
// whose tag is in tag_mask and, with min_dist > 0, for which not every enemy is nearer than min_dist (manhattan, in tiles).
struct AmiTilePick { int tx, ty, tag, count; };
__device__ AmiTilePick ami_random_tile(const AmiDev& d, int env, uint32_t seed, uint32_t draw, uint32_t env_offset, uint32_t tag_mask, int min_dist)
{
    const size_t N = (size_t)d.n;
    const int ne = d.sc[(size_t)A_N_ENEMIES * N + env];
    int ex[TBX_AMI_MAX_ENEMIES], ey[TBX_AMI_MAX_ENEMIES];
    for (int i = 0; i < TBX_AMI_MAX_ENEMIES; i++) {
        ex[i] = i < ne ? ami_floor_div(ami_mover_read(d, env, M_X, i), TBX_AMI_TILE_WX) : 0;
        ey[i] = i < ne ? ami_floor_div(ami_mover_read(d, env, M_Y, i), TBX_AMI_TILE_WY) : 0;
    }
    auto accepted = [&](int tx, int ty) {
        if (!((tag_mask >> ami_tile_tag(d, env, tx, ty)) & 1u)) return false;
        if (min_dist <= 0) return true;
        bool all_near = true;                                 // `not all(d < min for every enemy)`; no enemies: all([]) is True
        for (int i = 0; i < ne; i++) all_near = all_near && (abs(ex[i] - tx) + abs(ey[i] - ty) < min_dist);
        return !all_near;
    };
    AmiTilePick p{-1, -1, -1, 0};
    for (int ty = 0; ty < BH; ty++)
        for (int tx = 0; tx < BW; tx++) p.count += accepted(tx, ty) ? 1 : 0;
    if (p.count == 0) return p;
    const uint64_t r = tbx_splitmix64((uint64_t)seed ^ ((uint64_t)(env_offset + (uint32_t)env) << 32) ^ (uint64_t)draw);
    int k = (int)(r % (uint64_t)p.count);
    for (int ty = 0; ty < BH && p.tx < 0; ty++)
        for (int tx = 0; tx < BW; tx++)
            if (accepted(tx, ty) && k-- == 0) { p.tx = tx; p.ty = ty; p.tag = ami_tile_tag(d, env, tx, ty); break; }
    return p;
}

__global__ __launch_bounds__(256) void ami_edit_kernel(AmiDev d, int op, TbxEditArgs a, const uint8_t* __restrict__ mask)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.n || (mask && !mask[env])) return;
    const size_t N = (size_t)d.n;
    switch (op) {
    case TBX_EDIT_SET_LIVES: d.sc[(size_t)A_LIVES * N + env] = a.geti(env, 0); break;
    case TBX_EDIT_SET_SCORE: d.sc[(size_t)A_SCORE * N + env] = a.geti(env, 0); break;
    case TBX_EDIT_SET_LEVEL: d.sc[(size_t)A_LEVEL * N + env] = a.geti(env, 0); break;
    case TBX_EDIT_AMI_JUMPS: d.sc[(size_t)A_JUMPS * N + env] = a.geti(env, 0); break;
    case TBX_EDIT_AMI_TIMERS:
        if (a.geti(env, 0) >= 0) d.sc[(size_t)A_JUMP_TIMER * N + env] = a.geti(env, 0);
        if (a.geti(env, 1) >= 0) d.sc[(size_t)A_CHASE_TIMER * N + env] = a.geti(env, 1);
        break;
    case TBX_EDIT_AMI_TILE: {
        const int tx = a.geti(env, 0), ty = a.geti(env, 1), tag = a.geti(env, 2) & 3;
        if (tx >= 0 && ty >= 0 && tx < BW && ty < BH) {
            uint64_t& w = d.tiles[(size_t)env * 32 + ty];
            w = (w & ~(3ull << (2 * tx))) | ((uint64_t)tag << (2 * tx));
        }
        break;
    }
    case TBX_EDIT_AMI_ENEMY_AI: {
        const int slot = a.geti(env, 0);
        if (slot >= 0 && slot < d.sc[(size_t)A_N_ENEMIES * N + env])
            for (int k = 0; k < 14; k++) ami_mover_write(d, env, M_KIND + k, slot, a.geti(env, 1 + k));
        break;
    }
    case TBX_EDIT_AMI_PLAYER_TILE:
        ami_mover_write(d, env, M_X, PLAYER_SLOT, a.geti(env, 0) * TBX_AMI_TILE_WX);
        ami_mover_write(d, env, M_Y, PLAYER_SLOT, a.geti(env, 1) * TBX_AMI_TILE_WY);
        break;
    case TBX_EDIT_AMI_PLAYER_RANDOM_START: {
        const AmiTilePick p = ami_random_tile(d, env, a.getu(env, 0), a.getu(env, 1), a.getu(env, 2), 0xFu, a.geti(env, 3));
        if (p.count > 0) {
            ami_mover_write(d, env, M_X, PLAYER_SLOT, p.tx * TBX_AMI_TILE_WX);
            ami_mover_write(d, env, M_Y, PLAYER_SLOT, p.ty * TBX_AMI_TILE_WY);
        }
        break;
    }
    default: break;
    }
}

__global__ __launch_bounds__(256) void ami_reduce_kernel(AmiDev d, int query, TbxEditArgs a, double* __restrict__ out, int width)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.n) return;
    const size_t N = (size_t)d.n;
    double* o = out + (size_t)env * width;
    const int ne = d.sc[(size_t)A_N_ENEMIES * N + env];
    const int ptx = ami_floor_div(ami_mover_read(d, env, M_X, PLAYER_SLOT), TBX_AMI_TILE_WX);
    const int pty = ami_floor_div(ami_mover_read(d, env, M_Y, PLAYER_SLOT), TBX_AMI_TILE_WY);
    auto distances = [&](int tx, int ty) {
        for (int i = 0; i < TBX_AMI_MAX_ENEMIES; i++) {
            if (i >= ne) { o[i] = -1.0; continue; }
            const int ex = ami_floor_div(ami_mover_read(d, env, M_X, i), TBX_AMI_TILE_WX), ey = ami_floor_div(ami_mover_read(d, env, M_Y, i), TBX_AMI_TILE_WY);
            o[i] = abs(ex - tx) + abs(ey - ty);
        }
    };
    switch (query) {
    case TBX_QUERY_AMI_MODE: o[0] = d.sc[(size_t)A_JUMP_TIMER * N + env]; o[1] = d.sc[(size_t)A_CHASE_TIMER * N + env]; break;
    case TBX_QUERY_AMI_ANY_CAUGHT: {
        int any = 0;
        for (int i = 0; i < ne; i++) any |= ami_mover_read(d, env, M_CAUGHT, i) != 0;
        o[0] = any;
        break;
    }
    case TBX_QUERY_AMI_TILE: o[0] = ami_tile_tag(d, env, a.geti(env, 0), a.geti(env, 1)); break;
    case TBX_QUERY_AMI_COUNT_TILES: {
        const int tag = a.geti(env, 0);
        int c = 0;
        for (int ty = 0; ty < BH; ty++)
            for (int tx = 0; tx < BW; tx++) c += ami_tile_tag(d, env, tx, ty) == tag;
        o[0] = c;
        break;
    }
    case TBX_QUERY_AMI_TILES_MASK: {
        const uint32_t tm = a.getu(env, 0);
        int c = 0;
        for (int ty = 0; ty < BH; ty++) {
            uint32_t bits = 0;
            for (int tx = 0; tx < BW; tx++)
                if ((tm >> ami_tile_tag(d, env, tx, ty)) & 1u) { bits |= 1u << tx; c++; }
            o[ty] = bits;
        }
        o[31] = c;
        break;
    }
    case TBX_QUERY_AMI_RANDOM_TILE: {
        const AmiTilePick p = ami_random_tile(d, env, a.getu(env, 0), a.getu(env, 1), a.getu(env, 2), a.getu(env, 3), a.geti(env, 4));
        o[0] = p.tx; o[1] = p.ty; o[2] = p.tag; o[3] = p.count;
        break;
    }
    case TBX_QUERY_AMI_RANDOM_DIR: {
        const int tx = a.geti(env, 3), ty = a.geti(env, 4);
        const int nx[4] = {tx, tx, tx - 1, tx + 1}, ny[4] = {ty - 1, ty + 1, ty, ty};     // TBX_DIR_UP, DOWN, LEFT, RIGHT
        int valid = 0;
        for (int k = 0; k < 4; k++) valid += ami_tile_tag(d, env, nx[k], ny[k]) > TBX_TILE_EMPTY;
        int dir = -1;
        if (valid) {
            const uint64_t r = tbx_splitmix64((uint64_t)a.getu(env, 0) ^ ((uint64_t)(a.getu(env, 2) + (uint32_t)env) << 32) ^ (uint64_t)a.getu(env, 1));
            int pick = (int)(r % (uint64_t)valid);
            for (int k = 0; k < 4 && dir < 0; k++)
                if (ami_tile_tag(d, env, nx[k], ny[k]) > TBX_TILE_EMPTY && pick-- == 0) dir = k;
        }
        o[0] = dir; o[1] = valid;
        break;
    }
    case TBX_QUERY_AMI_ADJACENT: {
        const int tx = a.geti(env, 0), ty = a.geti(env, 1);
        o[0] = ami_tile_tag(d, env, tx, ty - 1); o[1] = ami_tile_tag(d, env, tx - 1, ty);
        o[2] = ami_tile_tag(d, env, tx + 1, ty); o[3] = ami_tile_tag(d, env, tx, ty + 1);
        break;
    }
    case TBX_QUERY_AMI_ENEMY_DISTANCES: distances(a.geti(env, 0), a.geti(env, 1)); break;
    case TBX_QUERY_AMI_PLAYER_TILE: o[0] = ptx; o[1] = pty; o[2] = ami_tile_tag(d, env, ptx, pty); break;
    case TBX_QUERY_AMI_PLAYER_ENEMY_DISTANCES: distances(ptx, pty); break;
    case TBX_QUERY_AMI_PLAYER_ON_PAINTED: o[0] = ami_tile_tag(d, env, ptx, pty) == TBX_TILE_PAINTED; break;
    case TBX_QUERY_AMI_PLAYER_NEAR_UNPAINTED: {
        const int radius = a.geti(env, 0);
        int near = 0, painted = 0;
        for (int ty = 0; ty < BH; ty++)
            for (int tx = 0; tx < BW; tx++) {
                const int tag = ami_tile_tag(d, env, tx, ty);
                if (abs(tx - ptx) + abs(ty - pty) < radius && tag != TBX_TILE_EMPTY) { near++; painted += tag == TBX_TILE_PAINTED; }
            }
        o[0] = painted != near;
        break;
    }
    default: break;
    }
}

__global__ void ami_scalars_kernel(AmiDev d, int32_t* score, int32_t* lives, int32_t* level)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    const size_t N = (size_t)d.n;
    if (score) score[i] = d.sc[(size_t)A_SCORE * N + i];
    if (lives) lives[i] = d.sc[(size_t)A_LIVES * N + i];
    if (level) level[i] = d.sc[(size_t)A_LEVEL * N + i];
}

// ------------------------------------------------------------------ host ops

struct AmiOps : GameOps {
    AmiDev d{};
    tbx_amidar_config_t cfg{};
    AmiTables tab{};
    AmiTables* tab_dev = nullptr;
    int height() const override { return TBX_AMI_H; }
    int width() const override { return TBX_AMI_W; }
    size_t state_size() const override { return sizeof(tbx_amidar_state_t); }
    size_t config_size() const override { return sizeof(tbx_amidar_config_t); }

    static bool walk(const tbx_amidar_config_t& k, int tx, int ty)
    {
        return tx >= 0 && ty >= 0 && tx < BW && ty < BH && k.board[ty][tx] != TBX_TILE_EMPTY;
    }

    // board-derived tables (boxes = maximal empty rectangles recorded by their track corners, row-major scan)
    int build_tables(tbx_engine* e, const tbx_amidar_config_t& k)
    {
        if (k.n_enemies < 0 || k.n_enemies > TBX_AMI_MAX_ENEMIES) return e->fail(TBX_E_UNSUPPORTED, "amidar: at most 8 enemies");
        AmiTables t;
        memset(&t, 0, sizeof t);
        for (int y = 0; y < BH; y++)
            for (int x = 0; x < BW; x++) t.board_rows[y] |= (uint64_t)(k.board[y][x] & 3) << (2 * x);
        for (int y = 0; y < BH; y++)
            for (int x = 0; x < BW; x++)
                if (k.board[y][x] == TBX_TILE_CHASE_MARKER && t.n_chase < TBX_AMI_MAX_CHASE_J) t.chase_j[t.n_chase++] = y * BW + x;
        for (int ty = 0; ty < BH - 1; ty++)
            for (int tx = 0; tx < BW - 1; tx++) {
                if (!walk(k, tx, ty) || !walk(k, tx + 1, ty) || !walk(k, tx, ty + 1) || walk(k, tx + 1, ty + 1)) continue;
                int x1 = tx + 1, y1 = ty + 1;
                while (x1 < BW && !walk(k, x1, ty + 1)) x1++;
                while (y1 < BH && !walk(k, tx + 1, y1)) y1++;
                if (x1 >= BW || y1 >= BH || t.n_boxes >= TBX_AMI_MAX_BOXES) continue;
                uint32_t fl = 4u;
                for (int q = 0; q < t.n_chase; q++)
                    if (t.chase_j[q] == ty * BW + tx) fl |= 2u;
                t.box_geom[t.n_boxes] = (uint32_t)tx | ((uint32_t)ty << 8) | ((uint32_t)x1 << 16) | ((uint32_t)y1 << 24);
                t.box_flags[t.n_boxes] = fl;
                t.n_boxes++;
            }
        t.n_enemies = k.n_enemies;
        for (int i = 0; i < k.n_enemies; i++) memcpy(t.ai[i], &k.enemies[i], sizeof(int32_t) * 14);
        t.player_start_tx = k.player_start_tx; t.player_start_ty = k.player_start_ty; t.player_hist0 = -1;
        t.start_lives = k.start_lives; t.start_jumps = k.start_jumps; t.jump_time = k.jump_time; t.chase_time = k.chase_time;
        t.box_bonus = k.box_bonus; t.chase_score_bonus = k.chase_score_bonus;
        t.bg = pack_color(k.bg_color); t.player = pack_color(k.player_color); t.unpainted = pack_color(k.unpainted_color);
        t.painted = pack_color(k.painted_color); t.enemy = pack_color(k.enemy_color); t.inner = pack_color(k.inner_painted_color);
        cfg = k;
        tab = t;
        return TBX_OK;
    }

    int upload(tbx_engine* e)
    {
        TBX_HIP(hipMemcpy(tab_dev, &tab, sizeof tab, hipMemcpyHostToDevice));
        return TBX_OK;
    }

    int init(tbx_engine* e, const void* cfg_pod, size_t cfg_size) override
    {
        static_assert(sizeof(tbx_amidar_ai_t) == 14 * sizeof(int32_t), "tbx_amidar_ai_t is 14 int32 fields");
        if (!cfg_pod || cfg_size != sizeof(tbx_amidar_config_t)) return e->fail(TBX_E_INVALID, "amidar: config size mismatch");
        tbx_amidar_config_t k;
        memcpy(&k, cfg_pod, sizeof k);
        int rc = build_tables(e, k);
        if (rc) return rc;
        const size_t N = (size_t)e->n;
        d.n = e->n;
        d.sim_rng = e->sim_rng; d.prev_score = e->prev_score; d.reward = e->reward; d.done = e->done;
        d.lives_out = e->lives_out; d.score_out = e->score_out; d.packed = e->packed; d.err_flag = e->err_flag;
        TBX_HIP(hipMalloc((void**)&d.rng, 2 * N * sizeof(uint64_t)));
        TBX_HIP(hipMalloc((void**)&d.sc, (size_t)ANF * N * sizeof(int32_t)));
        TBX_HIP(hipMalloc((void**)&d.tiles, N * 32 * sizeof(uint64_t)));
        TBX_HIP(hipMalloc((void**)&d.boxes, N * 128 * sizeof(uint32_t)));
        TBX_HIP(hipMalloc((void**)&d.movers, N * NMF * 16 * sizeof(int32_t)));
        TBX_HIP(hipMalloc((void**)&d.mh, N * NMH * MSLOTS * sizeof(int32_t)));
        TBX_HIP(hipMalloc((void**)&tab_dev, sizeof(AmiTables)));
        d.tab = tab_dev;
        return upload(e);
    }

    void destroy(tbx_engine*) override
    {
        hipFree(d.rng); hipFree(d.sc); hipFree(d.tiles); hipFree(d.boxes); hipFree(d.movers); hipFree(d.mh); hipFree(tab_dev);
        hipFree(dA.rng); hipFree(dA.sc); hipFree(dA.tiles); hipFree(dA.boxes); hipFree(dA.movers); hipFree(dA.mh);
        hipFree(dB.rng); hipFree(dB.sc); hipFree(dB.tiles); hipFree(dB.boxes); hipFree(dB.movers); hipFree(dB.mh);
    }

    int get_config(tbx_engine*, void* pod) override { memcpy(pod, &cfg, sizeof cfg); return TBX_OK; }
    int set_config(tbx_engine* e, const void* pod) override
    {
        tbx_amidar_config_t k;
        memcpy(&k, pod, sizeof k);
        int rc = build_tables(e, k);
        if (rc) return rc;
        return upload(e);
    }

    static dim3 grid_for(int count) { return dim3((count + TBX_WAVES_PER_BLOCK - 1) / TBX_WAVES_PER_BLOCK); }

    int new_game(tbx_engine* e, const uint8_t* mask_dev, hipStream_t s) override
    {
        hipLaunchKernelGGL(ami_new_game_kernel, grid_for(e->n), dim3(TBX_BLOCK), 0, s, d, mask_dev);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    // The rasteriser reads live state, so a step never runs beside or inside a rasteriser launch (GameOps::pipeline_ok and
    // render_step_fused stay false; tbx_render_step_synthetic is the two launches in stream order).  Both were built and measured
    // slower than stream order at every batch size: records behind the step on a second stream (round 3: the step took
    // 420-520 us beside a rasteriser instead of 45, the rasteriser 1.69-1.72 ms instead of 1.37) and the step's thread-per-env
    // form riding in the rasteriser's launch with the records written by the step itself (round 4, 80-VGPR budget of the
    // six-waves-per-SIMD launch: 1.61 against 1.47 ms per step at 65 536 envs, 0.160 against 0.106 at 4 096) -- the step is a chain
    // of dependent loads, and beside a kernel that saturates the memory system with stores every one of them queues.  A third
    // form needs no records at all: the WAVE-per-env step (one round of loads, then registers) as the first blocks of the
    // rasteriser's launch, writing the whole next state into a second state buffer that becomes the current one after the
    // launch.  Built, bit-identical, and no faster either: ms per step fused / two launches 0.1072 / 0.1085 at 4 096 envs,
    // 0.214 / 0.204 at 8 192, 0.421 / 0.391 at 16 384, 1.68 / 1.53 at 65 536 (five waves per SIMD, no spills; six: 0.209 at
    // 8 192, 1.62 at 65 536) -- a step wave holds a rasteriser wave's slot for its whole latency-bound life.  Removed.
    // Round 5, the one form left: the same wave-per-env step over two state buffers, but on the engine's STEP
    // STREAM beside the rasteriser of the previous frame (TBX_OPT_PIPELINE = 2 / 3, the fences of the pipelined mode; parity-green
    // in tests/test_gpu_paths.py's pipelined-mode tests).  scripts/pipeline_sweep.py amidar, one box, ms per step
    // stream order / value 2 / value 3: 0.0605 / 0.0695 / 0.0622 at 2 048 envs, 0.1085 / 0.1181 / 0.1130 at 4 096,
    // 0.200 / 0.212 / 0.2085 at 8 192, 0.3886 / 0.4324 / 0.4055 at 16 384 -- slower everywhere (the kill criterion was a gain of
    // 3 % at 4 096): beside a kernel that saturates the memory system with stores the step's loads and its full-state write-back
    // queue, and the rasteriser loses the slots the step's waves hold.  Removed; Amidar's step stays in stream order.
    void rebind_outputs(tbx_engine* e) override
    {
        d.reward = e->reward; d.done = e->done; d.lives_out = e->lives_out; d.score_out = e->score_out; d.packed = e->packed;
    }

    int step(tbx_engine* e, const ActionSource& src, uint32_t flags, hipStream_t s) override
    {
        int first = 0, count = e->n;
        if (src.single_env >= 0) { first = src.single_env; count = 1; }
        dA.tab = dB.tab = d.tab;
        // TBX_OPT_STEP_FORM: 2 = never, 1 = always, 0 = by batch size.  The thread form is one wave per 64 envs with a long
        // serial path per thread (~45 us whatever the batch), the wave form scales with the batch.  Measured in the step + render
        // loop (scripts/pipeline_sweep.py amidar, PS_STEP_FORM=1|2, one box): ms per step thread / wave form 0.246 / 0.215 at
        // 8 192 envs, 0.419 / 0.394 at 16 384, 0.601 / 0.591 at 24 576, 0.778 / 0.779 at 32 768, 1.130 / 1.168 at 49 152,
        // 1.483 / 1.546 at 65 536 (round 2 had put the switch at 16 384 from step-only timings)
        const int form = e->opt[TBX_OPT_STEP_FORM];
        const bool use_tpe = form == 2 ? false : form == 1 ? true : e->n >= 32768;
        if (use_tpe && src.single_env < 0) {
            // large batches: one THREAD per env (the wave-per-env form stays for small batches, single-env calls and the
            // in-kernel reset procedure)
            if (src.acc_reward || src.buf_valid || src.exec_flag || src.frames > 1) {    // an agent step's frames (never auto-reset)
                if (flags & TBX_STEP_AUTO_RESET) return e->fail(TBX_E_INVALID, "an agent step cannot auto-reset");
                hipLaunchKernelGGL(ami_step_tpe_kernel<true>, dim3((e->n + 63) / 64), dim3(64), 0, s, d, dA, dB, src, flags);
            } else
                TBX_LAUNCH_STEP(e, s, (ami_step_tpe_kernel<false>), dim3((e->n + 63) / 64), dim3(64), d, d, d, src, flags);
            TBX_HIP(hipGetLastError());
            return TBX_OK;
        }
        if (src.acc_reward || src.buf_valid || src.exec_flag || src.frames > 1) {    // an agent step's frames (never auto-reset)
            if (flags & TBX_STEP_AUTO_RESET) return e->fail(TBX_E_INVALID, "an agent step cannot auto-reset");
            hipLaunchKernelGGL(ami_agent_step_kernel, grid_for(count), dim3(TBX_BLOCK), 0, s, d, dA, dB, src, flags, first, count);
        } else
            if (src.single_env < 0) TBX_LAUNCH_STEP(e, s, ami_step_kernel, grid_for(count), dim3(TBX_BLOCK), d, src, flags, first, count);
            else hipLaunchKernelGGL(ami_step_kernel, grid_for(count), dim3(TBX_BLOCK), 0, s, d, src, flags, first, count);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    bool serve_paints() const override { return true; }
    int serve(tbx_engine* e, TbxServeCtl* ctl_dev, hipStream_t s) override
    {
        hipLaunchKernelGGL(ami_serve_kernel, dim3(1), dim3(64 * TBX_SERVE_WAVES), 0, s, d, ctl_dev);   // (paints on request: serve_paints)
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    // ---- agent layer: MaxAndSkipEnv's two-frame buffer is two snapshots of the dynamic SoA state per env
    AmiDev dA{}, dB{};
    bool agent_fused() const override { return true; }
    bool multi_frame_step() const override { return true; }
    bool agent_reset_supported() const override { return true; }

    int alloc_slot(tbx_engine* e, AmiDev& x)
    {
        if (x.sc) { x.tab = d.tab; return TBX_OK; }
        const size_t N = (size_t)e->n;
        x = d;
        x.sc = nullptr; x.tiles = nullptr; x.boxes = nullptr; x.movers = nullptr; x.rng = nullptr; x.mh = nullptr;
        TBX_HIP(hipMalloc((void**)&x.rng, 2 * N * sizeof(uint64_t)));
        TBX_HIP(hipMalloc((void**)&x.sc, (size_t)ANF * N * sizeof(int32_t)));
        TBX_HIP(hipMalloc((void**)&x.tiles, N * 32 * sizeof(uint64_t)));
        TBX_HIP(hipMalloc((void**)&x.boxes, N * 128 * sizeof(uint32_t)));
        TBX_HIP(hipMalloc((void**)&x.movers, N * NMF * 16 * sizeof(int32_t)));
        TBX_HIP(hipMalloc((void**)&x.mh, N * NMH * MSLOTS * sizeof(int32_t)));
        return TBX_OK;
    }

    int agent_prepare(tbx_engine* e) override
    {
        int rc = alloc_slot(e, dA);
        if (rc) return rc;
        return alloc_slot(e, dB);
    }

    int agent_warp(tbx_engine* e, const AgentWarpArgs& a, hipStream_t s) override
    {
        dA.tab = dB.tab = d.tab;
        const dim3 grid = grid_for(a.end - a.first), block(TBX_BLOCK);
        switch (a.obs ? a.stack : 0) {
        case 0: hipLaunchKernelGGL(ami_agent_warp_kernel<0>, grid, block, 0, s, d, dA, dB, a, e->n); break;      // the plane ring (new_plane = 2), any depth
        case 1: hipLaunchKernelGGL(ami_agent_warp_kernel<1>, grid, block, 0, s, d, dA, dB, a, e->n); break;
        case 2: hipLaunchKernelGGL(ami_agent_warp_kernel<2>, grid, block, 0, s, d, dA, dB, a, e->n); break;
        case 3: hipLaunchKernelGGL(ami_agent_warp_kernel<3>, grid, block, 0, s, d, dA, dB, a, e->n); break;
        default: hipLaunchKernelGGL(ami_agent_warp_kernel<4>, grid, block, 0, s, d, dA, dB, a, e->n); break;
        }
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int agent_reset_envs(tbx_engine* e, const AgentResetArgs& r, hipStream_t s) override
    {
        dA.tab = dB.tab = d.tab;
        const dim3 grid = r.list ? dim3(std::min<unsigned>(grid_for(e->n).x, 512u)) : grid_for(e->n);
        hipLaunchKernelGGL(ami_agent_reset_kernel, grid, dim3(TBX_BLOCK), 0, s, d, dA, dB, r);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int render_from(tbx_engine* e, int source, const uint8_t* pick_live, uint8_t* out_dev, int channels, hipStream_t s) override
    {
        dA.tab = dB.tab = d.tab;
        const AmiDev& src = source == 1 ? dA : source == 2 ? dB : d;
        return render_impl(e, src, d, source ? pick_live : nullptr, out_dev, channels, 0, e->n, s);
    }

    int render(tbx_engine* e, uint8_t* out_dev, int channels, int first_env, int n_envs, hipStream_t s) override
    {
        return render_impl(e, d, d, nullptr, out_dev, channels, first_env, n_envs, s);
    }

    int render_impl(tbx_engine* e, const AmiDev& src, const AmiDev& alt, const uint8_t* pick_alt, uint8_t* out_dev, int channels, int first_env,
                    int n_envs, hipStream_t s)
    {
        const int split_env = e->opt[TBX_OPT_RENDER_SPLIT];
        // RGB: nine waves per frame (2-3 of the 25 units each) measured 5.45-5.55 TB/s against 4.9 for one wave per frame;
        // gray and RGBA show no such effect (scripts/ab_render.py over TBX_OPT_RENDER_SPLIT)
        // (gray: one wave per frame under-fills the chip at small batches -- four per frame up to 4 096 envs: 0.026 against 0.047 ms
        // at 1 024, 0.064 against 0.066 at 4 096; at 16 384 it is the slower form, 0.196 against 0.187, and so it is for RGBA at 4 096)
        const int split = split_env > 0 ? split_env : channels == 3 ? 9 : (channels == 1 && n_envs <= 4096) ? 4 : 1;
        switch (channels) {
        case 1: if (pick_alt) hipLaunchKernelGGL((ami_render_kernel<1, true>), grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, split, alt, pick_alt); else hipLaunchKernelGGL((ami_render_kernel<1, false>), grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, split, alt, pick_alt); break;
        case 3: if (pick_alt) hipLaunchKernelGGL((ami_render_kernel_w6<3, true>), grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, split, alt, pick_alt); else hipLaunchKernelGGL((ami_render_kernel_w6<3, false>), grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, split, alt, pick_alt); break;
        case 4: if (pick_alt) hipLaunchKernelGGL((ami_render_kernel<4, true>), grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, split, alt, pick_alt); else hipLaunchKernelGGL((ami_render_kernel<4, false>), grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, split, alt, pick_alt); break;
        default: return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
        }
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int pack_state(tbx_engine* e, int env, int count, hipStream_t s) override
    {
        hipLaunchKernelGGL(ami_pack_kernel, dim3(count), dim3(64), 0, s, d, env, (tbx_amidar_state_t*)e->staging);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int unpack_state(tbx_engine* e, int env, int count, const void* pod_host, hipStream_t s) override
    {
        const auto* sts = (const tbx_amidar_state_t*)pod_host;
        for (int i = 0; i < count; i++) {
            const auto& st = sts[i];
            if (st.n_enemies < 0 || st.n_enemies > TBX_AMI_MAX_ENEMIES) return e->fail(TBX_E_UNSUPPORTED, "amidar: the device engine holds at most 8 enemies per env");
            if (st.n_boxes < 0 || st.n_boxes > TBX_AMI_MAX_BOXES) return e->fail(TBX_E_UNSUPPORTED, "amidar: the device engine holds at most 64 boxes per env");
        }
        TBX_HIP(hipMemcpyAsync(e->staging, pod_host, sizeof(tbx_amidar_state_t) * (size_t)count, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(ami_unpack_kernel, dim3(count), dim3(64), 0, s, d, env, (const tbx_amidar_state_t*)e->staging);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int edit(tbx_engine* e, int op, const TbxEditArgs& a, const uint8_t* mask_dev, hipStream_t s) override
    {
        switch (op) {
        case TBX_EDIT_SET_LIVES: case TBX_EDIT_SET_SCORE: case TBX_EDIT_SET_LEVEL: case TBX_EDIT_AMI_TIMERS: case TBX_EDIT_AMI_JUMPS:
        case TBX_EDIT_AMI_TILE: case TBX_EDIT_AMI_ENEMY_AI: case TBX_EDIT_AMI_PLAYER_TILE: case TBX_EDIT_AMI_PLAYER_RANDOM_START: break;
        default: return e->fail(TBX_E_INVALID, "amidar: unknown edit");
        }
        hipLaunchKernelGGL(ami_edit_kernel, dim3((e->n + 255) / 256), dim3(256), 0, s, d, op, a, mask_dev);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int reduce(tbx_engine* e, int query, const TbxEditArgs& a, double* out_dev, int width, hipStream_t s) override
    {
        hipLaunchKernelGGL(ami_reduce_kernel, dim3((e->n + 255) / 256), dim3(256), 0, s, d, query, a, out_dev, width);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int scalars(tbx_engine* e, int32_t* score_dev, int32_t* lives_dev, int32_t* level_dev, hipStream_t s) override
    {
        hipLaunchKernelGGL(ami_scalars_kernel, dim3((e->n + 255) / 256), dim3(256), 0, s, d, score_dev, lives_dev, level_dev);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }
};

}  // namespace

GameOps* tbx_make_amidar_ops() { return new AmiOps(); }
