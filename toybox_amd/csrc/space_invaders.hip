// space_invaders.hip -- SpaceInvaders on gfx950: one 64-lane wavefront per env.
//
// Replaces the per-env Rust transition + rasteriser behind ctoybox.Toybox.apply_ale_action /
// get_state (call sites: /root/reference/toybox/envs/atari/base.py:126,109) for the game the
// reference registers as SpaceInvadersToyboxNoFrameskip-v4 (toybox/__init__.py:20-24).  Rules:
// SPEC.md "SpaceInvaders"; independently restated in scalar C by the CPU checker under oracle/
// and compared bit for bit by tests/test_gpu_parity.py.  All arithmetic is int32 except the
// jitter test (one binary64 compare).
//
// Layout in HBM: everything of an env is env-major, so that the 64 lanes of its wave read it coalesced:
//   head     [N][64]    int32   lane i < NF = scalar field i; words 60..63 = the env's RNG (s0 lo, s0 hi, s1 lo, s1 hi);
//                               words 33, 34 = a MIRROR of enemy 0's x, y (the formation origin), written with every head
//                               row and read only by the step of an engine whose states are plain (si_load_canonical).
//                               (A [field][N] table costs a wave-per-env kernel 8x its bytes: the 32 envs of a 128-byte
//                               line sit in 8 consecutive blocks, which run on the 8 XCDs, and every L2 fetches the line.)
//   enemies  [N][5][64] int32   lane = enemy index (x, y, row | col << 8 | id << 16, points, status)
//   shields  [N][64]    uint32  lane = shield*18 + row (16-bit pixel mask)
//   lasers   [N][8][16] int32   lane (0..8) = laser slot (8 = the ship's laser), field-major
//
// Lane roles in the step: lane = enemy for the march / hit tests / shooter election (ballots and
// wave reductions), lane = shield row for laser erosion, lanes 0..8 = laser slots.

#include "tbx_common.hpp"
#include "raster.hpp"
#include "agent_device.hpp"
#include "../../include/toybox_amd_spec.h"

#include <climits>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>

namespace {

enum SiField {
    F_SCORE, F_LIVES, F_LEVEL, F_LIFE_TIMER, F_SHOT_DELAY, F_N_ENEMIES, F_N_LASERS, F_HAS_SHIP_LASER,
    F_SHIP_X, F_SHIP_Y, F_SHIP_W, F_SHIP_H, F_SHIP_SPEED, F_SHIP_DC, F_SHIP_COLOR, F_SHIP_FLAGS,
    F_UFO_X, F_UFO_Y, F_UFO_APP, F_UFO_DC, F_MOVE_COUNTER, F_MOVE_DIR, F_ORIENT, F_N_SHIELDS,
    F_SHIELD_X0, F_SHIELD_X1, F_SHIELD_X2, F_SHIELD_Y0, F_SHIELD_Y1, F_SHIELD_Y2,
    F_SHIELD_C0, F_SHIELD_C1, F_SHIELD_C2, NF
};
// rci: row | col << 8 | id << 16 (the three fields no rule ever writes; row, col <= 255 and id <= 65535 are checked where
// states come in); status: bit0 alive, bits 8.. death_counter+1
enum { EF_X, EF_Y, EF_RCI, EF_POINTS, EF_STATUS, NEF };
enum { LF_X, LF_Y, LF_W, LF_H, LF_T, LF_MOV, LF_SPEED, LF_COLOR, NLF };
constexpr int SHIP_SLOT = TBX_SI_MAX_LASERS;   // lane 8 carries the ship's laser
constexpr int HEAD_WORDS = 64, HEAD_RNG = 60;  // the head row: one 256-byte load per env
constexpr int HEAD_ORIGIN = 33;                // enemy 0's x, y
static_assert(NF <= HEAD_ORIGIN && HEAD_ORIGIN + 2 <= HEAD_RNG, "the head row's sections");

struct SiDev {
    int n;
    uint64_t* sim_rng; int32_t* prev_score; int32_t* reward; uint8_t* done; int32_t* lives_out; int32_t* score_out;
    uint64_t* packed; uint32_t* err_flag;
    int32_t* sc;          // [N][HEAD_WORDS]: the head row (scalars + RNG)
    int32_t* enemies;     // [N][NEF][64]
    uint32_t* shields;    // [N][64]
    int32_t* lasers;      // [N][NLF][16]
};

struct SiCfg {
    double jitter;
    int32_t start_lives, n_rows, n_shields;
    int32_t row_scores[TBX_SI_MAX_ROWS];
    int32_t shield_x[TBX_SI_MAX_SHIELDS], shield_y[TBX_SI_MAX_SHIELDS];
};

__constant__ uint16_t SI_SHIELD_DEFAULT[TBX_SI_SHIELD_H] = {
    0x0FF0, 0x0FF0, 0x3FFC, 0x3FFC, 0x3FFC, 0x3FFC, 0x3FFC, 0x3FFC, 0x3FFC, 0x3FFC,
    0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xF00F, 0xF00F};

__host__ __device__ constexpr uint32_t rgb_u32(int r, int g, int b) { return (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16) | 0xFF000000u; }

// wave-uniform scalars + per-lane entity slices of one env
struct SiRegs {
    Rng rng;
    int32_t f[NF];
    // lane = enemy
    int32_t ex, ey, erow, ecol, eid, epoints, estatus;
    // lane = shield row
    uint32_t srow;
    // lanes 0..8 = laser slots
    int32_t lf[NLF];
};

// what si_load fetched, as it lies in memory: si_store_changed() writes back only the rows a frame changed
struct SiLoaded {
    int32_t fv;                                   // lane i = scalar field i
    int32_t ex, ey, rci, epoints, estatus;
    uint32_t srow;
    int32_t lf[NLF];
};

__device__ __forceinline__ int32_t si_rci(const SiRegs& s) { return (int32_t)((uint32_t)(s.erow & 0xFF) | ((uint32_t)(s.ecol & 0xFF) << 8) | ((uint32_t)s.eid << 16)); }

__device__ __forceinline__ void si_load_head(const SiDev& d, int env, int lane, SiRegs& s, SiLoaded& o)
{
    o.fv = d.sc[(size_t)env * HEAD_WORDS + lane];
#pragma unroll
    for (int i = 0; i < NF; i++) s.f[i] = __builtin_amdgcn_readlane(o.fv, i);
    s.rng.s0 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(o.fv, HEAD_RNG) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(o.fv, HEAD_RNG + 1) << 32);
    s.rng.s1 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(o.fv, HEAD_RNG + 2) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(o.fv, HEAD_RNG + 3) << 32);
}

__device__ __forceinline__ void si_load(const SiDev& d, int env, int lane, SiRegs& s, SiLoaded& o)
{
    // the env's NF scalars and its RNG with ONE coalesced load (lane i fetches word i of the head row) and a v_readlane per
    // field actually used: they land in SGPRs instead of occupying NF VGPRs
    si_load_head(d, env, lane, s, o);
    const int32_t* e = d.enemies + (size_t)env * NEF * 64;
    o.ex = e[EF_X * 64 + lane]; o.ey = e[EF_Y * 64 + lane]; o.rci = e[EF_RCI * 64 + lane];
    o.epoints = e[EF_POINTS * 64 + lane]; o.estatus = e[EF_STATUS * 64 + lane];
    s.ex = o.ex; s.ey = o.ey; s.epoints = o.epoints; s.estatus = o.estatus;
    s.erow = o.rci & 0xFF; s.ecol = (o.rci >> 8) & 0xFF; s.eid = (int32_t)((uint32_t)o.rci >> 16);
    s.srow = o.srow = d.shields[(size_t)env * 64 + lane];
    const int32_t* l = d.lasers + (size_t)env * NLF * 16;
#pragma unroll
    for (int i = 0; i < NLF; i++) s.lf[i] = o.lf[i] = l[i * 16 + (lane & 15)];
}

__device__ __forceinline__ void si_load(const SiDev& d, int env, int lane, SiRegs& s)
{
    SiLoaded o;
    si_load(d, env, lane, s, o);
}

// The same for an engine whose states are CANONICAL (SiOps::custom == false: every env's enemy i sits at formation origin +
// (32 (i % 6), 18 (i / 6)) with row i / 6, col i % 6, id i and the config's points of its row, and every laser in flight has
// the size, speed, direction and colour of its kind): what never changes is derived instead of loaded.  Of the five 256-byte
// enemy rows only the status row is read -- the formation origin comes out of the head row's mirror -- and of the eight laser
// rows only x, y and t: ~1.3 KB instead of ~2.2 KB per env-frame.  `o` receives exactly what a full load would have found, so
// si_store_changed() keeps writing back the rows a frame changed (x / y when the formation marches, the constant laser rows
// when a laser spawns or dies) and every other kernel keeps reading the full table.
// (Measured and not kept: the status as bytes and the lasers' (x, y, t) as packed words in the head row as well -- two loads per
// env-frame, 844 B = 1.7x the algorithmic bytes -- is 3 % SLOWER (51.1 against 49.6 us): the kernel is bound by its instruction
// stream and exposed latency, not by bytes, and the packing costs more issue slots than the six loads it removes.)
__device__ __forceinline__ void si_load_canonical(const SiDev& d, const SiCfg& c, int env, int lane, SiRegs& s, SiLoaded& o)
{
    si_load_head(d, env, lane, s, o);
    const int32_t fx = __builtin_amdgcn_readlane(o.fv, HEAD_ORIGIN), fy = __builtin_amdgcn_readlane(o.fv, HEAD_ORIGIN + 1);   // enemy 0 = the formation origin
    const bool on = lane < s.f[F_N_ENEMIES];
    const int32_t* e = d.enemies + (size_t)env * NEF * 64;
    o.estatus = e[EF_STATUS * 64 + lane];
    const int row = lane / TBX_SI_COLS, col = lane - row * TBX_SI_COLS;
    int32_t pts = 0;
#pragma unroll
    for (int r = 0; r < TBX_SI_MAX_ROWS; r++) pts = row == r ? c.row_scores[r] : pts;
    o.ex = on ? fx + TBX_SI_ENEMY_DX * col : 0; o.ey = on ? fy + TBX_SI_ENEMY_DY * row : 0;
    o.rci = on ? (int32_t)((uint32_t)row | ((uint32_t)col << 8) | ((uint32_t)lane << 16)) : 0;
    o.epoints = on ? pts : 0;
    s.ex = o.ex; s.ey = o.ey; s.epoints = o.epoints; s.estatus = o.estatus;
    s.erow = on ? row : 0; s.ecol = on ? col : 0; s.eid = on ? lane : 0;
    s.srow = o.srow = d.shields[(size_t)env * 64 + lane];
    const int slot = lane & 15;
    const bool enemy_laser = slot < s.f[F_N_LASERS] && slot < TBX_SI_MAX_LASERS, ship_laser = slot == SHIP_SLOT && s.f[F_HAS_SHIP_LASER] != 0;
    const bool lz = enemy_laser || ship_laser;
    const int32_t* l = d.lasers + (size_t)env * NLF * 16;
    o.lf[LF_X] = l[LF_X * 16 + slot]; o.lf[LF_Y] = l[LF_Y * 16 + slot]; o.lf[LF_T] = l[LF_T * 16 + slot];
    o.lf[LF_W] = lz ? TBX_SI_LASER_W : 0; o.lf[LF_H] = lz ? TBX_SI_LASER_H : 0;
    o.lf[LF_MOV] = ship_laser ? TBX_DIR_UP : enemy_laser ? TBX_DIR_DOWN : 0;
    o.lf[LF_SPEED] = ship_laser ? TBX_SI_SHIP_LASER_V : enemy_laser ? TBX_SI_ENEMY_LASER_V : 0;
    o.lf[LF_COLOR] = ship_laser ? (int32_t)rgb_u32(TBX_SI_COL_SHIP_LASER) : enemy_laser ? (int32_t)rgb_u32(TBX_SI_COL_ENEMY_LASER) : 0;
#pragma unroll
    for (int i = 0; i < NLF; i++) s.lf[i] = o.lf[i];
}

__device__ __forceinline__ int32_t si_scalar_row(int lane, const SiRegs& s)
{
    int32_t fv = 0;                               // v_writelane: one instruction per field (a select chain is two)
#pragma unroll
    for (int i = 0; i < NF; i++) asm("v_writelane_b32 %0, %1, %2" : "+v"(fv) : "s"(wave_uniform(s.f[i])), "n"(i));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(fv) : "s"(__builtin_amdgcn_readlane(s.ex, 0)), "n"(HEAD_ORIGIN));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(fv) : "s"(__builtin_amdgcn_readlane(s.ey, 0)), "n"(HEAD_ORIGIN + 1));
    const uint64_t r0 = wave_uniform64(s.rng.s0), r1 = wave_uniform64(s.rng.s1);
    asm("v_writelane_b32 %0, %1, %2" : "+v"(fv) : "s"((int32_t)(uint32_t)r0), "n"(HEAD_RNG));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(fv) : "s"((int32_t)(uint32_t)(r0 >> 32)), "n"(HEAD_RNG + 1));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(fv) : "s"((int32_t)(uint32_t)r1), "n"(HEAD_RNG + 2));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(fv) : "s"((int32_t)(uint32_t)(r1 >> 32)), "n"(HEAD_RNG + 3));
    return fv;
}

__device__ __forceinline__ void si_store(const SiDev& d, int env, int lane, const SiRegs& s)
{
    d.sc[(size_t)env * HEAD_WORDS + lane] = si_scalar_row(lane, s);      // the head row: scalars, zeros, RNG
    int32_t* e = d.enemies + (size_t)env * NEF * 64;
    e[EF_X * 64 + lane] = s.ex; e[EF_Y * 64 + lane] = s.ey; e[EF_RCI * 64 + lane] = si_rci(s);
    e[EF_POINTS * 64 + lane] = s.epoints; e[EF_STATUS * 64 + lane] = s.estatus;
    d.shields[(size_t)env * 64 + lane] = s.srow;
    if (lane < 16) {
        int32_t* l = d.lasers + (size_t)env * NLF * 16;
#pragma unroll
        for (int i = 0; i < NLF; i++) l[i * 16 + lane] = s.lf[i];
    }
}

// the same state back to where si_load(.., o) took it from: only the 256-byte rows (and scalar words) that differ.  A frame
// moves the lasers and a few counters; the formation marches every TBX_SI_MOVE_PERIOD frames, shields and statuses change on
// hits -- the step kernel is bound by its HBM bytes, and most of them were rows written back unchanged
__device__ __forceinline__ void si_store_changed(const SiDev& d, int env, int lane, const SiRegs& s, const SiLoaded& o)
{
    const int32_t fv = si_scalar_row(lane, s);
    if (fv != o.fv) d.sc[(size_t)env * HEAD_WORDS + lane] = fv;
    int32_t* e = d.enemies + (size_t)env * NEF * 64;
    const int32_t rci = si_rci(s);
    if (__ballot(s.ex != o.ex)) e[EF_X * 64 + lane] = s.ex;
    if (__ballot(s.ey != o.ey)) e[EF_Y * 64 + lane] = s.ey;
    if (__ballot(rci != o.rci)) e[EF_RCI * 64 + lane] = rci;
    if (__ballot(s.epoints != o.epoints)) e[EF_POINTS * 64 + lane] = s.epoints;
    if (__ballot(s.estatus != o.estatus)) e[EF_STATUS * 64 + lane] = s.estatus;
    if (__ballot(s.srow != o.srow)) d.shields[(size_t)env * 64 + lane] = s.srow;
    int32_t* l = d.lasers + (size_t)env * NLF * 16;
#pragma unroll
    for (int i = 0; i < NLF; i++)
        if (__ballot(lane < 16 && s.lf[i] != o.lf[i]) && lane < 16) l[i * 16 + lane] = s.lf[i];
}

// k == 0 ? a : k == 1 ? b : c over VALUES
__device__ __forceinline__ int32_t sel3(int k, int32_t a, int32_t b, int32_t c) { return k == 0 ? a : k == 1 ? b : c; }

__device__ __forceinline__ bool e_alive(const SiRegs& s) { return (s.estatus & 1) != 0; }
__device__ __forceinline__ int e_dc(const SiRegs& s) { return (s.estatus >> 8) - 1; }
__device__ __forceinline__ int32_t mk_status(bool alive, int dc) { return (alive ? 1 : 0) | ((dc + 1) << 8); }

__device__ __forceinline__ void si_reset_formation(SiRegs& s, int lane)
{
    if (lane < s.f[F_N_ENEMIES]) {
        s.ex = TBX_SI_ENEMY_X0 + TBX_SI_ENEMY_DX * s.ecol;
        s.ey = TBX_SI_ENEMY_Y0 + TBX_SI_ENEMY_DY * s.erow;
        s.estatus = mk_status(true, -1);
    }
    s.f[F_MOVE_COUNTER] = TBX_SI_MOVE_PERIOD;
    s.f[F_MOVE_DIR] = TBX_DIR_RIGHT;
    s.f[F_ORIENT] = 1;
    s.f[F_N_LASERS] = 0;
    s.f[F_HAS_SHIP_LASER] = 0;
#pragma unroll
    for (int i = 0; i < NLF; i++) s.lf[i] = 0;
}

__device__ __forceinline__ void si_new_game(const SiCfg& c, int lane, Rng& sim, SiRegs& s)
{
    s.rng = sim.child();
#pragma unroll
    for (int i = 0; i < NF; i++) s.f[i] = 0;
    s.f[F_LIVES] = c.start_lives;
    s.f[F_LEVEL] = 1;
    s.f[F_LIFE_TIMER] = TBX_SI_NEW_LIFE_TIME;
    s.f[F_SHOT_DELAY] = TBX_SI_SHOT_DELAY;
    const int ne = TBX_SI_COLS * c.n_rows;
    s.f[F_N_ENEMIES] = ne;
    s.ex = s.ey = s.erow = s.ecol = s.eid = s.epoints = 0;
    s.estatus = 0;   // slots beyond n_enemies are all-zero records
    if (lane < ne) {
        s.erow = lane / TBX_SI_COLS; s.ecol = lane % TBX_SI_COLS; s.eid = lane;
        s.epoints = c.row_scores[s.erow];
    }
    si_reset_formation(s, lane);
    if (lane >= ne) s.estatus = 0;
    s.f[F_SHIP_X] = TBX_SI_SHIP_X0; s.f[F_SHIP_Y] = TBX_SI_SHIP_Y; s.f[F_SHIP_W] = TBX_SI_SHIP_W; s.f[F_SHIP_H] = TBX_SI_SHIP_H;
    s.f[F_SHIP_SPEED] = TBX_SI_SHIP_SPEED; s.f[F_SHIP_DC] = -1; s.f[F_SHIP_COLOR] = (int32_t)rgb_u32(TBX_SI_COL_SHIP);
    s.f[F_SHIP_FLAGS] = 2;   // alive = 0, death_hit_1 = 1
    s.f[F_UFO_X] = TBX_SI_UFO_X0; s.f[F_UFO_Y] = TBX_SI_UFO_Y; s.f[F_UFO_APP] = TBX_SI_UFO_PERIOD; s.f[F_UFO_DC] = -1;
    s.f[F_N_SHIELDS] = c.n_shields;
#pragma unroll
    for (int k = 0; k < TBX_SI_MAX_SHIELDS; k++) {
        const bool on = k < c.n_shields;
        s.f[F_SHIELD_X0 + k] = on ? c.shield_x[k] : 0;
        s.f[F_SHIELD_Y0 + k] = on ? c.shield_y[k] : 0;
        s.f[F_SHIELD_C0 + k] = on ? (int32_t)rgb_u32(TBX_SI_COL_SHIELD) : 0;
    }
    {
        const int k = lane / TBX_SI_SHIELD_H, r = lane - k * TBX_SI_SHIELD_H;
        s.srow = (lane < TBX_SI_MAX_SHIELDS * TBX_SI_SHIELD_H && k < c.n_shields) ? SI_SHIELD_DEFAULT[r] : 0u;
    }
}

__device__ __forceinline__ bool overlap(int ax, int ay, int aw, int ah, int bx, int by, int bw, int bh)
{
    return ax < bx + bw && bx < ax + aw && ay < by + bh && by < ay + ah;
}

struct Laser { int32_t x, y, w, h, t, mov, speed, color; };

__device__ __forceinline__ Laser get_laser(const SiRegs& s, int slot)
{
    Laser l;
    l.x = __builtin_amdgcn_readlane(s.lf[LF_X], slot); l.y = __builtin_amdgcn_readlane(s.lf[LF_Y], slot);
    l.w = __builtin_amdgcn_readlane(s.lf[LF_W], slot); l.h = __builtin_amdgcn_readlane(s.lf[LF_H], slot);
    l.t = __builtin_amdgcn_readlane(s.lf[LF_T], slot); l.mov = __builtin_amdgcn_readlane(s.lf[LF_MOV], slot);
    l.speed = __builtin_amdgcn_readlane(s.lf[LF_SPEED], slot); l.color = __builtin_amdgcn_readlane(s.lf[LF_COLOR], slot);
    return l;
}

__device__ __forceinline__ void put_laser(SiRegs& s, int lane, int slot, const Laser& l)
{
    if (lane == slot) {
        s.lf[LF_X] = l.x; s.lf[LF_Y] = l.y; s.lf[LF_W] = l.w; s.lf[LF_H] = l.h;
        s.lf[LF_T] = l.t; s.lf[LF_MOV] = l.mov; s.lf[LF_SPEED] = l.speed; s.lf[LF_COLOR] = l.color;
    }
}

__device__ __forceinline__ void clear_laser(SiRegs& s, int lane, int slot)
{
    if (lane == slot) {
#pragma unroll
        for (int i = 0; i < NLF; i++) s.lf[i] = 0;
    }
}

__device__ __forceinline__ void move_laser(Laser& l)
{
    if (l.mov == TBX_DIR_UP) l.y -= l.speed;
    else if (l.mov == TBX_DIR_DOWN) l.y += l.speed;
    else if (l.mov == TBX_DIR_LEFT) l.x -= l.speed;
    else l.x += l.speed;
    l.t += 1;
}

// lane = shield row.  Returns true (wave-uniform) and erodes the first shield, in index order,
// that has a live pixel under the laser rect.
__device__ __forceinline__ bool shield_hit(SiRegs& s, int lane, const Laser& l)
{
    const int k = lane / TBX_SI_SHIELD_H, r = lane - k * TBX_SI_SHIELD_H;
    const bool valid = lane < TBX_SI_MAX_SHIELDS * TBX_SI_SHIELD_H && k < s.f[F_N_SHIELDS];
    // (values, not lvalues, in the selects: a conditional over array elements becomes a select of addresses and pushes
    // the whole register array into scratch)
    const int sx = sel3(k, s.f[F_SHIELD_X0], s.f[F_SHIELD_X1], s.f[F_SHIELD_X2]);
    const int sy = sel3(k, s.f[F_SHIELD_Y0], s.f[F_SHIELD_Y1], s.f[F_SHIELD_Y2]);
    int cx0 = l.x - sx, cx1 = l.x + l.w - sx, cy0 = l.y - sy, cy1 = l.y + l.h - sy;
    if (cx0 < 0) cx0 = 0;
    if (cy0 < 0) cy0 = 0;
    if (cx1 > TBX_SI_SHIELD_W) cx1 = TBX_SI_SHIELD_W;
    if (cy1 > TBX_SI_SHIELD_H) cy1 = TBX_SI_SHIELD_H;
    const bool inside = valid && cx0 < cx1 && r >= cy0 && r < cy1;
    uint32_t colmask = 0, dmask = 0;
    if (inside) {
        colmask = ((1u << cx1) - 1u) & ~((1u << cx0) - 1u);
        const int dx0 = cx0 > 0 ? cx0 - 1 : 0, dx1 = cx1 < TBX_SI_SHIELD_W ? cx1 + 1 : TBX_SI_SHIELD_W;
        dmask = ((1u << dx1) - 1u) & ~((1u << dx0) - 1u);
    }
    const uint64_t hits = __ballot(inside && (s.srow & colmask) != 0);
    if (!hits) return false;
    const int kh = (int)__builtin_ctzll(hits) / TBX_SI_SHIELD_H;   // first shield in index order
    if (inside && k == kh) s.srow &= ~dmask;
    return true;
}

// the clipped rectangle shield_hit() builds is not empty for some shield (wave-uniform arithmetic: the cheap way to say no)
__device__ __forceinline__ bool near_shield(const SiRegs& s, const Laser& l)
{
    bool near = false;
#pragma unroll
    for (int k = 0; k < TBX_SI_MAX_SHIELDS; k++) {
        const int sx = sel3(k, s.f[F_SHIELD_X0], s.f[F_SHIELD_X1], s.f[F_SHIELD_X2]);
        const int sy = sel3(k, s.f[F_SHIELD_Y0], s.f[F_SHIELD_Y1], s.f[F_SHIELD_Y2]);
        const int cx0 = max(l.x - sx, 0), cx1 = min(l.x + l.w - sx, TBX_SI_SHIELD_W);
        const int cy0 = max(l.y - sy, 0), cy1 = min(l.y + l.h - sy, TBX_SI_SHIELD_H);
        near = near || (k < s.f[F_N_SHIELDS] && cx0 < cx1 && cy0 < cy1);
    }
    return near;
}

__device__ __forceinline__ bool dec_counter(int32_t& c)
{
    if (c < 0) return false;
    c -= 1;
    if (c <= 0) { c = -1; return true; }
    return false;
}

__device__ __forceinline__ uint64_t wave_or64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, o), hi = __shfl_xor((uint32_t)(v >> 32), o);
        v |= (uint64_t)lo | ((uint64_t)hi << 32);
    }
    return v;
}

__device__ __forceinline__ int64_t wave_max64(int64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, o), hi = __shfl_xor((uint32_t)((uint64_t)v >> 32), o);
        int64_t w = (int64_t)((uint64_t)lo | ((uint64_t)hi << 32));
        v = w > v ? w : v;
    }
    return v;
}

// lowest-on-screen alive enemy of column `col` (max y, ties lowest index); -1 if none
__device__ __forceinline__ int column_shooter(const SiRegs& s, int lane, int col)
{
    const bool cand = lane < s.f[F_N_ENEMIES] && e_alive(s) && (s.ecol & 63) == col;
    const int64_t key = cand ? ((int64_t)s.ey * 256 + (int64_t)(63 - lane)) : INT64_MIN;
    const int64_t best = wave_max64(key);
    if (best == INT64_MIN) return -1;
    return 63 - (int)(best & 0xFF);
}

__device__ __forceinline__ void si_step(const SiCfg& c, int lane, uint32_t buttons, SiRegs& s)
{
    int32_t* f = s.f;
    // A. "get ready" phase of a life
    if (f[F_LIFE_TIMER] > 0) {
        f[F_LIFE_TIMER] -= 1;
        if (f[F_LIFE_TIMER] == 0) { f[F_SHIP_FLAGS] = 3; f[F_SHIP_DC] = -1; }
        return;
    }
    // B. ship explosion: the rest of the world is frozen
    if (f[F_SHIP_DC] >= 0) {
        if (dec_counter(f[F_SHIP_DC])) {
            f[F_LIVES] -= 1;
            f[F_SHIP_FLAGS] |= 2;
            if (f[F_LIVES] > 0) {
                f[F_LIFE_TIMER] = TBX_SI_NEW_LIFE_TIME;
                f[F_SHIP_X] = TBX_SI_SHIP_X0;
                f[F_N_LASERS] = 0; f[F_HAS_SHIP_LASER] = 0;
#pragma unroll
                for (int i = 0; i < NLF; i++) s.lf[i] = 0;
            }
        } else {
            const bool h1 = ((f[F_SHIP_DC] >> 2) & 1) == 0;
            f[F_SHIP_FLAGS] = (f[F_SHIP_FLAGS] & 1) | (h1 ? 2 : 0);
        }
        return;
    }
    if (!(f[F_SHIP_FLAGS] & 1)) return;

    const int ne = f[F_N_ENEMIES];
    // C. ship
    if (buttons & TBX_BTN_LEFT) f[F_SHIP_X] -= f[F_SHIP_SPEED];
    else if (buttons & TBX_BTN_RIGHT) f[F_SHIP_X] += f[F_SHIP_SPEED];
    if (f[F_SHIP_X] < TBX_SI_SHIP_X_MIN) f[F_SHIP_X] = TBX_SI_SHIP_X_MIN;
    if (f[F_SHIP_X] > TBX_SI_SHIP_X_MAX) f[F_SHIP_X] = TBX_SI_SHIP_X_MAX;

    // D. fire
    if ((buttons & TBX_BTN_BUTTON1) && !f[F_HAS_SHIP_LASER]) {
        Laser l;
        l.x = f[F_SHIP_X] + f[F_SHIP_W] / 2 - 1; l.y = f[F_SHIP_Y] - TBX_SI_LASER_H;
        l.w = TBX_SI_LASER_W; l.h = TBX_SI_LASER_H; l.t = 0; l.mov = TBX_DIR_UP;
        l.speed = TBX_SI_SHIP_LASER_V; l.color = (int32_t)rgb_u32(TBX_SI_COL_SHIP_LASER);
        put_laser(s, lane, SHIP_SLOT, l);
        f[F_HAS_SHIP_LASER] = 1;
    }

    // E. ship laser
    if (f[F_HAS_SHIP_LASER]) {
        Laser l = get_laser(s, SHIP_SLOT);
        move_laser(l);
        bool has = true;
        if (l.y + l.h <= 0 || l.y >= TBX_SI_GROUND_Y || l.x + l.w <= 0 || l.x >= TBX_SI_W) has = false;
        if (has) {
            const uint64_t m = __ballot(lane < ne && e_alive(s) && overlap(l.x, l.y, l.w, l.h, s.ex, s.ey, TBX_SI_ENEMY_W, TBX_SI_ENEMY_H));
            if (m) {
                const int idx = (int)__builtin_ctzll(m);
                f[F_SCORE] += __builtin_amdgcn_readlane(s.epoints, idx);
                if (lane == idx) s.estatus = mk_status(false, TBX_SI_ENEMY_DEATH_T);
                has = false;
            }
        }
        if (has && f[F_UFO_APP] == 0 && f[F_UFO_DC] < 0 &&
            overlap(l.x, l.y, l.w, l.h, f[F_UFO_X], f[F_UFO_Y], TBX_SI_UFO_W, TBX_SI_UFO_H)) {
            f[F_UFO_DC] = TBX_SI_UFO_DEATH_T;
            f[F_SCORE] += TBX_SI_UFO_BONUS;
            has = false;
        }
        if (has && near_shield(s, l) && shield_hit(s, lane, l)) has = false;
        if (has) { if (lane == SHIP_SLOT) { s.lf[LF_X] = l.x; s.lf[LF_Y] = l.y; s.lf[LF_T] = l.t; } }   // what move_laser() changes
        else { clear_laser(s, lane, SHIP_SLOT); f[F_HAS_SHIP_LASER] = 0; }
    }

    // F. enemy explosions
    if (lane < ne) {
        int dc = e_dc(s);
        dec_counter(dc);
        s.estatus = mk_status(e_alive(s), dc);
    }

    // G. formation march
    f[F_MOVE_COUNTER] -= 1;
    if (f[F_MOVE_COUNTER] <= 0) {
        const bool alive = lane < ne && e_alive(s);
        const int n_alive = __popcll(__ballot(alive));
        const int dx = f[F_MOVE_DIR] == TBX_DIR_RIGHT ? TBX_SI_STEP_X : -TBX_SI_STEP_X;
        const bool at_edge = alive && (dx > 0 ? s.ex + TBX_SI_ENEMY_W + dx > TBX_SI_FIELD_X_MAX : s.ex + dx < TBX_SI_FIELD_X_MIN);
        const bool edge = __ballot(at_edge) != 0;
        if (lane < ne) {
            if (edge) s.ey += TBX_SI_STEP_Y;
            else s.ex += dx;
        }
        if (edge) f[F_MOVE_DIR] = f[F_MOVE_DIR] == TBX_DIR_RIGHT ? TBX_DIR_LEFT : TBX_DIR_RIGHT;
        f[F_ORIENT] = f[F_ORIENT] ? 0 : 1;
        f[F_MOVE_COUNTER] = TBX_SI_MOVE_PERIOD_MIN + (ne > 0 ? ((TBX_SI_MOVE_PERIOD - TBX_SI_MOVE_PERIOD_MIN) * n_alive) / ne : 0);
        if (__ballot(alive && s.ey + TBX_SI_ENEMY_H >= f[F_SHIP_Y])) f[F_LIVES] = 0;   // invasion
    }

    // H. enemy fire
    f[F_SHOT_DELAY] -= 1;
    if (f[F_SHOT_DELAY] <= 0) {
        f[F_SHOT_DELAY] = TBX_SI_SHOT_DELAY;
        const bool alive = lane < ne && e_alive(s);
        const uint64_t colmask = wave_or64(alive ? 1ull << (s.ecol & 63) : 0ull);
        if (colmask && f[F_N_LASERS] < TBX_SI_MAX_LASERS) {
            const uint64_t draw = s.rng.next();
            const double u = (double)(draw >> 11) * (1.0 / 9007199254740992.0);
            int col = -1;
            if (u < c.jitter) {
                int k = (int)s.rng.range((uint64_t)__popcll(colmask));
                uint64_t m = colmask;
                while (k > 0) { m &= m - 1; k--; }
                col = (int)__builtin_ctzll(m);
            } else {
                int best = 1 << 30;
                const int target = f[F_SHIP_X] + f[F_SHIP_W] / 2;
                uint64_t m = colmask;
                while (m) {
                    const int b = (int)__builtin_ctzll(m);
                    m &= m - 1;
                    const int sh = column_shooter(s, lane, b);
                    int dd = __builtin_amdgcn_readlane(s.ex, sh) + TBX_SI_ENEMY_W / 2 - target;
                    if (dd < 0) dd = -dd;
                    if (dd < best) { best = dd; col = b; }
                }
            }
            const int sh = column_shooter(s, lane, col);
            Laser l;
            l.x = __builtin_amdgcn_readlane(s.ex, sh) + TBX_SI_ENEMY_W / 2 - 1;
            l.y = __builtin_amdgcn_readlane(s.ey, sh) + TBX_SI_ENEMY_H;
            l.w = TBX_SI_LASER_W; l.h = TBX_SI_LASER_H; l.t = 0; l.mov = TBX_DIR_DOWN;
            l.speed = TBX_SI_ENEMY_LASER_V; l.color = (int32_t)rgb_u32(TBX_SI_COL_ENEMY_LASER);
            put_laser(s, lane, f[F_N_LASERS], l);
            f[F_N_LASERS] += 1;
        }
    }

    // I. enemy lasers.  The rules take them in slot order (bounds, then shields, then the ship), but only two things carry
    // from one laser to the next: the shields' pixels and whether the ship is still alive.  So lane = slot moves every laser
    // and tests the bounds at once, only the lasers whose rectangle reaches a shield's box take the eroding shield test one
    // after the other, and the first of the rest that overlaps the living ship kills it.
    {
        const int nl = f[F_N_LASERS];
        const bool mine = lane < nl;                      // nl <= TBX_SI_MAX_LASERS; the ship's laser sits in the slot above
        uint32_t keepmask = 0;
        if (nl > 0) {
            int32_t lx = s.lf[LF_X], ly = s.lf[LF_Y];
            const int32_t lw = s.lf[LF_W], lh = s.lf[LF_H], mov = s.lf[LF_MOV], sp = s.lf[LF_SPEED];
            if (mov == TBX_DIR_UP) ly -= sp;
            else if (mov == TBX_DIR_DOWN) ly += sp;
            else if (mov == TBX_DIR_LEFT) lx -= sp;
            else lx += sp;
            bool gone = ly + lh >= TBX_SI_GROUND_Y || ly + lh <= 0 || lx + lw <= 0 || lx >= TBX_SI_W;
            bool near = false;                            // the clipped rectangle shield_hit() would build is not empty
#pragma unroll
            for (int k = 0; k < TBX_SI_MAX_SHIELDS; k++) {
                const int sx = sel3(k, f[F_SHIELD_X0], f[F_SHIELD_X1], f[F_SHIELD_X2]);
                const int sy = sel3(k, f[F_SHIELD_Y0], f[F_SHIELD_Y1], f[F_SHIELD_Y2]);
                const int cx0 = max(lx - sx, 0), cx1 = min(lx + lw - sx, TBX_SI_SHIELD_W);
                const int cy0 = max(ly - sy, 0), cy1 = min(ly + lh - sy, TBX_SI_SHIELD_H);
                near = near || (k < f[F_N_SHIELDS] && cx0 < cx1 && cy0 < cy1);
            }
            for (uint64_t todo = __ballot(mine && !gone && near); todo; todo &= todo - 1) {
                const int i = (int)__builtin_ctzll(todo);
                Laser l;
                l.x = __builtin_amdgcn_readlane(lx, i); l.y = __builtin_amdgcn_readlane(ly, i);
                l.w = __builtin_amdgcn_readlane(lw, i); l.h = __builtin_amdgcn_readlane(lh, i);
                if (shield_hit(s, lane, l) && lane == i) gone = true;
            }
            if (f[F_SHIP_FLAGS] & 1) {
                const uint64_t hit = __ballot(mine && !gone && overlap(lx, ly, lw, lh, f[F_SHIP_X], f[F_SHIP_Y], f[F_SHIP_W], f[F_SHIP_H]));
                if (hit) {
                    f[F_SHIP_FLAGS] = 2; f[F_SHIP_DC] = TBX_SI_SHIP_DEATH_T;
                    if (lane == (int)__builtin_ctzll(hit)) gone = true;
                }
            }
            if (mine) { s.lf[LF_X] = lx; s.lf[LF_Y] = ly; s.lf[LF_T] += 1; }
            keepmask = (uint32_t)__ballot(mine && !gone);
        }
        // compaction: slot j takes the j-th kept laser; freed slots are zeroed (nothing to do while no laser went: the slots
        // above the count are zero already)
        const int keep = __popc(keepmask);
        if (keep != nl) {
            int src = -1;
            if (lane < TBX_SI_MAX_LASERS && lane < keep) {
                uint32_t m = keepmask;
                for (int j = 0; j < lane; j++) m &= m - 1;
                src = __builtin_ctz(m);
            }
            const int from = src < 0 ? lane : src;
#pragma unroll
            for (int i = 0; i < NLF; i++) {
                const int32_t v = __shfl(s.lf[i], from);
                if (lane < TBX_SI_MAX_LASERS) s.lf[i] = src >= 0 ? v : 0;
            }
            f[F_N_LASERS] = keep;
        }
    }

    // J. ufo
    if (f[F_UFO_DC] >= 0) {
        if (dec_counter(f[F_UFO_DC])) { f[F_UFO_X] = TBX_SI_UFO_X0; f[F_UFO_APP] = TBX_SI_UFO_PERIOD; }
    } else if (f[F_UFO_APP] > 0) {
        f[F_UFO_APP] -= 1;
    } else if (f[F_UFO_APP] == 0) {
        f[F_UFO_X] += TBX_SI_UFO_STEP;
        if (f[F_UFO_X] >= TBX_SI_W) { f[F_UFO_X] = TBX_SI_UFO_X0; f[F_UFO_APP] = TBX_SI_UFO_PERIOD; }
    }

    // K. wave cleared
    {
        const bool busy = lane < ne && (e_alive(s) || e_dc(s) >= 0);
        if (!__ballot(busy) && ne > 0) { f[F_LEVEL] += 1; si_reset_formation(s, lane); }
    }
}

// ------------------------------------------------------------------ rasteriser input record
//
// What the rasteriser needs of one env, digested by whoever holds the env's state in registers -- the batch step kernel, or
// si_rec_prep_kernel after anything else touched the state.  The painter's set-up becomes a few scalar loads and ~300
// instructions instead of a 2.2 KB gather of state rows followed by ~1 500 dependent instructions (five times per RGB
// frame), and -- as for Breakout -- a step may run while the previous frame is still being painted, because the painter no
// longer reads live state (there are two buffers of records: GameOps::step_ahead).
//
// CANONICAL formation only: enemy i sits at (fx + 32 (i % 6), fy + 18 (i / 6)) with row = i / 6, col = i % 6.  Every game the
// engine starts is like that and stays like that (the whole formation marches in lockstep, dead enemies included); an
// intervention that writes other enemy coordinates switches the engine to the state-reading rasteriser (SiOps::custom).
// Coordinates are clamped to +-1000 (far off screen either way), rectangles are clipped to the screen.
struct alignas(128) SiRenderRec {
    int32_t fx, fy;                    // formation origin
    uint32_t vis_lo, vis_hi;           // bit i: enemy i is drawn (alive or exploding)
    uint32_t alive_lo, alive_hi;       // bit i: ... with its marching sprite (else the explosion)
    uint32_t flags;                    // REC_* below
    uint32_t hud;                      // 4-bit digits: score 10^4..10^0 (bits 0..19), lives (20..23), level (24..27)
    int32_t ship_x, ship_y;
    uint32_t ship_color;
    int32_t ufo_x, ufo_y;
    uint32_t shield_xy[TBX_SI_MAX_SHIELDS];      // (uint16)x | (uint16)y << 16, two's complement
    uint32_t shield_color[TBX_SI_MAX_SHIELDS];
    uint32_t _pad0;
    uint32_t laser_x[12], laser_y[12];           // slot 0..7 enemy lasers, 8 the ship's: x0 | x1 << 16, y0 | y1 << 16 (clipped; 0 = none)
    uint32_t laser_color[12];
    uint16_t shield_rows[64];                    // lane = shield * 18 + row
    uint32_t _pad1[8];
};
static_assert(sizeof(SiRenderRec) == 384, "render record layout");
constexpr int REC_HDR_DWORDS = 20;
enum { REC_ORIENT = 1u << 7, REC_UFO = 1u << 8, REC_SHIP_SHIFT = 9, REC_SHIELDS_SHIFT = 11 };   // bits 0..6: n_enemies

__device__ __forceinline__ int32_t clamp_coord(int32_t v) { return v < -1000 ? -1000 : v > 1000 ? 1000 : v; }

__device__ __forceinline__ uint32_t si_hud_word(int sc, int lv, int le)
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

// every lane of the env's wave calls this with the env's state in `s`
__device__ __forceinline__ void si_write_rec(SiRenderRec* __restrict__ rec, int lane, const SiRegs& s)
{
    const int32_t* f = s.f;
    const int ne = f[F_N_ENEMIES];
    const uint64_t alive = __ballot(lane < ne && e_alive(s));
    const uint64_t vis = __ballot(lane < ne && (e_alive(s) || e_dc(s) >= 0));
    const int pose = (f[F_SHIP_FLAGS] & 1) ? 1 : f[F_SHIP_DC] >= 0 ? ((f[F_SHIP_FLAGS] & 2) ? 2 : 3) : 0;
    const bool ufo_on = f[F_UFO_APP] == 0 || f[F_UFO_DC] >= 0;
    uint32_t h[REC_HDR_DWORDS];
    h[0] = (uint32_t)clamp_coord(ne > 0 ? __builtin_amdgcn_readlane(s.ex, 0) : 0);
    h[1] = (uint32_t)clamp_coord(ne > 0 ? __builtin_amdgcn_readlane(s.ey, 0) : 0);
    h[2] = (uint32_t)vis; h[3] = (uint32_t)(vis >> 32);
    h[4] = (uint32_t)alive; h[5] = (uint32_t)(alive >> 32);
    h[6] = (uint32_t)(ne & 127) | (f[F_ORIENT] ? REC_ORIENT : 0u) | (ufo_on ? REC_UFO : 0u) | ((uint32_t)pose << REC_SHIP_SHIFT) |
           ((uint32_t)(f[F_N_SHIELDS] & 3) << REC_SHIELDS_SHIFT);
    h[7] = si_hud_word(f[F_SCORE], f[F_LIVES], f[F_LEVEL]);
    h[8] = (uint32_t)clamp_coord(f[F_SHIP_X]); h[9] = (uint32_t)clamp_coord(f[F_SHIP_Y]);
    h[10] = (uint32_t)f[F_SHIP_COLOR];
    h[11] = (uint32_t)clamp_coord(f[F_UFO_X]); h[12] = (uint32_t)clamp_coord(f[F_UFO_Y]);
#pragma unroll
    for (int k = 0; k < TBX_SI_MAX_SHIELDS; k++) {
        h[13 + k] = ((uint32_t)clamp_coord(f[F_SHIELD_X0 + k]) & 0xFFFFu) | ((uint32_t)clamp_coord(f[F_SHIELD_Y0 + k]) << 16);
        h[16 + k] = (uint32_t)f[F_SHIELD_C0 + k];
    }
    h[19] = 0u;
    uint32_t hv = 0u;                              // lane i holds header dword i: one 80-byte store
#pragma unroll
    for (int i = 0; i < REC_HDR_DWORDS; i++) asm("v_writelane_b32 %0, %1, %2" : "+v"(hv) : "s"(wave_uniform((int)h[i])), "n"(i));
    uint32_t* out = reinterpret_cast<uint32_t*>(rec);
    if (lane < REC_HDR_DWORDS) out[lane] = hv;
    if (lane < 12) {
        const bool on = lane == SHIP_SLOT ? f[F_HAS_SHIP_LASER] != 0 : lane < f[F_N_LASERS];
        const long lx = s.lf[LF_X], ly = s.lf[LF_Y];
        long x0 = lx, x1 = lx + s.lf[LF_W], y0 = ly, y1 = ly + s.lf[LF_H];
        x0 = x0 < 0 ? 0 : x0 > TBX_SI_W ? TBX_SI_W : x0; x1 = x1 < x0 ? x0 : x1 > TBX_SI_W ? TBX_SI_W : x1;
        y0 = y0 < 0 ? 0 : y0 > TBX_SI_H ? TBX_SI_H : y0; y1 = y1 < y0 ? y0 : y1 > TBX_SI_H ? TBX_SI_H : y1;
        const bool show = on && x0 < x1 && y0 < y1;
        rec->laser_x[lane] = show ? (uint32_t)x0 | ((uint32_t)x1 << 16) : 0u;
        rec->laser_y[lane] = show ? (uint32_t)y0 | ((uint32_t)y1 << 16) : 0u;
        rec->laser_color[lane] = show ? (uint32_t)s.lf[LF_COLOR] : 0u;
    }
    {
        const int k = lane / TBX_SI_SHIELD_H;
        const bool valid = lane < TBX_SI_MAX_SHIELDS * TBX_SI_SHIELD_H && k < f[F_N_SHIELDS];
        rec->shield_rows[lane] = valid ? (uint16_t)s.srow : (uint16_t)0;
    }
}

// ------------------------------------------------------------------ kernels

__global__ __launch_bounds__(TBX_BLOCK) void si_new_game_kernel(SiDev d, SiCfg c, const uint8_t* mask)
{
    const int lane = threadIdx.x & 63;
    const int env = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (env >= d.n) return;
    if (mask && !mask[env]) return;
    const size_t N = (size_t)d.n;
    Rng sim;
    sim.s0 = d.sim_rng[env];
    sim.s1 = d.sim_rng[N + env];
    SiRegs s;
    si_new_game(c, lane, sim, s);
    si_store(d, env, lane, s);
    if (lane == 0) {
        d.sim_rng[env] = sim.s0;
        d.sim_rng[N + env] = sim.s1;
        d.prev_score[env] = s.f[F_SCORE];
    }
}

// one frame (AGENT: the agent layer's whole action repeat, with MaxAndSkipEnv's bookkeeping) of one env on one wave.  Two
// instantiations because the kernel is bound by latency x occupancy: the batch protocol's form carries neither the slot
// structs nor the frame loop and needs fewer registers
// CANON: the engine's states are canonical (si_load_canonical) -- the batch protocol of an engine no intervention has taken off the grid
template <bool AGENT, bool CANON = false>
__device__ __forceinline__ void si_step_body(const SiDev& d, const SiDev& slot_a, const SiDev& slot_b, const SiCfg& c, const ActionSource& src, uint32_t flags, int env, int lane,
                                             SiRenderRec* __restrict__ recs = nullptr)
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
            a = tbx_legal_action(TBX_GAME_SPACE_INVADERS, (int)(h % 6ull));
        }
        buttons = tbx_ale_buttons(a);
        if (buttons == 0xFFu) {
            buttons = 0;
            if (lane == 0) atomicOr(d.err_flag, 1u);
        }
    }

    SiRegs s;
    SiLoaded loaded;
    if (CANON) si_load_canonical(d, c, env, lane, s, loaded);
    else si_load(d, env, lane, s, loaded);
    int32_t prev = d.prev_score[env];
    const int frames = AGENT && src.frames > 1 ? src.frames : 1;
    int32_t rew = 0, out_lives = 0, out_score = 0;
    bool is_done = false;
    for (int fr = 0; fr < frames; fr++) {                  // > 1: the agent layer's action repeat, state stays in registers
        si_step(c, lane, buttons, s);
        rew = s.f[F_SCORE] - prev;
        if (rew < 0) rew = 0;
        out_lives = s.f[F_LIVES]; out_score = s.f[F_SCORE];
        is_done = out_lives <= 0;
        prev = out_score;
        if (!AGENT && is_done && (flags & TBX_STEP_AUTO_RESET)) {   // (the agent layer resets through its own procedure)
            Rng sim;
            sim.s0 = d.sim_rng[env];
            sim.s1 = d.sim_rng[N + env];
            si_new_game(c, lane, sim, s);
            if (lane == 0) { d.sim_rng[env] = sim.s0; d.sim_rng[N + env] = sim.s1; }
            prev = s.f[F_SCORE];
        }
        if (AGENT) {
            if (lane == 0) tbx_accumulate(src, env, rew, is_done, fr);
            if (src.buf_valid) {                             // MaxAndSkipEnv's frame buffer: slot A after frame skip-2, B after skip-1
                const uint32_t slots = tbx_snap_slots(src, fr);
                if (slots & 1u) si_store(slot_a, env, lane, s);
                if (slots & 2u) si_store(slot_b, env, lane, s);
                if (slots && lane == 0) src.buf_valid[env] |= (uint8_t)slots;
                if (is_done) break;                          // ... and its loop ends with the game
            }
        }
    }
    if (AGENT) si_store(d, env, lane, s);                 // (the agent form keeps its registers for the frame loop)
    else si_store_changed(d, env, lane, s, loaded);
    if (!AGENT && recs) si_write_rec(recs + env, lane, s);  // the rasteriser's input, while the state is in registers
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

template <bool CANON>
__global__ __launch_bounds__(TBX_BLOCK) void si_step_kernel(SiDev d, SiCfg c, ActionSource src, uint32_t flags, int first_env, int count,
                                                            SiRenderRec* __restrict__ recs)
{
    const int lane = threadIdx.x & 63;
    const int rel = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (rel >= count) return;
    si_step_body<false, CANON>(d, d, d, c, src, flags, first_env + rel, lane, recs);
}

// records of envs [first_env, first_env + count) from their state (after a new game, a state write, an agent step ...)
__global__ __launch_bounds__(TBX_BLOCK) void si_rec_prep_kernel(SiDev d, SiRenderRec* __restrict__ recs, int first_env, int count)
{
    const int lane = threadIdx.x & 63;
    const int rel = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (rel >= count) return;
    SiRegs s;
    si_load(d, first_env + rel, lane, s);
    si_write_rec(recs + first_env + rel, lane, s);
}

__global__ __launch_bounds__(TBX_BLOCK) void si_agent_step_kernel(SiDev d, SiDev slot_a, SiDev slot_b, SiCfg c, ActionSource src, uint32_t flags, int first_env, int count)
{
    const int lane = threadIdx.x & 63;
    const int rel = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (rel >= count) return;
    si_step_body<true>(d, slot_a, slot_b, c, src, flags, first_env + rel, lane);
}

// reset-time wrappers of the agent layer for the envs flagged in r.kind (agent_device.hpp, AgentResetProc)
struct SiAgentEnv {
    const SiCfg& c;
    int lane;
    SiRegs& s;
    Rng& sim;
    const SiDev& slot_a;
    const SiDev& slot_b;
    int env;
    __device__ __forceinline__ void snapshot(int slot) { si_store(slot ? slot_b : slot_a, env, lane, s); }
    __device__ __forceinline__ void step(uint32_t buttons) { si_step(c, lane, buttons, s); }
    __device__ __forceinline__ void new_game() { si_new_game(c, lane, sim, s); }
    __device__ __forceinline__ int lives() const { return wave_uniform(s.f[F_LIVES]); }
    __device__ __forceinline__ int score() const { return wave_uniform(s.f[F_SCORE]); }
};

__global__ __launch_bounds__(TBX_BLOCK) void si_agent_reset_kernel(SiDev d, SiDev slot_a, SiDev slot_b, SiCfg c, AgentResetArgs r)
{
    const int lane = threadIdx.x & 63;
    // a persistent grid walks the compact list of flagged envs (or every env when there is no list)
    const int wave_id = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + (threadIdx.x >> 6)), n_waves = gridDim.x * TBX_WAVES_PER_BLOCK;
    const int total = r.list ? wave_uniform(*r.count) : d.n;
    for (int it = wave_id; it < total; it += n_waves) {
        const int env = r.list ? wave_uniform(r.list[it]) : it;
        if (wave_uniform((int)r.kind[env]) == 0) continue;
        const size_t N = (size_t)d.n;
        SiRegs s;
        si_load(d, env, lane, s);
        Rng sim;
        sim.s0 = d.sim_rng[env]; sim.s1 = d.sim_rng[N + env];
        AgentMonitor m = agent_monitor_load(r, env);
        SiAgentEnv ops{c, lane, s, sim, slot_a, slot_b, env};
        AgentResetProc<SiAgentEnv> proc{ops, r, m, r.env_offset + (uint64_t)env, wave_uniform(d.prev_score[env]),
                                        (uint32_t)wave_uniform((int)r.buf_valid[env]), r.noop_override ? wave_uniform(r.noop_override[env]) : 0, false};
        proc.run();
        si_store(d, env, lane, s);
        if (lane == 0) {
            d.sim_rng[env] = sim.s0; d.sim_rng[N + env] = sim.s1;
            d.prev_score[env] = proc.prev;
            agent_monitor_store(r, env, m, proc.valid, proc.obs_raw);
        }
    }
}

// ------------------------------------------------------------------ render

__constant__ uint32_t SI_SPR_A[TBX_SI_ENEMY_H] = TBX_SI_SPRITE_ENEMY_A;
__constant__ uint32_t SI_SPR_B[TBX_SI_ENEMY_H] = TBX_SI_SPRITE_ENEMY_B;
__constant__ uint32_t SI_SPR_BOOM[TBX_SI_ENEMY_H] = TBX_SI_SPRITE_BOOM;
__constant__ uint32_t SI_SPR_SHIP[TBX_SI_SHIP_H] = TBX_SI_SPRITE_SHIP;
__constant__ uint32_t SI_SPR_D1[TBX_SI_SHIP_H] = TBX_SI_SPRITE_SHIP_D1;
__constant__ uint32_t SI_SPR_D2[TBX_SI_SHIP_H] = TBX_SI_SPRITE_SHIP_D2;
__constant__ uint32_t SI_SPR_UFO[TBX_SI_UFO_H] = TBX_SI_SPRITE_UFO;


constexpr int SI_UNIT_ROWS = 6;   // 210 = 35 units; 6 x 960 B (RGB) = 5760 B of LDS per wave

// paints sprite row bits (bit k = column k, `w` <= 32 columns) at x position sx into the lane's pixel groups
template <int NG>
__device__ __forceinline__ void paint_bits(uint32_t (&px)[NG][4], const int (&gx)[NG], int sx, uint32_t bits, int w, uint32_t col)
{
    const uint64_t live = (uint64_t)(w >= 32 ? bits : (bits & ((1u << w) - 1u))) << 4;   // 4 guard bits below column 0
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int k = gx[g] - sx + 4;                       // bit index of this group's first pixel in `live`
        if (k >= 0 && k < 40) {
            const uint32_t four = (uint32_t)(live >> k) & 15u;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if ((four >> i) & 1u) px[g][i] = col;
        }
    }
}

template <int NG>
__device__ __forceinline__ void paint_span(uint32_t (&px)[NG][4], const int (&gx)[NG], long rx0, long rx1, uint32_t col)
{
#pragma unroll
    for (int g = 0; g < NG; g++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const long x = gx[g] + i;
            if (x >= rx0 && x < rx1) px[g][i] = col;
        }
    }
}

// Everything one wave needs to paint scanlines of one env: lane l makes the 4-pixel groups l and l+64 (320 px = 80
// groups); lanes also hold the entities (lane = enemy / shield row / laser slot).  Built once per frame by setup();
// paint_row() then composes one busy scanline in the checker's order (shields, enemies, ufo, ship, lasers, HUD).
constexpr int SI_NG = 2;
// the block's LDS copy of every sprite row (si_fill_sprites): enemy pose A, pose B, explosion, ship, ship death 1 / 2, ufo
constexpr int SPR_SHIP = 3 * TBX_SI_ENEMY_H, SPR_D1 = SPR_SHIP + TBX_SI_SHIP_H, SPR_D2 = SPR_D1 + TBX_SI_SHIP_H;
constexpr int SPR_UFO = SPR_D2 + TBX_SI_SHIP_H, SPR_WORDS = SPR_UFO + TBX_SI_UFO_H;

template <int C>
struct SiPainter {
    typedef SiDev Dev;
    static constexpr int W = TBX_SI_W, H = TBX_SI_H, NG = SI_NG;
    enum { CLS_ENEMY, CLS_SHIELD, CLS_LASER, CLS_HUD, CLS_UFO, CLS_SHIP, NCLS, NLDS = NCLS };
    SiRegs s;
    int lane;
    int gx[SI_NG];
    bool gact[SI_NG];
    uint32_t hud[SI_NG][4];                 // bit 3*r = lit in glyph row r
    uint32_t c_enemy, c_ufo, c_ground, c_hud, c_black, c_ship, l_col, s_c;
    bool ufo_on, s_valid, e_vis, l_on;
    int s_x, s_y;
    int e_y0, e_y1, s_y0, s_y1;             // scanline ranges that can hold enemies / shield rows at all (wave-uniform; clipped to the
    int l_lo, l_hi;                         // ... lasers                                           256 rows of the class masks)
    uint64_t cand[SI_NG];                   // per pixel group: visible enemies whose columns overlap it (bit e)
    int e_tab;                              // lane = enemy: its sprite table base in spr_lds
    uint64_t busy[4];                       // scanlines that show anything but black (wave-uniform, 256 bits)
    const uint32_t* spr_lds;
    mutable uint64_t ym_cached;             // enemy lookups of the current formation row (paint_row)
    mutable int ec_shift[SI_NG], ec_row0[SI_NG];
    mutable bool ec_multi;
    // fused agent observation, scanlines that hold nothing but enemies (fast_*): per output column of this lane the visible
    // enemies whose columns reach its 8-pixel tap window, and the lookups of the current formation row
    static constexpr bool FAST_ROWS = true;
    static constexpr bool SPARSE_ROWS = true;    // (agent_fused_wave: walk the active scanlines only -- about half of the 210 scanlines hold nothing)
    int f_start[2];                         // first source pixel of this lane's two output columns (-1: column not in use)
    uint64_t fcand[2];                      // per window: visible enemies whose columns reach it (bit e)
    mutable uint64_t fym_cached;
    mutable uint32_t f_hit[2];              // (enemy x + 64) << 16 | (sprite-table row of scanline 0 + 1024); ~0u: no enemy under the window
    mutable bool f_multi;

    // spr_lds (set by the caller first): the block's copy of the three enemy sprites (si_fill_sprites); cls: [NCLS][8]
    // dwords of LDS private to this wave
    __device__ __forceinline__ void setup(const SiDev& d, int env, int lane_, uint32_t* cls)
    {
        lane = lane_;
        si_load(d, env, lane, s);
        const int32_t* f = s.f;
        const int ne = f[F_N_ENEMIES];
        gx[0] = lane * 4; gx[1] = (lane + 64) * 4;
        gact[0] = true; gact[1] = lane + 64 < TBX_SI_W / 4;
        {
            int sc = f[F_SCORE];
            if (sc < 0) sc = 0;
            sc %= 100000;
            int lv = f[F_LIVES];
            lv = lv < 0 ? 0 : lv > 9 ? 9 : lv;
            int le = f[F_LEVEL];
            if (le < 0) le = 0;
            le %= 10;
            const int hud_x0[7] = {36, 44, 52, 60, 68, 148, 196};
#pragma unroll
            for (int g = 0; g < SI_NG; g++)
#pragma unroll
                for (int i = 0; i < 4; i++) hud[g][i] = 0;
            int div = 10000;
#pragma unroll
            for (int q = 0; q < 7; q++) {
                int digit;
                if (q < 5) { digit = (sc / div) % 10; div /= 10; }
                else digit = q == 5 ? lv : le;
                const uint32_t glyph = tbx_digit_glyph((uint32_t)digit);
#pragma unroll
                for (int g = 0; g < SI_NG; g++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int dx = gx[g] + i - hud_x0[q];
                        if (dx >= 0 && dx < 6) hud[g][i] = (glyph >> (dx >> 1)) & 0x1249u;
                    }
            }
        }
        // every colour goes through pix_of<C>() once here; the scanline loop only moves finished pixel values
        c_enemy = pix_of<C>(rgb_u32(TBX_SI_COL_ENEMY)); c_ufo = pix_of<C>(rgb_u32(TBX_SI_COL_UFO));
        c_ground = pix_of<C>(rgb_u32(TBX_SI_COL_GROUND)); c_hud = pix_of<C>(rgb_u32(TBX_SI_COL_HUD));
        c_black = pix_of<C>(0xFF000000u); c_ship = pix_of<C>((uint32_t)f[F_SHIP_COLOR]);
        l_col = pix_of<C>((uint32_t)s.lf[LF_COLOR]);
        ufo_on = f[F_UFO_APP] == 0 || f[F_UFO_DC] >= 0;
        const int sk = lane / TBX_SI_SHIELD_H, sr = lane - sk * TBX_SI_SHIELD_H;
        s_valid = lane < TBX_SI_MAX_SHIELDS * TBX_SI_SHIELD_H && sk < f[F_N_SHIELDS];
        s_x = sel3(sk, f[F_SHIELD_X0], f[F_SHIELD_X1], f[F_SHIELD_X2]);
        s_y = sel3(sk, f[F_SHIELD_Y0], f[F_SHIELD_Y1], f[F_SHIELD_Y2]) + sr;
        s_c = pix_of<C>((uint32_t)sel3(sk, f[F_SHIELD_C0], f[F_SHIELD_C1], f[F_SHIELD_C2]));
        e_vis = lane < ne && (e_alive(s) || e_dc(s) >= 0);
        l_on = lane == SHIP_SLOT ? f[F_HAS_SHIP_LASER] != 0 : lane < f[F_N_LASERS];
        // (the scanline ranges e_y0 .. l_hi come out of the class masks below: first and last set bit, scalar -- round 5; until then
        // they were six 6-step wave reductions, 36 ds_bpermute per set-up, and the set-ups are 29 % of the agent observation kernel)

        cand[0] = cand[1] = 0ull;
        for (uint64_t m = __ballot(e_vis); m;) {                // one turn per distinct x (six for a formation), not per enemy
            const int e = (int)__builtin_ctzll(m);
            const int ex = __builtin_amdgcn_readlane(s.ex, e);
            const uint64_t same = __ballot(e_vis && s.ex == ex);
            m &= ~same;
#pragma unroll
            for (int g = 0; g < SI_NG; g++)
                if (gx[g] + 3 >= ex && gx[g] < ex + TBX_SI_ENEMY_W) cand[g] |= same;
        }
        e_tab = (s.estatus & 1) ? (f[F_ORIENT] ? 0 : TBX_SI_ENEMY_H) : 2 * TBX_SI_ENEMY_H;
        ym_cached = 0ull; ec_multi = false;
        ec_shift[0] = ec_shift[1] = 31; ec_row0[0] = ec_row0[1] = 0;

        // scanline masks per class in LDS: every lane ORs the rows of the entities it holds (enemy, shield row, laser);
        // HUD + ground, ufo and ship come from wave-uniform fields.  busy = their union.
        for (int i = lane; i < NCLS * 8; i += 64) cls[i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const long ly0 = s.lf[LF_Y];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t we = e_vis ? row_range_bits(s.ey, (long)s.ey + TBX_SI_ENEMY_H, k) : 0ull;
            const uint64_t ws = (s_valid && s.srow != 0) ? row_range_bits(s_y, (long)s_y + 1, k) : 0ull;
            const uint64_t wl = l_on ? row_range_bits(ly0, ly0 + s.lf[LF_H], k) : 0ull;
            if ((uint32_t)we) atomicOr(&cls[CLS_ENEMY * 8 + 2 * k], (uint32_t)we);
            if ((uint32_t)(we >> 32)) atomicOr(&cls[CLS_ENEMY * 8 + 2 * k + 1], (uint32_t)(we >> 32));
            if ((uint32_t)ws) atomicOr(&cls[CLS_SHIELD * 8 + 2 * k], (uint32_t)ws);
            if ((uint32_t)(ws >> 32)) atomicOr(&cls[CLS_SHIELD * 8 + 2 * k + 1], (uint32_t)(ws >> 32));
            if ((uint32_t)wl) atomicOr(&cls[CLS_LASER * 8 + 2 * k], (uint32_t)wl);
            if ((uint32_t)(wl >> 32)) atomicOr(&cls[CLS_LASER * 8 + 2 * k + 1], (uint32_t)(wl >> 32));
            if (lane == 0) {
                const uint64_t wh = row_range_bits(2, 12, k) | row_range_bits(TBX_SI_GROUND_Y, TBX_SI_GROUND_Y + 1, k);
                const uint64_t wu = ufo_on ? row_range_bits(f[F_UFO_Y], (long)f[F_UFO_Y] + TBX_SI_UFO_H, k) : 0ull;
                const uint64_t wp = ((f[F_SHIP_FLAGS] & 1) || f[F_SHIP_DC] >= 0) ? row_range_bits(f[F_SHIP_Y], (long)f[F_SHIP_Y] + TBX_SI_SHIP_H, k) : 0ull;
                cls[CLS_HUD * 8 + 2 * k] = (uint32_t)wh; cls[CLS_HUD * 8 + 2 * k + 1] = (uint32_t)(wh >> 32);
                cls[CLS_UFO * 8 + 2 * k] = (uint32_t)wu; cls[CLS_UFO * 8 + 2 * k + 1] = (uint32_t)(wu >> 32);
                cls[CLS_SHIP * 8 + 2 * k] = (uint32_t)wp; cls[CLS_SHIP * 8 + 2 * k + 1] = (uint32_t)(wp >> 32);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int y0[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, y1[3] = {INT32_MIN, INT32_MIN, INT32_MIN};   // CLS_ENEMY, CLS_SHIELD, CLS_LASER
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t lo = 0u, hi = 0u;
#pragma unroll
            for (int c = 0; c < NCLS; c++) {
                const uint32_t cl = (uint32_t)__builtin_amdgcn_readfirstlane(cls[c * 8 + 2 * k]), ch = (uint32_t)__builtin_amdgcn_readfirstlane(cls[c * 8 + 2 * k + 1]);
                lo |= cl; hi |= ch;
                if (c <= CLS_LASER) {
                    const uint64_t m = (uint64_t)cl | ((uint64_t)ch << 32);
                    if (m) {
                        if (y0[c] == INT32_MAX) y0[c] = 64 * k + (int)__builtin_ctzll(m);
                        y1[c] = 64 * k + 64 - (int)__builtin_clzll(m);
                    }
                }
            }
            busy[k] = (uint64_t)lo | ((uint64_t)hi << 32);
        }
        e_y0 = y0[CLS_ENEMY]; e_y1 = y1[CLS_ENEMY]; s_y0 = y0[CLS_SHIELD]; s_y1 = y1[CLS_SHIELD]; l_lo = y0[CLS_LASER]; l_hi = y1[CLS_LASER];
        __builtin_amdgcn_wave_barrier();
    }

    // ---- agent_fused_wave's shortcut: on a scanline that shows only enemies the gray row is c_enemy where a sprite bit is set and
    // 0 elsewhere, so an output column's horizontal sum is c_enemy x (tap weights . sprite bits under the window) -- no painting,
    // no staging in LDS.  Exactly the sum hsum() takes of the painted row.
    // dword w (of 8) of the scanline mask: enemy rows that no other class touches
    static __device__ __forceinline__ uint32_t fast_row_word(const uint32_t* cls, int w)
    {
        uint32_t other = 0u;
#pragma unroll
        for (int c = 0; c < NCLS; c++)
            if (c != CLS_ENEMY) other |= cls[c * 8 + w];
        return cls[CLS_ENEMY * 8 + w] & ~other;
    }

    __device__ __forceinline__ void fast_init(const ColTaps& c0, const ColTaps& c1, bool on0, bool on1)
    {
        f_start[0] = on0 ? c0.start : -1; f_start[1] = on1 ? c1.start : -1;
        fym_cached = 0ull; f_multi = false;
        f_hit[0] = f_hit[1] = ~0u;
        // per window: the visible enemies whose columns reach it -- one loop turn per distinct enemy x, like `cand` in setup()
        fcand[0] = fcand[1] = 0ull;
        for (uint64_t m = __ballot(e_vis); m;) {
            const int e = (int)__builtin_ctzll(m);
            const int ex = __builtin_amdgcn_readlane(s.ex, e);
            const uint64_t same = __ballot(e_vis && s.ex == ex);
            m &= ~same;
#pragma unroll
            for (int q = 0; q < 2; q++)
                if (f_start[q] >= 0 && f_start[q] + 8 > ex && f_start[q] < ex + TBX_SI_ENEMY_W) fcand[q] |= same;
        }
    }

    // looks up the enemies crossing scanline y under this lane's two windows (once per formation row: the set is the same
    // for its ten scanlines); false if some window holds two of them (states written by hand): the scanline then takes the
    // painted path.  Round 5: the one enemy of a window comes out of fcand & ym with three cross-lane reads, where a serial loop
    // over the row's enemies (find-first-set -> readlane -> compare, six dependent turns per formation row) used to look for it.
    __device__ __forceinline__ bool fast_ready(int y) const
    {
        const uint64_t ym = __ballot(e_vis && y >= s.ey && y < s.ey + TBX_SI_ENEMY_H);
        if (ym != fym_cached) {
            fym_cached = ym;
            bool multi = false;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint64_t c = fcand[q] & ym;
                const int e = c ? (int)__builtin_ctzll(c) : 0;
                const int ex = __shfl(s.ex, e), ey = __shfl(s.ey, e), tab = __shfl(e_tab, e);
                f_hit[q] = c ? ((uint32_t)(ex + 64) << 16) | (uint32_t)(tab - ey + 1024) : ~0u;
                multi |= (c & (c - 1)) != 0;
            }
            f_multi = __ballot(multi) != 0;
        }
        return !f_multi;
    }

    __device__ __forceinline__ void fast_sums(int y, const ColTaps& c0, const ColTaps& c1, bool, bool, uint32_t& h0, uint32_t& h1) const
    {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            uint32_t sum = 0u;
            if (f_hit[q] != ~0u) {
                const int ex = (int)(f_hit[q] >> 16) - 64, row0 = (int)(f_hit[q] & 0xFFFFu) - 1024;
                const uint32_t bits = spr_lds[row0 + y] & ((1u << TBX_SI_ENEMY_W) - 1u);      // bit k = pixel ex + k
                const int sh = f_start[q] - ex;                        // window pixel i is sprite bit sh + i; -8 < sh < 16
                const uint32_t wb = (sh >= 0 ? bits >> sh : bits << -sh) & 0xFFu;
                const uint32_t lo = ((wb & 15u) * 0x00204081u) & 0x01010101u, hi = ((wb >> 4) * 0x00204081u) & 0x01010101u;
                sum = c_enemy * __builtin_amdgcn_udot4(lo, q ? c1.wlo : c0.wlo, __builtin_amdgcn_udot4(hi, q ? c1.whi : c0.whi, 0u, false), false);
            }
            (q ? h1 : h0) = sum;
        }
    }

    // the classes whose entities differ between two states of one env (wave-uniform bit mask)
    static __device__ __forceinline__ uint32_t diff_classes(const SiPainter& a, const SiPainter& b)
    {
        uint32_t m = 0u;
        if (__ballot(a.e_vis != b.e_vis || (b.e_vis && (a.s.ex != b.s.ex || a.s.ey != b.s.ey || a.e_tab != b.e_tab)))) m |= 1u << CLS_ENEMY;
        if (__ballot(a.s_valid != b.s_valid || (b.s_valid && (a.s.srow != b.s.srow || a.s_x != b.s_x || a.s_y != b.s_y || a.s_c != b.s_c))))
            m |= 1u << CLS_SHIELD;
        if (__ballot(a.l_on != b.l_on || (b.l_on && (a.s.lf[LF_X] != b.s.lf[LF_X] || a.s.lf[LF_Y] != b.s.lf[LF_Y] || a.s.lf[LF_W] != b.s.lf[LF_W] ||
                                                     a.s.lf[LF_H] != b.s.lf[LF_H] || a.l_col != b.l_col))))
            m |= 1u << CLS_LASER;
        bool hud_diff = false;
#pragma unroll
        for (int g = 0; g < SI_NG; g++)
#pragma unroll
            for (int i = 0; i < 4; i++) hud_diff |= a.hud[g][i] != b.hud[g][i];
        if (__ballot(hud_diff)) m |= 1u << CLS_HUD;
        const int32_t *fa = a.s.f, *fb = b.s.f;
        if (a.ufo_on != b.ufo_on || (b.ufo_on && (fa[F_UFO_X] != fb[F_UFO_X] || fa[F_UFO_Y] != fb[F_UFO_Y]))) m |= 1u << CLS_UFO;
        const bool sa = (fa[F_SHIP_FLAGS] & 1) || fa[F_SHIP_DC] >= 0, sb = (fb[F_SHIP_FLAGS] & 1) || fb[F_SHIP_DC] >= 0;
        if (sa != sb || (sb && (fa[F_SHIP_X] != fb[F_SHIP_X] || fa[F_SHIP_Y] != fb[F_SHIP_Y] || fa[F_SHIP_FLAGS] != fb[F_SHIP_FLAGS] ||
                                (fa[F_SHIP_DC] >= 0) != (fb[F_SHIP_DC] >= 0) || a.c_ship != b.c_ship)))
            m |= 1u << CLS_SHIP;
        return wave_uniform((int)m);
    }

    // one busy scanline as finished pixel values
    __device__ __forceinline__ void paint_row(int y, uint32_t (&px)[SI_NG][4]) const
    {
        constexpr int NG = SI_NG;
        const int32_t* f = s.f;
        const uint32_t base = y == TBX_SI_GROUND_Y ? c_ground : c_black;
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int i = 0; i < 4; i++) px[g][i] = base;
        // shields (ascending shield index, then row: one row per shield can match y)
        if (y >= s_y0 && y < s_y1) {
            uint64_t m = __ballot(s_valid && s_y == y && s.srow != 0);
            while (m) {
                const int src = (int)__builtin_ctzll(m);
                m &= m - 1;
                paint_bits<NG>(px, gx, bcast(s_x, src), bcast(s.srow, src), TBX_SI_SHIELD_W, bcast(s_c, src));
            }
        }
        // enemies in index order.  The set of enemies crossing a scanline (ym) is the same for the ten scanlines of a
        // formation row, so what each lane needs of its (usually single) overlapping enemy is looked up once per set:
        // the shift of the sprite row into its pixel group and the sprite-table offset.  A lane with two overlapping
        // enemies in a group (only states written by hand) takes the general walk below.
        if (y >= e_y0 && y < e_y1) {
            const uint64_t ym = __ballot(e_vis && y >= s.ey && y < s.ey + TBX_SI_ENEMY_H);
            if (ym != ym_cached) {
                ym_cached = ym;
                bool multi = false;
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    const uint64_t c = cand[g] & ym;
                    const int e = c ? (int)__builtin_ctzll(c) : 0;
                    const int ex = __shfl(s.ex, e), ey = __shfl(s.ey, e), tab = __shfl(e_tab, e);
                    ec_shift[g] = c ? gx[g] - ex + 4 : 31;        // 31: every sprite bit shifted out
                    ec_row0[g] = tab - ey;
                    multi |= (c & (c - 1)) != 0;
                }
                ec_multi = __ballot(multi) != 0;
            }
            if (!ec_multi) {
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    const int idx = ec_row0[g] + y;
                    const uint32_t bits = ec_shift[g] != 31 ? spr_lds[idx] & ((1u << TBX_SI_ENEMY_W) - 1u) : 0u;
                    const uint32_t four = ((bits << 4) >> ec_shift[g]) & 15u;
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if ((four >> i) & 1u) px[g][i] = c_enemy;
                }
            } else {
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    uint64_t c = cand[g] & ym;
                    while (__ballot(c != 0)) {
                        const bool on = c != 0;
                        const int e = on ? (int)__builtin_ctzll(c) : 0;
                        const int ex = __shfl(s.ex, e), ey = __shfl(s.ey, e), tab = __shfl(e_tab, e);
                        if (on) {
                            const uint32_t bits = spr_lds[tab + (y - ey)] & ((1u << TBX_SI_ENEMY_W) - 1u);
                            const uint32_t four = ((bits << 4) >> (gx[g] - ex + 4)) & 15u;    // shift in [1, 19]
#pragma unroll
                            for (int i = 0; i < 4; i++)
                                if ((four >> i) & 1u) px[g][i] = c_enemy;
                            c &= c - 1;
                        }
                    }
                }
            }
        }
        if (ufo_on && y >= f[F_UFO_Y] && y < f[F_UFO_Y] + TBX_SI_UFO_H)
            paint_bits<NG>(px, gx, f[F_UFO_X], spr_lds[SPR_UFO + (y - f[F_UFO_Y])], TBX_SI_UFO_W, c_ufo);
        if (y >= f[F_SHIP_Y] && y < f[F_SHIP_Y] + TBX_SI_SHIP_H) {
            const int ry = y - f[F_SHIP_Y];
            if (f[F_SHIP_FLAGS] & 1) paint_bits<NG>(px, gx, f[F_SHIP_X], spr_lds[SPR_SHIP + ry], 16, c_ship);
            else if (f[F_SHIP_DC] >= 0)
                paint_bits<NG>(px, gx, f[F_SHIP_X], spr_lds[((f[F_SHIP_FLAGS] & 2) ? SPR_D1 : SPR_D2) + ry], 16, c_ship);
        }
        // lasers: the ship's first, then enemy lasers in slot order
        if (y >= l_lo && y < l_hi) {
            const long ly0 = s.lf[LF_Y], ly1 = (long)s.lf[LF_Y] + s.lf[LF_H];
            uint64_t m = __ballot(l_on && y >= ly0 && y < ly1);
            if ((m >> SHIP_SLOT) & 1) {
                const Laser l = get_laser(s, SHIP_SLOT);
                paint_span<NG>(px, gx, l.x, (long)l.x + l.w, bcast(l_col, SHIP_SLOT));
            }
            m &= (1ull << SHIP_SLOT) - 1;
            while (m) {
                const int src = (int)__builtin_ctzll(m);
                m &= m - 1;
                const long lx = bcast(s.lf[LF_X], src), lw = bcast(s.lf[LF_W], src);
                paint_span<NG>(px, gx, lx, lx + lw, bcast(l_col, src));
            }
        }
        if (y >= 2 && y < 12) {
            const int gr = ((y - 2) >> 1) * 3;
#pragma unroll
            for (int g = 0; g < NG; g++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if ((hud[g][i] >> gr) & 1u) px[g][i] = c_hud;
        }
    }
};

// (SiPainter, continued) packed gray bytes for the fused agent path
template <int C>
__device__ __forceinline__ void si_row_dwords(const SiPainter<C>& p, int y, uint32_t (&v)[SI_NG])
{
    uint32_t px[SI_NG][4];
    p.paint_row(y, px);
#pragma unroll
    for (int g = 0; g < SI_NG; g++) v[g] = px[g][0] | (px[g][1] << 8) | (px[g][2] << 16) | (px[g][3] << 24);
}

struct SiGrayPainter : SiPainter<1> {
    static __device__ __forceinline__ uint32_t diff_classes(const SiGrayPainter& a, const SiGrayPainter& b) { return SiPainter<1>::diff_classes(a, b); }
    __device__ __forceinline__ uint32_t blank_dword() const { return 0u; }       // black is gray 0
    __device__ __forceinline__ void row_dwords(int y, uint32_t (&v)[SI_NG]) const { si_row_dwords<1>(*this, y, v); }
    uint64_t rep[4] = {0ull, 0ull, 0ull, 0ull};                                   // sprite rows differ scanline by scanline
};

// fills the block's LDS copy of the sprite rows (scalar loads from constant memory inside the scanline loop stalled the
// wave once per sprite row); ends with a block barrier
__device__ __forceinline__ void si_fill_sprites(uint32_t* spr_lds)
{
    for (int t = threadIdx.x; t < SPR_WORDS; t += blockDim.x) {
        uint32_t v;
        if (t < TBX_SI_ENEMY_H) v = SI_SPR_A[t];
        else if (t < 2 * TBX_SI_ENEMY_H) v = SI_SPR_B[t - TBX_SI_ENEMY_H];
        else if (t < SPR_SHIP) v = SI_SPR_BOOM[t - 2 * TBX_SI_ENEMY_H];
        else if (t < SPR_D1) v = SI_SPR_SHIP[t - SPR_SHIP];
        else if (t < SPR_D2) v = SI_SPR_D1[t - SPR_D1];
        else if (t < SPR_UFO) v = SI_SPR_D2[t - SPR_D2];
        else v = SI_SPR_UFO[t - SPR_UFO];
        spr_lds[t] = v;
    }
    __syncthreads();
}

// units part, part + split, ... of one env's frame from a painter that has been set up (SiPainter or SiRecPainter), on one
// wave: the body of the render kernels and of the resident single-env kernel's paint request.  Blank units are stored directly.
template <int C, class Painter>
__device__ __forceinline__ void si_paint_units(const Painter& p, uint8_t* __restrict__ frame, int env, int lane, const RowStager<C, TBX_SI_W, SI_UNIT_ROWS>& st,
                                               int part, int split, int skip_blank = 1)
{
    constexpr int NG = SI_NG;
    using Stager = RowStager<C, TBX_SI_W, SI_UNIT_ROWS>;
    constexpr int NUNITS = TBX_SI_H / SI_UNIT_ROWS;
    const int u0 = split > 1 ? 0 : (int)(((uint32_t)env * 11u) % (uint32_t)NUNITS);
    for (int k = part; k < NUNITS; k += split) {
        int u = u0 + k;
        if (u >= NUNITS) u -= NUNITS;
        const uint32_t rows_busy = (skip_blank & 1) ? row_mask_chunk<SI_UNIT_ROWS>(p.busy, u * SI_UNIT_ROWS) : (1u << SI_UNIT_ROWS) - 1u;
        if (rows_busy == 0 && C != 4) {                      // nothing but background: no staging (for RGBA the
            Stager::fill_unit(frame + (size_t)u * Stager::UNIT_BYTES, lane, p.c_black);   // staged path measured faster)
            continue;
        }
#pragma unroll 1
        for (int r = 0; r < SI_UNIT_ROWS; r++) {
            const int y = u * SI_UNIT_ROWS + r;
            uint32_t px[NG][4];
            if (((rows_busy >> r) & 1u) && !(skip_blank & 2)) p.paint_row(y, px);
            else {
#pragma unroll
                for (int g = 0; g < NG; g++)
#pragma unroll
                    for (int i = 0; i < 4; i++) px[g][i] = p.c_black;
            }
#pragma unroll
            for (int g = 0; g < NG; g++)
                if (p.gact[g]) st.put4p(r, lane + 64 * g, px[g][0], px[g][1], px[g][2], px[g][3]);
        }
        if (!(skip_blank & 4)) st.flush(frame + (size_t)u * Stager::UNIT_BYTES, lane);
    }
}

// One wave rasterises one env, scanline by scanline; SI_UNIT_ROWS scanlines are staged in LDS and flushed as 16-byte
// stores, blank units are stored directly.  `split` > 1: that many waves share a frame.
// (agent layer, generic path: envs flagged in pick_alt are painted from d_alt)
template <int C, bool ALT>
__global__ __launch_bounds__(TBX_BLOCK) void si_render_kernel(SiDev d, uint8_t* out, int first_env, int count, int skip_blank, int split, SiDev d_alt,
                                                              const uint8_t* __restrict__ pick_alt)
{
    constexpr int W = TBX_SI_W, H = TBX_SI_H;
    using Stager = RowStager<C, W, SI_UNIT_ROWS>;
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[TBX_WAVES_PER_BLOCK * Stager::UNIT_BYTES];
    __shared__ uint32_t lds_mask[TBX_WAVES_PER_BLOCK][SiPainter<C>::NCLS * 8];
    __shared__ uint32_t spr_lds[SPR_WORDS];
    si_fill_sprites(spr_lds);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wid = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + wave);   // `split` waves share a frame (see breakout.hip)
    const int rel = wid / split, part = wid - rel * split;
    if (rel >= count) return;
    const int env = first_env + rel;
    Stager st{lds_all + wave * Stager::UNIT_BYTES};
    SiPainter<C> p;
    p.spr_lds = spr_lds;
    SiDev src = d;                                            // by VALUE: a select between references to kernel arguments puts both into scratch
    if (ALT && pick_alt && wave_uniform((int)pick_alt[env])) src = d_alt;   // (ALT: the agent layer's generic path only)
    if (C == 3) tbx_stagger_first_waves(wid);
    p.setup(src, env, lane, lds_mask[wave]);

    si_paint_units<C>(p, out + (size_t)rel * H * W * C, env, lane, st, part, split, skip_blank);
}

// ------------------------------------------------------------------ the rasteriser over render records
//
// Same scanline composition and paint order as SiPainter (shields, enemies, ufo, ship, lasers, HUD), from a SiRenderRec.  The
// header arrives by scalar loads; the enemy under a pixel group follows from the formation origin by arithmetic (no candidate
// masks, no lane shuffles); lane = shield row and lane = laser slot keep their roles.  Which scanlines show anything but
// background is worked out lane-parallel (lane l judges scanlines l, l + 64, l + 128, l + 192; four ballots).
template <int C>
struct SiRecPainter {
    static constexpr int NG = SI_NG;
    int lane;
    int gx[SI_NG];
    bool gact[SI_NG];
    uint32_t hud[SI_NG][4];
    uint32_t c_enemy, c_ufo, c_ground, c_hud, c_black, c_ship;
    // wave-uniform
    int fx, fy, ship_x, ship_y, ufo_x, ufo_y, n_enemies, n_rows, n_shields, pose;
    uint64_t vis, alive;
    bool orient, ufo_on;
    int sh_x[TBX_SI_MAX_SHIELDS], sh_y[TBX_SI_MAX_SHIELDS];
    uint32_t sh_c[TBX_SI_MAX_SHIELDS];
    int e_y0, e_y1, l_lo, l_hi;
    uint64_t busy[4];
    // per lane
    uint32_t srow;                         // lane = shield * 18 + row
    uint32_t lz_x, lz_y, lz_c;             // lane = laser slot (clipped spans, finished colour)
    int ecol[SI_NG], eshift[SI_NG];        // formation column whose sprites reach this pixel group (-1: none), sprite-row shift
    const uint32_t* spr_lds;

    __device__ __forceinline__ void setup(const SiRenderRec* __restrict__ recs, int env, int lane_)
    {
        lane = lane_;
        const SiRenderRec* rec = recs + env;
        // the header by ONE vector load (lane i = dword i) and a v_readlane per dword (same-box A/B against twenty scalar loads:
        // 2.39 against 2.43 ms per launch at 65 536 envs)
        const uint32_t hv = lane < REC_HDR_DWORDS ? reinterpret_cast<const uint32_t*>(rec)[lane] : 0u;
        uint32_t h[REC_HDR_DWORDS];
#pragma unroll
        for (int i = 0; i < REC_HDR_DWORDS; i++) h[i] = (uint32_t)__builtin_amdgcn_readlane((int)hv, i);
        fx = (int)h[0]; fy = (int)h[1];
        vis = (uint64_t)h[2] | ((uint64_t)h[3] << 32);
        alive = (uint64_t)h[4] | ((uint64_t)h[5] << 32);
        const uint32_t flags = h[6];
        const uint32_t hudw = h[7];
        ship_x = (int)h[8]; ship_y = (int)h[9];
        c_ship = pix_of<C>(h[10]);
        ufo_x = (int)h[11]; ufo_y = (int)h[12];
        n_enemies = (int)(flags & 127u); orient = (flags & REC_ORIENT) != 0; ufo_on = (flags & REC_UFO) != 0;
        pose = (int)((flags >> REC_SHIP_SHIFT) & 3u); n_shields = (int)((flags >> REC_SHIELDS_SHIFT) & 3u);
        n_rows = (n_enemies + TBX_SI_COLS - 1) / TBX_SI_COLS;
#pragma unroll
        for (int k = 0; k < TBX_SI_MAX_SHIELDS; k++) {
            sh_x[k] = (int)(int16_t)(h[13 + k] & 0xFFFFu); sh_y[k] = (int)(int16_t)(h[13 + k] >> 16);
            sh_c[k] = pix_of<C>(h[16 + k]);
        }
        srow = rec->shield_rows[lane];
        lz_x = lane < 12 ? rec->laser_x[lane] : 0u;
        lz_y = lane < 12 ? rec->laser_y[lane] : 0u;
        lz_c = pix_of<C>(lane < 12 ? rec->laser_color[lane] : 0u);

        gx[0] = lane * 4; gx[1] = (lane + 64) * 4;
        gact[0] = true; gact[1] = lane + 64 < TBX_SI_W / 4;
        {
            const int hud_x0[7] = {36, 44, 52, 60, 68, 148, 196};
#pragma unroll
            for (int g = 0; g < SI_NG; g++)
#pragma unroll
                for (int i = 0; i < 4; i++) hud[g][i] = 0;
#pragma unroll
            for (int q = 0; q < 7; q++) {
                const uint32_t glyph = tbx_digit_glyph((hudw >> (4 * q)) & 15u);
#pragma unroll
                for (int g = 0; g < SI_NG; g++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int dx = gx[g] + i - hud_x0[q];
                        if (dx >= 0 && dx < 6) hud[g][i] = (glyph >> (dx >> 1)) & 0x1249u;
                    }
            }
        }
        c_enemy = pix_of<C>(rgb_u32(TBX_SI_COL_ENEMY)); c_ufo = pix_of<C>(rgb_u32(TBX_SI_COL_UFO));
        c_ground = pix_of<C>(rgb_u32(TBX_SI_COL_GROUND)); c_hud = pix_of<C>(rgb_u32(TBX_SI_COL_HUD));
        c_black = pix_of<C>(0xFF000000u);
        // the one formation column whose 16-pixel sprites can reach this lane's 4-pixel group (columns are 32 apart)
#pragma unroll
        for (int g = 0; g < SI_NG; g++) {
            const int t = gx[g] + 3 - fx;
            const int c = t >> 5, m = t & 31;
            ecol[g] = (c >= 0 && c < TBX_SI_COLS && m < TBX_SI_ENEMY_W + 3) ? c : -1;
            eshift[g] = m + 1;                                   // gx - (fx + 32 c) + 4, in [1, 19]
        }
        // scanline ranges that can hold enemies / lasers at all
        e_y0 = INT32_MAX; e_y1 = INT32_MIN;
        for (int r = 0; r < n_rows; r++)
            if ((vis >> (TBX_SI_COLS * r)) & 63ull) { e_y0 = min(e_y0, fy + TBX_SI_ENEMY_DY * r); e_y1 = max(e_y1, fy + TBX_SI_ENEMY_DY * r + TBX_SI_ENEMY_H); }
        l_lo = INT32_MAX; l_hi = INT32_MIN;
#pragma unroll
        for (int i = 0; i <= SHIP_SLOT; i++) {
            const uint32_t ly = (uint32_t)__builtin_amdgcn_readlane((int)lz_y, i);
            if (ly) { l_lo = min(l_lo, (int)(ly & 0xFFFFu)); l_hi = max(l_hi, (int)(ly >> 16)); }
        }
        // busy scanlines: lane l judges l, l + 64, l + 128, l + 192
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = lane + 64 * k;
            bool on = (y >= 2 && y < 12) || y == TBX_SI_GROUND_Y;
            on |= ufo_on && y >= ufo_y && y < ufo_y + TBX_SI_UFO_H;
            on |= pose != 0 && y >= ship_y && y < ship_y + TBX_SI_SHIP_H;
            if (y >= e_y0 && y < e_y1) {
                const int dy = y - fy, r = dy / TBX_SI_ENEMY_DY;             // dy >= 0 here
                on |= dy - r * TBX_SI_ENEMY_DY < TBX_SI_ENEMY_H && r < n_rows && ((vis >> (TBX_SI_COLS * r)) & 63ull) != 0;
            }
#pragma unroll
            for (int q = 0; q < TBX_SI_MAX_SHIELDS; q++) on |= q < n_shields && y >= sh_y[q] && y < sh_y[q] + TBX_SI_SHIELD_H;
            if (y >= l_lo && y < l_hi) {
#pragma unroll
                for (int i = 0; i <= SHIP_SLOT; i++) {
                    const uint32_t ly = (uint32_t)__builtin_amdgcn_readlane((int)lz_y, i);
                    on |= y >= (int)(ly & 0xFFFFu) && y < (int)(ly >> 16);
                }
            }
            busy[k] = __ballot(on && y < TBX_SI_H);
        }
    }

    __device__ __forceinline__ void paint_row(int y, uint32_t (&px)[SI_NG][4]) const
    {
        constexpr int NG = SI_NG;
        const uint32_t base = y == TBX_SI_GROUND_Y ? c_ground : c_black;
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int i = 0; i < 4; i++) px[g][i] = base;
        // shields, ascending index
#pragma unroll
        for (int k = 0; k < TBX_SI_MAX_SHIELDS; k++) {
            const int r = y - sh_y[k];
            if (k < n_shields && r >= 0 && r < TBX_SI_SHIELD_H) {
                const uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)srow, wave_uniform(k * TBX_SI_SHIELD_H + r));
                if (bits) paint_bits<NG>(px, gx, sh_x[k], bits, TBX_SI_SHIELD_W, sh_c[k]);
            }
        }
        // enemies of the formation row that crosses this scanline
        if (y >= e_y0 && y < e_y1) {
            const int dy = y - fy, r = dy / TBX_SI_ENEMY_DY, ry = dy - r * TBX_SI_ENEMY_DY;
            if (ry < TBX_SI_ENEMY_H && r < n_rows) {
                const uint32_t v6 = (uint32_t)(vis >> (TBX_SI_COLS * r)) & 63u, a6 = (uint32_t)(alive >> (TBX_SI_COLS * r)) & 63u;
                if (v6) {
                    const uint32_t march = spr_lds[(orient ? 0 : TBX_SI_ENEMY_H) + ry] & ((1u << TBX_SI_ENEMY_W) - 1u);
                    const uint32_t boom = spr_lds[2 * TBX_SI_ENEMY_H + ry] & ((1u << TBX_SI_ENEMY_W) - 1u);
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        const int c = ecol[g];
                        if (c >= 0 && ((v6 >> c) & 1u)) {
                            const uint32_t bits = ((a6 >> c) & 1u) ? march : boom;
                            const uint32_t four = ((bits << 4) >> eshift[g]) & 15u;
#pragma unroll
                            for (int i = 0; i < 4; i++)
                                if ((four >> i) & 1u) px[g][i] = c_enemy;
                        }
                    }
                }
            }
        }
        if (ufo_on && y >= ufo_y && y < ufo_y + TBX_SI_UFO_H) paint_bits<NG>(px, gx, ufo_x, spr_lds[SPR_UFO + (y - ufo_y)], TBX_SI_UFO_W, c_ufo);
        if (pose != 0 && y >= ship_y && y < ship_y + TBX_SI_SHIP_H)
            paint_bits<NG>(px, gx, ship_x, spr_lds[(pose == 1 ? SPR_SHIP : pose == 2 ? SPR_D1 : SPR_D2) + (y - ship_y)], 16, c_ship);
        // lasers: the ship's first, then the enemies' in slot order
        if (y >= l_lo && y < l_hi) {
            uint64_t m = __ballot(lane <= SHIP_SLOT && y >= (int)(lz_y & 0xFFFFu) && y < (int)(lz_y >> 16));
            if ((m >> SHIP_SLOT) & 1) {
                const uint32_t lx = bcast(lz_x, SHIP_SLOT);
                paint_span<NG>(px, gx, (long)(lx & 0xFFFFu), (long)(lx >> 16), bcast(lz_c, SHIP_SLOT));
            }
            m &= (1ull << SHIP_SLOT) - 1;
            while (m) {
                const int src = (int)__builtin_ctzll(m);
                m &= m - 1;
                const uint32_t lx = bcast(lz_x, src);
                paint_span<NG>(px, gx, (long)(lx & 0xFFFFu), (long)(lx >> 16), bcast(lz_c, src));
            }
        }
        if (y >= 2 && y < 12) {
            const int gr = ((y - 2) >> 1) * 3;
#pragma unroll
            for (int g = 0; g < NG; g++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if ((hud[g][i] >> gr) & 1u) px[g][i] = c_hud;
        }
    }
};

template <int C>
__global__ __launch_bounds__(TBX_BLOCK) void si_rec_render_kernel(const SiRenderRec* __restrict__ recs, uint8_t* __restrict__ out, int first_env, int count, int split)
{
    constexpr int W = TBX_SI_W, H = TBX_SI_H;
    using Stager = RowStager<C, W, SI_UNIT_ROWS>;
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[TBX_WAVES_PER_BLOCK * Stager::UNIT_BYTES];
    __shared__ uint32_t spr_lds[SPR_WORDS];
    si_fill_sprites(spr_lds);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wid = wave_uniform(blockIdx.x * TBX_WAVES_PER_BLOCK + wave);
    const int rel = wid / split, part = wid - rel * split;
    if (rel >= count) return;
    const int env = first_env + rel;
    Stager st{lds_all + wave * Stager::UNIT_BYTES};
    SiRecPainter<C> p;
    p.spr_lds = spr_lds;
    if (C == 3) tbx_stagger_first_waves(wid);
    p.setup(recs, env, lane);
    si_paint_units<C>(p, out + (size_t)rel * H * W * C, env, lane, st, part, split);
}

// ------------------------------------------------------------------ resident single-env form (tbx_serve_loop, tbx_common.hpp)
//
// One wave, env 0: steps on request and, when the request asks for it, rasterises the env straight into the engine's mapped
// pinned frame buffer (ToyboxBaseEnv.step = apply_ale_action + get_state without a launch, a copy or a synchronisation).
// recs != nullptr: the formation is canonical, the step leaves env 0's record and the record painter paints it; else the
// state-reading painter does.
template <int C>
__device__ __forceinline__ void si_serve_paint(const SiDev& d, const SiRenderRec* recs, uint8_t* frame, int lane, uint8_t* lds, uint32_t* cls, const uint32_t* spr_lds,
                                               int part, int split)
{
    const RowStager<C, TBX_SI_W, SI_UNIT_ROWS> st{lds};
    if (recs) {
        SiRecPainter<C> p;
        p.spr_lds = spr_lds;
        p.setup(recs, 0, lane);
        si_paint_units<C>(p, frame, 0, lane, st, part, split);
    } else {
        SiPainter<C> p;
        p.spr_lds = spr_lds;
        p.setup(d, 0, lane, cls);
        si_paint_units<C>(p, frame, 0, lane, st, part, split);
    }
}

__global__ __launch_bounds__(64 * TBX_SERVE_WAVES) void si_serve_kernel(SiDev d, SiCfg c, SiRenderRec* recs, TbxServeCtl* ctl)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[TBX_SERVE_WAVES][RowStager<4, TBX_SI_W, SI_UNIT_ROWS>::UNIT_BYTES];
    __shared__ uint32_t cls[TBX_SERVE_WAVES][SiPainter<1>::NCLS * 8];
    __shared__ uint32_t spr_lds[SPR_WORDS];
    si_fill_sprites(spr_lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    tbx_serve_loop(ctl, lane, [&](const ActionSource& src, uint32_t flags) { si_step_body<false>(d, d, d, c, src, flags, 0, lane, recs); },
                   [&](int channels, uint8_t* frame, int part, int split) {
                       // (with split > 1 every wave starts at unit `part`: 35 units over the block's waves)
                       switch (channels) {
                       case 1: si_serve_paint<1>(d, recs, frame, lane, lds[wave], cls[wave], spr_lds, part, split); break;
                       case 3: si_serve_paint<3>(d, recs, frame, lane, lds[wave], cls[wave], spr_lds, part, split); break;
                       default: si_serve_paint<4>(d, recs, frame, lane, lds[wave], cls[wave], spr_lds, part, split); break;
                       }
                       return true;
                   },
                   d.reward, d.done, d.lives_out, d.score_out, d.err_flag);
}

// ------------------------------------------------------------------ fused agent observation
//
// max(frame A, frame B) -> gray -> area warp -> frame stack without the two full-resolution gray frames ever reaching
// HBM: agent_fused_wave (agent_device.hpp) with two SiGrayPainters in one wave per env.
// (three waves per SIMD: the two painters sit at 168-170 VGPRs, one register either side of the step from three waves to two,
// and the kernel is issue-bound -- 2.98 ms per agent step with three, 4.07 with two)
template <int S>
__global__ __launch_bounds__(TBX_BLOCK) __attribute__((amdgpu_waves_per_eu(3))) void si_agent_warp_kernel(SiDev dLive, SiDev dA, SiDev dB, AgentWarpArgs a, int n)
{
    __shared__ AgentFusedLds<SiGrayPainter> lds[TBX_WAVES_PER_BLOCK];
    __shared__ uint32_t spr_lds[SPR_WORDS];
    si_fill_sprites(spr_lds);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int env = wave_uniform(a.first + blockIdx.x * TBX_WAVES_PER_BLOCK + wave);
    if (env >= a.end) return;
    SiGrayPainter pa, pb;
    pa.spr_lds = spr_lds; pb.spr_lds = spr_lds;
    agent_fused_wave<S, SiGrayPainter>(pa, pb, dLive, dA, dB, a, env, lane, lds[wave]);
}

// ------------------------------------------------------------------ state pack / unpack, scalars

__global__ void si_pack_kernel(SiDev d, int env, tbx_si_state_t* out)
{
    env += blockIdx.x;      // one block per env of the requested range
    out += blockIdx.x;
    const int lane = threadIdx.x & 63;
    SiRegs s;
    si_load(d, env, lane, s);
    const int32_t* f = s.f;
    if (lane == 0) {
        out->rand[0] = s.rng.s0; out->rand[1] = s.rng.s1;
        out->score = f[F_SCORE]; out->lives = f[F_LIVES]; out->level = f[F_LEVEL];
        out->life_display_timer = f[F_LIFE_TIMER]; out->enemy_shot_delay = f[F_SHOT_DELAY];
        out->n_enemies = f[F_N_ENEMIES]; out->n_enemy_lasers = f[F_N_LASERS]; out->has_ship_laser = f[F_HAS_SHIP_LASER];
        out->ship_x = f[F_SHIP_X]; out->ship_y = f[F_SHIP_Y]; out->ship_w = f[F_SHIP_W]; out->ship_h = f[F_SHIP_H];
        out->ship_speed = f[F_SHIP_SPEED]; out->ship_death_counter = f[F_SHIP_DC];
        out->ship_color = unpack_color((uint32_t)f[F_SHIP_COLOR]);
        out->ship_alive = f[F_SHIP_FLAGS] & 1; out->ship_death_hit_1 = (f[F_SHIP_FLAGS] >> 1) & 1;
        out->_pad0[0] = out->_pad0[1] = 0;
        out->ufo_x = f[F_UFO_X]; out->ufo_y = f[F_UFO_Y]; out->ufo_appearance_counter = f[F_UFO_APP]; out->ufo_death_counter = f[F_UFO_DC];
        out->move_counter = f[F_MOVE_COUNTER]; out->move_dir = f[F_MOVE_DIR];
        out->visual_orientation = f[F_ORIENT] ? 1 : 0;
        out->_pad1[0] = out->_pad1[1] = out->_pad1[2] = 0;
        out->n_shields = f[F_N_SHIELDS];
        for (int k = 0; k < TBX_SI_MAX_SHIELDS; k++) {
            out->shield_x[k] = f[F_SHIELD_X0 + k]; out->shield_y[k] = f[F_SHIELD_Y0 + k];
            out->shield_color[k] = unpack_color((uint32_t)f[F_SHIELD_C0 + k]);
        }
    }
    if (lane < TBX_SI_MAX_SHIELDS * TBX_SI_SHIELD_H) out->shield_rows[lane / TBX_SI_SHIELD_H][lane % TBX_SI_SHIELD_H] = (uint16_t)s.srow;
    {
        tbx_si_enemy_t e;
        memset(&e, 0, sizeof e);
        if (lane < f[F_N_ENEMIES]) {
            e.x = s.ex; e.y = s.ey; e.row = s.erow; e.col = s.ecol; e.id = s.eid; e.points = s.epoints;
            e.death_counter = e_dc(s); e.alive = e_alive(s) ? 1 : 0;
        }
        out->enemies[lane] = e;
    }
    if (lane <= SHIP_SLOT) {
        tbx_si_laser_t l;
        memset(&l, 0, sizeof l);
        const bool on = lane == SHIP_SLOT ? f[F_HAS_SHIP_LASER] != 0 : lane < f[F_N_LASERS];
        if (on) {
            l.x = s.lf[LF_X]; l.y = s.lf[LF_Y]; l.w = s.lf[LF_W]; l.h = s.lf[LF_H]; l.t = s.lf[LF_T];
            l.movement = s.lf[LF_MOV]; l.speed = s.lf[LF_SPEED]; l.color = unpack_color((uint32_t)s.lf[LF_COLOR]);
        }
        if (lane == SHIP_SLOT) out->ship_laser = l;
        else out->enemy_lasers[lane] = l;
    }
}

__global__ void si_unpack_kernel(SiDev d, int env, const tbx_si_state_t* in)
{
    env += blockIdx.x;
    in += blockIdx.x;
    const int lane = threadIdx.x & 63;
    SiRegs s;
    int32_t* f = s.f;
    s.rng.s0 = in->rand[0]; s.rng.s1 = in->rand[1];
    f[F_SCORE] = in->score; f[F_LIVES] = in->lives; f[F_LEVEL] = in->level;
    f[F_LIFE_TIMER] = in->life_display_timer; f[F_SHOT_DELAY] = in->enemy_shot_delay;
    f[F_N_ENEMIES] = in->n_enemies; f[F_N_LASERS] = in->n_enemy_lasers; f[F_HAS_SHIP_LASER] = in->has_ship_laser ? 1 : 0;
    f[F_SHIP_X] = in->ship_x; f[F_SHIP_Y] = in->ship_y; f[F_SHIP_W] = in->ship_w; f[F_SHIP_H] = in->ship_h;
    f[F_SHIP_SPEED] = in->ship_speed; f[F_SHIP_DC] = in->ship_death_counter < 0 ? -1 : in->ship_death_counter;
    f[F_SHIP_COLOR] = (int32_t)pack_color(in->ship_color);
    f[F_SHIP_FLAGS] = (in->ship_alive ? 1 : 0) | (in->ship_death_hit_1 ? 2 : 0);
    f[F_UFO_X] = in->ufo_x; f[F_UFO_Y] = in->ufo_y; f[F_UFO_APP] = in->ufo_appearance_counter;
    f[F_UFO_DC] = in->ufo_death_counter < 0 ? -1 : in->ufo_death_counter;
    f[F_MOVE_COUNTER] = in->move_counter; f[F_MOVE_DIR] = in->move_dir & 3; f[F_ORIENT] = in->visual_orientation ? 1 : 0;
    f[F_N_SHIELDS] = in->n_shields;
    for (int k = 0; k < TBX_SI_MAX_SHIELDS; k++) {
        const bool on = k < in->n_shields;
        f[F_SHIELD_X0 + k] = on ? in->shield_x[k] : 0; f[F_SHIELD_Y0 + k] = on ? in->shield_y[k] : 0;
        f[F_SHIELD_C0 + k] = on ? (int32_t)pack_color(in->shield_color[k]) : 0;
    }
    {
        const int k = lane / TBX_SI_SHIELD_H, r = lane % TBX_SI_SHIELD_H;
        s.srow = (lane < TBX_SI_MAX_SHIELDS * TBX_SI_SHIELD_H && k < in->n_shields) ? in->shield_rows[k][r] : 0u;
    }
    {
        const tbx_si_enemy_t& e = in->enemies[lane];
        const bool on = lane < in->n_enemies;
        s.ex = on ? e.x : 0; s.ey = on ? e.y : 0; s.erow = on ? e.row : 0; s.ecol = on ? e.col : 0;
        s.eid = on ? e.id : 0; s.epoints = on ? e.points : 0;
        s.estatus = on ? mk_status(e.alive != 0, e.death_counter < 0 ? -1 : e.death_counter) : 0;
    }
    for (int i = 0; i < NLF; i++) s.lf[i] = 0;
    {
        const int slot = lane & 15;
        const bool on = slot == SHIP_SLOT ? in->has_ship_laser != 0 : slot < in->n_enemy_lasers;
        if (slot <= SHIP_SLOT && on) {
            const tbx_si_laser_t& l = slot == SHIP_SLOT ? in->ship_laser : in->enemy_lasers[slot];
            s.lf[LF_X] = l.x; s.lf[LF_Y] = l.y; s.lf[LF_W] = l.w; s.lf[LF_H] = l.h; s.lf[LF_T] = l.t;
            s.lf[LF_MOV] = l.movement & 3; s.lf[LF_SPEED] = l.speed; s.lf[LF_COLOR] = (int32_t)pack_color(l.color);
        }
    }
    si_store(d, env, lane, s);
}

// ------------------------------------------------------------------ batched interventions (tbx_edit / tbx_reduce)
// SpaceInvadersIntervention (toybox/interventions/space_invaders.py:159-176) over the batch, one thread per env on the [field][N] scalars
__global__ __launch_bounds__(256) void si_edit_kernel(SiDev d, int op, TbxEditArgs a, const uint8_t* __restrict__ mask)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.n || (mask && !mask[env])) return;
    const int f = op == TBX_EDIT_SET_LIVES ? F_LIVES : op == TBX_EDIT_SET_SCORE ? F_SCORE : op == TBX_EDIT_SET_LEVEL ? F_LEVEL : F_UFO_APP;
    d.sc[(size_t)env * HEAD_WORDS + f] = a.geti(env, 0);
}

__global__ __launch_bounds__(256) void si_reduce_kernel(SiDev d, int query, TbxEditArgs a, double* __restrict__ out, int width)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.n) return;
    double* o = out + (size_t)env * width;
    auto F = [&](int f) { return d.sc[(size_t)env * HEAD_WORDS + f]; };
    if (query == TBX_QUERY_SI_SHIP) {
        o[0] = F(F_SHIP_X); o[1] = F(F_SHIP_Y); o[2] = F(F_SHIP_W); o[3] = F(F_SHIP_H); o[4] = F(F_SHIP_SPEED);
        o[5] = F(F_SHIP_FLAGS) & 1; o[6] = F(F_SHIP_DC); o[7] = (F(F_SHIP_FLAGS) >> 1) & 1;
    }
}

__global__ void si_scalars_kernel(SiDev d, int32_t* score, int32_t* lives, int32_t* level)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    if (score) score[i] = d.sc[(size_t)i * HEAD_WORDS + F_SCORE];
    if (lives) lives[i] = d.sc[(size_t)i * HEAD_WORDS + F_LIVES];
    if (level) level[i] = d.sc[(size_t)i * HEAD_WORDS + F_LEVEL];
}

// ------------------------------------------------------------------ host ops

struct SiOps : GameOps {
    SiDev d{};
    SiCfg c{};
    tbx_si_config_t cfg{};
    // rasteriser input records (SiRenderRec): two buffers, `recs` the current one.  Written by the batch step kernel; anything
    // else that touches state clears recs_valid and the next render rebuilds them (si_rec_prep_kernel).  custom: an
    // intervention wrote enemies off the formation grid -- records cannot describe that, the state-reading rasteriser paints.
    SiRenderRec* recs = nullptr;
    SiRenderRec* recs_other = nullptr;
    int recs_par = 0;
    bool recs_valid = false;
    bool custom = false;
    bool plain = true;          // every state's ids, points and laser constants are what si_load_canonical derives (is_plain)
    bool want_recs = false;     // a batch render has been asked for since creation: steps leave records from now on (a loop that
                                // never renders keeps the 12 us the record costs the step kernel: 61 against 50 us at 65 536 envs)
    SiRenderRec* recs_chunk[2] = {nullptr, nullptr};   // [k][N] records of a rollout chunk of parity q (tbx_rollout_synthetic), made on first use
    int recs_chunk_k[2] = {0, 0};

    int height() const override { return TBX_SI_H; }
    int width() const override { return TBX_SI_W; }
    size_t state_size() const override { return sizeof(tbx_si_state_t); }
    size_t config_size() const override { return sizeof(tbx_si_config_t); }

    int load_cfg(tbx_engine* e, const tbx_si_config_t& k)
    {
        if (k.n_rows < 1 || k.n_rows > TBX_SI_MAX_ROWS) return e->fail(TBX_E_UNSUPPORTED, "space_invaders: n_rows must be 1..10");
        if (k.n_shields < 0 || k.n_shields > TBX_SI_MAX_SHIELDS) return e->fail(TBX_E_UNSUPPORTED, "space_invaders: at most 3 shields");
        if (k.enemy_protocol != 0) return e->fail(TBX_E_UNSUPPORTED, "space_invaders: only the TargetPlayer firing protocol is implemented");
        cfg = k;
        c.jitter = k.jitter; c.start_lives = k.start_lives; c.n_rows = k.n_rows; c.n_shields = k.n_shields;
        for (int i = 0; i < TBX_SI_MAX_ROWS; i++) c.row_scores[i] = k.row_scores[i];
        for (int i = 0; i < TBX_SI_MAX_SHIELDS; i++) { c.shield_x[i] = k.shield_x[i]; c.shield_y[i] = k.shield_y[i]; }
        return TBX_OK;
    }

    int init(tbx_engine* e, const void* cfg_pod, size_t cfg_size) override
    {
        if (!cfg_pod || cfg_size != sizeof(tbx_si_config_t)) return e->fail(TBX_E_INVALID, "space_invaders: config size mismatch");
        tbx_si_config_t k;
        memcpy(&k, cfg_pod, sizeof k);
        int rc = load_cfg(e, k);
        if (rc) return rc;
        const size_t N = (size_t)e->n;
        d.n = e->n;
        d.sim_rng = e->sim_rng; d.prev_score = e->prev_score; d.reward = e->reward; d.done = e->done;
        d.lives_out = e->lives_out; d.score_out = e->score_out; d.packed = e->packed; d.err_flag = e->err_flag;
        TBX_HIP(hipMalloc((void**)&d.sc, N * HEAD_WORDS * sizeof(int32_t)));
        TBX_HIP(hipMalloc((void**)&d.enemies, N * NEF * 64 * sizeof(int32_t)));
        TBX_HIP(hipMalloc((void**)&d.shields, N * 64 * sizeof(uint32_t)));
        TBX_HIP(hipMalloc((void**)&d.lasers, N * NLF * 16 * sizeof(int32_t)));
        TBX_HIP(hipMalloc((void**)&recs, N * sizeof(SiRenderRec)));
        TBX_HIP(hipMalloc((void**)&recs_other, N * sizeof(SiRenderRec)));
        return TBX_OK;
    }

    void destroy(tbx_engine*) override
    {
        hipFree(recs); hipFree(recs_other); hipFree(recs_chunk[0]); hipFree(recs_chunk[1]);
        hipFree(d.sc); hipFree(d.enemies); hipFree(d.shields); hipFree(d.lasers);
        hipFree(dA.sc); hipFree(dA.enemies); hipFree(dA.shields); hipFree(dA.lasers);
        hipFree(dB.sc); hipFree(dB.enemies); hipFree(dB.shields); hipFree(dB.lasers);
    }

    int get_config(tbx_engine*, void* pod) override { memcpy(pod, &cfg, sizeof cfg); return TBX_OK; }
    int set_config(tbx_engine* e, const void* pod) override
    {
        tbx_si_config_t k;
        memcpy(&k, pod, sizeof k);
        // enemies keep the points they were made with until their next game: a config with other row scores (or rows) means the
        // running games no longer carry "the config's points of their row" -- the step kernel loads every row from here on
        if (k.n_rows != cfg.n_rows || memcmp(k.row_scores, cfg.row_scores, sizeof k.row_scores) != 0) plain = false;
        return load_cfg(e, k);
    }

    static dim3 grid_for(int count) { return dim3((count + TBX_WAVES_PER_BLOCK - 1) / TBX_WAVES_PER_BLOCK); }

    int new_game(tbx_engine* e, const uint8_t* mask_dev, hipStream_t s) override
    {
        hipLaunchKernelGGL(si_new_game_kernel, grid_for(e->n), dim3(TBX_BLOCK), 0, s, d, c, mask_dev);
        TBX_HIP(hipGetLastError());
        recs_valid = false;
        return TBX_OK;
    }

    int step(tbx_engine* e, const ActionSource& src, uint32_t flags, hipStream_t s) override
    {
        int first = 0, count = e->n;
        if (src.single_env >= 0) { first = src.single_env; count = 1; }
        if (src.acc_reward || src.buf_valid || src.exec_flag || src.frames > 1) {    // an agent step's frames (never auto-reset)
            if (flags & TBX_STEP_AUTO_RESET) return e->fail(TBX_E_INVALID, "an agent step cannot auto-reset");
            hipLaunchKernelGGL(si_agent_step_kernel, grid_for(count), dim3(TBX_BLOCK), 0, s, d, dA, dB, c, src, flags, first, count);
            recs_valid = false;
        } else {
            // a whole-batch step leaves the rasteriser's records behind (canonical formations only)
            const bool whole = src.single_env < 0 && !custom && want_recs;
            const bool canon = !custom && plain;               // the short load (si_load_canonical)
            SiRenderRec* const wr = whole ? recs : nullptr;
            if (src.single_env < 0) {
                if (canon) TBX_LAUNCH_STEP(e, s, si_step_kernel<true>, grid_for(count), dim3(TBX_BLOCK), d, c, src, flags, first, count, wr);
                else TBX_LAUNCH_STEP(e, s, si_step_kernel<false>, grid_for(count), dim3(TBX_BLOCK), d, c, src, flags, first, count, wr);
            } else if (canon) hipLaunchKernelGGL(si_step_kernel<true>, grid_for(count), dim3(TBX_BLOCK), 0, s, d, c, src, flags, first, count, wr);
            else hipLaunchKernelGGL(si_step_kernel<false>, grid_for(count), dim3(TBX_BLOCK), 0, s, d, c, src, flags, first, count, wr);
            recs_valid = whole;
        }
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    // the rasteriser reads nothing but the records, and there are two buffers of them: a batch step can run while the previous
    // frame is still being painted (engine.hip, pipelined mode)
    bool pipeline_ok() const override { return !custom && recs_other != nullptr; }
    // scripts/pipeline_sweep.py, stream order against value 3, ms per step without a gather: 1 024 envs 0.0546 / 0.0514, 2 048
    // 0.0926 / 0.0799, 4 096 0.162 / 0.150, 8 192 0.306 / 0.296, 12 288 0.452 / 0.447
    int pipeline_auto(int n, bool gather) const override { return (!gather && n < 16384) ? 3 : 0; }
    int records_parity() const override { return recs_par; }
    bool records_valid() const override { return recs_valid; }
    void rebind_outputs(tbx_engine* e) override
    {
        d.reward = e->reward; d.done = e->done; d.lives_out = e->lives_out; d.score_out = e->score_out; d.packed = e->packed;
    }
    int step_ahead(tbx_engine* e, const ActionSource& src, uint32_t flags, hipStream_t s) override
    {
        if (plain) hipLaunchKernelGGL(si_step_kernel<true>, grid_for(e->n), dim3(TBX_BLOCK), 0, s, d, c, src, flags, 0, e->n, recs_other);
        else hipLaunchKernelGGL(si_step_kernel<false>, grid_for(e->n), dim3(TBX_BLOCK), 0, s, d, c, src, flags, 0, e->n, recs_other);
        TBX_HIP(hipGetLastError());
        std::swap(recs, recs_other);
        recs_par ^= 1;
        recs_valid = true;
        return TBX_OK;
    }

    // tbx_render_step_synthetic stays the two launches in stream order (GameOps::render_step_fused false).  A fused launch -- one step
    // block (four envs, a wave each) in front of the twelve rasteriser blocks of the same four envs, other records buffer -- was built
    // and measured in round 4 (scripts/strong_sweep.py, ms per step two launches / fused on one box): 4 096 envs 0.1625 / 0.1645,
    // 8 192 envs 0.3103 / 0.3147, 65 536 envs 2.299 / 2.421.  The rasteriser is not waiting for anything the step's waves could hide
    // in: inside the launch they cost what they cost in front of it, plus the block slots (23 KB of LDS each) they hold.  Removed.
    // Breakout's form -- ALL the step blocks first, then the rasteriser's -- was built too (bit-identical): 0.1561 / 0.1610 at 4 096
    // envs (the overlapped launches of pipeline_auto get 0.150), 0.2999 / 0.3070 at 8 192, 2.311 / 2.309 at 65 536 (fused / two
    // launches; staggered first waves inside it: no better).  Nothing at the size that matters, less than mode 3 below it: removed.
    bool serve_paints() const override { return true; }
    int serve(tbx_engine* e, TbxServeCtl* ctl_dev, hipStream_t s) override
    {
        hipLaunchKernelGGL(si_serve_kernel, dim3(1), dim3(64 * TBX_SERVE_WAVES), 0, s, d, c, custom ? nullptr : recs, ctl_dev);
        TBX_HIP(hipGetLastError());
        recs_valid = false;
        return TBX_OK;
    }

    // ---- agent layer: MaxAndSkipEnv's two-frame buffer is two snapshots of the dynamic SoA state per env
    SiDev dA{}, dB{};
    bool agent_fused() const override { return true; }
    bool multi_frame_step() const override { return true; }
    bool agent_reset_supported() const override { return true; }

    int alloc_slot(tbx_engine* e, SiDev& x)
    {
        if (x.sc) return TBX_OK;
        const size_t N = (size_t)e->n;
        x = d;
        x.sc = nullptr; x.enemies = nullptr; x.shields = nullptr; x.lasers = nullptr;
        TBX_HIP(hipMalloc((void**)&x.sc, N * HEAD_WORDS * sizeof(int32_t)));
        TBX_HIP(hipMalloc((void**)&x.enemies, N * NEF * 64 * sizeof(int32_t)));
        TBX_HIP(hipMalloc((void**)&x.shields, N * 64 * sizeof(uint32_t)));
        TBX_HIP(hipMalloc((void**)&x.lasers, N * NLF * 16 * sizeof(int32_t)));
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
        const dim3 grid = grid_for(a.end - a.first), block(TBX_BLOCK);
        switch (a.obs ? a.stack : 0) {
        case 0: hipLaunchKernelGGL(si_agent_warp_kernel<0>, grid, block, 0, s, d, dA, dB, a, e->n); break;      // the plane ring (new_plane = 2), any depth
        case 1: hipLaunchKernelGGL(si_agent_warp_kernel<1>, grid, block, 0, s, d, dA, dB, a, e->n); break;
        case 2: hipLaunchKernelGGL(si_agent_warp_kernel<2>, grid, block, 0, s, d, dA, dB, a, e->n); break;
        case 3: hipLaunchKernelGGL(si_agent_warp_kernel<3>, grid, block, 0, s, d, dA, dB, a, e->n); break;
        default: hipLaunchKernelGGL(si_agent_warp_kernel<4>, grid, block, 0, s, d, dA, dB, a, e->n); break;
        }
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int agent_reset_envs(tbx_engine* e, const AgentResetArgs& r, hipStream_t s) override
    {
        const dim3 grid = r.list ? dim3(std::min<unsigned>(grid_for(e->n).x, 512u)) : grid_for(e->n);
        hipLaunchKernelGGL(si_agent_reset_kernel, grid, dim3(TBX_BLOCK), 0, s, d, dA, dB, c, r);
        TBX_HIP(hipGetLastError());
        recs_valid = false;
        return TBX_OK;
    }

    int render_from(tbx_engine* e, int source, const uint8_t* pick_live, uint8_t* out_dev, int channels, hipStream_t s) override
    {
        const SiDev& src = source == 1 ? dA : source == 2 ? dB : d;
        return render_impl(e, src, d, source ? pick_live : nullptr, out_dev, channels, 0, e->n, s);
    }

    // ---- rollout chunks (engine.hip, rollout_chunked): the step lane runs the record of the current state and then the k single-frame
    // step launches of the chunk back to back, step j leaving the record of frame j + 1 in the chunk's buffer; the chunk's rasteriser
    // launches (engine.hip: per frame on the two lanes, or one over the chunk) read them.  (No multi-frame kernel: the wave-per-env step keeps its state in HBM rows anyway.)
    bool rollout_ok(int) const override { return pipeline_ok(); }
    // scripts/rollout_ab.py (RA_GAME=space_invaders), k = 4, ms per step, single calls in stream order / the pipelined two-launch loop
    // (the engine's choice, off under a record ring) / chunks; no gather | K = 4 ring (r06_rollout_ab_si.txt): 2 048 envs 0.0918 / 0.0793 /
    // 0.0736 | 0.0972 / 0.0972 / 0.0809; 4 096: 0.1614 / 0.1481 / 0.1439 | 0.1658 / 0.1658 / 0.1487; 8 192: 0.3080 / 0.2966 / 0.2950 |
    // 0.3124 / 0.3119 / 0.2950; 16 384: 0.5943 / 0.5987 / 0.5950 | 0.6010 / 0.6026 / 0.6301
    bool rollout_auto(int n, int /*gather_kind*/) const override { return n <= 8192; }
    int rollout_step(tbx_engine* e, const ActionSource& src, uint32_t flags, int k, int q, uint64_t* packed, size_t stride, hipStream_t s) override
    {
        const size_t N = (size_t)e->n;
        if (recs_chunk_k[q] < k) {                             // (the caller has made sure nothing reads the old buffer any more)
            TBX_HIP(hipStreamSynchronize(s));
            hipFree(recs_chunk[q]);
            recs_chunk[q] = nullptr;
            recs_chunk_k[q] = 0;
            TBX_HIP(hipMalloc((void**)&recs_chunk[q], sizeof(SiRenderRec) * (size_t)k * N));
            recs_chunk_k[q] = k;
        }
        hipLaunchKernelGGL(si_rec_prep_kernel, grid_for(e->n), dim3(TBX_BLOCK), 0, s, d, recs_chunk[q], 0, e->n);
        for (int j = 0; j < k; j++) {
            SiDev dj = d;
            dj.packed = packed + (size_t)j * stride;
            ActionSource sj = src;
            sj.t = src.t + (uint64_t)j;
            SiRenderRec* const wr = j + 1 < k ? recs_chunk[q] + (size_t)(j + 1) * N : nullptr;
            if (plain) hipLaunchKernelGGL(si_step_kernel<true>, grid_for(e->n), dim3(TBX_BLOCK), 0, s, dj, c, sj, flags, 0, e->n, wr);
            else hipLaunchKernelGGL(si_step_kernel<false>, grid_for(e->n), dim3(TBX_BLOCK), 0, s, dj, c, sj, flags, 0, e->n, wr);
        }
        TBX_HIP(hipGetLastError());
        recs_valid = false;                                    // the single-frame records no longer show the state
        return TBX_OK;
    }
    int rollout_render(tbx_engine* e, uint8_t* out, int channels, int q, int j, hipStream_t s) override
    {
        return launch_rec_render(e, recs_chunk[q] + (size_t)j * (size_t)e->n, out, channels, 0, e->n, s);
    }

    bool rollout_span_ok() const override { return true; }
    bool rollout_span_auto(int /*n*/, int /*gather_kind*/) const override { return false; }
    int rollout_render_span(tbx_engine* e, uint8_t* out, int channels, int q, int j0, int count, bool /*behind_rasteriser*/, hipStream_t s) override
    {
        return launch_rec_render(e, recs_chunk[q] + (size_t)j0 * (size_t)e->n, out, channels, 0, count * e->n, s);
    }

    int launch_rec_render(tbx_engine* e, const SiRenderRec* rr, uint8_t* out_dev, int channels, int first_env, int n_envs, hipStream_t s)
    {
        // waves per frame: the set-up is light (a record, ~300 instructions), so RGB frames are cut finer than the
        // state-reading rasteriser could afford: 35 six-row units over twelve waves (measured 3 / 5 / 7 / 9 / 12 / 18 / 35 waves per frame:
        // 2.45 / 2.50 / 2.49 / 2.49 / 2.43 / 2.49 / 3.67 ms at 65 536 envs, 0.174 / 0.171 / 0.166 / 0.164 / 0.151 / 0.167 / 0.238 ms at 4 096)
        const int split_opt = e->opt[TBX_OPT_RENDER_SPLIT];
        const int split = split_opt > 0 ? split_opt : channels == 3 ? 12 : (channels == 4 && n_envs <= 32768) ? 5 : (channels == 1 && n_envs <= 4096) ? 4 : 1;   // (gray, small batches: 0.035 against 0.067 ms at 1 024 envs, 0.094 against 0.102 at 4 096)
        switch (channels) {
        case 1: hipLaunchKernelGGL(si_rec_render_kernel<1>, grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, rr, out_dev, first_env, n_envs, split); break;
        case 3: hipLaunchKernelGGL(si_rec_render_kernel<3>, grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, rr, out_dev, first_env, n_envs, split); break;
        case 4: hipLaunchKernelGGL(si_rec_render_kernel<4>, grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, rr, out_dev, first_env, n_envs, split); break;
        default: return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
        }
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int render(tbx_engine* e, uint8_t* out_dev, int channels, int first_env, int n_envs, hipStream_t s) override
    {
        if (custom) return render_impl(e, d, d, nullptr, out_dev, channels, first_env, n_envs, s);
        if (n_envs == e->n) want_recs = true;
        if (!recs_valid) {
            hipLaunchKernelGGL(si_rec_prep_kernel, grid_for(n_envs), dim3(TBX_BLOCK), 0, s, d, recs, first_env, n_envs);
            TBX_HIP(hipGetLastError());
            if (first_env == 0 && n_envs == e->n) recs_valid = true;
        }
        return launch_rec_render(e, recs, out_dev, channels, first_env, n_envs, s);
    }

    int render_impl(tbx_engine* e, const SiDev& src, const SiDev& alt, const uint8_t* pick_alt, uint8_t* out_dev, int channels, int first_env,
                    int n_envs, hipStream_t s)
    {
#ifdef TBX_DIAG
        // measurement builds only (make DIAG=1; scripts/render_probe.py): bit 0 skip blank units; TBX_SI_DIAG bit 0 = no
        // painting (stores only), bit 1 = no stores (painting only) -- frames are WRONG with either set
        static const int skip_blank = (getenv("TBX_SI_NO_SKIP") ? 0 : 1) | (getenv("TBX_SI_DIAG") ? atoi(getenv("TBX_SI_DIAG")) << 1 : 0);
#else
        constexpr int skip_blank = 1;
#endif
        const int split_env = e->opt[TBX_OPT_RENDER_SPLIT];
        // the painter set-up (~1 500 instructions behind a state load) is too heavy to repeat many times per frame: five waves
        // per RGB frame (A/B on two boxes at 65 536 envs, scripts/ab_render.py: 2.41-2.43 ms in every round against 2.41-2.56
        // for one wave per frame, which drops into the GPU's slower rate state more often; whole step 25.0 -> 25.8 M
        // env-steps/s), five per RGBA frame only for batches that would leave the chip under-filled, one per gray frame.  A
        // two-stage form (set-up once per env into a 3.5 KB record, 9-18 light waves per frame) and a set-up shared by a
        // block's waves through LDS were built and measured this round: both slower (profiles/HISTORY.md).
        const int split = split_env > 0 ? split_env : channels == 3 ? 5 : (channels == 4 && n_envs <= 32768) ? 5 : 1;
        switch (channels) {
        case 1: if (pick_alt) hipLaunchKernelGGL((si_render_kernel<1, true>), grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, skip_blank, split, alt, pick_alt); else hipLaunchKernelGGL((si_render_kernel<1, false>), grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, skip_blank, split, alt, pick_alt); break;
        case 3: if (pick_alt) hipLaunchKernelGGL((si_render_kernel<3, true>), grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, skip_blank, split, alt, pick_alt); else hipLaunchKernelGGL((si_render_kernel<3, false>), grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, skip_blank, split, alt, pick_alt); break;
        case 4: if (pick_alt) hipLaunchKernelGGL((si_render_kernel<4, true>), grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, skip_blank, split, alt, pick_alt); else hipLaunchKernelGGL((si_render_kernel<4, false>), grid_for(n_envs * split), dim3(TBX_BLOCK), 0, s, src, out_dev, first_env, n_envs, skip_blank, split, alt, pick_alt); break;
        default: return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
        }
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    // the formation grid the render records assume: enemy i at (x0 + 32 (i % 6), y0 + 18 (i / 6)), row i / 6, col i % 6
    static bool is_canonical(const tbx_si_state_t& st)
    {
        for (int i = 0; i < st.n_enemies; i++) {
            const tbx_si_enemy_t& en = st.enemies[i];
            const int row = i / TBX_SI_COLS, col = i % TBX_SI_COLS;
            if (en.row != row || en.col != col) return false;
            if ((long)en.x != (long)st.enemies[0].x + TBX_SI_ENEMY_DX * col || (long)en.y != (long)st.enemies[0].y + TBX_SI_ENEMY_DY * row) return false;
        }
        return true;
    }
    // ... and what the step kernel's short load (si_load_canonical) derives on top of that: ids, the config's points, and lasers
    // that have the size, speed, direction and colour of their kind
    bool is_plain(const tbx_si_state_t& st) const
    {
        for (int i = 0; i < st.n_enemies; i++) {
            const int row = i / TBX_SI_COLS;
            if (st.enemies[i].id != i || row >= TBX_SI_MAX_ROWS || st.enemies[i].points != cfg.row_scores[row]) return false;
        }

        auto laser_ok = [](const tbx_si_laser_t& l, int mov, int speed, uint32_t color) {
            return l.w == TBX_SI_LASER_W && l.h == TBX_SI_LASER_H && (l.movement & 3) == mov && l.speed == speed && pack_color(l.color) == color;
        };
        for (int i = 0; i < st.n_enemy_lasers; i++)
            if (!laser_ok(st.enemy_lasers[i], TBX_DIR_DOWN, TBX_SI_ENEMY_LASER_V, rgb_u32(TBX_SI_COL_ENEMY_LASER))) return false;
        if (st.has_ship_laser && !laser_ok(st.ship_laser, TBX_DIR_UP, TBX_SI_SHIP_LASER_V, rgb_u32(TBX_SI_COL_SHIP_LASER))) return false;
        return true;
    }

    int pack_state(tbx_engine* e, int env, int count, hipStream_t s) override
    {
        hipLaunchKernelGGL(si_pack_kernel, dim3(count), dim3(64), 0, s, d, env, (tbx_si_state_t*)e->staging);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int unpack_state(tbx_engine* e, int env, int count, const void* pod_host, hipStream_t s) override
    {
        const auto* sts = (const tbx_si_state_t*)pod_host;
        for (int i = 0; i < count; i++) {
            const auto& st = sts[i];
            if (st.n_enemies < 0 || st.n_enemies > TBX_SI_MAX_ENEMIES) return e->fail(TBX_E_UNSUPPORTED, "space_invaders: the device engine holds at most 64 enemies per env");
            if (st.n_enemy_lasers < 0 || st.n_enemy_lasers > TBX_SI_MAX_LASERS) return e->fail(TBX_E_UNSUPPORTED, "space_invaders: the device engine holds at most 8 enemy lasers per env");
            if (st.n_shields < 0 || st.n_shields > TBX_SI_MAX_SHIELDS) return e->fail(TBX_E_UNSUPPORTED, "space_invaders: the device engine holds at most 3 shields per env");
            for (int k = 0; k < st.n_enemies; k++) {
                const auto& en = st.enemies[k];
                if (en.row < 0 || en.row > 255 || en.col < 0 || en.col > 255 || en.id < 0 || en.id > 65535)
                    return e->fail(TBX_E_UNSUPPORTED, "space_invaders: the device engine keeps an enemy's row, col (0..255) and id (0..65535) in one word");
            }
        }
        if (!custom)
            for (int i = 0; i < count && !custom; i++)
                if (!is_canonical(sts[i])) custom = true;        // from now on the state-reading rasteriser paints, steps stay in stream order
        if (plain)
            for (int i = 0; i < count && plain; i++)
                if (!is_plain(sts[i])) plain = false;            // from now on the step kernel loads every row
        TBX_HIP(hipMemcpyAsync(e->staging, pod_host, sizeof(tbx_si_state_t) * (size_t)count, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(si_unpack_kernel, dim3(count), dim3(64), 0, s, d, env, (const tbx_si_state_t*)e->staging);
        TBX_HIP(hipGetLastError());
        recs_valid = false;
        return TBX_OK;
    }

    int edit(tbx_engine* e, int op, const TbxEditArgs& a, const uint8_t* mask_dev, hipStream_t s) override
    {
        if (op != TBX_EDIT_SET_LIVES && op != TBX_EDIT_SET_SCORE && op != TBX_EDIT_SET_LEVEL && op != TBX_EDIT_SI_UFO_APPEARANCE)
            return e->fail(TBX_E_INVALID, "space_invaders: unknown edit");
        hipLaunchKernelGGL(si_edit_kernel, dim3((e->n + 255) / 256), dim3(256), 0, s, d, op, a, mask_dev);
        TBX_HIP(hipGetLastError());
        recs_valid = false;
        return TBX_OK;
    }

    int reduce(tbx_engine* e, int query, const TbxEditArgs& a, double* out_dev, int width, hipStream_t s) override
    {
        hipLaunchKernelGGL(si_reduce_kernel, dim3((e->n + 255) / 256), dim3(256), 0, s, d, query, a, out_dev, width);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }

    int scalars(tbx_engine* e, int32_t* score_dev, int32_t* lives_dev, int32_t* level_dev, hipStream_t s) override
    {
        hipLaunchKernelGGL(si_scalars_kernel, dim3((e->n + 255) / 256), dim3(256), 0, s, d, score_dev, lives_dev, level_dev);
        TBX_HIP(hipGetLastError());
        return TBX_OK;
    }
};

}  // namespace

GameOps* tbx_make_si_ops() { return new SiOps(); }
