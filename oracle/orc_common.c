/* orc_common.c -- RNG, action tables and the batch driver of the CPU oracle.
 * TEST INFRASTRUCTURE ONLY (see oracle.h). */
#include "oracle.h"
#include <string.h>

/* ---------------------------------------------------------------- RNG
 * xoroshiro128+ with (a,b,c) = (55,14,36), output s0+s1.  Derived from, and pinned by, the
 * rand.state words of /root/reference/toybox/interventions/defaults/{amidar,breakout,
 * space_invaders}_{config,state}_default.json (tests/golden/rng_kat.json). */
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void orc_rng_seed(uint64_t s[2], uint32_t seed)
{
    s[0] = 0x193a6754a8a7d469ULL ^ (uint64_t)seed;
    s[1] = 0x97830e05113ba7bbULL;
}

uint64_t orc_rng_next(uint64_t s[2])
{
    uint64_t s0 = s[0], s1 = s[1];
    uint64_t r = s0 + s1;
    s1 ^= s0;
    s[0] = rotl64(s0, 55) ^ s1 ^ (s1 << 14);
    s[1] = rotl64(s1, 36);
    return r;
}

void orc_rng_child(uint64_t parent[2], uint64_t child[2])
{
    child[0] = orc_rng_next(parent);
    child[1] = orc_rng_next(parent);
}

/* Uniform integer in [0,n): 64x64->128 widening multiply with the power-of-two-scaled
 * rejection zone of rand's UniformInt::sample_single.  Pinned for n=4 by the Breakout golden:
 * the state RNG there is its initial child advanced by exactly two draws, the first of which
 * this rule rejects and the second of which maps to start position #2 (the golden ball). */
uint64_t orc_rng_range(uint64_t s[2], uint64_t n)
{
    if (n <= 1) return 0;
    uint64_t zone = (n << __builtin_clzll(n)) - 1;
    for (;;) {
        uint64_t v = orc_rng_next(s);
        unsigned __int128 m = (unsigned __int128)v * n;
        uint64_t lo = (uint64_t)m;
        if (lo <= zone) return (uint64_t)(m >> 64);
    }
}

uint64_t orc_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

/* ---------------------------------------------------------------- actions */

int orc_ale_action_to_buttons(int a)
{
    /* names: envs/atari/constants.py:16-35; a name containing UP/DOWN/LEFT/RIGHT presses that
       direction, containing FIRE presses button1 */
    static const uint8_t tab[18] = {
        0,                                            /* 0 NOOP */
        TBX_BTN_BUTTON1,                              /* 1 FIRE */
        TBX_BTN_UP, TBX_BTN_RIGHT, TBX_BTN_LEFT, TBX_BTN_DOWN,   /* 2..5 */
        TBX_BTN_UP | TBX_BTN_RIGHT, TBX_BTN_UP | TBX_BTN_LEFT,   /* 6 UPRIGHT 7 UPLEFT */
        TBX_BTN_DOWN | TBX_BTN_RIGHT, TBX_BTN_DOWN | TBX_BTN_LEFT, /* 8 9 */
        TBX_BTN_UP | TBX_BTN_BUTTON1, TBX_BTN_RIGHT | TBX_BTN_BUTTON1,   /* 10 11 */
        TBX_BTN_LEFT | TBX_BTN_BUTTON1, TBX_BTN_DOWN | TBX_BTN_BUTTON1,  /* 12 13 */
        TBX_BTN_UP | TBX_BTN_RIGHT | TBX_BTN_BUTTON1, TBX_BTN_UP | TBX_BTN_LEFT | TBX_BTN_BUTTON1,
        TBX_BTN_DOWN | TBX_BTN_RIGHT | TBX_BTN_BUTTON1, TBX_BTN_DOWN | TBX_BTN_LEFT | TBX_BTN_BUTTON1,
    };
    if (a < 0 || a > 17) return -1;
    return tab[a];
}

int orc_legal_actions(int game, int32_t* out, int cap)
{
    static const int32_t brk[] = {0, 1, 3, 4};
    static const int32_t ami[] = {0, 1, 2, 3, 4, 5};
    static const int32_t spi[] = {0, 1, 3, 4, 11, 12};
    static const int32_t grd[] = {0, 2, 3, 4, 5};
    const int32_t* src; int n;
    switch (game) {
    case TBX_GAME_BREAKOUT: src = brk; n = 4; break;
    case TBX_GAME_AMIDAR: src = ami; n = 6; break;
    case TBX_GAME_SPACE_INVADERS: src = spi; n = 6; break;
    case TBX_GAME_GRIDWORLD: src = grd; n = 5; break;
    default: return -1;
    }
    for (int i = 0; i < n && i < cap; i++) out[i] = src[i];
    return n;
}

int orc_frame_dims(int game, int* h, int* w)
{
    switch (game) {
    case TBX_GAME_BREAKOUT: *h = TBX_BRK_H; *w = TBX_BRK_W; return 0;
    case TBX_GAME_SPACE_INVADERS: *h = TBX_SI_H; *w = TBX_SI_W; return 0;
    case TBX_GAME_AMIDAR: *h = TBX_AMI_H; *w = TBX_AMI_W; return 0;
    case TBX_GAME_GRIDWORLD: *h = TBX_GW_H; *w = TBX_GW_W; return 0;
    default: return -1;
    }
}

int32_t orc_synthetic_action(int game, uint64_t seed, uint64_t env_global, uint64_t t)
{
    int32_t legal[18];
    int n = orc_legal_actions(game, legal, 18);
    uint64_t h = orc_splitmix64(seed ^ (env_global << 32) ^ t);
    return legal[h % (uint64_t)n];
}

/* ---------------------------------------------------------------- batch driver */

static size_t state_size(int game)
{
    switch (game) {
    case TBX_GAME_BREAKOUT: return sizeof(tbx_breakout_state_t);
    case TBX_GAME_SPACE_INVADERS: return sizeof(tbx_si_state_t);
    case TBX_GAME_AMIDAR: return sizeof(tbx_amidar_state_t);
    case TBX_GAME_GRIDWORLD: return sizeof(tbx_gridworld_state_t);
    default: return 0;
    }
}

static void one_new_game(int game, const void* cfg, void* st, uint64_t* sim)
{
    switch (game) {
    case TBX_GAME_BREAKOUT:
        orc_breakout_new_game((const tbx_breakout_config_t*)cfg, sim, (tbx_breakout_state_t*)st);
        break;
    case TBX_GAME_SPACE_INVADERS:
        orc_si_new_game((const tbx_si_config_t*)cfg, sim, (tbx_si_state_t*)st);
        break;
    case TBX_GAME_AMIDAR:
        orc_amidar_new_game((const tbx_amidar_config_t*)cfg, sim, (tbx_amidar_state_t*)st);
        break;
    case TBX_GAME_GRIDWORLD:
        orc_gridworld_new_game((const tbx_gridworld_config_t*)cfg, sim, (tbx_gridworld_state_t*)st);
        break;
    }
}

static void one_step(int game, const void* cfg, void* st, uint32_t buttons)
{
    switch (game) {
    case TBX_GAME_BREAKOUT:
        orc_breakout_step((const tbx_breakout_config_t*)cfg, (tbx_breakout_state_t*)st, buttons);
        break;
    case TBX_GAME_SPACE_INVADERS:
        orc_si_step((const tbx_si_config_t*)cfg, (tbx_si_state_t*)st, buttons);
        break;
    case TBX_GAME_AMIDAR:
        orc_amidar_step((const tbx_amidar_config_t*)cfg, (tbx_amidar_state_t*)st, buttons);
        break;
    case TBX_GAME_GRIDWORLD:
        orc_gridworld_step((const tbx_gridworld_config_t*)cfg, (tbx_gridworld_state_t*)st, buttons);
        break;
    }
}

static void one_scalars(int game, const void* st, int32_t* score, int32_t* lives, int32_t* level)
{
    *score = 0; *lives = 0; *level = 0;
    switch (game) {
    case TBX_GAME_BREAKOUT: {
        const tbx_breakout_state_t* s = (const tbx_breakout_state_t*)st;
        *score = s->score; *lives = s->lives; *level = s->level;
        break; }
    case TBX_GAME_SPACE_INVADERS: {
        const tbx_si_state_t* s = (const tbx_si_state_t*)st;
        *score = s->score; *lives = s->lives; *level = s->level;
        break; }
    case TBX_GAME_AMIDAR: {
        const tbx_amidar_state_t* s = (const tbx_amidar_state_t*)st;
        *score = s->score; *lives = s->lives; *level = s->level;
        break; }
    case TBX_GAME_GRIDWORLD: {                  /* one life, spent on reaching a goal cell; a single level */
        const tbx_gridworld_state_t* s = (const tbx_gridworld_state_t*)st;
        *score = s->score; *lives = s->game_over ? 0 : 1; *level = 1;
        break; }
    }
}

int orc_get_scalars(int game, const void* states, int n, int32_t* score, int32_t* lives, int32_t* level)
{
    size_t sz = state_size(game);
    if (!sz) return -1;
    for (int i = 0; i < n; i++) {
        int32_t a, b, c;
        one_scalars(game, (const char*)states + sz * (size_t)i, &a, &b, &c);
        if (score) score[i] = a;
        if (lives) lives[i] = b;
        if (level) level[i] = c;
    }
    return 0;
}

int orc_new_game_batch(int game, const void* cfg, void* states, uint64_t* sim_rng, int32_t* prev_score,
                       int n, const uint8_t* mask)
{
    size_t sz = state_size(game);
    if (!sz) return -1;
    for (int i = 0; i < n; i++) {
        if (mask && !mask[i]) continue;
        void* st = (char*)states + sz * (size_t)i;
        one_new_game(game, cfg, st, sim_rng + 2 * (size_t)i);
        int32_t sc, lv, le;
        one_scalars(game, st, &sc, &lv, &le);
        if (prev_score) prev_score[i] = sc;
    }
    return 0;
}

int orc_step_batch(int game, const void* cfg, void* states, uint64_t* sim_rng, int32_t* prev_score,
                   int n, const int32_t* actions, uint32_t flags,
                   int32_t* reward, uint8_t* done, int32_t* lives, int32_t* score, int threads)
{
    size_t sz = state_size(game);
    if (!sz) return -1;
    int bad = 0;
#pragma omp parallel for schedule(static) num_threads(threads > 1 ? threads : 1) reduction(| : bad)
    for (int i = 0; i < n; i++) {
        void* st = (char*)states + sz * (size_t)i;
        int b = orc_ale_action_to_buttons(actions[i]);
        if (b < 0) { b = 0; bad |= 1; }
        one_step(game, cfg, st, (uint32_t)b);
        int32_t sc, lv, le;
        one_scalars(game, st, &sc, &lv, &le);
        int32_t r = sc - prev_score[i];
        if (r < 0) r = 0;
        prev_score[i] = sc;
        uint8_t d = lv <= 0;
        if (reward) reward[i] = r;
        if (done) done[i] = d;
        if (lives) lives[i] = lv;
        if (score) score[i] = sc;
        if (d && (flags & TBX_STEP_AUTO_RESET)) {
            one_new_game(game, cfg, st, sim_rng + 2 * (size_t)i);
            one_scalars(game, st, &sc, &lv, &le);
            prev_score[i] = sc;
        }
    }
    return bad ? TBX_E_ACTION : 0;
}

int orc_render_batch(int game, const void* cfg, const void* states, int n, uint8_t* out, int channels, int threads)
{
    size_t sz = state_size(game);
    int h, w;
    if (!sz || orc_frame_dims(game, &h, &w)) return -1;
    size_t fsz = (size_t)h * w * channels;
#pragma omp parallel for schedule(static) num_threads(threads > 1 ? threads : 1)
    for (int i = 0; i < n; i++) {
        const void* st = (const char*)states + sz * (size_t)i;
        switch (game) {
        case TBX_GAME_BREAKOUT:
            orc_breakout_render((const tbx_breakout_config_t*)cfg, (const tbx_breakout_state_t*)st,
                                out + fsz * (size_t)i, channels);
            break;
        case TBX_GAME_SPACE_INVADERS:
            orc_si_render((const tbx_si_config_t*)cfg, (const tbx_si_state_t*)st, out + fsz * (size_t)i, channels);
            break;
        case TBX_GAME_AMIDAR:
            orc_amidar_render((const tbx_amidar_config_t*)cfg, (const tbx_amidar_state_t*)st, out + fsz * (size_t)i, channels);
            break;
        case TBX_GAME_GRIDWORLD:
            orc_gridworld_render((const tbx_gridworld_config_t*)cfg, (const tbx_gridworld_state_t*)st, out + fsz * (size_t)i, channels);
            break;
        }
    }
    return 0;
}
