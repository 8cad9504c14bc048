/* orc_abi.c -- the C-ABI of include/toybox_amd.h restated over the CPU oracle.
 * TEST INFRASTRUCTURE ONLY (see oracle.h): it lets tests/ drive the product's Python host
 * (toybox_amd.Engine, Toybox, envs) over the oracle on a machine without a GPU, and lets the
 * GPU parity tests run the very same call sequence against both libraries.  "Device" pointers
 * are host pointers here and streams are ignored.  The product never loads this library. */
#define _GNU_SOURCE
#include "oracle.h"
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct tbx_engine {
    int game, n;
    size_t ssz, csz;
    void* cfg;
    char* states;
    uint64_t* sim;        /* [n][2] */
    int32_t* prev;
    int32_t *reward, *lives, *score;
    uint8_t* done;
    uint64_t* packed;
    uint8_t* frame;
    size_t frame_bytes;
    int pending_action_error;
    int threads;
    char err[256];
    /* agent-side preprocessing */
    int agent_on;
    tbx_agent_config_t acfg;
    uint8_t *gray_a, *gray_b, *aobs, *afin, *adone;
    uint8_t* aplane;      /* [n][out_h][out_w] the newest plane of every stack (acfg.new_plane = 1), else NULL */
    uint8_t* aring;       /* [stack][n][out_h][out_w] the last `stack` planes INSTEAD of the stack (acfg.new_plane = 2: aobs == NULL) */
    int ahead;            /* the ring slot that holds the newest plane */
    float* areward;
    /* host delivery: the oracle's "_begin" calls do everything at once; "_end" hands out the error the step left */
    int host_pending, agent_host_pending, host_rc;
    /* record gather: the oracle restates the one-rank case only (there is no second process to talk to) */
    int gather_on, gather_width, gather_ranks, gather_rank;
    uint64_t* gathered;
    void* gshm;
    size_t gshm_len;
    char gshm_name[TBX_GATHER_ID_BYTES];
    uint64_t gather_calls[2];
    /* TBX_OPT_GATHER_EVERY = K > 1: batch steps write their records into slot `gather_fill` of ring[ring_par] ([K][width]),
     * tbx_gather only counts and the K-th call exchanges the whole ring */
    int gather_every, gather_fill, ring_par, gather_advance;
    uint64_t* ring[2];
    uint64_t* packed_own;
    /* Monitor / EpisodicLifeEnv state and this step's episode records */
    int32_t *ep_ret, *ep_len, *ep_index, *prev_lives, *ep_len_out;
    uint8_t* ep_done;
    float* ep_ret_out;
    uint8_t *was_real_done, *needs_reset;
    int32_t* noop_override;
    int pending_needs_reset;
    int opt[TBX_OPT_COUNT];
    uint8_t* one_frame;     /* tbx_step1_frame's buffer */
    /* tbx_rollout_synthetic: the chunk's frames [k][n][H][W][C] and, without a record ring, its step records [k][n] */
    uint8_t* chunk_frames;
    size_t chunk_frame_bytes;
    uint64_t *chunk_packed, *chunk_packed_base;
    size_t chunk_packed_records, chunk_stride;
    int chunk_k, chunk_channels;
};

static char g_err[256];

static int fail(tbx_engine* e, int code, const char* msg)
{
    snprintf(e ? e->err : g_err, 256, "%s", msg);
    return code;
}

int tbx_abi_version(void) { return TBX_ABI_VERSION; }
const char* tbx_last_error(const tbx_engine* e) { return e ? e->err : g_err; }
int tbx_frame_dims(int game, int* h, int* w) { return orc_frame_dims(game, h, w) ? TBX_E_INVALID : TBX_OK; }
int tbx_legal_actions(int game, int32_t* out, int cap) { int n = orc_legal_actions(game, out, cap); return n < 0 ? TBX_E_INVALID : n; }
int tbx_ale_action_to_buttons(int a) { int b = orc_ale_action_to_buttons(a); return b < 0 ? TBX_E_INVALID : b; }

size_t tbx_state_size(int game)
{
    switch (game) {
    case TBX_GAME_BREAKOUT: return sizeof(tbx_breakout_state_t);
    case TBX_GAME_SPACE_INVADERS: return sizeof(tbx_si_state_t);
    case TBX_GAME_AMIDAR: return sizeof(tbx_amidar_state_t);
    case TBX_GAME_GRIDWORLD: return sizeof(tbx_gridworld_state_t);
    default: return 0;
    }
}
size_t tbx_config_size(int game)
{
    switch (game) {
    case TBX_GAME_BREAKOUT: return sizeof(tbx_breakout_config_t);
    case TBX_GAME_SPACE_INVADERS: return sizeof(tbx_si_config_t);
    case TBX_GAME_AMIDAR: return sizeof(tbx_amidar_config_t);
    case TBX_GAME_GRIDWORLD: return sizeof(tbx_gridworld_config_t);
    default: return 0;
    }
}

static void agent_free(tbx_engine* e);
static void gather_close(tbx_engine* e);

int tbx_destroy(tbx_engine* e)
{
    if (!e) return TBX_OK;
    free(e->cfg); free(e->states); free(e->sim); free(e->prev); free(e->reward); free(e->lives);
    gather_close(e);                       /* (puts e->packed back onto the engine's own array) */
    free(e->score); free(e->done); free(e->packed); free(e->frame); free(e->one_frame); free(e->chunk_frames); free(e->chunk_packed);
    agent_free(e);
    free(e);
    return TBX_OK;
}

/* the checker has no device: ordinal -1 and an arch string that says so */
int tbx_device_identity(tbx_engine* e, tbx_device_identity_t* out)
{
    if (!e) return TBX_E_INVALID;
    if (!out) return fail(e, TBX_E_INVALID, "out is NULL");
    memset(out, 0, sizeof *out);
    out->ordinal = -1; out->pci_domain = out->pci_bus = out->pci_device = -1;
    snprintf(out->arch, sizeof out->arch, "cpu-oracle");
    snprintf(out->name, sizeof out->name, "scalar C restatement (test infrastructure)");
    return TBX_OK;
}

int tbx_create(int game, int n, int device, const void* cfg, size_t cfg_size, tbx_engine** out)
{
    (void)device;
    if (!out) return TBX_E_INVALID;
    *out = NULL;
    if (n < 1) return fail(NULL, TBX_E_INVALID, "n_envs must be >= 1");
    size_t ssz = tbx_state_size(game), csz = tbx_config_size(game);
    if (!ssz) return fail(NULL, TBX_E_INVALID, "unknown game id");
    if (cfg && cfg_size != csz) return fail(NULL, TBX_E_INVALID, "config size mismatch");
    tbx_engine* e = (tbx_engine*)calloc(1, sizeof *e);
    e->game = game; e->n = n; e->ssz = ssz; e->csz = csz;
    e->cfg = calloc(1, csz);
    e->states = (char*)calloc((size_t)n, ssz);
    e->sim = (uint64_t*)calloc((size_t)n * 2, 8);
    e->prev = (int32_t*)calloc((size_t)n, 4);
    e->reward = (int32_t*)calloc((size_t)n, 4);
    e->lives = (int32_t*)calloc((size_t)n, 4);
    e->score = (int32_t*)calloc((size_t)n, 4);
    e->done = (uint8_t*)calloc((size_t)n, 1);
    e->packed = (uint64_t*)calloc((size_t)n, 8);
    const char* th = getenv("TBX_ORACLE_THREADS");
    e->threads = th ? atoi(th) : 1;
    e->opt[TBX_OPT_RESIDENT_STEP] = 1;
    e->opt[TBX_OPT_GATHER_EVERY] = 1;
    e->gather_every = 1;
    if (cfg) memcpy(e->cfg, cfg, csz);
    else switch (game) {
        case TBX_GAME_BREAKOUT: orc_breakout_default_config((tbx_breakout_config_t*)e->cfg); break;
        case TBX_GAME_SPACE_INVADERS: orc_si_default_config((tbx_si_config_t*)e->cfg); break;
        case TBX_GAME_AMIDAR: orc_amidar_default_config((tbx_amidar_config_t*)e->cfg); break;
        case TBX_GAME_GRIDWORLD: orc_gridworld_default_config((tbx_gridworld_config_t*)e->cfg); break;
    }
    for (int i = 0; i < n; i++) memcpy(e->sim + 2 * (size_t)i, e->cfg, 16);
    orc_new_game_batch(game, e->cfg, e->states, e->sim, e->prev, n, NULL);
    *out = e;
    return TBX_OK;
}

int tbx_num_envs(const tbx_engine* e) { return e ? e->n : TBX_E_INVALID; }
int tbx_game(const tbx_engine* e) { return e ? e->game : TBX_E_INVALID; }

int tbx_seed(tbx_engine* e, int env, uint32_t seed)
{
    if (!e) return TBX_E_INVALID;
    if (env < -1 || env >= e->n) return fail(e, TBX_E_INVALID, "env index out of range");
    for (int i = 0; i < e->n; i++) {
        if (env >= 0 && i != env) continue;
        orc_rng_seed(e->sim + 2 * (size_t)i, env >= 0 ? seed : seed + (uint32_t)i);
    }
    return TBX_OK;
}

int tbx_seed_array(tbx_engine* e, const uint32_t* seeds)
{
    if (!e) return TBX_E_INVALID;
    if (!seeds) return fail(e, TBX_E_INVALID, "seeds pointer is NULL");
    for (int i = 0; i < e->n; i++) orc_rng_seed(e->sim + 2 * (size_t)i, seeds[i]);
    return TBX_OK;
}

int tbx_get_sim_rng(tbx_engine* e, int env, uint64_t out[2])
{
    if (!e) return TBX_E_INVALID;
    if (env < 0 || env >= e->n || !out) return fail(e, TBX_E_INVALID, "env index out of range");
    out[0] = e->sim[2 * (size_t)env]; out[1] = e->sim[2 * (size_t)env + 1];
    return TBX_OK;
}

int tbx_set_sim_rng(tbx_engine* e, int env, const uint64_t st[2])
{
    if (!e) return TBX_E_INVALID;
    if (env < -1 || env >= e->n || !st) return fail(e, TBX_E_INVALID, "env index out of range");
    for (int i = 0; i < e->n; i++) {
        if (env >= 0 && i != env) continue;
        e->sim[2 * (size_t)i] = st[0]; e->sim[2 * (size_t)i + 1] = st[1];
    }
    return TBX_OK;
}

int tbx_new_game(tbx_engine* e, const uint8_t* mask)
{
    if (!e) return TBX_E_INVALID;
    orc_new_game_batch(e->game, e->cfg, e->states, e->sim, e->prev, e->n, mask);
    return TBX_OK;
}

static void pack_outputs(tbx_engine* e)
{
    if (e->gather_advance) {               /* ring mode: the first step after a tbx_gather opens the next slot */
        e->gather_advance = 0;
        e->packed = e->ring[e->ring_par] + (size_t)e->gather_fill * (size_t)e->gather_width;
    }
    for (int i = 0; i < e->n; i++) {
        int32_t lv = e->lives[i];
        uint32_t l8 = lv < 0 ? 0u : lv > 255 ? 255u : (uint32_t)lv;
        e->packed[i] = (uint64_t)(uint32_t)e->reward[i] | ((uint64_t)(e->done[i] ? 1u : 0u) << 32) | ((uint64_t)l8 << 40);
    }
}

int tbx_step_device(tbx_engine* e, const int32_t* actions, uint32_t flags, void* stream)
{
    (void)stream;
    if (!e) return TBX_E_INVALID;
    if (!actions) return fail(e, TBX_E_INVALID, "actions pointer is NULL");
    int rc = orc_step_batch(e->game, e->cfg, e->states, e->sim, e->prev, e->n, actions, flags,
                            e->reward, e->done, e->lives, e->score, e->threads);
    pack_outputs(e);
    if (rc == TBX_E_ACTION) e->pending_action_error = 1;
    return TBX_OK;
}

int tbx_step_synthetic(tbx_engine* e, uint64_t seed, uint64_t t, uint64_t env_offset, uint32_t flags, void* stream)
{
    if (!e) return TBX_E_INVALID;
    int32_t* a = (int32_t*)malloc((size_t)e->n * 4);
    for (int i = 0; i < e->n; i++) a[i] = orc_synthetic_action(e->game, seed, env_offset + (uint64_t)i, t);
    int rc = tbx_step_device(e, a, flags, stream);
    free(a);
    return rc;
}

static int take_action_error(tbx_engine* e)
{
    if (e->pending_needs_reset) {
        e->pending_needs_reset = 0;
        e->pending_action_error = 0;
        return fail(e, TBX_E_NEEDS_RESET, "an env was stepped after its game ended inside EpisodicLifeEnv's no-op step (bench.Monitor raises here)");
    }
    if (e->pending_action_error) {
        e->pending_action_error = 0;
        return fail(e, TBX_E_ACTION, "an illegal ALE action id was passed (treated as NOOP)");
    }
    return TBX_OK;
}

int tbx_step(tbx_engine* e, const int32_t* actions, uint32_t flags, int32_t* reward, uint8_t* done,
             int32_t* lives, int32_t* score)
{
    int rc = tbx_step_device(e, actions, flags, NULL);
    if (rc) return rc;
    size_t n = (size_t)e->n;
    if (reward) memcpy(reward, e->reward, n * 4);
    if (done) memcpy(done, e->done, n);
    if (lives) memcpy(lives, e->lives, n * 4);
    if (score) memcpy(score, e->score, n * 4);
    return take_action_error(e);
}

int tbx_apply_input(tbx_engine* e, int env, uint32_t buttons)
{
    if (!e) return TBX_E_INVALID;
    if (env < 0 || env >= e->n) return fail(e, TBX_E_INVALID, "env index out of range");
    void* st = e->states + e->ssz * (size_t)env;
    switch (e->game) {
    case TBX_GAME_BREAKOUT:
        orc_breakout_step((const tbx_breakout_config_t*)e->cfg, (tbx_breakout_state_t*)st, buttons & 0x3Fu);
        break;
    case TBX_GAME_SPACE_INVADERS:
        orc_si_step((const tbx_si_config_t*)e->cfg, (tbx_si_state_t*)st, buttons & 0x3Fu);
        break;
    case TBX_GAME_AMIDAR:
        orc_amidar_step((const tbx_amidar_config_t*)e->cfg, (tbx_amidar_state_t*)st, buttons & 0x3Fu);
        break;
    case TBX_GAME_GRIDWORLD:
        orc_gridworld_step((const tbx_gridworld_config_t*)e->cfg, (tbx_gridworld_state_t*)st, buttons & 0x3Fu);
        break;
    }
    int32_t sc, lv, le;
    orc_get_scalars(e->game, st, 1, &sc, &lv, &le);
    int32_t r = sc - e->prev[env];
    if (r < 0) r = 0;
    e->prev[env] = sc;
    e->reward[env] = r; e->done[env] = lv <= 0; e->lives[env] = lv; e->score[env] = sc;
    pack_outputs(e);
    return TBX_OK;
}

int tbx_step1(tbx_engine* e, int env, int32_t ale_action, uint32_t flags, int32_t out[4])
{
    if (!e) return TBX_E_INVALID;
    if (env < 0 || env >= e->n) return fail(e, TBX_E_INVALID, "env index out of range");
    int b = orc_ale_action_to_buttons(ale_action);
    const int bad = b < 0;
    int rc = tbx_apply_input(e, env, bad ? 0u : (uint32_t)b);
    if (rc) return rc;
    if (out) { out[0] = e->reward[env]; out[1] = e->done[env]; out[2] = e->lives[env]; out[3] = e->score[env]; }
    if ((flags & TBX_STEP_AUTO_RESET) && e->done[env]) {
        uint8_t* mask = (uint8_t*)calloc((size_t)e->n, 1);
        mask[env] = 1;
        orc_new_game_batch(e->game, e->cfg, e->states, e->sim, e->prev, e->n, mask);
        free(mask);
    }
    return bad ? fail(e, TBX_E_ACTION, "an illegal ALE action id was passed (treated as NOOP)") : TBX_OK;
}

/* the step of tbx_step1 followed by the picture of the state it left, in a buffer the engine owns (ToyboxBaseEnv.step) */
int tbx_step1_frame(tbx_engine* e, int env, int32_t ale_action, uint32_t flags, int channels, int32_t out[4], const uint8_t** frame_host)
{
    if (!e) return TBX_E_INVALID;
    if (frame_host) *frame_host = NULL;
    if (channels != 1 && channels != 3 && channels != 4) return fail(e, TBX_E_INVALID, "channels must be 1, 3 or 4");
    int rc = tbx_step1(e, env, ale_action, flags, out);
    if (rc != TBX_OK && rc != TBX_E_ACTION) return rc;
    int h = 0, w = 0;
    orc_frame_dims(e->game, &h, &w);
    if (!e->one_frame) e->one_frame = (uint8_t*)malloc((size_t)h * w * 4);
    int rc2 = tbx_render_env(e, env, e->one_frame, channels);
    if (rc2) return rc2;
    if (frame_host) *frame_host = e->one_frame;
    return rc;
}

int tbx_get_scalars(tbx_engine* e, int32_t* score, int32_t* lives, int32_t* level, uint8_t* over)
{
    if (!e) return TBX_E_INVALID;
    int32_t* lv = (int32_t*)malloc((size_t)e->n * 4);
    orc_get_scalars(e->game, e->states, e->n, score, lv, level);
    for (int i = 0; i < e->n; i++) {
        if (lives) lives[i] = lv[i];
        if (over) over[i] = lv[i] <= 0;
    }
    free(lv);
    return TBX_OK;
}

int tbx_render_device(tbx_engine* e, uint8_t* out, int channels, void* stream)
{
    (void)stream;
    if (!e) return TBX_E_INVALID;
    if (channels != 1 && channels != 3 && channels != 4) return fail(e, TBX_E_INVALID, "channels must be 1, 3 or 4");
    int h, w;
    orc_frame_dims(e->game, &h, &w);
    if (!out) {
        size_t bytes = (size_t)e->n * h * w * channels;
        if (e->frame_bytes < bytes) { free(e->frame); e->frame = (uint8_t*)malloc(bytes); e->frame_bytes = bytes; }
        out = e->frame;
    }
    orc_render_batch(e->game, e->cfg, e->states, e->n, out, channels, e->threads);
    return TBX_OK;
}

/* the random-rollout loop body as one call: the picture of the current state, then one frame with device-rule actions */
int tbx_render_step_synthetic(tbx_engine* e, uint8_t* out, int channels, uint64_t seed, uint64_t t, uint64_t env_offset, uint32_t flags, void* stream)
{
    int rc = tbx_render_device(e, out, channels, stream);
    if (rc) return rc;
    return tbx_step_synthetic(e, seed, t, env_offset, flags, stream);
}

/* A rollout chunk = the k single calls, one after the other (include/toybox_amd.h): frame j into row j of the chunk buffer; the
 * step records into the ring (K-step ring: tbx_gather after every step, the k-th one exchanges it) or copied out row by row. */
int tbx_rollout_synthetic(tbx_engine* e, int channels, uint64_t seed, uint64_t t0, int k, uint64_t env_offset, uint32_t flags, void* stream)
{
    if (!e) return TBX_E_INVALID;
    if (channels != 1 && channels != 3 && channels != 4) return fail(e, TBX_E_INVALID, "channels must be 1, 3 or 4");
    if (k < 1 || k > 64) return fail(e, TBX_E_INVALID, "tbx_rollout_synthetic: k must be in 1 .. 64");
    const int ring = e->gather_on && e->gather_every > 1;
    if (ring && e->gather_every != k)
        return fail(e, TBX_E_INVALID, "tbx_rollout_synthetic: the chunk must be as long as the gather's record ring (TBX_OPT_GATHER_EVERY)");
    if (ring && e->gather_fill != 0)
        return fail(e, TBX_E_INVALID, "tbx_rollout_synthetic: the record ring is partly filled (finish it with single steps + tbx_gather)");
    int h = 0, w = 0;
    tbx_frame_dims(e->game, &h, &w);
    const size_t n = (size_t)e->n, fb = n * (size_t)h * (size_t)w * (size_t)channels;
    if (e->chunk_frame_bytes < fb * (size_t)k) {
        free(e->chunk_frames);
        e->chunk_frames = (uint8_t*)malloc(fb * (size_t)k);
        e->chunk_frame_bytes = fb * (size_t)k;
    }
    if (!ring && e->chunk_packed_records < n * (size_t)k) {
        free(e->chunk_packed);
        e->chunk_packed = (uint64_t*)calloc(n * (size_t)k, 8);
        e->chunk_packed_records = n * (size_t)k;
    }
    uint64_t* ring_base = NULL;
    for (int j = 0; j < k; j++) {
        int rc = tbx_render_step_synthetic(e, e->chunk_frames + fb * (size_t)j, channels, seed, t0 + (uint64_t)j, env_offset, flags, stream);
        if (rc) return rc;
        if (ring && j == 0) ring_base = e->packed;
        if (!ring) memcpy(e->chunk_packed + n * (size_t)j, e->packed, n * 8);
        if (e->gather_on) {
            rc = tbx_gather(e, NULL, stream);
            if (rc) return rc;
        }
    }
    e->chunk_k = k; e->chunk_channels = channels;
    e->chunk_packed_base = ring ? ring_base : e->chunk_packed;
    e->chunk_stride = ring ? (size_t)e->gather_width : n;
    /* TBX_BUF_FRAME: the chunk's last frame, as after the k-th single call into a caller's buffer it is whatever it was -- the HIP
     * library names the last frame of the chunk; so does this one */
    free(e->frame);
    e->frame = (uint8_t*)malloc(fb);
    memcpy(e->frame, e->chunk_frames + fb * (size_t)(k - 1), fb);
    e->frame_bytes = fb;
    return TBX_OK;
}

int tbx_render(tbx_engine* e, uint8_t* out, int channels)
{
    if (!e) return TBX_E_INVALID;
    if (!out) return fail(e, TBX_E_INVALID, "output pointer is NULL");
    return tbx_render_device(e, out, channels, NULL);
}

int tbx_render_env(tbx_engine* e, int env, uint8_t* out, int channels)
{
    if (!e) return TBX_E_INVALID;
    if (env < 0 || env >= e->n || !out) return fail(e, TBX_E_INVALID, "env index out of range");
    if (channels != 1 && channels != 3 && channels != 4) return fail(e, TBX_E_INVALID, "channels must be 1, 3 or 4");
    orc_render_batch(e->game, e->cfg, e->states + e->ssz * (size_t)env, 1, out, channels, 1);
    return TBX_OK;
}

int tbx_get_state(tbx_engine* e, int env, void* pod, size_t size)
{
    if (!e) return TBX_E_INVALID;
    if (env < 0 || env >= e->n || !pod) return fail(e, TBX_E_INVALID, "env index out of range");
    if (size != e->ssz) return fail(e, TBX_E_INVALID, "state record size mismatch");
    memcpy(pod, e->states + e->ssz * (size_t)env, size);
    return TBX_OK;
}

/* tbx_set_state stores the CANONICAL form of a record, the one tbx_get_state returns on both libraries: slots beyond
 * the counts zeroed, flags 0 / 1, "no counter" = -1, directions two bits wide, tile tags two bits, paddings zero.  States a
 * game produces itself are canonical already; this only matters for hand-written records (interventions). */
static void normalize_state(int game, void* pod)
{
    if (game == TBX_GAME_BREAKOUT) {
        tbx_breakout_state_t* s = (tbx_breakout_state_t*)pod;
        s->is_dead = s->is_dead ? 1 : 0; s->reset = s->reset ? 1 : 0;
        s->_pad0[0] = s->_pad0[1] = 0;
        for (int b = s->n_balls; b < TBX_BRK_MAX_BALLS; b++) s->ball_x[b] = s->ball_y[b] = s->ball_vx[b] = s->ball_vy[b] = 0.0;
        for (int j = 0; j < TBX_BRK_MAX_BRICKS; j++) {
            if (j >= s->n_bricks) { memset(&s->bricks[j], 0, sizeof s->bricks[j]); continue; }
            s->bricks[j].alive = s->bricks[j].alive ? 1 : 0;
            s->bricks[j].destructible = s->bricks[j].destructible ? 1 : 0;
            s->bricks[j]._pad[0] = s->bricks[j]._pad[1] = 0;
        }
    } else if (game == TBX_GAME_SPACE_INVADERS) {
        tbx_si_state_t* s = (tbx_si_state_t*)pod;
        s->has_ship_laser = s->has_ship_laser ? 1 : 0;
        if (s->ship_death_counter < 0) s->ship_death_counter = -1;
        s->ship_alive = s->ship_alive ? 1 : 0; s->ship_death_hit_1 = s->ship_death_hit_1 ? 1 : 0;
        s->_pad0[0] = s->_pad0[1] = 0;
        if (s->ufo_death_counter < 0) s->ufo_death_counter = -1;
        s->move_dir &= 3;
        s->visual_orientation = s->visual_orientation ? 1 : 0;
        s->_pad1[0] = s->_pad1[1] = s->_pad1[2] = 0;
        for (int k = s->n_shields; k < TBX_SI_MAX_SHIELDS; k++) {
            s->shield_x[k] = s->shield_y[k] = 0;
            memset(&s->shield_color[k], 0, sizeof s->shield_color[k]);
            memset(s->shield_rows[k], 0, sizeof s->shield_rows[k]);
        }
        for (int j = 0; j < TBX_SI_MAX_ENEMIES; j++) {
            tbx_si_enemy_t* e = &s->enemies[j];
            if (j >= s->n_enemies) { memset(e, 0, sizeof *e); continue; }
            e->alive = e->alive ? 1 : 0;
            if (e->death_counter < 0) e->death_counter = -1;
            e->_pad[0] = e->_pad[1] = e->_pad[2] = 0;
        }
        for (int j = 0; j < TBX_SI_MAX_LASERS; j++) {
            if (j >= s->n_enemy_lasers) memset(&s->enemy_lasers[j], 0, sizeof s->enemy_lasers[j]);
            else s->enemy_lasers[j].movement &= 3;
        }
        if (!s->has_ship_laser) memset(&s->ship_laser, 0, sizeof s->ship_laser);
        else s->ship_laser.movement &= 3;
    } else if (game == TBX_GAME_AMIDAR) {
        tbx_amidar_state_t* s = (tbx_amidar_state_t*)pod;
        for (int j = s->n_enemies; j < TBX_AMI_MAX_ENEMIES; j++) memset(&s->enemies[j], 0, sizeof s->enemies[j]);
        for (int j = 0; j < TBX_AMI_MAX_BOXES; j++) {
            tbx_amidar_box_t* b = &s->boxes[j];
            if (j >= s->n_boxes) { memset(b, 0, sizeof *b); continue; }
            b->tl_tx &= 255; b->tl_ty &= 255; b->br_tx &= 255; b->br_ty &= 255;
            b->painted = b->painted ? 1 : 0; b->triggers_chase = b->triggers_chase ? 1 : 0;
            b->_pad[0] = b->_pad[1] = 0;
        }
        for (int y = 0; y < TBX_AMI_BOARD_H; y++)
            for (int x = 0; x < TBX_AMI_BOARD_W; x++) s->tiles[y][x] &= 3;
    } else if (game == TBX_GAME_GRIDWORLD) {
        tbx_gridworld_state_t* s = (tbx_gridworld_state_t*)pod;
        s->game_over = s->game_over ? 1 : 0;
        for (int t = 0; t < TBX_GW_MAX_TILES; t++) {
            s->tiles[t].goal = s->tiles[t].goal ? 1 : 0; s->tiles[t].walkable = s->tiles[t].walkable ? 1 : 0;
            s->tiles[t]._pad[0] = s->tiles[t]._pad[1] = 0;
        }
    }
}

int tbx_set_state(tbx_engine* e, int env, const void* pod, size_t size)
{
    if (!e) return TBX_E_INVALID;
    if (env < 0 || env >= e->n || !pod) return fail(e, TBX_E_INVALID, "env index out of range");
    if (size != e->ssz) return fail(e, TBX_E_INVALID, "state record size mismatch");
    if (e->game == TBX_GAME_BREAKOUT) {
        const tbx_breakout_state_t* s = (const tbx_breakout_state_t*)pod;
        if (s->n_balls < 0 || s->n_balls > TBX_BRK_MAX_BALLS) return fail(e, TBX_E_UNSUPPORTED, "breakout: the device engine holds at most 4 balls per env");
        if (s->n_bricks < 0 || s->n_bricks > TBX_BRK_MAX_BRICKS) return fail(e, TBX_E_UNSUPPORTED, "breakout: the device engine holds at most 256 bricks per env");
    }
    if (e->game == TBX_GAME_AMIDAR) {
        const tbx_amidar_state_t* s = (const tbx_amidar_state_t*)pod;
        if (s->n_enemies < 0 || s->n_enemies > TBX_AMI_MAX_ENEMIES) return fail(e, TBX_E_UNSUPPORTED, "amidar: the device engine holds at most 8 enemies per env");
        if (s->n_boxes < 0 || s->n_boxes > TBX_AMI_MAX_BOXES) return fail(e, TBX_E_UNSUPPORTED, "amidar: the device engine holds at most 64 boxes per env");
    }
    if (e->game == TBX_GAME_SPACE_INVADERS) {
        const tbx_si_state_t* s = (const tbx_si_state_t*)pod;
        if (s->n_enemies < 0 || s->n_enemies > TBX_SI_MAX_ENEMIES) return fail(e, TBX_E_UNSUPPORTED, "space_invaders: the device engine holds at most 64 enemies per env");
        if (s->n_enemy_lasers < 0 || s->n_enemy_lasers > TBX_SI_MAX_LASERS) return fail(e, TBX_E_UNSUPPORTED, "space_invaders: the device engine holds at most 8 enemy lasers per env");
        if (s->n_shields < 0 || s->n_shields > TBX_SI_MAX_SHIELDS) return fail(e, TBX_E_UNSUPPORTED, "space_invaders: the device engine holds at most 3 shields per env");
        for (int k = 0; k < s->n_enemies; k++) {
            const tbx_si_enemy_t* en = &s->enemies[k];
            if (en->row < 0 || en->row > 255 || en->col < 0 || en->col > 255 || en->id < 0 || en->id > 65535)
                return fail(e, TBX_E_UNSUPPORTED, "space_invaders: the device engine keeps an enemy's row, col (0..255) and id (0..65535) in one word");
        }
    }
    if (e->game == TBX_GAME_GRIDWORLD) {
        const tbx_gridworld_state_t* s = (const tbx_gridworld_state_t*)pod;
        if (s->width < 1 || s->width > TBX_GW_MAX_DIM || s->height < 1 || s->height > TBX_GW_MAX_DIM)
            return fail(e, TBX_E_UNSUPPORTED, "gridworld: game_size must be 1..32 x 1..32");
        if (s->n_tiles < 1 || s->n_tiles > TBX_GW_MAX_TILES) return fail(e, TBX_E_UNSUPPORTED, "gridworld: 1..16 tiles");
    }
    memcpy(e->states + e->ssz * (size_t)env, pod, size);
    normalize_state(e->game, e->states + e->ssz * (size_t)env);
    return TBX_OK;
}

int tbx_get_states(tbx_engine* e, int first, int count, void* pods, size_t size)
{
    if (!e) return TBX_E_INVALID;
    if (first < 0 || count < 1 || first + count > e->n || !pods) return fail(e, TBX_E_INVALID, "env range out of bounds");
    for (int i = 0; i < count; i++) {
        int rc = tbx_get_state(e, first + i, (char*)pods + size * (size_t)i, size);
        if (rc) return rc;
    }
    return TBX_OK;
}

int tbx_set_states(tbx_engine* e, int first, int count, const void* pods, size_t size)
{
    if (!e) return TBX_E_INVALID;
    if (first < 0 || count < 1 || first + count > e->n || !pods) return fail(e, TBX_E_INVALID, "env range out of bounds");
    for (int i = 0; i < count; i++) {
        int rc = tbx_set_state(e, first + i, (const char*)pods + size * (size_t)i, size);
        if (rc) return rc;
    }
    return TBX_OK;
}

int tbx_get_config(tbx_engine* e, void* pod, size_t size)
{
    if (!e) return TBX_E_INVALID;
    if (!pod || size != e->csz) return fail(e, TBX_E_INVALID, "config record size mismatch");
    memcpy(pod, e->cfg, size);
    memcpy(pod, e->sim, 16);
    return TBX_OK;
}

int tbx_set_config(tbx_engine* e, const void* pod, size_t size)
{
    if (!e) return TBX_E_INVALID;
    if (!pod || size != e->csz) return fail(e, TBX_E_INVALID, "config record size mismatch");
    if (e->game == TBX_GAME_BREAKOUT) {
        const tbx_breakout_config_t* k = (const tbx_breakout_config_t*)pod;
        if (k->n_rows < 1 || k->n_rows > TBX_BRK_MAX_ROWS) return fail(e, TBX_E_UNSUPPORTED, "breakout: n_rows must be 1..14");
        if (k->n_starts < 1 || k->n_starts > TBX_BRK_MAX_STARTS) return fail(e, TBX_E_UNSUPPORTED, "breakout: 1..8 ball_start_positions");
        if (k->paddle_discrete_segments < 1 || k->paddle_discrete_segments > TBX_BRK_MAX_SEGMENTS)
            return fail(e, TBX_E_UNSUPPORTED, "breakout: paddle_discrete_segments must be 1..16");
    }
    if (e->game == TBX_GAME_AMIDAR) {
        const tbx_amidar_config_t* k = (const tbx_amidar_config_t*)pod;
        if (k->n_enemies < 0 || k->n_enemies > TBX_AMI_MAX_ENEMIES) return fail(e, TBX_E_UNSUPPORTED, "amidar: at most 8 enemies");
    }
    if (e->game == TBX_GAME_SPACE_INVADERS) {
        const tbx_si_config_t* k = (const tbx_si_config_t*)pod;
        if (k->n_rows < 1 || k->n_rows > TBX_SI_MAX_ROWS) return fail(e, TBX_E_UNSUPPORTED, "space_invaders: n_rows must be 1..10");
        if (k->n_shields < 0 || k->n_shields > TBX_SI_MAX_SHIELDS) return fail(e, TBX_E_UNSUPPORTED, "space_invaders: at most 3 shields");
        if (k->enemy_protocol != 0) return fail(e, TBX_E_UNSUPPORTED, "space_invaders: only the TargetPlayer firing protocol is implemented");
    }
    if (e->game == TBX_GAME_GRIDWORLD) {
        const tbx_gridworld_config_t* k = (const tbx_gridworld_config_t*)pod;
        if (k->width < 1 || k->width > TBX_GW_MAX_DIM || k->height < 1 || k->height > TBX_GW_MAX_DIM)
            return fail(e, TBX_E_UNSUPPORTED, "gridworld: game_size must be 1..32 x 1..32");
        if (k->n_tiles < 1 || k->n_tiles > TBX_GW_MAX_TILES) return fail(e, TBX_E_UNSUPPORTED, "gridworld: 1..16 tiles");
    }
    memcpy(e->cfg, pod, size);
    /* `rand` unchanged from what tbx_get_config reports (env 0's words): the per-env simulator RNGs stay (toybox_amd.h) */
    if (memcmp(e->sim, pod, 16) != 0)
        for (int i = 0; i < e->n; i++) memcpy(e->sim + 2 * (size_t)i, pod, 16);
    return TBX_OK;
}

int tbx_query(tbx_engine* e, int env, int query_id, const int32_t* args, int n_args, int32_t* out, int n_out)
{
    if (!e) return TBX_E_INVALID;
    if (env < 0 || env >= e->n || !args || !out) return fail(e, TBX_E_INVALID, "bad query arguments");
    if (e->game == TBX_GAME_AMIDAR && n_args >= 2 && n_out >= 2 && orc_amidar_query(query_id, args, out) == 0) return TBX_OK;
    return fail(e, TBX_E_INVALID, "unknown query for this game");
}

static uint8_t* agent_newest_plane(tbx_engine* e);
static int agent_no_stack(tbx_engine* e);

int tbx_device_buffer(tbx_engine* e, int which, void** out_ptr, size_t* out_bytes)
{
    if (!e) return TBX_E_INVALID;
    if (!out_ptr) return fail(e, TBX_E_INVALID, "out_ptr is NULL");
    size_t n = (size_t)e->n, b = 0;
    void* p = NULL;
    switch (which) {
    case TBX_BUF_REWARD: p = e->reward; b = n * 4; break;
    case TBX_BUF_DONE: p = e->done; b = n; break;
    case TBX_BUF_LIVES: p = e->lives; b = n * 4; break;
    case TBX_BUF_SCORE: p = e->score; b = n * 4; break;
    case TBX_BUF_FRAME: p = e->frame; b = e->frame_bytes; break;
    case TBX_BUF_PACKED: p = e->packed; b = n * 8; break;
    case TBX_BUF_ROLLOUT_FRAMES:
        if (!e->chunk_k) return fail(e, TBX_E_INVALID, "tbx_rollout_synthetic has not been called");
        { int hh = 0, ww = 0; tbx_frame_dims(e->game, &hh, &ww); p = e->chunk_frames; b = (size_t)e->chunk_k * n * hh * ww * e->chunk_channels; }
        break;
    case TBX_BUF_ROLLOUT_PACKED:
        if (!e->chunk_k) return fail(e, TBX_E_INVALID, "tbx_rollout_synthetic has not been called");
        p = e->chunk_packed_base; b = 8 * (size_t)e->chunk_k * e->chunk_stride;
        break;
    case TBX_BUF_AGENT_OBS: case TBX_BUF_AGENT_REWARD: case TBX_BUF_AGENT_DONE:
    case TBX_BUF_AGENT_EP_DONE: case TBX_BUF_AGENT_EP_RETURN: case TBX_BUF_AGENT_EP_LENGTH: case TBX_BUF_AGENT_PLANE:
    case TBX_BUF_AGENT_RING:
        if (!e->agent_on) return fail(e, TBX_E_INVALID, "tbx_agent_init has not been called");
        if (which == TBX_BUF_AGENT_PLANE) {
            if (!agent_newest_plane(e)) return fail(e, TBX_E_INVALID, "TBX_BUF_AGENT_PLANE needs tbx_agent_config_t::new_plane = 1 or 2");
            p = agent_newest_plane(e); b = n * e->acfg.out_h * e->acfg.out_w;
        }
        else if (which == TBX_BUF_AGENT_RING) {
            if (!e->aring) return fail(e, TBX_E_INVALID, "TBX_BUF_AGENT_RING needs tbx_agent_config_t::new_plane = 2");
            p = e->aring; b = n * e->acfg.out_h * e->acfg.out_w * e->acfg.stack;
        }
        else if (which == TBX_BUF_AGENT_OBS) {
            if (!e->aobs) return agent_no_stack(e);
            p = e->aobs; b = n * e->acfg.out_h * e->acfg.out_w * e->acfg.stack;
        }
        else if (which == TBX_BUF_AGENT_REWARD) { p = e->areward; b = n * 4; }
        else if (which == TBX_BUF_AGENT_DONE) { p = e->adone; b = n; }
        else if (which == TBX_BUF_AGENT_EP_DONE) { p = e->ep_done; b = n; }
        else if (which == TBX_BUF_AGENT_EP_RETURN) { p = e->ep_ret_out; b = n * 4; }
        else { p = e->ep_len_out; b = n * 4; }
        break;
    case TBX_BUF_GATHERED:
        if (!e->gather_on) return fail(e, TBX_E_INVALID, "tbx_gather_init has not been called");
        p = e->gathered; b = (size_t)e->gather_ranks * e->gather_every * e->gather_width * 8;
        break;
    default: return fail(e, TBX_E_INVALID, "unknown buffer id");
    }
    *out_ptr = p;
    if (out_bytes) *out_bytes = b;
    return TBX_OK;
}

/* ---------------------------------------------------------------- batched interventions (tbx_edit / tbx_reduce)
 * The reference's helper methods (toybox/interventions/breakout.py:303-429, amidar.py:360-615, space_invaders.py:165-176),
 * restated env by env over the POD state records -- plain loops over `bricks`, `tiles`, `enemies`, the way the Python walks
 * the decoded JSON. */

int tbx_reduce_width(int game, int query)
{
    switch (query) {
    case TBX_QUERY_BRK_BRICKS_REMAINING: case TBX_QUERY_BRK_NUM_BRICKS: case TBX_QUERY_BRK_IS_CHANNEL: case TBX_QUERY_BRK_CHANNEL_COUNT:
    case TBX_QUERY_BRK_FIND_CHANNEL: return game == TBX_GAME_BREAKOUT ? 1 : TBX_E_INVALID;
    case TBX_QUERY_BRK_COLUMN: case TBX_QUERY_BRK_ROW: return game == TBX_GAME_BREAKOUT ? 32 : TBX_E_INVALID;
    case TBX_QUERY_BRK_PADDLE: return game == TBX_GAME_BREAKOUT ? 4 : TBX_E_INVALID;
    case TBX_QUERY_BRK_BALLS: return game == TBX_GAME_BREAKOUT ? 1 + 4 * TBX_BRK_MAX_BALLS : TBX_E_INVALID;
    case TBX_QUERY_AMI_MODE: return game == TBX_GAME_AMIDAR ? 2 : TBX_E_INVALID;
    case TBX_QUERY_AMI_ANY_CAUGHT: case TBX_QUERY_AMI_TILE: case TBX_QUERY_AMI_COUNT_TILES: case TBX_QUERY_AMI_PLAYER_ON_PAINTED:
    case TBX_QUERY_AMI_PLAYER_NEAR_UNPAINTED: return game == TBX_GAME_AMIDAR ? 1 : TBX_E_INVALID;
    case TBX_QUERY_AMI_ADJACENT: return game == TBX_GAME_AMIDAR ? 4 : TBX_E_INVALID;
    case TBX_QUERY_AMI_ENEMY_DISTANCES: case TBX_QUERY_AMI_PLAYER_ENEMY_DISTANCES: return game == TBX_GAME_AMIDAR ? TBX_AMI_MAX_ENEMIES : TBX_E_INVALID;
    case TBX_QUERY_AMI_PLAYER_TILE: return game == TBX_GAME_AMIDAR ? 3 : TBX_E_INVALID;
    case TBX_QUERY_SI_SHIP: return game == TBX_GAME_SPACE_INVADERS ? 8 : TBX_E_INVALID;
    case TBX_QUERY_BRK_FIND_BRICK: return game == TBX_GAME_BREAKOUT ? 1 : TBX_E_INVALID;
    case TBX_QUERY_AMI_TILES_MASK: return game == TBX_GAME_AMIDAR ? 32 : TBX_E_INVALID;
    case TBX_QUERY_AMI_RANDOM_TILE: return game == TBX_GAME_AMIDAR ? 4 : TBX_E_INVALID;
    case TBX_QUERY_AMI_RANDOM_DIR: return game == TBX_GAME_AMIDAR ? 2 : TBX_E_INVALID;
    default: return TBX_E_INVALID;
    }
}

static uint32_t arg_u(const double* a, int n, int i)
{
    const double x = i < n ? a[i] : 0.0;
    return x >= 4294967295.0 ? 0xFFFFFFFFu : x > 0.0 ? (uint32_t)x : 0u;
}

static int arg_i(const double* a, int n, int i)
{
    double x = i < n ? a[i] : 0.0;
    if (!(x > -2.0e9)) x = -2.0e9;
    if (x > 2.0e9) x = 2.0e9;
    return (int)x;
}
static double arg_d(const double* a, int n, int i) { return i < n ? a[i] : 0.0; }

static int brk_edit_one(const tbx_breakout_config_t* cfg, tbx_breakout_state_t* st, int op, const double* a, int n)
{
    (void)cfg;
    switch (op) {
    case TBX_EDIT_SET_LIVES: st->lives = arg_i(a, n, 0); return 0;
    case TBX_EDIT_SET_SCORE: st->score = arg_i(a, n, 0); return 0;
    case TBX_EDIT_SET_LEVEL: st->level = arg_i(a, n, 0); return 0;
    case TBX_EDIT_BRK_COLUMN_ALIVE:
        for (int j = 0; j < st->n_bricks; j++) if (st->bricks[j].col == arg_i(a, n, 0)) st->bricks[j].alive = arg_i(a, n, 1) != 0;
        return 0;
    case TBX_EDIT_BRK_ROW_ALIVE:
        for (int j = 0; j < st->n_bricks; j++) if (st->bricks[j].row == arg_i(a, n, 0)) st->bricks[j].alive = arg_i(a, n, 1) != 0;
        return 0;
    case TBX_EDIT_BRK_ALL_ALIVE:
        for (int j = 0; j < st->n_bricks; j++) st->bricks[j].alive = arg_i(a, n, 0) != 0;
        return 0;
    case TBX_EDIT_BRK_BRICK_ALIVE: {
        const int j = arg_i(a, n, 0);
        if (j >= 0 && j < st->n_bricks) st->bricks[j].alive = arg_i(a, n, 1) != 0;
        return 0;
    }
    case TBX_EDIT_BRK_PADDLE:
        st->paddle_x = arg_d(a, n, 0);
        if (n >= 2) st->paddle_y = arg_d(a, n, 1);
        return 0;
    case TBX_EDIT_BRK_BALL: {
        const int b = arg_i(a, n, 0);
        if (b >= 0 && b < TBX_BRK_MAX_BALLS && b < st->n_balls) {
            st->ball_x[b] = arg_d(a, n, 1); st->ball_y[b] = arg_d(a, n, 2); st->ball_vx[b] = arg_d(a, n, 3); st->ball_vy[b] = arg_d(a, n, 4);
        }
        return 0;
    }
    default: return -1;
    }
}

static int brk_is_channel(const tbx_breakout_state_t* st, int col)
{
    int bricks = 0, live = 0;
    for (int j = 0; j < st->n_bricks; j++) if (st->bricks[j].col == col) { bricks++; live += st->bricks[j].alive != 0; }
    return bricks > 0 && live == 0;
}

static int brk_reduce_one(const tbx_breakout_config_t* cfg, const tbx_breakout_state_t* st, int q, const double* a, int n, double* o, int width)
{
    switch (q) {
    case TBX_QUERY_BRK_BRICKS_REMAINING: { int c = 0; for (int j = 0; j < st->n_bricks; j++) c += st->bricks[j].alive != 0; o[0] = c; return 0; }
    case TBX_QUERY_BRK_NUM_BRICKS: o[0] = st->n_bricks; return 0;
    case TBX_QUERY_BRK_FIND_BRICK: {                          /* find_brick :400-404: `for i, b in enumerate(bricks): if pred(b): return i` */
        const int want = arg_i(a, n, 0);
        o[0] = -1;
        for (int j = 0; j < st->n_bricks; j++) {
            const int in_mask = (arg_u(a, n, 1 + j / 32) >> (j % 32)) & 1u;
            if (in_mask && (want < 0 || (st->bricks[j].alive != 0) == (want != 0))) { o[0] = j; break; }
        }
        return 0;
    }
    case TBX_QUERY_BRK_COLUMN: case TBX_QUERY_BRK_ROW: {
        int m = 0;
        for (int j = 0; j < st->n_bricks && m < width; j++)
            if ((q == TBX_QUERY_BRK_COLUMN ? st->bricks[j].col : st->bricks[j].row) == arg_i(a, n, 0)) o[m++] = st->bricks[j].alive != 0;
        for (; m < width; m++) o[m] = -1.0;
        return 0;
    }
    case TBX_QUERY_BRK_IS_CHANNEL: o[0] = brk_is_channel(st, arg_i(a, n, 0)); return 0;
    case TBX_QUERY_BRK_CHANNEL_COUNT: case TBX_QUERY_BRK_FIND_CHANNEL: {
        const int ncols = cfg->n_rows > 0 ? st->n_bricks / cfg->n_rows : 0;       /* num_columns: bricks // rows */
        int count = 0, first = -1;
        for (int c = 0; c < ncols; c++) if (brk_is_channel(st, c)) { count++; if (first < 0) first = c; }
        o[0] = q == TBX_QUERY_BRK_CHANNEL_COUNT ? count : first;
        return 0;
    }
    case TBX_QUERY_BRK_PADDLE: o[0] = st->paddle_x; o[1] = st->paddle_y; o[2] = st->paddle_vx; o[3] = st->paddle_vy; return 0;
    case TBX_QUERY_BRK_BALLS:
        o[0] = st->n_balls;
        for (int b = 0; b < TBX_BRK_MAX_BALLS; b++) {
            const int on = b < st->n_balls;
            o[1 + b] = on ? st->ball_x[b] : -1.0; o[1 + TBX_BRK_MAX_BALLS + b] = on ? st->ball_y[b] : -1.0;
            o[1 + 2 * TBX_BRK_MAX_BALLS + b] = on ? st->ball_vx[b] : -1.0; o[1 + 3 * TBX_BRK_MAX_BALLS + b] = on ? st->ball_vy[b] : -1.0;
        }
        return 0;
    default: return -1;
    }
}

static int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

/* get_random_tile(pred) (interventions/amidar.py:360-378) with the draw replaced by the counter rule of toybox_amd.h: the list
 * filter_tiles(pred) would return -- `for row in tiles: for tile in row: if pred(tile)` -- and element (r mod len) of it.
 * pred: tag in tag_mask, and (min_dist > 0) set_player_random_start's within_min_manhattan :543-546 as written:
 * `not all(d < min_enemy_distance for d in enemy distances)`.  Returns the number of candidates. */
static int ami_pick_tile(const tbx_amidar_state_t* st, int env_global, uint32_t seed, uint32_t draw, uint32_t tag_mask, int min_dist,
                         int* tx_out, int* ty_out)
{
    static __thread int cand[TBX_AMI_BOARD_H * TBX_AMI_BOARD_W];
    int count = 0;
    for (int ty = 0; ty < TBX_AMI_BOARD_H; ty++)
        for (int tx = 0; tx < TBX_AMI_BOARD_W; tx++) {
            if (!((tag_mask >> st->tiles[ty][tx]) & 1u)) continue;
            if (min_dist > 0) {
                int all_near = 1;
                for (int i = 0; i < st->n_enemies; i++) {
                    const int ex = floor_div(st->enemies[i].x, TBX_AMI_TILE_WX), ey = floor_div(st->enemies[i].y, TBX_AMI_TILE_WY);
                    if (!(abs(ex - tx) + abs(ey - ty) < min_dist)) all_near = 0;
                }
                if (all_near) continue;
            }
            cand[count++] = ty * TBX_AMI_BOARD_W + tx;
        }
    *tx_out = *ty_out = -1;
    if (count) {
        const uint64_t r = orc_splitmix64((uint64_t)seed ^ ((uint64_t)(uint32_t)env_global << 32) ^ (uint64_t)draw);
        const int c = cand[r % (uint64_t)count];
        *tx_out = c % TBX_AMI_BOARD_W; *ty_out = c / TBX_AMI_BOARD_W;
    }
    return count;
}

static int ami_edit_one(tbx_amidar_state_t* st, int op, const double* a, int n, int env)
{
    switch (op) {
    case TBX_EDIT_AMI_PLAYER_RANDOM_START: {
        int tx, ty;
        if (ami_pick_tile(st, (int)(arg_u(a, n, 2) + (uint32_t)env), arg_u(a, n, 0), arg_u(a, n, 1), 0xFu, arg_i(a, n, 3), &tx, &ty)) {
            st->player.x = tx * TBX_AMI_TILE_WX; st->player.y = ty * TBX_AMI_TILE_WY;     /* tile_to_worldpoint(pos) */
        }
        return 0;
    }
    case TBX_EDIT_SET_LIVES: st->lives = arg_i(a, n, 0); return 0;
    case TBX_EDIT_SET_SCORE: st->score = arg_i(a, n, 0); return 0;
    case TBX_EDIT_SET_LEVEL: st->level = arg_i(a, n, 0); return 0;
    case TBX_EDIT_AMI_JUMPS: st->jumps = arg_i(a, n, 0); return 0;
    case TBX_EDIT_AMI_TIMERS:
        if (arg_i(a, n, 0) >= 0) st->jump_timer = arg_i(a, n, 0);
        if (arg_i(a, n, 1) >= 0) st->chase_timer = arg_i(a, n, 1);
        return 0;
    case TBX_EDIT_AMI_TILE: {
        const int tx = arg_i(a, n, 0), ty = arg_i(a, n, 1);
        if (tx >= 0 && ty >= 0 && tx < TBX_AMI_BOARD_W && ty < TBX_AMI_BOARD_H) st->tiles[ty][tx] = (uint8_t)(arg_i(a, n, 2) & 3);
        return 0;
    }
    case TBX_EDIT_AMI_ENEMY_AI: {
        const int e = arg_i(a, n, 0);
        if (e >= 0 && e < st->n_enemies) {
            int32_t* ai = &st->enemies[e].ai.kind;
            for (int k = 0; k < 14; k++) ai[k] = arg_i(a, n, 1 + k);
        }
        return 0;
    }
    case TBX_EDIT_AMI_PLAYER_TILE:
        st->player.x = arg_i(a, n, 0) * TBX_AMI_TILE_WX; st->player.y = arg_i(a, n, 1) * TBX_AMI_TILE_WY;
        return 0;
    default: return -1;
    }
}

static int ami_tag(const tbx_amidar_state_t* st, int tx, int ty)
{
    return (tx < 0 || ty < 0 || tx >= TBX_AMI_BOARD_W || ty >= TBX_AMI_BOARD_H) ? -1 : st->tiles[ty][tx];
}
static void ami_distances(const tbx_amidar_state_t* st, int tx, int ty, double* o)
{
    for (int i = 0; i < TBX_AMI_MAX_ENEMIES; i++) {
        if (i >= st->n_enemies) { o[i] = -1.0; continue; }
        const int ex = floor_div(st->enemies[i].x, TBX_AMI_TILE_WX), ey = floor_div(st->enemies[i].y, TBX_AMI_TILE_WY);
        o[i] = abs(ex - tx) + abs(ey - ty);
    }
}

static int ami_reduce_one(const tbx_amidar_state_t* st, int q, const double* a, int n, double* o, int env)
{
    const int ptx = floor_div(st->player.x, TBX_AMI_TILE_WX), pty = floor_div(st->player.y, TBX_AMI_TILE_WY);
    switch (q) {
    case TBX_QUERY_AMI_TILES_MASK: {                         /* filter_tiles(lambda t: t.tag in ...) as one bitmap row per board row */
        const uint32_t tm = arg_u(a, n, 0);
        int c = 0;
        for (int ty = 0; ty < TBX_AMI_BOARD_H; ty++) {
            uint32_t bits = 0;
            for (int tx = 0; tx < TBX_AMI_BOARD_W; tx++)
                if ((tm >> st->tiles[ty][tx]) & 1u) { bits |= 1u << tx; c++; }
            o[ty] = bits;
        }
        o[31] = c;
        return 0;
    }
    case TBX_QUERY_AMI_RANDOM_TILE: {
        int tx, ty;
        const int c = ami_pick_tile(st, (int)(arg_u(a, n, 2) + (uint32_t)env), arg_u(a, n, 0), arg_u(a, n, 1), arg_u(a, n, 3), arg_i(a, n, 4), &tx, &ty);
        o[0] = tx; o[1] = ty; o[2] = c ? st->tiles[ty][tx] : -1; o[3] = c;
        return 0;
    }
    case TBX_QUERY_AMI_RANDOM_DIR: {                         /* get_random_dir_for_tile :550-583: a direction whose neighbour is walkable */
        const int tx = arg_i(a, n, 3), ty = arg_i(a, n, 4);
        const int dx[4] = {0, 0, -1, 1}, dy[4] = {-1, 1, 0, 0};           /* Up, Down, Left, Right */
        int dirs[4], nd = 0;
        for (int k = 0; k < 4; k++)
            if (ami_tag(st, tx + dx[k], ty + dy[k]) > TBX_TILE_EMPTY) dirs[nd++] = k;
        o[0] = -1; o[1] = nd;
        if (nd) {
            const uint64_t r = orc_splitmix64((uint64_t)arg_u(a, n, 0) ^ ((uint64_t)(arg_u(a, n, 2) + (uint32_t)env) << 32) ^ (uint64_t)arg_u(a, n, 1));
            o[0] = dirs[r % (uint64_t)nd];
        }
        return 0;
    }
    case TBX_QUERY_AMI_MODE: o[0] = st->jump_timer; o[1] = st->chase_timer; return 0;
    case TBX_QUERY_AMI_ANY_CAUGHT: { int any = 0; for (int i = 0; i < st->n_enemies; i++) any |= st->enemies[i].caught != 0; o[0] = any; return 0; }
    case TBX_QUERY_AMI_TILE: o[0] = ami_tag(st, arg_i(a, n, 0), arg_i(a, n, 1)); return 0;
    case TBX_QUERY_AMI_COUNT_TILES: {
        int c = 0;
        for (int ty = 0; ty < TBX_AMI_BOARD_H; ty++) for (int tx = 0; tx < TBX_AMI_BOARD_W; tx++) c += st->tiles[ty][tx] == arg_i(a, n, 0);
        o[0] = c;
        return 0;
    }
    case TBX_QUERY_AMI_ADJACENT: {
        const int tx = arg_i(a, n, 0), ty = arg_i(a, n, 1);
        o[0] = ami_tag(st, tx, ty - 1); o[1] = ami_tag(st, tx - 1, ty); o[2] = ami_tag(st, tx + 1, ty); o[3] = ami_tag(st, tx, ty + 1);
        return 0;
    }
    case TBX_QUERY_AMI_ENEMY_DISTANCES: ami_distances(st, arg_i(a, n, 0), arg_i(a, n, 1), o); return 0;
    case TBX_QUERY_AMI_PLAYER_TILE: o[0] = ptx; o[1] = pty; o[2] = ami_tag(st, ptx, pty); return 0;
    case TBX_QUERY_AMI_PLAYER_ENEMY_DISTANCES: ami_distances(st, ptx, pty, o); return 0;
    case TBX_QUERY_AMI_PLAYER_ON_PAINTED: o[0] = ami_tag(st, ptx, pty) == TBX_TILE_PAINTED; return 0;
    case TBX_QUERY_AMI_PLAYER_NEAR_UNPAINTED: {
        int near = 0, painted = 0;
        for (int ty = 0; ty < TBX_AMI_BOARD_H; ty++)
            for (int tx = 0; tx < TBX_AMI_BOARD_W; tx++)
                if (abs(tx - ptx) + abs(ty - pty) < arg_i(a, n, 0) && st->tiles[ty][tx] != TBX_TILE_EMPTY) { near++; painted += st->tiles[ty][tx] == TBX_TILE_PAINTED; }
        o[0] = painted != near;
        return 0;
    }
    default: return -1;
    }
}

static int si_edit_one(tbx_si_state_t* st, int op, const double* a, int n)
{
    switch (op) {
    case TBX_EDIT_SET_LIVES: st->lives = arg_i(a, n, 0); return 0;
    case TBX_EDIT_SET_SCORE: st->score = arg_i(a, n, 0); return 0;
    case TBX_EDIT_SET_LEVEL: st->level = arg_i(a, n, 0); return 0;
    case TBX_EDIT_SI_UFO_APPEARANCE: st->ufo_appearance_counter = arg_i(a, n, 0); return 0;
    default: return -1;
    }
}

int tbx_edit_device(tbx_engine* e, int op, const double* args, int n_args, int per_env, const uint8_t* mask, void* stream)
{
    (void)stream;
    if (!e) return TBX_E_INVALID;
    if (n_args < 0 || n_args > TBX_EDIT_MAX_ARGS || (n_args > 0 && !args)) return fail(e, TBX_E_INVALID, "bad intervention arguments");
    for (int i = 0; i < e->n; i++) {
        if (mask && !mask[i]) continue;
        const double* a = per_env ? args + (size_t)i * n_args : args;
        void* st = e->states + e->ssz * (size_t)i;
        int rc = -1;
        if (e->game == TBX_GAME_BREAKOUT) rc = brk_edit_one((const tbx_breakout_config_t*)e->cfg, (tbx_breakout_state_t*)st, op, a, n_args);
        else if (e->game == TBX_GAME_AMIDAR) rc = ami_edit_one((tbx_amidar_state_t*)st, op, a, n_args, i);
        else if (e->game == TBX_GAME_SPACE_INVADERS) rc = si_edit_one((tbx_si_state_t*)st, op, a, n_args);
        if (rc) return fail(e, TBX_E_INVALID, "unknown edit for this game");
    }
    return TBX_OK;
}

int tbx_edit(tbx_engine* e, int op, const double* args, int n_args, int per_env, const uint8_t* mask)
{
    return tbx_edit_device(e, op, args, n_args, per_env, mask, NULL);
}

int tbx_reduce_device(tbx_engine* e, int query, const double* args, int n_args, int per_env, double* out, void* stream)
{
    (void)stream;
    if (!e) return TBX_E_INVALID;
    const int width = tbx_reduce_width(e->game, query);
    if (width < 0) return fail(e, TBX_E_INVALID, "unknown query for this game");
    if (!out) return fail(e, TBX_E_INVALID, "output pointer is NULL");
    if (n_args < 0 || n_args > TBX_EDIT_MAX_ARGS || (n_args > 0 && !args)) return fail(e, TBX_E_INVALID, "bad intervention arguments");
    for (int i = 0; i < e->n; i++) {
        const double* a = per_env ? args + (size_t)i * n_args : args;
        const void* st = e->states + e->ssz * (size_t)i;
        double* o = out + (size_t)i * width;
        int rc = -1;
        if (e->game == TBX_GAME_BREAKOUT) rc = brk_reduce_one((const tbx_breakout_config_t*)e->cfg, (const tbx_breakout_state_t*)st, query, a, n_args, o, width);
        else if (e->game == TBX_GAME_AMIDAR) rc = ami_reduce_one((const tbx_amidar_state_t*)st, query, a, n_args, o, i);
        else if (e->game == TBX_GAME_SPACE_INVADERS && query == TBX_QUERY_SI_SHIP) {
            const tbx_si_state_t* s = (const tbx_si_state_t*)st;
            o[0] = s->ship_x; o[1] = s->ship_y; o[2] = s->ship_w; o[3] = s->ship_h; o[4] = s->ship_speed;
            o[5] = s->ship_alive != 0; o[6] = s->ship_death_counter; o[7] = s->ship_death_hit_1 != 0;
            rc = 0;
        }
        if (rc) return fail(e, TBX_E_INVALID, "unknown query for this game");
    }
    return TBX_OK;
}

int tbx_reduce(tbx_engine* e, int query, const double* args, int n_args, int per_env, double* out)
{
    return tbx_reduce_device(e, query, args, n_args, per_env, out, NULL);
}

/* ---------------------------------------------------------------- record gather
 * The product does this with RCCL over xGMI (toybox_amd/csrc/gather.hip).  The checker restates the same calls over a POSIX
 * shared-memory segment named by the "unique id", so that the multi-process CPU tests (world size 2) drive the real control
 * flow -- id made by rank 0, handed over out of band, collective init, one all-gather per step, max-reduction -- through the
 * same C-ABI.  Layout of the segment: seq[2][64] call counters (records / scalar), then two parity buffers of
 * nranks * width records, then two parity buffers of nranks doubles. */

typedef struct {
    volatile uint64_t seq[3][64];   /* records, scalar, init barrier */
} gshm_head_t;

static size_t gshm_bytes(int nranks, int width) { return sizeof(gshm_head_t) + 2 * (size_t)nranks * width * 8 + 2 * (size_t)nranks * 8; }

int tbx_gather_unique_id(void* id_out, size_t id_bytes)
{
    if (!id_out || id_bytes != TBX_GATHER_ID_BYTES) return fail(NULL, TBX_E_INVALID, "id buffer must be TBX_GATHER_ID_BYTES long");
    memset(id_out, 0, id_bytes);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf((char*)id_out, id_bytes, "/tbx_orc_%ld_%lld_%ld", (long)getpid(), (long long)ts.tv_sec, (long)ts.tv_nsec);
    return TBX_OK;
}

static void gather_close(tbx_engine* e)
{
    if (e->gshm) munmap(e->gshm, e->gshm_len);
    if (e->gather_on && e->gather_rank == 0 && e->gshm_name[0]) shm_unlink(e->gshm_name);
    free(e->gathered);
    if (e->ring[0]) {
        e->packed = e->packed_own;
        free(e->ring[0]); free(e->ring[1]);
        e->ring[0] = e->ring[1] = NULL;
    }
    e->gather_every = 1; e->gather_fill = 0; e->ring_par = 0; e->gather_advance = 0;
    e->gshm = NULL; e->gathered = NULL; e->gather_on = 0;
}

int tbx_gather_init(tbx_engine* e, int nranks, int rank, int records_per_rank, const void* id, size_t id_bytes)
{
    if (!e) return TBX_E_INVALID;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(e, TBX_E_INVALID, "gather: rank / nranks out of range");
    if (records_per_rank < e->n) return fail(e, TBX_E_INVALID, "gather: records_per_rank must be >= the engine's env count");
    if (!id || id_bytes != TBX_GATHER_ID_BYTES) return fail(e, TBX_E_INVALID, "gather: id must be TBX_GATHER_ID_BYTES long");
    if (nranks > 64) return fail(e, TBX_E_UNSUPPORTED, "gather: the CPU checker handles at most 64 ranks");
    gather_close(e);
    e->gather_ranks = nranks; e->gather_rank = rank; e->gather_width = records_per_rank;
    e->gather_calls[0] = e->gather_calls[1] = 0;
    const int K = e->opt[TBX_OPT_GATHER_EVERY] > 1 ? e->opt[TBX_OPT_GATHER_EVERY] : 1;
    e->gather_every = K;
    e->gathered = (uint64_t*)calloc((size_t)nranks * K * records_per_rank, 8);
    if (K > 1) {
        for (int k = 0; k < 2; k++) e->ring[k] = (uint64_t*)calloc((size_t)K * records_per_rank, 8);
        memcpy(e->ring[0], e->packed, (size_t)e->n * 8);   /* TBX_BUF_PACKED keeps reading what it read */
        e->packed_own = e->packed;
        e->packed = e->ring[0];
    }
    snprintf(e->gshm_name, sizeof e->gshm_name, "%s", (const char*)id);
    e->gshm_len = gshm_bytes(nranks, K * records_per_rank);
    if (nranks > 1) {
        if (e->gshm_name[0] != '/') return fail(e, TBX_E_INVALID, "gather: not an id made by tbx_gather_unique_id");
        int fd = shm_open(e->gshm_name, O_CREAT | O_RDWR, 0600);
        if (fd < 0) return fail(e, TBX_E_NO_DEVICE, "gather: shm_open failed");
        if (ftruncate(fd, (off_t)e->gshm_len) != 0) { close(fd); return fail(e, TBX_E_NO_DEVICE, "gather: ftruncate failed"); }
        e->gshm = mmap(NULL, e->gshm_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);   /* a fresh segment reads as zeros */
        close(fd);
        if (e->gshm == MAP_FAILED) { e->gshm = NULL; return fail(e, TBX_E_NO_DEVICE, "gather: mmap failed"); }
        /* collective like ncclCommInitRank: returns once every rank has attached */
        gshm_head_t* h = (gshm_head_t*)e->gshm;
        __atomic_store_n(&h->seq[2][rank], 1, __ATOMIC_RELEASE);
        for (int r = 0; r < nranks; r++)
            while (__atomic_load_n(&h->seq[2][r], __ATOMIC_ACQUIRE) < 1) sched_yield();
    } else {
        e->gshm = NULL;
    }
    e->gather_on = 1;
    return TBX_OK;
}

/* one collective step over the segment: publish `bytes` at slot `rank` of parity buffer `base`, wait for every rank's
 * counter to reach this call's number, copy all slots out */
static void gshm_exchange(tbx_engine* e, int which, char* base, size_t slot, const void* mine, void* all)
{
    gshm_head_t* h = (gshm_head_t*)e->gshm;
    const uint64_t call = ++e->gather_calls[which];
    char* buf = base + (call & 1u) * (size_t)e->gather_ranks * slot;
    memcpy(buf + (size_t)e->gather_rank * slot, mine, slot);
    __atomic_store_n(&h->seq[which][e->gather_rank], call, __ATOMIC_RELEASE);
    for (int r = 0; r < e->gather_ranks; r++)
        while (__atomic_load_n(&h->seq[which][r], __ATOMIC_ACQUIRE) < call) sched_yield();
    memcpy(all, buf, (size_t)e->gather_ranks * slot);
}

int tbx_gather(tbx_engine* e, uint64_t* out, void* stream)
{
    (void)stream;
    if (!e) return TBX_E_INVALID;
    if (!e->gather_on) return fail(e, TBX_E_INVALID, "tbx_gather_init has not been called");
    uint64_t* dst = out ? out : e->gathered;
    if (e->gather_every > 1) {
        /* ring mode: K - 1 of K calls only count; the K-th exchanges the ring [K][width] as one slot */
        e->gather_advance = 1;
        if (++e->gather_fill < e->gather_every) return TBX_OK;
        const size_t ring_bytes = (size_t)e->gather_every * e->gather_width * 8;
        const uint64_t* ring = e->ring[e->ring_par];
        if (e->gather_ranks == 1) memcpy(dst, ring, ring_bytes);
        else gshm_exchange(e, 0, (char*)e->gshm + sizeof(gshm_head_t), ring_bytes, ring, dst);
        e->gather_fill = 0;
        e->ring_par ^= 1;
        return TBX_OK;
    }
    const size_t slot = (size_t)e->gather_width * 8;
    uint64_t* mine = (uint64_t*)calloc(1, slot);
    memcpy(mine, e->packed, (size_t)e->n * 8);
    if (e->gather_ranks == 1) memcpy(dst, mine, slot);
    else gshm_exchange(e, 0, (char*)e->gshm + sizeof(gshm_head_t), slot, mine, dst);
    free(mine);
    return TBX_OK;
}

int tbx_gather_wait(tbx_engine* e, void* stream)
{
    (void)stream;
    if (!e) return TBX_E_INVALID;
    if (!e->gather_on) return fail(e, TBX_E_INVALID, "tbx_gather_init has not been called");
    return TBX_OK;
}

int tbx_gather_host(tbx_engine* e, uint64_t* out)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather_on) return fail(e, TBX_E_INVALID, "tbx_gather_init has not been called");
    if (!out) return fail(e, TBX_E_INVALID, "output pointer is NULL");
    memcpy(out, e->gathered, (size_t)e->gather_ranks * e->gather_every * e->gather_width * 8);
    return TBX_OK;
}

int tbx_gather_reduce_max(tbx_engine* e, double* inout)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather_on) return fail(e, TBX_E_INVALID, "tbx_gather_init has not been called");
    if (!inout) return fail(e, TBX_E_INVALID, "value pointer is NULL");
    if (e->gather_ranks > 1) {
        double all[64];
        char* base = (char*)e->gshm + sizeof(gshm_head_t) + 2 * (size_t)e->gather_ranks * e->gather_every * e->gather_width * 8;
        gshm_exchange(e, 1, base, sizeof(double), inout, all);
        for (int r = 0; r < e->gather_ranks; r++) if (all[r] > *inout) *inout = all[r];
    }
    return TBX_OK;
}

/* launch-time choices of the device engine: stored and reported, without effect on a scalar CPU restatement */
int tbx_set_option(tbx_engine* e, int option, int value)
{
    static const int hi[TBX_OPT_COUNT] = {3, 2, 64, 1, 1, 64, 1, 2, 1 << 20, 4};
    if (!e) return TBX_E_INVALID;
    if (option < 0 || option >= TBX_OPT_COUNT) return fail(e, TBX_E_INVALID, "unknown option");
    if (value < (option == TBX_OPT_GATHER_EVERY ? 1 : 0) || value > hi[option]) return fail(e, TBX_E_INVALID, "option value out of range");
    e->opt[option] = value;
    return TBX_OK;
}

int tbx_get_option(tbx_engine* e, int option, int* value_out)
{
    if (!e) return TBX_E_INVALID;
    if (value_out && (option == TBX_OPT_PIPELINE_ACTIVE || option == TBX_OPT_RECORDS_ACTIVE || option == TBX_OPT_RENDER_STEP_FUSED || option == TBX_OPT_FUSED_OVERLAP_ACTIVE || option == TBX_OPT_ROLLOUT_CHUNKS_ACTIVE)) { *value_out = 0; return TBX_OK; }   /* nothing to overlap on one CPU thread */
    if (option < 0 || option >= TBX_OPT_COUNT || !value_out) return fail(e, TBX_E_INVALID, "unknown option");
    *value_out = e->opt[option];
    return TBX_OK;
}

int tbx_gather_nranks(tbx_engine* e)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather_on) return fail(e, TBX_E_INVALID, "tbx_gather_init has not been called");
    return e->gather_ranks;
}

const char* tbx_gather_library(tbx_engine* e) { (void)e; return "oracle: POSIX shared memory"; }

int tbx_gather_every(tbx_engine* e)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather_on) return fail(e, TBX_E_INVALID, "tbx_gather_init has not been called");
    return e->gather_every;
}

int tbx_gather_fill(tbx_engine* e)
{
    if (!e) return TBX_E_INVALID;
    if (!e->gather_on) return fail(e, TBX_E_INVALID, "tbx_gather_init has not been called");
    return e->gather_fill;
}

int tbx_sync(tbx_engine* e)
{
    if (!e) return TBX_E_INVALID;
    return take_action_error(e);
}

/* ---------------------------------------------------------------- agent-side preprocessing */

static uint8_t* agent_newest_plane(tbx_engine* e)
{
    if (e->aring) return e->aring + (size_t)e->ahead * e->n * e->acfg.out_h * e->acfg.out_w;
    return e->aplane;
}

static int agent_no_stack(tbx_engine* e)
{
    return fail(e, TBX_E_INVALID, "no rolled stack on the device with tbx_agent_config_t::new_plane = 2: the planes are in TBX_BUF_AGENT_RING (newest: TBX_BUF_AGENT_PLANE)");
}

static void agent_free(tbx_engine* e)
{
    free(e->gray_a); free(e->gray_b); free(e->aobs); free(e->afin); free(e->adone); free(e->areward); free(e->aplane); free(e->aring);
    e->aplane = e->aring = NULL;
    e->ahead = 0;
    free(e->ep_ret); free(e->ep_len); free(e->ep_index); free(e->prev_lives); free(e->ep_len_out); free(e->ep_done);
    free(e->ep_ret_out); free(e->was_real_done); free(e->needs_reset); free(e->noop_override);
    e->was_real_done = e->needs_reset = NULL; e->noop_override = NULL;
    e->gray_a = e->gray_b = e->aobs = e->afin = e->adone = e->ep_done = NULL;
    e->areward = e->ep_ret_out = NULL;
    e->ep_ret = e->ep_len = e->ep_index = e->prev_lives = e->ep_len_out = NULL;
}

int tbx_agent_init(tbx_engine* e, const tbx_agent_config_t* cfg)
{
    if (!e) return TBX_E_INVALID;
    if (!cfg) return fail(e, TBX_E_INVALID, "agent config is NULL");
    int H, W;
    orc_frame_dims(e->game, &H, &W);
    if (cfg->skip < 1 || cfg->skip > 64 || cfg->stack < 1 || cfg->stack > 4 || cfg->out_h < 1 || cfg->out_w < 1 ||
        cfg->out_h > H || cfg->out_w > W || cfg->out_w > 128 || cfg->out_h * cfg->out_w > 84 * 84 || cfg->noop_max < 0 ||
        cfg->noop_max > 1000 || cfg->stack_fill < 0 || cfg->stack_fill > 1 || cfg->new_plane < 0 || cfg->new_plane > 2)
        return fail(e, TBX_E_INVALID, "agent config out of range (skip 1..64, stack 1..4, 1 <= out <= frame, out_w <= 128, out_h*out_w <= 7056, noop_max 0..1000, stack_fill 0..1, new_plane 0..2)");
    if ((H + cfg->out_h - 1) / cfg->out_h + 1 > 8 || (W + cfg->out_w - 1) / cfg->out_w + 1 > 8)
        return fail(e, TBX_E_UNSUPPORTED, "agent: the resize ratio needs more than 8 taps per axis");
    agent_free(e);
    size_t n = (size_t)e->n;
    e->acfg = *cfg;
    e->gray_a = (uint8_t*)calloc(n, (size_t)H * W);
    e->gray_b = (uint8_t*)calloc(n, (size_t)H * W);
    if (cfg->new_plane == 2) e->aring = (uint8_t*)calloc(n, (size_t)cfg->out_h * cfg->out_w * cfg->stack);
    else e->aobs = (uint8_t*)calloc(n, (size_t)cfg->out_h * cfg->out_w * cfg->stack);
    e->aplane = cfg->new_plane == 1 ? (uint8_t*)calloc(n, (size_t)cfg->out_h * cfg->out_w) : NULL;
    e->afin = (uint8_t*)calloc(n, 1);
    e->adone = (uint8_t*)calloc(n, 1);
    e->areward = (float*)calloc(n, sizeof(float));
    e->ep_ret = (int32_t*)calloc(n, 4); e->ep_len = (int32_t*)calloc(n, 4); e->ep_index = (int32_t*)calloc(n, 4);
    e->prev_lives = (int32_t*)calloc(n, 4); e->ep_len_out = (int32_t*)calloc(n, 4);
    e->ep_done = (uint8_t*)calloc(n, 1);
    e->ep_ret_out = (float*)calloc(n, sizeof(float));
    e->was_real_done = (uint8_t*)malloc(n);
    memset(e->was_real_done, 1, n);               /* EpisodicLifeEnv.__init__: was_real_done = True (:164) */
    e->needs_reset = (uint8_t*)calloc(n, 1);
    e->agent_on = 1;
    return TBX_OK;
}

/* ---- the wrapper stack of the reference's agents, one env at a time, every class a function over the one below it:
 *   ToyboxBaseEnv (toybox/envs/atari/base.py:115-156)
 *   -> NoopResetEnv -> MaxAndSkipEnv           make_wrapper, baselines/baselines/common/atari_wrappers.py:324-335; classes :108-135, :193-219
 *   -> bench.Monitor                           common/cmd_util.py:32 (allow_early_resets=True); bench/monitor.py:45-76
 *   -> EpisodicLifeEnv -> FireResetEnv -> WarpFrame -> ClipRewardEnv   wrap_deepmind :346-360; classes :157-191, :137-155, :230-244, :221-227
 *   -> DummyVecEnv's reset-on-done             common/vec_env/dummy_vec_env.py:45-60
 *   -> VecFrameStack                           common/vec_env/vec_frame_stack.py:17-30
 * State kept exactly where the Python keeps it: MaxAndSkipEnv._obs_buffer is two persistent full frames per env, zero at
 * construction (gray_a / gray_b), never cleared by a reset; EpisodicLifeEnv.{lives, was_real_done}; Monitor.{rewards,
 * needs_reset}; ToyboxBaseEnv.score (e->prev).  gym's TimeLimit around the env (:326) has no limit configured
 * (toybox/__init__.py registers the ids without max_episode_steps) and is the identity.
 * The one thing that is not the Python's: where NoopResetEnv asks numpy's RandomState for the number of no-ops, the count is
 * 1 + splitmix64(noop_seed ^ (global env << 32) ^ episode index) % noop_max -- unless a count was injected with
 * tbx_agent_set_noops (NoopResetEnv.override_num_noops, :115-123). */
typedef struct {
    tbx_engine* e;
    int i;
    int H, W;
    const uint8_t* obs;     /* what the last wrapper call returned: a full gray frame (scratch `raw`, or `mx`) */
    uint8_t *raw, *mx, *keep; /* per-call scratch frames */
} wrap_t;

static void* env_state(tbx_engine* e, int i) { return e->states + (size_t)i * e->ssz; }

static void raw_scalars(tbx_engine* e, int i, int32_t* score, int32_t* lives)
{
    int32_t level;
    orc_get_scalars(e->game, env_state(e, i), 1, score, lives, &level);
}
static int raw_lives(tbx_engine* e, int i) { int32_t sc, lv; raw_scalars(e, i, &sc, &lv); return lv; }

static void raw_frame(wrap_t* w, uint8_t* out) { orc_render_batch(w->e->game, w->e->cfg, env_state(w->e, w->i), 1, out, 1, 1); }

/* ToyboxBaseEnv.step (base.py:115-149): one frame, reward = max(score - self.score, 0), done = lives <= 0 */
static int base_step(wrap_t* w, uint32_t buttons, int* reward)
{
    tbx_engine* e = w->e;
    const int i = w->i;
    void* st = env_state(e, i);
    switch (e->game) {
    case TBX_GAME_BREAKOUT: orc_breakout_step((const tbx_breakout_config_t*)e->cfg, (tbx_breakout_state_t*)st, buttons); break;
    case TBX_GAME_SPACE_INVADERS: orc_si_step((const tbx_si_config_t*)e->cfg, (tbx_si_state_t*)st, buttons); break;
    case TBX_GAME_GRIDWORLD: orc_gridworld_step((const tbx_gridworld_config_t*)e->cfg, (tbx_gridworld_state_t*)st, buttons); break;
    default: orc_amidar_step((const tbx_amidar_config_t*)e->cfg, (tbx_amidar_state_t*)st, buttons); break;
    }
    int32_t sc, lv;
    raw_scalars(e, i, &sc, &lv);
    *reward = sc - e->prev[i] > 0 ? sc - e->prev[i] : 0;
    e->prev[i] = sc;
    e->reward[i] = *reward; e->done[i] = lv <= 0; e->lives[i] = lv; e->score[i] = sc;
    return lv <= 0;
}
/* ToyboxBaseEnv.reset (base.py:151-156) */
static void base_reset(wrap_t* w)
{
    tbx_engine* e = w->e;
    const int i = w->i;
    uint64_t* sim = e->sim + 2 * (size_t)i;
    void* st = env_state(e, i);
    switch (e->game) {
    case TBX_GAME_BREAKOUT: orc_breakout_new_game((const tbx_breakout_config_t*)e->cfg, sim, (tbx_breakout_state_t*)st); break;
    case TBX_GAME_SPACE_INVADERS: orc_si_new_game((const tbx_si_config_t*)e->cfg, sim, (tbx_si_state_t*)st); break;
    case TBX_GAME_GRIDWORLD: orc_gridworld_new_game((const tbx_gridworld_config_t*)e->cfg, sim, (tbx_gridworld_state_t*)st); break;
    default: orc_amidar_new_game((const tbx_amidar_config_t*)e->cfg, sim, (tbx_amidar_state_t*)st); break;
    }
    int32_t sc, lv;
    raw_scalars(e, i, &sc, &lv);
    e->prev[i] = sc;
}

/* NoopResetEnv.reset (:117-132) */
static void noop_reset(wrap_t* w)
{
    tbx_engine* e = w->e;
    const int i = w->i;
    base_reset(w);
    raw_frame(w, w->raw);
    w->obs = w->raw;
    int k = 0;
    if (e->noop_override && e->noop_override[i] > 0) k = e->noop_override[i];
    else if (e->acfg.noop_max > 0) {
        const uint64_t env_global = e->acfg.env_offset + (uint64_t)i;
        k = 1 + (int)(orc_splitmix64(e->acfg.noop_seed ^ (env_global << 32) ^ (uint64_t)(uint32_t)e->ep_index[i]) % (uint64_t)e->acfg.noop_max);
    }
    for (int j = 0; j < k; j++) {
        int r;
        const int done = base_step(w, 0, &r);
        if (done) base_reset(w);
    }
    if (k > 0) raw_frame(w, w->raw);      /* obs of the last no-op step, or of the reset that followed it */
}

/* MaxAndSkipEnv.step (:201-216): the two-frame buffer persists between calls; a step cut short by `done` leaves the slots
 * it did not reach as they were */
static int skip_step(wrap_t* w, uint32_t buttons, int* total)
{
    tbx_engine* e = w->e;
    const int i = w->i, skip = e->acfg.skip;
    const size_t fsz = (size_t)w->H * w->W;
    uint8_t *ba = e->gray_a + fsz * (size_t)i, *bb = e->gray_b + fsz * (size_t)i;
    int done = 0;
    *total = 0;
    for (int f = 0; f < skip; f++) {
        int r;
        done = base_step(w, buttons, &r);
        if (f == skip - 2) raw_frame(w, ba);
        if (f == skip - 1) raw_frame(w, bb);
        *total += r;
        if (done) break;
    }
    for (size_t p = 0; p < fsz; p++) w->mx[p] = ba[p] > bb[p] ? ba[p] : bb[p];
    w->obs = w->mx;
    return done;
}

/* Monitor.step / update (bench/monitor.py:51-76).  Where Monitor raises "Tried to step environment that needs reset" the
 * step is still carried out -- as the stack without a Monitor would -- and the condition is reported (TBX_E_NEEDS_RESET). */
static int monitor_step(wrap_t* w, uint32_t buttons, int* total)
{
    tbx_engine* e = w->e;
    const int i = w->i;
    const int stale = e->needs_reset[i];
    const int done = skip_step(w, buttons, total);
    if (stale) { e->pending_needs_reset = 1; return done; }
    e->ep_ret[i] += *total; e->ep_len[i] += 1;
    if (done) {
        e->needs_reset[i] = 1;
        e->ep_done[i] = 1; e->ep_ret_out[i] = (float)e->ep_ret[i]; e->ep_len_out[i] = e->ep_len[i];
    }
    return done;
}
/* Monitor.reset (:36-49, allow_early_resets=True) over MaxAndSkipEnv.reset (:218-219) over NoopResetEnv.reset */
static void monitor_reset(wrap_t* w)
{
    tbx_engine* e = w->e;
    e->ep_ret[w->i] = 0; e->ep_len[w->i] = 0; e->needs_reset[w->i] = 0;
    e->ep_index[w->i] += 1;
    noop_reset(w);
}

/* EpisodicLifeEnv (:157-191); with the wrapper off these are Monitor's step / reset */
static int episodic_step(wrap_t* w, uint32_t buttons, int* total)
{
    tbx_engine* e = w->e;
    int done = monitor_step(w, buttons, total);
    if (!e->acfg.episodic_life) return done;
    e->was_real_done[w->i] = (uint8_t)done;
    const int lives = raw_lives(e, w->i);
    if (lives < e->prev_lives[w->i] && lives > 0) done = 1;
    e->prev_lives[w->i] = lives;
    return done;
}
static void episodic_reset(wrap_t* w)
{
    tbx_engine* e = w->e;
    if (!e->acfg.episodic_life) { monitor_reset(w); return; }
    if (e->was_real_done[w->i]) monitor_reset(w);
    else { int t; monitor_step(w, 0, &t); }      /* no-op step to advance from the lost-life state; its `done` is ignored (:186-187) */
    e->prev_lives[w->i] = raw_lives(e, w->i);
}

/* FireResetEnv.reset (:144-152); with the wrapper off it is the reset below it */
static void top_reset(wrap_t* w)
{
    tbx_engine* e = w->e;
    episodic_reset(w);
    if (e->acfg.fire_reset) {
        int32_t legal[18];
        int t;
        orc_legal_actions(e->game, legal, 18);
        if (episodic_step(w, (uint32_t)orc_ale_action_to_buttons(legal[1]), &t)) episodic_reset(w);
        const int done = episodic_step(w, (uint32_t)orc_ale_action_to_buttons(legal[2]), &t);
        if (done) {                               /* `obs` stays what step(2) returned; the reset does not replace it */
            memcpy(w->keep, w->obs, (size_t)w->H * w->W);
            episodic_reset(w);
            w->obs = w->keep;
        }
    }
}

/* WarpFrame + VecFrameStack for one env */
static void commit_obs(wrap_t* w, int zero_stack)
{
    tbx_engine* e = w->e;
    const int oh = e->acfg.out_h, ow = e->acfg.out_w, st = e->acfg.stack;
    uint8_t* small = (uint8_t*)malloc((size_t)oh * ow);
    orc_warp_area(w->obs, w->H, w->W, small, oh, ow);
    if (e->aring) {
        /* toybox_amd.h, new_plane = 2: the new frame into the head slot; a stack that starts afresh (reset / done) gets its other
         * slots rewritten -- zeros (VecFrameStack) or the observation (FrameStack.reset) */
        const size_t px = (size_t)oh * ow, slot = (size_t)e->n * px;
        for (int k = 0; k < st; k++) {
            uint8_t* dst = e->aring + (size_t)((e->ahead + k) % st) * slot + (size_t)w->i * px;
            if (k == 0 || (zero_stack && e->acfg.stack_fill)) memcpy(dst, small, px);
            else if (zero_stack) memset(dst, 0, px);
        }
    } else {
        orc_stack_push(e->aobs + (size_t)w->i * oh * ow * st, small, oh, ow, st, zero_stack ? (e->acfg.stack_fill ? 2 : 1) : 0);
    }
    if (e->aplane) memcpy(e->aplane + (size_t)w->i * oh * ow, small, (size_t)oh * ow);   /* what the worker sends: the new frame alone */
    free(small);
}

static void wrap_open(wrap_t* w, tbx_engine* e, int i)
{
    w->e = e; w->i = i;
    orc_frame_dims(e->game, &w->H, &w->W);
    w->raw = (uint8_t*)malloc((size_t)w->H * w->W);
    w->mx = (uint8_t*)malloc((size_t)w->H * w->W);
    w->keep = (uint8_t*)malloc((size_t)w->H * w->W);
    w->obs = w->raw;
}
static void wrap_close(wrap_t* w) { free(w->raw); free(w->mx); free(w->keep); }

int tbx_agent_set_noops(tbx_engine* e, const int32_t* counts)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent_on) return fail(e, TBX_E_INVALID, "tbx_agent_init has not been called");
    free(e->noop_override);
    e->noop_override = NULL;
    if (counts) {
        e->noop_override = (int32_t*)malloc((size_t)e->n * 4);
        memcpy(e->noop_override, counts, (size_t)e->n * 4);
    }
    return TBX_OK;
}

/* venv.reset(): DummyVecEnv.reset (:56-60) + VecFrameStack.reset (:29-33) */
int tbx_agent_reset(tbx_engine* e, uint8_t* obs)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent_on) return fail(e, TBX_E_INVALID, "tbx_agent_init has not been called");
    if (obs && !e->aobs) return agent_no_stack(e);
    const int n = e->n;
    memset(e->ep_done, 0, (size_t)n);
    if (e->aring) e->ahead = (e->ahead + 1) % e->acfg.stack;
#pragma omp parallel for schedule(static) num_threads(e->threads > 1 ? e->threads : 1)
    for (int i = 0; i < n; i++) {
        wrap_t w;
        wrap_open(&w, e, i);
        top_reset(&w);
        commit_obs(&w, 1);
        wrap_close(&w);
    }
    memset(e->ep_done, 0, (size_t)n);     /* records of games that ended inside the reset procedure are not reported here */
    if (obs) memcpy(obs, e->aobs, (size_t)n * e->acfg.out_h * e->acfg.out_w * e->acfg.stack);
    return TBX_OK;
}

int tbx_agent_episodes(tbx_engine* e, uint8_t* ep_done, float* ep_return, int32_t* ep_length)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent_on) return fail(e, TBX_E_INVALID, "tbx_agent_init has not been called");
    const size_t n = (size_t)e->n;
    if (ep_done) memcpy(ep_done, e->ep_done, n);
    if (ep_return) memcpy(ep_return, e->ep_ret_out, n * 4);
    if (ep_length) memcpy(ep_length, e->ep_len_out, n * 4);
    return TBX_OK;
}

/* venv.step(actions): DummyVecEnv.step_wait (:45-54) + VecFrameStack.step_wait (:19-27) */
int tbx_agent_step_device(tbx_engine* e, const int32_t* actions, void* stream)
{
    (void)stream;
    if (!e) return TBX_E_INVALID;
    if (!e->agent_on) return fail(e, TBX_E_INVALID, "tbx_agent_init has not been called");
    if (!actions) return fail(e, TBX_E_INVALID, "actions pointer is NULL");
    const int n = e->n;
    int bad = 0;
    if (e->aring) e->ahead = (e->ahead + 1) % e->acfg.stack;
#pragma omp parallel for schedule(static) num_threads(e->threads > 1 ? e->threads : 1) reduction(| : bad)
    for (int i = 0; i < n; i++) {
        wrap_t w;
        wrap_open(&w, e, i);
        int b = orc_ale_action_to_buttons(actions[i]);
        if (b < 0) { b = 0; bad |= 1; }
        e->ep_done[i] = 0;
        int total;
        const int done = episodic_step(&w, (uint32_t)b, &total);            /* FireResetEnv.step passes through */
        e->areward[i] = e->acfg.clip_reward ? (float)((total > 0) - (total < 0)) : (float)total;
        e->adone[i] = (uint8_t)done;
        if (done) top_reset(&w);                                           /* obs = env.reset() */
        commit_obs(&w, done);
        wrap_close(&w);
    }
    pack_outputs(e);
    if (bad) e->pending_action_error = 1;
    return TBX_OK;
}

int tbx_agent_step_synthetic(tbx_engine* e, uint64_t seed, uint64_t t, uint64_t env_offset, void* stream)
{
    if (!e) return TBX_E_INVALID;
    int32_t* a = (int32_t*)malloc((size_t)e->n * 4);
    for (int i = 0; i < e->n; i++) a[i] = orc_synthetic_action(e->game, seed, env_offset + (uint64_t)i, t);
    int rc = tbx_agent_step_device(e, a, stream);
    free(a);
    return rc;
}

int tbx_agent_ring_head(tbx_engine* e, int32_t* out_head)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent_on) return fail(e, TBX_E_INVALID, "tbx_agent_init has not been called");
    if (!e->aring) return fail(e, TBX_E_INVALID, "tbx_agent_ring_head needs tbx_agent_config_t::new_plane = 2");
    if (!out_head) return fail(e, TBX_E_INVALID, "output pointer is NULL");
    *out_head = e->ahead;
    return TBX_OK;
}

int tbx_agent_step(tbx_engine* e, const int32_t* actions, float* reward, uint8_t* done, uint8_t* obs)
{
    if (e && e->agent_on && obs && !e->aobs) return agent_no_stack(e);
    int rc = tbx_agent_step_device(e, actions, NULL);
    if (rc) return rc;
    size_t n = (size_t)e->n;
    if (reward) memcpy(reward, e->areward, n * 4);
    if (done) memcpy(done, e->adone, n);
    if (obs) memcpy(obs, e->aobs, n * e->acfg.out_h * e->acfg.out_w * e->acfg.stack);
    return take_action_error(e);
}

/* ---------------------------------------------------------------- host delivery (toybox_amd.h: step_async / step_wait)
 * The oracle has no device and no stream: "_begin" carries the whole step out and fills the caller's buffers, "_end" returns
 * what the synchronous form returns.  Same calls, same order, same results as the HIP library's asynchronous forms. */

int tbx_host_alloc(void** out_ptr, size_t bytes)
{
    if (!out_ptr) return TBX_E_INVALID;
    void* p = NULL;
    if (posix_memalign(&p, 4096, bytes ? bytes : 1)) { *out_ptr = NULL; return TBX_E_NOMEM; }
    *out_ptr = p;
    return TBX_OK;
}

int tbx_host_free(void* ptr) { free(ptr); return TBX_OK; }

static int agent_copy_out(tbx_engine* e, const tbx_agent_host_out_t* o)
{
    const size_t n = (size_t)e->n, px = (size_t)e->acfg.out_h * e->acfg.out_w;
    if (o->plane && !agent_newest_plane(e)) return fail(e, TBX_E_INVALID, "the newest plane needs tbx_agent_config_t::new_plane = 1 or 2");
    if (o->obs && !e->aobs) return agent_no_stack(e);
    if (o->reward) memcpy(o->reward, e->areward, n * 4);
    if (o->done) memcpy(o->done, e->adone, n);
    if (o->ep_done) memcpy(o->ep_done, e->ep_done, n);
    if (o->ep_return) memcpy(o->ep_return, e->ep_ret_out, n * 4);
    if (o->ep_length) memcpy(o->ep_length, e->ep_len_out, n * 4);
    if (o->plane) memcpy(o->plane, agent_newest_plane(e), n * px);
    if (o->obs) memcpy(o->obs, e->aobs, n * px * e->acfg.stack);
    return TBX_OK;
}

int tbx_agent_step_begin(tbx_engine* e, const int32_t* actions, const tbx_agent_host_out_t* out)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent_on) return fail(e, TBX_E_INVALID, "tbx_agent_init has not been called");
    if (!actions || !out) return fail(e, TBX_E_INVALID, "actions / output descriptor is NULL");
    if (e->agent_host_pending || e->host_pending) return fail(e, TBX_E_INVALID, "tbx_agent_step_begin: the previous step has not been ended (tbx_agent_step_end / tbx_step_end)");
    if (out->plane && !agent_newest_plane(e)) return fail(e, TBX_E_INVALID, "the newest plane needs tbx_agent_config_t::new_plane = 1 or 2");
    if (out->obs && !e->aobs) return agent_no_stack(e);
    int rc = tbx_agent_step_device(e, actions, NULL);
    if (rc) return rc;
    rc = agent_copy_out(e, out);
    if (rc) return rc;
    e->agent_host_pending = 1;
    return TBX_OK;
}

int tbx_agent_step_end(tbx_engine* e)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent_on) return fail(e, TBX_E_INVALID, "tbx_agent_init has not been called");
    if (!e->agent_host_pending) return fail(e, TBX_E_INVALID, "tbx_agent_step_end without tbx_agent_step_begin");
    e->agent_host_pending = 0;
    return take_action_error(e);
}

int tbx_agent_fetch(tbx_engine* e, const tbx_agent_host_out_t* out)
{
    if (!e) return TBX_E_INVALID;
    if (!e->agent_on) return fail(e, TBX_E_INVALID, "tbx_agent_init has not been called");
    if (!out) return fail(e, TBX_E_INVALID, "output descriptor is NULL");
    return agent_copy_out(e, out);
}

int tbx_step_begin(tbx_engine* e, const int32_t* actions, uint32_t flags, const tbx_step_host_out_t* out)
{
    if (!e) return TBX_E_INVALID;
    if (!actions || !out) return fail(e, TBX_E_INVALID, "actions / output descriptor is NULL");
    if (e->host_pending || e->agent_host_pending) return fail(e, TBX_E_INVALID, "tbx_step_begin: the previous step has not been ended (tbx_step_end / tbx_agent_step_end)");
    if (out->frame && out->channels != 1 && out->channels != 3 && out->channels != 4) return fail(e, TBX_E_INVALID, "channels must be 1, 3 or 4");
    int rc = tbx_step_device(e, actions, flags, NULL);
    if (rc) return rc;
    const size_t n = (size_t)e->n;
    if (out->reward) memcpy(out->reward, e->reward, n * 4);
    if (out->done) memcpy(out->done, e->done, n);
    if (out->lives) memcpy(out->lives, e->lives, n * 4);
    if (out->score) memcpy(out->score, e->score, n * 4);
    if (out->frame) {
        rc = tbx_render_device(e, out->frame, out->channels, NULL);
        if (rc) return rc;
    }
    e->host_pending = 1;
    return TBX_OK;
}

int tbx_step_end(tbx_engine* e)
{
    if (!e) return TBX_E_INVALID;
    if (!e->host_pending) return fail(e, TBX_E_INVALID, "tbx_step_end without tbx_step_begin");
    e->host_pending = 0;
    return take_action_error(e);
}

/* VecFrameStack.step_wait / .reset on the host (vec_frame_stack.py:17-33), pixel by pixel as the Python's np.roll + slice writes
 * spell it; `threads` is ignored */
int tbx_host_stack_push(uint8_t* dst, const uint8_t* src, const uint8_t* plane, const uint8_t* done, int reset, int n, int px, int stack,
                        int fill, int threads)
{
    (void)threads;
    if (!dst || !src || !plane || n < 0 || px < 1 || stack < 1 || stack > 16 || fill < 0 || fill > 1) return TBX_E_INVALID;
    for (int i = 0; i < n; i++) {
        const int fresh = reset || (done && done[i]);
        for (int k = 0; k < px; k++) {
            uint8_t* d = dst + ((size_t)i * px + k) * stack;
            const uint8_t* s = src + ((size_t)i * px + k) * stack;
            const uint8_t v = plane[(size_t)i * px + k];
            for (int c = 0; c + 1 < stack; c++) d[c] = fresh ? (fill ? v : (uint8_t)0) : s[c + 1];   /* (dst may be src: ascending c) */
            d[stack - 1] = v;
        }
    }
    return TBX_OK;
}
