/*
 * toybox_amd.h -- C-ABI boundary of the MI355X-native batched Toybox engine.
 *
 * What this replaces.  In the reference (toybox-rs/Toybox) every game-step call goes
 * through the Python module `ctoybox` (cffi over the Rust cdylib of ctoybox==0.5.0,
 * /root/reference/REQUIREMENTS.txt:12), one opaque pointer per env, one FFI call per
 * operation.  The call sites that define the contract are cited per entry point below
 * (paths relative to /root/reference).  This header is the batched, device-resident
 * replacement of that per-env FFI: one engine handle owns N envs of one game on one GPU.
 *
 * Conventions (all entry points):
 *   - return int: 0 = TBX_OK, <0 = error (never abort / throw across the ABI);
 *     tbx_last_error() returns an engine-owned message valid until the next call;
 *   - plain pointers and sizes only; buffers are caller-allocated;
 *   - "_host" arguments are host pointers (call is synchronous), "_dev" arguments are
 *     device (HBM) pointers (call is asynchronous on the given hipStream_t, passed as void*);
 *   - calls on one handle take effect in program order whatever streams they name: when an entry point uses another stream
 *     than the previous one did (the host-pointer forms run on an engine-owned non-blocking stream), the new stream first
 *     waits for an event recorded on the old one -- no tbx_sync is needed between the two API families.  STREAM LIFETIME: to
 *     record that event the library keeps the handle of the stream the LAST call named until the next call on the handle;
 *     a stream passed to a call must therefore stay alive until the next call on the handle has returned, or until
 *     tbx_sync(), which forgets it (so: tbx_sync, then hipStreamDestroy).  The engine-owned output buffers (TBX_BUF_*) keep their addresses and a result in
 *     them stays valid for whatever the caller queues on the stream of the call that produced it before its next call on the
 *     handle -- unless the pipelined mode is switched on (TBX_OPT_PIPELINE below), which double-buffers them;
 *   - a handle is not thread-safe; different handles may be used concurrently;
 *   - there is NO CPU fallback: tbx_create fails with TBX_E_NO_DEVICE when no gfx950
 *     device is visible.
 *
 * The POD structs below are the lossless per-env state/config records that the Python
 * host turns into the interventions JSON schema
 * (toybox/interventions/{core,breakout,amidar,space_invaders}.py) and back.
 */
#ifndef TOYBOX_AMD_H
#define TOYBOX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ constants */

#define TBX_ABI_VERSION 1

#define TBX_GAME_BREAKOUT       0
#define TBX_GAME_AMIDAR         1
#define TBX_GAME_SPACE_INVADERS 2
#define TBX_GAME_GRIDWORLD      3   /* toybox/envs/atari/gridworld.py:8-13; not gym-registered by the reference */
#define TBX_NUM_GAMES           4

#define TBX_OK            0
#define TBX_E_INVALID    -1   /* bad argument (range, NULL, size mismatch)          */
#define TBX_E_NO_DEVICE  -2   /* no usable gfx950 device / HIP runtime error        */
#define TBX_E_NOMEM      -3
#define TBX_E_UNSUPPORTED -4  /* state/config outside what the device engine holds  */
#define TBX_E_ACTION     -5   /* an illegal ALE action id was seen (treated as NOOP) */
#define TBX_E_NEEDS_RESET -6  /* agent layer: an env was stepped although its game had ended inside EpisodicLifeEnv's no-op
                                 step -- where bench.Monitor raises "Tried to step environment that needs reset"
                                 (baselines/bench/monitor.py:52-53).  The step is carried out as the wrapper stack does
                                 without a Monitor (the finished game reports done at once and is reset). */

/* Input buttons bitmask == ctoybox.Input fields
 * (scripts/utils/test_games.py:13, test/interventions/test_breakout_interventions.py:12-15). */
#define TBX_BTN_LEFT    1u
#define TBX_BTN_RIGHT   2u
#define TBX_BTN_UP      4u
#define TBX_BTN_DOWN    8u
#define TBX_BTN_BUTTON1 16u
#define TBX_BTN_BUTTON2 32u

/* tbx_step flags */
#define TBX_STEP_AUTO_RESET 1u  /* VecEnv semantics: a done env is reset inside the step
                                   (baselines/common/vec_env/dummy_vec_env.py:51-54) */

/* device buffers addressable with tbx_device_buffer() */
#define TBX_BUF_REWARD  0   /* int32[N]  max(score - prev_score, 0) of the last step (envs/atari/base.py:136-138) */
#define TBX_BUF_DONE    1   /* uint8[N]  lives <= 0 after the last step (envs/atari/base.py:25-27,142)          */
#define TBX_BUF_LIVES   2   /* int32[N]  (pre-reset value when auto-reset fired)                                */
#define TBX_BUF_SCORE   3   /* int32[N]  (pre-reset value when auto-reset fired)                                */
#define TBX_BUF_FRAME   4   /* uint8[N,H,W,C] last tbx_render_device() into the engine-owned frame buffer       */
#define TBX_BUF_PACKED  5   /* uint64[N] {reward:i32, done:u8, lives:u8, pad:u16} record for the multi-GPU gather */

/* RGBA colour, memory order r,g,b,a (interventions/core.py:179-187) */
typedef struct tbx_color { uint8_t r, g, b, a; } tbx_color_t;

/* ------------------------------------------------------------------ Breakout POD */

#define TBX_BRK_MAX_BALLS    4
#define TBX_BRK_COLS         18
#define TBX_BRK_MAX_ROWS     14
#define TBX_BRK_MAX_BRICKS   256   /* >= 18 * 14 */
#define TBX_BRK_MAX_STARTS   8
#define TBX_BRK_MAX_SEGMENTS 16

/* Breakout world (pixels == world units).  Anchored on the golden dump
 * toybox/interventions/defaults/breakout_state_default.json: bricks at (12+12c, 43+4r),
 * paddle (120,143), 18 columns => field x in [12,228). */
#define TBX_BRK_W            240
#define TBX_BRK_H            160
#define TBX_BRK_LEFT         12.0
#define TBX_BRK_RIGHT        228.0
#define TBX_BRK_TOP          25.0
#define TBX_BRK_BOTTOM       160.0
#define TBX_BRK_WALL_Y0      13
#define TBX_BRK_BRICK_Y0     43.0
#define TBX_BRK_BRICK_W      12.0
#define TBX_BRK_BRICK_H      4.0
#define TBX_BRK_PADDLE_H     3.0

typedef struct tbx_breakout_config {
    uint64_t rand[2];                 /* simulator RNG == config_to_json()['rand']['state'] */
    int32_t  start_lives;
    int32_t  n_rows;                  /* len(row_scores) == len(row_colors) */
    int32_t  row_scores[TBX_BRK_MAX_ROWS];
    tbx_color_t row_colors[TBX_BRK_MAX_ROWS];
    int32_t  ball_speed_row_depth;
    int32_t  n_starts;
    double   ball_speed_slow, ball_speed_fast;
    double   start_x[TBX_BRK_MAX_STARTS], start_y[TBX_BRK_MAX_STARTS];
    double   start_angle_deg[TBX_BRK_MAX_STARTS];
    /* host-evaluated trig (libm), so that no transcendental runs on the device:
       start_dir = (cos a, sin a) of start_angle_deg */
    double   start_dir_x[TBX_BRK_MAX_STARTS], start_dir_y[TBX_BRK_MAX_STARTS];
    int32_t  paddle_discrete_segments;   /* 1..16 */
    int32_t  _pad0;
    /* paddle_dir[i] = (cos a_i, -sin a_i), a_i = 150 - i*120/(S-1) degrees (S==1: 90) */
    double   paddle_dir_x[TBX_BRK_MAX_SEGMENTS], paddle_dir_y[TBX_BRK_MAX_SEGMENTS];
    tbx_color_t bg_color, frame_color, paddle_color, ball_color;
} tbx_breakout_config_t;

typedef struct tbx_brick {
    double   x, y, w, h;              /* position, size */
    int32_t  points, depth, row, col;
    tbx_color_t color;
    uint8_t  alive, destructible, _pad[2];
} tbx_brick_t;

typedef struct tbx_breakout_state {
    uint64_t rand[2];
    int32_t  score, lives, level;
    uint8_t  is_dead, reset, _pad0[2];
    double   paddle_x, paddle_y, paddle_vx, paddle_vy;
    double   paddle_width, paddle_speed, ball_radius;
    int32_t  n_balls, n_bricks;
    double   ball_x[TBX_BRK_MAX_BALLS], ball_y[TBX_BRK_MAX_BALLS];
    double   ball_vx[TBX_BRK_MAX_BALLS], ball_vy[TBX_BRK_MAX_BALLS];
    tbx_brick_t bricks[TBX_BRK_MAX_BRICKS];
} tbx_breakout_state_t;

/* ------------------------------------------------------------------ SpaceInvaders POD */

#define TBX_SI_W            320
#define TBX_SI_H            210
#define TBX_SI_COLS         6
#define TBX_SI_MAX_ROWS     10
#define TBX_SI_MAX_ENEMIES  64    /* one lane per enemy; 6 x 10 = 60 */
#define TBX_SI_MAX_SHIELDS  3
#define TBX_SI_SHIELD_W     16
#define TBX_SI_SHIELD_H     18
#define TBX_SI_MAX_LASERS   8

/* interventions/core.py:125-135 Direction.directions order */
#define TBX_DIR_UP    0
#define TBX_DIR_DOWN  1
#define TBX_DIR_LEFT  2
#define TBX_DIR_RIGHT 3

typedef struct tbx_si_config {
    uint64_t rand[2];
    double   jitter;                  /* probability of a random (vs player-targeted) firing column */
    int32_t  start_lives, n_rows, n_shields;
    int32_t  enemy_protocol;          /* 0 = "TargetPlayer" */
    int32_t  row_scores[TBX_SI_MAX_ROWS];
    int32_t  shield_x[TBX_SI_MAX_SHIELDS], shield_y[TBX_SI_MAX_SHIELDS];
} tbx_si_config_t;

typedef struct tbx_si_laser {
    int32_t  x, y, w, h, t, movement, speed;
    tbx_color_t color;
} tbx_si_laser_t;

typedef struct tbx_si_enemy {
    int32_t  x, y, row, col, id, points;
    int32_t  death_counter;           /* -1 == None */
    uint8_t  alive, _pad[3];
} tbx_si_enemy_t;

typedef struct tbx_si_state {
    uint64_t rand[2];
    int32_t  score, lives, level;
    int32_t  life_display_timer, enemy_shot_delay;
    int32_t  n_enemies, n_enemy_lasers, has_ship_laser;
    int32_t  ship_x, ship_y, ship_w, ship_h, ship_speed;
    int32_t  ship_death_counter;      /* -1 == None */
    tbx_color_t ship_color;
    uint8_t  ship_alive, ship_death_hit_1, _pad0[2];
    int32_t  ufo_x, ufo_y, ufo_appearance_counter;
    int32_t  ufo_death_counter;       /* -1 == None */
    int32_t  move_counter, move_dir;
    uint8_t  visual_orientation, _pad1[3];
    int32_t  n_shields;
    int32_t  shield_x[TBX_SI_MAX_SHIELDS], shield_y[TBX_SI_MAX_SHIELDS];
    tbx_color_t shield_color[TBX_SI_MAX_SHIELDS];
    uint16_t shield_rows[TBX_SI_MAX_SHIELDS][TBX_SI_SHIELD_H];   /* bit c of row r = pixel (c,r) present */
    tbx_si_laser_t ship_laser;
    tbx_si_laser_t enemy_lasers[TBX_SI_MAX_LASERS];
    tbx_si_enemy_t enemies[TBX_SI_MAX_ENEMIES];
} tbx_si_state_t;

/* ------------------------------------------------------------------ Amidar POD */

#define TBX_AMI_W             160
#define TBX_AMI_H             250
#define TBX_AMI_BOARD_W       32
#define TBX_AMI_BOARD_H       31
#define TBX_AMI_MAX_ENEMIES   8
#define TBX_AMI_MAX_BOXES     64
#define TBX_AMI_MAX_HISTORY   16
#define TBX_AMI_MAX_CHASE_J   8
#define TBX_AMI_TILE_WX       64    /* world units per tile: player_start {31,15} <-> position {1984,1200} in the goldens */
#define TBX_AMI_TILE_WY       80

/* tile tags (interventions/amidar.py:55-59) */
#define TBX_TILE_EMPTY        0
#define TBX_TILE_UNPAINTED    1
#define TBX_TILE_PAINTED      2
#define TBX_TILE_CHASE_MARKER 3

/* movement protocols (interventions/amidar.py:101-112) */
#define TBX_AI_PLAYER          0
#define TBX_AI_LOOKUP          1   /* EnemyLookupAI     {next, default_route_index}                       */
#define TBX_AI_PERIMETER       2   /* EnemyPerimeterAI  {start}                                           */
#define TBX_AI_AMIDAR          3   /* EnemyAmidarMvmt   {vert, horiz, start_vert, start_horiz, start}     */
#define TBX_AI_TARGET_PLAYER   4   /* EnemyTargetPlayer {start, start_dir, vision_distance, dir, player_seen} */
#define TBX_AI_RANDOM          5   /* EnemyRandomMvmt   {start, start_dir, dir}                           */

typedef struct tbx_amidar_ai {
    int32_t kind;
    int32_t next, default_route_index;
    int32_t start_tx, start_ty;
    int32_t vert, horiz, start_vert, start_horiz;
    int32_t start_dir, dir;
    int32_t vision_distance;
    int32_t seen_tx, seen_ty;          /* player_seen; -1,-1 == None */
} tbx_amidar_ai_t;

typedef struct tbx_amidar_mover {
    int32_t x, y, speed;
    int32_t step_tx, step_ty;          /* step; -1,-1 == None */
    int32_t n_history;
    int32_t history[TBX_AMI_MAX_HISTORY];
    int32_t caught;
    tbx_amidar_ai_t ai;
} tbx_amidar_mover_t;

typedef struct tbx_amidar_box {
    int32_t tl_tx, tl_ty, br_tx, br_ty;
    uint8_t painted, triggers_chase, _pad[2];
} tbx_amidar_box_t;

typedef struct tbx_amidar_config {
    uint64_t rand[2];
    int32_t  start_lives, start_jumps, jump_time, chase_time, box_bonus, chase_score_bonus;
    int32_t  player_start_tx, player_start_ty;
    int32_t  n_enemies;
    uint8_t  render_images, default_board_bugs, _pad0[2];
    tbx_amidar_ai_t enemies[TBX_AMI_MAX_ENEMIES];
    tbx_color_t bg_color, player_color, unpainted_color, painted_color, enemy_color, inner_painted_color;
    uint8_t  board[TBX_AMI_BOARD_H][TBX_AMI_BOARD_W];   /* tile tags */
} tbx_amidar_config_t;

typedef struct tbx_amidar_state {
    uint64_t rand[2];
    int32_t  score, lives, level;
    int32_t  jumps, jump_timer, chase_timer;
    int32_t  n_enemies, n_boxes, n_chase_junctions;
    int32_t  chase_junctions[TBX_AMI_MAX_CHASE_J];
    tbx_amidar_mover_t player;
    tbx_amidar_mover_t enemies[TBX_AMI_MAX_ENEMIES];
    tbx_amidar_box_t boxes[TBX_AMI_MAX_BOXES];
    uint8_t  tiles[TBX_AMI_BOARD_H][TBX_AMI_BOARD_W];
} tbx_amidar_state_t;

/* ------------------------------------------------------------------ GridWorld POD */

/* The reference only ships this game's two golden dumps (toybox/interventions/defaults/gridworld_{config,state}_default.json)
 * and the env class envs/atari/gridworld.py:8-13; the records below hold exactly the fields of those dumps.  The rules
 * and the picture are this repo's own (SPEC.md, "GridWorld"): a fixed 160x128 frame divided into
 * game_size cells of floor(160/w) x floor(128/h) pixels. */
#define TBX_GW_W          160
#define TBX_GW_H          128
#define TBX_GW_MAX_DIM    32    /* game_size <= 32 x 32; grid rows have a stride of 32 cells */
#define TBX_GW_MAX_TILES  16

typedef struct tbx_gw_tile {
    tbx_color_t color;
    int32_t  reward;
    uint8_t  goal, walkable, _pad[2];
} tbx_gw_tile_t;

typedef struct tbx_gridworld_config {
    uint64_t rand[2];                 /* simulator RNG words (kept for a uniform seeding surface; the rules draw nothing) */
    int32_t  width, height;           /* "game_size" */
    int32_t  n_tiles;
    int32_t  player_start_x, player_start_y;
    int32_t  reward_becomes;          /* tile index a collected reward cell turns into */
    tbx_color_t player_color;
    uint8_t  tile_keys[TBX_GW_MAX_TILES];   /* the one-character names the JSON config gives the tiles */
    tbx_gw_tile_t tiles[TBX_GW_MAX_TILES];
    uint8_t  grid[TBX_GW_MAX_DIM * TBX_GW_MAX_DIM];   /* tile index of cell (x, y) at [y * 32 + x] */
} tbx_gridworld_config_t;

typedef struct tbx_gridworld_state {
    int32_t  score, game_over;
    int32_t  player_x, player_y;
    int32_t  reward_becomes;
    int32_t  width, height, n_tiles;
    tbx_color_t player_color;
    tbx_gw_tile_t tiles[TBX_GW_MAX_TILES];
    uint8_t  grid[TBX_GW_MAX_DIM * TBX_GW_MAX_DIM];
} tbx_gridworld_state_t;

/* ------------------------------------------------------------------ engine */

typedef struct tbx_engine tbx_engine;

/* ABI version of the loaded library. */
int tbx_abi_version(void);

/* Last error text; engine may be NULL (errors of tbx_create). */
const char* tbx_last_error(const tbx_engine* engine);

/* Static game metadata.
 * replaces Toybox.get_height/get_width (envs/atari/base.py:64-65) and
 * Toybox.get_legal_action_set (envs/atari/base.py:57, test/benchmark.py:48). */
int tbx_frame_dims(int game, int* height, int* width);
int tbx_legal_actions(int game, int32_t* out_actions, int cap);   /* returns count or <0 */
/* ALE action id (envs/atari/constants.py:16-35) -> TBX_BTN_* mask; <0 if out of 0..17. */
int tbx_ale_action_to_buttons(int ale_action);
size_t tbx_state_size(int game);
size_t tbx_config_size(int game);

/* Which device an engine drives: one process per GPU shards the batch (cmd_util.py:31 gives every worker its own rank and seed),
 * and a benchmark line has to show that N ranks drove N DISTINCT GPUs -- bench.py gathers one of these per rank and refuses
 * (rc 7) when two ranks of an RCCL run report the same PCI address. */
typedef struct tbx_device_identity {
    int32_t  ordinal;                 /* HIP device ordinal inside this process (HIP_VISIBLE_DEVICES applied) */
    int32_t  pci_domain, pci_bus, pci_device;
    uint64_t total_memory;            /* bytes of device memory */
    int32_t  compute_units;
    int32_t  _pad;
    char     arch[64];                /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
    char     name[64];                /* marketing name */
} tbx_device_identity_t;
int tbx_device_identity(tbx_engine* engine, tbx_device_identity_t* out);

/* Create an engine of n_envs envs of one game on HIP device `device`.
 * config_pod may be NULL (game defaults == interventions/defaults/<game>_config_default.json).
 * Every env's simulator RNG starts at config.rand; all envs are then given a first game.
 * replaces ctoybox.Toybox(name) (envs/atari/breakout.py:8, interventions/breakout.py:38). */
int tbx_create(int game, int n_envs, int device, const void* config_pod, size_t config_size,
               tbx_engine** out_engine);
int tbx_destroy(tbx_engine* engine);
int tbx_num_envs(const tbx_engine* engine);
int tbx_game(const tbx_engine* engine);

/* Re-seed the simulator RNG: env >= 0 -> that env gets `seed`; env == -1 -> env i gets seed+i.
 * Takes effect at the next new game, as in the reference (envs/atari/base.py:95-97).
 * replaces Toybox.set_seed (envs/atari/base.py:95, scripts/utils/test_games.py:30). */
int tbx_seed(tbx_engine* engine, int env, uint32_t seed);
/* Per-env seeds in one upload + one launch: env i gets seeds_host[i].  This is how a vectorised env applies the reference's
 * per-worker derivation seed2_i = hash_seed(seed + rank_i + 1) % 2**31 (envs/atari/base.py:84-98 under
 * baselines/common/cmd_util.py:31) without N calls. */
int tbx_seed_array(tbx_engine* engine, const uint32_t* seeds_host);
int tbx_get_sim_rng(tbx_engine* engine, int env, uint64_t out_state[2]);
int tbx_set_sim_rng(tbx_engine* engine, int env, const uint64_t state[2]);

/* Start a new game in the envs whose mask byte is non-zero (mask_host == NULL: all).
 * replaces Toybox.new_game (envs/atari/base.py:97,153; test/benchmark.py:54). */
int tbx_new_game(tbx_engine* engine, const uint8_t* mask_host);

/* One game frame for every env.  actions are ALE action ids.  Host-pointer form: synchronous,
 * any output pointer may be NULL.
 * replaces Toybox.apply_ale_action + get_score/get_lives/game_over
 * (envs/atari/base.py:126,136,142,145; test/benchmark.py:52-56). */
int tbx_step(tbx_engine* engine, const int32_t* ale_actions_host, uint32_t flags,
             int32_t* reward_host, uint8_t* done_host, int32_t* lives_host, int32_t* score_host);
/* Device-pointer form: asynchronous on `stream`; results land in the TBX_BUF_* buffers. */
int tbx_step_device(tbx_engine* engine, const int32_t* ale_actions_dev, uint32_t flags, void* stream);
/* Same, with the actions generated on the device: env e at time t plays
 * legal[ splitmix64(action_seed ^ (e_global << 32) ^ t) mod n_legal ], e_global = env_offset + e.
 * (Such a step depends on nothing a rasteriser produces: with TBX_OPT_PIPELINE it runs beside the previous frame's
 * tbx_render_device.) */
int tbx_step_synthetic(tbx_engine* engine, uint64_t action_seed, uint64_t t, uint64_t env_offset,
                       uint32_t flags, void* stream);
/* One frame for ONE env by ALE action id, with the outputs of tbx_step: out[4] = {reward, done, lives, score} (may be NULL).
 * replaces the reference's per-frame call sequence Toybox.apply_ale_action + get_score / get_lives / game_over on a single
 * env (envs/atari/base.py:126-145, test/benchmark.py:50-56).  On a ONE-env engine this call does not launch anything: a
 * resident kernel (one wave) waits on a mailbox in host-coherent pinned memory, steps env 0 and posts the outputs back --
 * two PCIe hops per frame.  The wave leaves after 50 ms without a request and every other entry point of the handle stops it
 * first, so the two never run side by side.  On batch engines it is tbx_apply_input plus a read-back of the outputs
 * (TBX_OPT_RESIDENT_STEP = 0 forces that second form everywhere). */
int tbx_step1(tbx_engine* engine, int env, int32_t ale_action, uint32_t flags, int32_t out[4]);
/* The same step plus the env's picture, i.e. the whole of ToyboxBaseEnv.step (apply_ale_action, then get_state / get_rgb_frame:
 * envs/atari/base.py:126,109,131-145; the env.step() arm of test/benchmark.py:83-97).  *frame_host receives the address of an
 * engine-owned buffer of H * W * channels bytes (channels 1, 3 or 4) that holds the frame of the state AFTER the step and stays
 * valid until the next call on the handle.  On a one-env engine the resident kernel rasterises straight into that buffer
 * (pinned host memory mapped into the device): still no launch, no copy, no synchronisation.  Elsewhere it is tbx_step1
 * followed by tbx_render_env into the same buffer. */
int tbx_step1_frame(tbx_engine* engine, int env, int32_t ale_action, uint32_t flags, int channels, int32_t out[4],
                    const uint8_t** frame_host);
/* One frame for one env with a raw button mask.
 * replaces Toybox.apply_action(Input) (scripts/utils/test_games.py:13). */
int tbx_apply_input(tbx_engine* engine, int env, uint32_t buttons);

/* Scalar reads for every env (any pointer may be NULL).
 * replaces get_score/get_lives/get_level/game_over (envs/atari/base.py:20-27). */
int tbx_get_scalars(tbx_engine* engine, int32_t* score_host, int32_t* lives_host,
                    int32_t* level_host, uint8_t* game_over_host);

/* Rasterise every env's current state: out[N][H][W][channels], channels 1 (gray), 3 (RGB), 4 (RGBA).
 * replaces Toybox.get_state / get_rgb_frame (envs/atari/base.py:109,164). */
int tbx_render(tbx_engine* engine, uint8_t* out_host, int channels);
/* out_dev == NULL renders into the engine-owned TBX_BUF_FRAME buffer. */
int tbx_render_device(tbx_engine* engine, uint8_t* out_dev, int channels, void* stream);
/* The random-rollout loop body as ONE call: rasterise every env's CURRENT state into out_dev (NULL: TBX_BUF_FRAME) and step
 * every env one frame with device-generated actions (the rule of tbx_step_synthetic) -- i.e. tbx_render_device followed by
 * tbx_step_synthetic, asynchronous on `stream`, same results bit for bit: the frame shows the state BEFORE the step, the
 * TBX_BUF_* outputs and the state are the step's.  Where the rasteriser reads step-written render records (Breakout with the
 * canonical wall; RGB / RGBA) both halves are ONE launch: the step's few blocks ride in front of the rasteriser's, write the
 * other records buffer and hide in the launch's ramp-up, so a loop of these calls runs like a render-only loop -- no kernel
 * boundary per frame, no rasteriser that starts in lockstep behind a short kernel (8 192 envs + gather, the per-GPU share of
 * the strong-scaled headline batch: see DESIGN.md section 6).  Only a loop whose actions do not depend on the frame can use
 * it (the north star's random-action rollouts; replaces the loop body of test/benchmark.py:50-56 plus the frame);
 * a policy-driven loop calls tbx_step_device and tbx_render_device.  Everywhere else it is the two launches in stream order. */
int tbx_render_step_synthetic(tbx_engine* engine, uint8_t* out_dev, int channels, uint64_t action_seed, uint64_t t,
                              uint64_t env_offset, uint32_t flags, void* stream);
/* A rollout CHUNK: k consecutive tbx_render_step_synthetic calls (t = t0 .. t0 + k - 1; with a K-step record ring each followed by
 * tbx_gather) as one call -- what the reference's learners consume is a K-step rollout anyway (A2C nsteps = 5, PPO2 128:
 * baselines/baselines/a2c/runner.py:16, ppo2/ppo2.py:103).  Results bit for bit those of the k single calls:
 *   TBX_BUF_ROLLOUT_FRAMES  uint8[k][N][H][W][channels]: frame j shows the state BEFORE step t0 + j;
 *   TBX_BUF_ROLLOUT_PACKED  uint64[k][stride] step records (reward, done, lives) of step j in row j -- with a K-step record ring
 *                           in force (TBX_OPT_GATHER_EVERY = K; k must equal K and the ring must be empty) the rows ARE the ring
 *                           (stride = records_per_rank) and the call queues the collective itself: no tbx_gather for these steps;
 *                           otherwise an engine-owned array with stride = N;
 *   TBX_BUF_REWARD / DONE / LIVES / SCORE / PACKED and the state: those of the chunk's LAST step; TBX_BUF_FRAME: its last frame.
 * Where the rasteriser reads step-written records (Breakout, canonical wall, RGB / RGBA; TBX_OPT_ROLLOUT_CHUNKS) a chunk is ONE step
 * launch on an internal stream -- every env k frames with its state in registers, writing k render records, the k step records and
 * the state once -- and rasteriser launches that depend on that step launch alone, in one of two forms: ONE launch for the chunk's
 * k x N frames (the records and the frames of a chunk lie one behind the other), chunk behind chunk on a second internal stream --
 * a render-only loop of launches k times as long as a frame's --, or k launches of one frame each that alternate between two
 * internal streams and overlap freely.  Either way the next chunk's step launch runs beside this chunk's rasterisers (a whole
 * chunk ahead of its own), and with a ring the collective waits for the step launch only.  That is what N
 * worker processes stepping independently of each other give the reference (baselines/baselines/common/vec_env/subproc_vec_env.py:49-74),
 * and what the per-GPU share of a strong-scaled batch (8 192 envs) loses to ramp-up and tail between launches in stream order.
 * Contract in that form: as for overlapped fused launches (TBX_OPT_FUSED_OVERLAP) -- two chunk buffers alternate; ask
 * tbx_device_buffer after the call for what is to be read, and it is that call which makes `stream` wait for the chunk (lazy
 * join); results stay valid for readers queued on `stream` before the next chunk.  Everywhere else (other games, gray frames,
 * one collective per step, the option off) the call is the k single calls in stream order. */
int tbx_rollout_synthetic(tbx_engine* engine, int channels, uint64_t action_seed, uint64_t t0, int k, uint64_t env_offset,
                          uint32_t flags, void* stream);
#define TBX_BUF_ROLLOUT_FRAMES 15
#define TBX_BUF_ROLLOUT_PACKED 16
/* Rasterise one env (host pointer, synchronous). */
int tbx_render_env(tbx_engine* engine, int env, uint8_t* out_host, int channels);

/* Per-env lossless state record (tbx_<game>_state_t).
 * replaces Toybox.to_state_json / write_state_json (interventions/base.py:391,406).
 * tbx_set_state stores the CANONICAL form of the record and tbx_get_state returns it: slots beyond the counts (balls,
 * bricks, enemies, lasers, shields, boxes) zeroed, flags 0 / 1, an absent counter -1, direction fields two bits wide,
 * Amidar tile tags two bits wide, paddings zero.  Records a game produces itself are canonical already.
 * Capacity: counts beyond the fixed tables (TBX_*_MAX_*) and SpaceInvaders enemies whose row / col leave 0..255 or whose id
 * leaves 0..65535 (the device packs the three into one word) are refused with TBX_E_UNSUPPORTED, never truncated. */
int tbx_get_state(tbx_engine* engine, int env, void* pod_out, size_t size);
int tbx_set_state(tbx_engine* engine, int env, const void* pod, size_t size);
/* Batched forms for intervention sweeps over many envs: `count` consecutive records of
 * record_size bytes for envs [first_env, first_env + count), one pack/unpack launch and one copy. */
int tbx_get_states(tbx_engine* engine, int first_env, int count, void* pods_out, size_t record_size);
int tbx_set_states(tbx_engine* engine, int first_env, int count, const void* pods, size_t record_size);
/* Batch-wide config record (tbx_<game>_config_t).  `rand` is env 0's simulator RNG on get.  On set, a `rand` equal to
 * env 0's current simulator RNG (i.e. the caller did not edit it) leaves every env's simulator RNG untouched -- a config
 * edit must not put a seeded batch onto one random stream; any other value is written to every env.  For a one-env engine
 * both cases coincide with the reference.  Does not start a new game.
 * replaces Toybox.config_to_json / write_config_json (interventions/base.py:390,402). */
int tbx_get_config(tbx_engine* engine, void* pod_out, size_t size);
int tbx_set_config(tbx_engine* engine, const void* pod, size_t size);

/* Integer state queries.  Amidar: TBX_QUERY_TILE_TO_WORLD {tx,ty} -> {x,y}, TBX_QUERY_WORLD_TO_TILE {x,y} -> {tx,ty}.
 * replaces Toybox.query_state_json('tile_to_world' / 'world_to_tile') (interventions/amidar.py:510,518). */
#define TBX_QUERY_TILE_TO_WORLD 1
#define TBX_QUERY_WORLD_TO_TILE 2
int tbx_query(tbx_engine* engine, int env, int query_id, const int32_t* args, int n_args, int32_t* out, int n_out);

/* ------------------------------------------------------------------ batched interventions on the device
 * The reference's per-game intervention classes are helper methods over ONE env's decoded JSON state
 * (toybox/interventions/breakout.py:303-429, amidar.py:360-615, space_invaders.py:165-176).  Their batched forms here act on
 * the struct-of-arrays state in HBM directly -- one kernel over every selected env, no state record leaves the device:
 *   tbx_edit    a field write ("add_channel", "set_mode", "set_tile_tag", "set_enemy_protocol", ...) in the envs whose mask
 *               byte is non-zero (mask NULL: every env);
 *   tbx_reduce  a per-env feature ("num_bricks_remaining", "channel_count", "player_enemy_distances", ...) as
 *               out[N][tbx_reduce_width(game, query)] doubles (integers are exact in binary64; Breakout's positions are
 *               binary64 anyway); entries that do not exist (a fifth enemy of four, a tile outside the board) read -1.
 * args: up to TBX_EDIT_MAX_ARGS doubles, the same for every env (per_env = 0) or one row per env, args[N][n_args]
 * (per_env = 1: a sweep in which env i gets its own column, tile, timer ...).  The "_device" forms take device pointers for
 * mask / per-env args / out and are asynchronous on `stream`; args with per_env = 0 are always a host pointer.
 * Helpers that take a Python predicate have a MASK form here: find_brick(pred) -> TBX_QUERY_BRK_FIND_BRICK over a bit mask of
 * brick indices (the host evaluates the predicate once over the brick table, whose attributes other than `alive` are the same
 * in every env) plus the wanted `alive`; filter_tiles(pred) for predicates on the tag -> TBX_QUERY_AMI_TILES_MASK over a mask
 * of tags.  Helpers that sample with Python's `random` (get_random_tile, get_random_track_position, set_player_random_start,
 * get_random_dir_for_tile: amidar.py:360-399,541-583) have COUNTER-RNG forms: draw number k of env e is
 * splitmix64(seed ^ (e_global << 32) ^ k) -- the rule of tbx_step_synthetic -- taken modulo the number of candidates, and
 * the candidates are exactly the elements the reference's predicate accepts, in the order its loops visit them (tiles row by
 * row, directions Up, Down, Left, Right).  Not `random`-compatible (the reference's draws depend on Python's global
 * Mersenne Twister and on rejection sampling); uniform over the same candidate set, reproducible, and independent of the batch
 * layout.  Config-level helpers (set_jitter, add_row) go through tbx_set_config like the reference's dirty_config path. */
#define TBX_EDIT_MAX_ARGS 16
/* every game -- `intervention.game.lives = v` (interventions/space_invaders.py:199, test_breakout_interventions.py) */
#define TBX_EDIT_SET_LIVES          1   /* {lives} */
#define TBX_EDIT_SET_SCORE          2   /* {score} */
#define TBX_EDIT_SET_LEVEL          3   /* {level} */
/* Breakout (interventions/breakout.py) */
#define TBX_EDIT_BRK_COLUMN_ALIVE  10   /* {col, alive}: add_channel :392-396 (alive 0), fill_column :398-402 (alive 1) */
#define TBX_EDIT_BRK_ROW_ALIVE     11   /* {row, alive} */
#define TBX_EDIT_BRK_ALL_ALIVE     12   /* {alive}: clear_board :412-415 (alive 0) */
#define TBX_EDIT_BRK_BRICK_ALIVE   13   /* {brick index, alive} (test_breakout_interventions.py:48) */
#define TBX_EDIT_BRK_PADDLE        14   /* {x[, y]} paddle.position (get_paddle_position :379; test :111-118) */
#define TBX_EDIT_BRK_BALL          15   /* {ball, x, y, vx, vy} balls[ball] position and velocity (:365-377) */
/* Amidar (interventions/amidar.py) */
#define TBX_EDIT_AMI_TIMERS        20   /* {jump_timer, chase_timer}, -1 = leave: set_mode :402-416 */
#define TBX_EDIT_AMI_JUMPS         21   /* {jumps} (test_amidar_interventions.py:174) */
#define TBX_EDIT_AMI_TILE          22   /* {tx, ty, tag}: set_tile_tag :476-478 */
#define TBX_EDIT_AMI_ENEMY_AI      23   /* {enemy, the 14 fields of tbx_amidar_ai_t}: set_enemy_protocol :418-471 */
#define TBX_EDIT_AMI_PLAYER_TILE   24   /* {tx, ty}: player.position = tile_to_world(tile), the write of set_player_random_start :531-538 */
#define TBX_EDIT_AMI_PLAYER_RANDOM_START 25  /* {seed, draw, env_offset, min_enemy_distance}: set_player_random_start :541-548 -- the player goes
                                              * to the tile TBX_QUERY_AMI_RANDOM_TILE {seed, draw, env_offset, 15, min_enemy_distance} names
                                              * (ANY tag, as the reference draws it; no candidate: the env is left alone) */
/* SpaceInvaders (interventions/space_invaders.py) */
#define TBX_EDIT_SI_UFO_APPEARANCE 30   /* {appearance_counter}: remove_mothership :172-173 (-1) */

#define TBX_QUERY_BRK_BRICKS_REMAINING 110  /* -> 1  num_bricks_remaining :309-310 */
#define TBX_QUERY_BRK_NUM_BRICKS       111  /* -> 1  num_bricks :312-313 */
#define TBX_QUERY_BRK_COLUMN           112  /* {col} -> 32  alive flag of each brick with that col, in brick order (get_column :349-355) */
#define TBX_QUERY_BRK_ROW              113  /* {row} -> 32  the same by row (get_row :357-359, read as documented: the ith row) */
#define TBX_QUERY_BRK_IS_CHANNEL       114  /* {col} -> 1  the column has bricks and none is alive (is_channel :340-347) */
#define TBX_QUERY_BRK_CHANNEL_COUNT    115  /* {n_columns} -> 1  channel_count :361-366 */
#define TBX_QUERY_BRK_FIND_CHANNEL     116  /* {n_columns} -> 1  first channel column or -1 (find_channel :404-410) */
#define TBX_QUERY_BRK_PADDLE           117  /* -> 4  position x, y, velocity x, y (:379-383) */
#define TBX_QUERY_BRK_BALLS            118  /* -> 17 n_balls, x[4], y[4], vx[4], vy[4] (:365-377) */
#define TBX_QUERY_BRK_FIND_BRICK       119  /* {alive (-1: either), m0 .. m7} -> 1  index of the first brick, in brick order, whose bit is set in the
                                             * mask (brick i = bit i % 32 of m[i / 32], each m an integer < 2^32) and whose alive flag is the one
                                             * asked for, else -1 (find_brick :400-404 for predicates the host has turned into a mask) */
#define TBX_QUERY_AMI_MODE             120  /* -> 2  jump_timer, chase_timer (get_regular / jump / chase_mode :385-395) */
#define TBX_QUERY_AMI_ANY_CAUGHT       121  /* -> 1  any_enemy_caught :397-399 */
#define TBX_QUERY_AMI_TILE             122  /* {tx, ty} -> 1  tag (get_tile_by_pos :480-481, is_tile_walkable :472-474) */
#define TBX_QUERY_AMI_COUNT_TILES      123  /* {tag} -> 1  len(filter_tiles(tag == ...)) :483-488 */
#define TBX_QUERY_AMI_ADJACENT         124  /* {tx, ty} -> 4  tags of the up, left, right, down neighbours -- the order filter_tiles finds them (get_adjacent_tiles :512-524) */
#define TBX_QUERY_AMI_ENEMY_DISTANCES  125  /* {tx, ty} -> 8  manhattan tile distance of every enemy (enemy_distances_from_tile :526-530) */
#define TBX_QUERY_AMI_PLAYER_TILE      126  /* -> 3  tx, ty, tag (player_tile :573-576) */
#define TBX_QUERY_AMI_PLAYER_ENEMY_DISTANCES 127  /* -> 8  player_enemy_distances :579-583 */
#define TBX_QUERY_AMI_PLAYER_ON_PAINTED 128 /* -> 1  player_on_painted :586-589 */
#define TBX_QUERY_AMI_PLAYER_NEAR_UNPAINTED 129 /* {radius} -> 1  player_near_unpainted :592-603 */
#define TBX_QUERY_SI_SHIP              130  /* -> 8  x, y, w, h, speed, alive, death_counter, death_hit_1 (get_player :175-176) */
#define TBX_QUERY_AMI_TILES_MASK       131  /* {tag_mask} -> 32  filter_tiles(pred on the tag) :494-499: [ty] = bit tx set where tile (tx, ty)'s tag is in
                                             * tag_mask (bit t = tag t), ty = 0 .. 30; [31] = their number */
#define TBX_QUERY_AMI_RANDOM_TILE      133  /* {seed, draw, env_offset, tag_mask, min_enemy_distance} -> 4  tx, ty, tag, number of candidates: the
                                             * (r mod count)-th, row by row, of the tiles whose tag is in tag_mask and -- min_enemy_distance > 0 --
                                             * for which NOT every enemy is nearer than that (the predicate of set_player_random_start :543-546
                                             * as written); get_random_tile :360-378, get_random_track_position :380-386 (tag_mask 14); -1 x 3, 0 without a candidate */
#define TBX_QUERY_AMI_RANDOM_DIR       134  /* {seed, draw, env_offset, tx, ty} -> 2  direction (TBX_DIR_*) drawn among those of Up, Down, Left, Right whose
                                             * neighbour tile is walkable, and their number (get_random_dir_for_tile :550-583); -1, 0 if none */
int tbx_reduce_width(int game, int query);   /* doubles per env, or TBX_E_INVALID */
int tbx_edit(tbx_engine* engine, int op, const double* args_host, int n_args, int per_env, const uint8_t* mask_host);
int tbx_edit_device(tbx_engine* engine, int op, const double* args, int n_args, int per_env, const uint8_t* mask_dev, void* stream);
int tbx_reduce(tbx_engine* engine, int query, const double* args_host, int n_args, int per_env, double* out_host);
int tbx_reduce_device(tbx_engine* engine, int query, const double* args, int n_args, int per_env, double* out_dev, void* stream);

/* ------------------------------------------------------------------ agent-side preprocessing (SURVEY.md 8f rank 1 + 2)
 * The per-env wrapper stack that the reference's vendored baselines put between the env and the learner, fused on the
 * device so that only the small stacked observation ever leaves the chip.  The composition is the reference's, class by
 * class, with each class's state kept per env (baselines/baselines/common/...):
 *   NoopResetEnv(noop_max) atari_wrappers.py:108-135  1..noop_max no-op frames after a real reset (a game that ends inside them
 *                          is started again); the observation of such a reset is the raw frame after the last no-op
 *   MaxAndSkipEnv(skip)    atari_wrappers.py:193-219  repeat the action `skip` frames until the game ends, sum the rewards;
 *                          observation = per-pixel max of a PERSISTENT two-frame buffer that frame skip-2 and frame skip-1
 *                          write (zero frames at construction, never cleared by a reset: a step cut short by the end of
 *                          the game leaves the slots it did not reach as they were)
 *   bench.Monitor          bench/monitor.py:45-76     episode return (unclipped) and length in agent steps
 *   EpisodicLifeEnv        atari_wrappers.py:157-191  a lost life ends the agent's episode; its reset() is a real reset only
 *                          after a real game over, otherwise ONE no-op agent step whose `done` is ignored (:186-187)
 *   FireResetEnv           atari_wrappers.py:137-155  reset(), then agent steps with action #1 and action #2 of the action
 *                          set, each followed by a reset() if it reports done; the observation is the one of the second step
 *   WarpFrame              atari_wrappers.py:230-244  gray frame (Toybox frames are already gray, :241-242) resized to
 *                          out_w x out_h by area averaging (exact rational weights, round half up)
 *   ClipRewardEnv          atari_wrappers.py:221-227  sign(reward)
 *   DummyVecEnv            vec_env/dummy_vec_env.py:45-60  an env that reports done is reset and its observation is reset()'s
 *   VecFrameStack(stack)   vec_env/vec_frame_stack.py:17-30  roll the channel axis, zero the stack of envs that are done,
 *                          write the new frame last -- or, with stack_fill = 1, FrameStack(k) inside every env
 *                          (atari_wrappers.py:246-275): the same roll, but a reset fills the whole stack with its observation
 * obs = uint8[N][out_h][out_w][stack], reward float32[N], done uint8[N].
 * Not the Python's: NoopResetEnv draws its count from numpy's RandomState; here it is counter-based (below) unless counts
 * are injected with tbx_agent_set_noops.  bench.Monitor raises when an env is stepped after its game ended inside
 * EpisodicLifeEnv's ignored no-op step; here that is TBX_E_NEEDS_RESET and the stack carries on as it does without a Monitor. */
typedef struct tbx_agent_config {
    int32_t skip;          /* >= 1 (4) */
    int32_t out_h, out_w;  /* 84, 84 */
    int32_t stack;         /* 1..4 (4) */
    int32_t clip_reward;   /* 0 / 1 */
    int32_t episodic_life; /* EpisodicLifeEnv on / off */
    int32_t fire_reset;    /* FireResetEnv on / off */
    int32_t noop_max;      /* NoopResetEnv: 1..noop_max no-op frames after a real reset; 0 = off */
    uint64_t noop_seed;    /* count = 1 + splitmix64(noop_seed ^ (global env << 32) ^ episode index) % noop_max */
    uint64_t env_offset;   /* global index of env 0 of this engine (sharded batches) */
    int32_t stack_fill;    /* what a reset leaves in the OLDER stack slots: 0 = zeros (VecFrameStack, vec_frame_stack.py:22-33),
                            * 1 = the reset observation itself (the per-env FrameStack of wrap_deepmind(frame_stack=True),
                            * atari_wrappers.py:246-275: reset() appends the observation k times) */
    int32_t new_plane;     /* 1: the observation kernels also write the NEWEST plane alone, dense, into TBX_BUF_AGENT_PLANE
                            * (uint8[N][out_h][out_w]) -- what a host-side VecFrameStack receives per step (one new frame per
                            * env, vec_frame_stack.py:19-27), a quarter of the bytes of the whole stacks.
                            * 2: the plane INSTEAD of the stack -- the reference's own data flow, where a worker produces one
                            * frame per step and the stack exists only at the receiver (vec_frame_stack.py:17-30; LazyFrames
                            * shares the frames between observations, atari_wrappers.py:288-317).  The device keeps a ring of
                            * the last `stack` planes, TBX_BUF_AGENT_RING = uint8[stack][N][out_h][out_w]; every reset / agent
                            * step writes slot head = (head + 1) % stack (tbx_agent_ring_head; TBX_BUF_AGENT_PLANE is that slot's
                            * address and moves with it), so the stack of env i, oldest first, is ring[(head + 1 + c) % stack][i],
                            * c = 0 .. stack - 1 -- the same bytes as TBX_BUF_AGENT_OBS[i][y][x][c] of the other modes.  An env
                            * whose stack starts afresh (reset; done under VecFrameStack) gets its other slots rewritten (zeros,
                            * or the observation with stack_fill = 1).  7 KB written per env and step instead of the roll's 21 KB
                            * read + 28 KB written.  There is no TBX_BUF_AGENT_OBS and no obs output in this mode (TBX_E_INVALID). */
} tbx_agent_config_t;

int tbx_agent_init(tbx_engine* engine, const tbx_agent_config_t* cfg);
/* tbx_agent_init constructs the stack (zero frame buffers, EpisodicLifeEnv.was_real_done = True, Monitor idle). */
/* NoopResetEnv.override_num_noops (atari_wrappers.py:115-123) per env: counts_host[i] > 0 replaces the drawn count in env i,
 * 0 keeps the default rule; NULL removes the override. */
int tbx_agent_set_noops(tbx_engine* engine, const int32_t* counts_host);
/* venv.reset(): reset() of every env's stack (so with EpisodicLifeEnv on, a SECOND call in the middle of an episode only
 * advances one no-op agent step, as in the reference); stack = zeros with the warped observation last (VecFrameStack.reset).
 * obs_host may be NULL. */
int tbx_agent_reset(tbx_engine* engine, uint8_t* obs_host);
/* one agent step with host pointers (synchronous); any output pointer may be NULL */
int tbx_agent_step(tbx_engine* engine, const int32_t* ale_actions_host, float* reward_host, uint8_t* done_host, uint8_t* obs_host);
/* device-resident forms: actions in HBM, or generated on the device like tbx_step_synthetic with t = agent step index
 * (every sub-frame of an agent step repeats the same action); results in TBX_BUF_AGENT_* */
int tbx_agent_step_device(tbx_engine* engine, const int32_t* ale_actions_dev, void* stream);
int tbx_agent_step_synthetic(tbx_engine* engine, uint64_t action_seed, uint64_t t, uint64_t env_offset, void* stream);

#define TBX_BUF_AGENT_OBS    6   /* uint8[N][out_h][out_w][stack] */
#define TBX_BUF_AGENT_REWARD 7   /* float32[N] */
#define TBX_BUF_AGENT_DONE   8   /* uint8[N] */
/* episode monitor (bench.Monitor / VecMonitor: baselines/bench/monitor.py:51-76, common/vec_env/vec_monitor.py:21-37) */
#define TBX_BUF_AGENT_EP_DONE   9    /* uint8[N]   a real episode (game over) ended during the last agent step */
#define TBX_BUF_AGENT_EP_RETURN 10   /* float32[N] its unclipped return 'r' (valid where EP_DONE) */
#define TBX_BUF_AGENT_EP_LENGTH 11   /* int32[N]   its length 'l' in agent steps (valid where EP_DONE) */
/* host copy of the three episode-monitor arrays of the last agent step (any pointer may be NULL) */
int tbx_agent_episodes(tbx_engine* engine, uint8_t* ep_done_host, float* ep_return_host, int32_t* ep_length_host);
#define TBX_BUF_AGENT_PLANE    13   /* uint8[N][out_h][out_w] the newest plane of every stack (tbx_agent_config_t::new_plane) */
#define TBX_BUF_AGENT_RING     14   /* uint8[stack][N][out_h][out_w] the last `stack` planes (new_plane = 2), newest in slot head */
/* the ring slot that holds the newest plane (new_plane = 2; 0 .. stack - 1, advanced by every tbx_agent_reset / agent step) */
int tbx_agent_ring_head(tbx_engine* engine, int32_t* out_head);

/* ------------------------------------------------------------------ host delivery: step_async / step_wait
 * The reference's consumers sit on the HOST side of the boundary: VecEnv.step_async(actions) sends the actions to the workers,
 * step_wait() receives (obs, rews, dones, infos) (baselines/baselines/common/vec_env/__init__.py:26-131,
 * subproc_vec_env.py:63-74), and what a worker sends per step is ONE new frame per env -- VecFrameStack rolls its stack on the
 * receiving side (vec_frame_stack.py:17-30).  The "_begin" calls queue the whole step on the engine's stream -- actions host ->
 * device, the step (the agent step with all its wrappers), and the copies of every requested output into the caller's host
 * buffers -- and return at once; the "_end" call blocks until they have arrived and reports what the synchronous form reports
 * (TBX_E_ACTION, TBX_E_NEEDS_RESET).  Between the two the caller's thread is free (the learner's update of the previous
 * rollout step overlaps with the device step and the PCIe copy).  Every pointer of the out-struct may be NULL; the buffers must stay valid
 * until the "_end" call returns and should be page-locked (tbx_host_alloc): a copy into pageable memory is staged by the runtime and blocks the
 * "_begin" call.  The actions are copied out of the caller's array before "_begin" returns.  Any other call on the handle
 * between "_begin" and "_end" ends the step first (program order holds as everywhere). */
int tbx_host_alloc(void** out_ptr, size_t bytes);   /* page-locked host memory (hipHostMalloc), any device */
int tbx_host_free(void* ptr);
typedef struct tbx_agent_host_out {
    float*   reward;      /* float32[N] */
    uint8_t* done;        /* uint8[N] */
    uint8_t* obs;         /* uint8[N][out_h][out_w][stack]: the whole stacks (TBX_BUF_AGENT_OBS; not with new_plane = 2) */
    uint8_t* plane;       /* uint8[N][out_h][out_w]: the newest plane only (TBX_BUF_AGENT_PLANE; needs new_plane = 1 or 2) */
    uint8_t* ep_done;     /* uint8[N], float32[N], int32[N]: the episode monitor (tbx_agent_episodes) */
    float*   ep_return;
    int32_t* ep_length;
} tbx_agent_host_out_t;
int tbx_agent_step_begin(tbx_engine* engine, const int32_t* ale_actions_host, const tbx_agent_host_out_t* out);
int tbx_agent_step_end(tbx_engine* engine);
/* the current contents of the same buffers (after tbx_agent_reset or a step), synchronous: what venv.reset() hands out when
 * the host keeps the stacks (tbx_agent_reset(NULL), then the plane) */
int tbx_agent_fetch(tbx_engine* engine, const tbx_agent_host_out_t* out);
/* The plain env step in the same shape: ToyboxBaseEnv.step for every env (envs/atari/base.py:115-149 under
 * dummy_vec_env.py:45-54) = tbx_step + the frames of the states it leaves (channels 1, 3 or 4; frame may be NULL). */
typedef struct tbx_step_host_out {
    int32_t* reward;      /* int32[N] */
    uint8_t* done;        /* uint8[N] */
    int32_t* lives;       /* int32[N] */
    int32_t* score;       /* int32[N] */
    uint8_t* frame;       /* uint8[N][H][W][channels] */
    int32_t  channels;
    int32_t  _pad;
} tbx_step_host_out_t;
int tbx_step_begin(tbx_engine* engine, const int32_t* ale_actions_host, uint32_t flags, const tbx_step_host_out_t* out);
int tbx_step_end(tbx_engine* engine);
/* The receiving side's frame stack as host code: VecFrameStack.step_wait (vec_frame_stack.py:17-27) over channel-last stacks
 * uint8[n][px][stack] -- dst = src with every pixel's channels rolled by one and `plane` (uint8[n][px], what tbx_agent_host_out_t::
 * plane received) as the newest channel; the stack of an env whose `done` byte is set starts anew: zeros in the older channels
 * (fill = 0, VecFrameStack :21-23) or the new plane in every channel (fill = 1, FrameStack.reset inside the env,
 * atari_wrappers.py:257-261).  done == NULL: no env is done; reset != 0: every env is (VecFrameStack.reset :29-33).  dst may
 * be src.  Plain host loops on `threads` threads (<= 0: the usable cores, at most 16); no engine, no device. */
int tbx_host_stack_push(uint8_t* dst, const uint8_t* src, const uint8_t* plane, const uint8_t* done, int reset,
                        int n, int px, int stack, int fill, int threads);

/* ------------------------------------------------------------------ multi-GPU record gather
 * One process per GPU, one engine per process = one contiguous shard of the env batch; envs never interact, so the only
 * exchange is one all-gather per step of the packed TBX_BUF_PACKED records (8 B/env: reward i32, done u8, lives u8) over
 * RCCL / xGMI.  replaces the pipes of pickled (ob, rew, done, info) tuples between the reference's worker processes and
 * the learner (baselines/baselines/common/vec_env/subproc_vec_env.py:63-74); frames stay on their GPU.
 * librccl is resolved with dlopen inside these calls (no link-time dependency, no PyTorch); without it they return
 * TBX_E_UNSUPPORTED and a caller may fall back to gathering the records on the host. */
#define TBX_GATHER_ID_BYTES 128
/* rank 0: a fresh communicator id (ncclGetUniqueId) to hand to every rank out of band (file, env var, pipe).
 * Errors are reported through tbx_last_error(NULL). */
int tbx_gather_unique_id(void* id_out, size_t id_bytes);
/* Collective over all ranks (ncclCommInitRank).  records_per_rank >= this engine's env count is the per-rank slot width of
 * the gathered layout [nranks][records_per_rank] (ranks may hold unequal shards; unused slots read 0). */
int tbx_gather_init(tbx_engine* engine, int nranks, int rank, int records_per_rank, const void* id, size_t id_bytes);
/* What the communicator itself reports (ncclCommCount; == nranks after a successful tbx_gather_init) and the path of the
 * librccl that was loaded -- so that a benchmark line can say which collective really ran over how many ranks. */
int tbx_gather_nranks(tbx_engine* engine);
const char* tbx_gather_library(tbx_engine* engine);
/* K of the record ring this communicator was initialised with (TBX_OPT_GATHER_EVERY at tbx_gather_init; 1 = a collective per
 * step), and how many steps' records sit in the ring waiting for the next collective (0 right after one went out). */
int tbx_gather_every(tbx_engine* engine);
int tbx_gather_fill(tbx_engine* engine);
/* Queue the all-gather of the last step's records into out_dev (NULL: the engine-owned TBX_BUF_GATHERED) -- with a K-step ring
 * (TBX_OPT_GATHER_EVERY) only every K-th call queues anything, then for the last K steps at once.  Asynchronous: it
 * is ordered after everything queued through this handle so far, runs on an engine-owned communication stream, and the next
 * step that rewrites those records is ordered after it -- what the caller queues in between (the rasteriser) overlaps with it.  `stream` is not used to run it; make a
 * stream wait for the result with tbx_gather_wait.
 * With the fused call (tbx_render_step_synthetic) in stream order and one collective PER STEP there is nothing in between: the
 * collective waits for the whole launch (the step rides in it) and the next launch rewrites the same records, so it waits for
 * the collective -- no overlap in that combination; the K-step ring (TBX_OPT_GATHER_EVERY > 1), the two-launch loop or
 * overlapped fused launches (TBX_OPT_FUSED_OVERLAP: two output sets) overlap it. */
int tbx_gather(tbx_engine* engine, uint64_t* out_dev, void* stream);
/* Make `stream` wait for the last queued gather (device-side consumers of the gathered records). */
int tbx_gather_wait(tbx_engine* engine, void* stream);
/* Block until the last queued gather has finished and copy TBX_BUF_GATHERED to the host: uint64[nranks * K * records_per_rank]
 * (K = tbx_gather_every, 1 by default). */
int tbx_gather_host(tbx_engine* engine, uint64_t* out_host);
/* Blocking max-reduction of one double over all ranks (barrier + "slowest rank" timing of bench.py). */
int tbx_gather_reduce_max(tbx_engine* engine, double* inout_host);
#define TBX_BUF_GATHERED 12   /* uint64[nranks][K][records_per_rank] result of tbx_gather(out_dev = NULL) */

/* Address of an engine-owned device buffer (TBX_BUF_*).  In pipelined mode (TBX_OPT_PIPELINE) the step outputs and the
 * engine-owned frame buffer each alternate between two addresses: ask again after every step / render. */
int tbx_device_buffer(tbx_engine* engine, int which, void** out_ptr, size_t* out_bytes);

/* ------------------------------------------------------------------ engine options
 * Launch-time choices that used to be environment variables.  Every non-default path they select has a parity test against
 * the oracle (tests/test_gpu_paths.py, tests/test_preproc.py).  Unknown options / values: TBX_E_INVALID.
 *
 * TBX_OPT_PIPELINE -- random-rollout loops (tbx_step_synthetic + tbx_render_device) only; 0 = off (default).
 *   A step whose actions are generated on the device depends on nothing the previous frame's rasteriser produces, and a
 *   rasteriser that reads step-written render records (Breakout, SpaceInvaders) disturbs nothing the next step touches.  With the option
 *   on, such engines keep TWO sets of render records, of step outputs (TBX_BUF_REWARD / DONE / LIVES / SCORE / PACKED) and
 *   -- for tbx_render_device(out_dev = NULL) -- of TBX_BUF_FRAME, and run the calls on internal streams:
 *     value 2: tbx_step_synthetic N+1 runs beside the rasteriser of frame N;
 *     value 3: additionally consecutive rasteriser launches alternate between two internal streams and the two frame
 *              buffers, so that launch N+1 fills the ramp-down of launch N (small batches: BASELINE configs 2-4, the
 *              per-GPU share of a strong-scaled batch);
 *     value 1: the engine's choice: value 3 for Breakout from 4 096 to 16 383 envs and for SpaceInvaders below 16 384 envs while
 *              no per-step gather is initialised (BASELINE configs 2-4: 0.0985 -> 0.0868 ms and 0.162 -> 0.150 ms per step at
 *              4 096 envs), 0 otherwise.  (Rounds 2-3 chose 2 for large batches, 1-15 % faster "depending on the box": what it
 *              bought was a rasteriser launch that does not start against an idle memory system, and the rasterisers now see to
 *              that themselves -- csrc/raster.hpp, tbx_stagger_first_waves; scripts/pipeline_sweep.py, 16 384 .. 65 536 envs:
 *              stream order 0.5-3 % ahead of value 2.  Value 3 loses with a gather: 0.230 against 0.175 ms at 8 192 envs.)
 *   Contract in this mode: the stream a call names still waits for the call's work, so anything queued on it afterwards sees
 *   the result; the result of step N (render N) stays valid for readers queued on that stream BEFORE step N+1 (render N+1)
 *   is issued -- the same rule as without the option -- but it lives at the address tbx_device_buffer reports after the
 *   call, which alternates.  Every other call on the handle first joins the pipeline.  Engines whose rasteriser reads live
 *   state (Amidar, GridWorld, Breakout with intervention-written bricks, SpaceInvaders with intervention-written enemy
 *   positions) ignore the option.  (Round 3 let Amidar take part through a record kernel behind its step: slower than stream
 *   order at every size -- its step, a chain of dependent loads, runs ten times longer beside a rasteriser -- and removed.) */
#define TBX_OPT_PIPELINE      0
/* batch step kernel form of Breakout and Amidar: 0 = the engine's choice (by batch size), 1 = one thread per env, 2 = one
 * wavefront per env */
#define TBX_OPT_STEP_FORM     1
/* waves per frame of the rasteriser launches: 0 = the engine's choice (by game, channels and batch size) */
#define TBX_OPT_RENDER_SPLIT  2
/* 1: the agent observation goes through two full-resolution gray renders and the generic warp kernel instead of the per-game
 * fused kernels (the cross-check path); read by tbx_agent_init */
#define TBX_OPT_AGENT_GENERIC 3
/* 1 (default): tbx_step1 on a one-env engine talks to the resident step kernel; 0: single-env launch + read-back */
#define TBX_OPT_RESIDENT_STEP 4
/* K-step record ring of the multi-GPU gather, 1..64 (default 1 = one collective per step).  Read by tbx_gather_init (like
 * TBX_OPT_AGENT_GENERIC by tbx_agent_init; a later change takes effect at the next tbx_gather_init).  SURVEY.md 8e: "one
 * ncclAllGather per step (or per K steps)" -- and what the reference's learners consume is a K-step rollout anyway: the A2C
 * runner collects nsteps = 5 steps of (obs, rewards, dones) per update, PPO2 128 (baselines/baselines/a2c/runner.py:16,
 * a2c/a2c.py:87, ppo2/ppo2.py:103).  With K > 1 the step kernels write their 8-byte records straight into slot j of a ring
 * [K][records_per_rank] (j = steps since the last collective; TBX_BUF_PACKED names the slot of the most recent step), tbx_gather
 * only counts -- no stream operation at all -- and every K-th call sends the whole ring with ONE ncclAllGather of
 * K * records_per_rank records; the following K steps fill a second ring meanwhile.  Gathered layout
 * [nranks][K][records_per_rank] (TBX_BUF_GATHERED, tbx_gather_host).  The pipelined mode is off while a ring is in force. */
#define TBX_OPT_GATHER_EVERY  5
/* how the record gather travels, read by tbx_gather_init: 0 (default) = RCCL (ncclAllGather over xGMI); 1 = host-staged over a
 * POSIX shared-memory segment named after the id -- the ranks of ONE node, no librccl needed ("a host-staged
 * gather is the fallback if RCCL is missing").  Same calls, same gathered layout, same ordering rules; but every collective
 * blocks the calling thread until the step that wrote the records has finished and all ranks have exchanged them.  It also lets
 * several ranks share ONE device (RCCL refuses two ranks of a communicator on one GPU): `bench.py --gather host --one-device`
 * walks the whole N-process flow on a one-GPU box.  tbx_gather_library() names the transport in use. */
#define TBX_OPT_GATHER_TRANSPORT 6
/* Consecutive tbx_render_step_synthetic(out_dev = NULL) calls of an engine whose call is ONE launch (TBX_OPT_RENDER_STEP_FUSED):
 * 0 (default) = the engine's choice by batch size, 1 = overlap them, 2 = stream order.
 *   Launch N+1 needs only what the STEP BLOCKS of launch N write (state, render records, outputs) -- the first few blocks of a
 *   launch that otherwise paints for 70 .. 1 200 us.  Overlapped, consecutive launches alternate between two internal streams,
 *   the two sets of step outputs and two engine-owned frame buffers (as in the pipelined mode), over THREE buffers of render
 *   records, and launch N+1 is ordered behind a device counter that the step blocks of launch N bump once their stores are
 *   visible device-wide: a one-wave kernel in front of launch N+1 waits for it (bounded; a time-out is reported by tbx_sync as
 *   TBX_E_NO_DEVICE), the launch itself never spins.  So launch N+1 ramps up in the ramp-down of launch N -- what N worker
 *   processes stepping independently of each other give the reference (baselines/baselines/common/vec_env/subproc_vec_env.py:49-74),
 *   and what the per-GPU share of a strong-scaled batch (8 192 envs) loses between launches in stream order.  It works with the
 *   record gather in both forms (one collective per step: the collective of step N then runs beside launch N+1; K-step ring).
 *   Contract: TBX_BUF_FRAME / TBX_BUF_REWARD ... alternate between two addresses -- ask tbx_device_buffer after every call whose
 *   results are to be read -- and it is THAT call which makes the stream the step call named wait for the launch that wrote
 *   them (the stream joins lazily: a loop that only rolls, or whose only consumer is the record gather, queues nothing on the
 *   caller's stream -- with a gather on the device a wait and a fence per call there cost more than the overlap gains).  The
 *   results of call N stay valid for readers queued on that stream after tbx_device_buffer and before call N+1.  A call with
 *   out_dev != NULL, or any other call on the handle, first joins everything (program order holds as everywhere). */
#define TBX_OPT_FUSED_OVERLAP 7
/* How early an overlapped launch is released, in blocks of the launch before it: launch N+1 starts once the step blocks of
 * launch N are through AND the block `lead` blocks before the end of launch N's grid has STARTED (blocks start in index order:
 * from there on launch N only drains).  0 = the engine's choice; 1 .. 2^20 blocks; a value beyond the grid releases launch N+1
 * as soon as the step blocks are through (both launches then run side by side for most of their length -- measured slower,
 * profiles/r06_experiments.txt). */
#define TBX_OPT_FUSED_OVERLAP_LEAD 8
/* tbx_rollout_synthetic as ONE step launch + the chunk's rasteriser launches on internal streams (see there): 0 (default) = the
 * engine's choice by batch size, 1 = wherever the engine can (the rasteriser form its choice), 2 = never (the k single calls in
 * stream order); 3 / 4 = as 1 with the rasteriser form named -- 3 a launch per frame on two streams, 4 one launch per chunk on one
 * (games without that form: as 3) */
#define TBX_OPT_ROLLOUT_CHUNKS 9
#define TBX_OPT_COUNT         10
/* read-only (tbx_get_option): what TBX_OPT_PIPELINE resolves to on this engine right now -- 0, 2 or 3 */
#define TBX_OPT_PIPELINE_ACTIVE 100
/* read-only: 1 while the rasteriser reads step-written render records (Breakout with the canonical wall, SpaceInvaders with
 * the canonical formation), 0 once an intervention has switched the engine to the state-reading rasteriser, or for games
 * without records */
#define TBX_OPT_RECORDS_ACTIVE  101
/* read-only: 1 if tbx_render_step_synthetic(channels = 3) is ONE launch on this engine right now (the rasteriser reads
 * step-written records and the game has a fused kernel), 0 if it is the two launches in stream order */
#define TBX_OPT_RENDER_STEP_FUSED 102
/* read-only: 1 if tbx_render_step_synthetic(out_dev = NULL, channels = 3) would be an overlapped launch right now
 * (TBX_OPT_FUSED_OVERLAP resolved), 0 if it runs in stream order */
#define TBX_OPT_FUSED_OVERLAP_ACTIVE 103
/* read-only: 1 if tbx_rollout_synthetic(channels = 3) would run as overlapped chunks right now, 0 if as single calls */
#define TBX_OPT_ROLLOUT_CHUNKS_ACTIVE 104
int tbx_set_option(tbx_engine* engine, int option, int value);
int tbx_get_option(tbx_engine* engine, int option, int* value_out);
/* Block until all work queued by this engine has finished; reports a pending TBX_E_ACTION. */
int tbx_sync(tbx_engine* engine);

#ifdef __cplusplus
}
#endif
#endif /* TOYBOX_AMD_H */
