"""ctypes mirror of include/toybox_amd.h: POD records, constants and prototype binding.

Every Structure here must match the C header byte for byte; tests/test_abi.py checks the sizes
against tbx_state_size()/tbx_config_size() of the loaded library and that every declared symbol
is exported.
"""
import ctypes as C

ABI_VERSION = 1

GAME_BREAKOUT, GAME_AMIDAR, GAME_SPACE_INVADERS, GAME_GRIDWORLD = 0, 1, 2, 3
GAME_IDS = {"breakout": GAME_BREAKOUT, "amidar": GAME_AMIDAR, "space_invaders": GAME_SPACE_INVADERS,
            "spaceinvaders": GAME_SPACE_INVADERS, "gridworld": GAME_GRIDWORLD}
GAME_NAMES = {GAME_BREAKOUT: "breakout", GAME_AMIDAR: "amidar", GAME_SPACE_INVADERS: "space_invaders",
              GAME_GRIDWORLD: "gridworld"}

OK, E_INVALID, E_NO_DEVICE, E_NOMEM, E_UNSUPPORTED, E_ACTION, E_NEEDS_RESET = 0, -1, -2, -3, -4, -5, -6

BTN_LEFT, BTN_RIGHT, BTN_UP, BTN_DOWN, BTN_BUTTON1, BTN_BUTTON2 = 1, 2, 4, 8, 16, 32
STEP_AUTO_RESET = 1
BUF_REWARD, BUF_DONE, BUF_LIVES, BUF_SCORE, BUF_FRAME, BUF_PACKED = 0, 1, 2, 3, 4, 5
BUF_AGENT_OBS, BUF_AGENT_REWARD, BUF_AGENT_DONE = 6, 7, 8
BUF_AGENT_EP_DONE, BUF_AGENT_EP_RETURN, BUF_AGENT_EP_LENGTH = 9, 10, 11
BUF_GATHERED = 12
BUF_AGENT_PLANE = 13
BUF_AGENT_RING = 14
BUF_ROLLOUT_FRAMES = 15            # uint8[k][N][H][W][C] frames of the last tbx_rollout_synthetic
BUF_ROLLOUT_PACKED = 16            # uint64[k][stride] its step records (stride = the gather's records_per_rank under a K-step ring, else N)
GATHER_ID_BYTES = 128
# engine options (tbx_set_option)
OPT_PIPELINE, OPT_STEP_FORM, OPT_RENDER_SPLIT, OPT_AGENT_GENERIC, OPT_RESIDENT_STEP, OPT_GATHER_EVERY = 0, 1, 2, 3, 4, 5
OPT_GATHER_TRANSPORT = 6           # 0 RCCL, 1 host-staged over POSIX shared memory (one node)
GATHER_RCCL, GATHER_HOST = 0, 1
OPT_FUSED_OVERLAP = 7              # consecutive fused render+step launches: 0 the engine's choice, 1 overlapped on two lanes, 2 stream order
FUSED_OVERLAP_AUTO, FUSED_OVERLAP_ON, FUSED_OVERLAP_OFF = 0, 1, 2
OPT_FUSED_OVERLAP_LEAD = 8         # blocks before the end of launch N at which launch N+1 is released (0: the engine's choice)
OPT_PIPELINE_ACTIVE = 100          # read-only: what OPT_PIPELINE resolves to on the engine
OPT_RECORDS_ACTIVE = 101           # read-only: the rasteriser reads step-written render records
OPT_RENDER_STEP_FUSED = 102        # read-only: tbx_render_step_synthetic is one launch on this engine
OPT_FUSED_OVERLAP_ACTIVE = 103     # read-only: such launches would be overlapped right now
OPT_ROLLOUT_CHUNKS = 9             # tbx_rollout_synthetic as one step launch + the chunk's rasteriser launches on internal streams: 0 engine's choice, 1 on, 2 off
ROLLOUT_CHUNKS_AUTO, ROLLOUT_CHUNKS_ON, ROLLOUT_CHUNKS_OFF = 0, 1, 2
ROLLOUT_CHUNKS_PER_FRAME, ROLLOUT_CHUNKS_SPAN = 3, 4   # on, with the rasteriser form named: a launch per frame on two lanes, one per chunk on one
OPT_ROLLOUT_CHUNKS_ACTIVE = 104    # read-only: it would run that way right now
PIPELINE_OFF, PIPELINE_AUTO, PIPELINE_STEP_BESIDE_RENDER, PIPELINE_OVERLAP_RENDERS = 0, 1, 2, 3
STEP_FORM_AUTO, STEP_FORM_THREAD_PER_ENV, STEP_FORM_WAVE_PER_ENV = 0, 1, 2

# batched interventions (tbx_edit / tbx_reduce): ids as include/toybox_amd.h declares them
QUERY_TILE_TO_WORLD = 1
QUERY_WORLD_TO_TILE = 2
EDIT_MAX_ARGS = 16
EDIT_SET_LIVES = 1
EDIT_SET_SCORE = 2
EDIT_SET_LEVEL = 3
EDIT_BRK_COLUMN_ALIVE = 10
EDIT_BRK_ROW_ALIVE = 11
EDIT_BRK_ALL_ALIVE = 12
EDIT_BRK_BRICK_ALIVE = 13
EDIT_BRK_PADDLE = 14
EDIT_BRK_BALL = 15
EDIT_AMI_TIMERS = 20
EDIT_AMI_JUMPS = 21
EDIT_AMI_TILE = 22
EDIT_AMI_ENEMY_AI = 23
EDIT_AMI_PLAYER_TILE = 24
EDIT_AMI_PLAYER_RANDOM_START = 25
EDIT_SI_UFO_APPEARANCE = 30
QUERY_BRK_BRICKS_REMAINING = 110
QUERY_BRK_NUM_BRICKS = 111
QUERY_BRK_COLUMN = 112
QUERY_BRK_ROW = 113
QUERY_BRK_IS_CHANNEL = 114
QUERY_BRK_CHANNEL_COUNT = 115
QUERY_BRK_FIND_CHANNEL = 116
QUERY_BRK_PADDLE = 117
QUERY_BRK_BALLS = 118
QUERY_BRK_FIND_BRICK = 119
QUERY_AMI_MODE = 120
QUERY_AMI_ANY_CAUGHT = 121
QUERY_AMI_TILE = 122
QUERY_AMI_COUNT_TILES = 123
QUERY_AMI_ADJACENT = 124
QUERY_AMI_ENEMY_DISTANCES = 125
QUERY_AMI_PLAYER_TILE = 126
QUERY_AMI_PLAYER_ENEMY_DISTANCES = 127
QUERY_AMI_PLAYER_ON_PAINTED = 128
QUERY_AMI_PLAYER_NEAR_UNPAINTED = 129
QUERY_SI_SHIP = 130
QUERY_AMI_TILES_MASK = 131
QUERY_AMI_RANDOM_TILE = 133
QUERY_AMI_RANDOM_DIR = 134

BRK_MAX_BALLS, BRK_COLS, BRK_MAX_ROWS, BRK_MAX_BRICKS, BRK_MAX_STARTS, BRK_MAX_SEGMENTS = 4, 18, 14, 256, 8, 16


class Color(C.Structure):
    _fields_ = [("r", C.c_uint8), ("g", C.c_uint8), ("b", C.c_uint8), ("a", C.c_uint8)]

    def to_json(self):
        return {"r": self.r, "g": self.g, "b": self.b, "a": self.a}

    @staticmethod
    def from_json(d):
        clamp = lambda v: max(0, min(255, int(v)))
        return Color(clamp(d["r"]), clamp(d["g"]), clamp(d["b"]), clamp(d["a"]))


class BreakoutConfig(C.Structure):
    _fields_ = [
        ("rand", C.c_uint64 * 2),
        ("start_lives", C.c_int32),
        ("n_rows", C.c_int32),
        ("row_scores", C.c_int32 * BRK_MAX_ROWS),
        ("row_colors", Color * BRK_MAX_ROWS),
        ("ball_speed_row_depth", C.c_int32),
        ("n_starts", C.c_int32),
        ("ball_speed_slow", C.c_double),
        ("ball_speed_fast", C.c_double),
        ("start_x", C.c_double * BRK_MAX_STARTS),
        ("start_y", C.c_double * BRK_MAX_STARTS),
        ("start_angle_deg", C.c_double * BRK_MAX_STARTS),
        ("start_dir_x", C.c_double * BRK_MAX_STARTS),
        ("start_dir_y", C.c_double * BRK_MAX_STARTS),
        ("paddle_discrete_segments", C.c_int32),
        ("_pad0", C.c_int32),
        ("paddle_dir_x", C.c_double * BRK_MAX_SEGMENTS),
        ("paddle_dir_y", C.c_double * BRK_MAX_SEGMENTS),
        ("bg_color", Color),
        ("frame_color", Color),
        ("paddle_color", Color),
        ("ball_color", Color),
    ]


class Brick(C.Structure):
    _fields_ = [
        ("x", C.c_double), ("y", C.c_double), ("w", C.c_double), ("h", C.c_double),
        ("points", C.c_int32), ("depth", C.c_int32), ("row", C.c_int32), ("col", C.c_int32),
        ("color", Color),
        ("alive", C.c_uint8), ("destructible", C.c_uint8), ("_pad", C.c_uint8 * 2),
    ]


class BreakoutState(C.Structure):
    _fields_ = [
        ("rand", C.c_uint64 * 2),
        ("score", C.c_int32), ("lives", C.c_int32), ("level", C.c_int32),
        ("is_dead", C.c_uint8), ("reset", C.c_uint8), ("_pad0", C.c_uint8 * 2),
        ("paddle_x", C.c_double), ("paddle_y", C.c_double), ("paddle_vx", C.c_double), ("paddle_vy", C.c_double),
        ("paddle_width", C.c_double), ("paddle_speed", C.c_double), ("ball_radius", C.c_double),
        ("n_balls", C.c_int32), ("n_bricks", C.c_int32),
        ("ball_x", C.c_double * BRK_MAX_BALLS), ("ball_y", C.c_double * BRK_MAX_BALLS),
        ("ball_vx", C.c_double * BRK_MAX_BALLS), ("ball_vy", C.c_double * BRK_MAX_BALLS),
        ("bricks", Brick * BRK_MAX_BRICKS),
    ]


SI_COLS, SI_MAX_ROWS, SI_MAX_ENEMIES, SI_MAX_SHIELDS, SI_SHIELD_W, SI_SHIELD_H, SI_MAX_LASERS = 6, 10, 64, 3, 16, 18, 8
DIR_NAMES = ["Up", "Down", "Left", "Right"]       # interventions/core.py:125-135


class SIConfig(C.Structure):
    _fields_ = [
        ("rand", C.c_uint64 * 2),
        ("jitter", C.c_double),
        ("start_lives", C.c_int32), ("n_rows", C.c_int32), ("n_shields", C.c_int32), ("enemy_protocol", C.c_int32),
        ("row_scores", C.c_int32 * SI_MAX_ROWS),
        ("shield_x", C.c_int32 * SI_MAX_SHIELDS), ("shield_y", C.c_int32 * SI_MAX_SHIELDS),
    ]


class SILaser(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32), ("t", C.c_int32),
                ("movement", C.c_int32), ("speed", C.c_int32), ("color", Color)]


class SIEnemy(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("row", C.c_int32), ("col", C.c_int32), ("id", C.c_int32),
                ("points", C.c_int32), ("death_counter", C.c_int32), ("alive", C.c_uint8), ("_pad", C.c_uint8 * 3)]


class SIState(C.Structure):
    _fields_ = [
        ("rand", C.c_uint64 * 2),
        ("score", C.c_int32), ("lives", C.c_int32), ("level", C.c_int32),
        ("life_display_timer", C.c_int32), ("enemy_shot_delay", C.c_int32),
        ("n_enemies", C.c_int32), ("n_enemy_lasers", C.c_int32), ("has_ship_laser", C.c_int32),
        ("ship_x", C.c_int32), ("ship_y", C.c_int32), ("ship_w", C.c_int32), ("ship_h", C.c_int32), ("ship_speed", C.c_int32),
        ("ship_death_counter", C.c_int32), ("ship_color", Color),
        ("ship_alive", C.c_uint8), ("ship_death_hit_1", C.c_uint8), ("_pad0", C.c_uint8 * 2),
        ("ufo_x", C.c_int32), ("ufo_y", C.c_int32), ("ufo_appearance_counter", C.c_int32), ("ufo_death_counter", C.c_int32),
        ("move_counter", C.c_int32), ("move_dir", C.c_int32),
        ("visual_orientation", C.c_uint8), ("_pad1", C.c_uint8 * 3),
        ("n_shields", C.c_int32),
        ("shield_x", C.c_int32 * SI_MAX_SHIELDS), ("shield_y", C.c_int32 * SI_MAX_SHIELDS),
        ("shield_color", Color * SI_MAX_SHIELDS),
        ("shield_rows", (C.c_uint16 * SI_SHIELD_H) * SI_MAX_SHIELDS),
        ("ship_laser", SILaser),
        ("enemy_lasers", SILaser * SI_MAX_LASERS),
        ("enemies", SIEnemy * SI_MAX_ENEMIES),
    ]


AMI_BOARD_W, AMI_BOARD_H, AMI_MAX_ENEMIES, AMI_MAX_BOXES, AMI_MAX_HISTORY, AMI_MAX_CHASE_J = 32, 31, 8, 64, 16, 8
AMI_TILE_WX, AMI_TILE_WY = 64, 80
TILE_NAMES = ["Empty", "Unpainted", "Painted", "ChaseMarker"]          # interventions/amidar.py:55-59
AI_NAMES = ["Player", "EnemyLookupAI", "EnemyPerimeterAI", "EnemyAmidarMvmt", "EnemyTargetPlayer", "EnemyRandomMvmt"]
QUERY_TILE_TO_WORLD, QUERY_WORLD_TO_TILE = 1, 2


class AmidarAI(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "next", "default_route_index", "start_tx", "start_ty", "vert", "horiz",
                                         "start_vert", "start_horiz", "start_dir", "dir", "vision_distance", "seen_tx", "seen_ty")]


class AmidarMover(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("speed", C.c_int32), ("step_tx", C.c_int32), ("step_ty", C.c_int32),
                ("n_history", C.c_int32), ("history", C.c_int32 * AMI_MAX_HISTORY), ("caught", C.c_int32), ("ai", AmidarAI)]


class AmidarBox(C.Structure):
    _fields_ = [("tl_tx", C.c_int32), ("tl_ty", C.c_int32), ("br_tx", C.c_int32), ("br_ty", C.c_int32),
                ("painted", C.c_uint8), ("triggers_chase", C.c_uint8), ("_pad", C.c_uint8 * 2)]


class AmidarConfig(C.Structure):
    _fields_ = [
        ("rand", C.c_uint64 * 2),
        ("start_lives", C.c_int32), ("start_jumps", C.c_int32), ("jump_time", C.c_int32), ("chase_time", C.c_int32),
        ("box_bonus", C.c_int32), ("chase_score_bonus", C.c_int32),
        ("player_start_tx", C.c_int32), ("player_start_ty", C.c_int32), ("n_enemies", C.c_int32),
        ("render_images", C.c_uint8), ("default_board_bugs", C.c_uint8), ("_pad0", C.c_uint8 * 2),
        ("enemies", AmidarAI * AMI_MAX_ENEMIES),
        ("bg_color", Color), ("player_color", Color), ("unpainted_color", Color), ("painted_color", Color),
        ("enemy_color", Color), ("inner_painted_color", Color),
        ("board", (C.c_uint8 * AMI_BOARD_W) * AMI_BOARD_H),
    ]


class AmidarState(C.Structure):
    _fields_ = [
        ("rand", C.c_uint64 * 2),
        ("score", C.c_int32), ("lives", C.c_int32), ("level", C.c_int32),
        ("jumps", C.c_int32), ("jump_timer", C.c_int32), ("chase_timer", C.c_int32),
        ("n_enemies", C.c_int32), ("n_boxes", C.c_int32), ("n_chase_junctions", C.c_int32),
        ("chase_junctions", C.c_int32 * AMI_MAX_CHASE_J),
        ("player", AmidarMover),
        ("enemies", AmidarMover * AMI_MAX_ENEMIES),
        ("boxes", AmidarBox * AMI_MAX_BOXES),
        ("tiles", (C.c_uint8 * AMI_BOARD_W) * AMI_BOARD_H),
    ]


GW_MAX_DIM, GW_MAX_TILES = 32, 16


class GridWorldTile(C.Structure):
    _fields_ = [("color", Color), ("reward", C.c_int32), ("goal", C.c_uint8), ("walkable", C.c_uint8), ("_pad", C.c_uint8 * 2)]


class GridWorldConfig(C.Structure):
    _fields_ = [
        ("rand", C.c_uint64 * 2),
        ("width", C.c_int32), ("height", C.c_int32), ("n_tiles", C.c_int32),
        ("player_start_x", C.c_int32), ("player_start_y", C.c_int32), ("reward_becomes", C.c_int32),
        ("player_color", Color),
        ("tile_keys", C.c_uint8 * GW_MAX_TILES),
        ("tiles", GridWorldTile * GW_MAX_TILES),
        ("grid", C.c_uint8 * (GW_MAX_DIM * GW_MAX_DIM)),
    ]


class GridWorldState(C.Structure):
    _fields_ = [
        ("score", C.c_int32), ("game_over", C.c_int32), ("player_x", C.c_int32), ("player_y", C.c_int32),
        ("reward_becomes", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("n_tiles", C.c_int32),
        ("player_color", Color),
        ("tiles", GridWorldTile * GW_MAX_TILES),
        ("grid", C.c_uint8 * (GW_MAX_DIM * GW_MAX_DIM)),
    ]


STATE_TYPES = {GAME_BREAKOUT: BreakoutState, GAME_SPACE_INVADERS: SIState, GAME_AMIDAR: AmidarState,
               GAME_GRIDWORLD: GridWorldState}
CONFIG_TYPES = {GAME_BREAKOUT: BreakoutConfig, GAME_SPACE_INVADERS: SIConfig, GAME_AMIDAR: AmidarConfig,
                GAME_GRIDWORLD: GridWorldConfig}

class AgentConfig(C.Structure):
    _fields_ = [("skip", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32), ("stack", C.c_int32), ("clip_reward", C.c_int32),
                ("episodic_life", C.c_int32), ("fire_reset", C.c_int32), ("noop_max", C.c_int32),
                ("noop_seed", C.c_uint64), ("env_offset", C.c_uint64), ("stack_fill", C.c_int32), ("new_plane", C.c_int32)]


class AgentHostOut(C.Structure):
    """tbx_agent_host_out_t: host destinations of an agent step's outputs (addresses; 0 = not wanted)"""
    _fields_ = [("reward", C.c_void_p), ("done", C.c_void_p), ("obs", C.c_void_p), ("plane", C.c_void_p),
                ("ep_done", C.c_void_p), ("ep_return", C.c_void_p), ("ep_length", C.c_void_p)]


class StepHostOut(C.Structure):
    """tbx_step_host_out_t"""
    _fields_ = [("reward", C.c_void_p), ("done", C.c_void_p), ("lives", C.c_void_p), ("score", C.c_void_p), ("frame", C.c_void_p),
                ("channels", C.c_int32), ("_pad", C.c_int32)]


class DeviceIdentity(C.Structure):
    """tbx_device_identity_t"""
    _fields_ = [("ordinal", C.c_int32), ("pci_domain", C.c_int32), ("pci_bus", C.c_int32), ("pci_device", C.c_int32),
                ("total_memory", C.c_uint64), ("compute_units", C.c_int32), ("_pad", C.c_int32), ("arch", C.c_char * 64), ("name", C.c_char * 64)]


_p = C.POINTER
_vp, _i, _u32, _u64, _sz = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_size_t

# name -> (restype, argtypes); the list of symbols include/toybox_amd.h declares
PROTOTYPES = {
    "tbx_abi_version": (_i, []),
    "tbx_last_error": (C.c_char_p, [_vp]),
    "tbx_frame_dims": (_i, [_i, _p(_i), _p(_i)]),
    "tbx_legal_actions": (_i, [_i, _p(C.c_int32), _i]),
    "tbx_ale_action_to_buttons": (_i, [_i]),
    "tbx_state_size": (_sz, [_i]),
    "tbx_config_size": (_sz, [_i]),
    "tbx_create": (_i, [_i, _i, _i, _vp, _sz, _p(_vp)]),
    "tbx_destroy": (_i, [_vp]),
    "tbx_num_envs": (_i, [_vp]),
    "tbx_game": (_i, [_vp]),
    "tbx_seed": (_i, [_vp, _i, _u32]),
    "tbx_seed_array": (_i, [_vp, _vp]),
    "tbx_get_sim_rng": (_i, [_vp, _i, _p(_u64)]),
    "tbx_set_sim_rng": (_i, [_vp, _i, _p(_u64)]),
    "tbx_new_game": (_i, [_vp, _vp]),
    "tbx_step": (_i, [_vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "tbx_step_device": (_i, [_vp, _vp, _u32, _vp]),
    "tbx_step_synthetic": (_i, [_vp, _u64, _u64, _u64, _u32, _vp]),
    "tbx_step1": (_i, [_vp, _i, C.c_int32, _u32, _p(C.c_int32)]),
    "tbx_step1_frame": (_i, [_vp, _i, C.c_int32, _u32, _i, _p(C.c_int32), _p(_vp)]),
    "tbx_apply_input": (_i, [_vp, _i, _u32]),
    "tbx_get_scalars": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "tbx_render": (_i, [_vp, _vp, _i]),
    "tbx_render_device": (_i, [_vp, _vp, _i, _vp]),
    "tbx_device_identity": (_i, [_vp, _vp]),
    "tbx_render_step_synthetic": (_i, [_vp, _vp, _i, _u64, _u64, _u64, _u32, _vp]),
    "tbx_rollout_synthetic": (_i, [_vp, _i, _u64, _u64, _i, _u64, _u32, _vp]),
    "tbx_render_env": (_i, [_vp, _i, _vp, _i]),
    "tbx_get_state": (_i, [_vp, _i, _vp, _sz]),
    "tbx_set_state": (_i, [_vp, _i, _vp, _sz]),
    "tbx_get_states": (_i, [_vp, _i, _i, _vp, _sz]),
    "tbx_set_states": (_i, [_vp, _i, _i, _vp, _sz]),
    "tbx_get_config": (_i, [_vp, _vp, _sz]),
    "tbx_set_config": (_i, [_vp, _vp, _sz]),
    "tbx_query": (_i, [_vp, _i, _i, _p(C.c_int32), _i, _p(C.c_int32), _i]),
    "tbx_reduce_width": (_i, [_i, _i]),
    "tbx_edit": (_i, [_vp, _i, _vp, _i, _i, _vp]),
    "tbx_edit_device": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    "tbx_reduce": (_i, [_vp, _i, _vp, _i, _i, _vp]),
    "tbx_reduce_device": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    "tbx_agent_init": (_i, [_vp, _p(AgentConfig)]),
    "tbx_agent_set_noops": (_i, [_vp, _vp]),
    "tbx_agent_reset": (_i, [_vp, _vp]),
    "tbx_agent_episodes": (_i, [_vp, _vp, _vp, _vp]),
    "tbx_agent_step": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "tbx_agent_step_device": (_i, [_vp, _vp, _vp]),
    "tbx_agent_step_synthetic": (_i, [_vp, _u64, _u64, _u64, _vp]),
    "tbx_host_alloc": (_i, [_p(_vp), _sz]),
    "tbx_host_free": (_i, [_vp]),
    "tbx_agent_step_begin": (_i, [_vp, _vp, _p(AgentHostOut)]),
    "tbx_agent_step_end": (_i, [_vp]),
    "tbx_agent_fetch": (_i, [_vp, _p(AgentHostOut)]),
    "tbx_agent_ring_head": (_i, [_vp, _p(C.c_int32)]),
    "tbx_step_begin": (_i, [_vp, _vp, _u32, _p(StepHostOut)]),
    "tbx_step_end": (_i, [_vp]),
    "tbx_host_stack_push": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "tbx_gather_unique_id": (_i, [_vp, _sz]),
    "tbx_gather_init": (_i, [_vp, _i, _i, _i, _vp, _sz]),
    "tbx_gather": (_i, [_vp, _vp, _vp]),
    "tbx_gather_wait": (_i, [_vp, _vp]),
    "tbx_gather_host": (_i, [_vp, _vp]),
    "tbx_gather_reduce_max": (_i, [_vp, _p(C.c_double)]),
    "tbx_gather_nranks": (_i, [_vp]),
    "tbx_gather_library": (C.c_char_p, [_vp]),
    "tbx_gather_every": (_i, [_vp]),
    "tbx_gather_fill": (_i, [_vp]),
    "tbx_device_buffer": (_i, [_vp, _i, _p(_vp), _p(_sz)]),
    "tbx_set_option": (_i, [_vp, _i, _i]),
    "tbx_get_option": (_i, [_vp, _i, _p(_i)]),
    "tbx_sync": (_i, [_vp]),
}


def bind(lib, older_build=False):
    """Attach restype/argtypes for every symbol of the header; raises AttributeError when one is missing.  older_build=True (the A/B
    scripts, which load the previous round's library beside this one): entry points that build does not have yet are skipped."""
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if older_build:
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    return lib
