"""ctypes mirror of include/toybox_amd.h: POD records, constants and prototype binding.

Every Structure here must match the C header byte for byte; tests/test_abi.py checks the sizes
against tbx_state_size()/tbx_config_size() of the loaded library and that every declared symbol
is exported.
"""
import ctypes as C

ABI_VERSION = 1

GAME_BREAKOUT, GAME_AMIDAR, GAME_SPACE_INVADERS = 0, 1, 2
GAME_IDS = {"breakout": GAME_BREAKOUT, "amidar": GAME_AMIDAR, "space_invaders": GAME_SPACE_INVADERS,
            "spaceinvaders": GAME_SPACE_INVADERS}
GAME_NAMES = {GAME_BREAKOUT: "breakout", GAME_AMIDAR: "amidar", GAME_SPACE_INVADERS: "space_invaders"}

OK, E_INVALID, E_NO_DEVICE, E_NOMEM, E_UNSUPPORTED, E_ACTION = 0, -1, -2, -3, -4, -5

BTN_LEFT, BTN_RIGHT, BTN_UP, BTN_DOWN, BTN_BUTTON1, BTN_BUTTON2 = 1, 2, 4, 8, 16, 32
STEP_AUTO_RESET = 1
BUF_REWARD, BUF_DONE, BUF_LIVES, BUF_SCORE, BUF_FRAME, BUF_PACKED = 0, 1, 2, 3, 4, 5

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


STATE_TYPES = {GAME_BREAKOUT: BreakoutState}
CONFIG_TYPES = {GAME_BREAKOUT: BreakoutConfig}

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
    "tbx_get_sim_rng": (_i, [_vp, _i, _p(_u64)]),
    "tbx_set_sim_rng": (_i, [_vp, _i, _p(_u64)]),
    "tbx_new_game": (_i, [_vp, _vp]),
    "tbx_step": (_i, [_vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "tbx_step_device": (_i, [_vp, _vp, _u32, _vp]),
    "tbx_step_synthetic": (_i, [_vp, _u64, _u64, _u64, _u32, _vp]),
    "tbx_apply_input": (_i, [_vp, _i, _u32]),
    "tbx_get_scalars": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "tbx_render": (_i, [_vp, _vp, _i]),
    "tbx_render_device": (_i, [_vp, _vp, _i, _vp]),
    "tbx_render_env": (_i, [_vp, _i, _vp, _i]),
    "tbx_get_state": (_i, [_vp, _i, _vp, _sz]),
    "tbx_set_state": (_i, [_vp, _i, _vp, _sz]),
    "tbx_get_config": (_i, [_vp, _vp, _sz]),
    "tbx_set_config": (_i, [_vp, _vp, _sz]),
    "tbx_device_buffer": (_i, [_vp, _i, _p(_vp), _p(_sz)]),
    "tbx_sync": (_i, [_vp]),
}


def bind(lib):
    """Attach restype/argtypes for every symbol of the header; raises AttributeError when one is missing."""
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
