"""The C-ABI boundary: the product library loads on a machine without a GPU, exports every symbol that
include/toybox_amd.h declares, agrees with the ctypes mirror on record sizes, and fails loudly
(TBX_E_NO_DEVICE, no CPU fallback) when asked to compute without a GPU."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, has_gpu
from toybox_amd import Engine, ToyboxAmdError, _abi


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "toybox_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tbx_[a-z_0-9]+)\s*\(", text)))


def test_header_and_ctypes_mirror_agree():
    assert _declared_symbols() == sorted(_abi.PROTOTYPES.keys())


@pytest.mark.parametrize("which", ["hip", "oracle"])
def test_library_exports_every_symbol(which, hip_lib, oracle_lib):
    lib = hip_lib if which == "hip" else oracle_lib
    for name in _declared_symbols():
        assert hasattr(lib, name), "%s does not export %s" % (which, name)
    assert lib.tbx_abi_version() == _abi.ABI_VERSION


@pytest.mark.parametrize("which", ["hip", "oracle"])
def test_record_sizes(which, hip_lib, oracle_lib):
    lib = hip_lib if which == "hip" else oracle_lib
    for gid, t in _abi.STATE_TYPES.items():
        assert lib.tbx_state_size(gid) == C.sizeof(t)
    for gid, t in _abi.CONFIG_TYPES.items():
        assert lib.tbx_config_size(gid) == C.sizeof(t)


@pytest.mark.parametrize("which", ["hip", "oracle"])
def test_static_metadata(which, hip_lib, oracle_lib):
    lib = hip_lib if which == "hip" else oracle_lib
    buf = (C.c_int32 * 18)()
    assert lib.tbx_legal_actions(_abi.GAME_BREAKOUT, buf, 18) == 4 and list(buf[:4]) == [0, 1, 3, 4]
    h, w = C.c_int(), C.c_int()
    assert lib.tbx_frame_dims(_abi.GAME_BREAKOUT, C.byref(h), C.byref(w)) == 0 and (h.value, w.value) == (160, 240)
    # ALE action names (toybox/envs/atari/constants.py:16-35) -> buttons
    names = ["NOOP", "FIRE", "UP", "RIGHT", "LEFT", "DOWN", "UPRIGHT", "UPLEFT", "DOWNRIGHT", "DOWNLEFT", "UPFIRE",
             "RIGHTFIRE", "LEFTFIRE", "DOWNFIRE", "UPRIGHTFIRE", "UPLEFTFIRE", "DOWNRIGHTFIRE", "DOWNLEFTFIRE"]
    for a, name in enumerate(names):
        want = 0
        for word, bit in (("UP", _abi.BTN_UP), ("DOWN", _abi.BTN_DOWN), ("LEFT", _abi.BTN_LEFT),
                          ("RIGHT", _abi.BTN_RIGHT), ("FIRE", _abi.BTN_BUTTON1)):
            if word in name:
                want |= bit
        assert lib.tbx_ale_action_to_buttons(a) == want
    assert lib.tbx_ale_action_to_buttons(18) < 0 and lib.tbx_ale_action_to_buttons(-1) < 0


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback(hip_lib):
    with pytest.raises(ToyboxAmdError) as ei:
        Engine("breakout", 4)
    assert ei.value.code == _abi.E_NO_DEVICE


def test_product_does_not_reference_oracle():
    """No file of the product package may import, load or name the oracle."""
    pkg = os.path.join(ROOT, "toybox_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", "Makefile")):
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in text and "orc_" not in text, os.path.join(dp, f)
