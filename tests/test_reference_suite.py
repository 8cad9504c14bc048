"""The reference's OWN test-suite, unmodified, against this repo's host layer (build container only: it needs
/root/reference).  A `ctoybox`-named shim (tests/shim/ctoybox) provides the module the reference imports; engines
are backed by the CPU oracle because this container has no GPU.  34 tests: schema / strict JSON decode, state
round-trips through the engine, dirty-state tracking, Breakout and Amidar interventions, equality modes and
property paths."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REF = os.environ.get("TOYBOX_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "test", "interventions")),
                                reason="the reference tree is only present in the build container")


def _env():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests", "shim"), ROOT, REF])
    env["PYTHONDONTWRITEBYTECODE"] = "1"     # /root/reference is read-only
    return env


def test_reference_intervention_suite(oracle_lib):
    for attempt in range(2):
        p = subprocess.run([sys.executable, "-m", "unittest", "discover", "-s", os.path.join(REF, "test", "interventions"),
                            "-t", REF], cwd="/tmp", env=_env(), capture_output=True, text=True, timeout=900)
        tail = (p.stdout + p.stderr)[-3000:]
        # the reference's test_random_starts draws an unseeded random tile (interventions/amidar.py:373-375) and fails when it
        # draws the player's own: one in 992 runs.  Only that one failure is retried.
        if p.returncode == 0 or "failures=1" not in tail or "FAIL: test_random_starts" not in tail:
            break
    assert p.returncode == 0, tail
    assert "Ran 34 tests" in tail and "OK" in tail, tail


def test_reference_smoke_script(oracle_lib):
    """scripts/utils/test_games.py (CI entry of the reference, check.sh:5 -> unit_tests.sh:5): config/state JSON,
    legal actions, set_seed(1234), 100 NOOPs, RGB frame, write_config_json / write_state_json, for all three games."""
    p = subprocess.run([sys.executable, os.path.join(REF, "scripts", "utils", "test_games.py")], cwd="/tmp", env=_env(),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    for game in ("amidar", "breakout", "space_invaders"):
        assert "TEST  %s" % game in p.stdout


def test_reference_space_invaders_selftest(oracle_lib):
    """unit_tests.sh:8 -> python -m toybox.interventions.space_invaders: lives = 1 marks dirty_state."""
    p = subprocess.run([sys.executable, "-m", "toybox.interventions.space_invaders"], cwd="/tmp", env=_env(),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
