"""A fixed stretch of every mode of tests/fuzz_parity.py (the randomized sweep that found three defects in round 6: DESIGN.md section 13)
as part of the suite: seeds and case counts are fixed, so every run covers the same cases.  Each mode runs in a child process (one engine
per sequence on purpose -- arenas are re-carved over what earlier geometries left -- and a device fault must not take the session with it)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NEEDS_REF = {"variants", "graph"}


@pytest.mark.parametrize("mode,seed,cases", [("slic", 11, 60), ("variants", 11, 80), ("warm", 11, 120), ("cca", 11, 400),
                                             ("lsc", 11, 60), ("pipeline", 11, 150), ("graph", 11, 60)])
def test_fixed_stretch_of_the_randomized_sweep(mode, seed, cases):
    if mode in NEEDS_REF and not os.path.exists(os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libfslic_ref.so")):
        pytest.skip("oracle/_ref (the reference build) is not present")
    args = [sys.executable, os.path.join(HERE, "fuzz_parity.py"), str(seed), "600"] + ([] if mode == "slic" else [mode])
    r = subprocess.run(args, env=dict(os.environ, FUZZ_CASES=str(cases)), capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert "fuzz_parity" in r.stdout.splitlines()[-1], tail
