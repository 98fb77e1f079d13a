"""A short run of tools/probes/fuzz_env.py inside the GPU suite: random itscp environments (grid, lanes, lane length, speed limit, episode
and signal length, mode, training / evaluation, per-vehicle attributes) through ItscpEnv.step -- whatever path the environment picks --
against the CPU oracle on the environment's own tables.  The long runs are profiles/r06z_fuzz_env*.log."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [5, 6])
def test_random_environments_match_the_oracle(cuda, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probes", "fuzz_env.py"), "24", str(seed)], capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.strip().splitlines()[-6:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("environments: 24") and "mismatches: 0" in last, tail
    print(last)
