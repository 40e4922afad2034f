"""More than 2 048 points per alignment at random sizes: the team kernels optimize picks against the one-CU kernels (tools/fuzz_large_n.py).
Seed 0 holds the case that found an out-of-slot read of empty team members (8 722 points on 16 CUs, last slot of the handle)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,trials", [(0, 10), (5, 8)])
def test_large_point_counts_random(gpu, seed, trials):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_large_n.py"), str(seed), str(trials)], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "0 disagreements" in p.stdout
