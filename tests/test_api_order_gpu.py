"""Calls whose device work is not waited for (frame / depth hand-over, residual mirror, mapped outputs) in random orders: a solve on
a long-lived handle stays bit-identical to the same solve on a fresh handle (tools/fuzz_api_order.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_random_call_orders_match_fresh_handles(gpu, seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_api_order.py"), str(seed), "120"], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "0 disagreements" in p.stdout
