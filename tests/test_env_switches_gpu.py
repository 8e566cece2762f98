"""-m gpu: the A/B switches that select alternative kernels / paths of PARITY mode ("only the defaults are in
GPUTEST"). Each runs tests/env_switch_worker.py in a fresh process with the switch flipped: the smoke forward of the head vs the
oracle (1e-3 mask logits) and a ResNet-50 forward vs the module path."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = ['', 'CGG_X3=0', 'CGG_X3A=0', 'CGG_X3_STEM=0', 'CGG_FUSED_TAIL=0', 'CGG_MSDA_GENERIC=1',
            'CGG_EXACT_F32_LOGITS=0', 'CGG_XATTN_X3=0', 'CGG_X3_GSCALE=0']


@pytest.mark.parametrize('switch', SWITCHES)
def test_parity_mode_switch(dev, switch):
    env = dict(os.environ)
    if switch:
        k, v = switch.split('=', 1)
        env[k] = v
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'env_switch_worker.py')], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and 'env switch worker OK' in r.stdout, (switch, r.stdout[-1500:], r.stderr[-1500:])
