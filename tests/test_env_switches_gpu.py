"""-m gpu: the A/B switches that select alternative kernels / paths of PARITY mode ("only the defaults are in
GPUTEST"). Each runs tests/env_switch_worker.py in a fresh process with the switch flipped: the smoke forward of the head vs the
oracle (1e-3 mask logits) and a ResNet-50 forward vs the module path."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = ['', 'CGG_X3=0', 'CGG_X3A=0', 'CGG_X3_STEM=0', 'CGG_FUSED_TAIL=0', 'CGG_MSDA_GENERIC=1', 'CGG_MERGED_PROJ=0', 'CGG_TAIL_WAVES=4',
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


# the training-side switches (none was ever exercised off-default): each alternative path must give the oracle's
# losses and gradients too. One parity-mode forward_train + backward at batch 4 x 512^2 per switch (tests/env_switch_train_worker.py).
TRAIN_SWITCHES = ['', 'CGG_X3_TRAIN=0', 'CGG_X3_WGRAD=0', 'CGG_X3_LAYER_NODES=0', 'CGG_X3_FPN_ROWS=0', 'CGG_MSDA_BWD_2S=0',
                  'CGG_X3_GENERATOR=0', 'CGG_XATTN_X3_TRAIN=0', 'CGG_FUSED_TRAIN_LN=0', 'CGG_FUSED_TRAIN_MSDA=0', 'CGG_X3A=0', 'CGG_POINT_LOGITS_X3=0', 'CGG_XATTN_X3_BWD=0']


@pytest.fixture(scope='module')
def oracle_cache(tmp_path_factory):
    return str(tmp_path_factory.mktemp('envswitch') / 'oracle.pt')


@pytest.mark.parametrize('switch', TRAIN_SWITCHES)
def test_training_switch(dev, switch, oracle_cache):
    env = dict(os.environ)
    if switch:
        k, v = switch.split('=', 1)
        env[k] = v
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'env_switch_train_worker.py'), oracle_cache], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'env switch train worker OK' in r.stdout, (switch, r.stdout[-1500:], r.stderr[-2500:])
