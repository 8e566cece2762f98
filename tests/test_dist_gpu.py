"""-m gpu: the N > 1 training path with the REAL model on the device -- two fresh child processes, both on device 0, gloo on
127.0.0.1 (tests/dist_gpu_worker.py). The CPU twin with toy modules is tests/test_dist_gloo.py."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_train_step_on_one_device(dev):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dist_gpu_worker.py')], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f'rank {r} failed:\n{o[-3000:]}'
    assert all('step 1' in o for o in outs), outs
    print('\n'.join(l for o in outs for l in o.splitlines() if l.startswith('rank')))
