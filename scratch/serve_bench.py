"""End-to-end serving through tools/test.py (results on the HOST as numpy): bit-packed vs bool mask copies."""
import os, sys, importlib.util, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cgg_amd import synthetic
cfg = synthetic.model_config(num_things=65, num_stuff=0, num_unknown=17, num_queries=100, depth=50)
d = tempfile.mkdtemp()
f = os.path.join(d, 'cfg.py')
open(f, 'w').write('model = ' + repr(cfg) + '\n')
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
spec = importlib.util.spec_from_file_location('t', os.path.join(root, 'tools', 'test.py'))
drv = importlib.util.module_from_spec(spec); spec.loader.exec_module(drv)
for extra in ([], ['--mask-bits']):
    print(extra, flush=True)
    drv.main([f, 'none', '--num-images', '48', '--synthetic', '1024'] + extra)
