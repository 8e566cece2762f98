"""End-to-end serving through tools/test.py with results on the HOST: COCO RLE (evaluation format; pinned staging + async
copies + C++ encoder), bit-packed planes, bool arrays (the reference's format). One process per mode (captured pipelines and
pinned staging of a previous mode otherwise stay resident and slow the next one down).  python scratch/serve_bench.py [images]"""
import os, subprocess, sys, importlib.util, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == '--one':
    from cgg_amd import synthetic
    cfg = synthetic.model_config(num_things=65, num_stuff=0, num_unknown=17, num_queries=100, depth=50)
    d = tempfile.mkdtemp()
    f = os.path.join(d, 'cfg.py')
    open(f, 'w').write('model = ' + repr(cfg) + '\n')
    spec = importlib.util.spec_from_file_location('t', os.path.join(ROOT, 'tools', 'test.py'))
    drv = importlib.util.module_from_spec(spec); spec.loader.exec_module(drv)
    drv.main([f, 'none', '--num-images', sys.argv[2], '--synthetic', '1024'] + sys.argv[3:])
else:
    n = sys.argv[1] if len(sys.argv) > 1 else '64'
    for extra in (['--rle'], ['--mask-bits'], []):
        subprocess.run([sys.executable, os.path.abspath(__file__), '--one', n] + extra, check=True)
