import cProfile, pstats, sys, os, io
sys.argv = ['bench.py', '--mode', 'train', '--steps', '2', '--warmup', '2']
sys.path.insert(0, '/root/repo')
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('/root/repo/bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats('cumulative')
ps.print_stats(70)
print(s.getvalue()[:14000])
