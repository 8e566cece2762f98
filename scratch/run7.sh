python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "astat" 2>&1 | tail -3
python scratch/einsum_sweep.py 2>&1 | grep -v amdgpu | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    print(d.get('queries'), (d.get('mode') or d.get('reference'))[:200], 'graph us', round(d['launch_ms']*1e3,1), 'eager us', round(d.get('launch_ms_eager_loop',0)*1e3,1), 'frac_mfma', round(d.get('frac_mfma_peak',0),3), 'GBs', round(d.get('GBs',0)))
"
