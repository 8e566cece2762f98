for a in 0 1 2 3 4 8 12 5; do echo -n "abl=$a "; CGG_MLA_ABL=$a python scratch/einsum_sweep.py 2>&1 | grep astat | python -c "
import sys, json
print([ (json.loads(l)['queries'], round(json.loads(l)['launch_ms']*1e3,1)) for l in sys.stdin])"; done
