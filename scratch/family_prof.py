"""Kernel time by family over the last `frac` of a trace."""
import csv, collections, sys, glob, os, re
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows = rows[int(len(rows) * (1 - frac)):]
fam = collections.defaultdict(lambda: [0, 0])
def family(n):
    if 'msda_bwd' in n: return 'cgg msda bwd'
    if 'cgg_' in n: return 'cgg other'
    if 'conv_bwd_data' in n or 'igemm_bwd' in n or 'bwd_data' in n: return 'conv bwd data'
    if 'bwd_weight' in n or 'igemm_wrw' in n or 'wrw' in n: return 'conv bwd weight'
    if 'conv_fwd' in n or 'igemm_fwd' in n: return 'conv fwd'
    if n.startswith('Cijk'): return 'GEMM (hipBLASLt)'
    if 'batched_gemm' in n or 'ck::' in n or '2ck' in n: return 'CK other'
    if 'grid_sampler' in n: return 'grid_sample'
    if 'layer_norm' in n or 'GammaBeta' in n or 'GradInput' in n: return 'LayerNorm'
    if 'SubTensor' in n or 'transpose' in n.lower(): return 'MIOpen tensor ops'
    if 'elementwise' in n or 'Functor' in n or 'copy' in n.lower() or 'fill' in n.lower(): return 'elementwise/copy/fill'
    if 'reduce' in n or 'softmax' in n.lower(): return 'reduce/softmax'
    if 'adam' in n.lower() or 'multi_tensor' in n: return 'optimizer'
    return 'other'
for r in rows:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    k = family(r['Kernel_Name'])
    fam[k][0] += d; fam[k][1] += 1
tot = sum(v[0] for v in fam.values())
wall = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print('kernels %d  sum %.1f ms  wall %.1f ms' % (len(rows), tot / 1e6, wall / 1e6))
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print('%9.2f ms %5.1f%% x%-6d %s' % (v[0] / 1e6, 100.0 * v[0] / tot, v[1], k))
