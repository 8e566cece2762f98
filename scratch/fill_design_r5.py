"""Fills the round-5 numbers of DESIGN.md section 11 / README.md from gpurun_out/r5/prof (after scratch/collect_profiles_r5.sh +
scratch/publish_profiles_r5.py). Idempotent: the numbers sit between <!--r5:key--> ... <!--/r5--> markers."""
import json, os, re
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
P = os.path.join(R, 'gpurun_out', 'r5', 'prof')
line = json.loads(open(os.path.join(P, 'bench_line.json')).read().strip().splitlines()[-1])
ex = line['extra']
vals = {}
vals['headline'] = f"{line['value']:.1f}"
vals['headline_ms'] = f"{line['ms_per_step']:.2f}"
vals['headline_frac'] = f"{line['roofline']['frac']:.3f}"
vals['bf16'] = f"{line['bf16_mode']['value']:.0f}" if isinstance(line.get('bf16_mode'), dict) and 'value' in line['bf16_mode'] else '?'
ts = ex.get('train_step', {})


def find(d, *keys):
    for k in keys:
        if isinstance(d, dict) and k in d:
            d = d[k]
        else:
            return None
    return d


def step_table(name):
    p = os.path.join(P, name)
    if not os.path.exists(p):
        return None
    t = open(p).read()
    m = re.search(r'step \(eager, serial\): (\d+) launches, ([\d.]+) us of kernel time', t)
    s = re.search(r'hand-written \(cgg_\*\) kernels: ([\d.]+) us per step = ([\d.]+) %', t)
    ks = {}
    for l in t.splitlines():
        mm = re.match(r'\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\S.*)', l)
        if mm:
            ks[mm.group(4)[:60]] = (float(mm.group(1)), float(mm.group(2)))
    return dict(launches=int(m.group(1)), kernel_ms=float(m.group(2)) / 1e3, cgg_share=float(s.group(2)), kernels=ks) if m and s else None


for w in ('cfg2', 'cfg3'):
    for p in ('fp32', 'bf16'):
        t = step_table(f'{w}_train_step_kernels_{p}.txt')
        if t:
            vals[f'{w}_{p}_kernel_ms'] = f"{t['kernel_ms']:.1f}"
            vals[f'{w}_{p}_cgg'] = f"{t['cgg_share']:.1f}"
            vals[f'{w}_{p}_launches'] = str(t['launches'])
            for key, pat in (('msda_sorted', 'cgg_msda_bwd_sorted'), ('msda_gather', 'cgg_msda_bwd_gather4'), ('gemm_x3', 'cgg_gemm_x3_kernel<false, 2, 2>'),
                             ('wgrad256', 'cgg_wgrad_x3_kernel<256>'), ('absmax', 'cgg_absmax'), ('add', 'CUDAFunctor_add<float>, std::array')):
                for k, (us, calls) in t['kernels'].items():
                    if pat in k:
                        vals[f'{w}_{p}_{key}_ms'] = f"{us / 1e3:.1f}"
                        break
print(json.dumps({k: v for k, v in vals.items()}, indent=1))
for k in ('configs[3]', 'configs[4]', 'train_step'):
    v = ex.get(k)
    if isinstance(v, dict):
        print(k, {kk: vv for kk, vv in v.items() if not isinstance(vv, (dict, list))})
