"""A/B of the encoder-stream MSDeformAttn forward at configs[1] shapes: head-major value layout (what the step runs),
quad-shared-tap kernel on row-layout values, vector-row kernel
(CGG_MSDA_V1=1) and the generic kernel (CGG_MSDA_GENERIC=1), each in its own process; vector-row == generic bit for bit,
quad-shared within one bf16 ulp (different f32 summation order).  python scratch/msda_ab.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, ROOT)
    import cgg_amd
    from cgg_amd import ops
    dev = 'cuda'
    B = 2
    g = torch.Generator().manual_seed(0)
    shapes = [(32, 32), (64, 64), (128, 128)]; starts = [0, 1024, 5120]; N = 21504
    raw = torch.randn(B, N, 288, generator=g); raw[..., :192] *= 2.0
    ref = []
    for h, w in shapes:
        ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing='ij')
        ref.append(torch.stack([(xs.flatten() + .5) / w, (ys.flatten() + .5) / h], -1))
    ref = torch.cat(ref).to(dev)
    raw16 = raw.to(dev).to(torch.bfloat16)
    v = torch.randn(B, N, 8, 32, generator=g).to(dev).to(torch.bfloat16)
    hm = bool(os.environ.get('MSDA_AB_HEAD_MAJOR'))
    if hm:
        v = v.permute(0, 2, 1, 3).contiguous()          # (B, 8, N, 32): what encoder_proj(value_head_major=True) writes
    run = lambda: ops.msda_forward_fused_bf16(v, shapes, starts, raw16, ref, 4, head_major=hm)
    for _ in range(10):
        out = run()
    torch.cuda.synchronize()
    flush = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    for mode in ('warm', 'flushed'):
        ts = []
        for _ in range(50):
            if mode == 'flushed':
                flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = run(); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print('%s %s: median %.1f us, min %.1f us' % (sys.argv[1], mode, ts[len(ts) // 2], ts[0]), flush=True)
    torch.save(out.cpu(), sys.argv[2])
else:
    import torch
    for name, env in (('head-major', {'MSDA_AB_HEAD_MAJOR': '1'}), ('quad-shared', {}), ('vector-row', {'CGG_MSDA_V1': '1'}),
                      ('generic', {'CGG_MSDA_GENERIC': '1'})):
        subprocess.run([sys.executable, os.path.abspath(__file__), name, f'/tmp/msda_{name}.pt'], env=dict(os.environ, **env), check=True)
    a, b, c = torch.load('/tmp/msda_vector-row.pt'), torch.load('/tmp/msda_generic.pt'), torch.load('/tmp/msda_quad-shared.pt')
    print('vector-row vs generic: bit-identical:', torch.equal(a, b), ' max |diff|:', float((a.float() - b.float()).abs().max()))
    d = (c.float() - b.float()).abs()
    ulp = b.float().abs().clamp(min=1e-3) * 2.0 ** -7          # one bf16 ulp is <= 2^-7 of the value
    print('quad-shared vs generic: max |diff| %.3e, elements differing %.4f %%, max diff in bf16 ulps %.2f' %
          (float(d.max()), 100.0 * float((d > 0).float().mean()), float((d / ulp).max())))
    e = torch.load('/tmp/msda_head-major.pt')
    print('head-major values + 16 queries of one head per wavefront vs quad-shared: bit-identical:', torch.equal(e, c))
