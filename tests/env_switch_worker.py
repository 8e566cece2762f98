"""One configuration of tests/test_env_switches_gpu.py, in a FRESH process (the switches are read at import): the smoke forward of
the head vs the oracle (pixel decoder stream, encoder tail, query decoder, mask logits, post-processing) and a ResNet-50 forward in
parity mode vs the plain module path."""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import __graft_entry__ as ge
    ge.smoke()
    import cgg_amd  # noqa: F401
    from cgg_amd import ops, registry, runtime
    dev = torch.device('cuda', 0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        bb = registry.build_backbone(dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1,
                                          norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='pytorch'))
        bb.init_weights()
    bb = bb.to(dev).eval()
    with torch.no_grad():
        for m in bb.modules():                       # zero-initialised last BNs would make every block an identity
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.fill_(1.0)
    img = torch.randn(1, 3, 160, 128, device=dev)
    with runtime.precision_scope('fp32'):
        with torch.no_grad():
            fast = [ops.x3a_to_f32(f) if ops.is_x3a(f) else f.float() for f in bb(img)]
        ref = [f.detach().float() for f in bb(img)]          # autograd enabled: the plain module path
    for a, b in zip(fast, ref):
        err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6)
        assert err <= 1e-4, err
    print('env switch worker OK:', ' '.join(f'{k}={v}' for k, v in os.environ.items() if k.startswith('CGG_')), flush=True)


if __name__ == '__main__':
    main()
