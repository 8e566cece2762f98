"""-m gpu: randomized shape sweeps of round 4's new kernels against float64 definitions -- the cross-attention core on the f16 x 3
contraction (transpose reads), the x3 weight gradient, the MSDeformAttn forward's three query -> lane mappings -- at sizes the
deterministic tests do not enumerate (odd key counts, query counts around the 32-row tiles, row counts around the split / chunk
boundaries, pyramids that select each mapping)."""
import math
import random

import pytest
import torch

import cgg_amd  # noqa: F401
from cgg_amd import ops

pytestmark = pytest.mark.gpu


def _err(got, want64):
    return (got.detach().cpu().double() - want64).abs().max().item()


@pytest.mark.parametrize('seed', range(6))
def test_xattn_x3_random_shapes(dev, seed):
    rnd = random.Random(100 + seed)
    g = torch.Generator().manual_seed(100 + seed)
    B, Q, S = rnd.choice([1, 2, 3]), rnd.choice([1, 31, 32, 33, 64, 100, 127, 128]), rnd.choice([1, 17, 63, 64, 65, 200, 1023, 2500])
    E, H = 256, 8
    q = torch.randn(B, Q, E, generator=g) * rnd.choice([0.3, 1.0, 3.0])
    kv = torch.randn(B, S, 2 * E, generator=g) * rnd.choice([0.3, 1.0, 2.0])
    mask = torch.rand(B, Q, S, generator=g) < rnd.choice([0.0, 0.5, 0.95])
    mask[:, :, 0] = False                                            # no fully masked row (NaN by definition: covered elsewhere)
    logits = torch.einsum('bqhd,bshd->bhqs', q.double().view(B, Q, H, 32), kv[..., :E].double().reshape(B, S, H, 32)) / math.sqrt(32)
    logits = logits.masked_fill(mask[:, None], float('-inf'))
    want = torch.einsum('bhqs,bshd->bqhd', logits.softmax(-1), kv[..., E:].double().reshape(B, S, H, 32)).reshape(B, Q, E)
    words = (S + 31) // 32
    padded = torch.zeros(B, Q, words * 32, dtype=torch.bool)
    padded[..., :S] = mask
    w = (padded.view(B, Q, words, 32).long() << torch.arange(32)).sum(-1)
    bits = w.where(w < 2 ** 31, w - 2 ** 32).to(torch.int32).to(dev)
    got = ops.masked_xattn(q.to(dev), kv.to(dev), bits, H)
    assert _err(got, want) <= 3e-5 * max(1.0, want.abs().max().item()), (B, Q, S)


@pytest.mark.parametrize('seed', range(6))
def test_wgrad_x3_random_shapes(dev, seed):
    rnd = random.Random(200 + seed)
    g = torch.Generator().manual_seed(200 + seed)
    M = rnd.choice([1, 31, 32, 33, 255, 256, 257, 1000, 8191, 20000])
    N, K = 4 * rnd.randint(1, 80), 4 * rnd.randint(1, 80)
    dy = torch.randn(M, N, generator=g) * rnd.choice([0.01, 1.0, 30.0])
    x = torch.randn(M, K, generator=g) * rnd.choice([0.01, 1.0, 30.0])
    want = dy.double().t() @ x.double()
    f32_err = ((dy.t() @ x).double() - want).abs().max().item()
    got = ops.wgrad_x3(dy.to(dev), x.to(dev))
    assert _err(got, want) <= 4 * f32_err + 3e-7 * want.abs().max().item() + 1e-30, (M, N, K)


@pytest.mark.parametrize('shapes', [[(16, 24), (32, 48), (64, 96)], [(8, 8), (8, 16), (16, 16)], [(12, 20), (24, 40), (48, 80)],
                                    [(3, 5), (6, 10), (12, 20)]])
def test_msda_forward_mappings_random(dev, shapes):
    """pyramids that select the 8 x 8-block mapping, the 4 x 4-tile mapping and the strip mapping; offsets of several pixels."""
    from oracle import ops as ref
    g = torch.Generator().manual_seed(sum(h * w for h, w in shapes))
    B, H, D, L, P = 2, 8, 32, 3, 4
    starts, Nv = [], 0
    for h, w in shapes:
        starts.append(Nv)
        Nv += h * w
    value = torch.randn(B, Nv, H, D, generator=g)
    raw = torch.randn(B, Nv, H * L * P * 3, generator=g)
    raw[..., :H * L * P * 2] *= 3.0
    refp = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing='ij'), -1).flip(-1)
                      .reshape(-1, 2) for h, w in shapes], 0)
    off = raw[..., :H * L * P * 2].view(B, Nv, H, L, P, 2)
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
    loc = refp[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
    aw = raw[..., H * L * P * 2:].view(B, Nv, H, L * P).softmax(-1).view(B, Nv, H, L, P)
    want = ref.msda_core(value, torch.tensor(shapes), loc, aw)
    got = ops.msda_forward_fused(value.to(dev), shapes, starts, raw.to(dev), refp.to(dev), P).cpu()
    assert (got - want).abs().max().item() <= 1e-4
