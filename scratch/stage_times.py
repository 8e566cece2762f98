import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
class A: pass
args = A(); args.queries = 100
dev = torch.device('cuda')
import cgg_amd
from cgg_amd import runtime, synthetic
runtime.set_precision('bf16')
cfg, model = bench.build_model(args, dev)
B, H, W = 2, 1024, 1024
img = torch.randn(B, 3, H, W, device=dev)
metas = synthetic.img_metas(B, H, W)
def graph_time(fn, n=50):
    with torch.no_grad():
        for _ in range(3): out = fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=s):
            out = fn()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out, g
t_bb, feats, g1 = graph_time(lambda: model.extract_feat(img))
t_enc, enc, g2 = graph_time(lambda: model.panoptic_head._encode(feats))
t_dec, res, g3 = graph_time(lambda: model.stage_decode(enc, metas, rescale=True, device_results=True))
print('graph replay alone: backbone %.2f ms | head encode %.2f ms | decode+post %.2f ms | sum %.2f' % (t_bb, t_enc, t_dec, t_bb + t_enc + t_dec))
