"""bench.py's einsum_mfma_target object alone (mask-logit einsum at Q = 100 / 200, stored-logits and consumer-fused forms)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import cgg_amd  # noqa: F401
dev = torch.device('cuda')
for r in bench.einsum_q_sweep(dev, 2, 1024, 1024):
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))
