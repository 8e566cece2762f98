"""`V2lTranformHead` (open_set/models/heads/v2l_head.py:5-26): registered by the reference under this (sic) name; the
shipped head builds a plain `nn.Linear` instead (mask2former_head.py:218-219), so this class is only reachable from a
user config. Same constructor keys and the same forward (ONLY the first linear is applied, :24-26); unlike the
reference's class it calls `nn.Module.__init__`, without which the reference's version cannot be constructed."""
import torch.nn as nn

from .registry import HEADS


@HEADS.register_module()
class V2lTranformHead(nn.Module):

    def __init__(self, in_dim=256, hidden_dim=1024, out_dim=768, n_layers=1):
        super().__init__()
        self.n_layers = n_layers
        self.linears = nn.ModuleList()
        if n_layers == 1:
            self.linears.append(nn.Linear(in_dim, out_dim))
        else:
            dims = [in_dim] + [hidden_dim] * (n_layers - 1) + [out_dim]
            for a, b in zip(dims[:-1], dims[1:]):
                self.linears.append(nn.Linear(a, b))
        self.relu = nn.ReLU()

    def forward(self, x):
        return self.linears[0](x)
