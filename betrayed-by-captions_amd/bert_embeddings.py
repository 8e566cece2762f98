"""Frozen BERT word embeddings + LayerNorm (open_set/models/utils/bert_embeddings.py:4-14).

The reference copies `word_embeddings` (30522 x 768, padding_idx 0) and `LayerNorm` (eps 1e-12) out of
HF `bert-base-uncased`. Offline there are no BERT weights, so besides the reference constructor
(`BertEmbeddings(bert_model)`) a `from_config(...)` constructor builds the same modules with a
synthetic table; `state_dict` keys are identical either way.
"""
import torch
from torch import nn


class BertEmbeddings(nn.Module):

    def __init__(self, bert_model=None, vocab_size=30522, hidden_size=768, pad_token_id=0,
                 layer_norm_eps=1e-12):
        super().__init__()
        if bert_model is not None:
            cfg = bert_model.config
            vocab_size, hidden_size = cfg.vocab_size, cfg.hidden_size
            pad_token_id, layer_norm_eps = cfg.pad_token_id, cfg.layer_norm_eps
        self.word_embeddings = nn.Embedding(vocab_size, hidden_size, padding_idx=pad_token_id)
        self.LayerNorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        if bert_model is not None:
            self.word_embeddings.load_state_dict(bert_model.embeddings.word_embeddings.state_dict())
            self.LayerNorm.load_state_dict(bert_model.embeddings.LayerNorm.state_dict())

    @classmethod
    def synthetic(cls, seed=0, vocab_size=30522, hidden_size=768):
        """N(0, 0.04^2) table, LN(gamma=1, beta=0): the synthetic stand-in used by bench / tests."""
        m = cls(None, vocab_size, hidden_size)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            m.word_embeddings.weight.copy_(torch.randn(vocab_size, hidden_size, generator=g) * 0.04)
            m.word_embeddings.weight[0].zero_()
        return m

    def forward(self, ids, normalize=True):
        e = self.word_embeddings(ids)
        return self.LayerNorm(e) if normalize else e
