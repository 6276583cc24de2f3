"""A minimal GPT-style host around the fla `MHLA` drop-in: token embedding, pre-norm blocks of
[RMSNorm -> MHLA layer -> residual -> RMSNorm -> gated MLP -> residual], final norm and a tied-free LM head -- the shape of the
reference's GLA-family language model (mhla_nlp/fla/models/gla/modeling_gla.py:83-100 builds the attention the same way).
Plumbing for step-level numbers; stock PyTorch besides the attention layer.  The reference layer's mixing matrix has 32 chunks
(layers/mhla.py:196-200: 2048 tokens at chunk 64); `max_seq_len` sizes it for longer sequences (8192 -> 128 chunks, the
BASELINE.json configs[4] sequence length, through the drop-in layer's `max_chunks`)."""
import torch
import torch.nn.functional as F
from torch import nn

from ..modules import MHLA


class RMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def forward(self, x):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)).to(x.dtype) * self.weight


class GatedMLP(nn.Module):
    def __init__(self, dim, hidden_ratio=4):
        super().__init__()
        inter = 256 * ((int(dim * hidden_ratio * 2 / 3) + 255) // 256)
        self.gate_proj = nn.Linear(dim, inter * 2, bias=False)
        self.down_proj = nn.Linear(inter, dim, bias=False)

    def forward(self, x):
        g, y = self.gate_proj(x).chunk(2, dim=-1)
        return self.down_proj(F.silu(g) * y)


class Block(nn.Module):
    def __init__(self, dim, heads, expand_k, expand_v, layer_idx, max_chunks=32):
        super().__init__()
        self.attn_norm = RMSNorm(dim)
        self.attn = MHLA(mode="chunk", hidden_size=dim, expand_k=expand_k, expand_v=expand_v, num_heads=heads,
                         feature_map="relu", layer_idx=layer_idx, max_chunks=max_chunks)
        self.mlp_norm = RMSNorm(dim)
        self.mlp = GatedMLP(dim)

    def forward(self, x):
        x = x + self.attn(self.attn_norm(x))[0]
        return x + self.mlp(self.mlp_norm(x))


class GPT_MHLA(nn.Module):
    def __init__(self, vocab_size=32000, hidden_size=1024, num_layers=24, num_heads=4, expand_k=0.5, expand_v=1.0,
                 max_seq_len=2048):
        super().__init__()
        self.embeddings = nn.Embedding(vocab_size, hidden_size)
        max_chunks = max(32, (max_seq_len + 63) // 64)     # 32 = the reference layer's matrix; 128 for seq_len 8192 (config 5)
        self.layers = nn.ModuleList([Block(hidden_size, num_heads, expand_k, expand_v, i, max_chunks) for i in range(num_layers)])
        self.norm = RMSNorm(hidden_size)
        self.lm_head = nn.Linear(hidden_size, vocab_size, bias=False)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.normal_(m.weight, std=0.02)

    def forward(self, input_ids, labels=None):
        x = self.embeddings(input_ids)
        for blk in self.layers:
            x = blk(x)
        logits = self.lm_head(self.norm(x))
        if labels is None:
            return logits
        return F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]).float(), labels[:, 1:].reshape(-1))


def GPT_configs():
    return {"340M": dict(hidden_size=1024, num_layers=24, num_heads=4), "1.3B": dict(hidden_size=2048, num_layers=24, num_heads=4)}
