"""Drop-in for the fla attention layer `MHLA` (mhla_nlp/fla/layers/mhla.py:29-365).

Same constructor, `forward(hidden_states, attention_mask, past_key_values, use_cache,
output_attentions, **kw) -> (o, None, past_key_values)` and parameter names
(`q_proj/k_proj/v_proj/g_proj/o_proj.weight`, `mixing_matrix [32,32,1,1,1,1]` (side = `max_chunks`),
`g_norm_swish_gate.weight`).  The causal operator and the per-head RMSNorm x swish gate run as HIP
kernels; rotary is plain tensor math (NeoX half rotation, rotary.py:20-32) -- no Triton.

Deviations, all documented in SURVEY.md: sequences of <= 64 tokens use the single-chunk case of the
chunk operator (the reference's token-recurrent form equals it only on the first chunk and ignores
its initial state).  `use_short_conv=True` (off in the shipped configuration; the reference's
ShortConvolution is a sibling fla module outside the MHLA hot path) is served by a plain-PyTorch
`ShortConvolution` with the reference's parameters and cache protocol (no HIP kernel: not on the path).
"""
import warnings
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from ..ops import featmap_rotary, mhla_causal, mhla_causal_normgate, naive_recurrent_mhla, rmsnorm_gate
from ..weights import causal_mixing_init


class FusedRMSNormGated(nn.Module):
    """Parameter holder + HIP kernel call (fused_norm_gate.py:997-1058)."""

    def __init__(self, hidden_size, elementwise_affine=True, eps=1e-5, activation="swish"):
        super().__init__()
        if activation not in ("swish", "silu"):
            raise ValueError(f"Unsupported activation: {activation}")
        self.hidden_size = hidden_size
        self.eps = eps
        self.activation = activation
        if elementwise_affine:
            self.weight = nn.Parameter(torch.ones(hidden_size))
        else:
            self.register_parameter("weight", None)
        self.register_parameter("bias", None)

    def forward(self, x, g):
        return rmsnorm_gate(x, g, self.weight, self.eps)


class RotaryEmbedding(nn.Module):
    """NeoX-style rotary, non-interleaved, base 10000 (rotary.py:330-431): fp32 inv_freq, cos/sin cached
    in the activation dtype."""

    def __init__(self, dim: int, base: float = 10000.0):
        super().__init__()
        self.dim = dim
        self.base = base
        self._cache: Optional[Tuple] = None

    def _tables(self, seqlen: int, device, dtype):
        c = self._cache
        if c is None or c[0] < seqlen or c[1] != device or c[2] != dtype:
            inv_freq = 1.0 / (self.base ** (torch.arange(0, self.dim, 2, device=device, dtype=torch.float32) / self.dim))
            t = torch.arange(seqlen, device=device, dtype=torch.float32)
            fr = torch.outer(t, inv_freq)
            c = (seqlen, device, dtype, torch.cos(fr).to(dtype), torch.sin(fr).to(dtype))
            self._cache = c
        return c[3], c[4]

    def forward(self, q, k, seqlen_offset: int = 0, max_seqlen: Optional[int] = None, cu_seqlens=None, positions=None):
        """`positions` [T] (long): the rotary position of every token row -- packed sequences restart per sequence
        (rotary.py:68-72 with cu_seqlens), padded decoding adds each sequence's own offset (layers/mhla.py:305-309)."""
        T = q.shape[1]
        if positions is not None:
            cos, sin = self._tables(max(max_seqlen or 0, T + int(seqlen_offset if isinstance(seqlen_offset, int) else 0)), q.device, q.dtype)
            cos, sin = cos.index_select(0, positions)[None, :, None, :], sin.index_select(0, positions)[None, :, None, :]
        else:
            cos, sin = self._tables(max(max_seqlen or 0, T + seqlen_offset), q.device, q.dtype)
            cos = cos[seqlen_offset:seqlen_offset + T][None, :, None, :]
            sin = sin[seqlen_offset:seqlen_offset + T][None, :, None, :]

        def rot(x):
            x1, x2 = x.chunk(2, dim=-1)
            return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), dim=-1)

        return rot(q), rot(k)


class ShortConvolution(nn.Conv1d):
    """Depthwise causal 1-D convolution (+ SiLU) of `fla/modules/convolution.py:794-1010` in plain PyTorch: parameters
    `weight [D, 1, W]` (+ `bias`), `forward(x [B, T, D], residual, mask, cache [N, D, W], output_final_state, cu_seqlens)
    -> (y, cache)`.  y_t = act(sum_i w[i] x[t - (W - 1) + i] + b); positions before a sequence start read the cache's last
    W - 1 columns (zeros without a cache); with `cu_seqlens` (B = 1) the convolution does not cross sequence boundaries; a
    single new token per sequence (B T == N) is the decoding step, which rolls the cache in place (:963-1002).  Outside the
    MHLA hot path (SURVEY.md 2.3 marks it OUT), hence no HIP kernel."""

    def __init__(self, hidden_size: int, kernel_size: int, bias: bool = False, activation: Optional[str] = "silu", **kwargs):
        super().__init__(hidden_size, hidden_size, kernel_size, groups=hidden_size, bias=bias, padding=kernel_size - 1)
        self.hidden_size = hidden_size
        if activation is not None and activation not in ("silu", "swish"):
            raise ValueError(f"Activation `{activation}` not supported yet.")
        self.activation = activation

    def _act(self, y):
        return F.silu(y) if self.activation is not None else y

    def forward(self, x, residual=None, mask=None, cache=None, output_final_state=False, cu_seqlens=None, **kwargs):
        B, T, D = x.shape
        W = self.kernel_size[0]
        N = B if cu_seqlens is None else len(cu_seqlens) - 1
        if mask is not None:
            if cu_seqlens is not None:
                raise ValueError("`mask` and `cu_seqlens` cannot be provided at the same time")
            x = x * mask.unsqueeze(-1)
        w = self.weight.squeeze(1)                                           # [D, W]
        if B * T == N:                                                       # decoding step (:927-935)
            xt = x.reshape(N, D)
            if cache is None:
                cache = x.new_zeros(N, D, W)
            cache.copy_(torch.cat([cache[..., 1:], xt.unsqueeze(-1)], dim=-1))
            y = (cache * w).sum(-1)
            if self.bias is not None:
                y = y + self.bias
            y = self._act(y).reshape(x.shape)
            return (y + residual if residual is not None else y), cache

        def one(seq, init):                                                  # seq [b, t, D], init [b, D, W] or None
            hist = init[..., 1:] if init is not None else seq.new_zeros(seq.shape[0], D, W - 1)
            xin = torch.cat([hist.to(seq.dtype), seq.transpose(1, 2)], dim=-1)
            y = F.conv1d(xin, self.weight, self.bias, groups=D).transpose(1, 2)
            full = torch.cat([init.to(seq.dtype) if init is not None else seq.new_zeros(seq.shape[0], D, W), seq.transpose(1, 2)], dim=-1)
            return self._act(y), full[..., -W:]

        if cu_seqlens is None:
            y, final = one(x, cache)
        else:
            ys, finals = [], []
            cu = [int(c) for c in cu_seqlens]
            for i in range(N):
                yi, fi = one(x[:, cu[i]:cu[i + 1]], None if cache is None else cache[i:i + 1])
                ys.append(yi)
                finals.append(fi)
            y, final = torch.cat(ys, dim=1), torch.cat(finals, dim=0)
        if residual is not None:
            y = y + residual
        return y, (final if output_final_state else None)

    @property
    def state_size(self) -> int:
        return self.hidden_size * self.kernel_size[0]


def _elu1(x):
    return F.elu(x) + 1


class MHLA(nn.Module):
    def __init__(self, mode: str = "chunk", hidden_size: int = 1024, expand_k: float = 0.5, expand_v: float = 1.0,
                 num_heads: int = 4, num_kv_heads: Optional[int] = None, feature_map: Optional[str] = None,
                 use_short_conv: bool = False, conv_size: int = 4, conv_bias: bool = False,
                 use_output_gate: bool = True, gate_fn: str = "swish", elementwise_affine: Optional[bool] = True,
                 norm_eps: float = 1e-5, gate_logit_normalizer: int = 16, gate_low_rank_dim: int = 16,
                 clamp_min: Optional[float] = None, fuse_norm: bool = True, layer_idx: int = None, max_chunks: int = 32,
                 summaries: str = "tf32"):
        """`max_chunks` (not in the reference, default = its hard-coded 32): side of the mixing matrix, i.e. the longest
        sequence is 64 * max_chunks tokens -- 128 for the 8192-token configuration of BASELINE.json configs[4], which the
        reference layer itself cannot run (layers/mhla.py:196-200); the operator accepts any [n, n] matrix (naive.py:55).
        `summaries` (not in the reference): "tf32" (default) stores the operator's chunk summaries with 11 significand bits in 2
        bytes (the precision of the reference's TF32 matmuls), "split" with >= 16 bits in 4 bytes (bf16 hi + lo pairs, naive.py:39);
        "bf16" opts into the reduced-precision variant (2-3e-3 of the output's maximum) -- see mhla_amd.mhla_causal."""
        super().__init__()
        self.mode = mode
        self.hidden_size = hidden_size
        self.expand_k = expand_k
        self.expand_v = expand_v
        self.num_heads = num_heads
        self.num_kv_heads = num_kv_heads if num_kv_heads is not None else num_heads
        self.num_kv_groups = self.num_heads // self.num_kv_heads
        self.key_dim = int(hidden_size * expand_k)
        self.value_dim = int(hidden_size * expand_v)
        self.key_dim_per_group = self.key_dim // self.num_kv_groups
        self.value_dim_per_group = self.value_dim // self.num_kv_groups
        self.clamp_min = clamp_min
        self.layer_idx = layer_idx
        if summaries not in ("tf32", "split", "bf16"):
            raise ValueError(f"summaries={summaries!r}: 'tf32', 'split' or 'bf16'")
        self.summaries = summaries
        self.use_output_gate = use_output_gate
        assert mode in ["chunk", "fused_recurrent", "fused_chunk"], f"Not supported mode `{mode}`."
        assert self.key_dim % num_heads == 0, f"key dim must be divisible by num_heads of {num_heads}"
        assert self.value_dim % num_heads == 0, f"value dim must be divisible by num_heads of {num_heads}"
        self.head_k_dim = self.key_dim // num_heads
        self.head_v_dim = self.value_dim // num_heads

        self._fmap_name = feature_map
        if feature_map == "relu":
            self.feature_map_q = self.feature_map_k = nn.ReLU()
        elif feature_map == "identity":
            self.feature_map_q = self.feature_map_k = nn.Identity()
        elif feature_map == "elu":
            self.feature_map_q = self.feature_map_k = _elu1
        else:
            raise NotImplementedError(f"Not supported feature map `{feature_map}`.")
        self.use_short_conv = use_short_conv
        self.conv_size = conv_size
        self.conv_bias = conv_bias

        self.q_proj = nn.Linear(hidden_size, self.key_dim, bias=False)
        self.k_proj = nn.Linear(hidden_size, self.key_dim_per_group, bias=False)
        self.v_proj = nn.Linear(hidden_size, self.value_dim_per_group, bias=False)
        if self.use_output_gate:
            self.g_proj = nn.Linear(hidden_size, self.value_dim, bias=False)
        if use_short_conv:                                                   # layers/mhla.py:175-194
            self.q_conv1d = ShortConvolution(self.key_dim, conv_size, bias=conv_bias, activation="silu")
            self.k_conv1d = ShortConvolution(self.key_dim_per_group, conv_size, bias=conv_bias, activation="silu")
            self.v_conv1d = ShortConvolution(self.value_dim_per_group, conv_size, bias=conv_bias, activation="silu")
        self.max_chunks = int(max_chunks)
        self.mixing_matrix = nn.Parameter(causal_mixing_init(self.max_chunks))   # layers/mhla.py:196-200 (32 there)
        self.o_proj = nn.Linear(self.value_dim, hidden_size, bias=False)
        self.fuse_norm_and_gate = gate_fn == "swish" and fuse_norm and use_output_gate
        if self.fuse_norm_and_gate:
            self.g_norm_swish_gate = FusedRMSNormGated(self.head_v_dim, elementwise_affine, norm_eps)
        else:
            self.g_norm = FusedRMSNormGated(self.head_v_dim, elementwise_affine, norm_eps)
            self.gate_fn = {"swish": F.silu, "silu": F.silu, "sigmoid": torch.sigmoid, "relu": F.relu,
                            "gelu": F.gelu}[gate_fn]
        self.gate_logit_normalizer = gate_logit_normalizer
        assert self.head_k_dim <= 256, "head_k_dim must be less than or equal to 256"
        self.rotary = RotaryEmbedding(dim=self.head_k_dim)

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                past_key_values=None, use_cache: Optional[bool] = False, output_attentions: Optional[bool] = False,
                **kwargs: Dict):
        # clamp + tril of the mixing weights at the start of every forward, on .data (layers/mhla.py:237)
        self.mixing_matrix.data = torch.clamp(self.mixing_matrix.data, 1e-5, 1).tril()
        if attention_mask is not None:
            assert len(attention_mask.shape) == 2, (
                "Expected attention_mask as a 0-1 matrix with shape [batch_size, seq_len] for padding purposes "
                "(0 indicating padding). Arbitrary attention masks of shape [batch_size, seq_len, seq_len] are not allowed.")
        batch_size, q_len, _ = hidden_states.shape
        last_state = None
        if past_key_values is not None and self.layer_idx is not None and len(past_key_values) > self.layer_idx:
            last_state = past_key_values[self.layer_idx]                      # :249-251
        indices = None
        cu_seqlens = kwargs.get("cu_seqlens", None)
        if attention_mask is not None:                                       # layers/mhla.py:253-256 (get_unpad_data)
            m = attention_mask[:, -q_len:]
            indices = torch.nonzero(m.flatten(), as_tuple=False).flatten()
            cu_seqlens = F.pad(m.sum(-1, dtype=torch.int32).cumsum(0, dtype=torch.int32), (1, 0))
            hidden_states = hidden_states.reshape(batch_size * q_len, -1).index_select(0, indices).unsqueeze(0)
        B, T, _ = hidden_states.shape
        conv_states = None
        if self.use_short_conv:                                              # :258-279
            cq = ck = cv = None
            if last_state is not None and last_state.get("conv_state") is not None:
                cq, ck, cv = last_state["conv_state"]
            q, cq = self.q_conv1d(x=self.q_proj(hidden_states), cache=cq, output_final_state=use_cache, cu_seqlens=cu_seqlens)
            k, ck = self.k_conv1d(x=self.k_proj(hidden_states), cache=ck, output_final_state=use_cache, cu_seqlens=cu_seqlens)
            v, cv = self.v_conv1d(x=self.v_proj(hidden_states), cache=cv, output_final_state=use_cache, cu_seqlens=cu_seqlens)
            conv_states = (cq, ck, cv)
            q = q.reshape(B, T, self.num_heads, self.head_k_dim)
        else:
            q = self.q_proj(hidden_states).reshape(B, T, self.num_heads, self.head_k_dim)
            k = self.k_proj(hidden_states)
            v = self.v_proj(hidden_states)
        if self.num_kv_groups > 1:                                           # :290-292 (repeat '(h g) d')
            k = k.reshape(B, T, self.num_kv_heads, 1, self.head_k_dim).expand(-1, -1, -1, self.num_kv_groups, -1)
            v = v.reshape(B, T, self.num_kv_heads, 1, self.head_v_dim).expand(-1, -1, -1, self.num_kv_groups, -1)
        k = k.reshape(B, T, self.num_heads, self.head_k_dim)
        v = v.reshape(B, T, self.num_heads, self.head_v_dim)
        seqlen_offset = 0
        if past_key_values is not None and hasattr(past_key_values, "get_seq_length"):
            seqlen_offset = past_key_values.get_seq_length(self.layer_idx)    # :301-303
        # rotary position of every token row: packed sequences restart at every sequence start (rotary.py:68-72 with cu_seqlens);
        # with a padding mask AND a cache offset every sequence continues from its own length (prepare_lens_from_mask, :305-309)
        positions = None
        if cu_seqlens is not None:
            tpos = torch.arange(T, device=q.device)
            cu = cu_seqlens.to(q.device).long()
            seq = torch.searchsorted(cu, tpos, right=True) - 1
            offs = seqlen_offset
            if attention_mask is not None and seqlen_offset > 0:
                offs = (attention_mask.sum(-1).to(q.device).long() - q_len)[seq]
            positions = tpos - cu[seq] + offs
        table_len = T + seqlen_offset if positions is None else (
            (int(attention_mask.shape[1]) if attention_mask is not None and seqlen_offset > 0 else T + seqlen_offset))
        if self.head_k_dim % 8 == 0:
            # feature map (:297-299) + rotary (:311) in one HIP kernel per tensor and direction
            cos, sin = self.rotary._tables(table_len, q.device, q.dtype)
            t_off = seqlen_offset
            if positions is not None:   # per-token rows of the tables, gathered once
                cos, sin, t_off = cos.index_select(0, positions), sin.index_select(0, positions), 0
            q = featmap_rotary(q, cos, sin, self._fmap_name, t_off)
            k = featmap_rotary(k, cos, sin, self._fmap_name, t_off)
        else:
            if not getattr(self, "_warned_eager_rotary", False):
                warnings.warn(f"MHLA: head_k_dim={self.head_k_dim} is not a multiple of 8: feature map and rotary run as eager "
                              "PyTorch ops (about ten elementwise passes per tensor) instead of the fused HIP kernel", stacklevel=2)
                self._warned_eager_rotary = True
            q, k = self.feature_map_q(q), self.feature_map_k(k)              # :297-299
            q, k = self.rotary(q, k, seqlen_offset=seqlen_offset, max_seqlen=table_len, positions=positions)   # :311
        recurrent_state = last_state["recurrent_state"] if last_state is not None else None
        fused_epilogue = self.use_output_gate and self.fuse_norm_and_gate and q_len > 64
        if fused_epilogue:
            # operator + per-head RMSNorm x swish gate (:330-337 + :351-355) as one node: the epilogue runs in the operator's
            # output kernel where the shape allows, otherwise as the separate HIP kernel (mhla_causal_normgate decides)
            g = self.g_proj(hidden_states).reshape(B, T, self.num_heads, self.head_v_dim)
            gn = self.g_norm_swish_gate
            o = mhla_causal_normgate(q, k, v, self.mixing_matrix, g, gn.weight, gn.eps, summaries=self.summaries).reshape(B, T, self.value_dim)
            recurrent_state = None
        elif q_len <= 64:                                                    # :247, :318-327: the token-recurrent form
            if T > 64 and not getattr(self, "_warned_recurrent_packed", False):
                warnings.warn(f"MHLA: a padded batch of {batch_size} x {q_len} tokens unpads to one packed sequence of {T} > 64 tokens; "
                              "the recurrent branch then runs the multi-chunk chunk operator (the reference's recurrent form reads "
                              "shifted states beyond the first chunk -- not replicated, see naive_recurrent_mhla)", stacklevel=2)
                self._warned_recurrent_packed = True
            o, recurrent_state = naive_recurrent_mhla(q, k, v, self.mixing_matrix, initial_state=recurrent_state,
                                                      output_final_state=bool(use_cache))
        else:                                                                # :330-337
            o = mhla_causal(q, k, v, self.mixing_matrix, summaries=self.summaries)
            recurrent_state = None
        if past_key_values is not None and hasattr(past_key_values, "update"):   # :339-345
            past_key_values.update(recurrent_state=recurrent_state, conv_state=conv_states if self.use_short_conv else None,
                                   layer_idx=self.layer_idx, offset=q_len)
        if fused_epilogue:
            pass
        elif self.use_output_gate:
            g = self.g_proj(hidden_states)
            if self.fuse_norm_and_gate:
                o = self.g_norm_swish_gate(o, g.reshape(B, T, self.num_heads, self.head_v_dim))   # :351-355
                o = o.reshape(B, T, self.value_dim)
            else:
                o = self.g_norm(o, None).reshape(B, T, self.value_dim) * self.gate_fn(g)
        else:
            o = self.g_norm(o, None).reshape(B, T, self.value_dim)
        o = self.o_proj(o)
        if indices is not None:                                              # pad_input, :362-363
            full = o.new_zeros(batch_size * q_len, o.shape[-1])
            full.index_copy_(0, indices, o.squeeze(0))
            o = full.reshape(batch_size, q_len, -1)
        return o, None, past_key_values
