#!/usr/bin/env python
"""CPU model of the 2-byte chunk-summary format on the causal operator (see tools/sim_h16.py): S, P, dP, dS rounded to an fp16 payload
x a power-of-two multiplier per (chunk, 64 x 64 tile, 16-row strip) -- measured maximum for S and dP (k_csf_state2), bound
sum_j |m_ij| m_j for P and dS (the mixing kernels) -- everything else fp64; against the all-fp64 result."""
import sys
import torch

DT = torch.float64


def p2f(x): return torch.exp2(torch.floor(torch.log2(x.clamp_min(1e-300))))
def p2c(x): return torch.exp2(torch.ceil(torch.log2(x.clamp_min(1e-300))))


def strips(x):   # [bh, n, K, V] -> view [bh, n, K/16, 16, V/64, 64]
    bh, n, K, V = x.shape
    return x.reshape(bh, n, K // 16, 16, V // 64, 64)


def q16(x, m, fmt):   # m: [bh, n, K/16, V/64]
    if fmt == "bf16":
        return x.to(torch.bfloat16).to(DT)
    xs = strips(x)
    mm = m[:, :, :, None, :, None]
    return ((xs / mm).to(torch.float16).to(DT) * mm).reshape(x.shape)


def measured(x):
    return p2f(strips(x).abs().amax(dim=(3, 5))) * 2.0 ** -14


class Quant(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mf, mb_fn, fmt):
        ctx.mb_fn, ctx.fmt = mb_fn, fmt
        return q16(x, mf, fmt)

    @staticmethod
    def backward(ctx, g):
        return q16(g, ctx.mb_fn(g), ctx.fmt), None, None, None


def run(B, T, H, K, V, fmt, mixkind="tril", seed=0, signed=True):
    g = torch.Generator().manual_seed(seed)
    n = T // 64
    bh = B * H
    q = torch.randn(bh, n, 64, K, generator=g)
    k = torch.randn(bh, n, 64, K, generator=g)
    if not signed:
        q, k = torch.relu(q), torch.relu(k)
    q, k = q.bfloat16().to(DT), k.bfloat16().to(DT)
    v = torch.randn(bh, n, 64, V, generator=g).bfloat16().to(DT)
    do = torch.randn(bh, n, 64, V, generator=g).bfloat16().to(DT)
    if mixkind == "tril":
        mix = torch.tril(torch.ones(n, n, dtype=DT)) / torch.arange(1, n + 1, dtype=DT)[:, None]
    else:
        mix = torch.tril(torch.rand(n, n, generator=g).to(DT)).clamp_min(1e-5).tril()
    scale = K ** -0.5
    res = {}
    for mode in ("exact", fmt):
        qq, kk, vv, mm = (t.clone().requires_grad_(True) for t in (q, k, v, mix))
        S = kk.transpose(-2, -1) @ vv                                   # [bh, n, K, V]
        lower = torch.tril(mm, -1)
        holder = {}
        if mode != "exact":
            mS = measured(S.detach())
            Sq = Quant.apply(S, mS, lambda gr: p2c(torch.einsum("ij,bikv->bjkv", torch.tril(mm.detach(), -1).abs(), holder["mdP"])), mode)
        else:
            Sq = S
        P = torch.einsum("ij,bjkv->bikv", lower, Sq)
        if mode != "exact":
            mP = p2c(torch.einsum("ij,bjkv->bikv", lower.detach().abs(), mS))

            def mb(gr):
                holder["mdP"] = measured(gr)
                return holder["mdP"]
            Pq = Quant.apply(P, mP, mb, mode)
        else:
            Pq = P
        A = torch.tril(qq @ kk.transpose(-2, -1))
        out = scale * (qq @ Pq + torch.diagonal(mm)[None, :, None, None] * (A @ vv))
        out.backward(do)
        res[mode] = {"out": out.detach(), "dq": qq.grad, "dk": kk.grad, "dv": vv.grad, "dmix": torch.tril(mm.grad)}
    return {kx: ((res[fmt][kx] - res["exact"][kx]).abs().max() / res["exact"][kx].abs().max()).item() for kx in res["exact"]}


if __name__ == "__main__":
    for shp in [(1, 2048, 2, 128, 256), (1, 8192, 1, 128, 256), (1, 4096, 1, 256, 512), (2, 512, 2, 64, 64), (2, 130 // 64 * 64 + 128, 2, 64, 128)]:
        for mk in ("tril", "rand"):
            for signed in (True, False):
                for fmt in ("f16", "bf16"):
                    e = run(*shp, fmt, mk, 0, signed)
                    print(shp, mk, "signed" if signed else "relu", fmt, " ".join(f"{k_}={v_:.1e}" for k_, v_ in e.items()), flush=True)
