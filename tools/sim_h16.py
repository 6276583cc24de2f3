#!/usr/bin/env python
"""CPU model of the 2-byte block-summary format ("h16": fp16 payload x one power-of-two multiplier per block row) on the
block-mixing operator: KV, G, dG, dKV are rounded to the format -- KV / dG with the multiplier taken from the row's measured
maximum (the state kernels), G / dKV with the multiplier taken from the BOUND sum_r |W(o, r)| m_r (the mixing kernels know no
more) -- everything else in fp64; compared with the all-fp64 result.  max|err| / max|want| per result."""
import sys
import torch

torch.manual_seed(0)
DT = torch.float64


def pow2_floor(x):
    return torch.exp2(torch.floor(torch.log2(x.clamp_min(1e-300))))


def pow2_ceil(x):
    return torch.exp2(torch.ceil(torch.log2(x.clamp_min(1e-300))))


def q16(x, m, fmt):
    if fmt == "f16":
        return (x / m).to(torch.float16).to(DT) * m
    if fmt == "bf16":
        return x.to(torch.bfloat16).to(DT)
    return x


class Quant(torch.autograd.Function):
    """forward: round x [bh, M, D, D] to the format with multipliers mf [bh, M]; backward: round the gradient with multipliers from mb(grad)."""
    @staticmethod
    def forward(ctx, x, mf, mb_fn, fmt):
        ctx.mb_fn, ctx.fmt = mb_fn, fmt
        return q16(x, mf[..., None, None], fmt)

    @staticmethod
    def backward(ctx, g):
        return q16(g, ctx.mb_fn(g)[..., None, None], ctx.fmt), None, None, None


def measured(x):   # decode multiplier of a row from its maximum: payload maximum in [2^14, 2^15)
    return pow2_floor(x.abs().amax(dim=(-2, -1))) * 2.0 ** -14


def run(B, H, M, S, D, fmt, Wkind="linear", seed=0):
    sys.path.insert(0, ".")
    from oracle import mhla_oracle as orc
    g = torch.Generator().manual_seed(seed)
    bh = B * H
    q = (torch.relu(torch.randn(bh, M, S, D, generator=g)) + 1e-6).bfloat16().to(DT)
    k = (torch.relu(torch.randn(bh, M, S, D, generator=g)) + 1e-6).bfloat16().to(DT)
    v = torch.randn(bh, M, S, D, generator=g).bfloat16().to(DT)
    do = torch.randn(bh, M, S, D, generator=g).bfloat16().to(DT)
    side = int(round(M ** 0.5))
    if Wkind == "linear":
        W = orc.block_distance_weights((side, side) if side * side == M else (M,), "linear").to(DT)
    else:
        W = torch.rand(M, M, generator=g).to(DT)
    res = {}
    for mode in ("exact", fmt):
        qq, kk, vv, WW = (t.clone().requires_grad_(True) for t in (q, k, v, W))
        kv = kk.transpose(-2, -1) @ vv
        if mode != "exact":
            m_kv = measured(kv.detach())
            holder = {}

            def mb_kv(gr):   # dKV = W^T dG: bound from the dG multipliers
                return pow2_ceil(torch.einsum("ij,bi->bj", WW.detach().abs(), holder["m_dg"]))

            kvq = Quant.apply(kv, m_kv, mb_kv, mode)
        else:
            kvq = kv
        gm = torch.einsum("ij,bjde->bide", WW, kvq)
        if mode != "exact":
            m_g = pow2_ceil(torch.einsum("ij,bj->bi", WW.detach().abs(), m_kv))

            def mb_g(gr):    # dG: measured
                holder["m_dg"] = measured(gr)
                return holder["m_dg"]

            gq = Quant.apply(gm, m_g, mb_g, mode)
        else:
            gq = gm
        num = qq @ gq
        ksum = kk.sum(dim=-2)
        z = torch.einsum("bmsd,bmd->bms", qq, ksum)
        n = torch.einsum("ij,bjs->bis", WW, z) + 1e-6
        out = num / n[..., None]
        out.backward(do)
        res[mode] = {"out": out.detach(), "dq": qq.grad, "dk": kk.grad, "dv": vv.grad, "dW": WW.grad}
    err = {kx: ((res[fmt][kx] - res["exact"][kx]).abs().max() / res["exact"][kx].abs().max()).item() for kx in res["exact"]}
    return err


if __name__ == "__main__":
    shapes = [(1, 2, 64, 64, 64), (1, 2, 16, 256, 64), (1, 1, 256, 16, 64), (2, 2, 16, 16, 72), (1, 1, 36, 64, 128), (1, 2, 4, 16, 64), (1, 1, 2, 1, 64)]
    for shp in shapes:
        for wk in ("linear", "uniform"):
            for fmt in ("f16", "bf16"):
                e = run(*shp, fmt, wk)
                print(shp, wk, fmt, " ".join(f"{k_}={v_:.1e}" for k_, v_ in e.items()), flush=True)
