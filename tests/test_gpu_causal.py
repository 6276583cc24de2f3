"""GPU parity: causal chunk-mixing MHLA operator + per-head RMSNorm x gate vs oracle and golden fixtures."""
import pytest
import torch

from conftest import load_golden
from gpu_util import poison, DEV, GTOL_BF16SUM, TOL_BF16SUM, CAUSAL_TOL, CAUSAL_DMIX_TOL, check
from oracle import mhla_oracle as orc

pytestmark = pytest.mark.gpu


def causal_inputs(B, T, H, K, V, L, dtype, seed=1234, random_mix=True):
    g = torch.Generator().manual_seed(seed)
    q = (torch.relu(torch.randn(B, T, H, K, generator=g)) * torch.sign(torch.randn(B, T, H, K, generator=g))).to(dtype)
    k = (torch.relu(torch.randn(B, T, H, K, generator=g)) * torch.sign(torch.randn(B, T, H, K, generator=g))).to(dtype)
    v = torch.randn(B, T, H, V, generator=g).to(dtype)
    do = torch.randn(B, T, H, V, generator=g).to(dtype)
    mix = torch.tril(torch.rand(L, L, generator=g).clamp(1e-5, 1)) if random_mix else orc.causal_mixing_init(L)
    return q, k, v, mix, do


def causal_tols(dtype, summaries="tf32"):
    """(out / dq / dk / dv tolerance, dmix tolerance): one final rounding + 1e-3 at the reference's arithmetic; the reduced
    precision variant carries K = 1 (outputs) / 2 (gradients) bf16 intermediates (gpu_util)."""
    if summaries == "bf16":
        return TOL_BF16SUM[dtype], GTOL_BF16SUM[dtype], GTOL_BF16SUM[dtype]
    return CAUSAL_TOL[dtype], CAUSAL_TOL[dtype], CAUSAL_DMIX_TOL[dtype]


def run_causal(B, T, H, K, V, L, dtype, seed=1234, summaries="tf32"):
    import mhla_amd
    q, k, v, mix, do = causal_inputs(B, T, H, K, V, L, dtype, seed)
    want = orc.causal_fwd(q.float(), k.float(), v.float(), mix)
    wg = orc.causal_bwd(q.float(), k.float(), v.float(), mix, do.float())
    dq, dk, dv, dm = (t.to(DEV).requires_grad_(True) for t in (q, k, v, mix.view(L, L, 1, 1, 1, 1)))
    poison()
    if summaries == "tf32":   # the reference's own call form (the library's default arithmetic)
        out = mhla_amd.naive_chunk_simple_mhla_fixed(q=dq, k=dk, v=dv, mixing_matrix=dm)
    else:
        out = mhla_amd.mhla_causal(dq, dk, dv, dm, summaries=summaries)
    assert out.dtype == dtype and out.shape == (B, T, H, V)
    dod = do.to(DEV)
    poison()
    out.backward(dod)
    otol, gtol, mtol = causal_tols(dtype, summaries)
    check("out", out, want, otol)
    check("dq", dq.grad, wg["dq"], gtol)
    check("dk", dk.grad, wg["dk"], gtol)
    check("dv", dv.grad, wg["dv"], gtol)
    check("dmix", dm.grad.reshape(L, L), wg["dmix"], mtol)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_golden_causal(tag):
    import mhla_amd
    g = load_golden("causal_" + tag)
    bf16 = tag == "d"
    cast = (lambda t: t.bfloat16()) if bf16 else (lambda t: t)
    q, k, v = (cast(g[n]).to(DEV).requires_grad_(True) for n in ("q", "k", "v"))
    mix = g["mix"].to(DEV).requires_grad_(True)
    out = mhla_amd.mhla_causal(q, k, v, mix)
    out.backward(cast(g["dout"]).to(DEV))
    # Fixture d is the reference's own bf16 call (naive.py:39 upcasts, :82 rounds once): its outputs and gradients are THEMSELVES
    # bf16-rounded, so a result that differs from the reference's fp32 value by 1e-5 still lands on the neighbouring bf16 number
    # now and then -- a one-ulp flip, |err| <= 2 u |x| on that element, nothing in between.  Hence: max error within one ulp
    # (2u + 1e-3), and the rms-relative error -- which counts how OFTEN that happens -- under north_star's 1e-3 (observed 1.5e-4;
    # single-bf16 summaries flip 40 % of the elements: 3e-3).
    tol = 2 * 2.0 ** -8 + 1e-3 if bf16 else 1e-4
    check("out", out, g["out"], tol)
    for n, t in (("dq", q), ("dk", k), ("dv", v)):
        check(n, t.grad, g[n], tol if bf16 else 2e-4)
    check("dmix", mix.grad, g["dmix"], CAUSAL_DMIX_TOL[torch.bfloat16] if bf16 else 2e-4)
    if bf16:
        from conftest import rms_ratio
        for n, got in (("out", out), ("dq", q.grad), ("dk", k.grad), ("dv", v.grad)):
            r = rms_ratio(got.float().cpu(), g[n].float())
            assert r < 1e-3, f"{n}: rms-relative error {r:.2e} vs the reference's own bf16 result"
    if "out_recurrent" in g:   # T <= 64: the reference's token-recurrent form (naive.py:88-142), as the fla layer calls it
        o_rec, S = mhla_amd.naive_recurrent_mhla(q.detach(), k.detach(), v.detach(), mix.detach())
        check("naive_recurrent_mhla", o_rec, g["out_recurrent"], tol)
        assert S.shape == (q.shape[0], q.shape[2], q.shape[3], v.shape[3]) and float(S.abs().max()) == 0.0
        assert mhla_amd.naive_recurrent_mhla(q.detach(), k.detach(), v.detach(), mix.detach(), output_final_state=False)[1] is None


@pytest.mark.parametrize("T,K,V", [(256, 64, 64), (200, 32, 16), (50, 16, 24), (64, 128, 256), (1000, 128, 128),
                                   (129, 256, 512), (4096, 16, 16)])
def test_causal_shapes_fp32(T, K, V):
    run_causal(1 if K * V > 16384 else 2, T, 2, K, V, max(4, (T + 63) // 64), torch.float32)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_causal_lowp(dtype):
    run_causal(2, 512, 4, 128, 256, 8, dtype)


BF16_SHAPES = [(256, 64, 64), (200, 64, 128), (50, 128, 64), (1000, 128, 256), (129, 256, 512),
               (320, 64, 48), (320, 48, 64), (200, 192, 192), (130, 192, 320), (200, 64, 384),
               # chunk walks of the output / summaries kernels (four chunks per workgroup): a full group followed by a partial
               # one, ragged and one-token last chunks
               (330, 128, 256), (449, 128, 256), (321, 64, 64), (8192, 64, 64), (2100, 256, 256),
               # 129..256 chunks: the sixteen-wave mixing kernels (dS and dmix as two launches)
               (8256, 64, 64), (16384, 64, 64), (10000, 128, 128),
               # the token-gradient kernel's chunk walk (k_csf_bwd_tok4: with B H = 4 here, two chunks per workgroup from 128 chunks on and
               # four at 256 -- the shapes above cover K = 64 / 128 with an odd chunk count and a ragged tail; the full-size C5 test walks
               # eight): K = 192, an odd count and a 8-token last chunk
               (8200, 192, 192)]


@pytest.mark.parametrize("T,K,V", BF16_SHAPES)
def test_causal_shapes_bf16(T, K, V):
    """K and V multiples of 64 (K <= 256, at most 256 chunks) run the 16-bit-MFMA pipeline (ragged last chunk included); the
    others the generic kernels.  The eight-wave kernels are templated on K / 64 (1..4) and on the V slices per workgroup
    (192 x 192 -> <3>, <3>; V = 320 -> five workgroups of one slice; V = 384 -> two of three).  Tolerance: the reference's
    arithmetic -- one final rounding + 1e-3."""
    run_causal(2, T, 2, K, V, max(4, (T + 63) // 64), torch.bfloat16, seed=T + K)


@pytest.mark.parametrize("T,K,V", [(256, 64, 64), (1000, 128, 256), (129, 256, 512), (200, 192, 192), (200, 64, 384), (330, 128, 256),
                                   (9000, 64, 128)])
def test_causal_shapes_bf16_reduced_precision_variant(T, K, V):
    """summaries="bf16" (MHLA_CAUSAL_BF16_SUMMARIES): the opt-in variant with single-bf16 chunk summaries and score tiles, held
    to the K-intermediate bounds of gpu_util (2u outputs, 3u gradients)."""
    run_causal(2, T, 2, K, V, max(4, (T + 63) // 64), torch.bfloat16, seed=T + K, summaries="bf16")


@pytest.mark.parametrize("T,K,V", [(256, 64, 64), (1000, 128, 256), (129, 256, 512), (200, 192, 192), (200, 64, 384), (330, 128, 256), (2100, 256, 256),
                                   (9000, 64, 128)])
def test_causal_shapes_bf16_hi_lo_summaries(T, K, V):
    """summaries="split" (MHLA_CAUSAL_FP32_GRADE_SUMMARIES): chunk summaries as bf16 hi + lo pairs (>= 16 significand bits, round 5's
    default) instead of the 2-byte h16 form, at the same tolerance."""
    run_causal(2, T, 2, K, V, max(4, (T + 63) // 64), torch.bfloat16, seed=T + K, summaries="split")


@pytest.mark.parametrize("vscale,doscale", [(1e-18, 1e12), (3e14, 1e-25)])
def test_causal_h16_summaries_at_extreme_scales(vscale, doscale):
    """The chunk summaries' strip multipliers absorb the scale of v and dO (1e-18 .. 3e14); a zero chunk and one 2^40 above the rest included."""
    import mhla_amd
    B, T, H, K, V, L = 1, 512, 2, 64, 128, 8
    q, k, v, mix, do = causal_inputs(B, T, H, K, V, L, torch.float32, seed=21)
    v = v * vscale
    v[:, 64:128] = 0.0
    v[:, 192:256] *= 2.0 ** 40
    do = do * doscale
    q, k, v, do = (t.to(torch.bfloat16) for t in (q, k, v, do))
    assert mhla_amd.describe_causal_dispatch(T, K, V, torch.bfloat16)["summaries"].startswith("h16")
    want = orc.causal_fwd(q.float(), k.float(), v.float(), mix)
    wg = orc.causal_bwd(q.float(), k.float(), v.float(), mix, do.float())
    dq, dk, dv, dm = (t.to(DEV).requires_grad_(True) for t in (q, k, v, mix))
    out = mhla_amd.mhla_causal(dq, dk, dv, dm)
    out.backward(do.to(DEV))
    otol, gtol, mtol = causal_tols(torch.bfloat16)
    check("out", out, want, otol)
    check("dq", dq.grad, wg["dq"], gtol)
    check("dk", dk.grad, wg["dk"], gtol)
    check("dv", dv.grad, wg["dv"], gtol)
    check("dmix", dm.grad, wg["dmix"], mtol)


@pytest.mark.parametrize("T,K,V", [(16400, 64, 64), (300, 320, 64)])
def test_causal_bf16_beyond_the_pipeline(T, K, V):
    """bf16 tensors outside the 16-bit pipeline's range (more than 256 chunks; K > 256): the generic fp32-MFMA kernels."""
    run_causal(1, T, 2, K, V, max(4, (T + 63) // 64), torch.bfloat16, seed=T + K)


def test_causal_bf16_pipeline_vs_generic():
    """The 16-bit pipeline (hi + lo chunk summaries) against the generic fp32-compute kernels on the same bf16 inputs: both
    carry only the final rounding, so they agree to twice that + 1e-3."""
    import mhla_amd
    q, k, v, mix, do = causal_inputs(2, 700, 2, 128, 128, 16, torch.bfloat16, seed=77)
    res = {}
    for tag in ("fast", "generic"):
        dq, dk, dv, dm = (t.to(DEV).requires_grad_(True) for t in (q, k, v, mix.view(16, 16, 1, 1, 1, 1)))
        out = mhla_amd.mhla_causal(dq, dk, dv, dm, force_generic=(tag == "generic"))
        out.backward(do.to(DEV))
        res[tag] = (out, dq.grad, dk.grad, dv.grad, dm.grad)
    for name, a, b in zip(("out", "dq", "dk", "dv"), res["fast"], res["generic"]):
        check(name, a, b.float().cpu(), 2 * 2.0 ** -8 + 1e-3)
    check("dmix", res["fast"][4], res["generic"][4].float().cpu(), 1e-3)


@pytest.mark.parametrize("summaries", ["tf32", "split"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_causal_kept_summaries_match_recompute(dtype, summaries, monkeypatch):
    """The backward reuses the forward's chunk summaries when the workspace is kept; same gradients as recomputing them."""
    import mhla_amd
    from mhla_amd import ops
    q, k, v, mix, do = causal_inputs(2, 300, 2, 64, 128, 8, dtype, seed=5)
    res = []
    for limit in (1 << 30, 0):
        monkeypatch.setattr(ops, "CAUSAL_KEEP_STATE_LIMIT_BYTES", limit)
        t = [x.to(DEV).requires_grad_(True) for x in (q, k, v, mix)]
        mhla_amd.mhla_causal(*t, summaries=summaries).backward(do.to(DEV))
        res.append([x.grad for x in t])
    for name, a, b in zip(("dq", "dk", "dv", "dmix"), *res):
        assert torch.equal(a, b), name


def test_causal_more_than_2_31_elements():
    """Maximum sizes for the causal operator: V tensors of 2.1e9 elements (B=272, T=8192, H=4, V=256), chunk summaries of
    8.6 GB each; sampled (b, h) slices vs the oracle, forward and backward."""
    import mhla_amd
    B, T, H, K, V = 272, 8192, 4, 128, 256
    assert B * T * H * V > 2 ** 31
    gen = torch.Generator(device=DEV).manual_seed(3)
    mk = lambda d: torch.randn(B, T, H, d, device=DEV, dtype=torch.bfloat16, generator=gen)
    q, k, v, do = mk(K), mk(K), mk(V), mk(V)
    n = T // 64
    mix = torch.tril(torch.rand(n, n, generator=torch.Generator().manual_seed(1)).clamp(1e-5, 1))
    md = mix.to(DEV).requires_grad_(True)
    for t in (q, k, v):
        t.requires_grad_(True)
    out = mhla_amd.mhla_causal(q, k, v, md)
    out.backward(do)
    for (b, h) in [(0, 0), (B - 1, H - 1)]:
        sl = lambda t: t.detach()[b:b + 1, :, h:h + 1].float().cpu()
        want = orc.causal_fwd(sl(q), sl(k), sl(v), mix)
        wg = orc.causal_bwd(sl(q), sl(k), sl(v), mix, sl(do))
        s16 = lambda t: t.detach()[b:b + 1, :, h:h + 1].cpu()   # (bf16: check() charges the final rounding per element)
        check("out", s16(out), want, CAUSAL_TOL[torch.bfloat16])
        check("dq", s16(q.grad), wg["dq"], CAUSAL_TOL[torch.bfloat16])
        check("dk", s16(k.grad), wg["dk"], CAUSAL_TOL[torch.bfloat16])
        check("dv", s16(v.grad), wg["dv"], CAUSAL_TOL[torch.bfloat16])
    # dmix of the whole batch = sum over batch chunks (size-independent), first samples anchored on the oracle
    full = md.grad.detach().double().cpu()
    acc = torch.zeros_like(full)
    for b0 in range(0, B, 34):
        qs, ks, vs = (t.detach()[b0:b0 + 34].requires_grad_(True) for t in (q, k, v))
        mc = mix.to(DEV).requires_grad_(True)
        mhla_amd.mhla_causal(qs, ks, vs, mc).backward(do[b0:b0 + 34])
        acc += mc.grad.double().cpu()
    check("dmix additivity over batch chunks", full.float(), acc.float(), 2e-4)
    f = lambda t: t.detach()[:2].float().cpu()
    wg2 = orc.causal_bwd(f(q), f(k), f(v), mix, f(do))
    mc = mix.to(DEV).requires_grad_(True)
    mhla_amd.mhla_causal(q.detach()[:2], k.detach()[:2], v.detach()[:2], mc).backward(do[:2])
    check("dmix (first 2 samples vs oracle)", mc.grad, wg2["dmix"], CAUSAL_DMIX_TOL[torch.bfloat16])
    del q, k, v, do, out
    torch.cuda.empty_cache()


def test_causal_bf16_strided_views():
    """q/k/v as slices of one fused projection buffer (row stride 3 * H * K): the views stay 16-byte aligned."""
    import mhla_amd
    B, T, H, K, L = 2, 300, 3, 64, 8
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B, T, 3, H, K, generator=g).bfloat16()
    mix = torch.tril(torch.rand(L, L, generator=g).clamp(1e-5, 1))
    do = torch.randn(B, T, H, K, generator=g).bfloat16()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    want = orc.causal_fwd(q.float(), k.float(), v.float(), mix)
    wg = orc.causal_bwd(q.float(), k.float(), v.float(), mix, do.float())
    dqkv = qkv.to(DEV).requires_grad_(True)
    dm = mix.view(L, L, 1, 1, 1, 1).to(DEV).requires_grad_(True)
    out = mhla_amd.naive_chunk_simple_mhla_fixed(q=dqkv[:, :, 0], k=dqkv[:, :, 1], v=dqkv[:, :, 2], mixing_matrix=dm)
    out.backward(do.to(DEV))
    check("out", out, want, CAUSAL_TOL[torch.bfloat16])
    for i, n in enumerate(("dq", "dk", "dv")):
        check(n, dqkv.grad[:, :, i], wg[n], CAUSAL_TOL[torch.bfloat16])
    check("dmix", dm.grad.reshape(L, L), wg["dmix"], CAUSAL_DMIX_TOL[torch.bfloat16])


def test_causal_is_causal_and_recurrent_first_chunk():
    """Future tokens never influence earlier outputs; with T <= 64 the op is the single-chunk
    (token-recurrent) case."""
    import mhla_amd
    q, k, v, mix, _ = causal_inputs(1, 300, 2, 32, 32, 8, torch.float32)
    dq, dk, dv, dm = (t.to(DEV) for t in (q, k, v, mix))
    o1 = mhla_amd.mhla_causal(dq, dk, dv, dm)
    k2, v2 = dk.clone(), dv.clone()
    k2[:, 200:] = 7.0
    v2[:, 200:] = -3.0
    o2 = mhla_amd.mhla_causal(dq, k2, v2, dm)
    assert torch.equal(o1[:, :200], o2[:, :200])
    with pytest.raises(IndexError):
        mhla_amd.mhla_causal(dq, dk, dv, dm[:4, :4])     # 300 tokens need 5 chunks


@pytest.mark.parametrize("fmap", [None, "relu", "elu"])
@pytest.mark.parametrize("dtype,K,off", [(torch.float32, 64, 0), (torch.bfloat16, 128, 0), (torch.float32, 24, 5)])
def test_featmap_rotary(fmap, dtype, K, off):
    """Fused feature map + NeoX rotary (fla layers/mhla.py:297-299, :311) vs the oracle's rotary twin, with gradients."""
    import torch.nn.functional as F
    import mhla_amd
    g = torch.Generator().manual_seed(K + off)
    B, T, H = 2, 77, 3
    x = torch.randn(B, T, H, K, generator=g).to(dtype)
    dy = torch.randn(B, T, H, K, generator=g).to(dtype)
    inv = 1.0 / (10000.0 ** (torch.arange(0, K, 2, dtype=torch.float32) / K))
    fr = torch.outer(torch.arange(T + off, dtype=torch.float32), inv)
    cos, sin = torch.cos(fr).to(dtype), torch.sin(fr).to(dtype)
    f = {None: lambda t: t, "relu": torch.relu, "elu": lambda t: F.elu(t) + 1}[fmap]
    xr = x.float().clone().requires_grad_(True)
    # rotary_embedding_kernel semantics (rotary.py:97-135): fp32 math on the tables as cached in the activation dtype
    c, s_ = cos.float()[off:off + T][None, :, None, :], sin.float()[off:off + T][None, :, None, :]
    a, b = f(xr).chunk(2, dim=-1)
    want = torch.cat((a * c - b * s_, b * c + a * s_), dim=-1)
    if dtype == torch.float32:
        check("oracle twin", want.detach(), orc.neox_rotary(f(x.float()), offset=off), 1e-6)
    want.backward(dy.float())
    xd = x.detach().to(DEV).requires_grad_(True)
    got = mhla_amd.featmap_rotary(xd, cos.to(DEV), sin.to(DEV), fmap, off)
    got.backward(dy.to(DEV))
    tol = 2e-6 if dtype == torch.float32 else 8e-3
    check("y", got, want, tol)
    check("dx", xd.grad, xr.grad, tol)


@pytest.mark.parametrize("gate", [True, False])
@pytest.mark.parametrize("D,dtype", [(256, torch.float32), (128, torch.bfloat16), (512, torch.float32), (24, torch.float32)])
def test_rmsnorm_gate(D, dtype, gate):
    import mhla_amd
    g_ = torch.Generator().manual_seed(5)
    x = torch.randn(3, 37, 4, D, generator=g_).to(dtype)
    g = torch.randn(3, 37, 4, D, generator=g_).to(dtype) if gate else None
    w = torch.rand(D, generator=g_) + 0.5
    dy = torch.randn(3, 37, 4, D, generator=g_).to(dtype)
    xr, wr = x.float().clone().requires_grad_(True), w.clone().requires_grad_(True)
    gr = g.float().clone().requires_grad_(True) if gate else None
    if gate:
        yr = orc.rms_norm_swish_gate(xr, gr, wr, 1e-5)
    else:
        yr = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * wr
    (yr * dy.float()).sum().backward()
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    gd = g.to(DEV).requires_grad_(True) if gate else None
    y = mhla_amd.rmsnorm_gate(xd, gd, wd, 1e-5)
    y.backward(dy.to(DEV))
    lo = dtype != torch.float32
    check("y", y, yr.detach(), 8e-3 if lo else 1e-5)
    check("dx", xd.grad, xr.grad, 2e-2 if lo else 1e-4)
    check("dw", wd.grad, wr.grad, 2e-2 if lo else 1e-4)
    if gate:
        check("dg", gd.grad, gr.grad, 2e-2 if lo else 1e-4)


@pytest.mark.parametrize("T,K,V,gate,affine", [(300, 128, 256, True, True), (129, 64, 64, True, True), (200, 64, 128, False, True),
                                               (512, 128, 192, True, False),
                                               # wide heads: the workgroup walks the head in two halves (V = 512: the 1.3B-like fla shape)
                                               (330, 256, 512, True, True), (200, 64, 384, True, True), (130, 128, 512, False, False)])
@pytest.mark.parametrize("summaries", ["tf32", "split"])
def test_causal_normgate_fused_epilogue(T, K, V, gate, affine, summaries):
    """N1: per-head RMSNorm x swish gate inside the causal operator's output kernel (mhla_causal_normgate_fwd) vs the oracle's
    composition (causal_fwd -> rms_norm_swish_gate), forward and every gradient; and vs the unfused HIP composition."""
    import mhla_amd
    from mhla_amd import ops
    B, H, L = 2, 2, 8
    q, k, v, mix, do = causal_inputs(B, T, H, K, V, L, torch.bfloat16, seed=T + V)
    gen = torch.Generator().manual_seed(7)
    g = torch.randn(B, T, H, V, generator=gen).bfloat16() if gate else None
    w = (torch.rand(V, generator=gen) + 0.5) if affine else None
    assert ops.causal_normgate_fusable(q.to(DEV), v.to(DEV))
    # Oracle with the reference layer's dtype flow (layers/mhla.py:330-355): the operator returns o in the activation dtype
    # (naive.py:82), FusedRMSNormGated reads that bf16 o, and autograd hands the operator a bf16 do -- `_R16` rounds the value
    # on the way forward and the gradient on the way back.  Everything else in fp32.
    class _R16(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.bfloat16().float()

        @staticmethod
        def backward(ctx, g):
            return g.bfloat16().float()

    ref = [t.float().clone().requires_grad_(True) for t in (q, k, v, mix)]
    gr = g.float().clone().requires_grad_(True) if gate else None
    wr = w.clone().requires_grad_(True) if affine else None
    o_exact = orc.causal_fwd(ref[0], ref[1], ref[2], ref[3])
    o_ref = _R16.apply(o_exact)
    norm = lambda o: (orc.rms_norm_swish_gate(o, gr, wr if affine else torch.ones(V), 1e-5) if gate else
                      o * torch.rsqrt(o.pow(2).mean(-1, keepdim=True) + 1e-5) * (wr if affine else 1.0))
    y_ref = norm(o_ref)
    (y_ref * do.float()).sum().backward()
    dev = [t.to(DEV).requires_grad_(True) for t in (q, k, v, mix)]
    gd = g.to(DEV).requires_grad_(True) if gate else None
    wd = w.to(DEV).requires_grad_(True) if affine else None
    poison()
    y = mhla_amd.mhla_causal_normgate(dev[0], dev[1], dev[2], dev[3], gd, wd, 1e-5, summaries=summaries)
    assert y.dtype == torch.bfloat16
    poison()
    y.backward(do.to(DEV))
    u = 2.0 ** -8
    # y: the fused kernel normalises the fp32 accumulators (closer to the exact value than the reference flow, which normalises
    # the bf16-rounded o): against the exact composition one staging rounding of the normalised tile + the final one, 2u + 1e-3;
    # against the reference flow one more u for its rounding of o.
    with torch.no_grad():
        check("y vs exact composition", y, norm(o_exact), 2 * u + 1e-3)
    check("y", y, y_ref.detach(), 3 * u + 1e-3)
    # gradients: the norm's backward kernel and the operator's backward see the same rounded o / do as the oracle, up to
    # one-ulp flips of the bf16 tensor in between -- a `do` element that sits on a rounding boundary lands on the neighbouring
    # bf16 number (the two sides evaluate exp / rsqrt differently in the last fp32 bits), a perturbation of 2u of that element
    # which the reference's own Triton kernel has against eager PyTorch as well.  Observed: <= 1.1e-3 beyond the final rounding
    # (1.0e-3 on dmix over B H = 4 heads); bound: one final rounding + 2e-3, fp32-stored results 2e-3.
    # summaries="tf32" (the default, 11-bit summaries): the operator's o is delta = 3e-4 of its maximum away from the fp32 result
    # instead of 1e-5, so delta / u = 8 % of its bf16 elements land on the neighbouring bf16 number instead of 0.3 % -- each a
    # perturbation of 2u of that element, an rms of sqrt(delta u) = 1.1e-3 on the rounded o that the norm's backward and the
    # operator's backward then see (any 11-bit arithmetic in front of a bf16 store does this; it is the layered bf16 flow of
    # layers/mhla.py:330-355, not the operator, that amplifies).  Bound: + 2e-3 on top of the above; dmix, a difference of nearly
    # equal terms behind a norm (the loss does not depend on the scale of o), 6e-3 (observed 3.7e-3 on four chunks).  The operator
    # alone is held to one rounding + 1e-3 on the same shapes in test_causal_shapes_bf16.
    extra = 2e-3 if summaries == "tf32" else 0.0
    for name, a, b in zip(("dq", "dk", "dv"), dev, ref):
        check(name, a.grad, b.grad, u + 2e-3 + extra)
    check("dmix", dev[3].grad, ref[3].grad, 2e-3 + 2 * extra)
    if gate:
        check("dgate", gd.grad, gr.grad, u + 2e-3 + extra)
    if affine:
        check("dweight", wd.grad, wr.grad, 2e-3 + extra)
    # the unfused composition of the two HIP operators agrees (same kernels downstream, one more bf16 rounding of o)
    with torch.no_grad():
        y2 = mhla_amd.rmsnorm_gate(mhla_amd.mhla_causal(dev[0], dev[1], dev[2], dev[3], summaries=summaries), gd, wd, 1e-5)
        y3 = mhla_amd.mhla_causal_normgate(dev[0], dev[1], dev[2], dev[3], gd, wd, 1e-5, summaries=summaries)   # inference: o is not stored
    check("fused vs unfused", y3, y2.float().cpu(), 3 * u + 1e-3)
    check("inference vs training path", y3, y.detach().float().cpu(), 1e-6)


def test_golden_fla_neighbours_gate():
    import mhla_amd
    g = load_golden("fla_neighbours")
    y = mhla_amd.rmsnorm_gate(g["o"].to(DEV), g["g"].to(DEV), g["w"].to(DEV), 1e-5)
    check("gated", y, g["gated"], 1e-5)


@pytest.mark.parametrize("summaries", ["tf32", "split"])
def test_full_size_c5_sampled_head(summaries):
    """BASELINE config C5 shape (fla 340M: T = 8192, H = 4, K = 128, V = 256, 128 chunks), bf16: one (b, h) vs the oracle,
    forward and backward."""
    import mhla_amd
    B, T, H, K, V, L = 2, 8192, 4, 128, 256, 128
    q, k, v, mix, do = causal_inputs(B, T, H, K, V, L, torch.bfloat16, seed=8, random_mix=False)
    dq, dk, dv, dm = (t.to(DEV).requires_grad_(True) for t in (q, k, v, mix))
    out = mhla_amd.mhla_causal(dq, dk, dv, dm, summaries=summaries)
    out.backward(do.to(DEV))
    b, h = 1, 2
    sl = lambda t: t[b:b + 1, :, h:h + 1].float()
    s16 = lambda t: t.detach()[b:b + 1, :, h:h + 1]   # (bf16: check() charges the final rounding per element)
    want = orc.causal_fwd(sl(q), sl(k), sl(v), mix)
    wg = orc.causal_bwd(sl(q), sl(k), sl(v), mix, sl(do))
    check("out", s16(out), want, CAUSAL_TOL[torch.bfloat16])
    check("dq", s16(dq.grad), wg["dq"], CAUSAL_TOL[torch.bfloat16])
    check("dk", s16(dk.grad), wg["dk"], CAUSAL_TOL[torch.bfloat16])
    check("dv", s16(dv.grad), wg["dv"], CAUSAL_TOL[torch.bfloat16])
    # dmix sums over every (b, h): the whole batch through the oracle
    wg_all = orc.causal_bwd(q.float(), k.float(), v.float(), mix, do.float())
    check("dmix (all heads)", dm.grad, wg_all["dmix"], CAUSAL_DMIX_TOL[torch.bfloat16])


@pytest.mark.parametrize("summaries", ["tf32", "split"])
def test_full_size_c5_1p3b_like_shape(summaries):
    """The 1.3B-like fla shape of SURVEY.md 8 (K = 256, V = 512, T = 8192, 128 chunks), bf16: every output and gradient of
    one (b, h) and dmix over all heads vs the oracle (the K <= 256 token-gradient kernel with four K slices)."""
    import mhla_amd
    B, T, H, K, V, L = 1, 8192, 2, 256, 512, 128
    q, k, v, mix, do = causal_inputs(B, T, H, K, V, L, torch.bfloat16, seed=18, random_mix=True)
    dq, dk, dv, dm = (t.to(DEV).requires_grad_(True) for t in (q, k, v, mix))
    poison()
    out = mhla_amd.mhla_causal(dq, dk, dv, dm, summaries=summaries)
    poison()
    out.backward(do.to(DEV))
    want = orc.causal_fwd(q.float(), k.float(), v.float(), mix)
    wg = orc.causal_bwd(q.float(), k.float(), v.float(), mix, do.float())
    check("out", out, want, CAUSAL_TOL[torch.bfloat16])
    check("dq", dq.grad, wg["dq"], CAUSAL_TOL[torch.bfloat16])
    check("dk", dk.grad, wg["dk"], CAUSAL_TOL[torch.bfloat16])
    check("dv", dv.grad, wg["dv"], CAUSAL_TOL[torch.bfloat16])
    check("dmix", dm.grad, wg["dmix"], CAUSAL_DMIX_TOL[torch.bfloat16])
