"""GPU parity: drop-in modules load the reference modules' state dicts and reproduce their outputs."""
import pytest
import torch

from conftest import load_golden
from gpu_util import DEV, TOL, UNIT_ROUNDOFF as UNIT, check, poison
from oracle import mhla_oracle as orc

pytestmark = pytest.mark.gpu


def _sd(g):
    return {k[3:]: v for k, v in g.items() if k.startswith("sd.")}


def _check_module_grads(m, x, y, g, tol):
    """Module-level backward against the reference's own autograd (fixtures `dY`, `dx_mod`, `gsd.<param>`): d<y, dY>/dx and
    the gradient of every parameter."""
    for prm in m.parameters():
        prm.grad = None
    x.grad = None
    poison()
    (y * g["dY"].to(DEV)).sum().backward()
    check("dx (module)", x.grad, g["dx_mod"], tol)
    want = {k[4:]: v for k, v in g.items() if k.startswith("gsd.")}
    got = dict(m.named_parameters())
    assert set(want) <= set(got), sorted(set(want) - set(got))
    for name, ref in want.items():
        assert got[name].grad is not None, f"{name}: no gradient"
        check(f"grad {name}", got[name].grad.reshape(ref.shape), ref, tol, atol=1e-7)


@pytest.mark.parametrize("tag,cls,kw", [
    ("dit_a", "MHLA4DiT", dict(heads=2, dim_head=32, block_size=16, embed_len=256, qkv_bias=True, transform="linear")),
    ("dit_b", "MHLA4DiT", dict(heads=1, dim_head=72, block_size=49, embed_len=441, qkv_bias=True, transform="exp")),
    ("vit_a", "MHLA_Normed_Torch", dict(heads=2, dim_head=64, window_size=16, embed_len=256, qk_norm=True)),
])
def test_dit_vit_module_matches_reference(tag, cls, kw):
    from mhla_amd import modules
    g = load_golden("blockmix2d_" + tag)
    dim = kw["heads"] * kw["dim_head"]
    m = getattr(modules, cls)(dim, dropout=0.0, **kw)
    missing = m.load_state_dict(_sd(g), strict=True)
    m = m.to(DEV).eval()
    x = g["x"].to(DEV).requires_grad_(True)
    y = m(x)
    check("y", y, g["y"], 1e-4)
    # 3-D [B, N, C] entry used by the DiT host gives the same tokens
    y3 = m(x.reshape(x.shape[0], -1, x.shape[-1]))
    assert torch.equal(y3.reshape(y.shape), y)
    _check_module_grads(m, x, y, g, 2e-4)


def test_dit_module_with_head_dim_above_128_matches_oracle_restatement():
    """dim_head = 192 (the reference module takes any): the drop-in routes the operator through the sliced composition
    (ops._blockmix_wide_head) and LePE separately; output, dx and every parameter gradient against the oracle's module restatement
    run under autograd on the CPU."""
    from mhla_amd import modules
    torch.manual_seed(3)
    heads, dh, bs, el = 2, 192, 16, 64
    m = modules.MHLA4DiT(heads * dh, heads=heads, dim_head=dh, block_size=bs, embed_len=el, qkv_bias=True, transform="linear")
    with torch.no_grad():
        m.lepe.weight.normal_(0, 0.2)
        m.piece_attn.conv.weight.copy_(torch.rand_like(m.piece_attn.conv.weight))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, el // bs, bs, heads * dh)
    dy = torch.randn(2, el // bs, bs, heads * dh)
    # oracle restatement on the CPU, gradients by autograd
    sdr = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = orc.dit_module_forward(sdr, xr, heads, bs, el)
    (yr * dy).sum().backward()
    m = m.to(DEV).eval()
    xg = x.to(DEV).requires_grad_(True)
    y = m(xg)
    (y * dy.to(DEV)).sum().backward()
    check("y", y, yr.detach(), 2e-4)
    check("dx", xg.grad, xr.grad, 5e-4)
    for name, prm in m.named_parameters():
        if sdr[name].grad is not None:
            check(f"grad {name}", prm.grad, sdr[name].grad, 5e-4, atol=1e-7)


def test_dit_module_initial_weights_match_reference_init():
    from mhla_amd import modules
    g = load_golden("weight_init")
    m = modules.MHLA4DiT(128, heads=2, block_size=16, embed_len=256, transform="cos")
    assert torch.allclose(m.piece_attn.get_weight_matrix(), g["w2d_cos_16_16"], atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_wan_module_matches_reference(tag):
    from mhla_amd import modules
    g = load_golden("wan_" + tag)
    B, H, D, M, S, fb, hb, wb, F_, H_, W_, normalize, gated = [int(x) for x in g["meta"]]
    m = modules.MHLA_Video_Uni(H * D, num_heads=H, block_layout=(fb, hb, wb), normalize_out=bool(normalize),
                               is_gated=bool(gated))
    m.load_state_dict(_sd(g), strict=True)
    m = m.to(DEV).eval()
    x = g["x"].to(DEV).requires_grad_(True)
    N = F_ * H_ * W_
    grid_sizes = torch.tensor([[F_, H_, W_]] * B, dtype=torch.long)
    y = m(x, torch.tensor([N] * B), grid_sizes, modules.wan_freqs(D))
    check("y", y, g["y"], 1e-4)
    _check_module_grads(m, x, y, g, 2e-4)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_wan_module_fused_inference_path(tag):
    """Under no_grad the module takes the fused prologue (norm + relu + eps kernel, rotation inside the operator's loads):
    same result as the reference fixture and as the unfused training path."""
    from mhla_amd import modules
    g = load_golden("wan_" + tag)
    B, H, D, M, S, fb, hb, wb, F_, H_, W_, normalize, gated = [int(x) for x in g["meta"]]
    m = modules.MHLA_Video_Uni(H * D, num_heads=H, block_layout=(fb, hb, wb), normalize_out=bool(normalize),
                               is_gated=bool(gated))
    m.load_state_dict(_sd(g), strict=True)
    m = m.to(DEV).eval()
    x = g["x"].to(DEV)
    N = F_ * H_ * W_
    grid_sizes = torch.tensor([[F_, H_, W_]] * B, dtype=torch.long)
    import mhla_amd.modules.wan as wanmod
    calls = []
    orig = wanmod.mhla_blockmix_wan
    wanmod.mhla_blockmix_wan = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            y = m(x, torch.tensor([N] * B), grid_sizes, modules.wan_freqs(D))
    finally:
        wanmod.mhla_blockmix_wan = orig
    assert calls, "fused path not taken under no_grad"
    check("y", y, g["y"], 1e-4)
    y2 = m(x.clone().requires_grad_(True), torch.tensor([N] * B), grid_sizes, modules.wan_freqs(D))
    check("fused vs unfused", y, y2.detach().cpu(), 1e-4)


@pytest.mark.parametrize("dtype,out_rows_partial", [(torch.bfloat16, False), (torch.float16, True)])
@pytest.mark.parametrize("M,S,gate,rope,gather", [(50, 210, True, True, False), (80, 130, False, False, True), (44, 251, True, True, True)])
def test_wan_flat_tile_list_matches_the_block_per_workgroup_kernel(M, S, gate, rope, gather, dtype, out_rows_partial):
    """The Wan inference output kernel as one persistent workgroup per CU over a flat list of token tiles (split.hpp k_sp_out<.., FLAT>:
    blocks of at least 128 tokens, enough tiles to cut) against the block-per-workgroup launch it replaced
    (mhla_set_option("recut_kernels", 0)) and against the materialising composition: per tile the same numbers in the same order, so all
    three agree to the last bit.  Block lengths that are not multiples of 16 (210, 130, 251: partial last tiles), ranges that start and end
    inside blocks, a gather map."""
    import mhla_amd
    from mhla_amd import ops
    B, H, D = 1, 12, 128
    N, C = M * S, H * D
    g = torch.Generator().manual_seed(M * 1000 + S)
    q, k, v = (torch.randn(B, N, C, generator=g).to(dtype).to(DEV) for _ in range(3))
    wq, wk = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    W = torch.rand(M, M, generator=g).to(DEV)
    nw = (torch.rand(D, generator=g) + 0.5).to(DEV)
    gt = torch.randn(B, N, H, D, generator=g).to(dtype).to(DEV) if gate else None
    cos = sin = None
    if rope:
        ang = torch.rand(N, D // 2, generator=g) * 6.28
        cos, sin = torch.cos(ang).to(DEV), torch.sin(ang).to(DEV)
    idx = torch.randperm(N, generator=torch.Generator().manual_seed(3)).int().to(DEV) if gather else None
    r4 = lambda t: t.reshape(B, N, H, D)
    assert ops.wan_pro_supported(r4(q), M)
    assert B * H * M * ((S + 15) // 16) >= 256 * 32 and S >= 113, "the shape must take the flat tile list"
    run = lambda: mhla_amd.mhla_blockmix_wan_pro(r4(q), r4(k), r4(v), wq, wk, 1e-6, W, cos, sin, nw, 1e-6, gt, eps=1e-6, normalize=True,
                                                 block_index=idx, qk_norm=True)
    poison()
    y = run()
    prev = mhla_amd.set_option("recut_kernels", 0)
    try:
        poison()
        y0 = run()
    finally:
        mhla_amd.set_option("recut_kernels", prev)
    assert torch.equal(y, y0), f"flat tile list differs from the block-per-workgroup kernel: max |diff| {(y.float() - y0.float()).abs().max().item():.3e}"
    qf = r4(mhla_amd.qk_prologue(q, wq, 1e-6, 1e-6))
    kf = r4(mhla_amd.qk_prologue(k, wk, 1e-6, 1e-6))
    y2 = mhla_amd.mhla_blockmix_wan(qf, kf, r4(v).float(), W, cos, sin, nw, 1e-6, gt, dtype, eps=1e-6, normalize=True, block_index=idx)
    assert torch.equal(y, y2), f"flat tile list differs from the materialised prologue: max |diff| {(y.float() - y2.float()).abs().max().item():.3e}"
    assert torch.isfinite(y.float()).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,S,gate,rope,norm,normalize,gather", [(6, 40, True, True, True, True, True), (40, 21, False, True, True, False, False),
                                                                 (150, 6, True, False, True, True, True), (9, 33, True, True, False, True, False)])
def test_wan_prologue_on_load_matches_materialised_prologue(M, S, gate, rope, norm, normalize, gather, dtype):
    """SURVEY N2 / VERDICT r5 item 3: mhla_blockmix_wan_pro reads the 16-bit q, k, v projections in place and applies
    relu(rmsnorm_C(x) w) + eps while loading (mhla_rms_rstd gives the per-token rstd) -- against the composition it replaces,
    mhla_qk_prologue (fp32 q, k materialised) + v.float() + mhla_blockmix_wan: the same fp32 numbers in the same order, so the two
    agree to the last bit of the 16-bit output; and against the oracle's module arithmetic (wan/mhla_utils.py:268-272, :308-362)."""
    import mhla_amd
    from mhla_amd import ops
    B, H, D = 2, 3, 128
    N, C = M * S, H * D
    g = torch.Generator().manual_seed(M * 1000 + S)
    q, k, v = (torch.randn(B, N, C, generator=g).to(dtype).to(DEV) for _ in range(3))
    wq, wk = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    W = torch.rand(M, M, generator=g).to(DEV)
    nw = (torch.rand(D, generator=g) + 0.5).to(DEV)
    gt = torch.randn(B, N, H, D, generator=g).to(dtype).to(DEV) if gate else None
    cos = sin = None
    if rope:
        ang = torch.rand(N, D // 2, generator=g) * 6.28
        cos, sin = torch.cos(ang).to(DEV), torch.sin(ang).to(DEV)
    idx = torch.randperm(N, generator=torch.Generator().manual_seed(3)).int().to(DEV) if gather else None
    r4 = lambda t: t.reshape(B, N, H, D)
    assert ops.wan_pro_supported(r4(q), M)
    poison()
    y = mhla_amd.mhla_blockmix_wan_pro(r4(q), r4(k), r4(v), wq if norm else None, wk if norm else None, 1e-6, W, cos, sin, nw, 1e-6, gt,
                                       eps=1e-6, normalize=normalize, block_index=idx, qk_norm=norm)
    assert y.dtype == dtype and y.shape == (B, N, H, D)
    qf = r4(mhla_amd.qk_prologue(q, wq if norm else None, 1e-6, 1e-6))
    kf = r4(mhla_amd.qk_prologue(k, wk if norm else None, 1e-6, 1e-6))
    y2 = mhla_amd.mhla_blockmix_wan(qf, kf, r4(v).float(), W, cos, sin, nw, 1e-6, gt, dtype, eps=1e-6, normalize=normalize, block_index=idx)
    assert torch.equal(y, y2), f"prologue on load differs from the materialised prologue: max |diff| {(y.float() - y2.float()).abs().max().item():.3e}"
    # ... and the oracle: fp32 prologue, rope, operator (gather map undone), per-head norm x gate
    qc, kc, vc = q.float().cpu(), k.float().cpu(), v.float().cpu()
    if norm:
        qc = orc.rms_norm(qc, wq.cpu(), 1e-6)
        kc = orc.rms_norm(kc, wk.cpu(), 1e-6)
    qc, kc = r4(torch.relu(qc) + 1e-6), r4(torch.relu(kc) + 1e-6)
    vc = r4(vc)

    def rot(x):
        if not rope:
            return x
        xs = x.reshape(B, N, H, D // 2, 2)
        c, s_ = cos.cpu()[None, :, None, :], sin.cpu()[None, :, None, :]
        return torch.stack((xs[..., 0] * c - xs[..., 1] * s_, xs[..., 0] * s_ + xs[..., 1] * c), dim=-1).reshape(B, N, H, D)
    perm = idx.long().cpu() if gather else torch.arange(N)
    bm = lambda t: t[:, perm]            # block-major order
    o = orc.blockmix_fwd(bm(rot(qc)), bm(rot(kc)), bm(vc), W.cpu(), 1e-6, bm(qc) if normalize else None, bm(kc) if normalize else None, normalize)
    o = o.to(dtype).float()
    want_bm = orc.rms_norm_swish_gate(o, bm(gt.float().cpu()), nw.cpu(), 1e-6) if gate else orc.rms_norm(o, nw.cpu(), 1e-6)
    want = torch.empty_like(want_bm)
    want[:, perm] = want_bm
    check("y vs oracle", y, want, TOL[dtype] + UNIT[dtype])   # (the host rounds O to the activation dtype before the norm: one more rounding)


def test_wan_module_bf16_inference_takes_the_prologue_on_load_path():
    """A bf16 MHLA_Video_Uni under no_grad (the Wan2.1 inference configuration) routes through mhla_blockmix_wan_pro -- no k_qk_prologue,
    no fp32 q / k / v -- and agrees with the same module's materialising path."""
    from mhla_amd import modules
    import mhla_amd.modules.wan as wanmod
    B, H, D, layout, grid = 1, 4, 128, (2, 2, 3), (4, 6, 9)
    N = grid[0] * grid[1] * grid[2]
    torch.manual_seed(5)
    m = modules.MHLA_Video_Uni(H * D, num_heads=H, block_layout=layout, is_gated=True).to(DEV).to(torch.bfloat16).eval()
    x = torch.randn(B, N, H * D, device=DEV, dtype=torch.bfloat16)
    args = (torch.tensor([N] * B), torch.tensor([list(grid)] * B, dtype=torch.long), modules.wan_freqs(D))
    calls = []
    orig = wanmod.mhla_blockmix_wan_pro
    wanmod.mhla_blockmix_wan_pro = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            y = m(x, *args)
            wanmod.wan_pro_supported, keep = (lambda *a, **k: False), wanmod.wan_pro_supported
            try:
                y2 = m(x, *args)
            finally:
                wanmod.wan_pro_supported = keep
    finally:
        wanmod.mhla_blockmix_wan_pro = orig
    assert len(calls) == 1, "the prologue-on-load path was not taken"
    assert torch.equal(y, y2)


@pytest.mark.parametrize("kind", ["mhla", "mhla_nope", "gated_mhla", "mhla_lepe", "gated_mhla_lepe"])
def test_older_wan_variants_match_reference(kind):
    """the five older Wan MHLA classes behind the registry keys of wan/model.py:1592-1605 -- strict load of the
    reference class's state dict, module output (training and no_grad paths), d<y, dY>/dx and every parameter gradient against the
    reference class's own autograd."""
    from mhla_amd import modules
    g = load_golden("wanv_" + kind)
    B, H, D, fb, hb, wb, F_, H_, W_, normalize, out_rms = [int(x) for x in g["meta"]]
    cls = modules.WAN_SELFATTENTION_CLASSES[kind]
    m = cls(H * D, num_heads=H, block_layout=(fb, hb, wb), normalize_out=bool(normalize), out_rmsnorm=bool(out_rms))
    m.load_state_dict(_sd(g), strict=True)
    m = m.to(DEV).eval()
    N = F_ * H_ * W_
    args = (torch.tensor([N] * B), torch.tensor([[F_, H_, W_]] * B, dtype=torch.long), modules.wan_freqs(D))
    with torch.no_grad():
        y_inf = m(g["x"].to(DEV), *args)
    check("y (no_grad path)", y_inf, g["y"], 1e-4)
    x = g["x"].to(DEV).requires_grad_(True)
    poison()
    y = m(x, *args)
    check("y", y, g["y"], 1e-4)
    _check_module_grads(m, x, y, g, 2e-4)


def test_wan_module_with_lepe_matches_reference():
    """is_lepe=True (wan/mhla_utils.py:226-231, 283-285, 363-364): the Conv3d branch runs on the HIP 3-D LePE kernels; module
    output and d(sum y)/dx against the reference fixture, training and fused inference paths."""
    from mhla_amd import modules
    g = load_golden("wan_c")
    assert int(g["is_lepe"][0]) == 1
    B, H, D, M, S, fb, hb, wb, F_, H_, W_, normalize, gated = [int(x) for x in g["meta"]]
    m = modules.MHLA_Video_Uni(H * D, num_heads=H, block_layout=(fb, hb, wb), normalize_out=bool(normalize),
                               is_gated=bool(gated), is_lepe=True)
    m.load_state_dict(_sd(g), strict=True)
    m = m.to(DEV).eval()
    N = F_ * H_ * W_
    grid_sizes = torch.tensor([[F_, H_, W_]] * B, dtype=torch.long)
    import mhla_amd.modules.wan as wanmod
    calls = []
    orig = wanmod.lepe3d
    wanmod.lepe3d = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        x = g["x"].to(DEV).requires_grad_(True)
        y = m(x, torch.tensor([N] * B), grid_sizes, modules.wan_freqs(D))
        y.sum().backward()
        with torch.no_grad():
            y_inf = m(g["x"].to(DEV), torch.tensor([N] * B), grid_sizes, modules.wan_freqs(D))
    finally:
        wanmod.lepe3d = orig
    assert len(calls) == 2, "HIP LePE not on the module's path"
    check("y", y, g["y"], 1e-4)
    check("y (fused inference)", y_inf, g["y"], 1e-4)
    check("dx", x.grad, g["dx"], 2e-4)
    assert m.lepe.weight.grad is not None and m.lepe.bias.grad is not None
    x2 = g["x"].to(DEV).requires_grad_(True)
    _check_module_grads(m, x2, m(x2, torch.tensor([N] * B), grid_sizes, modules.wan_freqs(D)), g, 2e-4)


@pytest.mark.parametrize("grid,C,dtype", [((4, 6, 9), 64, torch.float32), ((1, 5, 7), 24, torch.float32), ((3, 1, 1), 8, torch.float32),
                                          ((5, 8, 10), 1536, torch.bfloat16), ((2, 3, 4), 136, torch.float16)])
def test_lepe3d_matches_conv3d(grid, C, dtype):
    """HIP 3-D LePE kernels on the raster token layout vs nn.functional.conv3d on the rearranged video
    (wan/mhla_utils.py:283-285), V as a strided slice of a packed buffer; gradients w.r.t. v, weight, bias, add."""
    import torch.nn.functional as F
    import mhla_amd
    F_, H_, W_ = grid
    g = torch.Generator().manual_seed(C + F_)
    B, N = 2, F_ * H_ * W_
    qkv = torch.randn(B, N, 3, C, generator=g).to(dtype)
    w = (torch.randn(C, 1, 3, 3, 3, generator=g) * 0.2).to(dtype)
    bias = torch.randn(C, generator=g).to(dtype)
    add = torch.randn(B, N, C, generator=g).to(dtype)
    dy = torch.randn(B, N, C, generator=g).to(dtype)
    rv, rw, rb, ra = (t.float().clone().requires_grad_(True) for t in (qkv[:, :, 2], w, bias, add))
    vid = rv.reshape(B, F_, H_, W_, C).permute(0, 4, 1, 2, 3)
    want = F.conv3d(vid, rw, rb, padding=1, groups=C).permute(0, 2, 3, 4, 1).reshape(B, N, C) + ra
    want.backward(dy.float())
    dq = qkv.to(DEV).requires_grad_(True)
    dw, db, da = (t.to(DEV).requires_grad_(True) for t in (w, bias, add))
    poison()
    got = mhla_amd.lepe3d(dq[:, :, 2], dw, db, grid, add=da)
    assert got.dtype == dtype
    poison()
    got.backward(dy.to(DEV))
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    check("y", got, want, tol)
    check("dv", dq.grad[:, :, 2], rv.grad, tol)
    assert float(dq.grad[:, :, :2].abs().max()) == 0.0
    check("dw", dw.grad, rw.grad, tol)
    check("dbias", db.grad, rb.grad, tol)
    check("dadd", da.grad, dy, 1e-6)


@pytest.mark.parametrize("K,pl,bl,C,dtype", [(3, 4, 4, 64, torch.float32), (5, 2, 7, 48, torch.float32), (3, 4, 4, 1152, torch.bfloat16),
                                             (5, 4, 4, 384, torch.bfloat16), (3, 1, 6, 8, torch.float32)])
def test_lepe2d_matches_conv2d(K, pl, bl, C, dtype):
    """HIP LePE kernels on the block-major token layout vs nn.Conv2d on the rearranged image (mhla.py:246-247), with
    V as a strided slice of a fused QKV buffer; gradients w.r.t. v, weight, bias and the added tensor."""
    import torch.nn.functional as F
    import mhla_amd
    g = torch.Generator().manual_seed(K * 100 + C)
    B, N = 3, (pl * bl) ** 2
    qkv = torch.randn(B, N, 3, C, generator=g).to(dtype)
    w = (torch.randn(C, 1, K, K, generator=g) * 0.3).to(dtype)
    bias = torch.randn(C, generator=g).to(dtype)
    add = torch.randn(B, N, C, generator=g).to(dtype)
    dy = torch.randn(B, N, C, generator=g).to(dtype)

    def to_img(t):      # [B, N, C] block-major -> [B, C, side, side]
        return t.reshape(B, pl, pl, bl, bl, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, pl * bl, pl * bl)

    def from_img(t):
        return t.reshape(B, C, pl, bl, pl, bl).permute(0, 2, 4, 3, 5, 1).reshape(B, N, C)

    rv, rw, rb, ra = (t.float().clone().requires_grad_(True) for t in (qkv[:, :, 2], w, bias, add))
    want = from_img(F.conv2d(to_img(rv), rw, rb, padding=K // 2, groups=C)) + ra
    want.backward(dy.float())
    dq = qkv.to(DEV).requires_grad_(True)
    dw, db, da = (t.to(DEV).requires_grad_(True) for t in (w, bias, add))
    got = mhla_amd.lepe2d(dq[:, :, 2], dw, db, pl, bl, add=da)
    assert got.dtype == dtype
    got.backward(dy.to(DEV))
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    check("y", got, want, tol)
    check("dv", dq.grad[:, :, 2], rv.grad, tol)
    assert float(dq.grad[:, :, :2].abs().max()) == 0.0
    check("dw", dw.grad, rw.grad, tol)
    check("dbias", db.grad, rb.grad, tol)
    check("dadd", da.grad, dy, 1e-6)


@pytest.mark.parametrize("H,D,pl,bl,K,dtype", [(16, 72, 4, 4, 3, torch.bfloat16), (2, 64, 2, 7, 5, torch.float32), (4, 64, 8, 8, 3, torch.bfloat16)])
def test_dit_core_packed_node_matches_composition(H, D, pl, bl, K, dtype):
    """mhla_dit_core (operator + LePE on the packed QKV buffer, one packed gradient) == mhla_blockmix + lepe2d."""
    import mhla_amd
    g = torch.Generator().manual_seed(H * D)
    B, M, S = 2, pl * pl, bl * bl
    N, C = M * S, H * D
    qkv = torch.randn(B, N, 3, H, D, generator=g).to(dtype)
    W = orc.block_distance_weights((pl, pl), "linear")
    lw = (torch.randn(C, 1, K, K, generator=g) * 0.2).to(dtype)
    lb = torch.randn(C, generator=g).to(dtype)
    dy = torch.randn(B, N, C, generator=g).to(dtype)
    res = []
    for packed in (True, False):
        t = [x.clone().to(DEV).requires_grad_(True) for x in (qkv, W, lw, lb)]
        poison()
        if packed:
            y = mhla_amd.mhla_dit_core(t[0], t[1], t[2], t[3], pl, bl, eps=1e-6, relu_eps=True)
        else:
            o = mhla_amd.mhla_blockmix(t[0][:, :, 0], t[0][:, :, 1], t[0][:, :, 2], t[1], eps=1e-6, relu_eps=True)
            y = mhla_amd.lepe2d(t[0][:, :, 2].reshape(B, N, C), t[2], t[3], pl, bl, add=o.reshape(B, N, C))
        poison()
        y.backward(dy.to(DEV))
        res.append([y] + [x.grad for x in t])
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    for name, a, b in zip(("y", "dqkv", "dW", "dlepe_w", "dlepe_b"), *res):
        check(name, a, b.float().cpu(), tol)


def test_qk_prologue_and_rope_op():
    """mhla_qk_prologue vs the oracle's rms_norm + relu_eps; mhla_blockmix_rope vs rope_apply + the split-pair op."""
    import mhla_amd
    g = torch.Generator().manual_seed(21)
    B, F_, H_, W_, H, D = 1, 4, 6, 10, 3, 64
    N, C = F_ * H_ * W_, H * D
    for dtype in (torch.bfloat16, torch.float32):
        x = torch.randn(B, N, C, generator=g).to(dtype)
        w = (torch.rand(C, generator=g) + 0.5)
        want = orc.relu_eps(orc.rms_norm(x.float(), w, 1e-5), 1e-6)
        got = mhla_amd.qk_prologue(x.to(DEV), w.to(DEV), 1e-5, 1e-6)
        assert got.dtype == torch.float32
        check("prologue", got, want, 1e-5)
        check("prologue no norm", mhla_amd.qk_prologue(x.to(DEV), None, 0.0, 1e-6), torch.relu(x.float()) + 1e-6, 1e-6)
    q = torch.rand(B, N, H, D, generator=g) + 1e-6
    k = torch.rand(B, N, H, D, generator=g) + 1e-6
    v = torch.randn(B, N, H, D, generator=g)
    layout = (2, 3, 5)
    W = orc.block_distance_weights(layout, "linear")
    idx = orc.block_index_3d((F_, H_, W_), layout).int()
    freqs = orc.wan_freqs(D)
    qr, kr = orc.wan_rope_apply(q, (F_, H_, W_), freqs), orc.wan_rope_apply(k, (F_, H_, W_), freqs)
    gather = lambda t: t[:, idx.long()]
    from mhla_amd.modules.wan import _rope_table
    cos, sin = _rope_table(freqs, (F_, H_, W_), DEV)
    for normalize in (True, False):
        want = orc.blockmix_fwd(gather(qr), gather(kr), gather(v), W, 1e-6, q_den=gather(q), k_den=gather(k), normalize=normalize)
        got = mhla_amd.mhla_blockmix_rope(q.to(DEV), k.to(DEV), v.to(DEV), W.to(DEV), cos, sin, eps=1e-6, normalize=normalize,
                                          block_index=idx.to(DEV))
        check(f"rope op normalize={normalize}", got[:, idx.long().to(DEV)], want, 1e-4)
    # backward (mhla_blockmix_rope_bwd): gradients w.r.t. the UN-rotated q, k -- the transposed rotation of the numerator pair's
    # gradients plus the normaliser pair's -- against autograd through the oracle's rope_apply + split-pair operator
    do = torch.randn(B, N, H, D, generator=g)
    for normalize in (True, False):
        ref = [t.clone().requires_grad_(True) for t in (q, k, v, W)]
        qr_, kr_ = orc.wan_rope_apply(ref[0], (F_, H_, W_), freqs), orc.wan_rope_apply(ref[1], (F_, H_, W_), freqs)
        o_ref = orc.blockmix_fwd(gather(qr_), gather(kr_), gather(ref[2]), ref[3], 1e-6, q_den=gather(ref[0]), k_den=gather(ref[1]),
                                 normalize=normalize)
        (o_ref * gather(do)).sum().backward()
        dev = [t.to(DEV).requires_grad_(True) for t in (q, k, v, W)]
        poison()
        o = mhla_amd.mhla_blockmix_rope(*dev, cos, sin, eps=1e-6, normalize=normalize, block_index=idx.to(DEV))
        poison()
        o.backward(do.to(DEV))
        for name, a_, b_ in zip(("dq", "dk", "dv", "dW"), dev, ref):
            check(f"rope op {name} normalize={normalize}", a_.grad, b_.grad, 2e-4)
    with pytest.raises(RuntimeError):   # the rotary backward is built for fp32 tensors
        mhla_amd.mhla_blockmix_rope(q.to(DEV).bfloat16().requires_grad_(True), k.to(DEV).bfloat16(), v.to(DEV).bfloat16(), W.to(DEV), cos, sin)
    # prologue + epilogue fused: rotary inside, per-head RMSNorm x SiLU gate before the store, output in the host dtype
    nw = torch.rand(D, generator=g) + 0.5
    for odt in (torch.bfloat16, torch.float32):
        for gated in (True, False):
            gate = torch.randn(B, N, H, D, generator=g).to(odt) if gated else None
            o = mhla_amd.mhla_blockmix_rope(q.to(DEV), k.to(DEV), v.to(DEV), W.to(DEV), cos, sin, eps=1e-6, block_index=idx.to(DEV))
            want = mhla_amd.rmsnorm_gate(o.to(odt), gate.to(DEV) if gated else None, nw.to(DEV), 1e-5)
            got = mhla_amd.mhla_blockmix_wan(q.to(DEV), k.to(DEV), v.to(DEV), W.to(DEV), cos, sin, nw.to(DEV), 1e-5,
                                             gate.to(DEV) if gated else None, odt, eps=1e-6, block_index=idx.to(DEV))
            assert got.dtype == odt
            check(f"wan fused {odt} gate={gated}", got, want.float().cpu(), 1e-5 if odt == torch.float32 else 8e-3)


@pytest.mark.parametrize("dtype,C,D,norm", [(torch.float32, 256, 64, True), (torch.bfloat16, 1536, 128, True), (torch.float32, 192, 32, False)])
def test_qk_prologue_backward_and_rope_output(dtype, C, D, norm):
    """Differentiable prologue: y = relu(rmsnorm(x) w) + eps and y_rope = rope(y) in one kernel; gradients w.r.t. x and w
    from both outputs vs autograd through the oracle's rms_norm / relu_eps / wan_rope_apply."""
    import mhla_amd
    g = torch.Generator().manual_seed(C + D)
    B, grid = 2, (2, 3, 5)
    N = grid[0] * grid[1] * grid[2]
    H = C // D
    x = torch.randn(B, N, C, generator=g).to(dtype)
    w = (torch.rand(C, generator=g) + 0.5) if norm else None
    dy, dyr = torch.randn(B, N, C, generator=g), torch.randn(B, N, C, generator=g)
    freqs = orc.wan_freqs(D)
    from mhla_amd.modules.wan import _rope_table
    cos, sin = _rope_table(freqs, grid, DEV)
    xr = x.float().clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True) if norm else None
    yref = orc.relu_eps(orc.rms_norm(xr, wr, 1e-5) if norm else xr, 1e-6)
    yrref = orc.wan_rope_apply(yref.reshape(B, N, H, D), grid, freqs).reshape(B, N, C)
    (yref * dy).sum().backward(retain_graph=True)
    gx_y, gw_y = xr.grad.clone(), (wr.grad.clone() if norm else None)
    (yrref * dyr).sum().backward()
    xd = x.detach().to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True) if norm else None
    poison()
    y, yr = mhla_amd.qk_prologue(xd, wd, 1e-5, 1e-6, rope=(cos, sin), head_dim=D)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    check("y", y, yref.detach(), 1e-5)
    check("y_rope", yr, yrref.detach(), 1e-5)
    poison()
    ((y * dy.to(DEV)).sum() + (yr * dyr.to(DEV)).sum()).backward()
    check("dx", xd.grad, xr.grad, tol)
    if norm:
        check("dw", wd.grad, wr.grad, tol)
    # single-output form
    xd2 = x.detach().to(DEV).requires_grad_(True)
    wd2 = w.to(DEV).requires_grad_(True) if norm else None
    y2 = mhla_amd.qk_prologue(xd2, wd2, 1e-5, 1e-6)
    (y2 * dy.to(DEV)).sum().backward()
    check("dx (y only)", xd2.grad, gx_y, tol)
    if norm:
        check("dw (y only)", wd2.grad, gw_y, tol)


def test_fla_layer_matches_oracle_restatement():
    """The reference fla layer cannot run on CPU (Triton neighbours); it is pinned piecewise by
    golden vectors (rotary, norm-gate, causal op) and the oracle composes them (fla_layer_forward)."""
    from mhla_amd import modules
    torch.manual_seed(3)
    m = modules.MHLA(mode="chunk", hidden_size=256, expand_k=0.5, expand_v=1.0, num_heads=2, feature_map="relu",
                     norm_eps=1e-6)
    with torch.no_grad():
        m.g_norm_swish_gate.weight.uniform_(0.5, 1.5)
        m.mixing_matrix.copy_(torch.rand(32, 32).view(32, 32, 1, 1, 1, 1))
    x = torch.randn(2, 200, 256)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    want = orc.fla_layer_forward(sd, x, 2, 64, 128, norm_eps=1e-6)
    m = m.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    o, attn, cache = m(xd)
    assert attn is None and cache is None
    check("o", o, want, 1e-4)
    # layer-level gradients: autograd through the oracle's restatement of the layer (fp64 would change nothing: fp32 ref)
    dY = torch.randn(o.shape, generator=torch.Generator().manual_seed(9))
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    sdr["mixing_matrix"] = sd["mixing_matrix"].clamp(1e-5, 1).clone().requires_grad_(True)   # the forward's in-place clamp
    xr = x.clone().requires_grad_(True)
    (orc.fla_layer_forward(sdr, xr, 2, 64, 128, norm_eps=1e-6) * dY).sum().backward()
    (o * dY.to(DEV)).sum().backward()
    check("dx (layer)", xd.grad, xr.grad, 2e-4)
    for name, prm in m.named_parameters():
        check(f"grad {name}", prm.grad, sdr[name].grad, 2e-4, atol=1e-7)
    xd.grad = None
    # forward clamps the mixing weights in place on .data (layers/mhla.py:237); the reference's .tril() there
    # acts on the trailing 1x1 dims of the [32,32,1,1,1,1] parameter, i.e. it is a no-op, reproduced as is
    mm = m.mixing_matrix.detach().reshape(32, 32)
    assert mm.min() >= 1e-5 and mm.max() <= 1
    # short sequences (T <= 64) and padded batches run too
    o2, _, _ = m(xd[:, :50])
    want2 = orc.fla_layer_forward(sd, x[:, :50], 2, 64, 128, norm_eps=1e-6)
    check("o_short", o2, want2, 1e-4)
    mask = torch.ones(2, 200, dtype=torch.long)
    mask[1, 150:] = 0
    o3, _, _ = m(xd, attention_mask=mask.to(DEV))
    assert o3.shape == o.shape and torch.all(o3[1, 150:] == 0)
    # padded batch: one packed sequence, rotary positions restarting per sequence (cu_seqlens), operator over the packed tokens
    want3 = orc.fla_layer_forward(sd, x, 2, 64, 128, norm_eps=1e-6, attention_mask=mask)
    check("o_padded", o3, want3, 1e-4)
    # cache protocol (:339-345): the layer reports its tokens to the cache and hands the recurrent state over for T <= 64

    class Cache:
        def __init__(self):
            self.seen, self.states = 0, []

        def __len__(self):
            return len(self.states)

        def __getitem__(self, i):
            return self.states[i]

        def get_seq_length(self, layer_idx=0):
            return self.seen

        def update(self, recurrent_state=None, conv_state=None, layer_idx=0, offset=1):
            self.states = [dict(recurrent_state=recurrent_state, conv_state=conv_state)]
            self.seen += offset

    m.layer_idx = 0
    cache = Cache()
    o4, _, c4 = m(xd[:, :40], past_key_values=cache, use_cache=True)
    assert c4 is cache and cache.seen == 40 and cache.states[0]["recurrent_state"].shape == (2, 2, 64, 128)
    check("o_cached_first_call", o4, orc.fla_layer_forward(sd, x[:, :40], 2, 64, 128, norm_eps=1e-6), 1e-4)
    # longer sequences than the reference's 32-chunk matrix allows: the max_chunks knob (8192 tokens = 128 chunks)
    m128 = modules.MHLA(mode="chunk", hidden_size=128, expand_k=0.5, expand_v=1.0, num_heads=2, feature_map="relu", max_chunks=128).to(DEV)
    assert m128.mixing_matrix.shape == (128, 128, 1, 1, 1, 1)
    xl = torch.randn(1, 8192, 128, device=DEV)
    ol, _, _ = m128(xl)
    sdl = {k: v.detach().cpu() for k, v in m128.state_dict().items()}
    check("o_8192", ol, orc.fla_layer_forward(sdl, xl.cpu(), 2, 32, 64, norm_eps=1e-5), 1e-4)


def test_thin_dit_host_matches_cpu_composition():
    """The in-repo DiT host on the GPU vs the same host on the CPU with every attention module replaced by
    the oracle's restatement of MHLA4DiT.forward: checks the block-major adapter, the adaLN plumbing and the module in context."""
    from mhla_amd.hosts import DiT_MHLA
    torch.manual_seed(0)
    m = DiT_MHLA(input_size=16, patch_size=2, in_channels=4, hidden_size=128, depth=2, num_heads=2, num_classes=10,
                 class_dropout_prob=0.0, block_kwargs={"block_size": 16})
    with torch.no_grad():
        for prm in m.parameters():
            if prm.requires_grad and float(prm.abs().max()) == 0.0:
                prm.normal_(std=0.05)
    m.eval()
    x, t, y = torch.randn(3, 4, 16, 16), torch.tensor([1, 500, 999]), torch.tensor([0, 3, 9])
    # round trip of the token adapter
    tok = torch.arange(64)
    assert torch.equal(tok[m.to_block_major][m.to_raster], tok)
    assert sorted(m.state_dict())[:3] == ["blocks.0.adaLN_modulation.1.bias", "blocks.0.adaLN_modulation.1.weight", "blocks.0.attn.lepe.bias"]

    import copy
    ref = copy.deepcopy(m)
    for blk in ref.blocks:
        sd = {k: v.detach() for k, v in blk.attn.state_dict().items()}
        heads, bs, el = blk.attn.num_heads, blk.attn.block_size, blk.attn.embed_len
        blk.attn.forward = (lambda sd, heads, bs, el: lambda z: orc.dit_module_forward(
            sd, z.reshape(z.shape[0], el // bs, bs, z.shape[-1]), heads, bs, el).reshape(z.shape))(sd, heads, bs, el)
    with torch.no_grad():
        want = ref(x, t, y)
        got = m.to(DEV)(x.to(DEV), t.to(DEV), y.to(DEV))
    assert got.shape == (3, 8, 16, 16)
    check("dit host", got, want, 2e-4)
    # one training step runs (autograd through the HIP ops, clamp as in mhla_dit/train.py:308-310)
    m.train()
    out = m(x.to(DEV), t.to(DEV), y.to(DEV))
    out.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad)


def test_minimal_gpt_host_trains_a_step():
    """The minimal GPT host around the fla MHLA layer: finite loss and gradients for every parameter,
    causal (logits at position t do not change when later tokens change)."""
    from mhla_amd.hosts import GPT_MHLA
    torch.manual_seed(0)
    m = GPT_MHLA(vocab_size=97, hidden_size=128, num_layers=2, num_heads=2).to(DEV)
    ids = torch.randint(0, 97, (2, 200), device=DEV)
    loss = m(ids, labels=ids)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    with torch.no_grad():
        a = m(ids)
        ids2 = ids.clone()
        ids2[:, 150:] = (ids2[:, 150:] + 1) % 97
        b = m(ids2)
    check("causal prefix", a[:, :150], b[:, :150].cpu(), 1e-5)
    # parity: the same host on the CPU with every attention layer replaced by the oracle's restatement of the fla layer
    import copy
    ref = copy.deepcopy(m).cpu()
    for blk in ref.layers:
        sd = {k: v.detach() for k, v in blk.attn.state_dict().items()}
        lay = blk.attn
        blk.attn.forward = (lambda sd, lay: lambda z, **kw: (orc.fla_layer_forward(
            sd, z, lay.num_heads, lay.head_k_dim, lay.head_v_dim, norm_eps=lay.g_norm_swish_gate.eps), None, None))(sd, lay)
    with torch.no_grad():
        want = ref(ids.cpu())
    check("gpt host logits", a, want, 2e-4)


def test_wan_block_shell():
    """Thin Wan block host (model.py:1605-1766 shell around MHLA_Video_Uni): reference parameter names, inference (fused)
    and training paths agree, and with the residual gates at zero only the cross-attention branch remains."""
    from mhla_amd import modules
    from mhla_amd.hosts import WanAttentionBlock_MHLA
    torch.manual_seed(3)
    dim, heads, grid = 128, 2, (4, 6, 9)
    N = grid[0] * grid[1] * grid[2]
    blk = WanAttentionBlock_MHLA(dim=dim, ffn_dim=256, num_heads=heads, block_layout=(2, 2, 3), is_lepe=True).to(DEV).eval()
    keys = set(blk.state_dict())
    for k in ("modulation", "norm3.weight", "self_attn.block_attn.conv.weight", "self_attn.lepe.weight", "self_attn.g_norm.weight",
              "cross_attn.norm_q.weight", "cross_attn.o.bias", "ffn.0.weight", "ffn.2.bias"):
        assert k in keys, k
    B = 2
    x = torch.randn(B, N, dim, device=DEV)
    e = torch.randn(B, 6, dim, device=DEV) * 0.1
    ctx = torch.randn(B, 7, dim, device=DEV)
    gs = torch.tensor([list(grid)] * B, dtype=torch.long)
    sl = torch.tensor([N] * B)
    fr = modules.wan_freqs(dim // heads)
    with torch.no_grad():
        y_inf = blk(x, e, sl, gs, fr, ctx, torch.tensor([7, 5]))
    xg = x.clone().requires_grad_(True)
    y_tr = blk(xg, e, sl, gs, fr, ctx, torch.tensor([7, 5]))
    assert y_inf.shape == x.shape and torch.isfinite(y_inf).all()
    check("inference vs training path", y_inf, y_tr.detach().cpu(), 1e-4)
    y_tr.sum().backward()
    assert torch.isfinite(xg.grad).all() and blk.self_attn.block_attn.conv.weight.grad is not None
    e0 = e.clone()
    e0[:, 2] = -blk.modulation.detach()[0, 2]
    e0[:, 5] = -blk.modulation.detach()[0, 5]
    with torch.no_grad():
        y0 = blk(x, e0, sl, gs, fr, ctx)
        want = x + blk.cross_attn(blk.norm3(x), ctx)
    check("gates at zero", y0, want.cpu(), 1e-5)


@pytest.mark.parametrize("opts", [
    dict(num_kv_heads=2),                                   # grouped k / v heads (layers/mhla.py:290-292)
    dict(use_output_gate=False),                            # plain per-head RMSNorm (:357-358)
    dict(gate_fn="sigmoid"),                                # RMSNorm, then o * sigmoid(g) (:355-356)
    dict(feature_map="elu"),                                # elu + 1 (:130-134)
    dict(num_kv_heads=1, feature_map="identity", gate_fn="sigmoid"),
    dict(use_short_conv=True, conv_size=4),                 # q / k / v through the short convolutions (:258-279)
    dict(use_short_conv=True, conv_size=3, conv_bias=True, num_kv_heads=2),
])
def test_fla_layer_options_match_oracle_restatement(opts):
    """The fla layer's non-default constructor options on the GPU -- GQA, no output gate, a non-swish gate, other feature maps --
    forward, dx and every parameter gradient against autograd through the oracle's restatement of the layer."""
    from mhla_amd import modules
    torch.manual_seed(11)
    heads, hk, hv = 4, 32, 64
    m = modules.MHLA(mode="chunk", hidden_size=128, expand_k=1.0, expand_v=2.0, num_heads=heads, norm_eps=1e-6,
                     **{"feature_map": "relu", **opts})
    with torch.no_grad():
        (m.g_norm_swish_gate if hasattr(m, "g_norm_swish_gate") else m.g_norm).weight.uniform_(0.5, 1.5)
        m.mixing_matrix.copy_(torch.rand(32, 32).view(32, 32, 1, 1, 1, 1).clamp_(1e-5, 1))
    x = torch.randn(2, 300, 128)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    okw = dict(norm_eps=1e-6, num_kv_heads=opts.get("num_kv_heads"), feature_map=opts.get("feature_map", "relu"),
               use_output_gate=opts.get("use_output_gate", True), gate_fn=opts.get("gate_fn", "swish"),
               use_short_conv=opts.get("use_short_conv", False))
    if opts.get("use_short_conv"):
        x = x[:, :140]   # (the oracle's short convolution is a plain Python loop)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    want = orc.fla_layer_forward(sdr, xr, heads, hk, hv, **okw)
    dY = torch.randn(want.shape, generator=torch.Generator().manual_seed(5))
    (want * dY).sum().backward()
    m = m.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    o, _, _ = m(xd)
    check("o", o, want.detach(), 1e-4)
    (o * dY.to(DEV)).sum().backward()
    check("dx (layer)", xd.grad, xr.grad, 2e-4)
    for name, prm in m.named_parameters():
        check(f"grad {name}", prm.grad, sdr[name].grad, 2e-4, atol=1e-7)


def test_wan_block_host_matches_cpu_composition():
    """The thin Wan block host (hosts/wan.py) on the GPU against the same host on the CPU with its self-attention module
    replaced by the oracle's restatement of MHLA_Video_Uni (wan_module_forward): block output (inference and training path),
    dx, and the gradients of the block's parameters -- as the DiT and GPT hosts are checked."""
    import copy
    from mhla_amd import modules
    from mhla_amd.hosts import WanAttentionBlock_MHLA
    torch.manual_seed(5)
    dim, heads, grid, layout = 128, 2, (4, 6, 9), (2, 2, 3)
    N = grid[0] * grid[1] * grid[2]
    blk = WanAttentionBlock_MHLA(dim=dim, ffn_dim=256, num_heads=heads, block_layout=layout, is_gated=True, is_lepe=False,
                                 norm_output=True)
    with torch.no_grad():
        blk.self_attn.block_attn.conv.weight.copy_(torch.rand_like(blk.self_attn.block_attn.conv.weight))
        blk.self_attn.g_norm.weight.uniform_(0.5, 1.5)
    B = 2
    x = torch.randn(B, N, dim)
    e = torch.randn(B, 6, dim) * 0.1
    ctx = torch.randn(B, 7, dim)
    gs = torch.tensor([list(grid)] * B, dtype=torch.long)
    sl = torch.tensor([N] * B)
    clens = torch.tensor([7, 5])
    ref = copy.deepcopy(blk)
    sa = ref.self_attn
    fr_cpu = orc.wan_freqs(dim // heads)
    ref.self_attn.forward = lambda z, seq_lens, grid_sizes, freqs: orc.wan_module_forward(
        {k: v for k, v in sa.named_parameters()}, z, grid, fr_cpu, heads, layout=layout, eps=sa.eps if hasattr(sa, "eps") else 1e-6,
        normalize_out=True, is_gated=True)
    xr = x.clone().requires_grad_(True)
    want = ref(xr, e, sl, gs, None, ctx, clens)
    dY = torch.randn(want.shape, generator=torch.Generator().manual_seed(2))
    (want * dY).sum().backward()
    blk = blk.to(DEV)
    fr = modules.wan_freqs(dim // heads)
    args = (e.to(DEV), sl, gs, fr, ctx.to(DEV), clens)
    with torch.no_grad():
        y_inf = blk(x.to(DEV), *args)
    check("wan block (inference path)", y_inf, want.detach(), 2e-4)
    xd = x.to(DEV).requires_grad_(True)
    y = blk(xd, *args)
    check("wan block (training path)", y, want.detach(), 2e-4)
    (y * dY.to(DEV)).sum().backward()
    check("dx (wan block)", xd.grad, xr.grad, 2e-4)
    refp = dict(ref.named_parameters())
    for name, prm in blk.named_parameters():
        if prm.grad is None:
            assert refp[name].grad is None or float(refp[name].grad.abs().max()) == 0.0, name
            continue
        check(f"grad {name}", prm.grad, refp[name].grad, 2e-4, atol=1e-7)


def test_fla_layer_padded_decoding_offsets():
    """layers/mhla.py:305-309: with a padding mask AND a non-empty cache the rotary offset of every sequence is its own length
    so far (prepare_lens_from_mask(mask) - q_len), not the cache's scalar length.  A left-aligned mask over [past + new] tokens,
    a cache that reports `past` tokens; the oracle restatement gets the same per-sequence offsets."""
    from mhla_amd import modules
    torch.manual_seed(4)
    heads, hk, hv, q_len = 2, 32, 64, 96
    m = modules.MHLA(mode="chunk", hidden_size=64, expand_k=1.0, expand_v=2.0, num_heads=heads, feature_map="relu", norm_eps=1e-6,
                     layer_idx=0)
    with torch.no_grad():
        m.mixing_matrix.copy_(torch.rand(32, 32).view(32, 32, 1, 1, 1, 1).clamp_(1e-5, 1))
    x = torch.randn(3, q_len, 64)
    total = torch.tensor([q_len + 40, q_len + 7, q_len + 25])          # tokens each sequence has seen including the new ones
    mask = (torch.arange(int(total.max()))[None, :] < total[:, None]).long()    # [B, past + new]; the layer looks at the last q_len columns

    class Cache:
        def __len__(self):
            return 1

        def __getitem__(self, i):
            return {"recurrent_state": None, "conv_state": None}

        def get_seq_length(self, layer_idx=0):
            return 40

        def update(self, **kw):
            self.last = kw

    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    new_mask = mask[:, -q_len:]
    want = orc.fla_layer_forward(sd, x, heads, hk, hv, norm_eps=1e-6, attention_mask=new_mask,
                                 position_offsets=mask.sum(-1) - q_len)
    o, _, cache = m.to(DEV)(x.to(DEV), attention_mask=mask.to(DEV), past_key_values=Cache(), use_cache=True)
    check("o (padded decoding)", o, want, 1e-4)
    assert cache.last["offset"] == q_len
