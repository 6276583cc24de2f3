"""CPU: pin the oracle (oracle/mhla_oracle.py) against fixtures generated from the
reference's own files (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import mhla_oracle as orc

TOL = 2e-5   # fp32 oracle vs fp32 reference: only summation-order differences


@pytest.mark.parametrize("key", [
    ("w2d", "linear", (4, 4), "16_16"), ("w2d", "cos", (3, 3), "21_49"), ("w2d", "exp", (8, 8), "16_4"),
    ("w2d", "gaussian", (4, 4), "16_16"), ("w2d", "local", (8, 8), "16_4"),
    ("w3d", "linear", (3, 5, 10), "3_5_10"), ("w3d", "cos", (2, 3, 4), "2_3_4"), ("w3d", "exp", (1, 4, 4), "1_4_4"),
    ("w3d", "gaussian", (3, 5, 10), "3_5_10"), ("w3d", "local", (2, 3, 4), "2_3_4"),
])
def test_weight_init(key):
    kind, tr, layout, suffix = key
    g = load_golden("weight_init")[f"{kind}_{tr}_{suffix}"]
    w = orc.block_distance_weights(layout, tr)
    assert w.shape == g.shape
    assert rel_err(w, g) < 1e-6
    if tr != "gaussian":
        assert torch.allclose(w.sum(0), torch.ones(w.shape[0]), atol=1e-5)   # column-normalised


@pytest.mark.parametrize("tag", ["dit_a", "dit_b", "vit_a"])
def test_blockmix2d_op_and_grads(tag):
    g = load_golden("blockmix2d_" + tag)
    out = orc.blockmix_fwd(g["q"], g["k"], g["v"], g["W"], eps=1e-6)
    assert rel_err(out, g["out"]) < TOL
    grads = orc.blockmix_bwd(g["q"], g["k"], g["v"], g["W"], g["dout"], eps=1e-6)
    for name in ("dq", "dk", "dv", "dW"):
        assert rel_err(grads[name], g[name]) < 5e-5, name



def test_blockmix2d_bf16_fixture_fp32_evaluation_of_bf16_inputs():
    """`dit_c`: the reference module's operator evaluated in fp32 on bf16-rounded q, k, v, dO (make_golden.gen_blockmix_bf16).  The oracle
    reproduces it like the fp32 fixtures; the reference module run literally in bf16 is two orders of magnitude further away."""
    g = load_golden("blockmix2d_dit_c")
    q, k, v, W, do = (g[n] for n in ("q", "k", "v", "W", "dout"))
    for t in (q, k, v, do):
        assert torch.equal(t, t.bfloat16().float())   # the fixture's operator inputs ARE bf16 values
    out = orc.blockmix_fwd(q, k, v, W, eps=1e-6)
    assert rel_err(out, g["out"]) < 1e-5
    gr = orc.blockmix_bwd(q, k, v, W, do, eps=1e-6)
    for n in ("dq", "dk", "dv", "dW"):
        assert rel_err(gr[n], g[n]) < 2e-5, n
    assert rel_err(g["out_reference_module_in_bf16"], g["out"]) > 3e-3

@pytest.mark.parametrize("tag", ["dit_a", "dit_b", "vit_a"])
def test_blockmix2d_closed_form_matches_autograd_fp64(tag):
    g = load_golden("blockmix2d_" + tag)
    q, k, v, W = (g[n].double().requires_grad_(True) for n in ("q", "k", "v", "W"))
    out = orc.blockmix_fwd(q, k, v, W, eps=1e-6)
    (out * g["dout"].double()).sum().backward()
    grads = orc.blockmix_bwd(q.detach(), k.detach(), v.detach(), W.detach(), g["dout"].double(), eps=1e-6)
    for name, t in (("dq", q), ("dk", k), ("dv", v), ("dW", W)):
        assert rel_err(grads[name], t.grad) < 1e-12, name


@pytest.mark.parametrize("tag,heads,block,embed,qkn,lk", [
    ("dit_a", 2, 16, 256, False, 3), ("dit_b", 1, 49, 441, False, 3), ("vit_a", 2, 16, 256, True, 5)])
def test_blockmix2d_module(tag, heads, block, embed, qkn, lk):
    g = load_golden("blockmix2d_" + tag)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    y = orc.dit_module_forward(sd, g["x"], heads, block, embed, qk_norm=qkn, lepe_k=lk)
    assert rel_err(y, g["y"]) < TOL


@pytest.mark.parametrize("tag", ["a", "b"])
def test_wan_op_module_and_grads(tag):
    g = load_golden("wan_" + tag)
    B, H, D, M, S, fb, hb, wb, F_, H_, W_, normalize, gated = [int(x) for x in g["meta"]]
    grid, layout = (F_, H_, W_), (fb, hb, wb)
    freqs = torch.complex(g["freqs_re"], g["freqs_im"])
    # prologue pieces
    assert rel_err(orc.wan_rope_apply(g["q"], grid, freqs), g["q_rope"]) < 1e-6
    assert rel_err(torch.view_as_real(orc.wan_freqs(D)[:64]), torch.view_as_real(freqs)) < 1e-12
    idx = orc.block_index_3d(grid, layout)
    args = [g[n][:, idx] for n in ("q_rope", "k_rope", "v")]
    den = dict(q_den=g["q"][:, idx], k_den=g["k"][:, idx])
    out_b = orc.blockmix_fwd(*args, g["W"], 1e-6, normalize=bool(normalize), **den)
    out = torch.empty_like(out_b)
    out[:, idx] = out_b
    assert rel_err(out, g["out"]) < TOL
    grads = orc.blockmix_bwd(*args, g["W"], g["dout"][:, idx], 1e-6, normalize=bool(normalize), **den)

    def unblock(t):
        r = torch.empty_like(t)
        r[:, idx] = t
        return r
    assert rel_err(unblock(grads["dq"]), g["dq_rope"]) < 5e-5
    assert rel_err(unblock(grads["dk"]), g["dk_rope"]) < 5e-5
    assert rel_err(unblock(grads["dv"]), g["dv"]) < 5e-5
    assert rel_err(grads["dW"], g["dW"]) < 5e-5
    # total gradient wrt the un-roped q: den part + rope^T(num part)
    q = g["q"].clone().requires_grad_(True)
    (orc.wan_rope_apply(q, grid, freqs) * g["dq_rope"]).sum().backward()
    dq_total = q.grad + (unblock(grads["dq_den"]) if normalize else 0)
    assert rel_err(dq_total, g["dq"]) < 5e-5
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    y = orc.wan_module_forward(sd, g["x"], grid, orc.wan_freqs(D), H, layout, 1e-6, bool(normalize), bool(gated))
    assert rel_err(y, g["y"]) < TOL


@pytest.mark.parametrize("kind", ["mhla", "mhla_nope", "gated_mhla", "mhla_lepe", "gated_mhla_lepe"])
def test_older_wan_variants_module_and_grads(kind):
    """The oracle's restatement of the five older Wan classes vs the reference classes' own outputs and autograd gradients
    (fixtures made by AST-extracting each class from wan/model.py: tests/golden/make_golden.py section 2b)."""
    g = load_golden("wanv_" + kind)
    B, H, D, fb, hb, wb, F_, H_, W_, normalize, _ = [int(x) for x in g["meta"]]
    sd = {k[3:]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith("sd.")}
    x = g["x"].clone().requires_grad_(True)
    y = orc.wan_variant_forward(kind, sd, x, (F_, H_, W_), orc.wan_freqs(D), H, (fb, hb, wb), 1e-6, bool(normalize))
    assert rel_err(y.detach(), g["y"]) < TOL
    (y * g["dY"]).sum().backward()
    assert rel_err(x.grad, g["dx_mod"]) < 5e-5
    for name, ref in ((k[4:], v) for k, v in g.items() if k.startswith("gsd.")):
        assert rel_err(sd[name].grad.reshape(ref.shape), ref) < 5e-5, name


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_causal_op_and_grads(tag):
    g = load_golden("causal_" + tag)
    bf16 = tag == "d"
    q, k, v = g["q"], g["k"], g["v"]
    if bf16:
        q, k, v = q.bfloat16(), k.bfloat16(), v.bfloat16()
    out = orc.causal_fwd(q, k, v, g["mix"])
    assert out.dtype == q.dtype
    assert rel_err(out.float(), g["out"]) < (1e-2 if bf16 else TOL)
    grads = orc.causal_bwd(q, k, v, g["mix"], g["dout"])
    for name in ("dq", "dk", "dv", "dmix"):
        assert rel_err(grads[name].float(), g[name]) < (1e-2 if bf16 else 5e-5), name
    if "out_recurrent" in g:   # T <= 64: token-recurrent form == chunk form (first chunk only; naive.py:124-136)
        assert rel_err(out, g["out_recurrent"]) < 1e-4


def test_causal_closed_form_matches_autograd_fp64():
    g = load_golden("causal_b")
    q, k, v, mix = (g[n].double().requires_grad_(True) for n in ("q", "k", "v", "mix"))
    # causal_fwd computes in fp32 like the reference; use fp32 autograd and compare closed form at fp32 accuracy
    qf, kf, vf, mf = (t.detach().float().requires_grad_(True) for t in (q, k, v, mix))
    out = orc.causal_fwd(qf, kf, vf, mf)
    (out * g["dout"]).sum().backward()
    grads = orc.causal_bwd(qf.detach(), kf.detach(), vf.detach(), mf.detach(), g["dout"])
    for name, t in (("dq", qf), ("dk", kf), ("dv", vf), ("dmix", mf)):
        assert rel_err(grads[name], t.grad) < 2e-5, name


def test_fla_neighbours():
    g = load_golden("fla_neighbours")
    assert rel_err(orc.neox_rotary(g["x"]), g["rot"]) < 1e-6
    yb = orc.neox_rotary(g["x_bf16"].bfloat16())
    assert yb.dtype == torch.bfloat16
    assert rel_err(yb.float(), g["rot_bf16"]) < 1e-2
    gated = orc.rms_norm_swish_gate(g["o"], g["g"], g["w"], 1e-5)
    assert rel_err(gated, g["gated"]) < 1e-6


def test_block_index_maps_are_permutations():
    i2 = orc.block_index_2d(4, 4)
    assert sorted(i2.tolist()) == list(range(256))
    # block 0 = top-left 4x4 patch of the 16x16 raster
    assert i2[:16].tolist() == [r * 16 + c for r in range(4) for c in range(4)]
    i3 = orc.block_index_3d((3, 10, 20), (3, 5, 10))
    assert sorted(i3.tolist()) == list(range(600))
