"""Generate golden fixtures by running the REFERENCE's own files (build container only).

Run:  TORCHDYNAMO_DISABLE=1 python tests/golden/make_golden.py

Imports single files from /root/reference by path (never as packages; recipes
from SURVEY.md section 8(c)), runs them on seeded inputs and writes small
``.npz`` fixtures next to this script.  /root/reference does not exist on the
GPU box and nothing under ``tests/`` reads it at run time -- only the
committed ``.npz`` data travels.  No reference source text is stored.
"""
import ast
import importlib.util
import os
import sys
import types

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def np32(t):
    # always a copy: a float32 CPU tensor would otherwise share its memory with the array, and later backward passes that
    # accumulate into .grad in place would silently change fixtures already collected
    return t.detach().to(torch.float32).cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB, keys={sorted(arrs)}")


# ---------------------------------------------------------------------------
# 1. DiT / ViT block-mix module + op (mhla_dit/mhla/mhla.py, attention/mhla.py)
# ---------------------------------------------------------------------------
def gen_blockmix_2d(tag, path, cls_name, seed, B, heads, dim_head, block_size, embed_len,
                    transform, qk_norm, size_kw, perturb_w):
    mod = load_by_path(f"ref_{tag}", path)
    cls = getattr(mod, cls_name)
    torch.manual_seed(seed)
    dim = heads * dim_head
    m = cls(dim, heads=heads, dim_head=dim_head, dropout=0.0, qk_norm=qk_norm, transform=transform,
            qkv_bias=(tag.startswith("dit")), embed_len=embed_len, **{size_kw: block_size})
    m.eval()
    M = embed_len // block_size
    with torch.no_grad():
        # de-trivialise: random LePE / norm weights, and (optionally) a post-training-like W
        m.lepe.weight.normal_(0, 0.2)
        m.lepe.bias.normal_(0, 0.1)
        m.norm.weight.uniform_(0.5, 1.5)
        m.norm.bias.normal_(0, 0.1)
        if qk_norm:
            m.q_norm.weight.uniform_(0.5, 1.5)
            m.k_norm.weight.uniform_(0.5, 1.5)
        if perturb_w:
            m.piece_attn.conv.weight.copy_(torch.rand(M, M, 1, 1))
    x = torch.randn(B, M, block_size, dim)
    dO = torch.randn(B * heads, M, block_size, dim_head)

    cap = {}
    orig = m._process_qkv_impl

    def wrapped(q, k, v, B_, N_, H_, D_):
        q2, kT, v2 = orig(q, k, v, B_, N_, H_, D_)
        for t in (q2, kT, v2):
            t.retain_grad()
        cap["q"], cap["kT"], cap["v"] = q2, kT, v2
        return q2, kT, v2

    m._process_qkv_impl = wrapped
    m.lepe.register_forward_hook(lambda mod_, inp, out: cap.__setitem__("lepe_img", out))
    m.to_out.register_forward_pre_hook(lambda mod_, inp: cap.__setitem__("pre_out", inp[0]))
    y = m(x)
    # op output = (out + lepe) - lepe, both taken from the reference's own graph
    from einops import rearrange
    pl, bl = m.pieces_len, int(block_size ** 0.5)
    lepe = rearrange(cap["lepe_img"], "b d (h p1) (w p2) -> b (h w) (p1 p2) d", h=pl, w=pl, p1=bl, p2=bl)
    op_out = cap["pre_out"] - lepe                                 # [B, M, S, (h d)]
    op_out_bh = rearrange(op_out, "b n w (h d) -> (b h) n w d", h=heads)
    loss = (op_out_bh * dO).sum()
    loss.backward()

    def to_bnhd(t):  # [(b h), M, S, D] -> [B, N, H, D]
        return rearrange(t, "(b h) n w d -> b (n w) h d", b=B)

    sd = {k: np32(v) for k, v in m.state_dict().items()}
    # module-level backward (own generator: the op-level values above keep their random stream): d<y, dY>/dx and the
    # gradient of every parameter, through the reference's own autograd graph
    m._process_qkv_impl = orig
    dW_op = np32(m.piece_attn.conv.weight.grad.reshape(M, M))       # op-level gradient, before the grads are cleared
    for prm in m.parameters():
        prm.grad = None
    dY = torch.randn(y.shape, generator=torch.Generator().manual_seed(seed + 1000))
    xg = x.clone().requires_grad_(True)
    (m(xg) * dY).sum().backward()
    gsd = {k: np32(prm.grad) for k, prm in m.named_parameters() if prm.grad is not None}
    save(
        f"blockmix2d_{tag}",
        dY=np32(dY), dx_mod=np32(xg.grad), **{"gsd." + k: v for k, v in gsd.items()},
        meta=np.array([B, heads, dim_head, M, block_size, embed_len, int(qk_norm)], dtype=np.int64),
        x=np32(x), y=np32(y),
        q=np32(to_bnhd(cap["q"])), k=np32(to_bnhd(cap["kT"].transpose(-2, -1))), v=np32(to_bnhd(cap["v"])),
        W=np32(m.piece_attn.get_weight_matrix()),
        out=np32(to_bnhd(op_out_bh)), dout=np32(to_bnhd(dO)),
        dq=np32(to_bnhd(cap["q"].grad)), dk=np32(to_bnhd(cap["kT"].grad.transpose(-2, -1))),
        dv=np32(to_bnhd(cap["v"].grad)),
        dW=dW_op,
        **{"sd." + k: v for k, v in sd.items()},
    )


def gen_weight_inits():
    dit = load_by_path("ref_dit_w", f"{REF}/mhla_dit/mhla/mhla.py")
    arrs = {}
    for tr in ("linear", "cos", "exp", "gaussian", "local"):
        for side, group in ((16, 16), (21, 49), (16, 4)):
            c = dit.BlockDistanceConv(num_patches_per_side=side, patch_group_size=group, transform=tr)
            arrs[f"w2d_{tr}_{side}_{group}"] = np32(c.get_weight_matrix())
    wan = load_wan_utils()
    for tr in ("linear", "cos", "exp", "gaussian", "local"):
        for layout in ((3, 5, 10), (2, 3, 4), (1, 4, 4)):
            c = wan.BlockDistanceConv3D(blocks_layout=layout, transform=tr)
            arrs[f"w3d_{tr}_{layout[0]}_{layout[1]}_{layout[2]}"] = np32(c.get_weight_matrix())
    save("weight_init", **arrs)


def gen_blockmix_bf16(tag, path, cls_name, seed, B, heads, dim_head, block_size, embed_len):
    """bf16 fixture of the block-mixing operator (verdict r4 item 1): the reference module's operator evaluated IN FP32 ON
    BF16-ROUNDED q, k, v, dO -- what an implementation that takes bf16 tensors and keeps fp32-grade intermediates must
    reproduce up to the final rounding -- and, beside it, the same module run literally in bf16 (`module.bfloat16()`), whose
    operator output shows what the reference's own bf16 arithmetic loses."""
    from einops import rearrange
    mod = load_by_path(f"ref_{tag}", path)
    cls = getattr(mod, cls_name)
    torch.manual_seed(seed)
    dim = heads * dim_head
    m = cls(dim, heads=heads, dim_head=dim_head, dropout=0.0, qk_norm=False, transform="linear", qkv_bias=True,
            embed_len=embed_len, block_size=block_size)
    m.eval()
    M = embed_len // block_size
    with torch.no_grad():
        m.lepe.weight.normal_(0, 0.2)
        m.lepe.bias.normal_(0, 0.1)
        m.piece_attn.conv.weight.copy_(torch.rand(M, M, 1, 1))
    x = torch.randn(B, M, block_size, dim)
    dO = torch.randn(B * heads, M, block_size, dim_head).bfloat16().float()
    rnd = lambda t: t.detach().bfloat16().float().requires_grad_(True)
    cap = {}
    orig = m._process_qkv_impl

    def wrapped(q, k, v, B_, N_, H_, D_):
        q2, kT, v2 = orig(q, k, v, B_, N_, H_, D_)
        cap["q"], cap["kT"], cap["v"] = rnd(q2), rnd(kT), rnd(v2)       # leaves holding bf16 values, everything after them in fp32
        return cap["q"], cap["kT"], cap["v"]

    m._process_qkv_impl = wrapped
    m.lepe.register_forward_hook(lambda mod_, inp, out: cap.__setitem__("lepe_img", out))
    m.to_out.register_forward_pre_hook(lambda mod_, inp: cap.__setitem__("pre_out", inp[0]))
    m(x)
    pl, bl = m.pieces_len, int(block_size ** 0.5)
    lepe = rearrange(cap["lepe_img"], "b d (h p1) (w p2) -> b (h w) (p1 p2) d", h=pl, w=pl, p1=bl, p2=bl)
    op_out_bh = rearrange(cap["pre_out"] - lepe, "b n w (h d) -> (b h) n w d", h=heads)
    (op_out_bh * dO).sum().backward()
    to_bnhd = lambda t: rearrange(t, "(b h) n w d -> b (n w) h d", b=B)
    fx = dict(q=np32(to_bnhd(cap["q"])), k=np32(to_bnhd(cap["kT"].transpose(-2, -1))), v=np32(to_bnhd(cap["v"])),
              W=np32(m.piece_attn.get_weight_matrix()), out=np32(to_bnhd(op_out_bh)), dout=np32(to_bnhd(dO)),
              dq=np32(to_bnhd(cap["q"].grad)), dk=np32(to_bnhd(cap["kT"].grad.transpose(-2, -1))), dv=np32(to_bnhd(cap["v"].grad)),
              dW=np32(m.piece_attn.conv.weight.grad.reshape(M, M)))
    # the literal bf16 run of the same module on the same (bf16-rounded) operator inputs
    mb = cls(dim, heads=heads, dim_head=dim_head, dropout=0.0, qk_norm=False, transform="linear", qkv_bias=True,
             embed_len=embed_len, block_size=block_size)
    mb.load_state_dict(m.state_dict())
    mb = mb.eval().bfloat16()
    capb = {}
    q0, k0, v0 = cap["q"].detach().bfloat16(), cap["kT"].detach().bfloat16(), cap["v"].detach().bfloat16()
    mb._process_qkv_impl = lambda q, k, v, B_, N_, H_, D_: (q0, k0, v0)
    mb.lepe.register_forward_hook(lambda mod_, inp, out: capb.__setitem__("lepe_img", out))
    mb.to_out.register_forward_pre_hook(lambda mod_, inp: capb.__setitem__("pre_out", inp[0]))
    with torch.no_grad():
        mb(x.bfloat16())
    lepeb = rearrange(capb["lepe_img"].float(), "b d (h p1) (w p2) -> b (h w) (p1 p2) d", h=pl, w=pl, p1=bl, p2=bl)
    outb = rearrange(capb["pre_out"].float() - lepeb, "b n w (h d) -> (b h) n w d", h=heads)
    fx["out_reference_module_in_bf16"] = np32(to_bnhd(outb))
    err = (outb - op_out_bh.detach()).abs().max() / op_out_bh.detach().abs().max()
    print(f"   the reference module run in bf16 is {err.item():.2e} of the maximum away from its fp32 operator output")
    save(f"blockmix2d_{tag}", meta=np.array([B, heads, dim_head, M, block_size, embed_len, 0], dtype=np.int64), **fx)


# ---------------------------------------------------------------------------
# 2. Wan MHLA_Video_Uni (wan/mhla_utils.py) -- stub module for its lazy import
# ---------------------------------------------------------------------------
_WAN = None


def load_wan_utils():
    global _WAN
    if _WAN is not None:
        return _WAN
    src_path = f"{REF}/mhla_videogen/diffusion/model/wan/model.py"
    tree = ast.parse(open(src_path).read())
    wanted = {"WanRMSNorm", "rope_params"}
    stub = types.ModuleType("diffusion.model.wan.model")
    ns = stub.__dict__
    exec("import torch\nimport torch.nn as nn\nfrom torch.cuda import amp\n", ns)
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in wanted:
            code = compile(ast.Module(body=[node], type_ignores=[]), src_path, "exec")
            exec(code, ns)
    for pkg in ("diffusion", "diffusion.model", "diffusion.model.wan"):
        sys.modules.setdefault(pkg, types.ModuleType(pkg))
    sys.modules["diffusion.model.wan.model"] = stub
    _WAN = load_by_path("ref_wan_utils", f"{REF}/mhla_videogen/diffusion/model/wan/mhla_utils.py")
    _WAN._stub = stub
    return _WAN


def x_grad_of_sum(m, x, seq_lens, grid_sizes, freqs):
    """d(sum of the module output)/dx: pins the module-level backward (LePE branch included)."""
    xg = x.clone().requires_grad_(True)
    m(xg, seq_lens, grid_sizes, freqs).sum().backward()
    return xg.grad


def gen_wan(tag, seed, B, heads, dim_head, layout, grid, normalize_out, is_gated, is_lepe=False):
    wan = load_wan_utils()
    stub = wan._stub
    torch.manual_seed(seed)
    dim = heads * dim_head
    m = wan.MHLA_Video_Uni(dim, num_heads=heads, block_layout=layout, normalize_out=normalize_out,
                           is_gated=is_gated, is_lepe=is_lepe)
    m.eval()
    M = layout[0] * layout[1] * layout[2]
    with torch.no_grad():
        m.norm_q.weight.uniform_(0.5, 1.5)
        m.norm_k.weight.uniform_(0.5, 1.5)
        m.g_norm.weight.uniform_(0.5, 1.5)
    d = dim_head
    freqs = torch.cat([stub.rope_params(1024, d - 4 * (d // 6)), stub.rope_params(1024, 2 * (d // 6)),
                       stub.rope_params(1024, 2 * (d // 6))], dim=1)      # wan/model.py:1932-1936
    N = grid[0] * grid[1] * grid[2]
    x = torch.randn(B, N, dim)
    grid_sizes = torch.tensor([list(grid)] * B, dtype=torch.long)
    seq_lens = torch.tensor([N] * B, dtype=torch.long)
    dO = torch.randn(B, N, heads, dim_head)

    cap = {"rope": []}
    orig_rope = wan.rope_apply

    def rope_wrapped(t, gs, fr):
        cap.setdefault("rope_in", []).append(t)
        t.retain_grad()
        r = orig_rope(t, gs, fr)
        r.retain_grad()
        cap["rope"].append(r)
        return r

    wan.rope_apply = rope_wrapped
    orig_proc = m._process_qkv_impl

    def proc_wrapped(q, k, v, *a):
        q2, k2, v2 = orig_proc(q, k, v, *a)
        v2 = v2 * 1.0           # fresh node so the op-only gradient of v can be retained
        v2.retain_grad()
        cap["v"] = v2
        return q2, k2, v2

    m._process_qkv_impl = proc_wrapped
    m.g_norm.register_forward_pre_hook(lambda mod_, inp: cap.__setitem__("op_out", inp[0]))
    y = m(x, seq_lens, grid_sizes, freqs)
    wan.rope_apply = orig_rope
    op_out = cap["op_out"]                                               # [B, N, H, D] raster order
    (op_out.float() * dO).sum().backward()
    q_in, k_in = cap["rope_in"]
    q_rope, k_rope = cap["rope"]
    v4 = cap["v"].reshape(B, N, heads, dim_head)
    sd = {k: np32(v) for k, v in m.state_dict().items()}
    m._process_qkv_impl = orig_proc
    dW_op = np32(m.block_attn.conv.weight.grad.reshape(M, M))        # op-level gradient, before the grads are cleared
    for prm in m.parameters():
        prm.grad = None
    dY = torch.randn(y.shape, generator=torch.Generator().manual_seed(seed + 1000))
    xg = x.clone().requires_grad_(True)
    (m(xg, seq_lens, grid_sizes, freqs) * dY).sum().backward()
    gsd = {k: np32(prm.grad) for k, prm in m.named_parameters() if prm.grad is not None}
    save(
        f"wan_{tag}",
        dY=np32(dY), dx_mod=np32(xg.grad), **{"gsd." + k: v for k, v in gsd.items()},
        meta=np.array([B, heads, dim_head, M, N // M, *layout, *grid, int(normalize_out), int(is_gated)], dtype=np.int64),
        x=np32(x), y=np32(y),
        q=np32(q_in), k=np32(k_in), v=np32(v4), q_rope=np32(q_rope), k_rope=np32(k_rope),
        W=np32(m.block_attn.get_weight_matrix()),
        out=np32(op_out), dout=np32(dO),
        dq=np32(q_in.grad), dk=np32(k_in.grad), dq_rope=np32(q_rope.grad), dk_rope=np32(k_rope.grad),
        dv=np32(cap["v"].grad.reshape(B, N, heads, dim_head)),
        dW=dW_op,
        freqs_re=freqs.real.numpy()[:64], freqs_im=freqs.imag.numpy()[:64],
        is_lepe=np.array([int(is_lepe)], dtype=np.int64),
        dx=np32(x_grad_of_sum(m, x, seq_lens, grid_sizes, freqs)),
        **{"sd." + k: v for k, v in sd.items()},
    )


# ---------------------------------------------------------------------------
# 2b. the five older Wan MHLA classes (wan/model.py:428-1389): each class is AST-extracted from model.py (which cannot be
#     imported: diffusers / timm / mmcv / flash-attn) together with WanRMSNorm, rope_params and rope_apply, and executed in
#     a stub namespace that gets BlockDistanceConv3D from the reference's own mhla_utils.py
# ---------------------------------------------------------------------------
def load_wan_variant(cls_name):
    wan = load_wan_utils()
    src_path = f"{REF}/mhla_videogen/diffusion/model/wan/model.py"
    tree = ast.parse(open(src_path).read())
    ns = {}
    exec("import math\nimport torch\nimport torch.nn as nn\nimport torch.nn.functional as F\nfrom torch.cuda import amp\n"
         "from einops import rearrange\n", ns)
    ns["BlockDistanceConv3D"] = wan.BlockDistanceConv3D
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in {"WanRMSNorm", "rope_params", "rope_apply", cls_name}:
            exec(compile(ast.Module(body=[node], type_ignores=[]), src_path, "exec"), ns)
    return ns


def gen_wan_variant(tag, cls_name, seed, B, heads, dim_head, layout, grid, normalize_out, out_rmsnorm):
    ns = load_wan_variant(cls_name)
    torch.manual_seed(seed)
    dim = heads * dim_head
    m = ns[cls_name](dim, num_heads=heads, block_layout=layout, normalize_out=normalize_out, out_rmsnorm=out_rmsnorm)
    m.eval()
    M = layout[0] * layout[1] * layout[2]
    with torch.no_grad():       # de-trivialise the norm weights and the mixing weights
        for name, prm in m.named_parameters():
            if name.endswith("norm.weight") or name.startswith("norm_") or name == "g_norm.weight" or name == "out_rmsnorm.weight":
                prm.uniform_(0.5, 1.5)
        m.block_attn.conv.weight.mul_(torch.rand(M, M, 1, 1) + 0.5)
    d = dim_head
    freqs = torch.cat([ns["rope_params"](1024, d - 4 * (d // 6)), ns["rope_params"](1024, 2 * (d // 6)),
                       ns["rope_params"](1024, 2 * (d // 6))], dim=1)
    N = grid[0] * grid[1] * grid[2]
    x = torch.randn(B, N, dim)
    grid_sizes = torch.tensor([list(grid)] * B, dtype=torch.long)
    seq_lens = torch.tensor([N] * B, dtype=torch.long)
    xg = x.clone().requires_grad_(True)
    y = m(xg, seq_lens, grid_sizes, freqs)
    dY = torch.randn(y.shape, generator=torch.Generator().manual_seed(seed + 1000))
    (y * dY).sum().backward()
    save(
        f"wanv_{tag}",
        meta=np.array([B, heads, dim_head, *layout, *grid, int(normalize_out), int(out_rmsnorm)], dtype=np.int64),
        x=np32(x), y=np32(y), dY=np32(dY), dx_mod=np32(xg.grad),
        **{"sd." + k: np32(v) for k, v in m.state_dict().items()},
        **{"gsd." + k: np32(prm.grad) for k, prm in m.named_parameters() if prm.grad is not None},
    )


# ---------------------------------------------------------------------------
# 3. fla causal op (fla/ops/mhla/naive.py)
# ---------------------------------------------------------------------------
def gen_causal(tag, seed, B, T, H, K, V, L, dtype=torch.float32, random_mix=False):
    naive = load_by_path("ref_fla_naive", f"{REF}/mhla_nlp/fla/ops/mhla/naive.py")
    torch.manual_seed(seed)
    q = torch.relu(torch.randn(B, T, H, K)).to(dtype).requires_grad_(True)
    k = torch.relu(torch.randn(B, T, H, K)).to(dtype).requires_grad_(True)
    # give q, k signs like post-rotary activations have
    with torch.no_grad():
        q.mul_(torch.sign(torch.randn_like(q)))
        k.mul_(torch.sign(torch.randn_like(k)))
    v = torch.randn(B, T, H, V).to(dtype).requires_grad_(True)
    lower = torch.tril(torch.ones(L, L)) / (torch.arange(L, dtype=torch.float32).unsqueeze(1) + 1.0)
    if random_mix:
        lower = torch.tril(torch.rand(L, L).clamp(1e-5, 1))
    mix = lower.view(L, L, 1, 1, 1, 1).clone().requires_grad_(True)
    dO = torch.randn(B, T, H, V)
    o = naive.naive_chunk_simple_mhla_fixed(q=q, k=k, v=v, mixing_matrix=mix, output_final_state=False)
    (o.float() * dO).sum().backward()
    arrs = dict(meta=np.array([B, T, H, K, V, L], dtype=np.int64), q=np32(q), k=np32(k), v=np32(v),
                mix=np32(mix.reshape(L, L)), out=np32(o), dout=np32(dO), dq=np32(q.grad), dk=np32(k.grad),
                dv=np32(v.grad), dmix=np32(mix.grad.reshape(L, L)))
    if T <= 64:
        # first-chunk cross-check: the token-recurrent form agrees with the chunk form only here
        o_rec, _ = naive.naive_recurrent_mhla(q=q.detach(), k=k.detach(), v=v.detach(), mixing_matrix=mix.detach())
        arrs["out_recurrent"] = np32(o_rec)
    save(f"causal_{tag}", **arrs)


# ---------------------------------------------------------------------------
# 4. fla layer neighbours: rotary / rms_norm twins (torch refs inside Triton files)
# ---------------------------------------------------------------------------
def gen_fla_neighbours():
    # import scaffolding only: helper names the Triton files pull from the fla package
    fu = types.ModuleType("fla.utils")
    fu.autotune_cache_kwargs = {}
    fu.get_multiprocessor_count = lambda *a, **k: 1
    fu.input_guard = lambda f: f
    fu.is_amd = True
    fou = types.ModuleType("fla.ops.utils")
    fou.prepare_chunk_indices = lambda *a, **k: None
    for name in ("fla", "fla.ops"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["fla.utils"] = fu
    sys.modules["fla.ops.utils"] = fou
    rot = load_by_path("ref_fla_rotary", f"{REF}/mhla_nlp/fla/modules/rotary.py")
    ln = load_by_path("ref_fla_layernorm", f"{REF}/mhla_nlp/fla/modules/layernorm.py")
    torch.manual_seed(77)
    B, T, H, D = 2, 96, 2, 32
    x = torch.randn(B, T, H, D)
    emb = rot.RotaryEmbedding(dim=D)
    emb._update_cos_sin_cache(T, device=x.device, dtype=x.dtype)
    cos, sin = emb._cos_cached[:T], emb._sin_cached[:T]
    y = rot.rotary_embedding_ref(x, cos, sin, interleaved=False)
    xb = x.to(torch.bfloat16)
    embb = rot.RotaryEmbedding(dim=D)
    embb._update_cos_sin_cache(T, device=x.device, dtype=torch.bfloat16)
    yb = rot.rotary_embedding_ref(xb, embb._cos_cached[:T], embb._sin_cached[:T], interleaved=False)
    # rms_norm twin (layernorm.py:59-80) + the gate formula of fused_norm_gate.py:93-95 applied with torch ops
    o = torch.randn(B, T, H, D)
    g = torch.randn(B, T, H, D)
    w = torch.rand(D) + 0.5
    normed = ln.rms_norm_ref(o, w, None, eps=1e-5, upcast=True)
    gated = normed * g * torch.sigmoid(g)
    save("fla_neighbours", x=np32(x), rot=np32(y), x_bf16=np32(xb), rot_bf16=np32(yb),
         o=np32(o), g=np32(g), w=np32(w), normed=np32(normed), gated=np32(gated))


if __name__ == "__main__":
    gen_weight_inits()
    gen_blockmix_2d("dit_a", f"{REF}/mhla_dit/mhla/mhla.py", "MHLA4DiT", seed=11, B=2, heads=2, dim_head=32,
                    block_size=16, embed_len=256, transform="linear", qk_norm=False, size_kw="block_size",
                    perturb_w=False)
    gen_blockmix_2d("dit_b", f"{REF}/mhla_dit/mhla/mhla.py", "MHLA4DiT", seed=12, B=1, heads=1, dim_head=72,
                    block_size=49, embed_len=441, transform="exp", qk_norm=False, size_kw="block_size",
                    perturb_w=True)
    gen_blockmix_2d("vit_a", f"{REF}/mhla_image_classification/models/modules/attention/mhla.py",
                    "MHLA_Normed_Torch", seed=13, B=1, heads=2, dim_head=64, block_size=16, embed_len=256,
                    transform="cos", qk_norm=True, size_kw="window_size", perturb_w=False)
    gen_blockmix_bf16("dit_c", f"{REF}/mhla_dit/mhla/mhla.py", "MHLA4DiT", seed=13, B=1, heads=3, dim_head=64, block_size=16, embed_len=256)
    gen_wan("a", seed=21, B=1, heads=2, dim_head=128, layout=(2, 3, 4), grid=(4, 6, 8), normalize_out=True,
            is_gated=False)
    gen_wan("b", seed=22, B=1, heads=2, dim_head=32, layout=(3, 5, 10), grid=(3, 10, 20), normalize_out=False,
            is_gated=True)
    gen_wan("c", seed=23, B=2, heads=2, dim_head=32, layout=(2, 2, 3), grid=(4, 6, 9), normalize_out=False,
            is_gated=True, is_lepe=True)
    gen_wan_variant("mhla", "MHLA_Video", seed=41, B=1, heads=2, dim_head=32, layout=(2, 2, 3), grid=(4, 6, 9),
                    normalize_out=True, out_rmsnorm=True)
    gen_wan_variant("mhla_nope", "MHLA_Video_Nope", seed=42, B=1, heads=2, dim_head=32, layout=(2, 3, 2), grid=(4, 6, 8),
                    normalize_out=False, out_rmsnorm=False)
    gen_wan_variant("gated_mhla", "Gated_MHLA_Video", seed=43, B=2, heads=2, dim_head=32, layout=(1, 2, 3), grid=(3, 4, 9),
                    normalize_out=True, out_rmsnorm=False)
    gen_wan_variant("mhla_lepe", "MHLA_Video_LePE", seed=44, B=1, heads=2, dim_head=32, layout=(2, 2, 3), grid=(4, 6, 9),
                    normalize_out=True, out_rmsnorm=True)
    gen_wan_variant("gated_mhla_lepe", "Gated_MHLA_Video_LePE", seed=45, B=1, heads=2, dim_head=32, layout=(2, 2, 3), grid=(4, 6, 9),
                    normalize_out=False, out_rmsnorm=False)
    gen_causal("a", seed=31, B=2, T=256, H=2, K=16, V=24, L=32)
    gen_causal("b", seed=32, B=1, T=200, H=2, K=32, V=16, L=32, random_mix=True)
    gen_causal("c", seed=33, B=2, T=50, H=1, K=16, V=16, L=32)
    gen_causal("d", seed=34, B=1, T=320, H=1, K=64, V=128, L=8, dtype=torch.bfloat16, random_mix=True)
    gen_fla_neighbours()
