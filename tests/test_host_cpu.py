"""CPU (-m "not gpu"): host logic, C-ABI surface, drop-in module contracts, and the world_size-2 gloo path."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

from conftest import ROOT, load_golden


def test_library_builds_and_exports_every_declared_symbol():
    from mhla_amd import build as b, _lib
    b.build()
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "mhla_hip.h")).read()
    declared = set(re.findall(r"\b(mhla_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mhla_hip.h but not exported"
    assert lib.mhla_abi_version() == _lib.ABI_VERSION == 9
    assert b"no-packed-fp32" in lib.mhla_build_flags()
    # workspace sizing is pure host arithmetic: callable without a GPU
    fwd = lib.mhla_blockmix_fwd_ws_bytes(8, 16, 64, 64, 64, 0, 0, 0)
    bwd = lib.mhla_blockmix_bwd_ws_bytes(8, 16, 64, 64, 64, 0, 0, 0)
    # The workspace follows the summary format of the call (capi_common.hpp bm_sumfmt).  fp32 tensors, D = 64: fp32 words (4 bytes per
    # summary element + 1152 bytes of row padding).  bf16 tensors: 2 bytes by default (fp16 payload + the row's multiplier), 3 bytes
    # (24-bit floats) with MHLA_FLAG_FP32_GRADE_SUMMARIES, 2 bytes (single bf16, reduced precision) with MHLA_FLAG_BF16_SUMMARIES;
    # the flags mean nothing for fp32 tensors.
    rows, small = 8 * 16 * 64, lambda M, S, D: 4 * (2 * 8 * 16 * M * S + 8 * 16 * M * D)     # summary rows per set; z, 1/n, ksum
    assert fwd == 2 * rows * (4096 + 288) * 4 + small(64, 64, 64)
    h16 = lib.mhla_blockmix_fwd_ws_bytes(8, 16, 64, 64, 64, 1, 0, 0)
    assert h16 == 2 * rows * (2048 + 288) * 4 + small(64, 64, 64)
    assert lib.mhla_blockmix_fwd_ws_bytes(8, 16, 64, 64, 64, 2, 0, 0) == h16                    # fp16 tensors alike
    assert lib.mhla_blockmix_fwd_ws_bytes(8, 16, 64, 64, 64, 1, 0, _lib.FLAG_FP32_GRADE_SUMMARIES) == 2 * rows * (3072 + 288) * 4 + small(64, 64, 64)
    assert lib.mhla_blockmix_fwd_ws_bytes(8, 16, 512, 8, 64, 1, 0, 0) > lib.mhla_blockmix_fwd_ws_bytes(8, 16, 512, 8, 64, 0, 0, 0) * 0.99   # > 256 blocks: fp32 words for every dtype
    assert lib.mhla_blockmix_fwd_ws_bytes(8, 16, 64, 8, 64, 1, 0, 0) == 2 * rows * (3072 + 288) * 4 + small(64, 8, 64)   # blocks of < 16 tokens: 24-bit floats
    # (C2-like shapes -- D <= 64, up to 128 blocks -- form the backward's row dots from G and keep nothing else; elsewhere the forward
    # keeps the bf16 residual of its store of O, B N H D * 2 bytes: here D = 72)
    f72 = lib.mhla_blockmix_fwd_ws_bytes(8, 16, 64, 64, 72, 0, 0, 0)
    assert f72 == 2 * rows * (5184 + 288) * 4 + small(64, 64, 72)
    assert lib.mhla_blockmix_fwd_ws_bytes(8, 16, 64, 64, 72, 1, 0, 0) == 2 * rows * (2592 + 288) * 4 + small(64, 64, 72) + 8 * 4096 * 16 * 72 * 2
    assert lib.mhla_blockmix_fwd_ws_bytes(8, 16, 64, 64, 64, 1, 0, _lib.FLAG_BF16_SUMMARIES) < 0.6 * fwd
    assert lib.mhla_blockmix_fwd_ws_bytes(8, 16, 64, 64, 64, 0, 0, _lib.FLAG_BF16_SUMMARIES) == fwd
    assert lib.mhla_blockmix_fwd_ws_bytes(8, 16, 64, 64, 64, 0, 0, _lib.FLAG_FP32_GRADE_SUMMARIES) == fwd
    assert lib.mhla_blockmix_fwd_keeps_state(8, 16, 64, 64, 64, 1, 0, 0) == 1
    assert 0 < fwd < bwd
    assert lib.mhla_causal_bwd_ws_bytes(1, 8192, 4, 128, 256, 64, 0, 0) > lib.mhla_causal_fwd_ws_bytes(1, 8192, 4, 128, 256, 64, 0, 0) > 0
    # bf16 tensors: chunk summaries in 2 bytes by default (h16), as bf16 hi + lo pairs (as many bytes as fp32) or single bf16 with the opt-in flags
    # (+ 2176 bytes of padding per chunk tile -- 8 tiles of 64 x 64 at K = 128, V = 256 -- and summary set on the 16-bit
    # pipeline: fast::cs_layout)
    f32 = lib.mhla_causal_fwd_ws_bytes(1, 8192, 4, 128, 256, 64, 0, 0)
    pad = 2 * 4 * 128 * 8 * 2176
    assert lib.mhla_causal_fwd_ws_bytes(1, 8192, 4, 128, 256, 64, 1, _lib.CAUSAL_FP32_GRADE_SUMMARIES) == f32 + pad
    assert lib.mhla_causal_fwd_ws_bytes(1, 8192, 4, 128, 256, 64, 1, 0) == f32 // 2 + pad      # default: 2-byte h16 summaries (the strip multipliers sit in the padding)
    assert lib.mhla_causal_fwd_ws_bytes(1, 8192, 4, 128, 256, 64, 1, _lib.CAUSAL_BF16_SUMMARIES) == f32 // 2 + pad
    assert lib.mhla_causal_fwd_ws_bytes(1, 8192, 4, 128, 256, 64, 1, _lib.CAUSAL_FORCE_GENERIC) == f32
    # the fused norm x gate epilogue: what the 16-bit pipeline covers with V <= 256
    assert lib.mhla_causal_normgate_fusable(8192, 128, 256, 64, 1, 0) == 1
    assert lib.mhla_causal_normgate_fusable(8192, 256, 512, 64, 1, 0) == 1    # wide heads: two halves per workgroup
    assert lib.mhla_causal_normgate_fusable(8192, 128, 320, 64, 1, 0) == 0
    assert lib.mhla_causal_normgate_fusable(8256, 128, 256, 64, 1, 0) == 1    # 129 chunks: sixteen-wave mixing kernels
    assert lib.mhla_causal_normgate_fusable(16448, 128, 256, 64, 1, 0) == 0   # 257 chunks: generic kernels
    assert lib.mhla_causal_normgate_fusable(8192, 128, 256, 64, 0, 0) == 0    # fp32 tensors


def test_argument_validation_without_gpu():
    """Shape / alignment errors are reported by code + message before anything touches the device."""
    from mhla_amd import _lib
    lib = _lib.load()
    V = _lib.View
    bad = V(8, 64, 64, 64)          # misaligned pointer
    ok = V(1 << 20, 64 * 16, 64, 64)
    rc = lib.mhla_blockmix_fwd(bad, ok, ok, ok, ok, 1 << 20, 4, ok, None, 1 << 20, 1 << 30, 1, 1, 4, 16, 64, 0, 1e-6, 0, None)
    assert rc == -22 and b"aligned" in lib.mhla_last_error()
    rc = lib.mhla_blockmix_fwd(ok, ok, ok, ok, ok, 1 << 20, 4, ok, None, 1 << 20, 1 << 30, 1, 1, 4, 16, 192, 0, 1e-6, 0, None)
    assert rc == -95
    # (S = 32: blocks of 16 tokens take the single-launch small-sequence kernels, which need no workspace)
    rc = lib.mhla_blockmix_fwd(ok, ok, ok, ok, ok, 1 << 20, 4, ok, None, 1 << 20, 16, 1, 1, 4, 32, 64, 0, 1e-6, 0, None)
    assert rc == -22 and b"workspace" in lib.mhla_last_error()
    rc = lib.mhla_causal_fwd(ok, ok, ok, 1 << 20, 4, ok, 1 << 20, 1 << 30, 1, 128, 1, 64, 64, 32, 0.125, 0, 0, None)
    assert rc == -95 and b"chunk" in lib.mhla_last_error()
    rc = lib.mhla_causal_fwd(ok, ok, ok, 1 << 20, 4, ok, 1 << 20, 1 << 30, 1, 128, 1, 64, 64, 64, 0.125, 0, 8, None)
    assert rc == -22 and b"flags" in lib.mhla_last_error()
    # the summary-format flags: single-bf16 summaries serve bf16 tensors only; the two opt-ins exclude each other
    rc = lib.mhla_blockmix_fwd(ok, ok, ok, ok, ok, 1 << 20, 4, ok, None, 1 << 20, 1 << 30, 1, 1, 4, 32, 64, _lib.F16, 1e-6, _lib.FLAG_BF16_SUMMARIES, None)
    assert rc == -22 and b"bf16 tensors only" in lib.mhla_last_error()
    rc = lib.mhla_blockmix_fwd(ok, ok, ok, ok, ok, 1 << 20, 4, ok, None, 1 << 20, 1 << 30, 1, 1, 4, 32, 64, _lib.BF16, 1e-6,
                               _lib.FLAG_BF16_SUMMARIES | _lib.FLAG_FP32_GRADE_SUMMARIES, None)
    assert rc == -22 and b"exclude" in lib.mhla_last_error()
    rc = lib.mhla_causal_fwd(ok, ok, ok, 1 << 20, 4, ok, 1 << 20, 1 << 30, 1, 128, 1, 64, 64, 64, 0.125, _lib.BF16,
                             _lib.CAUSAL_BF16_SUMMARIES | _lib.CAUSAL_FP32_GRADE_SUMMARIES, None)
    assert rc == -22 and b"exclude" in lib.mhla_last_error()
    # the forward of an inference call (MHLA_FLAG_NO_BWD_STATE) does not carve the O-residual region (D = 72: outside the row-dots-from-G shapes)
    assert (lib.mhla_blockmix_fwd_ws_bytes(8, 16, 64, 64, 72, _lib.BF16, 0, 0) - lib.mhla_blockmix_fwd_ws_bytes(8, 16, 64, 64, 72, _lib.BF16, 0, _lib.FLAG_NO_BWD_STATE)
            == 8 * 4096 * 16 * 72 * 2)


@pytest.mark.parametrize("tr", ["linear", "cos", "exp", "gaussian", "local"])
def test_weight_init_matches_reference(tr):
    from mhla_amd import block_distance_weights
    g = load_golden("weight_init")
    for key, layout in ((f"w2d_{tr}_16_16", (4, 4)), (f"w2d_{tr}_21_49", (3, 3)), (f"w2d_{tr}_16_4", (8, 8)),
                        (f"w3d_{tr}_3_5_10", (3, 5, 10)), (f"w3d_{tr}_2_3_4", (2, 3, 4)), (f"w3d_{tr}_1_4_4", (1, 4, 4))):
        w = block_distance_weights(layout, tr)
        assert torch.allclose(w, g[key], atol=2e-6), key


def test_block_index_maps_match_oracle():
    from mhla_amd import block_index_2d, block_index_3d
    from oracle import mhla_oracle as orc
    assert torch.equal(block_index_2d(4, 4).long(), orc.block_index_2d(4, 4))
    assert torch.equal(block_index_3d((3, 10, 20), (3, 5, 10)).long(), orc.block_index_3d((3, 10, 20), (3, 5, 10)))
    with pytest.raises(ValueError):
        block_index_3d((3, 10, 21), (3, 5, 10))


def test_module_state_dicts_are_reference_compatible():
    from mhla_amd import modules
    for tag, cls, kw in (
        ("dit_a", modules.MHLA4DiT, dict(heads=2, dim_head=32, block_size=16, embed_len=256, qkv_bias=True)),
        ("vit_a", modules.MHLA_Normed_Torch, dict(heads=2, dim_head=64, window_size=16, embed_len=256, qk_norm=True)),
    ):
        g = load_golden("blockmix2d_" + tag)
        sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
        m = cls(kw["heads"] * kw["dim_head"], **kw)
        assert set(m.state_dict()) == set(sd)
        m.load_state_dict(sd, strict=True)
    g = load_golden("wan_b")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    m = modules.MHLA_Video_Uni(64, num_heads=2, block_layout=(3, 5, 10), is_gated=True, normalize_out=False)
    assert set(m.state_dict()) == set(sd)
    f = modules.MHLA(hidden_size=256, num_heads=2, feature_map="relu")
    assert set(f.state_dict()) == {"q_proj.weight", "k_proj.weight", "v_proj.weight", "g_proj.weight", "o_proj.weight",
                                   "mixing_matrix", "g_norm_swish_gate.weight"}
    assert f.mixing_matrix.shape == (32, 32, 1, 1, 1, 1)
    from oracle import mhla_oracle as orc
    assert torch.equal(f.mixing_matrix.detach().reshape(32, 32), orc.causal_mixing_init(32))


def test_product_path_has_no_cpu_fallback():
    import mhla_amd
    from mhla_amd import modules
    q = torch.randn(1, 64, 2, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mhla_amd.mhla_blockmix(q, q, q, torch.eye(4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mhla_amd.mhla_causal(q, q, q, torch.eye(4))
    m = modules.MHLA4DiT(128, heads=2, block_size=16, embed_len=256)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 16, 16, 128))
    # and nothing in the product package imports the oracle
    pkg = os.path.join(ROOT, "mhla_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dp, fn)).read()
                assert "oracle" not in src.replace("no CPU", ""), f"{fn} mentions the oracle"


def test_wan_rope_and_fla_rotary_host_math_match_golden():
    from mhla_amd.modules import wan, fla
    g = load_golden("wan_a")
    B, H, D, M, S, fb, hb, wb, F_, H_, W_, normalize, gated = [int(x) for x in g["meta"]]
    cos, sin = wan._rope_table(wan.wan_freqs(D), (F_, H_, W_), "cpu")
    y = wan.rope_apply(g["q"], cos, sin)
    assert (y - g["q_rope"]).abs().max() < 1e-5
    gn = load_golden("fla_neighbours")
    r = fla.RotaryEmbedding(32)
    q, _ = r(gn["x"], gn["x"])
    assert (q - gn["rot"]).abs().max() < 1e-6


WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ["MHLA_ROOT"])
import torch.distributed as dist
from mhla_amd import dist as mdist
from oracle import mhla_oracle as orc
rank, local, world = mdist.init_from_env("gloo")
assert world == 2
# global batch of 6 samples sharded over 2 ranks; per-rank dW (oracle stands in for the GPU op here:
# this test covers the sharding + dW all-reduce + timing harness, not the kernels)
g = torch.Generator().manual_seed(0)
B, N, H, D, M = 6, 64, 2, 16, 4
q = torch.relu(torch.randn(B, N, H, D, generator=g)) + 1e-6
k = torch.relu(torch.randn(B, N, H, D, generator=g)) + 1e-6
v = torch.randn(B, N, H, D, generator=g)
do = torch.randn(B, N, H, D, generator=g)
W = orc.block_distance_weights((2, 2), "linear")
lo, hi = mdist.shard_batch(B, rank, world)
assert (lo, hi) == ((0, 3) if rank == 0 else (3, 6))
dW_local = orc.blockmix_bwd(q[lo:hi], k[lo:hi], v[lo:hi], W, do[lo:hi])["dW"]
dW = mdist.allreduce_mean_(dW_local.clone())
full = orc.blockmix_bwd(q, k, v, W, do)["dW"] / world
assert torch.allclose(dW, full, rtol=1e-4, atol=1e-6), (dW - full).abs().max()
red = mdist.OverlappedGradAllReduce()
t2 = dW_local.clone()
red.issue(t2)          # asynchronous mean all-reduce, as bench.py schedules it
red.wait()
assert torch.allclose(t2, full, rtol=1e-4, atol=1e-6)
calls = []
el = mdist.timed_steps(lambda: calls.append(1), steps=4, warmup=2, sync=lambda: None)
assert len(calls) == 6 and el >= 0
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_world_size_2_gloo_sharding_and_dw_allreduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MHLA_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    from mhla_amd.dist import shard_batch
    assert [shard_batch(7, r, 3) for r in range(3)] == [(0, 3), (3, 5), (5, 7)]


SPAWN_WORKER = r"""
import os, sys, torch
sys.path.insert(0, os.environ["MHLA_ROOT"])
import torch.distributed as dist
from mhla_amd import dist as mdist
assert mdist.launched_by_rendezvous()
rank, local, world = mdist.init_from_env("gloo")
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
assert world == int(sys.argv[1]) and t.item() == world * (world + 1) / 2
if len(sys.argv) > 2 and rank == int(sys.argv[2]):
    sys.exit(7)          # a failing rank: the parent must stop the others and report the code
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("SPAWN_OK", world, flush=True)
"""


def test_spawn_local_ranks_starts_fresh_children(tmp_path, capfd):
    """What `bench.py --gpus N` does without torch.distributed.run: N child ranks, rendezvous on 127.0.0.1, status relayed."""
    from mhla_amd.dist import spawn_local_ranks
    script = tmp_path / "spawn_worker.py"
    script.write_text(SPAWN_WORKER)
    env = {"MHLA_ROOT": ROOT, "OMP_NUM_THREADS": "1"}
    assert spawn_local_ranks(2, [sys.executable, str(script), "2"], env, timeout=240) == 0
    assert "SPAWN_OK 2" in capfd.readouterr().out
    assert spawn_local_ranks(2, [sys.executable, str(script), "2", "1"], env, timeout=240) == 7


def test_bench_gpus_flag_spawns_ranks_before_touching_the_gpu():
    """On a box without a GPU every spawned rank must fail loudly ("needs a GPU"), once per rank, and the parent relays it --
    proof that `python bench.py --gpus 2` starts two ranks (RANK 0 and 1) instead of benchmarking one device."""
    if torch.cuda.is_available():
        pytest.skip("CPU-side test of the spawn path")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "ranks share devices" in r.stderr and "needs a GPU" in r.stderr
    assert "spawn_local_ranks" in r.stderr


def test_c1_dit_s2_single_latent_cpu_plumbing():
    """BASELINE.json configs[0]: mhla_dit DiT-S/2, one 256x256 image (32x32x4 latent), eager MHLA on the CPU -- the plumbing
    configuration.  The thin in-repo host runs on the CPU with every attention module evaluated by the oracle's restatement of
    MHLA4DiT.forward (the reference's eager op sequence; the product modules have no CPU path), end to end: patch embedding,
    block-major adapter, 12 adaLN blocks, final layer, unpatchify.  Checks shapes, finiteness, the adapter's inverse and that
    the class / timestep conditioning reaches the output."""
    import time
    from mhla_amd.hosts import DiT_MHLA, DiT_configs
    from oracle import mhla_oracle as orc
    torch.manual_seed(0)
    m = DiT_MHLA(input_size=32, **DiT_configs()["DiT-S/2"])
    with torch.no_grad():   # adaLN-Zero leaves every block an identity at init: give the plumbing run non-trivial activations
        for prm in m.parameters():
            if prm.requires_grad and float(prm.abs().max()) == 0.0:
                prm.normal_(std=0.02)
    m.eval()
    assert len(m.blocks) == 12 and m.blocks[0].attn.num_heads == 6 and m.blocks[0].attn.head_dim == 64
    assert m.blocks[0].attn.num_pieces == 16 and m.blocks[0].attn.block_size == 16      # block_size = 16: mhla_dit/README.md:23
    for blk in m.blocks:
        sd = {k: v.detach() for k, v in blk.attn.state_dict().items()}
        at = blk.attn
        blk.attn.forward = (lambda sd, at: lambda z: orc.dit_module_forward(
            sd, z.reshape(z.shape[0], at.num_pieces, at.block_size, z.shape[-1]), at.num_heads, at.block_size, at.embed_len
        ).reshape(z.shape))(sd, at)
    x = torch.randn(1, 4, 32, 32)
    t0 = time.perf_counter()
    with torch.no_grad():
        y = m(x, torch.tensor([500]), torch.tensor([7]))
        y2 = m(x, torch.tensor([500]), torch.tensor([8]))
        y3 = m(x, torch.tensor([10]), torch.tensor([7]))
    dt = (time.perf_counter() - t0) / 3
    assert y.shape == (1, 8, 32, 32) and torch.isfinite(y).all()
    assert (y - y2).abs().max() > 0 and (y - y3).abs().max() > 0
    tok = torch.arange(256)
    assert torch.equal(tok[m.to_block_major][m.to_raster], tok)
    print(f"C1 DiT-S/2 256x256 single latent, CPU eager (oracle attention): {dt * 1e3:.0f} ms per forward, {256 / dt:.0f} tokens/s")


def test_short_convolution_matches_oracle_restatement():
    """The fla layer's optional ShortConvolution (plain PyTorch, host side): full sequences, continuation from a cache, the
    single-token decoding steps and packed sequences, against the oracle's explicit-loop restatement."""
    import torch
    from oracle import mhla_oracle as orc
    from mhla_amd.modules.fla import ShortConvolution
    torch.manual_seed(0)
    m = ShortConvolution(12, 4, bias=True)
    x = torch.randn(2, 9, 12)
    want = orc.short_conv(x, m.weight, m.bias)
    y, c = m(x, output_final_state=True)
    assert (y - want).abs().max() < 1e-6 and c.shape == (2, 12, 4)
    y1, c1 = m(x[:, :5], output_final_state=True)
    y2, c2 = m(x[:, 5:], cache=c1, output_final_state=True)
    assert (torch.cat([y1, y2], 1) - want).abs().max() < 1e-6 and torch.equal(c2, c)
    assert (y2 - orc.short_conv(x[:, 5:], m.weight, m.bias, initial_state=c1)).abs().max() < 1e-6
    steps, cc = [], None
    for t in range(9):
        yt, cc = m(x[:, t:t + 1], cache=cc, output_final_state=True)
        steps.append(yt)
    assert (torch.cat(steps, 1) - want).abs().max() < 1e-6
    xp, cu = torch.randn(1, 10, 12), torch.tensor([0, 4, 10])
    yv, cv = m(xp, cu_seqlens=cu, output_final_state=True)
    assert (yv - orc.short_conv(xp, m.weight, m.bias, cu_seqlens=cu)).abs().max() < 1e-6 and cv.shape == (2, 12, 4)
    assert sorted(m.state_dict()) == ["bias", "weight"] and m.weight.shape == (12, 1, 4)


def test_bench_line_is_small_strict_json():
    """The driver parses the LAST stdout line of bench.py; round 5's 20.6 KB line did not parse.  The line is now built by
    bench.compact_line() from the full record: contract keys only, strict JSON (no NaN), under 4 KB -- whatever the record holds."""
    import copy
    import json
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "r5_bench.json")).read().strip().splitlines()[-1])   # a real 20.6 KB record
    assert len(json.dumps(full)) > 16000
    line = bench.compact_line(copy.deepcopy(full))
    assert len(line) < 4096 and "\n" not in line
    got = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))   # NaN / Infinity would raise
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in got, k
    assert got["value"] == pytest.approx(full["value"], rel=1e-5) and got["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert got["config"]["workload"] == full["config"]["workload"] and "model" not in got["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in got["roofline"], k
    assert got["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in got["cpu_baseline"], k
    assert set(got["targets"]) == {"ge_10x_cpu_on_dit_xl2_256", "within_1e-3_of_reference", "ge_40pct_mfma", "ge_6x_at_8_gpus"}
    # a hostile record: NaNs, infinities, bloated strings and blocks -- still one small strict line
    bad = copy.deepcopy(full)
    bad["roofline"]["traffic"] = float("nan")
    bad["roofline"]["frac_from_gpu_events"] = float("inf")
    bad["config"]["launch"] = "x" * 5000
    bad["config"]["arithmetic"] = "y" * 5000
    bad["cpu_baseline"]["sample"] = "z" * 5000
    bad["dit_xl2_train_step"] = {"error": "e" * 9000}
    line = bench.compact_line(bad)
    assert len(line) < 4096
    got = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert got["roofline"]["traffic"] is None and got["value"] == pytest.approx(full["value"], rel=1e-5)


def test_describe_dispatch_names_the_path_of_every_baseline_config():
    """mhla_describe_dispatch / mhla_causal_describe_dispatch (pure host logic): the kernel family and block-summary format a
    BASELINE.json configuration takes, asserted instead of inferred from timings."""
    import mhla_amd
    bf, f32, f16 = torch.bfloat16, torch.float32, torch.float16
    c2 = mhla_amd.describe_dispatch(8, 16, 64, 64, 64, bf)                                      # configs[1]
    assert c2["family"].startswith("split-operand") and c2["summaries"].startswith("h16")
    assert c2["fwd"] == ["k_sp_state", "k_sp_mixh<0>", "k_sp_out"]
    assert c2["bwd"] == ["k_sp_state<1>", "k_sp_mixh<1,dw>", "k_dw_reduce", "k_sp_bwd_dq", "k_sp_bwd_dkv"]
    assert mhla_amd.describe_dispatch(8, 16, 64, 64, 64, bf, summaries="split")["summaries"].startswith("p24")
    assert mhla_amd.describe_dispatch(8, 16, 64, 64, 64, f16)["summaries"].startswith("h16")
    assert mhla_amd.describe_dispatch(8, 16, 64, 64, 64, bf, summaries="bf16")["family"].startswith("bf16 fast path")
    assert mhla_amd.describe_dispatch(8, 16, 64, 64, 64, f32)["summaries"] == "fp32 words"
    assert mhla_amd.describe_dispatch(8, 16, 16, 256, 64, bf)["summaries"].startswith("h16")   # C2 variant 16 x 256
    c2b = mhla_amd.describe_dispatch(8, 16, 256, 16, 64, bf)                                    # C2 variant 256 x 16: more than 128 blocks
    assert c2b["summaries"].startswith("h16") and c2b["bwd"][1:3] == ["k_sp_mixh2<1>", "k_sp_dwr<4,h16>"]   # 32 slices per workgroup: the re-cut mixing kernel
    assert mhla_amd.describe_dispatch(1, 6, 256, 256, 64, bf)["bwd"][1] == "k_sp_mixh<1>"                 # ... two slices per workgroup: not worth its rebuilds
    prev = mhla_amd.set_option("recut_kernels", 0)                                               # the A/B switch of the re-cut kernels (mhla_hip.h)
    try:
        assert mhla_amd.describe_dispatch(8, 16, 256, 16, 64, bf)["bwd"][1] == "k_sp_mixh<1>"
    finally:
        assert mhla_amd.set_option("recut_kernels", prev) == 0
    assert mhla_amd.describe_dispatch(8, 16, 256, 16, 64, bf)["fwd"][1] == "k_sp_mixh2<0>"
    assert mhla_amd.describe_dispatch(8, 16, 320, 16, 64, bf)["summaries"] == "fp32 words"     # more than 256 blocks: the tiled mixing
    assert mhla_amd.describe_dispatch(8, 16, 64, 8, 64, bf)["summaries"].startswith("p24")     # blocks of fewer than 16 tokens
    assert mhla_amd.describe_dispatch(8, 16, 2, 64, 64, bf)["summaries"].startswith("p24")     # fewer than 4 blocks
    c3 = mhla_amd.describe_dispatch(32, 16, 16, 16, 72, bf)                                     # configs[2]: DiT-XL/2 256^2
    assert c3["family"].startswith("small-sequence bf16") and c3["fwd"] == ["k_sn_fwd<5,hl>"] and c3["bwd"][0] == "k_sn_bwd<5,hl>"
    assert mhla_amd.describe_dispatch(32, 16, 16, 16, 72, f32)["family"].startswith("small-sequence fp32")
    assert mhla_amd.describe_dispatch(32, 16, 16, 16, 72, bf, no_smalln=True)["summaries"].startswith("h16")
    c4 = mhla_amd.describe_dispatch(1, 12, 150, 210, 128, f32, split=True)                     # configs[3]: Wan2.1-1.3B
    assert c4["summaries"].startswith("p24") and "k_sp_mixr<0>" in c4["fwd"] and "k_sp_dwt" in c4["bwd"]
    assert mhla_amd.describe_dispatch(2, 2, 16, 16, 36, f32)["family"].startswith("generic")   # D % 8 != 0
    for K, V in ((128, 256), (256, 512)):                                                      # configs[4]: fla 340M / 1.3B-like
        c5 = mhla_amd.describe_causal_dispatch(8192, K, V, bf)
        assert c5["family"].startswith("16-bit pipeline") and c5["summaries"].startswith("h16") and c5["bwd"][1] == "k_csf_mixb"
    assert mhla_amd.describe_causal_dispatch(8192, 128, 256, bf, summaries="split")["summaries"].startswith("bf16 hi + lo")
    assert mhla_amd.describe_causal_dispatch(8192, 128, 256, f32)["family"].startswith("generic")
    assert mhla_amd.describe_causal_dispatch(16448, 64, 64, bf)["family"].startswith("generic")   # 257 chunks
