#!/usr/bin/env python
"""Step-level measurements for BASELINE.json configs[3] and [4] with the thin in-repo hosts, importable by bench.py:

  wan_forward(...)   Wan2.1-1.3B Full-MHLA inference: the 30-block transformer body of one denoising step (hosts/wan.py around the
                     MHLA_Video_Uni drop-in), 81 frames at 832 x 480 = 31 500 video tokens, bf16 weights, fp32 attention path (the
                     module's .float(), wan/mhla_utils.py:308), no_grad; B = 1 and B = 2 (classifier-free guidance batches the two
                     passes, dpm_solver.py:462-463).  Reference: mhla_videogen/diffusion/model/wan/model.py:2525-2660.
  gpt_step(...)      fla GPT-style LM training step (fwd, cross-entropy, bwd, AdamW; bf16 autocast) at seq_len 8192 with the minimal
                     host (hosts/gpt.py around the fla MHLA drop-in); DDP over RCCL when the job has several ranks.
                     Reference: mhla_nlp/fla/models/gla/modeling_gla.py:240-300.

Each result carries ms, tokens/s and the share of the step's GPU time spent inside the library's kernels (`mhla_kernel_share`:
sum of the per-launch HIP-event durations of mhla_* launches over the GPU time of the same steps).
  python tools/bench_steps.py [wan] [gpt340m] [gpt1p3b]"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _kernel_share(step, iters=2):
    """(GPU ms per step, ms of it inside libmhla_hip.so kernels, {kernel: ms per step}) over `iters` steps, HIP events."""
    import mhla_amd
    lib = mhla_amd._lib.load()
    torch.cuda.synchronize()
    lib.mhla_prof_enable(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    lib.mhla_prof_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.mhla_prof_report(buf, len(buf))
    ks = {}
    for line in buf.value.decode().splitlines():
        name, cnt, tot = line.rsplit(" ", 2)
        ks[name] = float(tot) / iters
    return e0.elapsed_time(e1) / iters, sum(ks.values()), ks


def _time(step, iters, warm):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def wan_forward(dev, layers=30, iters=3, warm=1, batches=(1, 2)):
    from mhla_amd import modules
    from mhla_amd.hosts import WanStack_MHLA
    torch.manual_seed(0)
    dim, heads, grid = 1536, 12, (21, 30, 50)
    N = grid[0] * grid[1] * grid[2]
    net = WanStack_MHLA(num_layers=layers, dim=dim, ffn_dim=8960, num_heads=heads).to(dev).to(torch.bfloat16).eval()
    freqs = modules.wan_freqs(dim // heads)
    out = {"what": f"Wan2.1-1.3B Full-MHLA inference, {layers}-block transformer body of one denoising step (thin host, MHLA_Video_Uni "
                   "drop-in), 81 frames x 832 x 480 = 31 500 tokens, bf16 weights, fp32 attention path, no_grad",
           "config": "BASELINE.json configs[3]", "layers": layers, "tokens": N, "runs": []}
    for B in batches:
        x = torch.randn(B, N, dim, device=dev, dtype=torch.bfloat16)
        e = torch.randn(B, 6, dim, device=dev, dtype=torch.float32) * 0.1
        ctx = torch.randn(B, 512, dim, device=dev, dtype=torch.bfloat16)
        gs = torch.tensor([list(grid)] * B, dtype=torch.long)
        sl = torch.tensor([N] * B)

        def step():
            with torch.no_grad():
                return net(x, e, sl, gs, freqs, ctx)

        ms = _time(step, iters, warm)
        gpu_ms, lib_ms, ks = _kernel_share(step, 1)
        out["runs"].append({"B": B, "note": "classifier-free guidance batch" if B == 2 else "one sample", "ms_per_forward": ms,
                            "ms_per_block": ms / layers, "tokens_per_s": B * N / ms * 1e3, "mhla_kernel_share": lib_ms / gpu_ms,
                            "mhla_kernel_ms": lib_ms, "top_mhla_kernels_ms": dict(sorted(ks.items(), key=lambda kv: -kv[1])[:4])})
        del x, e, ctx
    del net
    torch.cuda.empty_cache()
    return out


def gpt_step(rank, local, world, model="340M", batch=2, seq=8192, steps=4, warmup=2, bucket_cap_mb=25):
    from mhla_amd import dist as mdist
    from mhla_amd.hosts import GPT_MHLA, GPT_configs
    dev = torch.device("cuda", local)
    torch.manual_seed(1234 + rank)
    net0 = GPT_MHLA(**GPT_configs()[model], max_seq_len=seq).to(dev)
    ddp = world > 1
    net = torch.nn.parallel.DistributedDataParallel(net0, device_ids=[local], bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True) if ddp else net0
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4)
    ids = torch.randint(0, 32000, (batch, seq), device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = net(ids, labels=ids)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    el = mdist.timed_steps(step, steps, warmup, torch.cuda.synchronize)
    gpu_ms, lib_ms, ks = _kernel_share(step, 1)
    nparam = sum(p.numel() for p in net0.parameters())
    cfg = GPT_configs()[model]
    hk, hv = cfg["hidden_size"] // 2 // cfg["num_heads"], cfg["hidden_size"] // cfg["num_heads"]
    res = {"what": f"GPT-{model} MHLA LM training step (fwd, cross-entropy, bwd, AdamW), seq_len {seq}, minimal host around the fla MHLA "
                   "drop-in, bf16 autocast, synthetic tokens",
           "config": "BASELINE.json configs[4]" + ("" if model == "1.3B" else " (the 340M model the reference ships a config for)"),
           "layer_shape": f"{cfg['num_heads']} heads, K={hk}, V={hv}, {seq // 64} chunks of 64",
           "n_gpus": world, "per_gpu_batch": batch, "seq_len": seq, "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
           "tokens_per_s": batch * seq * world / (el / steps), "params_M": nparam / 1e6,
           "mhla_kernel_share": lib_ms / gpu_ms, "mhla_kernel_ms": lib_ms,
           "top_mhla_kernels_ms": dict(sorted(ks.items(), key=lambda kv: -kv[1])[:5]),
           "gradient_exchange": ("none (one rank)" if not ddp else
                                 f"DistributedDataParallel over {torch.distributed.get_backend()}: {nparam * 4 / 1e9:.2f} GB of fp32 gradients per "
                                 f"step, bucket_cap_mb={bucket_cap_mb}, gradient_as_bucket_view=True, all-reduce overlapped with backward")}
    del opt, net, net0
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    which = sys.argv[1:] or ["wan", "gpt340m", "gpt1p3b"]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if "wan" in which:
        print(json.dumps(wan_forward(dev)), flush=True)
    if "gpt340m" in which:
        print(json.dumps(gpt_step(0, 0, 1, "340M")), flush=True)
    if "gpt1p3b" in which:
        print(json.dumps(gpt_step(0, 0, 1, "1.3B", batch=1)), flush=True)
