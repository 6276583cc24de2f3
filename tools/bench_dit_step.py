#!/usr/bin/env python
"""Step-level number for BASELINE.json configs[2]: a DiT training step (forward, MSE loss on the noise half, backward, AdamW)
with the thin in-repo host (mhla_amd.hosts.DiT_MHLA) around the MHLA4DiT drop-in, synthetic latents, bf16 autocast.

  python tools/bench_dit_step.py [--model DiT-XL/2] [--batch 32] [--steps 10] [--warmup 3]
  python tools/bench_dit_step.py --gpus N                                            (starts N ranks itself)
  python -m torch.distributed.run --nproc-per-node N ... tools/bench_dit_step.py     (DDP over RCCL, one process per GPU)

Informational (bench.py is the contract benchmark); prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mhla_amd import dist as mdist  # noqa: E402
from mhla_amd.hosts import DiT_MHLA, DiT_configs  # noqa: E402


def run_dit_step(rank, local, world, model_name="DiT-XL/2", batch=32, image=256, steps=10, warmup=3, bucket_cap_mb=25, force_ddp=False):
    """Training steps of the thin DiT host on this rank's GPU; with world > 1 the model is wrapped in DistributedDataParallel
    (the default process group must exist): bucketed RCCL all-reduce of the fp32 gradients (2.7 GB at DiT-XL/2), overlapped
    with the backward by DDP's reducer.  `force_ddp`: wrap the model also in a one-rank group (the test of the DDP + RCCL path on a
    one-GPU box).  Returns the result dict on every rank (max-over-ranks time)."""
    dev = torch.device("cuda", local)
    torch.manual_seed(1234 + rank)
    latent = image // 8
    model = DiT_MHLA(input_size=latent, **DiT_configs()[model_name]).to(dev)
    # zero-initialised adaLN / head would make every block an identity: give the benchmark non-trivial activations
    with torch.no_grad():
        for prm in model.parameters():
            if prm.requires_grad and float(prm.abs().max()) == 0.0:
                prm.normal_(std=0.02)
    ddp = world > 1 or force_ddp
    if ddp:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)
    else:
        net = model
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=0.0)
    x = torch.randn(batch, 4, latent, latent, device=dev)
    noise = torch.randn_like(x)
    t = torch.randint(0, 1000, (batch,), device=dev)
    y = torch.randint(0, 1000, (batch,), device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(x, t, y)
        loss = torch.nn.functional.mse_loss(out[:, :4].float(), noise)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        for blk in model.blocks:   # the post-step clamp of the mixing weights, mhla_dit/train.py:308-310
            blk.attn.piece_attn.conv.weight.data.clamp_(min=0)

    el = mdist.timed_steps(step, steps, warmup, torch.cuda.synchronize)
    tokens = batch * (latent // 2) ** 2 * world
    nparam = sum(p.numel() for p in model.parameters())
    res = {"what": f"{model_name} {image}x{image} training step (fwd, MSE, bwd, AdamW), thin host around the MHLA4DiT drop-in, bf16 autocast",
           "n_gpus": world, "per_gpu_batch": batch, "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
           "images_per_s": batch * world / (el / steps), "tokens_per_s": tokens / (el / steps),
           "params_M": nparam / 1e6,
           "gradient_exchange": ("none (one rank)" if not ddp else
                                 f"DistributedDataParallel over {torch.distributed.get_backend()}: {nparam * 4 / 1e9:.2f} GB of fp32 gradients per "
                                 f"step, bucket_cap_mb={bucket_cap_mb}, gradient_as_bucket_view=True, all-reduce overlapped with backward")}
    del opt, net, model
    torch.cuda.empty_cache()
    return res


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--model", default="DiT-XL/2", choices=sorted(DiT_configs()))
    p.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    p.add_argument("--image", type=int, default=256, help="image side in pixels (latent side = image / 8)")
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--bucket-cap-mb", type=int, default=25)
    p.add_argument("--gpus", type=int, default=1, help="ranks to start when not launched by torch.distributed.run")
    a = p.parse_args()
    if a.gpus > 1 and not mdist.launched_by_rendezvous():   # parent: start the ranks before any GPU call, relay the status
        extra = {"MHLA_DIST_BACKEND": "gloo"} if torch.cuda.device_count() < a.gpus else {}
        sys.exit(mdist.spawn_local_ranks(a.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], extra))
    rank, local, world = mdist.init_from_env()
    local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    res = run_dit_step(rank, local, world, a.model, a.batch, a.image, a.steps, a.warmup, a.bucket_cap_mb)
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
