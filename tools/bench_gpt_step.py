#!/usr/bin/env python
"""Step-level number for BASELINE.json configs[4]: a GPT-style LM training step (forward, cross-entropy, backward, AdamW) with
the minimal in-repo host (mhla_amd.hosts.GPT_MHLA) around the fla MHLA drop-in, synthetic tokens, bf16 autocast.

  python tools/bench_gpt_step.py [--model 340M] [--batch 8] [--seq 2048] [--steps 5] [--warmup 2]
  python -m torch.distributed.run --nproc-per-node N ... tools/bench_gpt_step.py          (DDP over RCCL)

`--seq 8192` is the BASELINE.json configs[4] sequence length (the host sizes the layer's mixing matrix: 128 chunks of 64).
`--gpus N` starts N ranks itself (before any GPU call).  Informational; one JSON line on rank 0."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mhla_amd import dist as mdist  # noqa: E402
from mhla_amd.hosts import GPT_MHLA, GPT_configs  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--model", default="340M", choices=sorted(GPT_configs()))
    p.add_argument("--batch", type=int, default=8)
    p.add_argument("--seq", type=int, default=2048)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--gpus", type=int, default=1, help="ranks to start when not launched by torch.distributed.run")
    a = p.parse_args()
    if a.gpus > 1 and not mdist.launched_by_rendezvous():
        extra = {"MHLA_DIST_BACKEND": "gloo"} if torch.cuda.device_count() < a.gpus else {}
        sys.exit(mdist.spawn_local_ranks(a.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], extra))
    rank, local, world = mdist.init_from_env()
    local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(1234 + rank)
    model = GPT_MHLA(**GPT_configs()[a.model], max_seq_len=a.seq).to(dev)
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local]) if world > 1 else model
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4)
    ids = torch.randint(0, 32000, (a.batch, a.seq), device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = net(ids, labels=ids)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    el = mdist.timed_steps(step, a.steps, a.warmup, torch.cuda.synchronize)
    if rank == 0:
        print(json.dumps({"what": f"GPT-{a.model} MHLA LM training step, seq {a.seq}, minimal host, bf16 autocast, AdamW",
                          "n_gpus": world, "per_gpu_batch": a.batch, "ms_per_step": el / a.steps * 1e3,
                          "tokens_per_s": a.batch * a.seq * world / (el / a.steps),
                          "params_M": sum(p.numel() for p in model.parameters()) / 1e6}))


if __name__ == "__main__":
    main()
