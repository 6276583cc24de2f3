#!/usr/bin/env python
"""Per-kernel HIP-event timings (mhla_prof_* hook) for the non-C2 shapes."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd
from mhla_amd import block_distance_weights, block_index_3d, causal_mixing_init
lib = mhla_amd._lib.load()
DEV = "cuda"

def prof(fn, name, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); lib.mhla_prof_enable(1)
    for _ in range(iters): fn()
    torch.cuda.synchronize(); lib.mhla_prof_enable(0)
    buf = ctypes.create_string_buffer(1 << 16); lib.mhla_prof_report(buf, len(buf))
    tot = 0
    print("==", name)
    for line in buf.value.decode().splitlines():
        k, c, t = line.rsplit(" ", 2); us = float(t) / iters * 1e3; tot += us
        print(f"   {k:22s} {int(c)//iters} x {float(t)/int(c)*1e3:8.1f} us")
    print(f"   total {tot:.0f} us")

g = torch.Generator().manual_seed(1)
B, T, H, K, V = 4, 8192, 4, 128, 256
q = torch.randn(B, T, H, K, generator=g).bfloat16().to(DEV).requires_grad_(True)
k = torch.randn(B, T, H, K, generator=g).bfloat16().to(DEV).requires_grad_(True)
v = torch.randn(B, T, H, V, generator=g).bfloat16().to(DEV).requires_grad_(True)
do = torch.randn(B, T, H, V, generator=g).bfloat16().to(DEV)
n = T // 64
mix = causal_mixing_init(n).reshape(n, n).to(DEV).requires_grad_(True)
def c5():
    o = mhla_amd.mhla_causal(q, k, v, mix); o.backward(do); q.grad = k.grad = v.grad = mix.grad = None
prof(c5, "C5 causal fwd+bwd")
B, N, H, D, M = 1, 31500, 12, 128, 150
mk = lambda: torch.randn(B, N, H, D, generator=g).to(DEV)
qq, kk, vv, q2, k2 = mk().abs(), mk().abs(), mk(), mk().abs(), mk().abs()
W = block_distance_weights((3, 5, 10), "linear").to(DEV)
idx = block_index_3d((21, 30, 50), (3, 5, 10)).to(DEV)
prof(lambda: mhla_amd.mhla_blockmix(qq, kk, vv, W, q_den=q2, k_den=k2, block_index=idx), "C4 Wan fwd normalised")
qs = [t.clone().requires_grad_(True) for t in (qq, kk, vv, q2, k2)]
Wg = W.clone().requires_grad_(True)
dd = mk()
def c4b():
    o = mhla_amd.mhla_blockmix(qs[0], qs[1], qs[2], Wg, q_den=qs[3], k_den=qs[4], block_index=idx); o.backward(dd)
    for t in qs + [Wg]: t.grad = None
prof(c4b, "C4 Wan fwd+bwd normalised")
B, N, H, D, M = 16, 1024, 16, 72, 16
mkb = lambda: torch.randn(B, N, H, D, generator=g).abs().bfloat16().to(DEV).requires_grad_(True)
xq, xk, xv = mkb(), mkb(), mkb()
xo = torch.randn(B, N, H, D, generator=g).bfloat16().to(DEV)
W2 = block_distance_weights((4, 4), "linear").to(DEV).requires_grad_(True)
def xl():
    o = mhla_amd.mhla_blockmix(xq, xk, xv, W2); o.backward(xo)
    for t in (xq, xk, xv, W2): t.grad = None
prof(xl, "DiT-XL/2 512x512 op bf16 D=72 fwd+bwd")
