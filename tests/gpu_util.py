"""Shared helpers for the -m gpu parity tests: seeded inputs and oracle calls."""
import torch

from conftest import rel_err, rms_ratio
from oracle import mhla_oracle as orc

DEV = "cuda"
# north_star: outputs within 1e-3 rel-err (max|a-b| / max|b|) of the reference in fp32-accumulate, fp32-output mode; bf16 /
# fp16 OUTPUT adds one rounding of the result (2^-9 / 2^-11 elementwise).  The bounds below are about twice the largest
# error observed over the whole -m gpu suite in round 2 (profiles/r2_parity_errors.md: fp32 7.2e-5; bf16 outputs 5.8e-3 --
# one bf16 rounding of the largest output, nothing to tighten; bf16 gradients 7.7e-3; fp16 4.3e-4 / 6.4e-4).
TOL = {torch.float32: 2e-4, torch.bfloat16: 6e-3, torch.float16: 1e-3}
GTOL = {torch.float32: 2e-4, torch.bfloat16: 1.2e-2, torch.float16: 1.5e-3}


def make_blockmix_inputs(B, H, M, S, D, dtype, seed=1234, w="linear", split=False):
    g = torch.Generator().manual_seed(seed)
    N = M * S
    q = (torch.relu(torch.randn(B, N, H, D, generator=g)) + 1e-6).to(dtype)
    k = (torch.relu(torch.randn(B, N, H, D, generator=g)) + 1e-6).to(dtype)
    v = torch.randn(B, N, H, D, generator=g).to(dtype)
    do = torch.randn(B, N, H, D, generator=g).to(dtype)
    if w == "rand":
        W = torch.rand(M, M, generator=g)
    else:
        side = int(round(M ** 0.5))
        layout = (side, side) if side * side == M else (M,)
        W = orc.block_distance_weights(layout, "linear") if M > 1 else torch.ones(1, 1)
    qd = kd = None
    if split:   # numerator pair with signs (like roped q, k), positive denominator pair
        qd, kd = q, k
        q = (q.float() * torch.sign(torch.randn(B, N, H, D, generator=g))).to(dtype)
        k = (k.float() * torch.sign(torch.randn(B, N, H, D, generator=g))).to(dtype)
    return q, k, v, W, do, qd, kd


def oracle_blockmix(q, k, v, W, do, qd, kd, eps, normalize):
    f = lambda t: None if t is None else t.float()
    out = orc.blockmix_fwd(f(q), f(k), f(v), W, eps, f(qd), f(kd), normalize)
    grads = orc.blockmix_bwd(f(q), f(k), f(v), W, f(do), eps, f(qd), f(kd), normalize)
    return out, grads


def to_dev(*ts):
    return [None if t is None else t.to(DEV) for t in ts]


# every comparison of a parity run: (test id, name, dtype of the HIP result, max-normalised error, rms-relative error, tolerance);
# conftest writes the summary to gpurun_out/parity_report.json at the end of a -m gpu session (profiles/ keeps a copy per round)
OBSERVED = []


def check(name, got, want, tol, atol=0.0):
    """rel-err = max|got - want| / max|want| < tol (or max|got - want| < atol for ~zero references); the rms-relative error
    (which, unlike the max-normalised one, sees errors on small-magnitude elements) must stay below the same bound."""
    import os
    g, w = got.float().cpu(), want.float()
    if atol and (g - w).abs().max().item() < atol:
        return 0.0
    e, r = rel_err(g, w), rms_ratio(g, w)
    OBSERVED.append((os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], name, str(got.dtype).replace("torch.", ""), e, r, tol))
    assert e < tol, f"{name}: rel_err {e:.3e} (rms ratio {r:.3e}) exceeds {tol:.1e}"
    assert r < 2 * tol, f"{name}: rms ratio {r:.3e} exceeds {2 * tol:.1e} (rel_err {e:.3e})"
    return e


def poison():
    """Fill ~350 MB of device memory with NaN and free it again: the next torch.empty() calls of the ops (workspaces,
    outputs, gradients) start as NaN, and the registers of idle CUs hold NaN -- reads of memory or registers that were never
    written show up as NaN in the parity checks."""
    slabs = [torch.full((n,), float("nan"), dtype=torch.float32, device=DEV) for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20)]
    del slabs
