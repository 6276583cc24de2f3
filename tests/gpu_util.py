"""Shared helpers for the -m gpu parity tests: seeded inputs and oracle calls."""
import torch

from conftest import rel_err, rms_ratio
from oracle import mhla_oracle as orc

DEV = "cuda"
# Tolerances, derived from the arithmetic rather than fitted to observations.  u = unit roundoff of the
# tensor dtype (round to nearest: |fl(x) - x| <= u |x|; bf16 has 8 significand bits: u = 2^-8; fp16 11: u = 2^-11).
#  * fp32 tensors: every product is exact (fp32 MFMA) or split into bf16 hi + lo parts (>= 16 significand bits, 2^-17 per
#    operand) with fp32 accumulation: 2e-4 of the tensor's maximum, five times under north_star's 1e-3.
#  * 16-bit tensors, DEFAULT arithmetic (round 5 for the block-mixing operator, round 4 for the causal one): the reference
#    computes in fp32 on the given tensors (mhla_dit/train.py:12-13: no autocast; naive.py:39) and so do the kernels -- every
#    intermediate that feeds a second contraction (block / chunk summaries, dP = dO / n, score tiles) keeps >= 16 significand
#    bits (fp32 summaries, bf16 hi + lo operands).  What is left is the ONE final rounding of a 16-bit result (<= u |x| PER
#    ELEMENT, which no implementation can avoid) plus north_star's 1e-3 for everything the kernels add:
#        max|got - want| <= (u + 1e-3) max|want|   and   max(|got - want| - u |want|) <= 1e-3 max|want|   (check() asserts both);
#    fp32-stored results of 16-bit problems (dW, dmix) get the 1e-3 alone (DW_TOL, CAUSAL_DMIX_TOL).
#  * 16-bit tensors, OPT-IN reduced precision (summaries="bf16": MHLA_FLAG_BF16_SUMMARIES / MHLA_CAUSAL_BF16_SUMMARIES): the
#    kernels keep K intermediate tiles as single bf16 values on their way through the matrix pipe -- K = 1 for outputs (the
#    block / chunk summary, or the score tile of the small-sequence path), K = 2 for gradients (additionally dP = dO / n, resp.
#    the dS / dP summaries).  An intermediate rounding perturbs the result by at most u times the magnitude of what it feeds,
#    i.e. <= u max|x| when nothing averages (contraction length 1: the S = 1, M <= 4 corner cases of the fuzz tests reach
#    0.8 u .. 1.7 u) and ~ u / sqrt(L) over a contraction of length L (BASELINE shapes, L >= 64: 0.2 u .. 0.8 u observed):
#        max|got - want| <= (1 + K) u max|want|     (TOL_BF16SUM = 2 u for outputs, GTOL_BF16SUM = 3 u for gradients).
UNIT_ROUNDOFF = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}
TOL = {torch.float32: 2e-4, torch.bfloat16: 2.0 ** -8 + 1e-3, torch.float16: 2.0 ** -11 + 1e-3}
GTOL = dict(TOL)
DW_TOL = {torch.float32: 2e-4, torch.bfloat16: 1e-3, torch.float16: 1e-3}
TOL_BF16SUM = {torch.float32: 2e-4, torch.bfloat16: 2 * 2.0 ** -8, torch.float16: 2 * 2.0 ** -11}
GTOL_BF16SUM = {torch.float32: 2e-4, torch.bfloat16: 3 * 2.0 ** -8, torch.float16: 3 * 2.0 ** -11}
CAUSAL_TOL = TOL
CAUSAL_DMIX_TOL = DW_TOL


def bm_tols(dtype, summaries="split"):
    """(output, token-gradient, dW) tolerances of the block-mixing operator for this dtype and arithmetic."""
    if summaries == "bf16" and dtype == torch.bfloat16:
        return TOL_BF16SUM[dtype], GTOL_BF16SUM[dtype], GTOL_BF16SUM[dtype]
    return TOL[dtype], GTOL[dtype], DW_TOL[dtype]


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
    # error beyond the final rounding of the result to its own dtype (|err| <= u |want| elementwise is what storing the exact
    # result in that dtype costs): what the kernels' internal arithmetic adds, normalised like rel_err
    u = UNIT_ROUNDOFF.get(got.dtype, 0.0)
    x = ((g.double() - w.double()).abs() - u * w.double().abs()).clamp_min(0).max().item() / max(w.double().abs().max().item(), 1e-30)
    OBSERVED.append((os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], name, str(got.dtype).replace("torch.", ""), e, r, tol, x))
    assert e < tol, f"{name}: rel_err {e:.3e} (rms ratio {r:.3e}) exceeds {tol:.1e}"
    if u and tol > u:
        assert x < tol - u, f"{name}: error beyond the final {got.dtype} rounding {x:.3e} exceeds {tol - u:.1e} (rel_err {e:.3e})"
    assert r < 2 * tol, f"{name}: rms ratio {r:.3e} exceeds {2 * tol:.1e} (rel_err {e:.3e})"
    return e


def poison():
    """Fill ~350 MB of device memory with NaN and free it again: the next torch.empty() calls of the ops (workspaces,
    outputs, gradients) start as NaN, and the registers of idle CUs hold NaN -- reads of memory or registers that were never
    written show up as NaN in the parity checks."""
    slabs = [torch.full((n,), float("nan"), dtype=torch.float32, device=DEV) for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20)]
    del slabs
