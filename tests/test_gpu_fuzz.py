"""GPU parity: seeded random shapes / options through the dispatcher (small-sequence, bf16 fast, split-operand and generic
kernel families; causal bf16 pipeline and generic) against the oracle.  Sizes keep the oracle at milliseconds per case."""
import random

import pytest
import torch

from gpu_util import DEV, GTOL, TOL, check
from oracle import mhla_oracle as orc
from test_gpu_blockmix import run_case
from test_gpu_causal import run_causal

pytestmark = pytest.mark.gpu

_DT = [torch.float32, torch.bfloat16, torch.float16]


def _blockmix_cases(n=48, seed=20261001):
    rnd = random.Random(seed)
    cases = []
    for i in range(n):
        dtype = rnd.choice(_DT)
        D = rnd.choice([4, 8, 12, 16, 24, 32, 40, 48, 56, 64, 64, 64, 72, 80, 96, 104, 128, 128])
        M = rnd.choice([1, 2, 3, 4, 5, 9, 16, 16, 25, 33, 64, 70])
        S = rnd.choice([1, 3, 7, 16, 16, 20, 33, 49, 64])
        if M * S * D > 400000:
            S = max(1, 400000 // (M * D))
        normalize = rnd.random() < 0.8
        split = normalize and rnd.random() < 0.35
        use_idx = rnd.random() < 0.3
        cases.append((i, dtype, M, S, D, normalize, split, use_idx, rnd.choice(["linear", "rand"])))
    return cases


@pytest.mark.parametrize("case", _blockmix_cases(), ids=lambda c: f"{c[0]}-{str(c[1]).split('.')[-1]}-M{c[2]}-S{c[3]}-D{c[4]}-n{int(c[5])}s{int(c[6])}i{int(c[7])}")
def test_blockmix_fuzz(case):
    i, dtype, M, S, D, normalize, split, use_idx, w = case
    idx = None
    if use_idx:
        idx = torch.randperm(M * S, generator=torch.Generator().manual_seed(i)).int()
    run_case(2 if M * S * D < 100000 else 1, 2, M, S, D, dtype, normalize=normalize, split=split, w=w, idx=idx, seed=1000 + i)


def _round4_cases(n=20, seed=4):
    """Shapes of the kernels added in round 4: many short blocks (wave-per-block token kernels, whole-matrix dW, LDS-DMA mixing,
    padded summary rows) and few long ones (chunk parts over several workgroups, pair prefetch of the summary kernels)."""
    rnd = random.Random(seed)
    cases = []
    for i in range(n):
        if i % 2 == 0:
            dtype = rnd.choice([torch.bfloat16, torch.bfloat16, torch.float32])
            M, S, D = rnd.choice([65, 100, 129, 193, 200, 256]), rnd.choice([8, 16, 16]), rnd.choice([32, 64, 64, 72])
        else:
            dtype = torch.bfloat16
            M, S, D = rnd.choice([3, 8, 17, 33]), rnd.choice([65, 128, 130, 192, 256, 320]), 64
        normalize = rnd.random() < 0.8
        cases.append((100 + i, dtype, M, S, D, normalize, False, rnd.random() < 0.3, "rand"))
    return cases


@pytest.mark.parametrize("case", _round4_cases(), ids=lambda c: f"{c[0]}-{str(c[1]).split('.')[-1]}-M{c[2]}-S{c[3]}-D{c[4]}-n{int(c[5])}i{int(c[7])}")
def test_blockmix_fuzz_many_short_and_few_long_blocks(case):
    i, dtype, M, S, D, normalize, split, use_idx, w = case
    idx = torch.randperm(M * S, generator=torch.Generator().manual_seed(i)).int() if use_idx else None
    run_case(2, 3, M, S, D, dtype, normalize=normalize, split=split, w=w, idx=idx, seed=2000 + i)


def _round5_cases(n=28, seed=5):
    """Shapes of the round-5 kernels: the resident mixing at two, four and eight waves with the fused dW and the normaliser slices (even
    and odd block lengths, a few and many (b, h) pairs), 24-bit summaries on 16-bit tensors (head dims 8 .. 96) and on fp32 tensors at
    head dim 128 with up to 192 blocks (k_sp_dw on the two planes above 128), the row dots from G (D <= 64), split q / k pairs, gather maps."""
    rnd = random.Random(seed)
    cases = []
    for i in range(n):
        if i % 4 == 3:
            dtype, D = torch.float32, rnd.choice([120, 128, 128])
            M, S = rnd.choice([33, 40, 100, 129, 150, 192]), rnd.choice([5, 6, 10, 14])
        else:
            dtype = rnd.choice([torch.bfloat16, torch.bfloat16, torch.float16])
            D = rnd.choice([8, 16, 24, 40, 56, 64, 64, 72, 80, 96])
            M = rnd.choice([2, 5, 16, 17, 31, 32, 33, 50, 64, 65, 100, 128])
            S = rnd.choice([2, 6, 7, 20, 30, 33, 64, 66, 130])
        if M * S * D > 500000:
            S = max(2, 500000 // (M * D))
        normalize = rnd.random() < 0.85
        split = normalize and rnd.random() < 0.3
        cases.append((200 + i, dtype, M, S, D, normalize, split, rnd.random() < 0.3, "rand", rnd.choice([1, 2, 5])))
    return cases


@pytest.mark.parametrize("case", _round5_cases(), ids=lambda c: f"{c[0]}-{str(c[1]).split('.')[-1]}-M{c[2]}-S{c[3]}-D{c[4]}-n{int(c[5])}s{int(c[6])}i{int(c[7])}-B{c[9]}")
def test_blockmix_fuzz_resident_mixing_and_24_bit_summaries(case):
    i, dtype, M, S, D, normalize, split, use_idx, w, B = case
    idx = torch.randperm(M * S, generator=torch.Generator().manual_seed(i)).int() if use_idx else None
    run_case(B, 3, M, S, D, dtype, normalize=normalize, split=split, w=w, idx=idx, seed=3000 + i)


def _causal_cases(n=16, seed=7):
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        dtype = rnd.choice(_DT)
        K = rnd.choice([8, 16, 32, 64, 64, 128, 192])
        V = rnd.choice([8, 16, 64, 64, 128, 256])
        T = rnd.choice([1, 17, 63, 64, 65, 130, 200, 333, 512])
        out.append((i, dtype, T, K, V))
    return out


@pytest.mark.parametrize("case", _causal_cases(), ids=lambda c: f"{c[0]}-{str(c[1]).split('.')[-1]}-T{c[2]}-K{c[3]}-V{c[4]}")
def test_causal_fuzz(case):
    i, dtype, T, K, V = case
    run_causal(2, T, 2, K, V, max(2, (T + 63) // 64 + (i % 3)), dtype, seed=500 + i)
