"""Seeded shape fuzzing of the tap-GEMM through the op-level C-ABI seams (linear, Conv1d, ConvTranspose1d, attention):
every tile variant (128/64-row tiles, 128/64/32-column tiles, 64/128-byte k-tiles), ragged tails and odd sizes against
float64 torch references."""
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _cases_linear():
    rng = random.Random(1234)
    out = []
    for _ in range(28):
        M = rng.choice([1, 7, 63, 64, 65, 129, 500, 1728, 4099, 20000])
        N = rng.choice([1, 18, 24, 33, 64, 80, 96, 130, 384, 515, 1152])
        K = rng.choice([8, 32, 64, 72, 200, 384, 520, 1024, 1536])
        out.append((M, N, K, rng.choice(["f16", "f32"])))
    return out


@pytest.mark.parametrize("M,N,K,dtype", _cases_linear())
def test_linear_fuzz(M, N, K, dtype):
    from seedvc_amd import ops
    g = torch.Generator().manual_seed(M * 31 + N * 7 + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    if dtype == "f16":
        ref = (a.half().double() @ w.half().double().t() + b.double()).float()
        tol = 2e-4
    else:
        ref = (a.double() @ w.double().t() + b.double()).float()
        tol = 2e-5
    y = ops.linear(a.cuda(), w.cuda(), b.cuda(), dtype=dtype).cpu()
    assert y.shape == ref.shape
    assert (y - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())


def _cases_conv():
    rng = random.Random(4321)
    out = []
    for _ in range(24):
        k = rng.choice([1, 3, 5, 7, 11, 16])
        dil = rng.choice([1, 1, 3, 5]) if k > 1 else 1
        stride = rng.choice([1, 1, 1, 2, 8]) if k > 1 else 1
        Cin = rng.choice([1, 18, 24, 48, 64, 80, 100, 256])
        Cout = rng.choice([1, 18, 32, 48, 96, 128, 200])
        L = rng.choice([1, 2, 9, 40, 129, 600])
        B = rng.choice([1, 2, 3])
        pad_left = rng.choice([0, (k - 1) * dil // 2, (k - 1) * dil])
        out.append((k, dil, stride, pad_left, Cin, Cout, L, B, rng.choice(["f32", "f16x3"])))
    return out


@pytest.mark.parametrize("k,dil,stride,pad_left,Cin,Cout,L,B,dtype", _cases_conv())
def test_conv1d_fuzz(k, dil, stride, pad_left, Cin, Cout, L, B, dtype):
    from seedvc_amd import ops
    span = (k - 1) * dil
    pad_right = max(0, span - pad_left)
    Lout = (L + pad_left + pad_right - span - 1) // stride + 1
    if Lout < 1:
        pytest.skip("empty output")
    g = torch.Generator().manual_seed(k * 131 + Cin * 17 + L)
    x = torch.randn(B, Cin, L, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv1d(F.pad(x.double(), (pad_left, pad_right)), w.double(), b.double(), stride=stride, dilation=dil).float()
    assert ref.shape[-1] == Lout
    y = ops.conv1d_cl(x.transpose(1, 2).contiguous().cuda(), w.cuda(), b.cuda(), dilation=dil, stride=stride,
                      pad_left=pad_left, Lout=Lout, dtype=dtype).cpu().transpose(1, 2)
    assert y.shape == ref.shape
    assert (y - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("N,T,H,lens", [(1, 1, 1, None), (2, 63, 2, [63, 5]), (3, 130, 3, [130, 1, 77]), (1, 257, 6, None),
                                        (2, 864, 2, [864, 431])])
def test_attention_fuzz(N, T, H, lens):
    from seedvc_amd import ops
    g = torch.Generator().manual_seed(N * 1000 + T)
    q, k, v = (torch.randn(N, T, H, 64, generator=g) for _ in range(3))
    y = ops.attention(q.cuda(), k.cuda(), v.cuda(), kv_lens=lens).cpu()
    qh, kh, vh = (t.half().double().permute(0, 2, 1, 3) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / 8.0
    if lens is not None:
        mask = torch.arange(T)[None, :] >= torch.tensor(lens)[:, None]
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, dim=-1) @ vh).permute(0, 2, 1, 3).float()
    assert (y - ref).abs().max().item() < 4e-3
