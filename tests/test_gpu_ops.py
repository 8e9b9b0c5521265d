"""GPU parity of the individual HIP kernels against plain torch fp32 references (through the C ABI)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def ops():
    from seedvc_amd import ops as o
    assert torch.cuda.is_available()
    return o


def _rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 384, 384), (1000, 1152, 512), (77, 80, 200), (5, 24, 40)])
@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_linear(ops, M, N, K, dtype):
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    if dtype == "f16":
        ref = F.linear(a.half().float(), w.half().float(), b)
        tol = 2e-5
    else:
        ref = F.linear(a.double(), w.double(), b.double()).float()
        tol = 2e-6
    out = ops.linear(a.cuda(), w.cuda(), b.cuda(), dtype=dtype).cpu()
    assert _rel(out, ref) < tol


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_linear_large_m(ops, dtype):
    """Large M (many row tiles, XCD-remapped order, tail tile)."""
    M, N, K = 45000, 384, 192
    g = torch.Generator().manual_seed(9)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    if dtype == "f16":
        ref = F.linear(a.half().float(), w.half().float(), b)
        tol = 2e-5
    else:
        ref = F.linear(a.double(), w.double(), b.double()).float()
        tol = 2e-6
    out = ops.linear(a.cuda(), w.cuda(), b.cuda(), dtype=dtype).cpu()
    assert _rel(out, ref) < tol


def test_linear_asymmetric_layout(ops):
    # A = I with an asymmetric W catches a transposed accumulator map
    n = 128
    a = torch.eye(n)
    w = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251)
    out = ops.linear(a.cuda(), w.cuda(), None, dtype="f32").cpu()
    assert torch.equal(out, w.t().contiguous())


@pytest.fixture
def attn_form(request, monkeypatch):
    """SVC_ATTN32 is read per launch: "0" (the default) = the 16x16x32 kernels, "1" = the 32x32x16 kernels (attn32_kernel,
    measured slower, kept selectable); both come as 64- and 128-query blocks chosen by grid size."""
    monkeypatch.setenv("SVC_ATTN32", request.param)
    return request.param


@pytest.mark.parametrize("attn_form", ["0", "1"], indirect=True)
@pytest.mark.parametrize("N,T,H,lens", [(2, 70, 2, None), (1, 862, 6, None), (3, 200, 3, [200, 130, 1]), (1, 64, 1, [64]),
                                        (2, 131, 2, [131, 67]), (1, 33, 1, [17])])
def test_attention(ops, N, T, H, lens, attn_form):
    g = torch.Generator().manual_seed(T)
    q, k, v = (torch.randn(N, T, H, 64, generator=g) for _ in range(3))
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), lens).cpu()
    qh, kh, vh = (t.half().float().permute(0, 2, 1, 3) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / 8.0
    if lens is not None:
        mask = torch.arange(T)[None, :] < torch.tensor(lens)[:, None]
        s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3)
    assert (out - ref).abs().max().item() < 4e-3      # fp16 q/k/v/p operands, fp32 statistics


@pytest.mark.parametrize("attn_form", ["0", "1"], indirect=True)
def test_attention_spike_forces_rescale(ops, attn_form):
    # one key dominates late in the sequence: exercises the online-softmax rescale path
    N, T, H = 1, 300, 1
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(N, T, H, 64, generator=g) for _ in range(3))
    k[0, 250] = q[0, 3] * 4.0
    out = ops.attention(q.cuda(), k.cuda(), v.cuda()).cpu()
    qh, kh, vh = (t.half().float().permute(0, 2, 1, 3) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, -1) @ vh).permute(0, 2, 1, 3)
    assert (out - ref).abs().max().item() < 4e-3


@pytest.mark.parametrize("N,K", [(512, 512), (512, 1536), (1024, 2560), (3072, 512), (384, 384)])
def test_linear_tile_forms_bit_identical(ops, N, K):
    """Small-M launches take 64x64 / 64x128 four-wave tiles, large ones 128x128: the K order of an output element is the
    same in every form, so a row computed alone equals the same row computed inside a large batch bit for bit."""
    g = torch.Generator().manual_seed(N + K)
    a = torch.randn(40000, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    big = ops.linear(a.cuda(), w.cuda(), None, dtype="f16")
    for m in (1720, 900, 70):
        small = ops.linear(a[:m].cuda(), w.cuda(), None, dtype="f16")
        assert torch.equal(small, big[:m]), (N, K, m)


@pytest.mark.parametrize("M,N,K,epi", [(3000, 768, 768, 0), (3000, 4096, 768, 1), (3000, 2304, 768, 3), (2100, 3072, 512, 1),
                                       (2100, 1536, 512, 3), (1300, 768, 2048, 0), (1300, 1024, 2560, 2), (500, 2304, 768, 3),
                                       (257, 256, 64, 0), (1000, 512, 192, 0)])
def test_tap_gemm_tile_forms_bit_identical_with_epilogues(ops, M, N, K, epi):
    """Every tile form of the tap-GEMM (128x128 in its ring variants, 256x128 with 8 or 4 waves -- the default of the D = 768
    SwiGLU / QKV launches --, 64x64 / 64x128 small-M tiles and the 256x256 AGPR-accumulator harness form) gives the same
    bits through the DiT's epilogues: bias + fp32 residual store, SwiGLU, tanh-sigmoid, QKV + RoPE + transposed V."""
    import ctypes as C
    from seedvc_amd import _lib
    L = _lib.lib()
    for form in (0x10, 0x20, 0x80, 0xB0, 0x90, 0x50, 0x70, 0x00):
        n = C.c_longlong(-1)
        _lib.check(L.svc_op_gemm_forms_diff(M, N, K, epi, 0x40, form, C.byref(n), None))
        assert n.value == 0, (hex(form), n.value)


def test_attention_block_forms_bit_identical(ops, monkeypatch):
    """64-query blocks (small grids) and 128-query blocks of the 16x16x32 kernel give the same bits, also across baseline
    moves.  (Large grids take the 32x32x16 kernel by default: see test_attention32_*.)"""
    monkeypatch.setenv("SVC_ATTN32", "0")
    T, H = 862, 6
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(1, T, H, 64, generator=g) for _ in range(3))
    k[0, 700, 2] = q[0, 5, 2] * 4.0                     # a late dominant key: forces the deferred rescale
    one = ops.attention(q.cuda(), k.cuda(), v.cuda()).cpu()                         # 7 x 6 blocks of 128 -> 64-query form
    rep = [t.repeat(8, 1, 1, 1) for t in (q, k, v)]
    many = ops.attention(rep[0].cuda(), rep[1].cuda(), rep[2].cuda()).cpu()         # 336 blocks -> 128-query form
    for n in (0, 7):
        assert torch.equal(many[n], one[0])


def test_attention32_independent_of_batch_and_neighbours(ops, monkeypatch):
    """attn32_kernel (SVC_ATTN32=1).  A query's arithmetic depends on its own keys only (the baseline
    moves per query) and a wave's arithmetic is the same in the 64- and 128-query block forms, so the same sequence gives
    the same bits alone (42 blocks -> 64-query form) and anywhere in a batch of 8 (336 blocks -> 128-query form), whatever
    its neighbours are; against the 16x16x32 kernel it agrees to fp16-operand rounding."""
    T, H = 862, 6
    g = torch.Generator().manual_seed(12)
    q, k, v = (torch.randn(1, T, H, 64, generator=g) for _ in range(3))
    k[0, 700, 2] = q[0, 5, 2] * 4.0                     # a late dominant key: forces the deferred rescale for some queries
    others = [torch.randn(7, T, H, 64, generator=g) for _ in range(3)]
    batch = [torch.cat([o[:3], t, o[3:]]) for o, t in zip(others, (q, k, v))]           # position 3 of 8: 336 blocks
    monkeypatch.setenv("SVC_ATTN32", "1")
    many = ops.attention(batch[0].cuda(), batch[1].cuda(), batch[2].cuda()).cpu()
    rep = [t.repeat(8, 1, 1, 1) for t in (q, k, v)]
    same = ops.attention(rep[0].cuda(), rep[1].cuda(), rep[2].cuda()).cpu()
    assert torch.equal(many[3], same[0]) and torch.equal(same[7], same[0])
    alone = ops.attention(q.cuda(), k.cuda(), v.cuda()).cpu()                          # 84 blocks of 64 queries
    assert torch.equal(alone[0], same[0])
    monkeypatch.setenv("SVC_ATTN32", "0")
    old = ops.attention(rep[0].cuda(), rep[1].cuda(), rep[2].cuda()).cpu()
    assert (old[0] - same[0]).abs().max().item() < 2e-3
    qh, kh, vh = (t.half().float().permute(0, 2, 1, 3) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, -1) @ vh).permute(0, 2, 1, 3)
    assert (same[0] - ref[0]).abs().max().item() < 4e-3


@pytest.mark.parametrize("rows,D", [(10, 128), (33, 384), (7, 512), (5, 768), (3, 192)])
def test_rmsnorm(ops, rows, D):
    g = torch.Generator().manual_seed(D)
    x = torch.randn(rows, D, generator=g) * 3
    gamma = 1 + 0.1 * torch.randn(D, generator=g)
    w, b = torch.randn(D, generator=g), torch.randn(D, generator=g)
    n = x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-5) * gamma

    def relerr(out, ref):
        return ((out - ref).abs() / (ref.abs() + 1)).max().item()

    assert relerr(ops.rmsnorm(x.cuda(), gamma.cuda(), w.cuda(), b.cuda(), add_one=False).cpu(), n * w + b) < 2e-3
    assert relerr(ops.rmsnorm(x.cuda(), gamma.cuda(), w.cuda(), b.cuda(), add_one=True).cpu(), n * (1 + w) + b) < 2e-3
    assert relerr(ops.rmsnorm(x.cuda(), gamma.cuda()).cpu(), n) < 2e-3


@pytest.mark.parametrize("name", ["act_small", "act_edge", "act_mid"])
def test_anti_alias_activation_golden(ops, golden, name):
    import cases
    x, alpha, beta = cases.act_case(name)
    filt = torch.from_numpy(golden[name + ".filter"])
    y = ops.anti_alias_activation_forward(x.cuda(), filt.cuda(), filt.cuda(), alpha.cuda(), beta.cuda()).cpu()
    assert (y - torch.from_numpy(golden[name + ".snakebeta"])).abs().max().item() < 2e-6
    y = ops.anti_alias_activation_forward(x.cuda(), filt.cuda(), filt.cuda(), alpha.cuda(), alpha.cuda()).cpu()
    assert (y - torch.from_numpy(golden[name + ".snake"])).abs().max().item() < 2e-6


# L % 4 == 0 takes the vectorised kernel (aa_act_rows4_kernel), anything else the scalar one: both at block edges
@pytest.mark.parametrize("B,C,L", [(1, 1, 1), (2, 3, 1023), (1, 2, 1024), (1, 2, 1025), (1, 24, 5000), (1, 2, 4), (1, 3, 8), (2, 2, 12),
                                   (1, 2, 1020), (1, 2, 1028), (1, 3, 2048), (2, 3, 1720), (1, 2, 4100)])
@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
def test_anti_alias_activation_shapes(ops, B, C, L, dt):
    import seedvc_oracle as O
    from seedvc_amd import weights
    g = torch.Generator().manual_seed(L)
    x = (torch.randn(B, C, L, generator=g) * 2).to(dt)
    al, be = torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    filt = weights.make_tensor("x.filter", (1, 1, 12)).reshape(-1)
    ref = O.anti_alias_act(x.float(), filt, torch.exp(al), 1.0 / (torch.exp(be) + 1e-9))
    y = ops.anti_alias_activation_forward(x.cuda(), filt.cuda(), filt.cuda(), al.cuda(), be.cuda())
    assert y.dtype == dt
    tol = {torch.float32: 1e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dt]
    assert ((y.float().cpu() - ref).abs() / (ref.abs() + 1)).max().item() < tol
    if L % 4 == 0 and L > 4:
        # an unaligned view of the same data (storage offset 1 element) must take the scalar kernel and agree
        buf = torch.zeros(B * C * L + 1, dtype=dt, device="cuda")
        xv = buf[1:].view(B, C, L)
        xv.copy_(x)
        y2 = ops.anti_alias_activation_forward(xv, filt.cuda(), filt.cuda(), al.cuda(), be.cuda())
        assert ((y2.float() - y.float()).abs() / (y.float().abs() + 1)).max().item() < 2 * tol


def test_anti_alias_activation_large_arguments(ops):
    """|a * u| far beyond the polynomial's verified range (caller data is arbitrary at this seam): libm path."""
    import seedvc_oracle as O
    from seedvc_amd import weights
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 2, 64, generator=g) * 40.0
    al, be = torch.tensor([1.5, 2.0]), torch.tensor([0.2, -0.1])
    filt = weights.make_tensor("x.filter", (1, 1, 12)).reshape(-1)
    ref = O.anti_alias_act(x, filt, torch.exp(al), 1.0 / (torch.exp(be) + 1e-9))
    y = ops.anti_alias_activation_forward(x.cuda(), filt.cuda(), filt.cuda(), al.cuda(), be.cuda()).cpu()
    assert ((y - ref).abs() / (ref.abs() + 1)).max().item() < 2e-4      # fp32 argument rounding at |a u| ~ 1e3 dominates


def test_anti_alias_activation_empty(ops):
    x = torch.zeros(0, 3, 7, device="cuda")
    f = torch.ones(12, device="cuda") / 12
    y = ops.anti_alias_activation_forward(x, f, f, torch.zeros(3, device="cuda"), torch.zeros(3, device="cuda"))
    assert y.shape == x.shape
