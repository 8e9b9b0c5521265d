"""GPU parity of the vocoders (HIP path through the C ABI) against the committed reference outputs and the
CPU oracle.  Tolerance: north_star's waveform RMS < 1e-4 (fp32-MFMA precision mode)."""
import pytest
import torch
import torch.nn.functional as F

import cases
import seedvc_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

WAVE_RMS = 1e-4


def _rms(a, b):
    return (a - b).pow(2).mean().sqrt().item()


@pytest.mark.parametrize("k,dil,stride,pad_left,Cin,Cout,L", [(3, 1, 1, 1, 20, 24, 50), (7, 3, 1, 9, 96, 48, 301),
                                                                (11, 5, 1, 25, 32, 96, 200), (16, 1, 8, 4, 18, 40, 129),
                                                                (1, 1, 1, 0, 18, 128, 65)])
@pytest.mark.parametrize("dtype", ["f32", "f16", "f16x3"])
def test_conv1d_channels_last(k, dil, stride, pad_left, Cin, Cout, L, dtype):
    from seedvc_amd import ops
    g = torch.Generator().manual_seed(k * 31 + L)
    B = 2
    x = torch.randn(B, Cin, L, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    if dtype == "f16":
        ref = F.conv1d(x.half().float(), w.half().float(), b, stride=stride, dilation=dil, padding=pad_left)
    else:
        ref = F.conv1d(x.double(), w.double(), b.double(), stride=stride, dilation=dil, padding=pad_left).float()
    Lout = ref.shape[-1]
    y = ops.conv1d_cl(x.transpose(1, 2).contiguous().cuda(), w.cuda(), b.cuda(), dilation=dil, stride=stride,
                      pad_left=pad_left, Lout=Lout, dtype=dtype).cpu().transpose(1, 2)
    assert (y - ref).abs().max().item() < (3e-5 if dtype == "f16" else 1e-5)      # fp16x3 is held to the fp32 bound


def test_conv1d_reflect_pad():
    from seedvc_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 64, 40, generator=g)
    w = torch.randn(32, 64, 5, generator=g) / 18.0
    ref = F.conv1d(F.pad(x, (2, 2), mode="reflect").double(), w.double()).float()
    y = ops.conv1d_cl(x.transpose(1, 2).contiguous().cuda(), w.cuda(), None, pad_left=2, Lout=40, pad_mode=1,
                      dtype="f32").cpu().transpose(1, 2)
    assert (y - ref).abs().max().item() < 3e-6


@pytest.mark.parametrize("dtype", ["f32", "f16x3"])
@pytest.mark.parametrize("s,Cin,Cout,L", [(2, 48, 24, 77), (4, 64, 32, 40), (8, 512, 256, 12)])
def test_conv_transpose1d(s, Cin, Cout, L, dtype):
    from seedvc_amd import ops
    g = torch.Generator().manual_seed(s)
    x = torch.randn(2, Cin, L, generator=g)
    w = torch.randn(Cin, Cout, 2 * s, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv_transpose1d(x.double(), w.double(), b.double(), stride=s, padding=s // 2).float()
    y = ops.conv_transpose1d_cl(x.transpose(1, 2).contiguous().cuda(), w.cuda(), b.cuda(), s, dtype=dtype).cpu().transpose(1, 2)
    assert y.shape == ref.shape
    assert (y - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("precision", ["fp32", "fp16x3", "fp16p8"])
@pytest.mark.parametrize("name", list(cases.BIGVGAN_CASES))
def test_bigvgan_vs_reference_golden(name, precision, golden):
    from seedvc_amd.vocoder import BigVGAN
    h, sd, mel, meta = cases.bigvgan_case(name)
    voc = BigVGAN(h, sd, "cuda:0", precision=precision)
    y = voc(mel.cuda()).cpu()
    ref = torch.from_numpy(golden[name + ".wave"])
    assert y.shape == ref.shape
    rms = _rms(y, ref)
    print(f"{name} [{precision}]: waveform RMS vs reference {rms:.3e} (signal rms {ref.pow(2).mean().sqrt():.3f})")
    assert rms < WAVE_RMS


def test_bigvgan_fp16_mode_reports_error(golden):
    from seedvc_amd.vocoder import BigVGAN
    h, sd, mel, meta = cases.bigvgan_case("bigvgan_r")
    y = BigVGAN(h, sd, "cuda:0", precision="fp16")(mel.cuda()).cpu()
    rms = _rms(y, torch.from_numpy(golden["bigvgan_r.wave"]))
    print(f"bigvgan_r fp16-operand mode: waveform RMS {rms:.3e}")
    assert rms < 2e-2


@pytest.mark.parametrize("precision", ["fp32", "fp16x3", "fp16p8"])
@pytest.mark.parametrize("name", list(cases.HIFT_CASES))
def test_hift_vs_reference_golden(name, precision, golden):
    from seedvc_amd.vocoder import HiFT
    c, sd, mel, phase0, noise, meta = cases.hift_case(name)
    voc = HiFT(c, sd, "cuda:0", precision=precision)
    # (1) f0 predictor
    y, f0 = voc(mel.cuda(), phase0=phase0.cuda(), noise=noise.cuda(), return_f0=True)
    f0_ref = torch.from_numpy(golden[name + ".f0"])
    rel = ((f0.cpu() - f0_ref).abs() / f0_ref.abs().clamp_min(1.0)).max().item()
    print(f"{name}: f0 max rel err {rel:.3e}")
    assert rel < 2e-5
    # (2) decoder with the f0 path pinned (phase integrates f0 over the utterance: chaotic in f0)
    for key, fkey in ((".wave_f0fixed", ".f0_fixed"), (".wave", ".f0")):
        f0_in = torch.from_numpy(golden[name + fkey])
        y = voc(mel.cuda(), f0=f0_in.cuda(), phase0=phase0.cuda(), noise=noise.cuda()).cpu()
        ref = torch.from_numpy(golden[name + key])
        rms = _rms(y, ref)
        print(f"{name}{key}: waveform RMS vs reference {rms:.3e} (signal rms {ref.pow(2).mean().sqrt():.3f})")
        assert rms < WAVE_RMS
    # (3) full path (own f0): short clips keep the phase drift small
    y = voc(mel.cuda(), phase0=phase0.cuda(), noise=noise.cuda()).cpu()
    rms = _rms(y, torch.from_numpy(golden[name + ".wave"]))
    print(f"{name}: full-path waveform RMS {rms:.3e}")
    assert rms < 1e-3


def test_vocoder_batch_equals_single():
    from seedvc_amd.vocoder import BigVGAN
    h, sd, mel, meta = cases.bigvgan_case("bigvgan_r")
    voc = BigVGAN(h, sd, "cuda:0")
    mel18 = torch.cat([mel * (1.0 - 0.05 * i) for i in range(9)], 0)       # B = 18 > the micro-batch set below
    assert mel18.shape[0] > 16
    voc.set_microbatch(16)                                                 # (default 32): a full group and a remainder group
    y = voc(mel18.cuda()).cpu()
    for b in (4, 17):                                                      # first group and the remainder group
        y0 = voc(mel18[b:b + 1].cuda()).cpu()
        assert torch.equal(y[b:b + 1], y0)
    voc.set_microbatch(5)                                                  # the grouping never changes a result
    assert torch.equal(voc(mel18.cuda()).cpu(), y)
