"""CPU tests of the int8 weight mode's oracle (oracle/int8_ref.py): optimum-quanto's qint8 rule restated — bf16 scale = amax / 127,
bf16 quotient, round half to even, clamp to [-128, 127], bf16 product."""
import torch

from oracle import int8_ref


def test_quanto_rule_scales_rounding_and_bf16_products():
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(64, 256, generator=g) * torch.logspace(-6, 3, 64).unsqueeze(1)).to(torch.bfloat16)
    w[3] = 0
    w[5, :] = 0; w[5, 7] = 127.0                                      # amax 127 -> scale exactly 1
    w[8, :] = 0; w[8, :6] = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, 127.0])      # ties: half to even
    w[9, :] = 0; w[9, 0] = 254.0; w[9, 1] = 1.0                       # scale 2: 1 / 2 = 0.5 -> 0
    q, s = int8_ref.quantize_rows(w)
    assert torch.equal(s, s.to(torch.bfloat16).float())               # the scale is a bf16 value
    assert s[3] == 1.0 and s[5] == 1.0 and s[9] == 2.0
    amax = w.float().abs().amax(-1)
    nz = amax > 0
    assert bool(((amax[nz] / 127.0 - s[nz]).abs() <= 2.0 ** -8 * s[nz]).all())        # bf16(amax / 127)
    qi = q.view(torch.int8)
    assert int(qi.min()) >= -128 and int(qi.max()) <= 127
    assert qi[8, :6].tolist() == [0, 2, 2, 0, -2, 127] and qi[9, :2].tolist() == [127, 0]
    # the bf16 quotient can land on 127.5 / 128: a few weights of a row reach +-128 before the clamp; 128 itself is clamped to 127
    dq = int8_ref.dequantize_rows(q, s)
    assert torch.equal(dq, dq.to(torch.bfloat16).float())             # the int8 model is a bf16 model
    err = (dq - w.float()).abs().amax(-1)[nz]
    assert bool((err <= 1.01 * s[nz]).all())                          # half a step + the two bf16 roundings (quotient, product)
    # idempotence is NOT a property (re-quantising W' may pick another scale): every tensor is quantised once, from the checkpoint
    w3 = torch.randn(5, 64, 96, generator=g).to(torch.bfloat16)
    q3, s3 = int8_ref.quantize_rows(w3)
    assert q3.shape == (5, 64, 96) and s3.shape == (5, 64)


def test_error_of_the_quanto_rule_against_round_4s_power_of_two_scales():
    """What the change of rule does to the weights: quanto's amax / 127 scale uses the whole int8 range (rms error ~0.8 % of the rms
    weight on Gaussian rows against ~1.4 % for the rounded-up power of two) at the price of a per-element bf16 rounding."""
    from oracle import fp8_ref
    g = torch.Generator().manual_seed(2)
    w = (torch.randn(512, 2048, generator=g) * 0.02).to(torch.bfloat16)
    quanto = int8_ref.fake_quant_rows(w)
    pow2 = int8_ref.pow2_rows(w)
    e4m3 = fp8_ref.fake_quant_rows(w)
    rms = lambda a: float(((a - w.float()) ** 2).mean().sqrt())
    assert rms(quanto) < rms(pow2) < rms(e4m3)
    print("rms error / rms weight: int8 quanto %.4f, pow2 %.4f, e4m3 %.4f" % tuple(rms(x) / float(w.float().pow(2).mean().sqrt()) for x in (quanto, pow2, e4m3)))
