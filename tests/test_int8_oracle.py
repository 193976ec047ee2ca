"""CPU tests of the int8 weight mode's oracle (oracle/int8_ref.py): power-of-two row scales, round-to-nearest-even, exactness of
the dequantised weights in bf16, and what the power-of-two scale costs against optimum-quanto's amax / 127 rule."""
import torch

from oracle import int8_ref


def test_row_scales_quantisation_and_exactness_in_bf16():
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(64, 256, generator=g) * torch.logspace(-6, 3, 64).unsqueeze(1)).to(torch.bfloat16)
    w[3] = 0
    w[5, :] = 0; w[5, 7] = 127.0                                      # amax exactly on the int8 maximum
    w[6, :] = 0; w[6, 9] = 1.984375 * 2.0 ** -20                      # ma == 127 / 64 exactly: no extra binade
    w[7, :] = 0; w[7, 9] = 1.9921875 * 2.0 ** 5                       # just above: one more
    w[8, :] = 0; w[8, :4] = torch.tensor([0.5, 1.5, 2.5, 127.0])      # ties: half to even
    q, s = int8_ref.quantize_rows(w)
    m, _ = torch.frexp(s)
    assert (m == 0.5).all() and s[3] == 1.0
    amax = w.float().abs().amax(-1)
    nz = amax > 0
    r = amax[nz] / s[nz]
    assert (r > 63.5).all() and (r <= 127).all()
    assert s[5] == 1.0 and s[6] == 2.0 ** -26 and s[7] == 2.0 ** 0 and s[8] == 1.0
    qi = q.view(torch.int8)
    assert int(qi.min()) >= -127 and int(qi.max()) <= 127
    assert qi[8, :4].tolist() == [0, 2, 2, 127]
    dq = int8_ref.dequantize_rows(q, s)
    assert torch.equal(dq, dq.to(torch.bfloat16).float())             # a bf16 model of its own right
    err = (dq - w.float()).abs().amax(-1)[nz]
    assert bool((err <= 0.5 * s[nz]).all())                           # half a step


def test_cost_of_the_power_of_two_scale_against_quanto():
    """The rounded-up scale at most doubles the step: rms error <= 2x that of amax / 127 scales; on Gaussian rows (amax ~ 4 sigma) the
    int8 form is about twice as close to the bf16 weights as e4m3 is (1.4 % vs 2.7 % of the rms weight; quanto's scale: 0.8 %)."""
    from oracle import fp8_ref
    g = torch.Generator().manual_seed(2)
    w = (torch.randn(512, 2048, generator=g) * 0.02).to(torch.bfloat16)
    ours = int8_ref.fake_quant_rows(w)
    quanto = int8_ref.quanto_rows(w)
    e4m3 = fp8_ref.fake_quant_rows(w)
    rms = lambda a: float(((a - w.float()) ** 2).mean().sqrt())
    assert rms(ours) <= 2.0 * rms(quanto) + 1e-12
    assert rms(ours) < 0.7 * rms(e4m3)
    print("rms error / rms weight: int8 pow2 %.4f, quanto %.4f, e4m3 %.4f" % tuple(rms(x) / float(w.float().pow(2).mean().sqrt()) for x in (ours, quanto, e4m3)))
