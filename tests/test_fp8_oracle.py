"""CPU tests of the fp8 weight mode's oracle (oracle/fp8_ref.py): torch's float8_e4m3fn casts, which the oracle uses, are held to
the OCP e4m3 definition restated in numpy; the power-of-two row scales keep the dequantised weights exact in bf16."""
import numpy as np
import torch

from oracle import fp8_ref


def test_decode_all_bytes_matches_ocp_definition():
    tab = fp8_ref.decode_e4m3_table()
    got = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).double().numpy()
    nan = np.isnan(tab)
    assert nan.sum() == 2 and nan[0x7F] and nan[0xFF] and np.isnan(got[nan]).all()
    assert np.array_equal(tab[~nan], got[~nan])
    assert tab[0x7E] == 448.0 and tab[0x01] == 2.0 ** -9 and tab[0x08] == 2.0 ** -6


def test_encode_round_to_nearest_even_matches_ocp_definition():
    tab = fp8_ref.decode_e4m3_table()[:127]
    g = torch.Generator().manual_seed(0)
    mids = (tab[:-1] + tab[1:]) / 2                                   # every tie between neighbours, both signs
    xs = np.concatenate([tab, mids, -tab, -mids, (torch.randn(20000, generator=g).double() * 60).clamp(-448, 448).numpy(),
                         (torch.randn(5000, generator=g).double() * 0.01).numpy()])
    xs = xs.astype(np.float32)                                        # the cast under test takes fp32
    want = fp8_ref.encode_e4m3_nearest_even(xs)
    got = torch.from_numpy(xs).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    zero = xs == 0                                                    # +-0: sign is irrelevant for a weight
    assert np.array_equal(want[~zero], got[~zero]), np.flatnonzero(want != got)[:10]


def test_row_scales_are_powers_of_two_and_dequant_is_exact_in_bf16():
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(64, 256, generator=g) * torch.logspace(-6, 3, 64).unsqueeze(1)).to(torch.bfloat16)
    w[3] = 0
    w[5, :] = 0; w[5, 7] = 448.0                                      # amax exactly on the e4m3 maximum
    w[6, :] = 0; w[6, 9] = 1.75 * 2.0 ** -20                          # ma == 1.75 exactly: no extra binade
    w[7, :] = 0; w[7, 9] = 1.7578125 * 2.0 ** 5                       # just above 1.75: one more
    q, s = fp8_ref.quantize_rows(w)
    m, _ = torch.frexp(s)
    assert (m == 0.5).all() and s[3] == 1.0
    amax = w.float().abs().amax(-1)
    nz = amax > 0
    r = amax[nz] / s[nz]
    assert (r > 224).all() and (r <= 448).all()
    assert s[5] == 1.0 and s[6] == 2.0 ** -28 and s[7] == 2.0 ** -2
    dq = fp8_ref.dequantize_rows(q, s)
    assert torch.equal(dq, dq.to(torch.bfloat16).float())             # a bf16 model of its own right
    rel = ((dq - w.float()).abs().amax(-1)[nz] / amax[nz])
    assert float(rel.max()) <= 2.0 ** -4 + 1e-7                        # half an ulp of a 3-bit mantissa at the top binade
