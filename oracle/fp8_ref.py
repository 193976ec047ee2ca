"""CPU oracle of the fp8 weight mode: row-wise OCP e4m3fn quantisation with power-of-two scales.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference has no fp8 arithmetic: its reduced-byte surface is the weight-only `dtype` switch of
mingunivision/mingunivisioninfer.py:46-70 (int8 through optimum-quanto, int4 through bitsandbytes; both third-party and
absent from this image), and the only fp8 trace is vllm/ming_lite.patch:171-190.  The format restated here is the published
OCP 8-bit floating point specification (OFP8 rev 1.0), e4m3 variant: 1 sign, 4 exponent bits (bias 7), 3 mantissa bits,
subnormals, no infinities, S.1111.111 = NaN, max finite 448 — `decode_e4m3_table` / `encode_e4m3_nearest_even` below are
that definition in plain numpy; tests/test_fp8_oracle.py pins torch's float8_e4m3fn casts (what `quantize_rows` uses for speed)
to them on every byte and on round-to-nearest-even ties.  PARITY PIN: there is no reference output to pin against — the
fp8 model is defined by this file: weights W -> dequantize_rows(*quantize_rows(W)), everything else as the bf16 model; the
HIP path is held to the oracle run on those weights (tests/test_gpu_fp8.py).
"""
import numpy as np
import torch

E4M3_MAX = 448.0


def decode_e4m3_table():
    """All 256 byte values of OCP e4m3fn as float64 (NaN at 0x7f / 0xff), straight from the format definition."""
    out = np.empty(256, dtype=np.float64)
    for b in range(256):
        s = -1.0 if b & 0x80 else 1.0
        e, m = (b >> 3) & 0xF, b & 0x7
        if e == 0xF and m == 0x7:
            out[b] = np.nan
        elif e == 0:
            out[b] = s * (m / 8.0) * 2.0 ** (1 - 7)
        else:
            out[b] = s * (1.0 + m / 8.0) * 2.0 ** (e - 7)
    return out


def encode_e4m3_nearest_even(x):
    """float array (|x| <= 448) -> bytes: the nearest e4m3 value, ties to the even mantissa, by search over the 127
    non-negative finite codes (small inputs only: this is the definition, not the fast path)."""
    tab = decode_e4m3_table()[:127]                      # codes 0x00..0x7e ascend with the value
    x = np.asarray(x, dtype=np.float64)
    a = np.abs(x)
    hi = np.searchsorted(tab, a, side="left").clip(0, 126)
    lo = (hi - 1).clip(0, 126)
    d_lo, d_hi = np.abs(a - tab[lo]), np.abs(tab[hi] - a)
    pick_hi = (d_hi < d_lo) | ((d_hi == d_lo) & (hi % 2 == 0))
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    return np.where(np.signbit(x), code | 0x80, code).astype(np.uint8)


def pow2_row_scale(w):
    """fp32 [..., N]: s = 2^e with amax / s in (224, 448]; 1 for an all-zero row.  amax = ma * 2^ea (1 <= ma < 2), 448 = 1.75 * 2^8
    -> e = ea - 8, + 1 when ma > 1.75."""
    amax = w.float().abs().amax(dim=-1)
    m, e = torch.frexp(amax)                              # amax = m * 2^e, 0.5 <= m < 1
    es = (e - 1) - 8 + (2.0 * m > 1.75).to(e.dtype)
    es = es.clamp(-126, 127)
    s = torch.ldexp(torch.ones_like(amax), es)
    return torch.where(amax == 0, torch.ones_like(s), s)


def quantize_rows(w):
    """W [..., N, K] (bf16 values) -> (uint8 [..., N, K], fp32 scale [..., N]):  q = e4m3_rne(W / s), s = pow2_row_scale."""
    s = pow2_row_scale(w)
    q = (w.float() / s.unsqueeze(-1)).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), s


def dequantize_rows(q, s):
    """-> fp32 [..., N, K] = e4m3(q) * s: the weights of the fp8 model (exactly representable in bf16)."""
    return q.view(torch.float8_e4m3fn).float() * s.unsqueeze(-1)


def fake_quant_rows(w):
    """W -> the fp8 model's weight values, fp32."""
    return dequantize_rows(*quantize_rows(w))
