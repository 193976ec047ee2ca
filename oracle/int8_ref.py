"""CPU oracle of the int8 weight mode: row-wise symmetric int8 quantisation with power-of-two scales.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's int8 surface is `MingUniVisionInfer(dtype="int8")` (mingunivision/mingunivisioninfer.py:59-68): HF `QuantoConfig(
weights="int8")`, i.e. the third-party `optimum-quanto` (absent from this image and from /root/reference; any 0.2.x): weight-only,
symmetric, one scale per output channel (axis 0), scale = amax / 127, q = clamp(round(w / scale), -128, 127), y = (x @ q^T) * scale.
Restated here with ONE deviation, stated in DESIGN.md section 5.3: the scale is rounded UP to a power of two,
    scale[n] = 2^ceil(log2(amax_n / 127)),   q = rne(w / scale) in [-127, 127],
which costs at most one of the seven magnitude bits (amax / scale lies in (63.5, 127]) and makes q * scale exactly representable
in bf16: the int8 model is a bf16 model of its own right (like the e4m3 form), runnable through every bf16 route and through the
fp32 oracle with bit-identical weights.  `quanto_rows` is the unrounded-scale form, kept to measure that cost.
PARITY PIN: no reference output exists (the dependency is absent) — the int8 model is defined by this file:
weights W -> dequantize_rows(*quantize_rows(W)); the HIP path is held to the oracle run on those weights (tests/test_gpu_int8.py).
"""
import torch


def pow2_row_scale(w):
    """fp32 [..., N]: s = 2^e with amax / s in (63.5, 127]; 1 for an all-zero row.  amax = ma * 2^ea (1 <= ma < 2), 127 = 1.984375 * 2^6
    -> e = ea - 6, + 1 when ma > 1.984375."""
    amax = w.float().abs().amax(dim=-1)
    m, e = torch.frexp(amax)                              # amax = m * 2^e, 0.5 <= m < 1
    es = (e - 1) - 6 + (2.0 * m > 1.984375).to(e.dtype)
    es = es.clamp(-126, 127)
    s = torch.ldexp(torch.ones_like(amax), es)
    return torch.where(amax == 0, torch.ones_like(s), s)


def quantize_rows(w):
    """W [..., N, K] (bf16 values) -> (uint8 [..., N, K] two's-complement bytes, fp32 scale [..., N])."""
    s = pow2_row_scale(w)
    q = torch.round(w.float() / s.unsqueeze(-1)).clamp(-127, 127).to(torch.int8)      # torch.round: half to even
    return q.view(torch.uint8), s


def dequantize_rows(q, s):
    """-> fp32 [..., N, K] = int8(q) * s: the weights of the int8 model (exactly representable in bf16)."""
    return q.view(torch.int8).float() * s.unsqueeze(-1)


def fake_quant_rows(w):
    """W -> the int8 model's weight values, fp32."""
    return dequantize_rows(*quantize_rows(w))


def quanto_rows(w):
    """optimum-quanto's own rule (scale = amax / 127, fp32): the values an unrounded scale would give — for the cost comparison only."""
    amax = w.float().abs().amax(dim=-1, keepdim=True)
    s = torch.where(amax == 0, torch.ones_like(amax), amax / 127.0)
    return torch.round(w.float() / s).clamp(-128, 127) * s
