"""The fp8-MFMA regime (mingnative.h section 8; BASELINE.json configs[4]'s "fp8 MFMA"): a LABELLED reduced-arithmetic regime with its own
stated tolerance — the reference has no fp8 arithmetic, the parity target is the fp32 oracle at a looser bar.

Two levels:
  * the KERNEL is exact arithmetic on what it is given: products of two e4m3 values are exact in fp32 and the accumulation is fp32, so
    `mn_gemm256_f8` must agree with an fp64 product of the SAME quantised operands to fp32-accumulation rounding (<= 2e-5), over ragged
    shapes, split-K and the SwiGLU epilogue — the 1e-3-class bar of every other kernel here, untouched by the regime;
  * the REGIME (quantising both operands to e4m3 with one power-of-two scale per row) is what costs accuracy: against the fp64 product of
    the UN-quantised operands the stated tolerance is 6e-2 relative in max-norm (e4m3 carries 3 mantissa bits: 2^-4 relative per operand)."""
import pytest
import torch

from ming_univision_amd import ops
from tests.util import rel_err

pytestmark = pytest.mark.gpu

KERNEL_TOL = 2e-5
REGIME_TOL = 6e-2


def _deq(q, s):
    return ops.dequant_rows(q, s, "fp8").double().cpu()


@pytest.mark.parametrize("M,N,K,ks", [(200, 132, 256, 1), (1536, 3072, 1024, 1), (333, 260, 1152, 3), (128, 64, 8192, 4), (2048, 512, 3072, 1)])
def test_gemm256_f8_is_exact_on_its_operands(M, N, K, ks):
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 3).to(torch.bfloat16).cuda()      # rows of different magnitude
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).cuda()
    x8, xs = ops.quant_rows(x, "fp8")
    w8, ws = ops.quant_rows(w, "fp8")
    y = ops.gemm256_f8(x8, xs, w8, ws, bias=b, ksplit=ks)
    ref_q = _deq(x8, xs) @ _deq(w8, ws).T + b.double().cpu()
    ref = x.double().cpu() @ w.double().cpu().T + b.double().cpu()
    e_k, e_r = rel_err(y, ref_q), rel_err(y, ref)
    print(f"gemm256_f8 {M}x{N}x{K} ks={ks}: vs fp64 on the quantised operands {e_k:.2e} | regime vs the bf16 operands {e_r:.2e}")
    assert e_k < KERNEL_TOL and e_r < REGIME_TOL


@pytest.mark.parametrize("M,H,K", [(200, 132, 256), (1536, 1024, 3072)])
def test_gemm256_f8_swiglu(M, H, K):
    g = torch.Generator().manual_seed(7 + M)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w12 = (torch.randn(2 * H, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    b12 = (torch.randn(2 * H, generator=g) * 0.1).to(torch.bfloat16).cuda()
    x8, xs = ops.quant_rows(x, "fp8")
    w8, ws = ops.quant_rows(w12, "fp8")
    y = ops.gemm256_f8(x8, xs, w8, ws, bias=b12, swiglu=True)
    r = _deq(x8, xs) @ _deq(w8, ws).T + b12.double().cpu()
    ref_q = torch.nn.functional.silu(r[:, :H]) * r[:, H:]
    e = rel_err(y, ref_q)
    print(f"gemm256_f8 SwiGLU {M}x{H}x{K}: vs fp64 on the quantised operands {e:.2e} (bf16 result: 1 ulp = 3.9e-3 of a value)")
    assert e < 4e-3          # the result is rounded to bf16


def test_gemm256_f8_rejects_what_it_cannot_run():
    x8 = torch.zeros(16, 192, dtype=torch.uint8, device="cuda"); s = torch.ones(16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gemm256_f8(x8, s, x8, s)                      # K % 128 != 0
