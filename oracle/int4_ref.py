"""CPU oracle of the int4 weight mode: bitsandbytes NF4 (4-bit NormalFloat), blockwise absmax.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's int4 surface is `MingUniVisionInfer(dtype="int4")` (mingunivision/mingunivisioninfer.py:46-58):
    BitsAndBytesConfig(load_in_4bit=True, bnb_4bit_compute_dtype=torch.bfloat16, bnb_4bit_quant_type="nf4",
                       llm_int8_skip_modules=["BailingAudioModel"])
i.e. the third-party `bitsandbytes` (requirements.txt:27, UNPINNED; absent from this image and from /root/reference) behind HF's
`replace_with_bnb_linear`: every nn.Linear except the modules of the caller's skip list becomes a `Linear4bit` — the reference's own
list (the audio tower) REPLACES HF's default one ("keep the output head"), so `lm_head` converts too, as `_lib.FULL_MODEL` and the
loaders here do — whose weight is stored as

    * 4-bit codes into the 16-entry NF4 table — the quantiles of N(0, 1) construction of the QLoRA paper (Dettmers et al. 2023, §3
      "4-bit NormalFloat", appendix E; bitsandbytes `functional.create_normal_map(offset=0.9677083)`), normalised to [-1, 1] with an
      exact 0;
    * one fp32 `absmax` per BLOCK of 64 consecutive elements of the flattened (row-major) weight (`blocksize=64`, the 4-bit default);
      `bnb_4bit_use_double_quant` is left at its default False, so the absmax values stay fp32;
    * quantisation (csrc/kernels.cu `kQuantizeBlockwise` / `dQuantizeNF4`): x = w * (1 / absmax) in fp32, code = the table entry
      nearest to x by a comparison tree over the midpoints of adjacent entries (`x > midpoint` goes up: a value exactly on a
      midpoint takes the LOWER entry); two codes per byte, the even element in the high nibble;
    * de-quantisation (`kDequantizeBlockwise`, what `bnb.matmul_4bit` multiplies with for more than one activation row):
      w' = table[code] * absmax in fp32, stored in the compute dtype -> bf16 (round to nearest even).

So the int4 model is the bf16 model with every converted Linear weight W replaced by W' = bf16(NF4[code(W)] * absmax(W)) — a bf16
model of its own right: the HIP path streams the 4-bit codes + absmax where the bytes matter (RF ResBlock matrices, adaLN, experts:
95 % of what a visual token reads) and holds W' as bf16 elsewhere; the parity tests feed W' to the fp32 oracle.

PARITY PIN: "parity unpinned" against bitsandbytes itself (absent, unpinned in the reference).  What IS pinned here:
`tests/test_int4_oracle.py` rebuilds the table from its published construction (scipy's normal quantile function) and holds the
constants below to it at fp32 resolution, checks the midpoints / tie rule / packing / block layout on hand-made cases, and the device
quantiser + every kernel's decoder are held bit-for-bit to this file (tests/test_gpu_int4.py).
"""
import numpy as np
import torch

BLOCK = 64

# bitsandbytes functional.get_4bit_type("nf4") — the table as published (fp32 values)
NF4_TABLE = (
    -1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635, -0.18477343022823334,
    -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
    0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0)


def table():
    return torch.tensor(NF4_TABLE, dtype=torch.float32)


def midpoints():
    """The 15 decision thresholds of dQuantizeNF4: fp32 midpoints of adjacent table entries."""
    t = table().double()
    return ((t[1:] + t[:-1]) / 2).float()


def quantize_blocks(w):
    """W [..., N, K] (bf16 values; K % 64 == 0, so no block straddles a row) -> (codes uint8 [..., N, K] in 0..15,
    absmax fp32 [..., N, K / 64])."""
    K = w.shape[-1]
    assert K % BLOCK == 0
    wf = w.float().reshape(*w.shape[:-1], K // BLOCK, BLOCK)
    absmax = wf.abs().amax(-1)
    inv = torch.where(absmax == 0, torch.zeros_like(absmax), 1.0 / absmax)          # fp32 reciprocal, then a multiply (kernels.cu)
    x = wf * inv.unsqueeze(-1)
    codes = torch.searchsorted(midpoints(), x.contiguous(), right=False)           # number of midpoints < x  ( `x > m` goes up )
    return codes.to(torch.uint8).reshape(w.shape), absmax


def dequantize_blocks(codes, absmax):
    """-> fp32 [..., N, K] holding bf16 values: bf16_rne(NF4[code] * absmax) — the weights of the int4 model."""
    K = codes.shape[-1]
    t = table()[codes.long()].reshape(*codes.shape[:-1], K // BLOCK, BLOCK)
    return (t * absmax.unsqueeze(-1)).to(torch.bfloat16).float().reshape(codes.shape)


def fake_quant(w):
    """W (any shape) -> the int4 model's weight values (fp32 tensor of bf16 values).  bitsandbytes blocks the FLATTENED tensor: for
    rows that are a multiple of 64 long that is the per-row blocking above; otherwise blocks straddle rows (e.g. the RF head's
    input_proj [w, 32]) and a last partial block stands alone (zero padding changes neither its absmax nor its codes)."""
    if w.shape[-1] % BLOCK == 0:
        return dequantize_blocks(*quantize_blocks(w))
    n = w.numel()
    flat = torch.zeros((n + BLOCK - 1) // BLOCK * BLOCK)
    flat[:n] = w.float().reshape(-1)
    return dequantize_blocks(*quantize_blocks(flat.view(-1, BLOCK))).reshape(-1)[:n].reshape(w.shape)


def pack_bnb(codes):
    """bitsandbytes' byte layout: element 2j in the HIGH nibble of byte j, element 2j + 1 in the low nibble."""
    c = codes.reshape(-1, 2)
    return ((c[:, 0] << 4) | c[:, 1]).to(torch.uint8)


def pack_kernel(codes):
    """The byte layout the HIP kernels stream (ming_univision_amd/csrc/w8_codec.h): rows of K / 2 bytes; inside every group of 8
    consecutive elements e0..e7 (one little-endian dword) the nibbles are, from bit 0 up, e0 e4 e1 e5 e2 e6 e3 e7 — so that
    `x & 0x0f0f0f0f` is (e0 e1 e2 e3) and `(x >> 4) & 0x0f0f0f0f` is (e4 e5 e6 e7), one byte each, ready for v_perm_b32 lookups.
    Same codes, same absmax, another order of nibbles: the dequantised weights are bitsandbytes'."""
    K = codes.shape[-1]
    assert K % 8 == 0
    c = codes.reshape(*codes.shape[:-1], K // 8, 8).to(torch.uint8)
    lo, hi = c[..., :4], c[..., 4:]
    return (lo | (hi << 4)).reshape(*codes.shape[:-1], K // 2)


def unpack_kernel(packed):
    b = packed.reshape(*packed.shape[:-1], packed.shape[-1] // 4, 4)
    return torch.cat([b & 15, b >> 4], dim=-1).reshape(*packed.shape[:-1], packed.shape[-1] * 2)


def nf4_from_construction():
    """The table rebuilt from its published construction (QLoRA appendix E / bitsandbytes create_normal_map, offset 0.9677083):
    8 positive quantiles of N(0, 1) at linspace(offset, 0.5, 9)[:-1], 7 negative ones at linspace(offset, 0.5, 8)[:-1], and an
    exact zero, normalised by the largest.  float64 -> compared with the fp32 constants by the CPU test."""
    from scipy.stats import norm
    offset = 0.9677083
    pos = norm.ppf(np.linspace(offset, 0.5, 9)[:-1])
    neg = -norm.ppf(np.linspace(offset, 0.5, 8)[:-1])
    v = np.sort(np.concatenate([pos, [0.0], neg]))
    return v / v.max()
