"""Tensor-level wrappers of the primitive C-ABI operators (torch tensors in, torch tensors out).

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic
runs in libmingnative.so.  Every wrapper validates dtype/device and raises on
any error code — nothing silently falls back to torch math.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import SkinnyArgs, check, current_stream, lib, ptr

PRO = dict(none=0, silu=1, add_silu=2, rmsnorm=3, ln=4, ln_mod=5)
EPI = dict(none=0, silu=1, gelu=2, swiglu=3, resid=4, resid_gate=5)
GEMM_EPI = dict(bf16=0, bf16_gelu=1, f32=2, f32_resid=3)


def _req(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA/HIP tensor (no CPU fallback exists)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


def as_bf16_bits(t):
    """bf16 tensor -> same storage (the ABI takes uint16 bit patterns)."""
    return t


def skinny_gemm(x, w, bias=None, *, prologue="none", epilogue="none", out=None, pro_a=None, pro_b=None,
                ln_g=None, ln_b=None, eps=1e-6, res=None, gate=None, n_out=None, use_mfma_route=True, wscale=None, wfmt="fp8"):
    """out[M,N] = epilogue(prologue(x)[M,K] @ w[N(,2N),K]^T + bias).  x fp32, w/bias/ln bf16.
    wscale (fp32 [N or 2N]) given: w holds 8-bit weight bytes (uint8: e4m3 for wfmt = "fp8", int8 for "int8") with one scale per
    weight row (the weight-only modes of mingnative.h section 7)."""
    _req(x, torch.float32, "x"); _req(w, torch.uint8 if wscale is not None else torch.bfloat16, "w"); _req(bias, torch.bfloat16, "bias")
    _req(wscale, torch.float32, "wscale")
    _req(ln_g, torch.bfloat16, "ln_g"); _req(ln_b, torch.bfloat16, "ln_b")
    _req(pro_a, torch.float32, "pro_a"); _req(pro_b, torch.float32, "pro_b")
    _req(res, torch.float32, "res"); _req(gate, torch.float32, "gate")
    M, K = x.shape
    N = w.shape[0] // 2 if epilogue == "swiglu" else w.shape[0]
    if n_out is not None:
        N = n_out
    assert w.shape[1] == (K // 2 if (wscale is not None and wfmt == "int4") else K) and x.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    a = SkinnyArgs()
    a.x, a.ldx = ptr(x), x.stride(0)
    a.w, a.ldw = ptr(w), w.stride(0)
    a.bias = ptr(bias)
    a.out, a.ldo = ptr(out), out.stride(0)
    a.M, a.N, a.K = M, N, K
    a.prologue, a.epilogue = PRO[prologue], EPI[epilogue]
    if pro_a is not None:
        a.pro_a, a.ld_pro_a = ptr(pro_a), (0 if pro_a.dim() == 1 else pro_a.stride(0))
    if pro_b is not None:
        a.pro_b, a.ld_pro_b = ptr(pro_b), (0 if pro_b.dim() == 1 else pro_b.stride(0))
    a.ln_g, a.ln_b, a.eps = ptr(ln_g), ptr(ln_b), eps
    if res is not None:
        a.res, a.ldres = ptr(res), res.stride(0)
    if gate is not None:
        a.gate, a.ldgate = ptr(gate), gate.stride(0)
    ws = None
    if wscale is not None:
        assert wscale.numel() == w.shape[0] * (K // 64 if wfmt == "int4" else 1) and w.is_contiguous()
        a.wfmt, a.wscale = _lib.WFMT[wfmt], ptr(wscale)
        a.ldw = K
        nb = lib().mn_skinny_workspace_bytes_wq(a.wfmt, M, N, K, a.epilogue)
    else:
        nb = lib().mn_skinny_workspace_bytes(M, N, K, a.epilogue)
    if nb > 0 and (M > 8 or use_mfma_route or wscale is not None):
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        a.ws, a.ws_bytes = ptr(ws), nb
    check(lib().mn_skinny_gemm(C.byref(a), current_stream()), "mn_skinny_gemm")
    return out


def moe_router(x, norm_w, eps, gate_w, image_gate_w, image_mask, top_k, norm_topk_prob=True, n_shared_slots=0):
    _req(x, torch.float32, "x"); _req(norm_w, torch.bfloat16, "norm_w"); _req(gate_w, torch.bfloat16, "gate_w")
    _req(image_gate_w, torch.bfloat16, "image_gate_w"); _req(image_mask, torch.uint8, "image_mask")
    M, H = x.shape
    E = gate_w.shape[0]
    n_slot = top_k + n_shared_slots
    xn = torch.empty(M, H, dtype=torch.float32, device=x.device)
    idx = torch.empty(M, n_slot, dtype=torch.int32, device=x.device)
    w = torch.empty(M, n_slot, dtype=torch.float32, device=x.device)
    lws = torch.empty(2 * M * E, dtype=torch.float32, device=x.device)
    check(lib().mn_moe_router(ptr(x), x.stride(0), ptr(norm_w), eps, ptr(gate_w), ptr(image_gate_w), ptr(image_mask),
                              M, H, E, top_k, int(norm_topk_prob), n_shared_slots, ptr(xn), ptr(idx), ptr(w), ptr(lws),
                              None, 0, current_stream()), "mn_moe_router")
    return xn, idx, w


def moe_experts(xn, idx, w, w_gate_up, w_down, res, gate_up_scale=None, down_scale=None, wfmt="fp8"):
    """Grouped expert MLPs of one MoE layer for M rows: returns res + sum_slot w * down(silu(gate x) * up x).
    w_gate_up bf16 [E', 2I, H], w_down bf16 [E', H, I]; idx/w [M, n_slot] from moe_router.
    8-bit weight modes: w_gate_up / w_down uint8 (e4m3 or int8 bytes, `wfmt`) with gate_up_scale fp32 [E', 2I] and down_scale fp32 [E', H]."""
    M, H = xn.shape
    n_slot = idx.shape[1]
    I = w_down.shape[2]
    hmid = torch.empty(M * n_slot, I, dtype=torch.float32, device=xn.device)
    a = SkinnyArgs()
    a.x, a.ldx, a.w, a.ldw, a.out, a.ldo = ptr(xn), H, ptr(w_gate_up), H, ptr(hmid), I
    a.M, a.N, a.K = 1, I, H
    a.epilogue = EPI["swiglu"]
    a.batch, a.w_index, a.w_batch_stride = M * n_slot, ptr(idx), 2 * I * H
    a.x_batch_stride, a.x_batch_div, a.out_batch_stride = H, n_slot, I
    if gate_up_scale is not None:
        a.wfmt, a.wscale, a.wscale_batch_stride = _lib.WFMT[wfmt], ptr(gate_up_scale), 2 * I
    check(lib().mn_skinny_gemm(C.byref(a), current_stream()), "mn_skinny_gemm(moe gate_up)")
    out = torch.empty(M, H, dtype=torch.float32, device=xn.device)
    b = SkinnyArgs()
    b.x, b.ldx, b.w, b.ldw, b.out, b.ldo = ptr(hmid), n_slot * I, ptr(w_down), I, ptr(out), H
    b.M, b.N, b.K = 1, H, I
    b.epilogue = EPI["resid"]
    b.res, b.ldres, b.res_batch_stride = ptr(res), res.stride(0), res.stride(0)
    b.batch, b.x_batch_stride, b.x_batch_div, b.out_batch_stride = M, n_slot * I, 1, H
    b.nseg, b.seg_index, b.seg_scale, b.seg_w_stride = n_slot, ptr(idx), ptr(w), H * I
    if down_scale is not None:
        b.wfmt, b.wscale, b.wscale_seg_stride = _lib.WFMT[wfmt], ptr(down_scale), H
    check(lib().mn_skinny_gemm(C.byref(b), current_stream()), "mn_skinny_gemm(moe down)")
    return out


def rope_kv_append(qkv, n_q, n_kv, hd, kv_cache, row_seq, row_slot, row_pos=None, cos=None, sin=None, q_scale=1.0,
                   mrope_section=None):
    """kv_cache fp32 [n_seq, 2, n_kv, t_max, hd].  Returns q_out [M, n_q*hd].
    mrope_section (e.g. [16, 24, 24]): 3D rotary, row_pos is then int32 [3, M] (t, h, w positions)."""
    _req(qkv, torch.float32, "qkv"); _req(kv_cache, torch.float32, "kv_cache")
    M = qkv.shape[0]
    t_max = kv_cache.shape[3]
    q = torch.empty(M, n_q * hd, dtype=torch.float32, device=qkv.device)
    rope = int(cos is not None)
    sec_t, sec_h = (0, 0) if mrope_section is None else (int(mrope_section[0]), int(mrope_section[1]))
    if mrope_section is not None:
        assert row_pos.shape == (3, M) and row_pos.is_contiguous() and sum(mrope_section) == hd // 2
    check(lib().mn_rope_kv_append_3d(ptr(qkv), qkv.stride(0), M, n_q, n_kv, hd, rope, ptr(cos), ptr(sin), ptr(row_seq),
                                     ptr(row_slot), ptr(row_pos), sec_t, sec_h, q_scale, ptr(q), ptr(kv_cache), t_max,
                                     current_stream()), "mn_rope_kv_append")
    return q


def attn_decode(q, n_q, n_kv, hd, kv_cache, row_seq, row_len, key_mask=None):
    _req(q, torch.float32, "q"); _req(key_mask, torch.uint8, "key_mask")
    M = q.shape[0]
    t_max = kv_cache.shape[3]
    nbytes = lib().mn_attn_decode_workspace_bytes(M, n_q, hd, t_max)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
    out = torch.empty(M, n_q * hd, dtype=torch.float32, device=q.device)
    check(lib().mn_attn_decode(ptr(q), M, n_q, n_kv, hd, ptr(kv_cache), t_max, ptr(row_seq), ptr(row_len),
                               ptr(key_mask), 0 if key_mask is None else key_mask.stride(0), ptr(out), ptr(ws), nbytes,
                               current_stream()), "mn_attn_decode")
    return out


def gemm_bf16(a, w, bias=None, epilogue="bf16", out=None):
    """a bf16 [M,K], w bf16 [N,K] -> bf16 or fp32 [M,N] (f32_resid accumulates into `out`)."""
    _req(a, torch.bfloat16, "a"); _req(w, torch.bfloat16, "w"); _req(bias, torch.bfloat16, "bias")
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        assert epilogue != "f32_resid"
        out = torch.empty(M, N, dtype=torch.bfloat16 if epilogue.startswith("bf16") else torch.float32, device=a.device)
    _req(out, torch.bfloat16 if epilogue.startswith("bf16") else torch.float32, "out")
    check(lib().mn_gemm_bf16(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(out), out.stride(0), M, N, K,
                             GEMM_EPI[epilogue], current_stream()), "mn_gemm_bf16")
    return out


def split_hilo(x):
    """fp32 [M, K] -> bf16 [2, M, K]: hi rows then lo rows (x = hi + lo to 2^-17), the operand layout of the hi/lo GEMMs."""
    _req(x, torch.float32, "x")
    x = x.contiguous()
    y = torch.empty((2,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    check(lib().mn_f32_split_bf16(ptr(x), ptr(y[0]), ptr(y[1]), x.numel(), current_stream()), "mn_f32_split_bf16")
    return y


NORM = dict(none=0, rms=1, ln=2)


def norm_act_split(x, norm="none", g=None, b=None, eps=1e-6, gelu=False, want_split=True, want_f32=False):
    """fp32 [M, D] -> act(norm(x)) as the bf16 hi/lo pair [2, M, D] of a hi/lo GEMM and / or as fp32 [M, D].
    Returns (pair | None, f32 | None)."""
    _req(x, torch.float32, "x"); _req(g, torch.bfloat16, "g"); _req(b, torch.bfloat16, "b")
    M, D = x.shape
    assert x.stride(1) == 1
    y = torch.empty(2, M, D, dtype=torch.bfloat16, device=x.device) if want_split else None
    o = torch.empty(M, D, dtype=torch.float32, device=x.device) if want_f32 else None
    check(lib().mn_norm_act_split(ptr(x), x.stride(0), NORM[norm], ptr(g), ptr(b), eps, int(gelu), ptr(y), D,
                                  0 if y is None else y.stride(0), ptr(o), D, M, D, current_stream()), "mn_norm_act_split")
    return y, o


def linear_hilo(a2, w, bias=None, out=None, resid=False):
    """fp32-class Linear on the bf16 MFMA: a2 bf16 [2, M, K] hi/lo pair, w bf16 [N, K] -> fp32 [M, N] (resid: out += ...).
    gemm256 when the shape allows (K % 64 == 0, N % 4 == 0), else the 128-tile hi/lo kernel (any K % 8 == 0; no resid form)."""
    _req(a2, torch.bfloat16, "a2"); _req(w, torch.bfloat16, "w"); _req(bias, torch.bfloat16, "bias")
    _, M, K = a2.shape
    N = w.shape[0]
    assert a2.is_contiguous() and w.shape[1] == K and w.stride(1) == 1
    if out is None:
        assert not resid
        out = torch.empty(M, N, dtype=torch.float32, device=a2.device)
    _req(out, torch.float32, "out")
    if lib().mn_gemm256_supported(K, a2.stride(0), w.stride(0), N, M, N, K) and out.stride(0) % 4 == 0:
        return gemm256(a2, w, bias, "f32_resid" if resid else "f32", out=out)
    if resid:
        raise RuntimeError(f"linear_hilo: residual form needs K % 64 == 0 and N % 4 == 0 (K={K}, N={N})")
    check(lib().mn_gemm_bf16_hilo(ptr(a2[0]), K, a2.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(out), out.stride(0), M, N, K,
                                  current_stream()), "mn_gemm_bf16_hilo")
    return out


def gemm256(a, w, bias=None, epilogue="bf16", out=None):
    """Wide-row GEMM (256 x 256 tiles).  a: bf16 [M, K], or a hi/lo pair bf16 [2, M, K] (fp32-class products).
    w bf16 [N, K] -> bf16 or fp32 [M, N] (f32_resid accumulates into `out`)."""
    _req(a, torch.bfloat16, "a"); _req(w, torch.bfloat16, "w"); _req(bias, torch.bfloat16, "bias")
    hilo = a.dim() == 3
    M, K = a.shape[-2:]
    N = w.shape[0]
    assert a.is_contiguous() and w.shape[1] == K and w.stride(1) == 1
    if out is None:
        assert epilogue != "f32_resid"
        out = torch.empty(M, N, dtype=torch.bfloat16 if epilogue.startswith("bf16") else torch.float32, device=a.device)
    check(lib().mn_gemm256(ptr(a), K, a.stride(0) if hilo else 0, ptr(w), w.stride(0), ptr(bias), ptr(out), out.stride(0), M, N, K,
                           GEMM_EPI[epilogue], current_stream()), "mn_gemm256")
    return out


def gemm256_splitk(a2, w, bias, ksplit):
    """Split-K form on a hi/lo pair a2 bf16 [2, M, K]: returns the fp32 partial slabs [nz, M, N] (bias in slab 0)."""
    _req(a2, torch.bfloat16, "a2"); _req(w, torch.bfloat16, "w"); _req(bias, torch.bfloat16, "bias")
    _, M, K = a2.shape
    N = w.shape[0]
    P = torch.empty(max(1, ksplit), M, N, dtype=torch.float32, device=a2.device)
    nz = lib().mn_gemm256_splitk(ptr(a2), K, a2.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(P), M, N, K, ksplit, current_stream())
    if nz < 0:
        check(nz, "mn_gemm256_splitk")
    return P[:nz]


def gemm256_swiglu_split(a2, w12, b12=None):
    """a2 bf16 [2, M, K] hi/lo pair; w12 bf16 [2 * hidden, K] -> bf16 [2, M, hidden] hi/lo pair of silu(gate) * up."""
    _req(a2, torch.bfloat16, "a2"); _req(w12, torch.bfloat16, "w12"); _req(b12, torch.bfloat16, "b12")
    _, M, K = a2.shape
    hidden = w12.shape[0] // 2
    y = torch.empty(2, M, hidden, dtype=torch.bfloat16, device=a2.device)
    check(lib().mn_gemm256_swiglu_split(ptr(a2), K, a2.stride(0), ptr(w12), w12.stride(0), ptr(b12), ptr(y), hidden, y.stride(0), M,
                                        hidden, K, current_stream()), "mn_gemm256_swiglu_split")
    return y


def gemm256_swiglu(a, w12, b12=None):
    """a bf16 [M, K]; w12 bf16 [2 * hidden, K] (gate rows, up rows) -> bf16 [M, hidden] = silu(gate) * up, one launch."""
    _req(a, torch.bfloat16, "a"); _req(w12, torch.bfloat16, "w12"); _req(b12, torch.bfloat16, "b12")
    M, K = a.shape
    hidden = w12.shape[0] // 2
    y = torch.empty(M, hidden, dtype=torch.bfloat16, device=a.device)
    check(lib().mn_gemm256_swiglu(ptr(a), a.stride(0), 0, ptr(w12), w12.stride(0), ptr(b12), ptr(y), hidden, M, hidden, K,
                                  current_stream()), "mn_gemm256_swiglu")
    return y


def gemm256_f8(a8, a_scale, w8, w_scale, bias=None, swiglu=False, ksplit=1):
    """The fp8-MFMA regime's GEMM (mingnative.h section 8; a labelled reduced-arithmetic regime): a8 uint8 [M, K] e4m3 bytes +
    a_scale fp32 [M], w8 uint8 [N or 2N, K] + w_scale fp32 [N or 2N] (quant_rows(x, "fp8") of both operands) ->
    fp32 [M, N] (swiglu=False; ksplit > 1: the sum of the split-K slabs) or bf16 [M, N] = silu(gate) * up (swiglu=True)."""
    _req(a8, torch.uint8, "a8"); _req(w8, torch.uint8, "w8"); _req(a_scale, torch.float32, "a_scale"); _req(w_scale, torch.float32, "w_scale")
    _req(bias, torch.bfloat16, "bias")
    M, K = a8.shape
    N = w8.shape[0] // (2 if swiglu else 1)
    assert w8.shape[1] == K and a_scale.numel() == M and w_scale.numel() == w8.shape[0] and a8.is_contiguous() and w8.is_contiguous()
    if swiglu:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a8.device)
        nz = lib().mn_gemm256_f8(ptr(a8), K, ptr(a_scale), ptr(w8), K, ptr(w_scale), ptr(bias), ptr(out), N, M, N, K, 1, 1, current_stream())
        if nz < 0:
            check(nz, "mn_gemm256_f8")
        return out
    nzr = lib().mn_gemm256_f8_slices(K, ksplit)
    out = torch.empty(nzr, M, N, dtype=torch.float32, device=a8.device)
    nz = lib().mn_gemm256_f8(ptr(a8), K, ptr(a_scale), ptr(w8), K, ptr(w_scale), ptr(bias), ptr(out), N, M, N, K, 0, ksplit, current_stream())
    if nz < 0:
        check(nz, "mn_gemm256_f8")
    assert nz == nzr
    return out[0] if nz == 1 else out.sum(0)


def gemm256_grouped(a2, a_rows, w, off, cnt, n_pos, m_max, swiglu):
    """Grouped (MoE) form: a2 bf16 [2, R, K] hi/lo pair; a_rows int32 [n_pos] or None; w bf16 [G, N or 2N, K];
    off / cnt int32 device arrays.  Returns fp32 [n_pos, N] or the bf16 hi/lo pair [2, n_pos, N] (swiglu)."""
    _req(a2, torch.bfloat16, "a2"); _req(w, torch.bfloat16, "w"); _req(off, torch.int32, "off"); _req(cnt, torch.int32, "cnt")
    _req(a_rows, torch.int32, "a_rows")
    G, NW, K = w.shape
    N = NW // 2 if swiglu else NW
    if swiglu:
        out = torch.zeros(2, n_pos, N, dtype=torch.bfloat16, device=a2.device)
        lo_off = out.stride(0)
    else:
        out = torch.zeros(n_pos, N, dtype=torch.float32, device=a2.device)
        lo_off = 0
    check(lib().mn_gemm256_grouped(ptr(a2), K, a2.stride(0), a2.shape[1], ptr(a_rows), ptr(w), K, w.stride(0), ptr(off), ptr(cnt), G, ptr(out), N,
                                   lo_off, m_max, N, K, int(swiglu), current_stream()), "mn_gemm256_grouped")
    return out


def gemm256_grouped_tiles(a2, topk_idx, w, n_groups, swiglu):
    """Grouped form over a device-built row-tile list, as the lock-step decoder step runs its experts (modeling_bailing_moe.py:605-639):
    a2 bf16 [2, T, K] hi/lo pair, topk_idx int32 [T, n_slot] group id of every (row, pick); w bf16 [G, N or 2N, K].  Returns (out, off, cnt, perm): out fp32 [T * n_slot, N] or the bf16 hi/lo pair
    [2, T * n_slot, N] (swiglu) in group-sorted order; perm[position] = source row."""
    _req(a2, torch.bfloat16, "a2"); _req(w, torch.bfloat16, "w"); _req(topk_idx, torch.int32, "topk_idx")
    tile_rows = 128
    T, n_slot = topk_idx.shape
    G, NW, K = w.shape
    assert G == n_groups
    N = NW // 2 if swiglu else NW
    dev, n_pos = a2.device, T * n_slot
    i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)
    cnt, off, perm, slot_of = i32(G), i32(G + 1), i32(n_pos), i32(n_pos)
    max_mtiles = n_pos // tile_rows + G
    tile_g, tile_m0, n_tiles = i32(max_mtiles), i32(max_mtiles), i32(1)
    st = current_stream()
    check(lib().mn_moe_sort_tiles(ptr(topk_idx), T, n_slot, G, ptr(cnt), ptr(off), ptr(perm), ptr(slot_of), tile_rows, ptr(tile_g), ptr(tile_m0),
                                  ptr(n_tiles), st), "mn_moe_sort_tiles")
    if swiglu:
        out = torch.zeros(2, n_pos, N, dtype=torch.bfloat16, device=dev)
        lo_off = out.stride(0)
    else:
        out = torch.zeros(n_pos, N, dtype=torch.float32, device=dev)
        lo_off = 0
    check(lib().mn_gemm256_grouped_tiles(ptr(a2), K, a2.stride(0), a2.shape[1], ptr(perm), ptr(w), K, w.stride(0), ptr(off), ptr(cnt), G, ptr(tile_g), ptr(tile_m0), ptr(n_tiles),
             max_mtiles, ptr(out), N, lo_off, N, K, 4 if swiglu else 0, st), "mn_gemm256_grouped_tiles")
    return out, off, cnt, perm


def gemm256_splitk_bf16(a, w, bias, ksplit):
    """Split-K form on plain bf16 activations a [M, K]: fp32 partial slabs [nz, M, N] (bias in slab 0)."""
    _req(a, torch.bfloat16, "a"); _req(w, torch.bfloat16, "w"); _req(bias, torch.bfloat16, "bias")
    M, K = a.shape
    N = w.shape[0]
    P = torch.empty(max(1, ksplit), M, N, dtype=torch.float32, device=a.device)
    nz = lib().mn_gemm256_splitk(ptr(a), a.stride(0), 0, ptr(w), w.stride(0), ptr(bias), ptr(P), M, N, K, ksplit, current_stream())
    if nz < 0:
        check(nz, "mn_gemm256_splitk")
    return P[:nz]


def slab_resid_norm(P, x, g=None, b=None, eps=1e-6, gelu=False, norm=True):
    """x fp32 [M, D] += sum of the slabs P [nz, M, D] (in place); with norm: returns bf16 LayerNorm(x) (then GELU), else None."""
    _req(P, torch.float32, "P"); _req(x, torch.float32, "x"); _req(g, torch.bfloat16, "g"); _req(b, torch.bfloat16, "b")
    nz, M, D = P.shape
    assert x.shape == (M, D) and P.is_contiguous() and x.stride(1) == 1
    y = torch.empty(M, D, dtype=torch.bfloat16, device=x.device) if norm else None
    check(lib().mn_slab_resid_norm(ptr(P), nz, M * D, ptr(x), x.stride(0), ptr(g), ptr(b), eps, int(gelu), ptr(y), D, M, D,
                                   current_stream()), "mn_slab_resid_norm")
    return y


def splitk_plan(M, N, K):
    """Split-K request for a residual GEMM on M rows that would leave 256 x 256 tiles on under half the chip: 0 = do not split."""
    tiles = -(-M // 256) * -(-N // 256)
    if tiles >= 128 or K < 512 or K % 64 or N % 4:
        return 0
    ks = max(2, min(8, 256 // tiles))
    while ks > 2 and K // ks < 256:
        ks -= 1
    return ks


def layernorm_bf16(x, g, b, eps=1e-6, gelu=False):
    _req(x, torch.float32, "x"); _req(g, torch.bfloat16, "g"); _req(b, torch.bfloat16, "b")
    M, D = x.shape
    y = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    check(lib().mn_layernorm_bf16(ptr(x), x.stride(0), ptr(g), ptr(b), eps, ptr(y), D, M, D, int(gelu),
                                  current_stream()), "mn_layernorm_bf16")
    return y


def swiglu_bf16(x12):
    _req(x12, torch.bfloat16, "x12")
    M, H2 = x12.shape
    h = torch.empty(M, H2 // 2, dtype=torch.bfloat16, device=x12.device)
    check(lib().mn_swiglu_bf16(ptr(x12), x12.stride(0), ptr(h), H2 // 2, M, H2 // 2, current_stream()), "mn_swiglu_bf16")
    return h


def attn_prefill_hd64(qkv, B, T, n_heads, causal):
    """qkv bf16 [B*T, 3*n_heads*64] -> bf16 [B*T, n_heads*64]."""
    _req(qkv, torch.bfloat16, "qkv")
    assert qkv.is_contiguous() and qkv.shape == (B * T, 3 * n_heads * 64)
    out = torch.empty(B * T, n_heads * 64, dtype=torch.bfloat16, device=qkv.device)
    check(lib().mn_attn_prefill_hd64(ptr(qkv), ptr(out), B, T, n_heads, int(causal), current_stream()),
          "mn_attn_prefill_hd64")
    return out


def attn_prefill_hd64_f32(qkv, B, T, n_heads, causal, want_f32=False):
    """fp32-class flash attention: qkv fp32 [B*T, 3*n_heads*64] -> the hi/lo pair bf16 [2, B*T, n_heads*64] of the result (the
    projection GEMM's operand) and, with want_f32, the fp32 result itself."""
    _req(qkv, torch.float32, "qkv")
    assert qkv.is_contiguous() and qkv.shape == (B * T, 3 * n_heads * 64)
    split = torch.empty(2, B * T, n_heads * 64, dtype=torch.bfloat16, device=qkv.device)
    out = torch.empty(B * T, n_heads * 64, dtype=torch.float32, device=qkv.device) if want_f32 else None
    check(lib().mn_attn_prefill_hd64_f32(ptr(qkv), ptr(out), ptr(split), B, T, n_heads, int(causal), current_stream()),
          "mn_attn_prefill_hd64_f32")
    return (split, out) if want_f32 else split


def f32_to_bf16(x):
    _req(x, torch.float32, "x")
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib().mn_f32_to_bf16(ptr(x), ptr(y), x.numel(), current_stream()), "mn_f32_to_bf16")
    return y


def bf16_to_f32(x):
    _req(x, torch.bfloat16, "x")
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().mn_bf16_to_f32(ptr(x), ptr(y), x.numel(), current_stream()), "mn_bf16_to_f32")
    return y


def f32_split_bf16(x):
    _req(x, torch.float32, "x")
    x = x.contiguous()
    hi = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib().mn_f32_split_bf16(ptr(x), ptr(hi), ptr(lo), x.numel(), current_stream()), "mn_f32_split_bf16")
    return hi, lo


def lmhead_argmax(hidden, w, vocab_offset=0):
    """Greedy pick: (idx int64 [M], val fp32 [M]) = arg-max / max over v of hidden[m] . w[v], idx offset by vocab_offset
    (w may be a vocabulary slice).  hidden fp32 [M, H], w bf16 [V, H]."""
    _req(hidden, torch.float32, "hidden"); _req(w, torch.bfloat16, "w")
    M, H = hidden.shape
    V = w.shape[0]
    assert hidden.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == H
    n = lib().mn_lmhead_argmax_workspace_bytes(M, V, H)
    ws = torch.empty(n, dtype=torch.uint8, device=hidden.device)
    idx = torch.empty(M, dtype=torch.int64, device=hidden.device)
    val = torch.empty(M, dtype=torch.float32, device=hidden.device)
    check(lib().mn_lmhead_argmax(ptr(hidden), hidden.stride(0), M, ptr(w), w.stride(0), V, H, vocab_offset, ptr(idx), ptr(val), ptr(ws), n,
                                 current_stream()), "mn_lmhead_argmax")
    return idx, val


SAMPLE_CANDIDATES = 2048                   # candidates mn_sample_logits ranks (top_k above this is refused)
SAMPLE_NUCLEUS_TRUNCATED, SAMPLE_TIES_TRUNCATED = 1, 2


def sample_logits(logits, u, temperature=1.0, top_k=0, top_p=1.0, vocab_offset=0, status=None):
    """Sampled pick of every row (mn_sample_logits: HF's temperature -> top-k -> top-p warpers, then the inverse CDF of the kept
    tokens — descending score, ties by ascending id — at u[m] in [0, 1)).  logits fp32 [M, V], u fp32 [M] -> int64 [M].
    status: optional int32 [M] device tensor that receives SAMPLE_* bits for rows whose kept set was cut at 2048 candidates."""
    _req(logits, torch.float32, "logits"); _req(u, torch.float32, "u")
    M, V = logits.shape
    assert logits.stride(1) == 1 and u.numel() == M and u.is_contiguous()
    if top_k > SAMPLE_CANDIDATES:
        raise ValueError(f"top_k = {top_k}: the sampler ranks at most {SAMPLE_CANDIDATES} candidates")
    if status is not None:
        _req(status, torch.int32, "status")
        assert status.numel() == M and status.is_contiguous()
    idx = torch.empty(M, dtype=torch.int64, device=logits.device)
    check(lib().mn_sample_logits(ptr(logits), logits.stride(0), M, V, float(temperature), int(top_k), float(top_p), ptr(u), vocab_offset,
                                 ptr(idx), ptr(status) if status is not None else None, current_stream()), "mn_sample_logits")
    return idx


# ---- fp8 weight mode (mingnative.h section 7) ----------------------------------------------------------------------------------
def quant_fp8_rows(w):
    """bf16 [..., N, K] -> (e4m3 bytes uint8 [..., N, K], fp32 scales [..., N]): one power-of-two scale per output row,
    W = e4m3(Wq) * scale exactly representable in bf16 (mn_quant_fp8_rows)."""
    _req(w, torch.bfloat16, "w")
    assert w.is_contiguous() and w.shape[-1] % 4 == 0
    K = w.shape[-1]
    n_rows = w.numel() // K
    q = torch.empty(w.shape, dtype=torch.uint8, device=w.device)
    scale = torch.empty(w.shape[:-1], dtype=torch.float32, device=w.device)
    check(lib().mn_quant_fp8_rows(ptr(w), K, ptr(q), K, ptr(scale), n_rows, K, current_stream()), "mn_quant_fp8_rows")
    return q, scale


def dequant_fp8_rows(q, scale):
    """(e4m3 bytes [..., N, K], scales [..., N]) -> bf16 [..., N, K] (exact for the power-of-two scales of quant_fp8_rows)."""
    _req(q, torch.uint8, "q"); _req(scale, torch.float32, "scale")
    assert q.is_contiguous() and scale.is_contiguous() and tuple(scale.shape) == tuple(q.shape[:-1])
    K = q.shape[-1]
    w = torch.empty(q.shape, dtype=torch.bfloat16, device=q.device)
    check(lib().mn_dequant_fp8_rows(ptr(q), K, ptr(scale), ptr(w), K, q.numel() // K, K, current_stream()), "mn_dequant_fp8_rows")
    return w


def quant_nf4_rows(w):
    """bf16 [..., N, K] (K % 64 == 0) -> (codes uint8 [..., N, K / 2] in the kernels' nibble order, absmax fp32 [..., N, K / 64]):
    bitsandbytes NF4, blocks of 64 consecutive k (mn_quant_nf4_rows; held bit for bit to the CPU oracle's restatement by the tests)."""
    _req(w, torch.bfloat16, "w")
    K = w.shape[-1]
    assert w.is_contiguous() and K % 64 == 0
    q = torch.empty(w.shape[:-1] + (K // 2,), dtype=torch.uint8, device=w.device)
    absmax = torch.empty(w.shape[:-1] + (K // 64,), dtype=torch.float32, device=w.device)
    check(lib().mn_quant_nf4_rows(ptr(w), K, ptr(q), K // 2, ptr(absmax), w.numel() // K, K, current_stream()), "mn_quant_nf4_rows")
    return q, absmax


def dequant_nf4_rows(q, absmax):
    """-> bf16 [..., N, K] = bf16_rne(NF4[code] * absmax): the int4 model's weights (what bitsandbytes' dequantize_4bit gives in bf16)."""
    _req(q, torch.uint8, "q"); _req(absmax, torch.float32, "absmax")
    K = q.shape[-1] * 2
    assert q.is_contiguous() and absmax.is_contiguous() and tuple(absmax.shape) == tuple(q.shape[:-1]) + (K // 64,)
    w = torch.empty(q.shape[:-1] + (K,), dtype=torch.bfloat16, device=q.device)
    check(lib().mn_dequant_nf4_rows(ptr(q), K // 2, ptr(absmax), ptr(w), K, q.numel() // (K // 2), K, current_stream()), "mn_dequant_nf4_rows")
    return w


def quant_rows(w, fmt):
    """bf16 [..., N, K] -> (codes uint8, fp32 scale table) in the weight-only format `fmt`: "fp8" (OCP e4m3 bytes [..., N, K] + one
    power-of-two scale per row [..., N]), "int8" (optimum-quanto's qint8 rule: two's complement bytes + one bf16-valued scale per row),
    "int4" (bitsandbytes NF4: [..., N, K / 2] + absmax [..., N, K / 64])."""
    if fmt == "fp8":
        return quant_fp8_rows(w)
    if fmt == "int4":
        return quant_nf4_rows(w)
    assert fmt == "int8", fmt
    _req(w, torch.bfloat16, "w")
    assert w.is_contiguous() and w.shape[-1] % 4 == 0
    K = w.shape[-1]
    q = torch.empty(w.shape, dtype=torch.uint8, device=w.device)
    scale = torch.empty(w.shape[:-1], dtype=torch.float32, device=w.device)
    check(lib().mn_quant_int8_rows(ptr(w), K, ptr(q), K, ptr(scale), w.numel() // K, K, current_stream()), "mn_quant_int8_rows")
    return q, scale


def dequant_rows(q, scale, fmt):
    """The inverse of quant_rows: the bf16 weights [..., N, K] the kernels multiply with."""
    if fmt == "fp8":
        return dequant_fp8_rows(q, scale)
    if fmt == "int4":
        return dequant_nf4_rows(q, scale)
    assert fmt == "int8", fmt
    _req(q, torch.uint8, "q"); _req(scale, torch.float32, "scale")
    assert q.is_contiguous() and scale.is_contiguous() and tuple(scale.shape) == tuple(q.shape[:-1])
    K = q.shape[-1]
    w = torch.empty(q.shape, dtype=torch.bfloat16, device=q.device)
    check(lib().mn_dequant_int8_rows(ptr(q), K, ptr(scale), ptr(w), K, q.numel() // K, K, current_stream()), "mn_dequant_int8_rows")
    return w


def fake_quant(w, fmt):
    """bf16 weights -> the bf16 weights of the `fmt` model (quantise, de-quantise): what a converted Linear that stays on the bf16
    kernels holds.  "int4" blocks run over the FLATTENED tensor like bitsandbytes' (64 consecutive elements, whatever the row
    length; a last partial block is padded with zeros, which changes neither its absmax nor its codes)."""
    w = w.contiguous()
    if fmt != "int4" or w.shape[-1] % 64 == 0:
        return dequant_rows(*quant_rows(w, fmt), fmt)
    n = w.numel()
    flat = torch.zeros((n + 63) // 64 * 64, dtype=w.dtype, device=w.device)
    flat[:n] = w.reshape(-1)
    return dequant_rows(*quant_rows(flat.view(-1, 64), fmt), fmt).reshape(-1)[:n].reshape(w.shape).contiguous()


def convert_linears(sd, fmt, skip=()):
    """{name: bf16 tensor} -> the same dict with every nn.Linear weight (2-D `*.weight`; `skip`: name fragments to leave alone, e.g.
    embeddings and the router gates, which are not nn.Linear in the reference) replaced by its `fmt` model value."""
    if fmt not in _lib.FULL_MODEL:
        return sd
    return {k: (fake_quant(v, fmt) if k.endswith(".weight") and v.dim() == 2 and not any(f in k for f in skip) else v) for k, v in sd.items()}


def stream_mfma_w8(y2, q, scale, wfmt="fp8"):
    """Weight-streaming MFMA launch on quantised weights (`wfmt`): y2 bf16 [2, M, K] (hi rows, lo rows), q uint8 [N, K] (int4: [N, K / 2]),
    scale fp32 [N] (int4: absmax [N, K / 64]) -> fp32 [M, N] (the K-slice partials summed here with torch: a test helper, the composites
    reduce them in their glue kernels)."""
    _req(y2, torch.bfloat16, "y2"); _req(q, torch.uint8, "q"); _req(scale, torch.float32, "scale")
    _, M, K = y2.shape
    N = q.shape[0]
    assert y2.is_contiguous() and q.is_contiguous() and q.shape[1] == (K // 2 if wfmt == "int4" else K) and scale.is_contiguous()
    nz = lib().mn_stream_mfma_wq_slices(_lib.WFMT[wfmt], M, N, K)
    P = torch.empty(nz, M, N, dtype=torch.float32, device=q.device)
    rc = lib().mn_stream_mfma_wq(ptr(y2), ptr(q), ptr(scale), ptr(P), M, N, K, _lib.WFMT[wfmt], current_stream())
    if rc < 0:
        check(rc, "mn_stream_mfma_wq")
    assert rc == nz
    return P.sum(0)


# ---- MingTok layout passes (layout_ops.hip) ---------------------------------------------------------------------------------------
def patchify_operand(image, P, hilo=False):
    """image fp32 [B,3,Hi,Wi] -> the patch-embed GEMM's A operand: bf16 [B*N, 3P^2], or the hi/lo pair [2, B*N, 3P^2] (hilo)."""
    _req(image, torch.float32, "image")
    B, C3, Hi, Wi = image.shape
    assert C3 == 3 and image.is_contiguous()
    rows, K = B * (Hi // P) * (Wi // P), 3 * P * P
    y = torch.empty((2, rows, K) if hilo else (rows, K), dtype=torch.bfloat16, device=image.device)
    check(lib().mn_patchify_operand(ptr(image), B, Hi, Wi, P, ptr(y), rows * K if hilo else 0, current_stream()), "mn_patchify_operand")
    return y


def tokens_assemble(tok, cls, pos, B, N):
    """tok fp32 [B*N, D], cls bf16 [D], pos fp32 [N+1, D] -> fp32 [B, N+1, D]: patch tokens, cls LAST, + position embedding."""
    _req(tok, torch.float32, "tok"); _req(cls, torch.bfloat16, "cls"); _req(pos, torch.float32, "pos")
    D = tok.shape[1]
    assert tok.is_contiguous() and pos.is_contiguous() and pos.numel() == (N + 1) * D and cls.numel() == D
    out = torch.empty(B, N + 1, D, dtype=torch.float32, device=tok.device)
    check(lib().mn_tokens_assemble(ptr(tok), ptr(cls), ptr(pos), ptr(out), B, N, D, current_stream()), "mn_tokens_assemble")
    return out


def subtoken_rearrange(y, B, h, w, r, Dp):
    """y fp32 [B*h*w, r*r*Dp] -> fp32 [B*h*r*w*r, Dp] ("b (h w) (x y c) -> b (h x w y) c")."""
    _req(y, torch.float32, "y")
    assert y.is_contiguous() and y.numel() == B * h * w * r * r * Dp
    x = torch.empty(B * h * r * w * r, Dp, dtype=torch.float32, device=y.device)
    check(lib().mn_subtoken_rearrange(ptr(y), ptr(x), B, h, w, r, Dp, current_stream()), "mn_subtoken_rearrange")
    return x


def unpatchify_clamp(o, B, hh, ww, p, lo=-1.0, hi=1.0):
    """o fp32 [B*hh*ww, p*p*3] -> image fp32 [B, 3, hh*p, ww*p] clamped to [lo, hi]."""
    _req(o, torch.float32, "o")
    assert o.is_contiguous() and o.numel() == B * hh * ww * p * p * 3
    img = torch.empty(B, 3, hh * p, ww * p, dtype=torch.float32, device=o.device)
    check(lib().mn_unpatchify_clamp(ptr(o), ptr(img), B, hh, ww, p, float(lo), float(hi), current_stream()), "mn_unpatchify_clamp")
    return img
