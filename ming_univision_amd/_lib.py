"""ctypes binding of libmingnative.so (the C ABI declared in include/mingnative.h).

The library is the product: there is NO fallback.  If the shared object is missing
or cannot be loaded, `lib()` raises and every operator of this package fails loudly.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C ming_univision_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmingnative.so")

c_bf16_p = C.c_void_p
c_f32_p = C.c_void_p
c_i32_p = C.c_void_p
c_u8_p = C.c_void_p


class SkinnyArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64),
        ("w", C.c_void_p), ("ldw", C.c_int64),
        ("bias", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("prologue", C.c_int32), ("epilogue", C.c_int32),
        ("pro_a", C.c_void_p), ("ld_pro_a", C.c_int64),
        ("pro_b", C.c_void_p), ("ld_pro_b", C.c_int64),
        ("ln_g", C.c_void_p), ("ln_b", C.c_void_p),
        ("eps", C.c_float),
        ("res", C.c_void_p), ("ldres", C.c_int64),
        ("gate", C.c_void_p), ("ldgate", C.c_int64),
        ("batch", C.c_int32), ("w_index", C.c_void_p), ("w_batch_stride", C.c_int64),
        ("x_batch_stride", C.c_int64), ("x_batch_div", C.c_int32), ("out_batch_stride", C.c_int64),
        ("res_batch_stride", C.c_int64),
        ("nseg", C.c_int32), ("seg_index", C.c_void_p), ("seg_scale", C.c_void_p), ("seg_w_stride", C.c_int64),
        ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
        ("wfmt", C.c_int32), ("wscale", C.c_void_p), ("wscale_batch_stride", C.c_int64), ("wscale_seg_stride", C.c_int64),
    ]


PP = C.POINTER(C.c_void_p)


class RfHead(C.Structure):
    _fields_ = [
        ("w", C.c_int32), ("depth", C.c_int32), ("hidden", C.c_int32), ("z_dim", C.c_int32),
        ("target", C.c_int32), ("steps", C.c_int32), ("llm_hidden", C.c_int32),
        ("vis_w", C.c_void_p), ("vis_b", C.c_void_p), ("vis_ln_g", C.c_void_p), ("vis_ln_b", C.c_void_p),
        ("cond_w", C.c_void_p), ("cond_b", C.c_void_p),
        ("in_w", C.c_void_p), ("in_b", C.c_void_p),
        ("temb", C.c_void_p),
        ("ada_w", C.c_void_p), ("ada_b", C.c_void_p),
        ("ln_g", PP), ("ln_b", PP), ("w12", PP), ("b12", PP), ("w3", PP), ("b3", PP),
        ("fin_w", C.c_void_p), ("fin_b", C.c_void_p),
        ("wfmt", C.c_int32), ("w12_scale", PP), ("w3_scale", PP), ("ada_q", C.c_void_p), ("ada_scale", C.c_void_p),
        ("arith", C.c_int32),
    ]


class Llm(C.Structure):
    _fields_ = [
        ("hidden", C.c_int32), ("n_layers", C.c_int32), ("n_q", C.c_int32), ("n_kv", C.c_int32),
        ("head_dim", C.c_int32), ("n_experts", C.c_int32), ("top_k", C.c_int32), ("n_shared_slots", C.c_int32),
        ("moe_inter", C.c_int32), ("norm_topk_prob", C.c_int32),
        ("rms_eps", C.c_float),
        ("ln1", PP), ("wqkv", PP), ("wdense", PP), ("ln2", PP), ("gate", PP), ("image_gate", PP),
        ("w_gate_up", PP), ("w_down", PP),
        ("final_norm", C.c_void_p),
        ("cos_tab", C.c_void_p), ("sin_tab", C.c_void_p),
        ("n_pos", C.c_int32),
        ("mrope_sec_t", C.c_int32), ("mrope_sec_h", C.c_int32),
        ("wfmt", C.c_int32), ("w_gate_up_scale", PP), ("w_down_scale", PP),
        ("arith", C.c_int32),
    ]


class SemDec(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("depth", C.c_int32), ("n_heads", C.c_int32), ("hidden", C.c_int32),
        ("in_dim", C.c_int32), ("proj_dim", C.c_int32), ("proj_depth", C.c_int32),
        ("mean", C.c_float), ("scale", C.c_float),
        ("in_w", C.c_void_p), ("in_b", C.c_void_p),
        ("ln1_g", PP), ("ln1_b", PP), ("wqkv", PP), ("bqkv", PP), ("wproj", PP), ("bproj", PP),
        ("ln2_g", PP), ("ln2_b", PP), ("w12", PP), ("b12", PP), ("w3", PP), ("b3", PP),
        ("norm_g", C.c_void_p), ("norm_b", C.c_void_p),
        ("proj_w", PP), ("proj_b", PP),
        ("hidden_pad", C.c_int32), ("w12p", PP), ("b12p", PP), ("w3p", PP),
    ]


class TpComm(C.Structure):
    _fields_ = [
        ("rank", C.c_int32), ("world", C.c_int32),
        ("inbox", PP), ("flags", PP),
        ("cap", C.c_int64), ("rows_cap", C.c_int32), ("epoch", C.c_uint32),
        ("err", C.c_void_p), ("wait_ms", C.c_uint32), ("two_shot_rows", C.c_int32),
    ]


class LlmTp(C.Structure):
    _fields_ = [
        ("expert0", C.c_int32), ("n_local_experts", C.c_int32), ("shared_inter", C.c_int32),
        ("ws_gate_up", PP), ("ws_down", PP),
        ("ws_gate_up_scale", PP), ("ws_down_scale", PP),
    ]


# (struct id of mn_struct_layout, its mn_sizeof_* export, the ctypes mirror): checked against the library at load time (`lib()`)
STRUCTS = ((0, "mn_sizeof_skinny_args", SkinnyArgs), (1, "mn_sizeof_rf_head", RfHead), (2, "mn_sizeof_llm", Llm),
           (3, "mn_sizeof_semdec", SemDec), (4, "mn_sizeof_tp_comm", TpComm), (5, "mn_sizeof_llm_tp", LlmTp))


def check_struct_layouts(handle):
    """Every ctypes Structure above against the library's own sizeof / offsetof (mn_sizeof_*, mn_struct_layout): a field added or
    moved on one side only raises here instead of corrupting calls.  Needs no GPU."""
    for sid, sizer, klass in STRUCTS:
        fn = getattr(handle, sizer)
        fn.restype, fn.argtypes = C.c_size_t, []
        if fn() != C.sizeof(klass):
            raise RuntimeError(f"{klass.__name__}: ctypes size {C.sizeof(klass)} != library {fn()} ({sizer})")
        lay = handle.mn_struct_layout
        lay.restype, lay.argtypes = C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.c_int]
        buf = (C.c_size_t * 128)()
        n = lay(sid, buf, 128)
        mine = [getattr(klass, name).offset for name, _ in klass._fields_]
        if n != len(mine) or list(buf[:n]) != mine:
            raise RuntimeError(f"{klass.__name__}: field offsets differ from the library's (struct id {sid}): {mine} vs {list(buf[:max(n, 0)])}")


W_BF16, W_FP8_E4M3, W_INT8, W_NF4 = 0, 1, 2, 3   # mingnative.h section 7: weight formats of the streaming route
WFMT = {"bf16": W_BF16, "fp8": W_FP8_E4M3, "int8": W_INT8, "int4": W_NF4}
W8 = ("fp8", "int8", "int4")                   # the weight-only modes: codes (one or half a byte per weight) + a scale table
# modes that convert EVERY nn.Linear of the model, as the reference's HF quantisation configs do (mingunivisioninfer.py:46-68:
# `llm_int8_skip_modules` / `modules_to_not_convert` are given, which REPLACES HF's default skip list — so lm_head is converted too;
# the "BailingAudioModel" entry matches no module path).  Not converted: nn.Embedding, Conv2d (patch embed), norms, biases, and the
# router gates (BailingMoeGate holds a bare nn.Parameter, modeling_bailing_moe.py:497).  "fp8" is this library's own byte format for
# the streamed tensors only (experts, RF ResBlocks, adaLN).
FULL_MODEL = ("int4", "int8")

_lib = None

# every symbol include/mingnative.h declares: (name, restype, argtypes)
_i, _i64, _f, _p, _sz = C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_size_t
SYMBOLS = {
    "mn_version": (_i, []),
    "mn_last_error": (C.c_char_p, []),
    "mn_num_cus": (_i, []),
    "mn_sizeof_skinny_args": (_sz, []), "mn_sizeof_rf_head": (_sz, []), "mn_sizeof_llm": (_sz, []), "mn_sizeof_semdec": (_sz, []),
    "mn_sizeof_tp_comm": (_sz, []), "mn_sizeof_llm_tp": (_sz, []),
    "mn_struct_layout": (_i, [_i, C.POINTER(C.c_size_t), _i]),
    "mn_skinny_gemm": (_i, [C.POINTER(SkinnyArgs), _p]),
    "mn_skinny_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "mn_skinny_workspace_bytes_w8": (_sz, [_i, _i, _i, _i]),
    "mn_skinny_workspace_bytes_wq": (_sz, [_i, _i, _i, _i, _i]),
    "mn_quant_fp8_rows": (_i, [_p, _i64, _p, _i64, _p, _i64, _i, _p]),
    "mn_dequant_fp8_rows": (_i, [_p, _i64, _p, _p, _i64, _i64, _i, _p]),
    "mn_quant_int8_rows": (_i, [_p, _i64, _p, _i64, _p, _i64, _i, _p]),
    "mn_quant_nf4_rows": (_i, [_p, _i64, _p, _i64, _p, _i64, _i, _p]),
    "mn_dequant_nf4_rows": (_i, [_p, _i64, _p, _p, _i64, _i64, _i, _p]),
    "mn_stream_mfma_wq_slices": (_i, [_i, _i, _i, _i]),
    "mn_tp_allreduce_segments": (_i, [C.POINTER(TpComm), _i, _i]),
    "mn_dequant_int8_rows": (_i, [_p, _i64, _p, _p, _i64, _i64, _i, _p]),
    "mn_stream_mfma_wq": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mn_stream_mfma_grouped_wq": (_i, [_p, _i, _p, _i64, _p, _i64, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mn_stream_mfma_w8": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "mn_stream_mfma_w8_slices": (_i, [_i, _i, _i]),
    "mn_stream_mfma_grouped_w8": (_i, [_p, _i, _p, _i64, _p, _i64, _p, _i, _p, _p, _i, _i, _i, _i, _p]),
    "mn_gemm_bf16_hilo": (_i, [_p, _i64, _i64, _p, _i64, _p, _p, _i64, _i, _i, _i, _p]),
    "mn_gemm_bf16_splitk": (_i, [_p, _i64, _p, _i64, _p, _i, _i, _i, _i, _p]),
    "mn_stream_mfma": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "mn_rmsnorm_bf16": (_i, [_p, _i64, _p, _f, _p, _i64, _i, _i, _p]),
    "mn_rope_kv_prefill": (_i, [_p, _i64, _i, _i, _i, _i, _p, _p, _p, _i, _f, _p, _p, _i64, _p]),
    "mn_attn_prefill_gqa_hd128": (_i, [_p, _p, _i64, _i, _i, _i, _i, _p, _p, _p]),
    "mn_moe_topk_logits": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "mn_moe_sort": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p]),
    "mn_flash_prefill_gqa_hd128": (_i, [_p, _p, _i64, _i, _i, _i, _p, _i, _i, _p, _i64, _p, _p]),
    "mn_norm_act_split": (_i, [_p, _i64, _i, _p, _p, _f, _i, _p, _i64, _i64, _p, _i64, _i, _i, _p]),
    "mn_slab_resid_norm": (_i, [_p, _i, _i64, _p, _i64, _p, _p, _f, _i, _p, _i64, _i, _i, _p]),
    "mn_rope_kv_prefill_spans": (_i, [_p, _i64, _i, _i, _i, _p, _p, _p, _i, _f, _p, _p, _i64, _p, _i, _i, _p]),
    "mn_moe_combine_norm": (_i, [_p, _p, _p, _i, _p, _i64, _p, _f, _p, _i64, _i, _i, _p]),
    "mn_moe_sort_tiles": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "mn_gemm256_grouped_tiles": (_i, [_p, _i64, _i64, _i64, _p, _p, _i64, _i64, _p, _p, _i, _p, _p, _p, _i, _p, _i64, _i64, _i, _i, _i, _p]),
    "mn_gather_rows_bf16": (_i, [_p, _i64, _p, _p, _i64, _i, _i, _p]),
    "mn_moe_combine": (_i, [_p, _i64, _p, _p, _i, _p, _i64, _i, _i, _p]),
    "mn_gemm_bf16_grouped": (_i, [_p, _i64, _p, _i64, _i64, _p, _p, _i, _p, _i64, _i, _i, _i, _i, _p]),
    "mn_gemm256_supported": (_i, [_i64, _i64, _i64, _i64, _i, _i, _i]),
    "mn_gemm256": (_i, [_p, _i64, _i64, _p, _i64, _p, _p, _i64, _i, _i, _i, _i, _p]),
    "mn_gemm256_splitk": (_i, [_p, _i64, _i64, _p, _i64, _p, _p, _i, _i, _i, _i, _p]),
    "mn_gemm256_swiglu_split": (_i, [_p, _i64, _i64, _p, _i64, _p, _p, _i64, _i64, _i, _i, _i, _p]),
    "mn_gemm256_swiglu": (_i, [_p, _i64, _i64, _p, _i64, _p, _p, _i64, _i, _i, _i, _p]),
    "mn_gemm256_grouped": (_i, [_p, _i64, _i64, _i64, _p, _p, _i64, _i64, _p, _p, _i, _p, _i64, _i64, _i, _i, _i, _i, _p]),
    "mn_stream_mfma_slices": (_i, [_i, _i, _i]),
    "mn_stream_mfma_grouped": (_i, [_p, _i, _p, _i64, _p, _i, _p, _p, _i, _i, _i, _i, _p]),
    "mn_stream_mfma_grouped_slices": (_i, [_i, _i, _i, _i]),
    "mn_moe_router": (_i, [_p, _i64, _p, _f, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "mn_rope_kv_append": (_i, [_p, _i64, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _i64, _p]),
    "mn_rope_kv_append_3d": (_i, [_p, _i64, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _f, _p, _p, _i64, _p]),
    "mn_attn_decode_workspace_bytes": (_sz, [_i, _i, _i, _i64]),
    "mn_attn_decode": (_i, [_p, _i, _i, _i, _i, _p, _i64, _p, _p, _p, _i64, _p, _p, _sz, _p]),
    "mn_gemm_bf16": (_i, [_p, _i64, _p, _i64, _p, _p, _i64, _i, _i, _i, _i, _p]),
    "mn_layernorm_bf16": (_i, [_p, _i64, _p, _p, _f, _p, _i64, _i, _i, _i, _p]),
    "mn_swiglu_bf16": (_i, [_p, _i64, _p, _i64, _i, _i, _p]),
    "mn_attn_prefill_hd64": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "mn_attn_prefill_hd64_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "mn_flash_prefill_gqa_hd128_f32": (_i, [_p, _p, _i64, _i, _i, _p, _i, _i, _p, _p, _i64, _p]),
    "mn_f32_to_bf16": (_i, [_p, _p, _i64, _p]),
    "mn_bf16_to_f32": (_i, [_p, _p, _i64, _p]),
    "mn_f32_split_bf16": (_i, [_p, _p, _p, _i64, _p]),
    "mn_rf_workspace_bytes": (_sz, [C.POINTER(RfHead), _i]),
    "mn_persist_set_status_word": (_i, [_p]),
    "mn_llm_route_capture": (_i, [_p]),
    "mn_gemm256_f8": (_i, [_p, _i64, _p, _p, _i64, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "mn_gemm256_f8_slices": (_i, [_i, _i]),
    "mn_rf_max_rows": (_i, [C.POINTER(RfHead)]),
    "mn_llm_max_rows": (_i, [C.POINTER(Llm)]),
    "mn_semdec_max_rows": (_i, [C.POINTER(SemDec)]),
    "mn_rf_sample": (_i, [C.POINTER(RfHead), _p, _i64, _i, _i, _p, _f, _f, _f, _p, _p, _sz, _p]),
    "mn_llm_workspace_bytes": (_sz, [C.POINTER(Llm), _i, _i64]),
    "mn_llm_step": (_i, [C.POINTER(Llm), _p, _i64, _i, _i, _p, _p, _p, _p, _p, _p, _i64, _p, _i, _i64, _p, _p, _sz, _p]),
    "mn_llm_step_ex": (_i, [C.POINTER(Llm), _p, _i64, _i, _i, _p, _p, _p, _p, _p, _p, _i64, _p, _i, _i64, _p, _p, _sz, _i, _p]),
    "mn_llm_step_spans": (_i, [C.POINTER(Llm), _p, _i64, _i, _p, _p, _p, _p, _p, _p, _i, _i64, _p, _i, _i, _p, _p, _sz, _p]),
    "mn_rows_advance": (_i, [_p, _p, _p, _i, _i, _p]),
    "mn_add_bcast_f32": (_i, [_p, _p, _p, _i64, _i64, _p]),
    "mn_group_mean_add": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "mn_repeat_add": (_i, [_p, _p, _p, _i, _i, _i, _f, _f, _p]),
    "mn_clamp_f32": (_i, [_p, _i64, _f, _f, _p]),
    "mn_patchify_operand": (_i, [_p, _i, _i, _i, _i, _p, _i64, _p]),
    "mn_tokens_assemble": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "mn_subtoken_rearrange": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "mn_unpatchify_clamp": (_i, [_p, _p, _i, _i, _i, _i, _f, _f, _p]),
    "mn_tp_alloc": (_i, [_sz, C.POINTER(C.c_void_p)]),
    "mn_tp_free": (_i, [_p]),
    "mn_tp_ipc_handle": (_i, [_p, _p]),
    "mn_tp_ipc_open": (_i, [_p, C.POINTER(C.c_void_p)]),
    "mn_tp_ipc_close": (_i, [_p]),
    "mn_allreduce_oneshot": (_i, [C.POINTER(TpComm), _p, _i64, _p, _i64, _i, _i, _i, _p]),
    "mn_ep_dispatch": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "mn_ep_combine": (_i, [C.POINTER(TpComm), _p, _p, _p, _p, _i, _i, _i, _p, _i, _i64, _p, _i64, _p, _i64, _i, _i, _p]),
    "mn_llm_tp_workspace_bytes": (_sz, [C.POINTER(Llm), C.POINTER(LlmTp), _i, _i64]),
    "mn_llm_tp_segments": (_i, [C.POINTER(Llm)]),
    "mn_llm_step_tp": (_i, [C.POINTER(Llm), C.POINTER(LlmTp), C.POINTER(TpComm), _p, _i64, _i, _i, _p, _p, _p, _p, _p, _p, _i64, _p, _i,
                            _i64, _p, _p, _sz, _i, _i, _p]),
    "mn_rf_tp_workspace_bytes": (_sz, [C.POINTER(RfHead), _i]),
    "mn_rf_tp_segments": (_i, [C.POINTER(RfHead)]),
    "mn_rf_sample_tp": (_i, [C.POINTER(RfHead), C.POINTER(TpComm), _p, _i64, _i, _i, _p, _f, _f, _f, _p, _p, _sz, _i, _i, _p]),
    "mn_lmhead_argmax_workspace_bytes": (_sz, [_i, _i, _i]),
    "mn_lmhead_argmax": (_i, [_p, _i64, _i, _p, _i64, _i, _i, _i64, _p, _p, _p, _sz, _p]),
    "mn_sample_logits": (_i, [_p, _i64, _i, _i, _f, _i, _f, _p, _i64, _p, _p, _p]),
    "mn_semdec_workspace_bytes": (_sz, [C.POINTER(SemDec), _i, _i64]),
    "mn_semdec_step": (_i, [C.POINTER(SemDec), _p, _i, _p, _p, _p, _p, _i, _i64, _p, _p, _p, _sz, _p]),
}


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `make -C {_HERE}/csrc` "
            "(or __graft_entry__.build()). There is no CPU/PyTorch fallback.")
    import torch  # noqa: F401  (before the library: the process must hold ONE HIP runtime — torch's bundled one; loaded first, the
    #                            library binds the system copy and torch's later launches fail with "no ROCm-capable device is detected")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(handle, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if handle.mn_version() < 124:
        raise RuntimeError("libmingnative.so is too old")
    check_struct_layouts(handle)
    _lib = handle
    return _lib


_persist_word = None


def persist_status_word(device):
    """The process's sticky status word of the persistent launches (mn_persist_set_status_word): one zeroed int32 in HBM, registered
    with the library at first use.  A grid-barrier wait that expires raises it; persist_check() reads it at a host sync."""
    global _persist_word
    if _persist_word is None:
        import torch
        _persist_word = torch.zeros(1, dtype=torch.int32, device=device)
        check(lib().mn_persist_set_status_word(ptr(_persist_word)), "mn_persist_set_status_word")
    return _persist_word


def persist_check():
    """Host sync point: raise if a persistent launch gave up waiting at its grid barrier since the last check (its results are NaN)."""
    if _persist_word is None:
        return
    e = int(_persist_word.item())
    if e:
        _persist_word.zero_()
        raise RuntimeError(
            "persistent RF-sampler launch: a grid-barrier wait expired (status 0x%x) — not every workgroup of the launch was resident "
            "(another process on this GPU, a CU mask, or a second persistent launch); the affected results are NaN.  Set "
            "MINGNATIVE_RF_PERSIST=0 to run the sampler as a launch chain." % e)


def check(rc, what=""):
    if rc != 0:
        msg = lib().mn_last_error().decode(errors="replace")
        raise RuntimeError(f"libmingnative {what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device (or host) address of a torch tensor, or None."""
    return None if t is None else C.c_void_p(t.data_ptr())


def ptr_array(tensors):
    """A ctypes array of device pointers (host-side table the composite structs point to)."""
    arr = (C.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])
    return arr


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
