"""Host-side mirror of the Bailing-MoE decoder stack and its image-generation loop.

Mirrors BailingMoeModel.forward (modeling_bailing_moe.py:1391-1540), the lm_head
(:1604-1620) and BailingMoeForCausalLM.generate_image (:1844-1965) for the hot-path
configuration (rope_scaling=None -> Legacy rotary, first_k_dense_replace=0).
All arithmetic runs in libmingnative; this module owns HBM residency:

  * weights: bf16, reference names; the 64 routed experts and the shared expert of a layer
    are re-packed ONCE at load into two grouped tensors
        w_gate_up [E + S, 2I, H]  (rows 0..I-1 gate_proj, I..2I-1 up_proj)
        w_down    [E + S, H, I]
    where the shared expert (intermediate S*I) is split into S pseudo-experts E..E+S-1 that
    every token selects with weight 1 (their partial down-projections add up to the shared
    expert's output), so routed and shared experts ride the same two grouped launches.
  * KV cache: one preallocated fp32 arena [L, n_seq, 2, n_kv, t_max, hd] instead of the
    reference's torch.cat per layer per step (DynamicCache.update, :789).
  * per-row bookkeeping (cache slot, rotary position, length, key mask) lives in device
    int32/uint8 arrays that the kernels read, so the AR loop never syncs with the host.
  * `weights="fp8"` (mingnative.h section 7; the reference's reduced-byte surface is the `dtype` switch of
    mingunivisioninfer.py:46-70): the packed experts — 15.5 of the stack's 16.2 B parameters, 87-92 % of the bytes a decode step
    streams — are quantised once at load to OCP e4m3 with one power-of-two scale per output row of every expert and streamed
    as bytes; attention, router, norms, embeddings and lm_head stay bf16.  Up to 64 rows per step the codes are decoded inside the
    weight-streaming kernels (the HBM-bound route); the wide route (65..2048 rows) expands a layer's experts into a bf16 scratch first.
"""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import Llm, check, current_stream, lib, ptr, ptr_array
from .configuration import BailingMoeConfig


def rope_tables(head_dim, base, n_pos, device):
    """cos/sin [n_pos, hd/2] fp32 — BailingMoeRotaryEmbeddingLegacy (modeling_bailing_moe.py:213-237);
    a load-time table (emb = cat(freqs, freqs) so only the first half is stored)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(n_pos, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq)
    return freqs.cos().to(device).contiguous(), freqs.sin().to(device).contiguous()


def pack_experts(sd, prefix, cfg):
    """-> (w_gate_up [E+S, 2I, H], w_down [E+S, H, I]) bf16, from reference-named per-expert weights."""
    E, S, I, H = cfg.num_experts, cfg.num_shared_experts or 0, cfg.moe_intermediate_size, cfg.hidden_size
    any_w = sd[f"{prefix}.experts.0.gate_proj.weight"]
    gu = torch.empty(E + S, 2 * I, H, dtype=torch.bfloat16, device=any_w.device)
    dn = torch.empty(E + S, H, I, dtype=torch.bfloat16, device=any_w.device)
    for e in range(E):
        gu[e, :I] = sd[f"{prefix}.experts.{e}.gate_proj.weight"]
        gu[e, I:] = sd[f"{prefix}.experts.{e}.up_proj.weight"]
        dn[e] = sd[f"{prefix}.experts.{e}.down_proj.weight"]
    for s in range(S):
        gu[E + s, :I] = sd[f"{prefix}.shared_experts.gate_proj.weight"][s * I:(s + 1) * I]
        gu[E + s, I:] = sd[f"{prefix}.shared_experts.up_proj.weight"][s * I:(s + 1) * I]
        dn[E + s] = sd[f"{prefix}.shared_experts.down_proj.weight"][:, s * I:(s + 1) * I]
    return gu, dn


def pass_spans(starts, seqs, past, r0, r1):
    """The pieces of stacked sequences that the row range [r0, r1) of one prefill pass holds, as mn_llm_step_spans wants them:
    (cache sequence, first row inside the pass, rows, slots the sequence already has in its cache).  starts: the row offset of every
    sequence in the stack (+ the total as last entry); a sequence cut by a pass boundary continues in the next pass with past > 0."""
    return [(seqs[i], max(starts[i], r0) - r0, min(starts[i + 1], r1) - max(starts[i], r0), past + max(starts[i], r0) - starts[i])
            for i in range(len(starts) - 1) if min(starts[i + 1], r1) > max(starts[i], r0)]


def quantize_layer_experts(ly, weights="fp8", n_shared=0):
    """Weight-only modes: replace a layer's packed bf16 experts by codes — e4m3 ("fp8") or int8 bytes + row scales [E + S, 2I] /
    [E + S, H], or NF4 codes [.., K / 2] + absmax [.., K / 64] ("int4") — in place.
    int8 follows optimum-quanto (one scale per output row of the Linear): the shared expert's down projection is ONE Linear whose rows
    run over all S column blocks of the packed layout, so its S pseudo-experts are quantised together and share their row scales
    (n_shared = S; rows of gate / up are output units of their own, and NF4's 64-element blocks never cross the I-wide column blocks)."""
    for k in ("w_gate_up", "w_down"):
        w = ly[k]
        ly[k], ly[k + "_scale"] = ops.quant_rows(w, weights)
        if k == "w_down" and weights == "int8" and n_shared > 1:
            E = w.shape[0] - n_shared
            I = w.shape[2]
            q, sc = ops.quant_rows(torch.cat([w[E + s] for s in range(n_shared)], dim=1).contiguous(), weights)     # [H, S * I]
            for s in range(n_shared):
                ly[k][E + s] = q[:, s * I:(s + 1) * I]
                ly[k + "_scale"][E + s] = sc
    return ly


MAX_ROWS = 64        # rows of one pass through the weight-streaming decode kernels (four 16-row MFMA tiles)
MAX_ROWS_WIDE = 2048 # rows of one pass through the wide route (every Linear a 256 x 256-tile MFMA GEMM on hi/lo operands)


class BailingMoeDecoder:
    """Weights + KV arena + C-ABI pointer table of the decoder stack."""

    def __init__(self, cfg: BailingMoeConfig, layers, final_norm, word_embeddings=None, lm_head=None,
                 t_max=2048, n_seq=3, n_pos=None, weights="bf16", arith=None, kv_cache=None):
        """layers: list of dicts with bf16 CUDA tensors: ln1, wqkv, wdense, ln2, gate, image_gate (or None),
        w_gate_up, w_down (see pack_experts).  Use `from_state_dict` / `synthetic` to build them.
        weights="fp8": w_gate_up / w_down are e4m3 bytes (uint8) with `w_gate_up_scale` / `w_down_scale` (quantize_layer_experts;
        bf16 experts are quantised here)."""
        assert weights in _lib.WFMT, f"weights={weights!r}: 'bf16', 'fp8', 'int8' or 'int4'"
        # arith="fp8_mfma" (mingnative.h section 8): the LABELLED reduced-arithmetic regime of the wide route — the grouped expert GEMMs on
        # e4m3 activations x the e4m3 expert bytes; needs weights="fp8".  None: fp32-class everywhere (the parity regime)
        assert arith in (None, "fp8_mfma") and (arith is None or weights == "fp8"), "arith='fp8_mfma' needs weights='fp8'"
        self.arith = arith
        self.weights = self.stream_fmt = weights         # the MODEL's mode / what the expert kernels read
        if weights == "int4" and (cfg.hidden_size % 64 or cfg.moe_intermediate_size % 64):
            # NF4 blocks (64 consecutive elements of the flattened matrix) would straddle rows: the experts keep the int4 model's VALUES
            # as bf16 tensors on the bf16 route (ops.fake_quant blocks the flattened tensors like bitsandbytes)
            self.stream_fmt = "bf16"
            for ly in layers:
                if not ly.get("_int4_values"):
                    for k in ("w_gate_up", "w_down"):
                        ly[k] = torch.stack([ops.fake_quant(ly[k][e], weights) for e in range(ly[k].shape[0])])
                    ly["_int4_values"] = True
        elif weights in _lib.W8:
            for ly in layers:
                if ly["w_gate_up"].dtype != torch.uint8:
                    quantize_layer_experts(ly, weights, cfg.num_shared_experts or 0)
        rs = cfg.rope_scaling
        assert rs is None or rs.get("type") == "3D", "Legacy or 3D rotary only (linear / NTK / YaRN are dead code, SURVEY.md a27)"
        # 3D rotary (:413-425, 463-469): same tables, per-frequency choice among the t / h / w position streams.
        # Every caller of the reference passes 2-D positions (= equal streams = Legacy, bit-identical); step() also
        # takes int32 [3, M] positions.
        self.mrope_section = [16, 24, 24] if rs is not None else None
        assert cfg.first_k_dense_replace == 0 and not cfg.use_qkv_bias and not cfg.use_bias
        self.cfg = cfg
        self.layers = layers
        self.final_norm = final_norm
        self.word_embeddings = word_embeddings
        self.lm_head = lm_head
        dev = final_norm.device
        self.device = dev
        self.t_max, self.n_seq = t_max, n_seq
        L, nkv, hd = cfg.num_hidden_layers, cfg.num_key_value_heads, cfg.head_dim
        if kv_cache is not None:                         # share another decoder's arena (same geometry): two weight modes, one conversation state
            assert kv_cache.shape == (L, n_seq, 2, nkv, t_max, hd) and kv_cache.dtype == torch.float32
            self.kv_cache = kv_cache
        else:
            self.kv_cache = torch.zeros(L, n_seq, 2, nkv, t_max, hd, dtype=torch.float32, device=dev)
        self.cos, self.sin = rope_tables(hd, cfg.rope_theta, n_pos or t_max, dev)
        self.n_shared = cfg.num_shared_experts or 0
        keys = ("ln1", "wqkv", "wdense", "ln2", "gate", "image_gate", "w_gate_up", "w_down")
        self._arrays = {k: ptr_array([ly.get(k) for ly in layers]) for k in keys}
        s = Llm()
        s.hidden, s.n_layers, s.n_q, s.n_kv, s.head_dim = cfg.hidden_size, L, cfg.num_attention_heads, nkv, hd
        s.n_experts, s.top_k, s.n_shared_slots = cfg.num_experts, cfg.num_experts_per_tok, self.n_shared
        s.moe_inter, s.norm_topk_prob, s.rms_eps = cfg.moe_intermediate_size, int(cfg.norm_topk_prob), cfg.rms_norm_eps
        for k in keys:
            setattr(s, k, C.cast(self._arrays[k], _lib.PP))
        s.wfmt = _lib.WFMT[self.stream_fmt]
        s.arith = 1 if (arith == "fp8_mfma" and self.stream_fmt == "fp8") else 0
        if self.stream_fmt in _lib.W8:
            for k in ("w_gate_up_scale", "w_down_scale"):
                self._arrays[k] = ptr_array([ly[k] for ly in layers])
                setattr(s, k, C.cast(self._arrays[k], _lib.PP))
        if not cfg.multi_gate:
            s.image_gate = None
        s.final_norm = ptr(final_norm)
        s.cos_tab, s.sin_tab, s.n_pos = ptr(self.cos), ptr(self.sin), self.cos.shape[0]
        if self.mrope_section is not None:
            assert sum(self.mrope_section) == hd // 2
            s.mrope_sec_t, s.mrope_sec_h = self.mrope_section[0], self.mrope_section[1]
        self.struct = s
        self._ws = {}

    # ---- construction -------------------------------------------------------------------
    @classmethod
    def from_state_dict(cls, cfg, sd, prefix="model.", **kw):
        """sd: reference-named bf16 CUDA tensors of BailingMoeForCausalLM (`model.layers.{i}.…`)."""
        layers = []
        for li in range(cfg.num_hidden_layers):
            p = f"{prefix}layers.{li}"
            gu, dn = pack_experts(sd, p + ".mlp", cfg)
            layers.append(dict(
                ln1=sd[p + ".input_layernorm.weight"], wqkv=sd[p + ".attention.query_key_value.weight"],
                wdense=sd[p + ".attention.dense.weight"], ln2=sd[p + ".post_attention_layernorm.weight"],
                gate=sd[p + ".mlp.gate.weight"],
                image_gate=sd.get(p + ".mlp.image_gate.weight") if cfg.multi_gate else None,
                w_gate_up=gu, w_down=dn))
        root = prefix[:-len("model.")] if prefix.endswith("model.") else ""
        lm = sd.get(root + "lm_head.weight")
        return cls(cfg, cls._convert_linears(layers, kw.get("weights", "bf16")), sd[prefix + "norm.weight"],
                   sd.get(prefix + "word_embeddings.weight"), cls._convert_lm_head(lm, kw.get("weights", "bf16")), **kw)

    @staticmethod
    def _convert_linears(layers, weights):
        """Modes that convert every nn.Linear (`_lib.FULL_MODEL`): the attention projections hold the mode's bf16 values (they stay on
        the bf16 kernels: 3 % of a step's bytes); the experts are quantised into codes by the constructor.  Raw weights in, once."""
        if weights in _lib.FULL_MODEL:
            for ly in layers:
                ly["wqkv"], ly["wdense"] = ops.fake_quant(ly["wqkv"], weights), ops.fake_quant(ly["wdense"], weights)
        return layers

    @staticmethod
    def _convert_lm_head(lm, weights):
        return ops.fake_quant(lm, weights) if (lm is not None and weights in _lib.FULL_MODEL) else lm

    @classmethod
    def synthetic(cls, cfg, device, seed=0, with_vocab=True, **kw):
        """Random-init weights of the exact architecture, generated layer by layer on `device`
        straight into the packed layout (no 2x peak), keyed by the reference parameter names."""
        from .configuration import llm_layer_param_shapes
        from .synth import synth_tensor
        layers = []
        for li in range(cfg.num_hidden_layers):
            shapes = llm_layer_param_shapes(cfg, li)
            sd = {k: synth_tensor(k, v, seed, device, torch.bfloat16) for k, v in shapes.items()}
            p = f"model.layers.{li}"
            gu, dn = pack_experts(sd, p + ".mlp", cfg)
            layers.append(dict(
                ln1=sd[p + ".input_layernorm.weight"], wqkv=sd[p + ".attention.query_key_value.weight"],
                wdense=sd[p + ".attention.dense.weight"], ln2=sd[p + ".post_attention_layernorm.weight"],
                gate=sd[p + ".mlp.gate.weight"],
                image_gate=sd.get(p + ".mlp.image_gate.weight") if cfg.multi_gate else None,
                w_gate_up=gu, w_down=dn))
            if kw.get("weights") in _lib.W8:        # layer by layer: the bf16 experts of one layer at a time
                if not (kw["weights"] == "int4" and (cfg.hidden_size % 64 or cfg.moe_intermediate_size % 64)):   # (else: __init__ keeps values)
                    quantize_layer_experts(layers[-1], kw["weights"], cfg.num_shared_experts or 0)
                cls._convert_linears(layers[-1:], kw["weights"])
            del sd, gu, dn
        H, V = cfg.hidden_size, cfg.vocab_size
        fn = synth_tensor("model.norm.weight", (H,), seed, device, torch.bfloat16)
        emb = lm = None
        if with_vocab:
            emb = synth_tensor("model.word_embeddings.weight", (V, H), seed, device, torch.bfloat16)
            lm = cls._convert_lm_head(synth_tensor("lm_head.weight", (V, H), seed, device, torch.bfloat16), kw.get("weights", "bf16"))
        return cls(cfg, layers, fn, emb, lm, **kw)

    def view(self, t_max, n_seq, n_pos=None):
        """A second decoder on the SAME weights with its own KV arena / rotary tables (e.g. long-context understanding next to an
        image batch's arena)."""
        return BailingMoeDecoder(self.cfg, self.layers, self.final_norm, self.word_embeddings, self.lm_head, t_max=t_max, n_seq=n_seq,
                                 n_pos=n_pos, weights=self.weights, arith=self.arith)

    def to_fp8(self, t_max=None, n_seq=None, weights="fp8", arith=None, share_kv=False):
        """A second decoder whose experts are 8-bit copies of this one's — e4m3 (default) or int8 (weights="int8") — (attention /
        router / vocabulary tensors shared; this bf16 decoder stays usable): + 0.5 bytes per expert parameter of HBM.
        arith="fp8_mfma": the labelled fp8-MFMA regime of the wide route (e4m3 only).  share_kv: use THIS decoder's KV arena (same
        t_max / n_seq) instead of allocating a second one."""
        assert self.weights == "bf16" and weights in _lib.W8
        layers = self._convert_linears([quantize_layer_experts(dict(ly), weights, self.n_shared) for ly in self.layers], weights)
        return BailingMoeDecoder(self.cfg, layers, self.final_norm, self.word_embeddings, self._convert_lm_head(self.lm_head, weights),
                                 t_max=t_max or self.t_max, n_seq=n_seq or self.n_seq, weights=weights, arith=arith,
                                 kv_cache=self.kv_cache if share_kv else None)

    def weight_bytes_active(self, distinct_experts_per_layer):
        """Weight bytes one decode step streams: attention + router (bf16) + the distinct routed and the shared experts
        (bf16, or e4m3 bytes + fp32 row scales)."""
        cfg = self.cfg
        H, I = cfg.hidden_size, cfg.moe_intermediate_size
        attn = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim * H + H * cfg.num_attention_heads * cfg.head_dim
        per_expert = (2 * 3 * I * H if self.stream_fmt == "bf16" else
                      (3 * I * H // 2 + 3 * I * H // 64 * 4 if self.stream_fmt == "int4" else 3 * I * H + (2 * I + H) * 4))
        return cfg.num_hidden_layers * (2 * (attn + cfg.num_experts * H) + (distinct_experts_per_layer + self.n_shared) * per_expert)

    def dequantized_state_dict(self, prefix="model."):
        """fp8 mode: {reference parameter name: bf16 tensor} of every expert / shared-expert projection as the kernels see it —
        the inverse of pack_experts on the dequantised packed tensors.  What the oracle is fed in the parity tests."""
        cfg = self.cfg
        E, S, I = cfg.num_experts, self.n_shared, cfg.moe_intermediate_size
        out = {}
        for li in range(cfg.num_hidden_layers):
            gu, dn = self.dequantized_experts(li)
            p = f"{prefix}layers.{li}.mlp"
            for e in range(E):
                out[f"{p}.experts.{e}.gate_proj.weight"] = gu[e, :I]
                out[f"{p}.experts.{e}.up_proj.weight"] = gu[e, I:]
                out[f"{p}.experts.{e}.down_proj.weight"] = dn[e]
            if S:
                out[f"{p}.shared_experts.gate_proj.weight"] = torch.cat([gu[E + s, :I] for s in range(S)], 0)
                out[f"{p}.shared_experts.up_proj.weight"] = torch.cat([gu[E + s, I:] for s in range(S)], 0)
                out[f"{p}.shared_experts.down_proj.weight"] = torch.cat([dn[E + s] for s in range(S)], 1)
            if self.weights in _lib.FULL_MODEL:      # every nn.Linear is converted: the attention projections hold the mode's values
                out[f"{prefix}layers.{li}.attention.query_key_value.weight"] = self.layers[li]["wqkv"]
                out[f"{prefix}layers.{li}.attention.dense.weight"] = self.layers[li]["wdense"]
        if self.weights in _lib.FULL_MODEL and self.lm_head is not None:
            out[(prefix[:-len("model.")] if prefix.endswith("model.") else "") + "lm_head.weight"] = self.lm_head
        return out

    def dequantized_experts(self, li):
        """fp8 mode: layer li's packed experts as the kernels see them (e4m3 * row scale, exact in bf16): (w_gate_up, w_down)."""
        assert self.weights in _lib.W8
        ly = self.layers[li]
        if self.stream_fmt not in _lib.W8:
            return ly["w_gate_up"], ly["w_down"]
        return (ops.dequant_rows(ly["w_gate_up"], ly["w_gate_up_scale"], self.weights), ops.dequant_rows(ly["w_down"], ly["w_down_scale"], self.weights))

    # ---- stepping -------------------------------------------------------------------------
    def _workspace(self, rows):
        key = (rows, torch.cuda.current_stream().cuda_stream)      # one scratch area per stream: groups overlap
        if key not in self._ws:
            n = lib().mn_llm_workspace_bytes(C.byref(self.struct), rows, self.t_max)
            self._ws[key] = torch.empty(n, dtype=torch.uint8, device=self.device)
        return self._ws[key]

    def step(self, x, row_seq, row_slot, row_pos, row_len, key_mask=None, image_mask=None, out=None, rows=None,
             x_row_div=1, distinct_sequences=False, spans=None):
        """One pass of the 28-layer stack over M <= 64 rows (weight-streaming kernels) or 65..2048 rows (wide route).
        x fp32 [M,H]; or [1,H] with rows=M to broadcast; or [M / x_row_div, H] with rows=M when the x_row_div CFG
        rows of an image share one embedding.  int32 device arrays per row; key_mask uint8 [M, >=len].
        distinct_sequences: every row is a different cache sequence and row_len == row_slot + 1 (a decode step, not a prefill
        chunk) — RoPE and the K / V append then ride the attention launch (mn_llm_step_ex; same results).
        spans: a PREFILL chunk described as whole spans of cache sequences — a list of (seq, first row, length, past): rows
        [r0, r0 + len) are the slots [past, past + len) of sequence seq, in order, no key mask (mn_llm_step_spans: the wide route
        then runs tiled flash attention on hi/lo operands instead of row-by-row decode attention; same results).
        Returns the post-final-norm hidden states [M,H] fp32."""
        M = rows or x.shape[0]
        ldx = 0 if (rows is not None and x.shape[0] == 1) else x.stride(0)
        assert x.shape[0] in (1, M, M // x_row_div)
        assert x.dtype == torch.float32 and x.is_cuda and x.stride(-1) == 1
        for t in (row_seq, row_slot, row_pos, row_len):
            assert t.dtype == torch.int32 and t.is_cuda and t.numel() >= M
        if self.mrope_section is not None:
            if row_pos.dim() == 1:                       # 2-D ids of the reference's callers: t = h = w
                row_pos = row_pos[:M].unsqueeze(0).expand(3, M).contiguous()
            assert row_pos.shape == (3, M) and row_pos.is_contiguous()
        else:
            assert row_pos.dim() == 1
        if key_mask is not None:
            assert key_mask.dtype == torch.uint8 and key_mask.is_cuda and key_mask.shape[0] >= M
        if image_mask is not None:
            assert image_mask.dtype == torch.uint8 and image_mask.is_cuda and image_mask.numel() >= M
        if out is None:
            out = torch.empty(M, self.cfg.hidden_size, dtype=torch.float32, device=self.device)
        ws = self._workspace(M)
        if spans is not None and M <= MAX_ROWS:
            spans = None                                  # <= 64 rows: the span table is not read (no wide route) — skip its blocking host-to-device copy
        if spans is not None:
            assert key_mask is None and x_row_div == 1 and x.shape[0] == M and not distinct_sequences
            assert sum(n for _, _, n, _ in spans) == M and all(p + n <= self.t_max for _, _, n, p in spans)
            tab = torch.tensor([list(map(int, sp)) for sp in spans], dtype=torch.int32).to(self.device)
            check(lib().mn_llm_step_spans(C.byref(self.struct), ptr(x), ldx, M, ptr(image_mask), ptr(row_seq), ptr(row_slot), ptr(row_pos),
                                          ptr(row_len), ptr(self.kv_cache), self.n_seq, self.t_max, ptr(tab), len(spans),
                                          max(n for _, _, n, _ in spans), ptr(out), ptr(ws), ws.numel(), current_stream()),
                  "mn_llm_step_spans")
            return out
        check(lib().mn_llm_step_ex(C.byref(self.struct), ptr(x), ldx, x_row_div, M, ptr(image_mask), ptr(row_seq), ptr(row_slot),
                                   ptr(row_pos), ptr(row_len), ptr(key_mask),
                                   0 if key_mask is None else key_mask.stride(0), ptr(self.kv_cache), self.n_seq,
                                   self.t_max, ptr(out), ptr(ws), ws.numel(), 1 if distinct_sequences else 0, current_stream()),
              "mn_llm_step")
        return out

    def prefill(self, embeds, seq=0, past=0, image_mask=None, chunk=MAX_ROWS):
        """Causal prefill of ONE sequence by chunks of <= 64 rows through the decode kernels
        (each row m of a chunk attends cache[0 : past + m + 1]).  embeds fp32 [T,H].
        Returns the hidden states [T,H]."""
        T = embeds.shape[0]
        assert past + T <= self.t_max
        outs = []
        for c0 in range(0, T, chunk):
            m = min(chunk, T - c0)
            slot = torch.arange(past + c0, past + c0 + m, dtype=torch.int32, device=self.device)
            seqs = torch.full((m,), seq, dtype=torch.int32, device=self.device)
            im = None if image_mask is None else image_mask[c0:c0 + m].to(self.device, torch.uint8).contiguous()
            outs.append(self.step(embeds[c0:c0 + m].contiguous(), seqs, slot, slot, slot + 1, None, im))
        return torch.cat(outs, 0)

    def max_rows(self):
        """Rows one step() accepts: 64, or 2048 when the wide route applies to this configuration."""
        return int(lib().mn_llm_max_rows(C.byref(self.struct)))

    def prefill_many(self, embeds, seqs, past=0):
        """Causal prefill of B sequences of the same length in lock-step (the prompts of an image batch): embeds fp32
        [B, T, H]; sequence b goes to cache sequence seqs[b].  Rows of all sequences share each pass through the stack
        (<= max_rows() rows per pass); row (b, t) attends cache[seqs[b]][0 : past + t + 1].  Returns hidden [B, T, H]."""
        B, T, H = embeds.shape
        assert past + T <= self.t_max and len(seqs) == B
        dev = self.device
        x = embeds.reshape(B * T, H).contiguous()
        seq = torch.tensor(list(seqs), dtype=torch.int32, device=dev).repeat_interleave(T)
        slot = (torch.arange(T, dtype=torch.int32, device=dev) + past).repeat(B)
        out = torch.empty(B * T, H, dtype=torch.float32, device=dev)
        step = self.max_rows()
        for r0 in range(0, B * T, step):
            r1 = min(B * T, r0 + step)
            sl = slot[r0:r1].contiguous()
            self.step(x[r0:r1], seq[r0:r1].contiguous(), sl, sl, (sl + 1).contiguous(), None, None, out=out[r0:r1])
        return out.reshape(B, T, H)

    def prefill_wide(self, embeds, seq=0, past=0, image_mask=None):
        """fp32-class causal prefill of ONE long prompt: all its rows go through the stack in passes of <= max_rows() rows on the
        wide route (every Linear a gemm256 launch on hi/lo operands — 2^-17 activations, the decode path's numerics —, masked GQA
        against the fp32 arena, image-gate routing on the rows flagged by image_mask).  Same results as `prefill` to summation
        order, at MFMA rate; `prefill_mfma` is the bf16-activation form (2x fewer MFMA passes, ~1e-2 of the fp32 result).
        embeds fp32 [T,H].  Returns the final-norm hidden states [T,H]."""
        T = embeds.shape[0]
        assert past + T <= self.t_max
        dev = self.device
        step = self.max_rows()
        out = torch.empty(T, embeds.shape[1], dtype=torch.float32, device=dev)
        x = embeds.to(dev, torch.float32).contiguous()
        im = None if image_mask is None else image_mask.reshape(-1).to(dev, torch.uint8).contiguous()
        for c0 in range(0, T, step):
            m = min(step, T - c0)
            slot = torch.arange(past + c0, past + c0 + m, dtype=torch.int32, device=dev)
            seqs = torch.full((m,), seq, dtype=torch.int32, device=dev)
            self.step(x[c0:c0 + m], seqs, slot, slot, slot + 1, None, None if im is None else im[c0:c0 + m], out=out[c0:c0 + m],
                      spans=[(seq, 0, m, past + c0)])
        return out

    def ensure_sequences(self, n_seq):
        """Grow the KV arena to hold `n_seq` cache sequences (contents of the existing ones are kept).  Every sequence costs
        L * 2 * n_kv * t_max * hd * 4 bytes (16B-A3B at t_max = 4096: 470 MB) and the old arena stays live during the copy:
        raises MemoryError naming the bytes instead of letting the allocator fail half-way."""
        if n_seq <= self.n_seq:
            return
        per_seq = self.kv_cache[:, 0].numel() * 4
        need = n_seq * per_seq
        free, _ = torch.cuda.mem_get_info(self.device)
        if need > free:
            raise MemoryError(f"KV arena for {n_seq} sequences at t_max = {self.t_max} needs {need / 2**30:.1f} GiB "
                              f"({per_seq / 2**20:.0f} MiB per sequence), {free / 2**30:.1f} GiB free: "
                              "use a smaller batch or build the model with a smaller t_max")
        kv = torch.zeros((self.kv_cache.shape[0], n_seq) + tuple(self.kv_cache.shape[2:]), dtype=torch.float32, device=self.device)
        kv[:, :self.n_seq] = self.kv_cache
        self.kv_cache, self.n_seq = kv, n_seq

    def copy_sequence(self, src, dst, n):
        """Cache sequence dst[:n] = src[:n] (KV replicate of generate_image, modeling_bailing_moe.py:1891-1902)."""
        self.kv_cache[:, dst, :, :, :n].copy_(self.kv_cache[:, src, :, :, :n])

    def release_sequences(self, n_keep):
        """Shrink the KV arena back to its first `n_keep` sequences (after a batch call: the batch sequences are dead)."""
        if n_keep >= self.n_seq:
            return
        self.kv_cache = self.kv_cache[:, :n_keep].clone()
        self.n_seq = n_keep

    def prefill_ragged(self, embeds_list, seqs, past=0, image_masks=None):
        """Causal prefill of several sequences of DIFFERENT lengths in shared passes through the stack: embeds_list[i] fp32
        [T_i, H] goes to cache sequence seqs[i] from slot `past`; image_masks[i] (optional, bool / uint8 [T_i]) flags the rows
        routed by the image gate.  fp32-class (decode-path numerics).  Returns the last-token hidden state of each, [B, H]."""
        dev = self.device
        lens = [int(e.shape[0]) for e in embeds_list]
        assert len(seqs) == len(lens) and past + max(lens) <= self.t_max
        x = torch.cat([e.to(dev, torch.float32) for e in embeds_list], 0).contiguous()
        seq = torch.cat([torch.full((n,), s, dtype=torch.int32) for n, s in zip(lens, seqs)]).to(dev)
        slot = torch.cat([torch.arange(past, past + n, dtype=torch.int32) for n in lens]).to(dev)
        out = torch.empty(x.shape[0], x.shape[1], dtype=torch.float32, device=dev)
        im = None
        if image_masks is not None and any(m is not None for m in image_masks):
            im = torch.cat([torch.zeros(n, dtype=torch.uint8, device=dev) if m is None else m.reshape(-1).to(dev, torch.uint8)
                            for m, n in zip(image_masks, lens)]).contiguous()
        step = self.max_rows()
        starts = [0]
        for n in lens:
            starts.append(starts[-1] + n)
        for r0 in range(0, x.shape[0], step):
            r1 = min(x.shape[0], r0 + step)
            sl = slot[r0:r1].contiguous()
            spans = pass_spans(starts, seqs, past, r0, r1)
            self.step(x[r0:r1], seq[r0:r1].contiguous(), sl, sl, (sl + 1).contiguous(), None, None if im is None else im[r0:r1],
                      out=out[r0:r1], spans=spans)
        last = torch.tensor(lens).cumsum(0) - 1
        return out[last.to(dev)]

    def prefill_mfma(self, embeds, seq=0, past=0, image_mask=None, positions=None, key_mask=None):
        """Causal prefill of ONE sequence for long prompts on the bf16 MFMA path (see prefill_mfma_many).  embeds fp32 [T,H].
        Returns the final-norm hidden state of the LAST token [1,H] fp32 (what the next-token logits need)."""
        return self.prefill_mfma_many([embeds], [seq], past=past, image_masks=None if image_mask is None else [image_mask],
                                      positions=None if positions is None else [positions],
                                      key_masks=None if key_mask is None else [key_mask])

    def prefill_mfma_many(self, embeds_list, seqs, past=0, image_masks=None, positions=None, key_masks=None):
        """Causal prefill of one or several sequences (lengths may differ) on the bf16 MFMA path, their tokens stacked into
        one row block: per layer
        RMSNorm -> QKV GEMM -> [per sequence: RoPE + KV append -> GQA flash attention (hd 128)] -> dense GEMM (+residual) ->
        RMSNorm -> gate GEMM -> top-k -> expert sort -> grouped gate/up GEMM -> SwiGLU -> grouped down GEMM ->
        weighted combine (+residual).  Every GEMM and the expert grouping see all sequences' tokens at once (the expert weights
        are read once per call instead of once per sequence; 8 x 1058 tokens put ~800 rows in front of every expert).
        fp32 residual stream, bf16 GEMM operands (like the reference's autocast path); the KV arena stays fp32.
        embeds_list[i] fp32 [T_i,H] goes to cache sequence seqs[i] from slot `past`.  Returns the final-norm hidden state of
        the LAST token of each sequence [B,H] fp32 (what the next-token logits need)."""
        import math
        if self.stream_fmt != "bf16":
            # bf16-autocast prefill kernels read bf16 expert weights: a quantised stack takes the shared passes through the step kernels instead
            assert positions is None and key_masks is None, "fp8 weight mode: default positions / masks only"
            return self.prefill_ragged(embeds_list, seqs, past=past, image_masks=image_masks)
        cfg, L_ = self.cfg, lib()
        lens = [int(e.shape[0]) for e in embeds_list]
        H = embeds_list[0].shape[1]
        assert len(seqs) == len(lens) and past + max(lens) <= self.t_max and cfg.head_dim == 128
        dev = self.device
        nq, nkv, hd, I = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.moe_intermediate_size
        E, k, S = cfg.num_experts, cfg.num_experts_per_tok, self.n_shared
        n_slot, G = k + S, E + S
        st = current_stream()
        h = torch.cat([e.to(dev, torch.float32) for e in embeds_list], 0).contiguous()
        if len(embeds_list) == 1:
            h = h.clone()
        T = h.shape[0]
        r0s = [sum(lens[:i]) for i in range(len(lens))]
        pos = [torch.arange(past, past + n, dtype=torch.int32, device=dev) if positions is None or positions[i] is None
               else positions[i].to(dev, torch.int32).contiguous() for i, n in enumerate(lens)]
        im = None
        if image_masks is not None and any(m is not None for m in image_masks):
            im = torch.cat([torch.zeros(n, dtype=torch.uint8, device=dev) if m is None else m.reshape(-1).to(dev, torch.uint8)
                            for m, n in zip(image_masks, lens)]).contiguous()
        kms = [None if key_masks is None or key_masks[i] is None else key_masks[i].to(dev, torch.uint8).contiguous()
               for i in range(len(lens))]
        seq_tab = km_all = None
        if nq == 4 * nkv and len(lens) > 1:
            seq_tab = torch.tensor([[int(s_), r0, n] for s_, r0, n in zip(seqs, r0s, lens)], dtype=torch.int32).to(dev)
            if any(m is not None for m in kms):
                km_all = torch.ones(len(lens), self.t_max, dtype=torch.uint8, device=dev)
                for i, m in enumerate(kms):
                    if m is not None:
                        km_all[i, :m.numel()] = m.reshape(-1)
        bf = torch.bfloat16
        xn = torch.empty(T, H, dtype=bf, device=dev)
        qkv = torch.empty(T, (nq + 2 * nkv) * hd, dtype=torch.float32, device=dev)
        qb = torch.empty(T, nq * hd, dtype=bf, device=dev)
        att = torch.empty(T, nq * hd, dtype=bf, device=dev)
        lg = torch.empty(2, T, E, dtype=torch.float32, device=dev)
        ti = torch.empty(T, n_slot, dtype=torch.int32, device=dev)
        tw = torch.empty(T, n_slot, dtype=torch.float32, device=dev)
        cnt = torch.empty(G, dtype=torch.int32, device=dev)
        off = torch.empty(G + 1, dtype=torch.int32, device=dev)
        perm = torch.empty(T * n_slot, dtype=torch.int32, device=dev)
        slot_of = torch.empty(T * n_slot, dtype=torch.int32, device=dev)
        hm = torch.empty(T * n_slot, I, dtype=bf, device=dev)
        yg = torch.empty(T * n_slot, H, dtype=torch.float32, device=dev)
        # expert GEMMs: gemm256 over the live 256-row tiles of the expert-sorted pairs (gather while staging, SwiGLU in the
        # epilogue) when the shapes allow, else the 128-tile grouped kernel with separate gather / SwiGLU passes
        tiled = (H % 64 == 0 and I % 64 == 0 and T * n_slot * max(H, I) * 2 < (1 << 32) and 2 * I * H * 2 < (1 << 32))
        if tiled:
            max_mt = T * n_slot // 256 + G
            tile_g = torch.empty(max_mt, dtype=torch.int32, device=dev)
            tile_m0 = torch.empty(max_mt, dtype=torch.int32, device=dev)
            n_tiles = torch.empty(1, dtype=torch.int32, device=dev)
        else:
            xg = torch.empty(T * n_slot, H, dtype=bf, device=dev)
            gu = torch.empty(T * n_slot, 2 * I, dtype=bf, device=dev)
        pos_all = torch.cat(pos).contiguous() if seq_tab is not None else None
        fuse_norm = H % 4 == 0 and H <= 4096          # the expert combine also applies the next layer's RMSNorm
        xn_ready = False
        for li, ly in enumerate(self.layers):
            if not xn_ready:
                check(L_.mn_rmsnorm_bf16(ptr(h), H, ptr(ly["ln1"]), cfg.rms_norm_eps, ptr(xn), H, T, H, st), "mn_rmsnorm_bf16")
            ops.gemm_bf16(xn, ly["wqkv"], None, "f32", out=qkv)
            if seq_tab is None:
                for i, n in enumerate(lens):
                    kv_seq, r0 = self.kv_cache[li, seqs[i]], r0s[i]
                    check(L_.mn_rope_kv_prefill(ptr(qkv[r0:]), qkv.stride(0), n, nq, nkv, hd, ptr(self.cos), ptr(self.sin), ptr(pos[i]),
                                                past, 1.0 / math.sqrt(hd), ptr(qb[r0:]), ptr(kv_seq), self.t_max, st), "mn_rope_kv_prefill")
                    check(L_.mn_attn_prefill_gqa_hd128(ptr(qb[r0:]), ptr(kv_seq), self.t_max, nq, nkv, past, n, ptr(kms[i]),
                                                       ptr(att[r0:]), st), "mn_attn_prefill_gqa_hd128")
            else:                        # all spans in one launch each (prefill_ops.hip, flash_prefill.hip)
                check(L_.mn_rope_kv_prefill_spans(ptr(qkv), qkv.stride(0), nq, nkv, hd, ptr(self.cos), ptr(self.sin), ptr(pos_all), past,
                                                  1.0 / math.sqrt(hd), ptr(qb), ptr(self.kv_cache[li]), self.t_max, ptr(seq_tab),
                                                  len(lens), max(lens), st), "mn_rope_kv_prefill_spans")
                check(L_.mn_flash_prefill_gqa_hd128(ptr(qb), ptr(self.kv_cache[li]), self.t_max, nq, nkv, past, ptr(seq_tab), len(lens),
                                                    max(lens), ptr(km_all), 0 if km_all is None else km_all.stride(0), ptr(att), st),
                      "mn_flash_prefill_gqa_hd128")
            ops.gemm_bf16(att, ly["wdense"], None, "f32_resid", out=h)
            check(L_.mn_rmsnorm_bf16(ptr(h), H, ptr(ly["ln2"]), cfg.rms_norm_eps, ptr(xn), H, T, H, st), "mn_rmsnorm_bf16")
            ops.gemm_bf16(xn, ly["gate"], None, "f32", out=lg[0])
            use_img = im is not None and ly.get("image_gate") is not None
            if use_img:
                ops.gemm_bf16(xn, ly["image_gate"], None, "f32", out=lg[1])
            check(L_.mn_moe_topk_logits(ptr(lg[0]), ptr(lg[1]) if use_img else None, ptr(im) if use_img else None, T, E, k,
                                        int(cfg.norm_topk_prob), S, ptr(ti), ptr(tw), st), "mn_moe_topk_logits")
            if tiled:
                check(L_.mn_moe_sort_tiles(ptr(ti), T, n_slot, G, ptr(cnt), ptr(off), ptr(perm), ptr(slot_of), 256, ptr(tile_g),
                                           ptr(tile_m0), ptr(n_tiles), st), "mn_moe_sort_tiles")
                check(L_.mn_gemm256_grouped_tiles(ptr(xn), H, 0, T, ptr(perm), ptr(ly["w_gate_up"]), H, 2 * I * H, ptr(off), ptr(cnt), G,
                                                  ptr(tile_g), ptr(tile_m0), ptr(n_tiles), max_mt, ptr(hm), I, 0, I, H, 6, st),
                      "mn_gemm256_grouped_tiles(gate_up)")
                check(L_.mn_gemm256_grouped_tiles(ptr(hm), I, 0, T * n_slot, None, ptr(ly["w_down"]), I, H * I, ptr(off), ptr(cnt), G,
                                                  ptr(tile_g), ptr(tile_m0), ptr(n_tiles), max_mt, ptr(yg), H, 0, H, I, 0, st),
                      "mn_gemm256_grouped_tiles(down)")
            else:
                check(L_.mn_moe_sort(ptr(ti), T, n_slot, G, ptr(cnt), ptr(off), ptr(perm), ptr(slot_of), st), "mn_moe_sort")
                check(L_.mn_gather_rows_bf16(ptr(xn), H, ptr(perm), ptr(xg), H, T * n_slot, H, st), "mn_gather_rows_bf16")
                check(L_.mn_gemm_bf16_grouped(ptr(xg), H, ptr(ly["w_gate_up"]), H, 2 * I * H, ptr(off), ptr(cnt), G, ptr(gu), 2 * I,
                                              T, 2 * I, H, ops.GEMM_EPI["bf16"], st), "mn_gemm_bf16_grouped(gate_up)")
                check(L_.mn_swiglu_bf16(ptr(gu), 2 * I, ptr(hm), I, T * n_slot, I, st), "mn_swiglu_bf16")
                check(L_.mn_gemm_bf16_grouped(ptr(hm), I, ptr(ly["w_down"]), I, H * I, ptr(off), ptr(cnt), G, ptr(yg), H, T, H, I,
                                              ops.GEMM_EPI["f32"], st), "mn_gemm_bf16_grouped(down)")
            if fuse_norm and li + 1 < len(self.layers):
                check(L_.mn_moe_combine_norm(ptr(yg), ptr(slot_of), ptr(tw), n_slot, ptr(h), H, ptr(self.layers[li + 1]["ln1"]),
                                             cfg.rms_norm_eps, ptr(xn), H, T, H, st), "mn_moe_combine_norm")
                xn_ready = True
            else:
                check(L_.mn_moe_combine(ptr(yg), H, ptr(slot_of), ptr(tw), n_slot, ptr(h), H, T, H, st), "mn_moe_combine")
                xn_ready = False
        last = torch.tensor([r0 + n - 1 for r0, n in zip(r0s, lens)], dtype=torch.long, device=dev)
        return self._final_norm_rows(h[last].contiguous())

    def _final_norm_rows(self, x):
        """model.norm on the prefill path (modeling_bailing_moe.py:1521): bf16 output like the rest of that path."""
        H = x.shape[1]
        y16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        check(lib().mn_rmsnorm_bf16(ptr(x), H, ptr(self.final_norm), self.cfg.rms_norm_eps, ptr(y16), H, x.shape[0], H,
                                    current_stream()), "mn_rmsnorm_bf16")
        return ops.bf16_to_f32(y16)

    def logits(self, hidden):
        """lm_head -> fp32 logits (compute_logit, :1604-1620, norm_head=False)."""
        if hidden.shape[0] > MAX_ROWS and lib().mn_gemm256_supported(hidden.shape[1], 8, self.lm_head.stride(0), self.lm_head.shape[0],
                                                                     hidden.shape[0], self.lm_head.shape[0], hidden.shape[1]):
            return ops.gemm256(ops.split_hilo(hidden.contiguous()), self.lm_head, None, "f32")    # many rows: one MFMA GEMM
        outs = [ops.skinny_gemm(hidden[i:i + 8].contiguous(), self.lm_head) for i in range(0, hidden.shape[0], 8)]
        return torch.cat(outs, 0)

    def greedy(self, hidden):
        """Greedy pick of every row: lm_head + arg-max in ONE C call (mn_lmhead_argmax: compute_logit, :1604-1620, + the argmax
        of greedy decoding; ties -> lowest id like torch.argmax).  hidden fp32 [M, H] -> int64 [M] on the device."""
        return ops.lmhead_argmax(hidden.contiguous(), self.lm_head)[0]

    def sample(self, hidden, u, temperature=1.0, top_k=50, top_p=1.0):
        """Sampled pick of every row (HF generate's `do_sample` branch: temperature -> top-k -> top-p -> one draw, at the caller's
        uniforms u fp32 [M]): lm_head logits + mn_sample_logits.  hidden fp32 [M, H] -> int64 [M] on the device."""
        st = torch.empty(hidden.shape[0], dtype=torch.int32, device=hidden.device)
        ids = ops.sample_logits(self.logits(hidden.contiguous()), u, temperature, top_k, top_p, status=st)
        # rows whose kept set was cut at the sampler's 2048 candidates (ops.SAMPLE_*): the status words are only KEPT here — no kernel on
        # the latency-critical decode loop — and OR-ed at the caller's next host sync (`sampling_truncated()`)
        pend = getattr(self, "_sample_status", None)
        if pend is None:
            pend = self._sample_status = []
        pend.append(st)
        if len(pend) >= 512:                       # bound the list on very long generations: one reduction per 512 tokens
            self._sample_status = [torch.stack([(torch.cat(pend) & 1).amax(), (torch.cat(pend) & 2).amax()]).to(torch.int32)]
        return ids

    def sampling_truncated(self):
        """Host sync: ops.SAMPLE_* bits seen by `sample` since the last call (0 = every draw was HF's distribution exactly)."""
        pend = getattr(self, "_sample_status", None)
        self._sample_status = None
        if not pend:
            return 0
        allst = torch.cat(pend)
        return int((allst & 1).amax() | (allst & 2).amax())

    def embed(self, ids):
        """word_embeddings lookup -> fp32 rows (gather = memory plumbing)."""
        return ops.bf16_to_f32(self.word_embeddings[ids.reshape(-1)])


def build_cfg_rows(attention_mask, uncond_attention_mask, text_uncond_attention_mask):
    """CFG row construction of generate_image (modeling_bailing_moe.py:1867-1889) — host integer logic.
    Masks are [1, T*] integer tensors (CPU or GPU); returns the stacked [rows, T] mask."""
    assert attention_mask.shape[0] == 1
    am = attention_mask
    if uncond_attention_mask is not None:
        n_c, n_u = am.shape[1], uncond_attention_mask.shape[1]
        if n_u < n_c:
            uncond_attention_mask = torch.cat((uncond_attention_mask, am[:, n_u:]), dim=1)
        am = torch.cat((am, uncond_attention_mask), dim=0)
    if text_uncond_attention_mask is not None and int(text_uncond_attention_mask.sum()) > 0:
        n_c, n_u = am.shape[1], text_uncond_attention_mask.shape[1]
        if n_u < n_c:
            text_uncond_attention_mask = torch.cat((text_uncond_attention_mask, am[0:1, n_u:]), dim=1)
        if int((text_uncond_attention_mask == uncond_attention_mask).sum()) != uncond_attention_mask.numel():
            am = torch.cat((am, text_uncond_attention_mask), dim=0)
    return am


class ImageGenState:
    """Device-resident bookkeeping of one (possibly batched) generate_image call.
    Rows are image-major: row r belongs to image r // rpi and is its (r % rpi)-th CFG row; the KV sequence
    of row r is `seq_base + r`."""

    def __init__(self, dec: BailingMoeDecoder, am_rows_list, past_lens, seq_base=0):
        dev = dec.device
        rpi = am_rows_list[0].shape[0]
        B = len(am_rows_list)
        rows = B * rpi
        assert seq_base + rows <= dec.n_seq, f"KV arena holds {dec.n_seq} sequences, need {seq_base + rows}"
        self.rows, self.rpi, self.n_images = rows, rpi, B
        km = torch.ones(rows, dec.t_max, dtype=torch.uint8)               # generated tokens are always attended
        pos, slot = [], []
        for i, (am, past) in enumerate(zip(am_rows_list, past_lens)):
            assert am.shape == (rpi, past + 1)
            km[i * rpi:(i + 1) * rpi, :past + 1] = am.to(torch.uint8)
            pos.append((am.long().cumsum(-1) - 1)[:, -1])                   # modeling_bailing_moe.py:1905-1907
            slot.append(torch.full((rpi,), past, dtype=torch.long))
        self.key_mask = km.to(dev)
        self.row_pos = torch.cat(pos).to(torch.int32).to(dev).contiguous()
        self.row_slot = torch.cat(slot).to(torch.int32).to(dev).contiguous()
        self.row_len = (self.row_slot + 1).contiguous()
        self.row_seq = (torch.arange(rows, dtype=torch.int32) + seq_base).to(dev)

    def advance(self):
        check(lib().mn_rows_advance(ptr(self.row_slot), ptr(self.row_pos), ptr(self.row_len), self.rows, 1,
                                    current_stream()), "mn_rows_advance")


def split_groups(n_images, n_groups):
    """Image ranges [(lo, hi), ...] of the lock-step groups: ceil(B / n_groups) images per group and only as many groups as
    that needs (5 images in 4 groups -> 2 + 2 + 1, never an empty group)."""
    assert 1 <= n_groups <= n_images
    per = (n_images + n_groups - 1) // n_groups
    return [(lo, min(n_images, lo + per)) for lo in range(0, n_images, per)]


_STREAM_POOL = {}


def _group_streams(device, n):
    """n side streams for lock-step groups (created once per device and reused)."""
    import os
    import warnings
    if n > 2 and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 2 * n:
        warnings.warn(f"{n} lock-step groups want GPU_MAX_HW_QUEUES >= {2 * n} (set before HIP initialises; the package sets 8 on "
                      "import): with fewer hardware queues some groups serialise (4 groups: 1447 vs 1686 tokens/s)")
    pool = _STREAM_POOL.setdefault(str(device), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


class _GroupRun:
    """One lock-step group of images of a generate_images call: its rows, outputs and per-token work."""

    def __init__(self, dec, rf, tok, ams, past_lens, noises, seq_base, start_embed, sampler_kw, skip_last_sample):
        cfg = dec.cfg
        self.dec, self.rf, self.tok, self.start_embed = dec, rf, tok, start_embed
        self.kw, self.skip_last = sampler_kw, skip_last_sample
        self.B, self.rpi = len(ams), ams[0].shape[0]
        self.rows = self.B * self.rpi
        self.n_tok = cfg.num_image_tokens_for_gen
        dev = dec.device
        self.st = ImageGenState(dec, ams, past_lens, seq_base=seq_base)
        self.latents = torch.empty(self.n_tok, self.B, rf.target, dtype=torch.float32, device=dev)
        self.sems = torch.empty(self.n_tok, self.B, tok.feature_dim, dtype=torch.float32, device=dev)
        self.embed = torch.empty(self.B, cfg.hidden_size, dtype=torch.float32, device=dev)
        self.hidden = torch.empty(self.rows, cfg.hidden_size, dtype=torch.float32, device=dev)
        self.noise_t = noises.reshape(self.B, -1, rf.target).transpose(0, 1).contiguous()     # [n+1, B, latent]
        self.sem_state = tok.new_decode_state(n_seq=self.B, t_max=self.n_tok)

    def token(self, ti):
        st, dec, rf = self.st, self.dec, self.rf
        if ti == 0:
            dec.step(self.start_embed, st.row_seq, st.row_slot, st.row_pos, st.row_len, st.key_mask, None, out=self.hidden,
                     rows=self.rows, distinct_sequences=True)        # every CFG row of every image has its own cache sequence
        else:
            dec.step(self.embed, st.row_seq, st.row_slot, st.row_pos, st.row_len, st.key_mask, None, out=self.hidden,
                     rows=self.rows, x_row_div=self.rpi, distinct_sequences=True)
        if ti < self.n_tok:
            rf.sample(self.hidden, self.noise_t[ti], out=self.latents[ti], n_images=self.B, **self.kw)
            self.tok.decode_step(self.latents[ti], self.sem_state, sem_out=self.sems[ti], embed_out=self.embed)
            st.advance()
        elif not self.skip_last:
            rf.sample(self.hidden, self.noise_t[ti], n_images=self.B, **self.kw)


def generate_images(dec: BailingMoeDecoder, rf, tok, start_embed, past_lens, attention_masks, uncond_attention_masks,
                    text_uncond_attention_masks, noises, temperature=1.0, text_cfg=3.0, image_cfg=1.1,
                    decode_pixels=True, skip_last_sample=True, n_groups=1, seq0=0):
    """BailingMoeForCausalLM.generate_image (modeling_bailing_moe.py:1844-1965) for B >= 1 independent images
    (B = 1 is the reference's call).  The images advance in lock-step, so every weight byte streamed from HBM is
    shared by all rows of a group (<= MAX_ROWS rows); with n_groups > 1 the batch is cut into groups that run on
    separate HIP streams, so one group's short kernels (norms, attention, reductions) run beside another's
    weight streaming.

    dec: decoder whose KV sequence seq0 + i*R already holds image i's `past_lens[i]` prompt tokens (R = CFG rows per
    image, equal for all images of the batch; seq0: first cache sequence of the batch).  start_embed fp32 [1,H]: the `<image>` token embedding.
    The three mask arguments are lists of [1, T*] tensors (one per image).  noises fp32 [B, n_tokens(+1), latent]:
    the noise RectifiedFlowLoss.sample would draw per iteration (torch.randn, diff_loss_rf_swiglu.py:117-122).
    CFG scales: the reference always runs 3.0 / 1.1 (its kwargs are swallowed).  Differences from
    the reference that do not change results: CFG rows that are bit-identical (semantic decoder, linear_proj,
    pixel decoder) are computed once per image; the sampler output of the 257th iteration (discarded at :1936)
    is not computed when skip_last_sample.
    Returns dict(image [B,3,R,R] | None, latents [B,n,32], sem [B,n,D], last_hidden [B*R,H], attention_mask list).
    """
    cfg = dec.cfg
    B = len(attention_masks)
    ams = [build_cfg_rows(a, u, t).cpu() for a, u, t in zip(attention_masks, uncond_attention_masks, text_uncond_attention_masks)]
    rpi = ams[0].shape[0]
    assert all(a.shape[0] == rpi for a in ams), "all images of a batch must have the same number of CFG rows"
    n_tok = cfg.num_image_tokens_for_gen
    groups = split_groups(B, n_groups)
    n_groups, per = len(groups), groups[0][1]
    if n_groups > 1 and getattr(dec, "single_stream", False):
        # a TP communicator is one epoch counter + a two-parity inbox: its all-reduces must be strictly sequential on ONE stream
        raise ValueError("tensor-parallel decoders run one lock-step group (n_groups = 1): the communicator serves a single stream")
    row_cap = min(dec.max_rows(), rf.max_rows(), tok.max_decode_rows() * rpi)
    assert max(past_lens) + n_tok + 1 <= dec.t_max and per * rpi <= row_cap, f"{per * rpi} rows per group > {row_cap}"
    if rpi > 1:   # replicate each prompt's KV to its CFG rows (:1891-1902) — device memcpy
        for i, past in enumerate(past_lens):
            for r in range(1, rpi):
                dec.copy_sequence(seq0 + i * rpi, seq0 + i * rpi + r, past)
    noises = noises.reshape(B, -1, rf.target)
    kw = dict(temperature=temperature, text_cfg=text_cfg, image_cfg=image_cfg)
    main = torch.cuda.current_stream()
    streams = [main] if n_groups == 1 else _group_streams(dec.device, n_groups)
    runs = []
    for (lo, hi), s in zip(groups, streams):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            runs.append(_GroupRun(dec, rf, tok, ams[lo:hi], past_lens[lo:hi], noises[lo:hi], seq0 + lo * rpi, start_embed, kw,
                                  skip_last_sample))
    for ti in range(n_tok + 1):
        for run, s in zip(runs, streams):
            with torch.cuda.stream(s):
                run.token(ti)
    for s in streams:
        main.wait_stream(s)
    if hasattr(dec, "check_err"):        # tensor-parallel decoders: an all-reduce wait that expired (NaN-poisoned rows) raises here
        dec.check_err()
    if n_groups > 1:      # side-stream allocations are consumed on the caller's stream below
        for r in runs:
            for t in (r.latents, r.sems, r.hidden):
                t.record_stream(main)
    latents = torch.cat([r.latents for r in runs], dim=1)
    sem_b = torch.cat([r.sems for r in runs], dim=1).transpose(0, 1).contiguous()
    hidden = torch.cat([r.hidden for r in runs], dim=0)
    image = tok.forward_pixel_decoder(sem_b) if decode_pixels else None
    if hasattr(rf, "check_err"):         # persistent sampler launches: an expired grid-barrier wait (NaN latents) raises here — one device
        rf.check_err()                   # word read AFTER the pixel decoder is enqueued (the caller's own sync comes next anyway)
    am_out = [torch.cat((a, torch.ones(rpi, n_tok, dtype=a.dtype)), dim=-1) for a in ams]
    return dict(image=image, latents=latents.transpose(0, 1), sem=sem_b, last_hidden=hidden, attention_mask=am_out,
                cache_len=[p + n_tok + 1 for p in past_lens])


def generate_image(dec: BailingMoeDecoder, rf, tok, start_embed, past_len, attention_mask, uncond_attention_mask,
                   text_uncond_attention_mask, noises, **kw):
    """Batch-size-1 form (the reference's): see generate_images."""
    out = generate_images(dec, rf, tok, start_embed, [past_len], [attention_mask], [uncond_attention_mask],
                          [text_uncond_attention_mask], noises.unsqueeze(0), **kw)
    out["latents"] = out["latents"][0]
    out["sem"] = out["sem"][0]
    out["attention_mask"] = out["attention_mask"][0]
    out["cache_len"] = out["cache_len"][0]
    return out
