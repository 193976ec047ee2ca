"""Host-side mirror of the rectified-flow continuous-token head.

Mirrors `vis_head` + `RectifiedFlowLoss` as set up by
BailingMoeForCausalLM.setup_vishead_diffloss (modeling_bailing_moe.py:1559-1584) and
sampled in forward_for_image_generation_inner (:1659-1670) /
RectifiedFlowLoss.sample (diff_loss_rf_swiglu.py:103-181).  All arithmetic runs in
libmingnative (mn_rf_sample); this class only owns the bf16 weights in HBM,
re-packs them once at load and builds the pointer table the C ABI takes.

Load-time re-packing (results unchanged):
  * the 12 blocks' adaLN projections and the final layer's are stacked into one
    [depth*3w + 2w, w] matrix so a step's modulations are ONE weight-streaming launch;
  * time_embed(t_s * 1000) only ever sees the `steps` fixed times of the Euler grid, so the
    [steps, w] table is computed once (with the same kernels) instead of per token.

`weights="fp8"` (mingnative.h section 7; the reference's reduced-byte surface is the `dtype` switch of
mingunivisioninfer.py:46-70): the ResBlock matrices w12 / w3 — 1.21 of the head's 1.29 B parameters, streamed 16 times per
visual token — are quantised once at load to OCP e4m3 with one power-of-two scale per output row and streamed as bytes; the
codes are decoded inside the weight-streaming kernels up to 64 rows per call (the HBM-bound route); the MFMA-bound wide route gains
nothing from narrower weights and expands the blocks into a bf16 scratch once per call (wide_rf.inl).
"""
import ctypes as C
import math

import torch

from . import _lib, ops
from ._lib import RfHead, check, current_stream, lib, ptr, ptr_array
from .configuration import DEFAULT_VISHEAD_DIFFLOSS, swiglu_hidden


class RectifiedFlowHead:
    def __init__(self, sd, llm_hidden, vishead_diffloss_config=None, latent_dim=32, prefix="", weights="bf16", arith=None):
        """sd: {reference parameter name: bf16 CUDA tensor} holding `vis_head.*` and `diffloss.net.*`.
        weights: "bf16", or a weight-only mode ("fp8" / "int8" / "int4") = w12 / w3 / adaLN quantised here into streamed codes.  A head
        BUILT in a weight-only mode keeps no reference to the bf16 originals (the mode exists for its footprint: 2.4 GB of bf16 ResBlock
        matrices at the 16B-A3B shapes — they are freed as soon as the caller drops `sd`); only a bf16 head keeps them, for to_fp8()."""
        assert weights in _lib.WFMT, f"weights={weights!r}: 'bf16', 'fp8', 'int8' or 'int4'"
        self.weights = weights
        # arith="fp8_mfma" (mingnative.h section 8; BASELINE configs[4]'s "fp8 MFMA"): a LABELLED reduced-arithmetic regime of the wide
        # route (> 64 rows) — e4m3 activations x e4m3 weights on the scaled fp8 MFMA for w12 / w3 / adaLN — with its own stated tolerance;
        # needs weights="fp8".  None: the fp32-class regime everywhere (the parity regime)
        assert arith in (None, "fp8_mfma") and (arith is None or weights == "fp8"), "arith='fp8_mfma' needs weights='fp8'"
        self.arith = arith
        cfg = {**DEFAULT_VISHEAD_DIFFLOSS, **(vishead_diffloss_config or {})}
        assert cfg["vis_head_arch"] == "linear2-norm"          # modeling_bailing_moe.py:1568
        assert cfg["gen_method"].startswith("flow_matching_swiglu-")
        self.w = int(cfg["diffloss_w"])
        self.depth = int(cfg["diffloss_d"])
        self.steps = int(cfg["num_sampling_steps"])
        self.hidden = swiglu_hidden(self.w, int(cfg["gen_method"].split("-")[1]))
        self.target = latent_dim
        self.llm_hidden = llm_hidden
        g = lambda k: sd[prefix + k]
        n = "diffloss.net."
        self.t = dict(
            vis_w=g("vis_head.0.weight"), vis_b=g("vis_head.0.bias"),
            vis_ln_g=g("vis_head.1.weight"), vis_ln_b=g("vis_head.1.bias"),
            cond_w=g(n + "cond_embed.weight"), cond_b=g(n + "cond_embed.bias"),
            in_w=g(n + "input_proj.weight"), in_b=g(n + "input_proj.bias"),
            fin_w=g(n + "final_layer.linear.weight"), fin_b=g(n + "final_layer.linear.bias"),
        )
        blocks = [n + f"res_blocks.{i}." for i in range(self.depth)]
        self.t["ada_w"] = torch.cat([g(b + "adaLN_modulation.1.weight") for b in blocks]
                                    + [g(n + "final_layer.adaLN_modulation.1.weight")], 0).contiguous()
        self.t["ada_b"] = torch.cat([g(b + "adaLN_modulation.1.bias") for b in blocks]
                                    + [g(n + "final_layer.adaLN_modulation.1.bias")], 0).contiguous()
        self.lists = dict(
            ln_g=[g(b + "in_ln.weight") for b in blocks], ln_b=[g(b + "in_ln.bias") for b in blocks],
            w12=[g(b + "mlp.w12.weight") for b in blocks], b12=[g(b + "mlp.w12.bias") for b in blocks],
            w3=[g(b + "mlp.w3.weight") for b in blocks], b3=[g(b + "mlp.w3.bias") for b in blocks],
        )
        for v in list(self.t.values()) + sum(self.lists.values(), []):
            assert v.is_cuda and v.dtype == torch.bfloat16 and v.is_contiguous()
        self._time = {k: g(n + "time_embed.mlp." + k) for k in ("0.weight", "0.bias", "2.weight", "2.bias")}
        self._raw_t, self._raw_lists = dict(self.t), dict(self.lists)      # the checkpoint's tensors: every mode is derived from them, once
        self._apply_mode(weights)
        if weights != "bf16":
            # a quantised head must not pin the bf16 matrices it was derived from (to_fp8 is a bf16 head's method)
            self._raw_t = self._raw_lists = None
            self._time = {k: v for k, v in self._time.items() if k.endswith("bias")}

    def _apply_mode(self, weights):
        """Build the head of weight mode `weights` from the raw bf16 tensors.  Modes that convert every nn.Linear (_lib.FULL_MODEL, the
        reference's int4 / int8 loads): the small Linears (vis_head, cond_embed, input_proj, time_embed, final layer) hold the mode's
        bf16 values and stay on the bf16 kernels; w12 / w3 / adaLN are quantised into streamed codes by _finalize."""
        full = weights in _lib.FULL_MODEL
        conv = (lambda t: ops.fake_quant(t, weights)) if full else (lambda t: t)
        n = "diffloss.net."
        self.t, self.lists = dict(self._raw_t), dict(self._raw_lists)
        for k in ("vis_w", "cond_w", "in_w", "fin_w"):
            self.t[k] = conv(self.t[k])
        self._time_embed_weights = {n + "time_embed.mlp.0.weight": conv(self._time["0.weight"]),
                                    n + "time_embed.mlp.2.weight": conv(self._time["2.weight"])}
        dev = self.t["vis_w"].device
        # time-embedding table for t_s = linspace(1, 0, steps+1)[:-1] (diff_loss_rf_swiglu.py:135, 372-373)
        ts = torch.linspace(1.0, 0.0, self.steps + 1)[:-1] * 1000.0
        half = 128
        freqs = torch.exp(-math.log(10000.0) * torch.arange(0, half, dtype=torch.float32) / half)
        args = ts[:, None].float() * freqs[None]
        femb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1).to(dev)
        temb = []
        for i in range(0, self.steps, 8):
            h = ops.skinny_gemm(femb[i:i + 8].contiguous(), self._time_embed_weights[n + "time_embed.mlp.0.weight"],
                                self._time["0.bias"], epilogue="silu")
            temb.append(ops.skinny_gemm(h, self._time_embed_weights[n + "time_embed.mlp.2.weight"], self._time["2.bias"]))
        self.t["temb"] = torch.cat(temb, 0).contiguous()
        fmt = weights
        if weights == "int4" and (self.w % 64 or self.hidden % 64):
            # NF4 blocks are 64 consecutive elements of the FLATTENED matrix: with rows that are not a multiple of 64 long they straddle
            # rows, which the streaming kernels' per-row absmax tables cannot express — such a head keeps the int4 MODEL (same values,
            # ops.fake_quant blocks the flattened tensor) as bf16 tensors on the bf16 route
            for k in ("w12", "w3"):
                self.lists[k] = [ops.fake_quant(w_, weights) for w_ in self.lists[k]]
            A3 = 3 * self.w
            self.t["ada_w"] = torch.cat([ops.fake_quant(self.t["ada_w"][i * A3:(i + 1) * A3], weights) for i in range(self.depth)]
                                        + [ops.fake_quant(self.t["ada_w"][self.depth * A3:], weights)], 0).contiguous()
            fmt = "bf16"
        self._finalize(fmt)
        self.weights = weights                       # the MODEL's mode; self.stream_fmt: what the streaming kernels read

    def _finalize(self, weights):
        """Quantise the ResBlock matrices when asked to and build the pointer table the C ABI takes."""
        self.weights = self.stream_fmt = weights
        self.scales = {}
        if weights in _lib.W8:
            self.lists = dict(self.lists)
            for k in ("w12", "w3"):
                qs = [ops.quant_rows(w, weights) for w in self.lists[k]]
                self.lists[k] = [q for q, _ in qs]
                self.scales[k] = [sc for _, sc in qs]
            # the stacked adaLN matrix: e4m3 bytes for the one-launch form (<= 4 CFG rows: all 16 steps' rows fit a streaming launch)
            # + its exact bf16 expansion for the MFMA GEMM of the larger row counts — the same model either way
            self.t = dict(self.t)
            self.t["ada_q"], self.t["ada_scale"] = ops.quant_rows(self.t["ada_w"], weights)
            self.t["ada_w"] = ops.dequant_rows(self.t["ada_q"], self.t["ada_scale"], weights)
        llm_hidden = self.llm_hidden
        self._arrays = {k: ptr_array(v) for k, v in self.lists.items()}
        self._scale_arrays = {k: ptr_array(v) for k, v in self.scales.items()}
        s = RfHead()
        s.w, s.depth, s.hidden, s.z_dim, s.target, s.steps, s.llm_hidden = (
            self.w, self.depth, self.hidden, self.w, self.target, self.steps, llm_hidden)
        for k in ("vis_w", "vis_b", "vis_ln_g", "vis_ln_b", "cond_w", "cond_b", "in_w", "in_b", "temb", "ada_w",
                  "ada_b", "fin_w", "fin_b"):
            setattr(s, k, ptr(self.t[k]))
        for k, arr in self._arrays.items():
            setattr(s, k, C.cast(arr, _lib.PP))
        s.wfmt = _lib.WFMT[weights]
        s.arith = 1 if (getattr(self, "arith", None) == "fp8_mfma" and weights == "fp8") else 0
        if weights in _lib.W8:
            s.w12_scale = C.cast(self._scale_arrays["w12"], _lib.PP)
            s.w3_scale = C.cast(self._scale_arrays["w3"], _lib.PP)
            s.ada_q, s.ada_scale = ptr(self.t["ada_q"]), ptr(self.t["ada_scale"])
        self.struct = s
        self._ws = {}

    def to_fp8(self, weights="fp8", arith=None):
        """A second head on the same HBM tensors in another weight mode — e4m3 (default), "int8" or "int4" — derived from the raw
        bf16 tensors (this head stays usable).  arith="fp8_mfma": the labelled fp8-MFMA regime of the wide route (e4m3 weights only)."""
        import copy
        assert self.weights == "bf16" and weights in _lib.W8 and self._raw_t is not None
        assert arith in (None, "fp8_mfma") and (arith is None or weights == "fp8")
        new = copy.copy(self)
        new.arith = arith
        new._apply_mode(weights)
        return new

    def weight_bytes_per_step(self, rows=2):
        """Weight bytes one Euler step must stream from HBM (bf16: 2 per ResBlock parameter; fp8: 1 + the row scales)."""
        n_w = 2 * self.hidden * self.w + self.w * self.hidden
        if self.stream_fmt == "int4":
            per_block = n_w // 2 + n_w // 64 * 4                       # codes + one fp32 absmax per 64 weights
        else:
            per_block = n_w * (1 if self.stream_fmt in _lib.W8 else 2)
            if self.stream_fmt in _lib.W8:
                per_block += (2 * self.hidden + self.w) * 4
        return self.depth * per_block + self.ada_bytes(rows)

    def ada_bytes(self, rows=2):
        """Bytes of the stacked adaLN matrix one visual token reads (once: all Euler steps in one launch): e4m3 in fp8 mode while
        steps x rows <= 64, else bf16."""
        n = self.t["ada_w"].numel()
        if self.stream_fmt == "int4" and self.steps * rows <= 64:
            return n // 2 + n // 64 * 4
        return n + 4 * self.t["ada_w"].shape[0] if (self.stream_fmt in _lib.W8 and self.steps * rows <= 64) else 2 * n

    def dequantized_blocks(self):
        """fp8 mode: {reference parameter name: bf16 tensor} of the ResBlock matrices as the kernels see them (e4m3 * row scale,
        exact in bf16) — what the oracle is fed in the parity tests, and a bf16 model of its own right."""
        assert self.weights in _lib.W8
        out = {}
        for i in range(self.depth):
            for k, name in (("w12", "mlp.w12.weight"), ("w3", "mlp.w3.weight")):
                out[f"diffloss.net.res_blocks.{i}.{name}"] = (ops.dequant_rows(self.lists[k][i], self.scales[k][i], self.stream_fmt)
                                                              if self.stream_fmt in _lib.W8 else self.lists[k][i])
            out[f"diffloss.net.res_blocks.{i}.adaLN_modulation.1.weight"] = self.t["ada_w"][i * 3 * self.w:(i + 1) * 3 * self.w]
        out["diffloss.net.final_layer.adaLN_modulation.1.weight"] = self.t["ada_w"][self.depth * 3 * self.w:]
        if self.weights in _lib.FULL_MODEL:          # every nn.Linear of the head is converted: the small ones as bf16 values
            out.update({"vis_head.0.weight": self.t["vis_w"], "diffloss.net.cond_embed.weight": self.t["cond_w"],
                        "diffloss.net.input_proj.weight": self.t["in_w"], "diffloss.net.final_layer.linear.weight": self.t["fin_w"]})
            out.update(self._time_embed_weights)
        return out

    def check_err(self):
        """Host sync point (reads one device word): raises if a persistent sampler launch gave up at its grid barrier — the latents
        of that call are NaN by construction; this names the cause (grid_bar.h)."""
        _lib.persist_check()

    def max_rows(self):
        """Rows one sample() accepts: 64, or 2048 when the wide route applies (all widths multiples of 64)."""
        return int(lib().mn_rf_max_rows(C.byref(self.struct)))

    def _workspace(self, rows, device):
        key = (rows, torch.cuda.current_stream().cuda_stream)      # one scratch area per stream: groups overlap
        if key not in self._ws:
            n = lib().mn_rf_workspace_bytes(C.byref(self.struct), rows)
            self._ws[key] = torch.empty(n, dtype=torch.uint8, device=device)
        return self._ws[key]

    def sample(self, hidden, noise, temperature=1.0, text_cfg=3.0, image_cfg=1.1, out=None, n_images=1):
        """hidden [rows, llm_hidden] fp32: last hidden state of each CFG row, image-major (rows = n_images x R).
        noise [n_images, target] (or [target] for one image) fp32.  Returns the sampled latents
        [n_images, target] ([target] when called with a 1-D noise) — identical for every CFG row of an image."""
        rows = hidden.shape[0]
        assert hidden.dtype == torch.float32 and hidden.is_cuda and hidden.stride(1) == 1
        assert noise.dtype == torch.float32 and noise.is_cuda and noise.is_contiguous()
        assert noise.numel() == n_images * self.target and rows % n_images == 0
        ws = self._workspace(rows, hidden.device)
        _lib.persist_status_word(hidden.device)          # (registered once per process: an expired grid-barrier wait is reported, check_err)
        if out is None:
            out = torch.empty(noise.shape, dtype=torch.float32, device=hidden.device)
        assert out.is_contiguous() and out.numel() == n_images * self.target
        check(lib().mn_rf_sample(C.byref(self.struct), ptr(hidden), hidden.stride(0), rows, n_images, ptr(noise),
                                 float(temperature), float(text_cfg), float(image_cfg), ptr(out), ptr(ws), ws.numel(),
                                 current_stream()), "mn_rf_sample")
        return out
