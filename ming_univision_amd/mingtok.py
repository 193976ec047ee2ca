"""Host-side mirror of MingTok-Vision (mingtok/modeling_mingtok.py:97-207).

Public surface kept from the reference: `MingTok(config)`, `.forward(x)`, `.forward_enc_dec(x)`,
`.forward_feature_decoder(h, past_key_values)`, `.forward_feature_decoder_wo_cache(h)`,
`.forward_pixel_decoder(x)`, `.latent_dim`, `.feature_dim`, `.patch_size`.

Two execution regimes, both entirely in libmingnative:
  * batched (encode, semantic-decoder prefill, pixel decoder): bf16 MFMA GEMMs with fp32
    accumulate and an fp32 residual stream, flash attention hd=64;
  * cached decode of the causal semantic decoder (<= 8 rows): weight-streaming skinny GEMMs with
    fp32 activations (mn_semdec_step), fused with linear_proj for the LLM's next input embedding.
torch is used for memory, views and the load-time pos-embed interpolation table only.
"""
import ctypes as C
import math

import torch
import torch.nn.functional as F

from . import _lib, ops
from ._lib import SemDec, check, current_stream, lib, ptr, ptr_array
from .configuration import MingTokConfig, mingtok_param_shapes, swiglu_hidden


class SemDecodeState:
    """KV arena + device row bookkeeping of a cached semantic-decoder decode (replaces DynamicCache)."""

    def __init__(self, depth, n_heads, n_seq, t_max, device):
        self.kv = torch.zeros(depth, n_seq, 2, n_heads, t_max, 64, dtype=torch.float32, device=device)
        self.n_seq, self.t_max = n_seq, t_max
        self.row_seq = torch.arange(n_seq, dtype=torch.int32, device=device)
        self.row_slot = torch.zeros(n_seq, dtype=torch.int32, device=device)
        self.row_len = torch.ones(n_seq, dtype=torch.int32, device=device)
        self.length = 0

    def get_seq_length(self):
        return self.length


class MingTok:
    config_class = MingTokConfig

    def __init__(self, config: MingTokConfig, state_dict=None, device="cuda", seed=0, linear_proj=None, precision="bf16", weights="bf16"):
        """state_dict: reference-named tensors (any float dtype; stored as bf16 in HBM); None -> synthetic.
        linear_proj: optional list of (weight, bias) bf16 CUDA tensors (modeling_bailingmm.py:111-115)
        fused behind the decode step.
        precision: regime of the BATCHED passes (encode, semantic-decoder prefill, pixel decoder); every method takes an override.
          "bf16" — bf16 activations on the MFMA, flash attention (the reference's own autocast precision: ~1e-2 of the fp32 result);
          "fp32" — fp32-class: every Linear a gemm256 launch on bf16 hi/lo activation pairs (2^-17), attention on the fp32 decode
                   kernels against a scratch K/V arena; within 1e-3 of the fp32 reference path (what feeds the LLM in understanding / editing)."""
        assert precision in ("bf16", "fp32")
        self.precision = precision
        self.config = config
        enc, sem, pix = config.low_level_encoder, config.semantic_decoder, config.pixel_decoder
        self.latent_dim = enc.get("out_dim", 32)
        self.feature_dim = sem.get("embed_dim", 1024)
        self.patch_size = enc.get("patch_size", 32)
        self.pix_patch = pix.get("patch_size", 16)
        self.sem_patch = sem.get("patch_size", 16)
        self.scaling_factor, self.mean = config.scaling_factor, config.mean
        self.device = torch.device(device)
        shapes = mingtok_param_shapes(config)
        if state_dict is None:
            from .synth import synth_tensor
            self.sd = {k: synth_tensor(k, s, seed, self.device, torch.bfloat16) for k, s in shapes.items()}
        else:
            missing = [k for k in shapes if k not in state_dict]
            if missing:
                raise KeyError(f"MingTok state dict is missing {missing[:5]}…")
            self.sd = {k: state_dict[k].to(self.device, torch.bfloat16).contiguous() for k in shapes}
        # weight-only modes that convert every nn.Linear (_lib.FULL_MODEL: the reference's int4 / int8 loads walk the vision tower too):
        # MingTok's Linears hold the mode's bf16 values and stay on the bf16 kernels (0.6 of the 37 GB a visual token reads).
        # Not converted: the patch-embed Conv2d (4-D), LayerNorm gains, cls / pos / mask tokens.  `linear_proj` is the caller's.
        self.weights = weights
        self.sd = ops.convert_linears(self.sd, weights)
        self.enc_depth, self.sem_depth, self.pix_depth = enc.get("depth", 24), sem.get("decoder_depth", 1), pix.get("decoder_depth", 1)
        self.enc_dim, self.pix_dim = enc.get("embed_dim", 1024), pix.get("embed_dim", 1024)
        self._pos_cache = {}
        self._semdec_struct = None
        self.linear_proj = linear_proj
        self._ws = {}

    # ---- helpers ---------------------------------------------------------------------------
    def _w(self, k):
        return self.sd[k]

    def _resid_linear(self, a, w, bias, x, next_norm):
        """x (fp32 residual stream) += a @ w^T + bias; returns bf16 next_norm(x) when it was fused into the tail, else None.
        next_norm: None or (weight, bias, gelu).  With few rows (a 256 x 256-tile GEMM would fill under half the chip) the
        product runs split-K and ONE launch reduces the slabs, updates the stream and applies the consumer's LayerNorm."""
        ks = ops.splitk_plan(a.shape[0], w.shape[0], a.shape[1])
        if ks and lib().mn_gemm256_supported(a.shape[1], 0, w.stride(0), w.shape[0], a.shape[0], w.shape[0], a.shape[1]):
            P = ops.gemm256_splitk_bf16(a, w, bias, ks)
            if next_norm is None:
                return ops.slab_resid_norm(P, x, norm=False)
            return ops.slab_resid_norm(P, x, next_norm[0], next_norm[1], gelu=next_norm[2])
        ops.gemm_bf16(a, w, bias, "f32_resid", out=x)
        return None

    def _block(self, x, prefix, D, B, T, causal, xn=None, next_norm=None):
        """Block.forward / CausalBlock.forward (layers/block.py:80-105, 301-327) on the fp32 residual x [B*T, D].
        xn: bf16 norm1(x) when the previous block's tail already produced it; next_norm = (weight, bias, gelu) of the LayerNorm
        that consumes this block's output.  Returns that LayerNorm's bf16 output when it was fused into the tail, else None."""
        nh = D // 64
        if xn is None:
            xn = ops.layernorm_bf16(x, self._w(prefix + ".norm1.weight"), self._w(prefix + ".norm1.bias"))
        qkv = ops.gemm_bf16(xn, self._w(prefix + ".attn.qkv.weight"), self._w(prefix + ".attn.qkv.bias"))
        att = ops.attn_prefill_hd64(qkv, B, T, nh, causal)
        xn = self._resid_linear(att, self._w(prefix + ".attn.proj.weight"), self._w(prefix + ".attn.proj.bias"), x,
                                (self._w(prefix + ".norm2.weight"), self._w(prefix + ".norm2.bias"), False))
        if xn is None:
            xn = ops.layernorm_bf16(x, self._w(prefix + ".norm2.weight"), self._w(prefix + ".norm2.bias"))
        if (prefix + ".mlp.w12.weight") in self.sd:
            w12, b12, w3 = self._w(prefix + ".mlp.w12.weight"), self._w(prefix + ".mlp.w12.bias"), self._w(prefix + ".mlp.w3.weight")
            if prefix.startswith("semantic_decoder.blocks.0."):
                # the zero-padded copies (hidden 2736 -> 2752): whole 64-k tiles for w3, results unchanged
                self._semdec()
                li = int(prefix.rsplit(".", 1)[1])
                w12, b12, w3 = self._sem_pad[0][li], self._sem_pad[1][li], self._sem_pad[2][li]
            hid = w12.shape[0] // 2
            if x.shape[0] >= 1024 and hid % 4 == 0 and D % 64 == 0:
                h = ops.gemm256_swiglu(xn, w12, b12)                 # SwiGLU in the GEMM epilogue: no [M, 2*hidden] round trip
            else:
                h = ops.swiglu_bf16(ops.gemm_bf16(xn, w12, b12))
            return self._resid_linear(h, w3, self._w(prefix + ".mlp.w3.bias"), x, next_norm)
        h = ops.gemm_bf16(xn, self._w(prefix + ".mlp.fc1.weight"), self._w(prefix + ".mlp.fc1.bias"), "bf16_gelu")
        return self._resid_linear(h, self._w(prefix + ".mlp.fc2.weight"), self._w(prefix + ".mlp.fc2.bias"), x, next_norm)

    def _blocks(self, x, stem, depth, D, B, T, causal, final_norm):
        """`depth` blocks `stem`.i in sequence; final_norm = (weight, bias, gelu) of the LayerNorm after the last one.
        Returns its bf16 output.  Each block's tail is asked for the next LayerNorm (fused when the tail runs split-K)."""
        xn = None
        for i in range(depth):
            nxt = (final_norm if i + 1 == depth else
                   (self._w(f"{stem}.{i + 1}.norm1.weight"), self._w(f"{stem}.{i + 1}.norm1.bias"), False))
            xn = self._block(x, f"{stem}.{i}", D, B, T, causal, xn=xn, next_norm=nxt)
        if xn is None:
            xn = ops.layernorm_bf16(x, final_norm[0], final_norm[1], gelu=final_norm[2])
        return xn

    def _pos_embed(self, npatch, w, h):
        """interpolate_pos_encoding (vision_transformer.py:183-215) — cached per resolution (load-time table)."""
        key = (npatch, w, h)
        if key in self._pos_cache:
            return self._pos_cache[key]
        pe = self._w("low_level_encoder.pos_embed").float()
        N = pe.shape[1] - 1
        if not (npatch == N and w == h):
            P, dim = self.patch_size, pe.shape[-1]
            w0, h0 = w // P, h // P
            M = int(math.sqrt(N))
            sx, sy = float(w0 + 0.1) / M, float(h0 + 0.1) / M
            patch = F.interpolate(pe[:, :-1].reshape(1, M, M, dim).permute(0, 3, 1, 2), mode="bicubic",
                                  antialias=False, scale_factor=(sx, sy))
            assert (w0, h0) == patch.shape[-2:]
            pe = torch.cat((patch.permute(0, 2, 3, 1).reshape(1, -1, dim), pe[:, -1:]), dim=1)
        pe = pe.reshape(-1, pe.shape[-1]).contiguous()
        self._pos_cache[key] = pe
        return pe

    # ---- fp32-class ("precise") regime: hi/lo GEMMs + fp32 attention ----------------------------------------------
    def _swiglu_padded(self, prefix):
        """(w12p, b12p, w3p): the block's SwiGLU weights with the hidden width zero-padded to a multiple of 64 (whole K-tiles for
        w3; the padded units are silu(0) * 0 against zero w3 columns) — load-time copies, cached."""
        if prefix.startswith("semantic_decoder.blocks.0."):       # the cached-decode struct already holds these copies
            self._semdec()
            li = int(prefix.rsplit(".", 1)[1])
            return self._sem_pad[0][li], self._sem_pad[1][li], self._sem_pad[2][li]
        key = ("pad", prefix)
        if key not in self._pos_cache:
            w12, b12, w3 = self._w(prefix + ".mlp.w12.weight"), self._w(prefix + ".mlp.w12.bias"), self._w(prefix + ".mlp.w3.weight")
            hid, D = w12.shape[0] // 2, w12.shape[1]
            hp = (hid + 63) // 64 * 64
            if hp == hid:
                self._pos_cache[key] = (w12, b12, w3)
            else:
                a = torch.zeros(2 * hp, D, dtype=torch.bfloat16, device=self.device)
                a[:hid], a[hp:hp + hid] = w12[:hid], w12[hid:]
                bb = torch.zeros(2 * hp, dtype=torch.bfloat16, device=self.device)
                bb[:hid], bb[hp:hp + hid] = b12[:hid], b12[hid:]
                c = torch.zeros(w3.shape[0], hp, dtype=torch.bfloat16, device=self.device)
                c[:, :hid] = w3
                self._pos_cache[key] = (a, bb, c)
        return self._pos_cache[key]

    def _attention_fp32(self, qkv, B, T, nh, causal):
        """softmax(q k^T / 8) v per (image, head), fp32-class: the hi/lo flash kernel (mn_attn_prefill_hd64_f32: operands as bf16
        hi + lo pairs, three MFMAs per product, fp32 softmax).  qkv fp32 [B*T, 3*nh*64] in the reshape of attention.py:83 -> the
        hi/lo pair bf16 [2, B*T, nh*64] of the result (the projection's operand).  (Round 3 borrowed the per-row decode kernels
        over a scratch arena: every query row re-read all keys — 8.5 of the 14.8 ms a 1024^2 image spent in MingTok.)"""
        return ops.attn_prefill_hd64_f32(qkv, B, T, nh, causal)

    def _block_fp32(self, x, prefix, D, B, T, causal):
        """Block.forward / CausalBlock.forward (layers/block.py:80-105, 301-327) on the fp32 residual x [B*T, D], fp32-class."""
        nh = D // 64
        xn, _ = ops.norm_act_split(x, "ln", self._w(prefix + ".norm1.weight"), self._w(prefix + ".norm1.bias"))
        qkv = ops.linear_hilo(xn, self._w(prefix + ".attn.qkv.weight"), self._w(prefix + ".attn.qkv.bias"))
        att = self._attention_fp32(qkv, B, T, nh, causal)
        ops.linear_hilo(att, self._w(prefix + ".attn.proj.weight"), self._w(prefix + ".attn.proj.bias"), out=x, resid=True)
        xn, _ = ops.norm_act_split(x, "ln", self._w(prefix + ".norm2.weight"), self._w(prefix + ".norm2.bias"))
        if (prefix + ".mlp.w12.weight") in self.sd:
            w12, b12, w3 = self._swiglu_padded(prefix)
            h = ops.gemm256_swiglu_split(xn, w12, b12)
            ops.linear_hilo(h, w3, self._w(prefix + ".mlp.w3.bias"), out=x, resid=True)
        else:
            h = ops.linear_hilo(xn, self._w(prefix + ".mlp.fc1.weight"), self._w(prefix + ".mlp.fc1.bias"))
            hs, _ = ops.norm_act_split(h, "none", gelu=True)
            ops.linear_hilo(hs, self._w(prefix + ".mlp.fc2.weight"), self._w(prefix + ".mlp.fc2.bias"), out=x, resid=True)

    def _blocks_fp32(self, x, stem, depth, D, B, T, causal):
        for i in range(depth):
            self._block_fp32(x, f"{stem}.{i}", D, B, T, causal)

    # ---- encoder ---------------------------------------------------------------------------
    def encode(self, x, precision=None):
        """low_level_encoder(x) -> RAW latent fp32 [B, N+1, latent_dim] (vision_transformer.py:225-233)."""
        fp32 = (precision or self.precision) == "fp32"
        B, _, W, H = x.shape
        P, D = self.patch_size, self.enc_dim
        gh, gw = W // P, H // P
        N = gh * gw
        # conv k = s = P == GEMM over (c, py, px) (patch_embed.py:76-78): ONE pass reads the image and writes the GEMM's bf16 (or
        # hi / lo) operand; a second one appends the cls token LAST (:221) and adds the position embedding
        wpe = self._w("low_level_encoder.patch_embed.proj.weight").reshape(D, 3 * P * P)
        cols = ops.patchify_operand(x.to(self.device, torch.float32).contiguous(), P, hilo=fp32)
        if fp32:
            tok = ops.linear_hilo(cols, wpe, self._w("low_level_encoder.patch_embed.proj.bias"))
        else:
            tok = ops.gemm_bf16(cols, wpe, self._w("low_level_encoder.patch_embed.proj.bias"), "f32")
        T = N + 1
        pe = self._pos_embed(N, W, H)
        xf = ops.tokens_assemble(tok, self._w("low_level_encoder.cls_token").reshape(D), pe, B, N).reshape(B * T, D)
        # blocks, then forward_out_layer (:173-178) whose LayerNorm + GELU rides the last block's tail
        if fp32:
            self._blocks_fp32(xf, "low_level_encoder.blocks.0", self.enc_depth, D, B, T, False)
            xn, _ = ops.norm_act_split(xf, "ln", self._w("low_level_encoder.out_norm.weight"), self._w("low_level_encoder.out_norm.bias"), gelu=True)
            y = ops.linear_hilo(xn, self._w("low_level_encoder.out_proj.weight"), self._w("low_level_encoder.out_proj.bias"))
        else:
            xn = self._blocks(xf, "low_level_encoder.blocks.0", self.enc_depth, D, B, T, False,
                              (self._w("low_level_encoder.out_norm.weight"), self._w("low_level_encoder.out_norm.bias"), True))
            y = ops.gemm_bf16(xn, self._w("low_level_encoder.out_proj.weight"), self._w("low_level_encoder.out_proj.bias"), "f32")
        out = torch.empty_like(y)
        check(lib().mn_group_mean_add(ptr(y), ptr(xf), ptr(out), B * T, D, self.latent_dim, current_stream()), "mn_group_mean_add")
        return out.reshape(B, T, self.latent_dim)

    # ---- semantic decoder, full causal pass -----------------------------------------------
    def _semantic_decoder_full(self, latent_raw, precision=None):
        """forward_features without cache (vision_transformer.py:382-451): [B,T,in] RAW latent -> x_norm fp32 [B,T,D]
        (rounded to bf16 in the bf16 regime, like the reference's autocast output)."""
        fp32 = (precision or self.precision) == "fp32"
        B, T, Cin = latent_raw.shape
        D = self.feature_dim
        lat = latent_raw.reshape(B * T, Cin).contiguous()
        if fp32:
            y = ops.linear_hilo(ops.split_hilo(lat), self._w("semantic_decoder.in_proj.weight"), self._w("semantic_decoder.in_proj.bias"))
        else:
            y = ops.gemm_bf16(ops.f32_to_bf16(lat), self._w("semantic_decoder.in_proj.weight"),
                              self._w("semantic_decoder.in_proj.bias"), "f32")
        x = torch.empty_like(y)
        check(lib().mn_repeat_add(ptr(y), ptr(lat), ptr(x), B * T, D, Cin, 1.0, 0.0, current_stream()), "mn_repeat_add")
        if fp32:
            self._blocks_fp32(x, "semantic_decoder.blocks.0", self.sem_depth, D, B, T, True)
            _, out = ops.norm_act_split(x, "ln", self._w("semantic_decoder.norm.weight"), self._w("semantic_decoder.norm.bias"),
                                        want_split=False, want_f32=True)
            return out.reshape(B, T, D)
        xn = self._blocks(x, "semantic_decoder.blocks.0", self.sem_depth, D, B, T, True,
                          (self._w("semantic_decoder.norm.weight"), self._w("semantic_decoder.norm.bias"), False))
        return ops.bf16_to_f32(xn).reshape(B, T, D)

    def forward(self, x, precision=None):
        """MingTok.forward (modeling_mingtok.py:156-163)."""
        latent = self.encode(x, precision)
        x_norm = self._semantic_decoder_full(latent, precision)
        return {"x_norm_patchtokens": x_norm[:, :-1], "latent": (latent - self.mean) / self.scaling_factor}

    __call__ = forward

    def forward_feature_decoder_wo_cache(self, hidden_states, precision=None):
        """(modeling_mingtok.py:176-177): RAW latent in, dict out."""
        x_norm = self._semantic_decoder_full(hidden_states.to(self.device, torch.float32), precision)
        return {"x_norm_patchtokens": x_norm[:, :-1] if x_norm.shape[1] > 1 else x_norm, "x_norm": x_norm}

    # ---- pixel decoder ----------------------------------------------------------------------
    def forward_pixel_decoder(self, sem, precision=None):
        """MingTok.forward_pixel_decoder (modeling_mingtok.py:179-196): sem [B,N,Dsem] -> image [B,3,R,R] fp32 in [-1,1]."""
        fp32 = (precision or self.precision) == "fp32"
        B, N, Ds = sem.shape
        r = self.sem_patch // self.pix_patch
        Dp = self.pix_dim
        s32 = sem.to(self.device, torch.float32).reshape(B * N, Ds).contiguous()
        if fp32:
            y = ops.linear_hilo(ops.split_hilo(s32), self._w("sem_to_pix.weight"), self._w("sem_to_pix.bias"))
        else:
            y = ops.gemm_bf16(ops.f32_to_bf16(s32), self._w("sem_to_pix.weight"), self._w("sem_to_pix.bias"), "f32")
        h = w = int(math.sqrt(N))
        x = ops.subtoken_rearrange(y, B, h, w, r, Dp)           # "b (h w) (x y c) -> b (h x w y) c" in one pass
        T = N * r * r
        if fp32:
            self._blocks_fp32(x, "pixel_decoder.blocks.0", self.pix_depth, Dp, B, T, False)
            xn, _ = ops.norm_act_split(x, "ln", self._w("pixel_decoder.norm.weight"), self._w("pixel_decoder.norm.bias"))
            o = ops.linear_hilo(xn, self._w("pixel_decoder.head.weight"), self._w("pixel_decoder.head.bias"))
        else:
            xn = self._blocks(x, "pixel_decoder.blocks.0", self.pix_depth, Dp, B, T, False,
                              (self._w("pixel_decoder.norm.weight"), self._w("pixel_decoder.norm.bias"), False))
            o = ops.gemm_bf16(xn, self._w("pixel_decoder.head.weight"), self._w("pixel_decoder.head.bias"), "f32")
        p = self.pix_patch
        hh = ww = int(math.sqrt(T))
        return ops.unpatchify_clamp(o, B, hh, ww, p, -1.0, 1.0)  # 'nhwpqc->nchpwq' (vision_transformer.py:515-527) + clamp_ in one pass

    def forward_enc_dec(self, x, precision=None):
        """MingTok.forward_enc_dec (modeling_mingtok.py:150-153)."""
        return self.forward_pixel_decoder(self.forward(x, precision)["x_norm_patchtokens"], precision)

    # ---- cached decode -----------------------------------------------------------------------
    def _semdec(self):
        if self._semdec_struct is not None:
            return self._semdec_struct
        D, L = self.feature_dim, self.sem_depth
        b = [f"semantic_decoder.blocks.0.{i}" for i in range(L)]
        names = dict(ln1_g=".norm1.weight", ln1_b=".norm1.bias", wqkv=".attn.qkv.weight", bqkv=".attn.qkv.bias",
                     wproj=".attn.proj.weight", bproj=".attn.proj.bias", ln2_g=".norm2.weight", ln2_b=".norm2.bias",
                     w12=".mlp.w12.weight", b12=".mlp.w12.bias", w3=".mlp.w3.weight", b3=".mlp.w3.bias")
        self._sem_arrays = {k: ptr_array([self._w(p + sfx) for p in b]) for k, sfx in names.items()}
        s = SemDec()
        s.dim, s.depth, s.n_heads, s.hidden = D, L, D // 64, swiglu_hidden(D)
        s.in_dim = self.latent_dim
        s.mean, s.scale = self.mean, self.scaling_factor
        s.in_w, s.in_b = ptr(self._w("semantic_decoder.in_proj.weight")), ptr(self._w("semantic_decoder.in_proj.bias"))
        for k, arr in self._sem_arrays.items():
            setattr(s, k, C.cast(arr, _lib.PP))
        s.norm_g, s.norm_b = ptr(self._w("semantic_decoder.norm.weight")), ptr(self._w("semantic_decoder.norm.bias"))
        # wide-row route (> 64 rows in lock-step): SwiGLU hidden width zero-padded to a multiple of 64 so that w3's K is whole
        # 64-k tiles — load-time copies, the padded units are silu(0) * 0 = 0 against zero w3 columns
        hid = s.hidden
        hp = (hid + 63) // 64 * 64
        w12p, b12p, w3p = [], [], []
        for p_ in b:
            w12, b12, w3 = self._w(p_ + ".mlp.w12.weight"), self._w(p_ + ".mlp.w12.bias"), self._w(p_ + ".mlp.w3.weight")
            a = torch.zeros(2 * hp, D, dtype=torch.bfloat16, device=self.device)
            a[:hid], a[hp:hp + hid] = w12[:hid], w12[hid:]
            bb = torch.zeros(2 * hp, dtype=torch.bfloat16, device=self.device)
            bb[:hid], bb[hp:hp + hid] = b12[:hid], b12[hid:]
            c = torch.zeros(D, hp, dtype=torch.bfloat16, device=self.device)
            c[:, :hid] = w3
            w12p.append(a); b12p.append(bb); w3p.append(c)
        self._sem_pad = (w12p, b12p, w3p)
        self._sem_pad_arrays = tuple(ptr_array(t) for t in self._sem_pad)
        s.hidden_pad = hp
        s.w12p, s.b12p, s.w3p = (C.cast(a_, _lib.PP) for a_ in self._sem_pad_arrays)
        if self.linear_proj:
            self._proj_arrays = (ptr_array([w for w, _ in self.linear_proj]), ptr_array([b_ for _, b_ in self.linear_proj]))
            s.proj_w, s.proj_b = C.cast(self._proj_arrays[0], _lib.PP), C.cast(self._proj_arrays[1], _lib.PP)
            s.proj_dim, s.proj_depth = self.linear_proj[0][0].shape[0], len(self.linear_proj)
        else:
            s.proj_dim, s.proj_depth = 8, 0
        self._semdec_struct = s
        return s

    def max_decode_rows(self):
        """Rows one decode_step() accepts: 64, or 2048 on the wide route."""
        return int(lib().mn_semdec_max_rows(C.byref(self._semdec())))

    def new_decode_state(self, n_seq=1, t_max=256):
        return SemDecodeState(self.sem_depth, self.feature_dim // 64, n_seq, t_max, self.device)

    def decode_step(self, latent_norm, state: SemDecodeState, sem_out=None, embed_out=None):
        """One cached causal step for state.n_seq rows. latent_norm fp32 [rows, latent_dim] (normalised).
        Writes x_norm rows to sem_out [rows, D] and linear_proj(x_norm) to embed_out [rows, H] (if given)."""
        s = self._semdec()
        M = latent_norm.shape[0]
        assert M == state.n_seq and state.length < state.t_max
        assert latent_norm.dtype == torch.float32 and latent_norm.is_cuda and latent_norm.is_contiguous()
        if sem_out is None:
            sem_out = torch.empty(M, self.feature_dim, dtype=torch.float32, device=self.device)
        key = (M, state.t_max, torch.cuda.current_stream().cuda_stream)
        if key not in self._ws:
            n = lib().mn_semdec_workspace_bytes(C.byref(s), M, state.t_max)
            self._ws[key] = torch.empty(n, dtype=torch.uint8, device=self.device)
        ws = self._ws[key]
        check(lib().mn_semdec_step(C.byref(s), ptr(latent_norm), M, ptr(state.row_seq), ptr(state.row_slot),
                                   ptr(state.row_len), ptr(state.kv), state.n_seq, state.t_max, ptr(sem_out),
                                   ptr(embed_out), ptr(ws), ws.numel(), current_stream()), "mn_semdec_step")
        check(lib().mn_rows_advance(ptr(state.row_slot), ptr(state.row_len), None, M, 1, current_stream()), "mn_rows_advance")
        state.length += 1
        return sem_out

    def forward_feature_decoder(self, hidden_states, past_key_values=None):
        """MingTok.forward_feature_decoder (modeling_mingtok.py:165-174): normalised latent [rows,1,32] in,
        dict(x_norm_patchtokens [rows,1,D], past_key_values) out."""
        rows = hidden_states.shape[0]
        if past_key_values is None:
            past_key_values = self.new_decode_state(n_seq=rows)
        lat = hidden_states.to(self.device, torch.float32).reshape(rows, self.latent_dim).contiguous()
        sem = self.decode_step(lat, past_key_values)
        return {"x_norm_patchtokens": sem.reshape(rows, 1, -1), "past_key_values": past_key_values}
