"""CPU oracle (plain fp32 PyTorch) for the Bailing-MoE LLM forward and the
`generate_image` autoregressive loop.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functional restatement of
mingunivision/modeling_bailing_moe.py on a state dict with the reference's
parameter names (`model.layers.{i}.attention.query_key_value.weight`, … — the
names of BailingMoeForCausalLM, i.e. without the outer `model.` that
MingUniVisionForConditionalGeneration adds).  Pinned by tests/golden/llm_*.npz
and tests/golden/genimg_*.npz, captured from the reference by gen_golden.py.

Hot-path configuration only: rope_scaling=None (BailingMoeRotaryEmbeddingLegacy),
eager attention with fp32 softmax, first_k_dense_replace=0.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import rf_ref


@dataclass
class LLMConfig:
    hidden_size: int = 2048
    num_hidden_layers: int = 28
    num_attention_heads: int = 16
    num_key_value_heads: int = 4
    head_dim: int = 128
    num_experts: int = 64
    num_experts_per_tok: int = 6
    num_shared_experts: int = 2
    moe_intermediate_size: int = 1408
    rms_norm_eps: float = 1e-5
    rope_theta: float = 600000.0
    norm_topk_prob: bool = True
    multi_gate: bool = True
    vocab_size: int = 126464
    num_image_tokens_for_gen: int = 256
    image_start_token: int = 126347
    image_patch_token: int = 126346


def rmsnorm(x, w, eps):
    """BailingMoeRMSNorm.forward (modeling_bailing_moe.py:131-136)."""
    x = x.float()
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def rope_cos_sin(head_dim, base, seq_len):
    """BailingMoeRotaryEmbeddingLegacy.forward (:213-237): emb = cat(freqs, freqs)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(seq_len, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    """:428-433"""
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin, position_ids):
    """apply_rotary_pos_emb (:436-461). q [B,h,T,hd]; position_ids [B,T]."""
    cos = cos[position_ids].unsqueeze(1)
    sin = sin[position_ids].unsqueeze(1)
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


def rope3d_cos_sin(head_dim, base, position_ids3):
    """BailingMoe3DRotaryEmbedding.forward (:413-425): position_ids3 [3,B,T] (t, h, w) -> cos, sin [3,B,T,hd]."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    freqs = position_ids3.float().unsqueeze(-1) * inv_freq            # [3,B,T,hd/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def apply_rope_3d(q, k, cos, sin, mrope_section=(16, 24, 24)):
    """apply_multimodal_rotary_pos_emb (:463-469): the 2 x 3 sections of the head dim take their angles from the
    t, h, w streams in turn.  q [B,h,T,hd]; cos/sin [3,B,T,hd]."""
    sec = list(mrope_section) * 2
    cos = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1).unsqueeze(1)
    sin = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1).unsqueeze(1)
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


def build_4d_mask(attention_mask, q_len, past_len):
    """Equivalent of transformers-4.52 `_prepare_4d_causal_attention_mask`
    (called at modeling_bailing_moe.py:1466): additive mask [B,1,q,kv], finfo.min
    where the key is in the future (bottom-right aligned) or attention_mask==0."""
    B, kv_len = attention_mask.shape
    assert kv_len == q_len + past_len
    neg = torch.finfo(torch.float32).min
    i = torch.arange(q_len).unsqueeze(1) + past_len
    j = torch.arange(kv_len).unsqueeze(0)
    causal = (j > i)
    pad = (attention_mask == 0)[:, None, None, :]
    m = torch.zeros(B, 1, q_len, kv_len)
    m = m.masked_fill(causal[None, None] | pad, neg)
    return m


def attention(x, sd, prefix, cfg, attn_mask4d, position_ids, kv):
    """BailingMoeAttention.forward (:743-829).  kv: dict(k=[B,hk,T,hd]|None, v=…), appended in place."""
    B, T, _ = x.shape
    nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    qkv = F.linear(x, sd[prefix + ".query_key_value.weight"], sd.get(prefix + ".query_key_value.bias"))
    qkv = qkv.view(B, T, nh + 2 * nkv, hd)
    q, k, v = qkv.split([nh, nkv, nkv], dim=-2)
    q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    past = 0 if kv.get("k") is None else kv["k"].shape[2]
    if position_ids.dim() == 3:      # rope_scaling.type == "3D" branch (:780-782)
        cos, sin = rope3d_cos_sin(hd, cfg.rope_theta, position_ids)
        q, k = apply_rope_3d(q, k, cos, sin)
    else:
        cos, sin = rope_cos_sin(hd, cfg.rope_theta, int(position_ids.max()) + 1)
        q, k = apply_rope(q, k, cos, sin, position_ids)
    if kv.get("k") is not None:
        k = torch.cat([kv["k"], k], dim=2)
        v = torch.cat([kv["v"], v], dim=2)
    kv["k"], kv["v"] = k, v
    rep = nh // nkv
    kk = k[:, :, None].expand(B, nkv, rep, k.shape[2], hd).reshape(B, nh, k.shape[2], hd)
    vv = v[:, :, None].expand(B, nkv, rep, v.shape[2], hd).reshape(B, nh, v.shape[2], hd)
    w = torch.matmul(q / math.sqrt(hd), kk.transpose(2, 3))
    if attn_mask4d is not None:
        w = w + attn_mask4d
    w = F.softmax(w, dim=-1, dtype=torch.float32)
    o = torch.matmul(w, vv).transpose(1, 2).reshape(B, T, nh * hd)
    return F.linear(o, sd[prefix + ".dense.weight"], sd.get(prefix + ".dense.bias"))


def gate(x2d, w, cfg):
    """BailingMoeGate.forward (:505-520)."""
    logits = F.linear(x2d, w)
    scores = logits.softmax(dim=-1, dtype=torch.float32)
    tw, ti = torch.topk(scores, k=cfg.num_experts_per_tok, dim=-1)
    if cfg.num_experts_per_tok > 1 and cfg.norm_topk_prob:
        tw = tw / tw.sum(dim=-1, keepdim=True)
    return ti, tw, logits


def expert_mlp(x, sd, prefix):
    """BailingMoeMLP.forward (:483-484)."""
    g = F.linear(x, sd[prefix + ".gate_proj.weight"])
    u = F.linear(x, sd[prefix + ".up_proj.weight"])
    return F.linear(F.silu(g) * u, sd[prefix + ".down_proj.weight"])


def moe_block(x, sd, prefix, cfg, image_mask=None, grouped=None):
    """BailingMoeSparseMoeBlock.forward (:556-606) + moe_infer (:608-639).
    image_mask: bool [B,T] or None.  Rows flagged take the image_gate's routing.
    grouped (default: above 64 rows): the reference's own order of work — moe_infer sorts the (row, slot) pairs by expert and runs
    every expert once on all its rows (:613-631) — instead of the row-by-row restatement; the weighted sum over a row's slots
    keeps the top-k order either way (:633-638).  tests/test_oracle_golden.py holds the two forms to each other."""
    B, T, H = x.shape
    x2 = x.reshape(-1, H)
    ti, tw, _ = gate(x2, sd[prefix + ".gate.weight"], cfg)
    if cfg.multi_gate and image_mask is not None:
        ii, iw, _ = gate(x2, sd[prefix + ".image_gate.weight"], cfg)
        m = image_mask.reshape(-1, 1)
        ti = torch.where(m, ii, ti)
        tw = torch.where(m, iw, tw)
    if grouped is None:
        grouped = x2.shape[0] > 64
    y = torch.zeros_like(x2)
    if grouped:
        per_slot = torch.zeros(x2.shape[0], cfg.num_experts_per_tok, H)
        for e in ti.unique().tolist():
            r, kk = (ti == e).nonzero(as_tuple=True)
            per_slot[r, kk] = expert_mlp(x2[r], sd, f"{prefix}.experts.{e}")
        for kk in range(cfg.num_experts_per_tok):
            y = y + tw[:, kk:kk + 1] * per_slot[:, kk]
    else:
        for r in range(x2.shape[0]):
            acc = torch.zeros(H)
            for kk in range(cfg.num_experts_per_tok):
                e = int(ti[r, kk])
                acc = acc + tw[r, kk] * expert_mlp(x2[r:r + 1], sd, f"{prefix}.experts.{e}")[0]
            y[r] = acc
    if cfg.num_shared_experts:
        y = y + expert_mlp(x2, sd, prefix + ".shared_experts")
    return y.view(B, T, H), (ti, tw)


def decoder_layer(x, sd, li, cfg, attn_mask4d, position_ids, kv, image_mask=None):
    """BailingMoeDecoderLayer.forward (:1165-1239)."""
    p = f"model.layers.{li}"
    h = rmsnorm(x, sd[p + ".input_layernorm.weight"], cfg.rms_norm_eps)
    x = x + attention(h, sd, p + ".attention", cfg, attn_mask4d, position_ids, kv)
    h = rmsnorm(x, sd[p + ".post_attention_layernorm.weight"], cfg.rms_norm_eps)
    y, _ = moe_block(h, sd, p + ".mlp", cfg, image_mask)
    return x + y


def new_kv(cfg):
    return [dict(k=None, v=None) for _ in range(cfg.num_hidden_layers)]


def model_forward(inputs_embeds, sd, cfg, attention_mask, position_ids, kvs, image_mask=None):
    """BailingMoeModel.forward (:1391-1540), eager-attention branch (:1464-1468).
    inputs_embeds [B,T,H]; attention_mask [B, past+T] (1 = attend) or None."""
    B, T, _ = inputs_embeds.shape
    past = 0 if kvs[0].get("k") is None else kvs[0]["k"].shape[2]
    if position_ids is None:
        position_ids = torch.arange(past, past + T).unsqueeze(0).expand(B, -1)
    if attention_mask is None:
        attention_mask = torch.ones(B, past + T, dtype=torch.long)
    m4 = build_4d_mask(attention_mask, T, past)
    x = inputs_embeds.float()
    for li in range(cfg.num_hidden_layers):
        x = decoder_layer(x, sd, li, cfg, m4, position_ids, kvs[li], image_mask)
    return rmsnorm(x, sd["model.norm.weight"], cfg.rms_norm_eps)


def lm_logits(h, sd):
    """compute_logit (:1604-1620) with norm_head=False -> fp32 logits (:1816-1817)."""
    return F.linear(h, sd["lm_head.weight"]).float()


def build_cfg_rows(attention_mask, uncond_attention_mask, text_uncond_attention_mask):
    """CFG row construction, generate_image (:1867-1889).  All masks [1,T*]."""
    assert attention_mask.shape[0] == 1
    am = attention_mask
    if uncond_attention_mask is not None:
        n_c, n_u = am.shape[1], uncond_attention_mask.shape[1]
        if n_u < n_c:
            uncond_attention_mask = torch.cat((uncond_attention_mask, am[:, n_u:]), dim=1)
        am = torch.cat((am, uncond_attention_mask), dim=0)
    if text_uncond_attention_mask is not None and text_uncond_attention_mask.sum() > 0:
        n_c, n_u = am.shape[1], text_uncond_attention_mask.shape[1]
        if n_u < n_c:
            text_uncond_attention_mask = torch.cat((text_uncond_attention_mask, am[0:1, n_u:]), dim=1)
        if (text_uncond_attention_mask == uncond_attention_mask).sum() != uncond_attention_mask.numel():
            am = torch.cat((am, text_uncond_attention_mask), dim=0)
    return am


def generate_image(start_embed, kvs, attention_mask, uncond_attention_mask, text_uncond_attention_mask,
                   sd, cfg, noises, latent_to_sem, linear_proj, sem_to_pix,
                   steps=16, temperature=1.0, text_cfg=3.0, image_cfg=1.1):
    """BailingMoeForCausalLM.generate_image (:1844-1965).

    start_embed [1,1,H]: embedding of the <image> token.  kvs: per-layer caches of
    the single conditional sequence (batch 1).  noises [n_tokens+1, 32]: the noise
    the RF sampler draws at each iteration (reference: torch.randn, replayed here).
    latent_to_sem(latent[rows,1,32]) -> sem token [rows,1,D]; linear_proj(sem) ->
    [rows,1,H]; sem_to_pix(sem[rows,N,D]) -> image.  CFG scales are the inner
    function's defaults 3.0/1.1 whatever the caller passes.
    Returns dict(image, latents [n,rows,32], sem [rows,n,D], last_hidden [rows,1,H], attention_mask).
    """
    am = build_cfg_rows(attention_mask, uncond_attention_mask, text_uncond_attention_mask)
    rows = am.shape[0]
    x = start_embed
    if rows > 1:
        x = x.repeat(rows, 1, 1)
        for kv in kvs:
            kv["k"] = kv["k"].repeat(rows, 1, 1, 1)
            kv["v"] = kv["v"].repeat(rows, 1, 1, 1)
    n_tok = cfg.num_image_tokens_for_gen
    rf_sd = {k[len("diffloss."):]: v for k, v in sd.items() if k.startswith("diffloss.")}
    latents, sems = [], []
    last_hidden = None
    for ti in range(n_tok + 1):
        pos = (am.long().cumsum(-1) - 1)[:, -1:]
        hidden = model_forward(x, sd, cfg, am, pos, kvs)
        last_hidden = hidden
        z = rf_ref.vis_head(hidden[:, -1], sd)
        lat = rf_ref.sample(z, noises[ti:ti + 1], rf_sd, steps=steps, temperature=temperature, text_cfg=text_cfg, image_cfg=image_cfg)
        if ti < n_tok:
            latents.append(lat)
            sem = latent_to_sem(lat.unsqueeze(1))
            sems.append(sem)
            x = linear_proj(sem)
            am = torch.cat((am, torch.ones(rows, 1, dtype=am.dtype)), dim=-1)
    for kv in kvs:
        kv["k"] = kv["k"][0:1]
        kv["v"] = kv["v"][0:1]
    sem_all = torch.cat(sems, dim=1)
    image = sem_to_pix(sem_all)
    return dict(image=image, latents=torch.stack(latents), sem=sem_all, last_hidden=last_hidden, attention_mask=am)


def linear_proj(x, sd, prefix="linear_proj"):
    """linear_proj = Linear, (GELU, Linear)*(mlp_depth-1) (modeling_bailingmm.py:111-115)."""
    i = 0
    x = F.linear(x, sd[f"{prefix}.{i}.weight"], sd[f"{prefix}.{i}.bias"])
    i += 2
    while f"{prefix}.{i}.weight" in sd:
        x = F.linear(F.gelu(x), sd[f"{prefix}.{i}.weight"], sd[f"{prefix}.{i}.bias"])
        i += 2
    return x
