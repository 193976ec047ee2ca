"""CPU oracle (plain fp32 PyTorch) for the MingTok-Vision tokenizer path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this.

A from-scratch functional restatement of the reference algorithm, operating on a
state dict with the reference's parameter names.  Each function
cites the reference file:line it follows (paths relative to /root/reference).
Pinned against the reference itself by tests/golden/mingtok_*.npz, which
oracle/gen_golden.py produced by importing the reference in the build container.

All math is fp32, eager softmax attention, no autocast: this is the
*mathematical* reference (SURVEY.md §8c caveat i).
"""
import math

import torch
import torch.nn.functional as F


def _lin(x, sd, prefix):
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def _ln(x, sd, prefix, eps=1e-6):
    # nn.LayerNorm(eps=1e-6): vision_transformer.py:103,291 (norm_layer partial)
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def _block_prefixes(sd, root):
    """Blocks are nested as `<root>.blocks.<chunk>.<i>` (BlockChunk, vision_transformer.py:152-159)."""
    idx = set()
    for k in sd:
        if k.startswith(root + ".blocks."):
            parts = k[len(root) + 8:].split(".")
            idx.add((int(parts[0]), int(parts[1])))
    return [f"{root}.blocks.{c}.{i}" for c, i in sorted(idx, key=lambda t: t[1])]


# --------------------------------------------------------------------------
# attention (layers/attention.py)
# --------------------------------------------------------------------------
def attention(x, sd, prefix, num_heads, causal=False, kv_cache=None):
    """Attention.forward (attention.py:61-74) / CausalAttention.forward (:138-163).

    kv_cache: None, or a dict {"k": [B,h,T,hd] or None, "v": ...} that is
    appended to (DynamicCache.update semantics, attention.py:148-150).  With a
    cache and N new tokens the new token i attends to past + new[0..i]
    (flash_attn causal=True bottom-right alignment, attention.py:232).
    """
    B, N, C = x.shape
    hd = C // num_heads
    qkv = _lin(x, sd, prefix + ".qkv").reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    if kv_cache is not None:
        if kv_cache.get("k") is not None:
            k = torch.cat([kv_cache["k"], k], dim=2)
            v = torch.cat([kv_cache["v"], v], dim=2)
        kv_cache["k"], kv_cache["v"] = k, v
    attn = q @ k.transpose(-2, -1)
    if causal:
        T = k.shape[2]
        # bottom-right aligned causal mask
        i = torch.arange(N).unsqueeze(1) + (T - N)
        j = torch.arange(T).unsqueeze(0)
        attn = attn.masked_fill(j > i, float("-inf"))
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return _lin(out, sd, prefix + ".proj")


def swiglu_ffn(x, sd, prefix):
    """SwiGLUFFN.forward (layers/swiglu_ffn.py:30-34)."""
    x12 = _lin(x, sd, prefix + ".w12")
    x1, x2 = x12.chunk(2, dim=-1)
    return _lin(F.silu(x1) * x2, sd, prefix + ".w3")


def gelu_mlp(x, sd, prefix):
    """Mlp.forward (layers/mlp.py:34-40), exact-erf GELU."""
    return _lin(F.gelu(_lin(x, sd, prefix + ".fc1")), sd, prefix + ".fc2")


def block(x, sd, prefix, num_heads, causal=False, kv_cache=None):
    """Block.forward (layers/block.py:80-105) / CausalBlock.forward (:301-327).
    init_values=None, drop_path=0 => LayerScale/DropPath are identity."""
    x = x + attention(_ln(x, sd, prefix + ".norm1"), sd, prefix + ".attn", num_heads, causal, kv_cache)
    h = _ln(x, sd, prefix + ".norm2")
    if (prefix + ".mlp.w12.weight") in sd:
        x = x + swiglu_ffn(h, sd, prefix + ".mlp")
    else:
        x = x + gelu_mlp(h, sd, prefix + ".mlp")
    return x


# --------------------------------------------------------------------------
# low-level encoder (vision_transformer.py:50-233)
# --------------------------------------------------------------------------
def interpolate_pos_encoding(pos_embed, npatch, w, h, patch_size, interpolate_offset=0.1):
    """VisionTransformerEncoder.interpolate_pos_encoding (vision_transformer.py:183-215).
    The class row is the LAST row of pos_embed (:190-191)."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    pos_embed = pos_embed.float()
    patch_pos = pos_embed[:, :-1]
    class_pos = pos_embed[:, -1]
    dim = pos_embed.shape[-1]
    w0, h0 = w // patch_size, h // patch_size
    M = int(math.sqrt(N))
    assert N == M * M
    sx = float(w0 + interpolate_offset) / M
    sy = float(h0 + interpolate_offset) / M
    patch_pos = F.interpolate(
        patch_pos.reshape(1, M, M, dim).permute(0, 3, 1, 2),
        mode="bicubic", antialias=False, scale_factor=(sx, sy))
    assert (w0, h0) == patch_pos.shape[-2:]
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((patch_pos, class_pos.unsqueeze(0)), dim=1)


def encoder_prepare_tokens(img, sd, root="low_level_encoder"):
    """PatchEmbed.forward (layers/patch_embed.py:69-82) + prepare_tokens
    (vision_transformer.py:218-223): conv k=s=P, cls appended at the END, + pos."""
    w_ = sd[root + ".patch_embed.proj.weight"]
    P = w_.shape[-1]
    B, _, W, H = img.shape
    x = F.conv2d(img, w_, sd[root + ".patch_embed.proj.bias"], stride=P)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((x, sd[root + ".cls_token"].expand(B, -1, -1)), dim=1)
    return x + interpolate_pos_encoding(sd[root + ".pos_embed"], x.shape[1] - 1, W, H, P)


def encoder_out_layer(x, sd, root="low_level_encoder"):
    """forward_out_layer (vision_transformer.py:173-178): group-mean shortcut
    (D -> out_dim by mean of D/out_dim consecutive channels) + Linear(GELU(LN(x)))."""
    out_dim = sd[root + ".out_proj.weight"].shape[0]
    B, N, D = x.shape
    shortcut = x.reshape(B, N, out_dim, D // out_dim).mean(-1)
    y = _lin(F.gelu(_ln(x, sd, root + ".out_norm")), sd, root + ".out_proj")
    return shortcut + y


def encoder_forward(img, sd, root="low_level_encoder"):
    """VisionTransformerEncoder.forward (vision_transformer.py:225-233) -> latent [B,N+1,out_dim]."""
    x = encoder_prepare_tokens(img, sd, root)
    D = x.shape[-1]
    for p in _block_prefixes(sd, root):
        x = block(x, sd, p, D // 64)
    return encoder_out_layer(x, sd, root)


# --------------------------------------------------------------------------
# semantic decoder (TransformerDecoder with CausalBlock, vision_transformer.py:373-451)
# --------------------------------------------------------------------------
def semdec_in_projection(latent, sd, root="semantic_decoder"):
    """forward_in_projection_layer (vision_transformer.py:373-380):
    Linear(in_dim->D) + shortcut that repeats each input channel D/in_dim times."""
    w = sd[root + ".in_proj.weight"]
    D, in_dim = w.shape
    shortcut = latent.unsqueeze(-1).repeat(1, 1, 1, D // in_dim).reshape(*latent.shape[:2], D)
    return _lin(latent, sd, root + ".in_proj") + shortcut


def semdec_forward(latent, sd, root="semantic_decoder", kv_caches=None):
    """forward_features (vision_transformer.py:382-451).

    latent: [B,N,in_dim] RAW (un-normalised) latent.  kv_caches: None (full
    causal pass) or a list of per-layer dicts (decode with cache, use_cache=True).
    Returns x_norm [B,N,D] (all rows; the caller drops the cls row, :431-439).
    """
    x = semdec_in_projection(latent, sd, root)
    D = x.shape[-1]
    for li, p in enumerate(_block_prefixes(sd, root)):
        x = block(x, sd, p, D // 64, causal=True, kv_cache=None if kv_caches is None else kv_caches[li])
    return _ln(x, sd, root + ".norm")


def semdec_new_cache(sd, root="semantic_decoder"):
    return [dict(k=None, v=None) for _ in _block_prefixes(sd, root)]


# --------------------------------------------------------------------------
# pixel decoder (modeling_mingtok.py:179-196; vision_transformer.py:515-527,572-597)
# --------------------------------------------------------------------------
def sem_to_pix(sem, sd, ratio=2):
    """sem_to_pix Linear + rearrange "b (h w) (x y c) -> b (h x w y) c" (modeling_mingtok.py:183-188)."""
    x = _lin(sem, sd, "sem_to_pix")
    B, N, C = x.shape
    h = w = int(math.sqrt(N))
    c = C // (ratio * ratio)
    x = x.reshape(B, h, w, ratio, ratio, c).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, h * ratio * w * ratio, c)


def unpatchify(x, p):
    """TransformerDecoder.unpatchify (vision_transformer.py:515-527)."""
    B, L, _ = x.shape
    h = w = int(L ** 0.5)
    assert h * w == L
    x = x.reshape(B, h, w, p, p, 3)
    x = torch.einsum("nhwpqc->nchpwq", x)
    return x.reshape(B, 3, h * p, w * p)


def pixel_decoder_forward(sem, sd, sem_patch=32, pix_patch=16, root="pixel_decoder"):
    """MingTok.forward_pixel_decoder (modeling_mingtok.py:179-196): sem_to_pix,
    24 bidirectional Blocks with GELU MLP and NO positional embedding, LN, head,
    unpatchify, clamp to [-1,1]."""
    x = sem_to_pix(sem, sd, sem_patch // pix_patch)
    D = x.shape[-1]
    for p in _block_prefixes(sd, root):
        x = block(x, sd, p, D // 64)
    x = _lin(_ln(x, sd, root + ".norm"), sd, root + ".head")
    return unpatchify(x, pix_patch).clamp(-1, 1)


# --------------------------------------------------------------------------
# MingTok (modeling_mingtok.py:150-177)
# --------------------------------------------------------------------------
MEAN = 1.46817409          # mingtok/config/config_mingtok.json:26
SCALING_FACTOR = 8.09449291  # mingtok/config/config_mingtok.json:25


def mingtok_forward(img, sd, mean=MEAN, scale=SCALING_FACTOR):
    """MingTok.forward (modeling_mingtok.py:156-163)."""
    latent = encoder_forward(img, sd)
    x_norm = semdec_forward(latent, sd)
    return {"x_norm_patchtokens": x_norm[:, :-1], "latent": (latent - mean) / scale}


def mingtok_forward_enc_dec(img, sd):
    """MingTok.forward_enc_dec (modeling_mingtok.py:150-153)."""
    return pixel_decoder_forward(mingtok_forward(img, sd)["x_norm_patchtokens"], sd)


def mingtok_feature_decoder_step(latent_norm, sd, kv_caches, mean=MEAN, scale=SCALING_FACTOR):
    """MingTok.forward_feature_decoder (modeling_mingtok.py:165-174): de-normalise the
    generated latent then one cached causal step.  latent_norm [B,1,32] -> [B,1,D]."""
    return semdec_forward(latent_norm * scale + mean, sd, kv_caches=kv_caches)
