"""Configuration classes, field-compatible with the reference's.

  MingTokConfig        <- mingtok/modeling_mingtok.py:56-89 + config/config_mingtok.json
  BailingMoeConfig     <- mingunivision/configuration_bailing_moe.py:6-84
  MingUniVisionConfig  <- mingunivision/configuration_bailingmm.py:20-31
                          (+ `vishead_diffloss_config`, asserted at modeling_bailingmm.py:118)

Plain Python (no transformers dependency): `from_dict` / `to_dict` /
`from_json_file` / `from_pretrained(dir)` read the same JSON the reference's HF
configs serialise to; unknown keys are kept as attributes like PretrainedConfig does.
"""
import copy
import json
import os


class _Config:
    model_type = ""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @classmethod
    def from_dict(cls, d):
        d = dict(d)
        d.pop("model_type", None)
        return cls(**d)

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls.from_dict(json.load(f))

    @classmethod
    def from_pretrained(cls, path):
        if os.path.isdir(path):
            path = os.path.join(path, "config.json")
        return cls.from_json_file(path)

    def to_dict(self):
        out = {}
        for k, v in self.__dict__.items():
            out[k] = v.to_dict() if isinstance(v, _Config) else copy.deepcopy(v)
        out["model_type"] = self.model_type
        return out

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True)

    def __repr__(self):
        return f"{type(self).__name__} {self.to_json_string()}"


class MingTokConfig(_Config):
    """MingTok-Vision hyper-parameters (mingtok/config/config_mingtok.json)."""
    model_type = "mingtok"

    def __init__(self, low_level_encoder=None, semantic_decoder=None, pixel_decoder=None,
                 scaling_factor=8.09449291, mean=1.46817409, model_dtype="bf16",
                 pretrained_checkpoint="", **kwargs):
        self.low_level_encoder = dict(low_level_encoder or dict(
            img_size=512, patch_size=32, depth=12, embed_dim=768, ffn_layer="swiglufused", out_dim=32))
        self.semantic_decoder = dict(semantic_decoder or dict(
            in_dim=32, patch_size=32, embed_dim=1024, decoder_depth=24, ffn_layer="swiglufused"))
        self.pixel_decoder = dict(pixel_decoder or dict(
            patch_size=16, decoder_depth=24, norm_pix_loss=True, embed_dim=1024, loss_type="L1-plain"))
        self.scaling_factor = scaling_factor
        self.mean = mean
        self.model_dtype = model_dtype
        self.pretrained_checkpoint = pretrained_checkpoint
        super().__init__(**kwargs)


class BailingMoeConfig(_Config):
    """Defaults follow configuration_bailing_moe.py:9-47."""
    model_type = "bailing_moe"

    def __init__(self, vocab_size=30592, hidden_size=1024, intermediate_size=None, num_hidden_layers=24,
                 num_attention_heads=16, num_key_value_heads=0, hidden_act="silu", use_qkv_bias=False,
                 use_bias=True, rms_norm_eps=1e-05, norm_head=False, tie_word_embeddings=False,
                 embedding_dropout=0.1, attention_dropout=0.1, output_dropout=0.1, initializer_range=0.02,
                 max_position_embeddings=16384, rope_theta=10000.0, use_cache=True, use_sliding_window=False,
                 sliding_window=4096, max_window_layers=28, rope_scaling=None, pad_token_id=126081,
                 num_experts=16, num_shared_experts=0, num_experts_per_tok=2, num_image_tokens_for_gen=256,
                 norm_topk_prob=True, moe_intermediate_size=None, first_k_dense_replace=0, head_dim=None,
                 output_router_logits=False, multi_gate=False, image_patch_token=126346,
                 image_start_token=126347, eos_token_id=126081, **kwargs):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads
        self.hidden_act = hidden_act
        self.use_qkv_bias = use_qkv_bias
        self.use_bias = use_bias
        self.norm_head = norm_head
        self.rms_norm_eps = rms_norm_eps
        self.tie_word_embeddings = tie_word_embeddings
        self.embedding_dropout = embedding_dropout
        self.attention_dropout = attention_dropout
        self.output_dropout = output_dropout
        self.initializer_range = initializer_range
        self.max_position_embeddings = max_position_embeddings
        self.rope_theta = rope_theta
        self.use_cache = use_cache
        self.use_sliding_window = use_sliding_window
        self.sliding_window = sliding_window
        self.max_window_layers = max_window_layers
        self.head_dim = head_dim or self.hidden_size // self.num_attention_heads
        self.rope_scaling = rope_scaling
        self.pad_token_id = pad_token_id
        self.eos_token_id = eos_token_id
        self.num_experts = num_experts
        self.num_shared_experts = num_shared_experts
        self.num_experts_per_tok = num_experts_per_tok
        self.num_image_tokens_for_gen = num_image_tokens_for_gen
        self.norm_topk_prob = norm_topk_prob
        self.moe_intermediate_size = moe_intermediate_size
        self.first_k_dense_replace = first_k_dense_replace
        self.output_router_logits = output_router_logits
        self.multi_gate = multi_gate
        self.image_patch_token = image_patch_token
        self.image_start_token = image_start_token
        super().__init__(**kwargs)

    @classmethod
    def ming_univision_16b_a3b(cls):
        """The `llm_config` block of mingunivision/config.json:27-119 (Ling-lite 16.8B-A2.75B);
        rope_scaling forced to None: the hot path uses the Legacy rotary."""
        return cls(vocab_size=126464, hidden_size=2048, intermediate_size=5632, num_hidden_layers=28,
                   num_attention_heads=16, num_key_value_heads=4, head_dim=128, use_qkv_bias=False,
                   use_bias=False, rms_norm_eps=1e-5, rope_theta=600000, max_position_embeddings=32768,
                   num_experts=64, num_shared_experts=2, num_experts_per_tok=6, moe_intermediate_size=1408,
                   norm_topk_prob=True, multi_gate=True, first_k_dense_replace=0, initializer_range=0.006,
                   embedding_dropout=0.0, attention_dropout=0.0, output_dropout=0.0, rope_scaling=None)


DEFAULT_VISHEAD_DIFFLOSS = dict(  # defaults of setup_vishead_diffloss, modeling_bailing_moe.py:1559-1567
    diffloss_w=3072, diffloss_d=12, num_sampling_steps="16", gen_method="flow_matching_swiglu-4",
    vis_head_arch="linear2-norm")


class MingUniVisionConfig(_Config):
    model_type = "mingunivision"

    def __init__(self, mlp_depth=1, llm_config=None, vishead_diffloss_config=None, mingtok_config=None, **kwargs):
        self.llm_config = BailingMoeConfig.from_dict(llm_config) if isinstance(llm_config, dict) else llm_config
        self.mlp_depth = mlp_depth
        self.vishead_diffloss_config = dict(vishead_diffloss_config) if vishead_diffloss_config is not None else None
        # The reference hard-codes MingTok.from_pretrained("./models/MingTok-Vision")
        # (modeling_bailingmm.py:102); we let the tokenizer config ride along.
        self.mingtok_config = (MingTokConfig.from_dict(mingtok_config) if isinstance(mingtok_config, dict)
                               else mingtok_config)
        super().__init__(**kwargs)

    @classmethod
    def ming_univision_16b_a3b(cls):
        return cls(mlp_depth=2, llm_config=BailingMoeConfig.ming_univision_16b_a3b(),
                   vishead_diffloss_config=dict(DEFAULT_VISHEAD_DIFFLOSS), mingtok_config=MingTokConfig())


def swiglu_hidden(dim, mlp_ratio=4.0):
    """SwiGLUFFNFused hidden size (layers/swiglu_ffn.py:54-66; diff_loss_rf_swiglu.py:54-66)."""
    return (int(int(dim * mlp_ratio) * 2 / 3) + 7) // 8 * 8


# --------------------------------------------------------------------------
# parameter name -> shape tables (reference state-dict names)
# --------------------------------------------------------------------------
def _vit_block_shapes(prefix, D, ffn):
    s = {
        f"{prefix}.norm1.weight": (D,), f"{prefix}.norm1.bias": (D,),
        f"{prefix}.attn.qkv.weight": (3 * D, D), f"{prefix}.attn.qkv.bias": (3 * D,),
        f"{prefix}.attn.proj.weight": (D, D), f"{prefix}.attn.proj.bias": (D,),
        f"{prefix}.norm2.weight": (D,), f"{prefix}.norm2.bias": (D,),
    }
    if ffn == "swiglufused":
        h = swiglu_hidden(D)
        s.update({f"{prefix}.mlp.w12.weight": (2 * h, D), f"{prefix}.mlp.w12.bias": (2 * h,),
                  f"{prefix}.mlp.w3.weight": (D, h), f"{prefix}.mlp.w3.bias": (D,)})
    else:
        s.update({f"{prefix}.mlp.fc1.weight": (4 * D, D), f"{prefix}.mlp.fc1.bias": (4 * D,),
                  f"{prefix}.mlp.fc2.weight": (D, 4 * D), f"{prefix}.mlp.fc2.bias": (D,)})
    return s


def mingtok_param_shapes(cfg: MingTokConfig):
    enc, sem, pix = cfg.low_level_encoder, cfg.semantic_decoder, cfg.pixel_decoder
    s = {}
    De, P = enc.get("embed_dim", 1024), enc.get("patch_size", 16)
    n_pos = (enc.get("img_size", 224) // P) ** 2 + 1
    s["low_level_encoder.cls_token"] = (1, 1, De)
    s["low_level_encoder.pos_embed"] = (1, n_pos, De)
    s["low_level_encoder.patch_embed.proj.weight"] = (De, 3, P, P)
    s["low_level_encoder.patch_embed.proj.bias"] = (De,)
    for i in range(enc.get("depth", 24)):
        s.update(_vit_block_shapes(f"low_level_encoder.blocks.0.{i}", De, enc.get("ffn_layer", "mlp")))
    s["low_level_encoder.out_norm.weight"] = (De,)
    s["low_level_encoder.out_norm.bias"] = (De,)
    s["low_level_encoder.out_proj.weight"] = (enc["out_dim"], De)
    s["low_level_encoder.out_proj.bias"] = (enc["out_dim"],)
    Ds = sem.get("embed_dim", 1024)
    s["semantic_decoder.in_proj.weight"] = (Ds, sem["in_dim"])
    s["semantic_decoder.in_proj.bias"] = (Ds,)
    for i in range(sem.get("decoder_depth", 1)):
        s.update(_vit_block_shapes(f"semantic_decoder.blocks.0.{i}", Ds, sem.get("ffn_layer", "mlp")))
    s["semantic_decoder.norm.weight"] = (Ds,)
    s["semantic_decoder.norm.bias"] = (Ds,)
    Dp, Pp = pix.get("embed_dim", 1024), pix.get("patch_size", 16)
    for i in range(pix.get("decoder_depth", 1)):
        s.update(_vit_block_shapes(f"pixel_decoder.blocks.0.{i}", Dp, "mlp"))
    s["pixel_decoder.norm.weight"] = (Dp,)
    s["pixel_decoder.norm.bias"] = (Dp,)
    s["pixel_decoder.head.weight"] = (Pp * Pp * 3, Dp)
    s["pixel_decoder.head.bias"] = (Pp * Pp * 3,)
    ratio = sem.get("patch_size", 16) // Pp
    s["sem_to_pix.weight"] = (Dp * ratio * ratio, Ds)
    s["sem_to_pix.bias"] = (Dp * ratio * ratio,)
    return s


def rf_param_shapes(w, depth, z_channels, target=32, mlp_mult=4, prefix="diffloss."):
    h = swiglu_hidden(w, mlp_mult)
    p = prefix + "net."
    s = {
        p + "time_embed.mlp.0.weight": (w, 256), p + "time_embed.mlp.0.bias": (w,),
        p + "time_embed.mlp.2.weight": (w, w), p + "time_embed.mlp.2.bias": (w,),
        p + "cond_embed.weight": (w, z_channels), p + "cond_embed.bias": (w,),
        p + "input_proj.weight": (w, target), p + "input_proj.bias": (w,),
    }
    for i in range(depth):
        b = f"{p}res_blocks.{i}."
        s.update({b + "in_ln.weight": (w,), b + "in_ln.bias": (w,),
                  b + "mlp.w12.weight": (2 * h, w), b + "mlp.w12.bias": (2 * h,),
                  b + "mlp.w3.weight": (w, h), b + "mlp.w3.bias": (w,),
                  b + "adaLN_modulation.1.weight": (3 * w, w), b + "adaLN_modulation.1.bias": (3 * w,)})
    f = p + "final_layer."
    s.update({f + "linear.weight": (target, w), f + "linear.bias": (target,),
              f + "adaLN_modulation.1.weight": (2 * w, w), f + "adaLN_modulation.1.bias": (2 * w,)})
    return s


def llm_layer_param_shapes(cfg: BailingMoeConfig, li, prefix="model.layers."):
    H, hd = cfg.hidden_size, cfg.head_dim
    nq, nkv, I = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.moe_intermediate_size
    p = f"{prefix}{li}."
    s = {p + "input_layernorm.weight": (H,), p + "post_attention_layernorm.weight": (H,),
         p + "attention.query_key_value.weight": ((nq + 2 * nkv) * hd, H),
         p + "attention.dense.weight": (H, nq * hd)}
    if cfg.use_qkv_bias:
        s[p + "attention.query_key_value.bias"] = ((nq + 2 * nkv) * hd,)
    if cfg.use_bias:
        s[p + "attention.dense.bias"] = (H,)
    gates = ["gate"] + (["image_gate", "audio_gate"] if cfg.multi_gate else [])
    for g in gates:
        s[p + f"mlp.{g}.weight"] = (cfg.num_experts, H)
    for e in range(cfg.num_experts):
        s[p + f"mlp.experts.{e}.gate_proj.weight"] = (I, H)
        s[p + f"mlp.experts.{e}.up_proj.weight"] = (I, H)
        s[p + f"mlp.experts.{e}.down_proj.weight"] = (H, I)
    if cfg.num_shared_experts:
        Is = I * cfg.num_shared_experts
        s[p + "mlp.shared_experts.gate_proj.weight"] = (Is, H)
        s[p + "mlp.shared_experts.up_proj.weight"] = (Is, H)
        s[p + "mlp.shared_experts.down_proj.weight"] = (H, Is)
    return s


def llm_param_shapes(cfg: BailingMoeConfig, vishead_diffloss_config=None, latent_dim=32):
    """Names of BailingMoeForCausalLM (+ vis_head / diffloss when configured)."""
    s = {"model.word_embeddings.weight": (cfg.vocab_size, cfg.hidden_size)}
    for li in range(cfg.num_hidden_layers):
        s.update(llm_layer_param_shapes(cfg, li))
    s["model.norm.weight"] = (cfg.hidden_size,)
    s["lm_head.weight"] = (cfg.vocab_size, cfg.hidden_size)
    if vishead_diffloss_config is not None:
        c = {**DEFAULT_VISHEAD_DIFFLOSS, **vishead_diffloss_config}
        w = c["diffloss_w"]
        s["vis_head.0.weight"] = (w, cfg.hidden_size)
        s["vis_head.0.bias"] = (w,)
        s["vis_head.1.weight"] = (w,)
        s["vis_head.1.bias"] = (w,)
        s.update(rf_param_shapes(w, c["diffloss_d"], w, latent_dim, int(c["gen_method"].split("-")[1])))
    return s


def linear_proj_param_shapes(feature_dim, hidden_size, mlp_depth, prefix="linear_proj."):
    s = {prefix + "0.weight": (hidden_size, feature_dim), prefix + "0.bias": (hidden_size,)}
    for i in range(1, mlp_depth):
        s[prefix + f"{2 * i}.weight"] = (hidden_size, hidden_size)
        s[prefix + f"{2 * i}.bias"] = (hidden_size,)
    return s
