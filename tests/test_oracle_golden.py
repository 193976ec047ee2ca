"""The oracle (oracle/*.py, our fp32 CPU restatement) against outputs of the
REFERENCE ITSELF captured in tests/golden/*.npz by oracle/gen_golden.py.
CPU only; runs in seconds."""
import pytest
import torch

from ming_univision_amd import configuration as C
from ming_univision_amd.synth import synth_state_dict
from oracle import bailing_ref, mingtok_ref, rf_ref
from tests.util import checksum, llm_sd, load_golden, mingtok_sd, rel_err

TOL = 2e-5


@pytest.fixture(scope="module")
def mt():
    g = load_golden("mingtok_tiny")
    sd = mingtok_sd(g["config"], g["seed"])
    assert abs(checksum(sd) - g["checksum"]) < 1e-6 * g["checksum"], "synthetic-weight RNG drifted"
    return g, sd


def test_mingtok_encoder_tokens(mt):
    g, sd = mt
    assert rel_err(mingtok_ref.encoder_prepare_tokens(g["img"], sd), g["enc_tokens"]) < TOL
    # 128x128 input on a 64x64-trained pos-embed: bicubic interpolation with the +0.1 kludge
    assert rel_err(mingtok_ref.encoder_prepare_tokens(g["img2"], sd), g["enc_tokens2"]) < TOL


def test_mingtok_encoder_block(mt):
    g, sd = mt
    y = mingtok_ref.block(g["enc_tokens"], sd, "low_level_encoder.blocks.0.0", 2)
    assert rel_err(y, g["enc_block0"]) < TOL


def test_mingtok_forward(mt):
    g, sd = mt
    assert rel_err(mingtok_ref.encoder_forward(g["img"], sd), g["latent_raw"]) < TOL
    out = mingtok_ref.mingtok_forward(g["img"], sd)
    assert rel_err(out["latent"], g["latent"]) < TOL
    assert rel_err(out["x_norm_patchtokens"], g["sem"]) < TOL
    out2 = mingtok_ref.mingtok_forward(g["img2"], sd)
    assert rel_err(out2["latent"], g["latent2"]) < TOL
    assert rel_err(out2["x_norm_patchtokens"], g["sem2"]) < TOL


def test_mingtok_pixel_decoder(mt):
    g, sd = mt
    assert rel_err(mingtok_ref.sem_to_pix(g["sem"], sd) if False else
                   torch.nn.functional.linear(g["sem"], sd["sem_to_pix.weight"], sd["sem_to_pix.bias"]),
                   g["sem_to_pix"]) < TOL
    assert rel_err(mingtok_ref.pixel_decoder_forward(g["sem"], sd), g["recon"]) < TOL
    assert rel_err(mingtok_ref.mingtok_forward_enc_dec(g["img2"], sd), g["recon2"]) < TOL


def test_mingtok_cached_decode(mt):
    g, sd = mt
    caches = mingtok_ref.semdec_new_cache(sd)
    outs = []
    for i in range(g["dec_latent_norm"].shape[1]):
        outs.append(mingtok_ref.mingtok_feature_decoder_step(g["dec_latent_norm"][:, i:i + 1], sd, caches))
    assert rel_err(torch.cat(outs, 1), g["dec_steps"]) < TOL


@pytest.mark.parametrize("name", ["rf_tiny", "rf_tiny16"])
def test_rf_head(name):
    g = load_golden(name)
    sd_full = synth_state_dict(C.rf_param_shapes(64, 2, 64, 32, 4), g["seed"])
    sd = {k[len("diffloss."):]: v for k, v in sd_full.items()}
    assert abs(checksum(sd) - g["checksum"]) < 1e-6 * g["checksum"]
    assert rel_err(rf_ref.time_embed(torch.tensor([1000.0, 937.5, 62.5]), sd), g["temb"]) < TOL
    assert rel_err(rf_ref.net_forward(g["x"], g["t"], g["z"], sd), g["v"]) < TOL
    n = g["noise"]
    s3 = rf_ref.sample(g["z"], n[0:1], sd, steps=g["steps"])
    assert rel_err(s3, g["sample3"]) < TOL
    s2 = rf_ref.sample(g["z"][:2], n[1:2], sd, steps=g["steps"], temperature=g["sample2_temperature"])
    assert rel_err(s2, g["sample2"]) < TOL
    s1 = rf_ref.sample(g["z"][:1], n[2:3], sd, steps=g["steps"], text_cfg=1.0, image_cfg=1.0)
    assert rel_err(s1, g["sample1"]) < TOL


@pytest.fixture(scope="module")
def llm():
    g = load_golden("llm_tiny")
    sd = llm_sd(g["config"], g["rf_config"], g["seed"])
    assert abs(checksum(sd) - g["checksum"]) < 1e-6 * g["checksum"]
    cfg = bailing_ref.LLMConfig(**{k: v for k, v in g["config"].items()
                                   if k in bailing_ref.LLMConfig.__dataclass_fields__})
    return g, sd, cfg


def test_llm_pieces(llm):
    g, sd, cfg = llm
    xn = bailing_ref.rmsnorm(g["emb"], sd["model.layers.0.input_layernorm.weight"], cfg.rms_norm_eps)
    assert rel_err(xn, g["x_norm"]) < TOL
    x2 = g["moe_in"].reshape(-1, cfg.hidden_size)
    ti, tw, _ = bailing_ref.gate(x2, sd["model.layers.0.mlp.gate.weight"], cfg)
    assert torch.equal(ti, g["gate_idx"]) and rel_err(tw, g["gate_w"]) < TOL
    ti, tw, _ = bailing_ref.gate(x2, sd["model.layers.0.mlp.image_gate.weight"], cfg)
    assert torch.equal(ti, g["igate_idx"]) and rel_err(tw, g["igate_w"]) < TOL
    y, (idx, _) = bailing_ref.moe_block(g["moe_in"], sd, "model.layers.0.mlp", cfg, g["moe_image_mask"].bool())
    assert torch.equal(idx, g["moe_topk_idx"])
    assert rel_err(y, g["moe_out"]) < TOL
    # the expert-sorted form the oracle switches to above 64 rows (moe_infer's own order of work) against the same golden output
    yg, (idxg, _) = bailing_ref.moe_block(g["moe_in"], sd, "model.layers.0.mlp", cfg, g["moe_image_mask"].bool(), grouped=True)
    assert torch.equal(idxg, idx) and rel_err(yg, g["moe_out"]) < TOL and rel_err(yg, y) < 2e-6


def test_llm_prefill_and_cfg_decode(llm):
    g, sd, cfg = llm
    kvs = bailing_ref.new_kv(cfg)
    T = g["emb"].shape[1]
    h = bailing_ref.model_forward(g["emb"], sd, cfg, torch.ones(1, T, dtype=torch.long), None, kvs,
                                  g["image_mask"].bool())
    assert rel_err(h, g["hidden"]) < TOL
    assert rel_err(kvs[0]["k"], g["k0"]) < TOL and rel_err(kvs[1]["v"], g["v1"]) < TOL
    assert rel_err(bailing_ref.lm_logits(h[:, -1:], sd), g["logits"]) < TOL
    rows = 3
    for kv in kvs:
        kv["k"], kv["v"] = kv["k"].repeat(rows, 1, 1, 1), kv["v"].repeat(rows, 1, 1, 1)
    am = g["dec_mask0"].clone()
    for s in range(g["dec_in"].shape[0]):
        pos = (am.cumsum(-1) - 1)[:, -1:]
        hd = bailing_ref.model_forward(g["dec_in"][s], sd, cfg, am, pos, kvs)
        assert rel_err(hd, g["dec_hidden"][s]) < TOL, s
        am = torch.cat([am, torch.ones(rows, 1, dtype=torch.long)], 1)
    assert rel_err(rf_ref.vis_head(hd[:, -1], sd), g["vis_z"]) < TOL


@pytest.mark.parametrize("tag", ["rows3", "rows2"])
def test_generate_image(tag):
    g = load_golden("genimg_tiny")
    sd = llm_sd(g["llm_config"], g["rf_config"], g["seed"])
    tsd = mingtok_sd(g["mingtok_config"], g["seed"])
    lsd = synth_state_dict(C.linear_proj_param_shapes(128, 256, 2), g["seed"])
    assert abs(checksum(sd) + checksum(tsd) - g["checksum"]) < 1e-6 * g["checksum"]
    cfg = bailing_ref.LLMConfig(**{k: v for k, v in g["llm_config"].items()
                                   if k in bailing_ref.LLMConfig.__dataclass_fields__})
    T = g["ids"].shape[1]
    kvs = bailing_ref.new_kv(cfg)
    emb = sd["model.word_embeddings.weight"][g["ids"]]
    bailing_ref.model_forward(emb, sd, cfg, torch.ones(1, T, dtype=torch.long), None, kvs)
    start = sd["model.word_embeddings.weight"][torch.tensor([[cfg.image_start_token]])]
    caches = mingtok_ref.semdec_new_cache(tsd)
    out = bailing_ref.generate_image(
        start, kvs, g["mask"], g["uncond"], g[tag + "_tuncond"], sd, cfg, g["noises"],
        latent_to_sem=lambda lat: mingtok_ref.mingtok_feature_decoder_step(lat, tsd, caches),
        linear_proj=lambda s: bailing_ref.linear_proj(s, lsd),
        sem_to_pix=lambda s: mingtok_ref.pixel_decoder_forward(s, tsd),
        steps=int(g["rf_config"]["num_sampling_steps"]))
    assert out["image"].shape == g[tag + "_image"].shape
    assert rel_err(out["image"], g[tag + "_image"]) < 1e-4
    assert rel_err(out["last_hidden"], g[tag + "_last_hidden"]) < 1e-4
    assert torch.equal(out["attention_mask"], g[tag + "_mask_out"])
    assert kvs[0]["k"].shape[2] == g[tag + "_cache_len"]
    assert rel_err(kvs[0]["k"], g[tag + "_k0"]) < 1e-4
    assert rel_err(bailing_ref.lm_logits(out["last_hidden"][0:1], sd), g[tag + "_logits"]) < 1e-4


@pytest.mark.parametrize("mode", ["KEEP", "DROP"])
def test_multiround_state_machine_vs_reference(mode):
    """the restated multi-round flow (tests/util.OracleConversation — what the GPU tests check the HIP façade against)
    versus the REFERENCE's own MingUniVisionForConditionalGeneration.generate over three rounds (image + instruction -> image; text ->
    image; text -> text), PAST_MODE KEEP and DROP: the greedy tokens, the three masks carried to the next round, the cache length after
    every round, the generated images (all CFG rows) and layer 0's K cache at the end (every appended line, generated image tokens
    included)."""
    from tests.util import OracleConversation
    g = load_golden("multiround_tiny")
    sd = llm_sd(g["llm_config"], g["rf_config"], g["seed"])
    tsd = mingtok_sd(g["mingtok_config"], g["seed"])
    lsd = synth_state_dict(C.linear_proj_param_shapes(128, 256, 2), g["seed"])
    assert abs(checksum(sd) - g[mode + "_checksum"]) < 1e-6 * g[mode + "_checksum"]
    cfg = bailing_ref.LLMConfig(**{k: v for k, v in g["llm_config"].items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    conv = OracleConversation(sd, lsd, tsd, cfg, int(g["rf_config"]["num_sampling_steps"]), past_mode=mode, decode_pixels=True)
    n_tok = g["llm_config"]["num_image_tokens_for_gen"]
    for r, spec in enumerate(g["rounds"]):
        t = f"{mode}_r{r}_"
        ids = g[t + "ids"]
        n0 = g[t + "noise0"]
        out = conv.round(ids, g[t + "unc"], g[t + "tunc"], pixel_values=g["pixel_values"] if spec["px"] else None,
                         patch_id=g["special_ids"]["PATCH"], max_new_tokens=spec["n_new"],
                         forced_first_token=g["llm_config"]["image_start_token"] if spec["force"] else None,
                         noises=[g["noises"][n0:n0 + n_tok + 1]])
        assert out["tokens"] == g[t + "seq"][0, ids.shape[1]:].tolist(), (r, out["tokens"])
        assert torch.equal(g[t + "seq"][0, :ids.shape[1]], ids[0])
        assert conv.cache_len == g[t + "cache_len"] == conv.kvs[0]["k"].shape[2]
        for mine, ref in zip(conv.past, (g[t + "past_am"], g[t + "past_unc"], g[t + "past_tunc"])):
            assert torch.equal(mine, ref), (r, mine, ref)
        if spec["force"]:
            img = out["images"][0]["image"]
            assert img.shape == g[t + "image"].shape and rel_err(img, g[t + "image"]) < 1e-4, r
            # the reference's bookkeeping of the image step: <image> is fed against the prompt's cache, the next text token against the
            # cache + 1 + n_tok generated lines at position = cache length (modeling_bailing_moe.py:1993 `past_length`)
            tr = g[t + "trace"]
            assert tr[1, 1] + 1 + n_tok == tr[2, 1] and tr[2, 4] == tr[2, 1] and tr[2, 2] == tr[2, 1] + 1
        else:
            assert not out["images"]
    assert rel_err(conv.kvs[0]["k"], g[mode + "_k0"]) < 1e-4


def test_rope3d_matches_reference():
    """3D rotary branch (rope_scaling.type == "3D", modeling_bailing_moe.py:413-425, 463-469): the restatement against
    the reference's own functions, with distinct t / h / w position streams and with equal streams (== Legacy)."""
    g = load_golden("rope3d")
    cos, sin = bailing_ref.rope3d_cos_sin(128, g["base"], g["pos3"])
    q3, k3 = bailing_ref.apply_rope_3d(g["q"], g["k"], cos, sin)
    assert rel_err(q3, g["q3"]) < 1e-6 and rel_err(k3, g["k3"]) < 1e-6
    same = g["pos3"][:1].expand(3, -1, -1)
    cos, sin = bailing_ref.rope3d_cos_sin(128, g["base"], same)
    qs, ks = bailing_ref.apply_rope_3d(g["q"], g["k"], cos, sin)
    cl, sl = bailing_ref.rope_cos_sin(128, g["base"], 64)
    ql, kl = bailing_ref.apply_rope(g["q"], g["k"], cl, sl, g["pos3"][0])
    assert rel_err(qs, g["q_same"]) < 1e-6 and rel_err(ql, g["q_same"]) < 1e-6 and rel_err(kl, g["k_same"]) < 1e-6


def test_streamed_oracle_and_decoder_backed_state_dict(llm):
    """Test infrastructure of tests/test_gpu_fullsize.py, checked on the CPU at the tiny golden configuration: the state-dict view
    over a decoder's packed tensors returns every per-layer reference parameter (inverse of pack_experts), and the layer-outermost
    oracle walk over rows grouped by cache length equals `model_forward` row by row."""
    from types import SimpleNamespace
    from ming_univision_amd import configuration as C
    from ming_univision_amd.bailing_moe import pack_experts
    from tests.test_gpu_fullsize import _oracle_step_streamed
    from tests.util import DecoderBackedSD
    g, sd, cfg = llm
    pcfg = C.BailingMoeConfig(**g["config"])
    bsd = {k: v.to(torch.bfloat16) for k, v in sd.items()}
    layers = []
    for li in range(pcfg.num_hidden_layers):
        p = f"model.layers.{li}"
        gu, dn = pack_experts(bsd, p + ".mlp", pcfg)
        layers.append(dict(ln1=bsd[p + ".input_layernorm.weight"], wqkv=bsd[p + ".attention.query_key_value.weight"],
                           wdense=bsd[p + ".attention.dense.weight"], ln2=bsd[p + ".post_attention_layernorm.weight"],
                           gate=bsd[p + ".mlp.gate.weight"], image_gate=bsd.get(p + ".mlp.image_gate.weight"), w_gate_up=gu, w_down=dn))
    fake = SimpleNamespace(layers=layers, cfg=pcfg, n_shared=pcfg.num_shared_experts or 0, final_norm=bsd["model.norm.weight"],
                           word_embeddings=None, lm_head=None)
    view = DecoderBackedSD(fake)
    names = [k for k in sd if k.startswith("model.layers.") and "audio_gate" not in k]     # the audio gate is not on the path
    assert len(names) > 20
    for k in names:
        assert torch.equal(view[k], sd[k]), k                      # synthetic weights are bf16-rounded: the round trip is exact
    assert view.get("model.layers.0.attention.query_key_value.bias") is None
    gen = torch.Generator().manual_seed(5)
    M, t_max = 7, 12
    L, nkv, hd, H = cfg.num_hidden_layers, cfg.num_key_value_heads, cfg.head_dim, cfg.hidden_size
    lens = torch.tensor([3, 5, 3, 8, 5, 3, 8])
    kv = torch.randn(L, M, 2, nkv, t_max, hd, generator=gen) * 0.5
    x = torch.randn(M, H, generator=gen) * 0.5
    km = torch.ones(M, t_max, dtype=torch.uint8)
    km[0, 1:2] = 0
    km[3, 1:6] = 0
    pos = torch.stack([(km[m, :int(lens[m]) + 1].long().cumsum(0) - 1)[-1] for m in range(M)])
    ref, margin, nk, nv = _oracle_step_streamed(view, cfg, x, lens, km, pos, kv)
    assert margin.shape == (M,) and bool((margin >= 0).all())
    for m in range(M):
        n = int(lens[m])
        kvs = [dict(k=kv[l, m:m + 1, 0, :, :n].clone(), v=kv[l, m:m + 1, 1, :, :n].clone()) for l in range(L)]
        h = bailing_ref.model_forward(x[m:m + 1, None], sd, cfg, km[m:m + 1, :n + 1].long(), pos[m:m + 1, None], kvs)
        assert rel_err(ref[m], h[0, 0]) < 2e-6
        assert rel_err(nk[:, m], torch.stack([kvs[l]["k"][0, :, n] for l in range(L)])) < 2e-6
        assert rel_err(nv[:, m], torch.stack([kvs[l]["v"][0, :, n] for l in range(L)])) < 2e-6
