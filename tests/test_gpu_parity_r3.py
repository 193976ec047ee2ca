"""Round-3 parity cases ("close the parity holes"), all through the C ABI against the CPU oracle / the reference's
golden vectors:

  * the bench's operating point: full-width generation with 400 and 1536 CFG rows in ONE lock-step group (SwiGLU-epilogue w12,
    split-K-3 w3, experts with a full + a remainder row tile), image 0 against the oracle;
  * image -> text end to end (BASELINE configs[2]): pixel_values -> extract_image_feature -> prompt_wrap_navit -> prefill with
    the image-gate rows -> greedy tokens, tiny and full-width; the ValueError on a token / feature count mismatch;
  * the fp32-class regime of the batched paths (MingTok encode / semantic decoder / pixel decoder, long-prompt prefill) at
    north_star's 1e-3, and the bf16 regime's measured distance from it;
  * full-size MingTok at BASELINE configs[1] (64 x 256^2): rel-err against the oracle and |dPSNR| <= 0.1 dB;
  * PAST_MODE=KEEP, and an EOS in the middle of a speculative decode chunk.
"""
import os

import pytest
import torch

from ming_univision_amd import configuration as C
from ming_univision_amd.synth import synth_state_dict
from tests.util import llm_sd, load_golden, mingtok_sd, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-3          # north_star: relative, against the fp32 reference path
TOL_BF16 = 3e-2     # bf16-activation regime (the reference's own autocast precision)


def _dev(sd):
    return {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}


def psnr(a, b):
    mse = float(((a.double().cpu() - b.double().cpu()) ** 2).mean())
    return 10 * torch.log10(torch.tensor(4.0 / max(mse, 1e-20))).item()


# ------------------------------------------------------------------------------------------------------------------------
# (a) the bench's operating point against the oracle
# ------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_ref():
    """Full-width model (16B-A3B layer shapes, full RF head, full semantic decoder; 2 LLM layers, 3 visual tokens, 2 CFG rows) and
    the fp32 oracle's generate_image for ONE prompt / noise."""
    from oracle import bailing_ref, mingtok_ref
    from ming_univision_amd.mingtok import MingTok
    from ming_univision_amd.rf_head import RectifiedFlowHead
    seed = 5
    d = C.BailingMoeConfig.ming_univision_16b_a3b().to_dict()
    d.pop("model_type", None)
    d.update(num_hidden_layers=2, vocab_size=1024, num_image_tokens_for_gen=3, image_start_token=1000, pad_token_id=0)
    rf_cfg = dict(C.DEFAULT_VISHEAD_DIFFLOSS)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    sd = llm_sd(d, rf_cfg, seed)
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in d.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    cfg = C.BailingMoeConfig(**d)
    dsd = _dev(sd)
    rf = RectifiedFlowHead(dsd, cfg.hidden_size, rf_cfg)
    lsd = synth_state_dict(C.linear_proj_param_shapes(1024, cfg.hidden_size, 2), seed)
    dl = _dev(lsd)
    tok = MingTok(C.MingTokConfig(), device="cuda", seed=seed,
                  linear_proj=[(dl["linear_proj.0.weight"], dl["linear_proj.0.bias"]), (dl["linear_proj.2.weight"], dl["linear_proj.2.bias"])])
    tsd = {k: v.float().cpu() for k, v in tok.sd.items()}
    g = torch.Generator().manual_seed(1)
    T = 12
    ids = torch.randint(0, 900, (1, T), generator=g)
    noises = torch.randn(cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    am = torch.ones(1, T + 1, dtype=torch.long)
    un = am.clone(); un[0, 2:T - 2] = 0
    kvs = bailing_ref.new_kv(ocfg)
    bailing_ref.model_forward(sd["model.word_embeddings.weight"][ids], sd, ocfg, torch.ones(1, T, dtype=torch.long), None, kvs)
    start = sd["model.word_embeddings.weight"][torch.tensor([[cfg.image_start_token]])]
    caches = mingtok_ref.semdec_new_cache(tsd)
    ref = bailing_ref.generate_image(
        start, kvs, am, un, un.clone(), sd, ocfg, noises,
        latent_to_sem=lambda lat: mingtok_ref.mingtok_feature_decoder_step(lat, tsd, caches),
        linear_proj=lambda s: bailing_ref.linear_proj(s, lsd), sem_to_pix=lambda s: None, steps=int(rf_cfg["num_sampling_steps"]))
    assert ref["last_hidden"].shape[0] == 2
    return dict(cfg=cfg, dsd=dsd, rf=rf, tok=tok, ids=ids, noises=noises, am=am, un=un, ref=ref, T=T, g=g)


@pytest.mark.parametrize("n_images", [200, 768, 1024])
def test_full_width_bench_operating_point_vs_oracle(full_ref, n_images):
    """400, 1536 and 2048 (the maximum) CFG rows in one lock-step group — bench.py's default is 768 images = 1536 rows: w12 with the SwiGLU epilogue
    (>= 385 rows), w3 split-K, grouped experts with full and remainder row tiles, prompts prefilled in lock-step on the wide route.
    Image 0 carries the oracle's prompt and noise; the other images have their own noise."""
    from ming_univision_amd.bailing_moe import BailingMoeDecoder, generate_images
    f = full_ref
    cfg, B, T, R = f["cfg"], n_images, f["T"], 2
    dec = BailingMoeDecoder.from_state_dict(cfg, f["dsd"], t_max=32, n_seq=R * B)
    emb = dec.embed(f["ids"][0].cuda())
    dec.prefill_many(emb.unsqueeze(0).expand(B, T, emb.shape[1]).contiguous(), [R * i for i in range(B)])
    g = torch.Generator().manual_seed(100 + B)
    nb = torch.randn(B, cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    nb[0] = f["noises"]
    out = generate_images(dec, f["rf"], f["tok"], dec.embed(torch.tensor([cfg.image_start_token]).cuda()), [T] * B, [f["am"]] * B,
                          [f["un"]] * B, [f["un"].clone()] * B, nb.cuda(), decode_pixels=False, n_groups=1)
    ref = f["ref"]
    errs = (rel_err(out["latents"][0], ref["latents"][:, 0]), rel_err(out["sem"][0], ref["sem"][0]),
            rel_err(out["last_hidden"][:R], ref["last_hidden"][:, 0]))
    print("bench operating point, %d rows vs oracle: latents %.2e sem %.2e hidden %.2e" % ((R * B,) + errs))
    assert torch.isfinite(out["latents"]).all()
    assert max(errs) < TOL, errs


# ------------------------------------------------------------------------------------------------------------------------
# (b) image -> text end to end
# ------------------------------------------------------------------------------------------------------------------------
def _oracle_image_to_text(sd, lsd, tsd, ocfg, ids, pixel_values, patch_id, n_new):
    """The reference's understanding path restated with the oracle's pieces (modeling_bailingmm.py:131-138, 152-204, 237-248):
    MingTok.forward -> x_norm_patchtokens -> linear_proj -> masked_scatter at `<imagePatch>` -> model forward with image_mask
    -> greedy tokens.  Returns (tokens, last hidden of the prompt, image features)."""
    from oracle import bailing_ref, mingtok_ref
    feat = mingtok_ref.mingtok_forward(pixel_values, tsd)["x_norm_patchtokens"]
    img = bailing_ref.linear_proj(feat.float(), lsd).reshape(-1, ocfg.hidden_size)
    emb = sd["model.word_embeddings.weight"][ids].clone()
    mask = ids == patch_id
    emb[mask] = img
    kvs = bailing_ref.new_kv(ocfg)
    h = bailing_ref.model_forward(emb, sd, ocfg, None, None, kvs, image_mask=mask)
    h_prompt = h[:, -1]
    toks = []
    for _ in range(n_new):
        t = int(bailing_ref.lm_logits(h[:, -1:], sd).argmax())
        toks.append(t)
        h = bailing_ref.model_forward(sd["model.word_embeddings.weight"][torch.tensor([[t]])], sd, ocfg, None, None, kvs)
    return toks, h_prompt, img


def _facade(llm_cfg, rf_cfg, tcfg_dict, seed, t_max, proj_in, sd=None, tsd=None, lsd=None):
    from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration
    cfg = C.MingUniVisionConfig(mlp_depth=2, llm_config=llm_cfg, vishead_diffloss_config=rf_cfg, mingtok_config=tcfg_dict)
    ckpt = {"model." + k: v for k, v in sd.items()}
    ckpt.update({"vision." + k: v for k, v in tsd.items()})
    ckpt.update(lsd)
    return MingUniVisionForConditionalGeneration(cfg, state_dict=ckpt, seed=seed, t_max=t_max)


def test_image_to_text_tiny_vs_oracle():
    """Tiny model, one 64 x 64 image (4 `<imagePatch>` tokens) inside a 12-token prompt: features, prompt hidden state and the
    greedy continuation against the oracle; the count mismatch raises the reference's ValueError; prompt_wrap_navit's forms."""
    from oracle import bailing_ref
    g = load_golden("genimg_tiny")
    llm_cfg = dict(g["llm_config"]); llm_cfg.update(eos_token_id=1, image_patch_token=498)
    sd = llm_sd(g["llm_config"], g["rf_config"], g["seed"])
    tsd = mingtok_sd(g["mingtok_config"], g["seed"])
    lsd = synth_state_dict(C.linear_proj_param_shapes(128, 256, 2), g["seed"])
    # the oracle sees the bf16-rounded weights the device holds
    sd_r, tsd_r, lsd_r = ({k: v.to(torch.bfloat16).float() for k, v in d.items()} for d in (sd, tsd, lsd))
    model = _facade(llm_cfg, g["rf_config"], g["mingtok_config"], g["seed"], 64, 128, sd, tsd, lsd)
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in llm_cfg.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    gen = torch.Generator().manual_seed(3)
    px = torch.rand(1, 3, 64, 64, generator=gen) * 2 - 1
    ids = torch.tensor([[5, 9, 11, 498, 498, 498, 498, 13, 17, 21, 30, 44]])
    ref_toks, ref_h, ref_img = _oracle_image_to_text(sd_r, lsd_r, tsd_r, ocfg, ids, px, 498, 5)
    feats = model.extract_image_feature(px.cuda())
    assert feats.shape == (4, 256) and rel_err(feats, ref_img) < TOL
    assert rel_err(model.extract_image_feature(px.cuda(), precision="bf16"), ref_img) < TOL_BF16
    seqs = model.generate(input_ids=ids, pixel_values=px, max_new_tokens=5)
    assert seqs[0, ids.shape[1]:].tolist() == ref_toks
    # prompt_wrap_navit / prompt_wrap_vision: forms and the error of modeling_bailingmm.py:163-166
    plain = model.prompt_wrap_navit(ids)
    assert torch.is_tensor(plain) and plain.shape == (12, 256)
    emb, im, am = model.prompt_wrap_navit(ids, feats)
    assert am is None and im.dtype == torch.bool and im.shape == ids.shape and im[0].tolist() == (ids[0] == 498).tolist()
    assert torch.equal(emb[3:7], feats) and torch.equal(emb[:3], plain[:3]) and torch.equal(emb[7:], plain[7:])
    emb2, none = model.prompt_wrap_vision(ids, plain, None)
    assert none is None and emb2 is plain
    with pytest.raises(ValueError, match="Image features and image tokens do not match: tokens: 4, features 3"):
        model.prompt_wrap_vision(ids, plain, feats[:3])
    with pytest.raises(ValueError, match="tokens: 3, features 4"):
        model.generate(input_ids=ids[:, :6], pixel_values=px, max_new_tokens=1)


def test_image_to_text_full_width_vs_oracle():
    """BASELINE configs[2] at the production width: a 512 x 512 image = 256 `<imagePatch>` tokens through the full-size MingTok
    and linear_proj, a 276-token prompt through two 16B-A3B layers with the image-gate rows, greedy tokens.  fp32-class regime
    (default) within 1e-3 of the oracle; the bf16 regime (the reference's autocast precision) measured beside it."""
    from oracle import bailing_ref
    seed = 21
    d = C.BailingMoeConfig.ming_univision_16b_a3b().to_dict()
    d.pop("model_type", None)
    d.update(num_hidden_layers=2, vocab_size=1024, num_image_tokens_for_gen=4, image_start_token=1000, image_patch_token=1001,
             pad_token_id=0, eos_token_id=1)
    rf_cfg = dict(C.DEFAULT_VISHEAD_DIFFLOSS)
    tcfg = C.MingTokConfig()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    sd = llm_sd(d, rf_cfg, seed)
    tsd = synth_state_dict(C.mingtok_param_shapes(tcfg), seed)
    lsd = synth_state_dict(C.linear_proj_param_shapes(1024, d["hidden_size"], 2), seed)
    sd_r, tsd_r, lsd_r = ({k: v.to(torch.bfloat16).float() for k, v in x.items()} for x in (sd, tsd, lsd))
    from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration
    cfg = C.MingUniVisionConfig(mlp_depth=2, llm_config=d, vishead_diffloss_config=rf_cfg)
    ckpt = {"model." + k: v for k, v in sd.items()}
    ckpt.update({"vision." + k: v for k, v in tsd.items()})
    ckpt.update(lsd)
    model = MingUniVisionForConditionalGeneration(cfg, state_dict=ckpt, seed=seed, t_max=320)
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in d.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    gen = torch.Generator().manual_seed(8)
    px = torch.rand(1, 3, 512, 512, generator=gen) * 2 - 1
    ids = torch.cat([torch.randint(2, 900, (1, 8), generator=gen), torch.full((1, 256), 1001), torch.randint(2, 900, (1, 12), generator=gen)], 1)
    n_new = 4
    ref_toks, ref_h, ref_img = _oracle_image_to_text(sd_r, lsd_r, tsd_r, ocfg, ids, px, 1001, n_new)
    feats = model.extract_image_feature(px.cuda())
    e_feat = rel_err(feats, ref_img)
    e_feat_bf16 = rel_err(model.extract_image_feature(px.cuda(), precision="bf16"), ref_img)
    # prompt hidden state: fp32-class prefill (default) and the bf16 MFMA prefill, both from the fp32-class image features
    emb, im, _ = model.prompt_wrap_navit(ids, feats)
    h32 = model.model.prefill_wide(emb, seq=0, past=0, image_mask=im.reshape(-1))[-1:]
    h16 = model.model.prefill_mfma(emb, seq=1, past=0, image_mask=im.reshape(-1))
    e_h32, e_h16 = rel_err(h32, ref_h), rel_err(h16, ref_h)
    print("image->text full width: features %.2e (bf16 regime %.2e); prompt hidden fp32-class %.2e, bf16 MFMA prefill %.2e"
          % (e_feat, e_feat_bf16, e_h32, e_h16))
    assert e_feat < TOL and e_h32 < TOL
    assert e_feat_bf16 < TOL_BF16 and e_h16 < TOL_BF16
    seqs = model.generate(input_ids=ids, pixel_values=px, max_new_tokens=n_new)
    assert seqs[0, ids.shape[1]:].tolist() == ref_toks
    # downstream effect of the bf16 regime on the greedy continuation (measured): same tokens on this model
    model.reset_inner_state()
    model.understanding_precision = "bf16"
    seqs16 = model.generate(input_ids=ids, pixel_values=px, max_new_tokens=n_new)
    n_same = sum(int(a == b) for a, b in zip(seqs16[0, ids.shape[1]:].tolist(), ref_toks))
    print("bf16 regime greedy tokens equal to the oracle's: %d of %d" % (n_same, n_new))


def test_edit_round_image_in_image_out_vs_oracle(tmp_path):
    """The editing round of BASELINE configs[4] on one GPU (tiny model): an input image (4 `<imagePatch>` tokens through MingTok +
    linear_proj, image-gate rows in the prefill) and an instruction, processor-style CFG masks — uncond hides the whole user turn,
    text-uncond keeps its image tokens, so generate_image runs THREE distinct CFG rows (:1867-1889) — then the generated image.
    Latents, semantic tokens and the last hidden states against the oracle driven the way modeling_bailingmm.py:206-301 drives the
    reference; then a follow-up text round on top of the cache (multi-round state)."""
    from oracle import bailing_ref, mingtok_ref
    from ming_univision_amd.processing import cfg_attention_masks
    g = load_golden("genimg_tiny")
    llm_cfg = dict(g["llm_config"]); llm_cfg.update(eos_token_id=1, image_patch_token=498)
    sd = llm_sd(g["llm_config"], g["rf_config"], g["seed"])
    tsd = mingtok_sd(g["mingtok_config"], g["seed"])
    lsd = synth_state_dict(C.linear_proj_param_shapes(128, 256, 2), g["seed"])
    sd_r, tsd_r, lsd_r = ({k: v.to(torch.bfloat16).float() for k, v in d.items()} for d in (sd, tsd, lsd))
    model = _facade(llm_cfg, g["rf_config"], g["mingtok_config"], g["seed"], 64, 128, sd, tsd, lsd)
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in llm_cfg.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    gen = torch.Generator().manual_seed(11)
    px = torch.rand(1, 3, 64, 64, generator=gen) * 2 - 1
    # <role>HUMAN</role> <image> 4 x <imagePatch> </image> text... <role>ASSISTANT</role>   (ids of a 512-word toy vocabulary)
    ROLE, ROLE_E, HUMAN, ASSIST, IMG, IMG_E, PATCH = 490, 491, 300, 301, 496, 497, 498
    ids = [ROLE, HUMAN, ROLE_E, IMG, PATCH, PATCH, PATCH, PATCH, IMG_E, 21, 22, 23, 24, ROLE, ASSIST, ROLE_E]
    unc, tunc = cfg_attention_masks(ids, [ROLE, HUMAN, ROLE_E], [ROLE, ASSIST, ROLE_E], {IMG, IMG_E, PATCH})
    assert unc != tunc and sum(tunc) > sum(unc)
    ids_t = torch.tensor([ids]); T = len(ids)
    am = torch.ones(1, T, dtype=torch.long)
    unc_t, tunc_t = torch.tensor([unc]), torch.tensor([tunc])
    n_tok = llm_cfg["num_image_tokens_for_gen"]
    # the noise generate() will draw (torch.randn on its generator, diff_loss_rf_swiglu.py:117-122) is replayed for the oracle
    model.noise_generator.manual_seed(123)
    noises = torch.randn(n_tok + 1, 32, generator=model.noise_generator, device="cuda").cpu()
    model.noise_generator.manual_seed(123)
    # ---- oracle, driven like the reference: features -> scatter -> prefill with image_mask -> generate_image (3 rows) ----
    feat = mingtok_ref.mingtok_forward(px, tsd_r)["x_norm_patchtokens"]
    img = bailing_ref.linear_proj(feat.float(), lsd_r).reshape(-1, ocfg.hidden_size)
    emb = sd_r["model.word_embeddings.weight"][ids_t].clone()
    mask = ids_t == PATCH
    emb[mask] = img
    kvs = bailing_ref.new_kv(ocfg)
    bailing_ref.model_forward(emb, sd_r, ocfg, None, None, kvs, image_mask=mask)
    caches = mingtok_ref.semdec_new_cache(tsd_r)
    one = torch.ones(1, 1, dtype=torch.long)
    ref = bailing_ref.generate_image(
        sd_r["model.word_embeddings.weight"][torch.tensor([[llm_cfg["image_start_token"]]])], kvs, torch.cat((am, one), 1), unc_t, tunc_t,
        sd_r, ocfg, noises, latent_to_sem=lambda lat: mingtok_ref.mingtok_feature_decoder_step(lat, tsd_r, caches),
        linear_proj=lambda s_: bailing_ref.linear_proj(s_, lsd_r),
        sem_to_pix=lambda s_: mingtok_ref.pixel_decoder_forward(s_, tsd_r), steps=int(g["rf_config"]["num_sampling_steps"]))
    assert ref["last_hidden"].shape[0] == 3
    # ---- HIP path through the facade ----
    seqs = model.generate(input_ids=ids_t, attention_mask=am, uncond_attention_mask=unc_t, text_uncond_attention_mask=tunc_t, pixel_values=px,
                          max_new_tokens=2, forced_first_token=llm_cfg["image_start_token"], output_image_prefix=str(tmp_path / "edit"))
    out = model.last_generation
    assert out["last_hidden"].shape[0] == 3
    errs = (rel_err(out["latents"], ref["latents"][:, 0]), rel_err(out["sem"], ref["sem"][0]), rel_err(out["last_hidden"], ref["last_hidden"][:, 0]))
    print("edit round (image in, 3 CFG rows, image out) vs oracle: latents %.2e sem %.2e hidden %.2e" % errs)
    assert max(errs) < TOL, errs
    assert psnr(model.last_image[0], ref["image"][0]) > 40.0            # the pixel decoder of generate_image runs in the bf16 regime
    assert os.path.exists(str(tmp_path / "edit.png"))
    assert model.past_len == T + 1 + n_tok and seqs.shape[1] == T + 2
    # follow-up text round on the cache the edit left (DROP policy: the uncond row forgets the generated tokens)
    nxt = model.generate(input_ids=torch.tensor([[ROLE, HUMAN, ROLE_E, 31, 32, ROLE, ASSIST, ROLE_E]]), max_new_tokens=3)
    assert nxt.shape[1] == 8 + 3 and model.past_len == T + 1 + n_tok + 8 + 2


# ------------------------------------------------------------------------------------------------------------------------
# (d) fp32-class regime of the batched paths on the reference's golden vectors
# ------------------------------------------------------------------------------------------------------------------------
def test_mingtok_fp32_regime_vs_reference_golden():
    """MingTok encode (+ interpolated pos-embed), semantic decoder and pixel decoder in the fp32-class regime against the
    REFERENCE's outputs (tests/golden/mingtok_tiny.npz), at north_star's 1e-3 — the bf16 regime sits at ~1e-2."""
    from ming_univision_amd.mingtok import MingTok
    g = load_golden("mingtok_tiny")
    tok = MingTok(C.MingTokConfig(**g["config"]), state_dict=mingtok_sd(g["config"], g["seed"]), precision="fp32")
    out = tok.forward(g["img"].cuda())
    e = dict(latent=rel_err(out["latent"], g["latent"]), sem=rel_err(out["x_norm_patchtokens"], g["sem"]))
    out2 = tok.forward(g["img2"].cuda())
    e.update(latent2=rel_err(out2["latent"], g["latent2"]), sem2=rel_err(out2["x_norm_patchtokens"], g["sem2"]))
    rec = tok.forward_pixel_decoder(g["sem"].cuda())
    e.update(recon=rel_err(rec, g["recon"]))
    rec2 = tok.forward_enc_dec(g["img2"].cuda())
    e.update(recon2=rel_err(rec2, g["recon2"]))
    print("MingTok fp32-class regime vs reference:", {k: "%.1e" % v for k, v in e.items()})
    assert max(e.values()) < TOL, e
    assert psnr(rec2, g["recon2"]) > 70.0
    b = tok.forward(g["img"].cuda(), precision="bf16")
    assert rel_err(b["x_norm_patchtokens"], g["sem"]) < TOL_BF16


def test_prefill_wide_vs_reference_golden():
    """Long-prompt prefill in the fp32-class regime (wide route, image-gate rows) against the reference's golden hidden states and
    the chunked fp32 prefill; then the reference's CFG decode steps on top of the cache it wrote."""
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    g = load_golden("llm_tiny")
    sd = llm_sd(g["config"], g["rf_config"], g["seed"])
    cfg = C.BailingMoeConfig(**g["config"])
    dec = BailingMoeDecoder.from_state_dict(cfg, _dev(sd), t_max=64, n_seq=3)
    assert dec.max_rows() > 64, "the tiny configuration must be able to take the wide route"
    emb = g["emb"][0].cuda()
    T = emb.shape[0]
    # the golden prompt is short: put n copies of it into one row block of > 64 rows so that the pass takes the wide route
    n = 64 // T + 2
    dec.ensure_sequences(n)
    hs = dec.prefill_ragged([emb] * n, list(range(n)), past=0, image_masks=[g["image_mask"][0]] * n)
    assert n * T > 64
    assert rel_err(hs[0:1], g["hidden"][0, -1:]) < TOL
    assert rel_err(hs[n - 1:n], g["hidden"][0, -1:]) < TOL
    kv_wide = dec.kv_cache[:, n - 1, :, :, :T].clone()
    h_ref = dec.prefill(emb, seq=0, past=0, image_mask=g["image_mask"][0], chunk=8)
    assert rel_err(hs[0:1], h_ref[-1:]) < 2e-4
    assert rel_err(kv_wide, dec.kv_cache[:, 0, :, :, :T]) < 2e-4
    # one sequence alone through prefill_wide (the form generate() calls); a short prompt stays on the <= 64-row kernels
    h_one = dec.prefill_wide(emb, seq=1, past=0, image_mask=g["image_mask"][0])
    assert rel_err(h_one[-1:], g["hidden"][0, -1:]) < TOL


# ------------------------------------------------------------------------------------------------------------------------
# (c) full-size MingTok at BASELINE configs[1]
# ------------------------------------------------------------------------------------------------------------------------
def test_mingtok_full_size_batch64_256_vs_oracle():
    """MingTok-Vision (697.7 M parameters) encode -> decode of 64 images at 256 x 256 (BASELINE configs[1]).  The fp32 oracle runs
    on the first 4 images (host time); against it: rel-err of latent / x_norm_patchtokens / reconstruction in both regimes and
    north_star's reconstruction criterion |PSNR_hip - PSNR_oracle| <= 0.1 dB (PSNR against the input image)."""
    from oracle import mingtok_ref
    from ming_univision_amd.mingtok import MingTok
    torch.set_num_threads(min(32, torch.get_num_threads()))
    tcfg = C.MingTokConfig()
    tok = MingTok(tcfg, device="cuda", seed=13)
    tsd = {k: v.float().cpu() for k, v in tok.sd.items()}
    gen = torch.Generator().manual_seed(1234)
    imgs = torch.rand(64, 3, 256, 256, generator=gen) * 2 - 1
    n_or = 4
    with torch.no_grad():
        ref = mingtok_ref.mingtok_forward(imgs[:n_or], tsd)
        ref_rec = mingtok_ref.pixel_decoder_forward(ref["x_norm_patchtokens"], tsd).clamp(-1, 1)
    out = tok.forward(imgs.cuda())
    rec = tok.forward_pixel_decoder(out["x_norm_patchtokens"])
    assert rec.shape == (64, 3, 256, 256) and torch.isfinite(rec).all()
    e16 = dict(latent=rel_err(out["latent"][:n_or], ref["latent"]), sem=rel_err(out["x_norm_patchtokens"][:n_or], ref["x_norm_patchtokens"]),
               recon=rel_err(rec[:n_or], ref_rec))
    out32 = tok.forward(imgs[:n_or].cuda(), precision="fp32")
    rec32 = tok.forward_pixel_decoder(out32["x_norm_patchtokens"], precision="fp32")
    e32 = dict(latent=rel_err(out32["latent"], ref["latent"]), sem=rel_err(out32["x_norm_patchtokens"], ref["x_norm_patchtokens"]),
               recon=rel_err(rec32, ref_rec))
    d16 = [abs(psnr(rec[i], imgs[i]) - psnr(ref_rec[i], imgs[i])) for i in range(n_or)]
    d32 = [abs(psnr(rec32[i], imgs[i]) - psnr(ref_rec[i], imgs[i])) for i in range(n_or)]
    print("MingTok 64 x 256^2: bf16 regime", {k: "%.1e" % v for k, v in e16.items()}, "PSNR(hip, oracle) %.1f dB, |dPSNR| max %.4f dB"
          % (psnr(rec[:n_or], ref_rec), max(d16)), "| fp32-class regime", {k: "%.1e" % v for k, v in e32.items()},
          "PSNR(hip, oracle) %.1f dB, |dPSNR| max %.5f dB" % (psnr(rec32, ref_rec), max(d32)))
    assert max(e32.values()) < TOL, e32
    # bf16 regime = the reference's own autocast precision (SURVEY.md §7 calibration: the reference under bf16 autocast sits at
    # 6.7e-3 / 1.4e-2 / 1.9e-2 rel-L2 and 45.4-48.3 dB from its fp32 run): features within 3e-2, the image >= 45 dB from the oracle's
    assert e16["latent"] < TOL_BF16 and e16["sem"] < TOL_BF16 and e16["recon"] < 0.1, e16
    assert psnr(rec[:n_or], ref_rec) > 45.0
    assert max(d16) <= 0.1 and max(d32) <= 0.1                      # north_star: reconstruction PSNR within 0.1 dB of the reference
    # every image of the batch is what it is alone (batch composition does not leak between images)
    one = tok.forward_pixel_decoder(tok.forward(imgs[63:64].cuda())["x_norm_patchtokens"])
    assert psnr(one[0], rec[63]) > 45.0


# ------------------------------------------------------------------------------------------------------------------------
# (e) multi-round state: PAST_MODE=KEEP, and EOS inside a speculative chunk
# ------------------------------------------------------------------------------------------------------------------------
def _tiny_model(t_max=96, eos=1):
    g = load_golden("genimg_tiny")
    llm_cfg = dict(g["llm_config"]); llm_cfg["eos_token_id"] = eos
    sd = llm_sd(g["llm_config"], g["rf_config"], g["seed"])
    tsd = mingtok_sd(g["mingtok_config"], g["seed"])
    lsd = synth_state_dict(C.linear_proj_param_shapes(128, 256, 2), g["seed"])
    return g, llm_cfg, _facade(llm_cfg, g["rf_config"], g["mingtok_config"], g["seed"], t_max, 128, sd, tsd, lsd)


def test_past_mode_keep_and_drop_masks(tmp_path, monkeypatch):
    """modeling_bailingmm.py:273-299: after a round, KEEP carries the round's own uncond / text-uncond masks forward (holes
    included), DROP replaces both by the cond mask; the uncond mask is padded with ZEROS over the generated tokens in both.  A
    second round must see exactly those masks: its image differs between KEEP and DROP, and the KEEP state round-trips."""
    g, llm_cfg, model = _tiny_model()
    ids = g["ids"]
    T = ids.shape[1]
    n_tok = llm_cfg["num_image_tokens_for_gen"]
    am = torch.ones(1, T, dtype=torch.long)
    unc, tunc = g["uncond"][:, :-1], g["rows3_tuncond"][:, :-1]
    state = {}
    for mode in ("KEEP", "DROP"):
        monkeypatch.setenv("PAST_MODE", mode)
        model.reset_inner_state()
        model.noise_generator.manual_seed(7)
        model.generate(input_ids=ids, attention_mask=am, uncond_attention_mask=unc, text_uncond_attention_mask=tunc, max_new_tokens=3,
                       forced_first_token=llm_cfg["image_start_token"], output_image_prefix=str(tmp_path / ("r1" + mode)))
        L = model.past_len
        assert L == T + 1 + n_tok + 1                                    # prompt, <image>, the image tokens, one fed text token
        pad1, pad0 = torch.ones(1, L - T, dtype=torch.long), torch.zeros(1, L - T, dtype=torch.long)
        assert torch.equal(model.past_attention_mask, torch.cat((am, pad1), 1))
        if mode == "KEEP":
            assert torch.equal(model.past_text_uncond_attention_mask, torch.cat((tunc, pad1), 1))
            assert torch.equal(model.past_uncond_attention_mask, torch.cat((unc, pad0), 1))
        else:
            assert torch.equal(model.past_text_uncond_attention_mask, torch.cat((am, pad1), 1))
            assert torch.equal(model.past_uncond_attention_mask, torch.cat((am, pad0), 1))
        # round 2 on top of that state: 5 new prompt tokens, another image
        ids2 = ids[:, :5]
        am2 = torch.ones(1, 5, dtype=torch.long)
        un2 = am2.clone(); un2[0, 1:4] = 0
        model.generate(input_ids=ids2, attention_mask=am2, uncond_attention_mask=un2, text_uncond_attention_mask=am2.clone(), max_new_tokens=2,
                       forced_first_token=llm_cfg["image_start_token"], output_image_prefix=str(tmp_path / ("r2" + mode)))
        assert model.past_len == L + 5 + 1 + n_tok                       # the second new token is returned but never fed
        assert model.past_attention_mask.shape[1] == model.past_len
        if mode == "KEEP":       # round 1's holes are still there, followed by round 2's
            assert torch.equal(model.past_uncond_attention_mask[:, :T], unc)
            assert torch.equal(model.past_uncond_attention_mask[:, L:L + 5], un2)
        state[mode] = model.last_image.clone()
    # the second image was conditioned on different uncond rows: the two policies are really different computations
    assert float((state["KEEP"] - state["DROP"]).abs().max()) > 1e-4


@pytest.mark.parametrize("mode", ["KEEP", "DROP"])
def test_multiround_conversation_vs_the_references_own_generate(mode, tmp_path, monkeypatch):
    """SURVEY §8 a24 against the REFERENCE ITSELF (tests/golden/multiround_tiny.npz: MingUniVisionForConditionalGeneration.generate,
    modeling_bailingmm.py:206-301 + modeling_bailing_moe.py:1769-1796, 1968-2080, run by oracle/gen_golden.gen_multiround): image +
    instruction -> image (3 CFG rows), text -> image on the carried cache and masks, text -> text, under PAST_MODE KEEP and DROP.
    The façade on the HIP path must return the reference's greedy tokens, carry the reference's three masks and cache length into
    every next round, and produce its images (all rows the reference saves: row 0) — the second image depends on every K / V line and
    every mask bit the first round left behind."""
    g = load_golden("multiround_tiny")
    llm_cfg = dict(g["llm_config"])
    sd = llm_sd(g["llm_config"], g["rf_config"], g["seed"])
    tsd = mingtok_sd(g["mingtok_config"], g["seed"])
    lsd = synth_state_dict(C.linear_proj_param_shapes(128, 256, 2), g["seed"])
    model = _facade(llm_cfg, g["rf_config"], g["mingtok_config"], g["seed"], 96, 128, sd, tsd, lsd)
    monkeypatch.setenv("PAST_MODE", mode)
    n_tok = llm_cfg["num_image_tokens_for_gen"]
    for r, spec in enumerate(g["rounds"]):
        t = f"{mode}_r{r}_"
        ids, n0 = g[t + "ids"], g[t + "noise0"]
        seq = model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), uncond_attention_mask=g[t + "unc"],
                             text_uncond_attention_mask=g[t + "tunc"], pixel_values=g["pixel_values"] if spec["px"] else None,
                             max_new_tokens=spec["n_new"], forced_first_token=llm_cfg["image_start_token"] if spec["force"] else None,
                             image_noises=g["noises"][n0:n0 + n_tok + 1], output_image_prefix=str(tmp_path / f"{mode}{r}"), eos_token_id=None)
        assert seq.cpu().tolist() == g[t + "seq"].tolist(), (r, seq.tolist(), g[t + "seq"].tolist())
        assert model.past_len == g[t + "cache_len"]
        assert torch.equal(model.past_attention_mask, g[t + "past_am"])
        assert torch.equal(model.past_uncond_attention_mask, g[t + "past_unc"])
        assert torch.equal(model.past_text_uncond_attention_mask, g[t + "past_tunc"])
        if spec["force"]:
            ref_img = g[t + "image"]
            assert model.last_generation["last_hidden"].shape[0] == ref_img.shape[0]          # the same number of CFG rows
            p = psnr(model.last_image[0], ref_img[0])
            print(f"multi-round {mode} round {r}: image PSNR vs the reference's {p:.1f} dB")
            assert p > 40.0                                              # generate_image's pixel decoder runs in the bf16 regime
    # layer 0's K cache of the whole conversation (the conditional sequence = cache sequence 0)
    k0 = model.model.kv_cache[0, 0, 0, :, :model.past_len].cpu()
    e = rel_err(k0, g[mode + "_k0"][0])
    print(f"multi-round {mode}: layer-0 K lines of {model.past_len} slots vs the reference's: {e:.2e}")
    assert e < TOL


def test_generate_eos_in_the_middle_of_a_chunk_and_arena_end():
    """Greedy decoding runs in speculative chunks of 8 tokens.  An EOS at position 2 of a chunk must leave exactly the state the
    token-by-token loop leaves (past_len, masks, the next round's tokens); a conversation that ends a few slots short of t_max
    must never be fed past the arena, and a full arena raises instead of corrupting the cache."""
    g, llm_cfg, model = _tiny_model(t_max=40, eos=-1)
    ids = g["ids"]
    T = ids.shape[1]
    model.decode_chunk = 1
    free_run = model.generate(input_ids=ids, max_new_tokens=7)[0, T:].tolist()
    j = next((i for i in range(1, 6) if free_run[i] not in free_run[:i]), None)
    assert j is not None, free_run
    eos = free_run[j]
    model.config.llm_config.eos_token_id = eos
    results = {}
    for chunk in (1, 8):
        model.decode_chunk = chunk
        model.reset_inner_state()
        r1 = model.generate(input_ids=ids, max_new_tokens=7)[0, T:].tolist()
        assert r1 == free_run[:j + 1]
        assert model.past_len == T + j                                   # the EOS itself is never fed
        assert model.past_attention_mask.shape[1] == T + j
        r2 = model.generate(input_ids=ids[:, :4], max_new_tokens=3)[0, 4:].tolist()
        results[chunk] = (r1, r2, model.past_len)
    assert results[1] == results[8], results
    # arena end: t_max = 40; prompt 12 -> at most 28 generated slots
    model.config.llm_config.eos_token_id = -1
    model.decode_chunk = 8
    model.reset_inner_state()
    sentinel = model.model.kv_cache[:, 1].clone()                        # sequence 1 follows sequence 0 in the arena
    out = model.generate(input_ids=ids, max_new_tokens=40 - T + 1)       # one more token than slots: the last is never fed
    assert out.shape[1] == 40 + 1 and model.past_len == 40
    assert torch.equal(model.model.kv_cache[:, 1], sentinel)             # nothing was written past sequence 0's slots
    with pytest.raises(ValueError, match="KV arena"):
        model.generate(input_ids=ids[:, :2], max_new_tokens=2)
