"""Full-WIDTH parity: the 16B-A3B layer shapes (H = 2048, 64 experts top-6 + 2 shared, I = 1408), the full RF head
(w = 3072, depth 12, hidden 8192, 16 Euler steps) and the full MingTok semantic decoder (D = 1024, 24 layers), with the
LLM cut to 2 layers and 3 visual tokens so that the fp32 CPU oracle finishes in seconds.  Covers what the tiny-config
fixtures cannot: the production launch plans, tile counts and K ranges of every streaming kernel, end to end."""
import pytest
import torch

from ming_univision_amd import configuration as C
from ming_univision_amd.synth import synth_state_dict
from tests.util import llm_sd, mingtok_sd, rel_err, rel_err_rows

pytestmark = pytest.mark.gpu

TOL = 1e-3
ROW_TOL = 3e-3       # the worst single ROW (its own max-norm): `rel_err`'s global max-norm alone would let a small-magnitude row be far off


def _dev(sd):
    return {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}


@pytest.fixture(scope="module")
def full():
    from oracle import bailing_ref
    seed = 5
    d = C.BailingMoeConfig.ming_univision_16b_a3b().to_dict()
    d.pop("model_type", None)
    d.update(num_hidden_layers=2, vocab_size=1024, num_image_tokens_for_gen=3, image_start_token=1000, pad_token_id=0)
    rf_cfg = dict(C.DEFAULT_VISHEAD_DIFFLOSS)
    tcfg = C.MingTokConfig()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    sd = llm_sd(d, rf_cfg, seed)                                   # fp32 values of bf16-rounded weights (CPU)
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in d.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    return d, rf_cfg, tcfg, sd, ocfg, seed


@pytest.mark.parametrize("rows_tag", ["rows2", "rows3"])
def test_full_width_generate_image_vs_oracle(full, rows_tag):
    from oracle import bailing_ref, mingtok_ref
    from ming_univision_amd.bailing_moe import BailingMoeDecoder, generate_image, generate_images
    from ming_univision_amd.mingtok import MingTok
    from ming_univision_amd.rf_head import RectifiedFlowHead
    d, rf_cfg, tcfg, sd, ocfg, seed = full
    cfg = C.BailingMoeConfig(**d)
    dsd = _dev(sd)
    B = 32 if rows_tag == "rows2" else 21                           # batched run below: 64 / 63 rows in one lock-step group
    dec = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=32, n_seq=3 * B)
    rf = RectifiedFlowHead(dsd, cfg.hidden_size, rf_cfg)
    lsd = synth_state_dict(C.linear_proj_param_shapes(1024, cfg.hidden_size, 2), seed)
    dl = _dev(lsd)
    tok = MingTok(tcfg, device="cuda", seed=seed,
                  linear_proj=[(dl["linear_proj.0.weight"], dl["linear_proj.0.bias"]),
                               (dl["linear_proj.2.weight"], dl["linear_proj.2.bias"])])
    tsd = {k: v.float().cpu() for k, v in tok.sd.items()}           # the oracle sees the same (bf16-rounded) MingTok weights
    g = torch.Generator().manual_seed(1)
    T = 12
    ids = torch.randint(0, 900, (1, T), generator=g)
    noises = torch.randn(cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    am = torch.ones(1, T + 1, dtype=torch.long)
    un = am.clone(); un[0, 2:T - 2] = 0
    tu = am.clone(); tu[0, 2:5] = 0
    if rows_tag == "rows2":
        tu = un.clone()
    # oracle
    kvs = bailing_ref.new_kv(ocfg)
    emb = sd["model.word_embeddings.weight"][ids]
    bailing_ref.model_forward(emb, sd, ocfg, torch.ones(1, T, dtype=torch.long), None, kvs)
    start = sd["model.word_embeddings.weight"][torch.tensor([[cfg.image_start_token]])]
    caches = mingtok_ref.semdec_new_cache(tsd)
    ref = bailing_ref.generate_image(
        start, kvs, am, un, tu, sd, ocfg, noises,
        latent_to_sem=lambda lat: mingtok_ref.mingtok_feature_decoder_step(lat, tsd, caches),
        linear_proj=lambda s: bailing_ref.linear_proj(s, lsd), sem_to_pix=lambda s: None,
        steps=int(rf_cfg["num_sampling_steps"]))
    # HIP path
    dec.prefill(dec.embed(ids[0].cuda()), seq=0, past=0)
    out = generate_image(dec, rf, tok, dec.embed(torch.tensor([cfg.image_start_token]).cuda()), T, am, un, tu,
                         noises.cuda(), decode_pixels=False)
    rows = ref["last_hidden"].shape[0]
    assert rows == (2 if rows_tag == "rows2" else 3)
    assert rel_err(out["latents"], ref["latents"][:, 0]) < TOL
    assert rel_err(out["sem"], ref["sem"][0]) < TOL
    assert rel_err(out["last_hidden"], ref["last_hidden"][:, 0]) < TOL
    rows_b1 = (rel_err_rows(out["latents"], ref["latents"][:, 0]), rel_err_rows(out["sem"], ref["sem"][0]),
               rel_err_rows(out["last_hidden"], ref["last_hidden"][:, 0]))
    print("batch 1 (%s), worst single row (latents / sem / hidden): %.2e / %.2e / %.2e" % ((rows_tag,) + rows_b1))
    assert max(rows_b1) < ROW_TOL, rows_b1
    # the same image inside a full lock-step group (K-loop form with four row tiles, grouped experts with every expert
    # active, unfused decoder sequence above 32 rows), other images with their own noise: image 0 must still match the oracle
    R = rows
    for i in range(B):
        dec.prefill(dec.embed(ids[0].cuda()), seq=i * R, past=0)
    nb = torch.randn(B, cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    nb[0] = noises
    outb = generate_images(dec, rf, tok, dec.embed(torch.tensor([cfg.image_start_token]).cuda()), [T] * B, [am] * B, [un] * B,
                           [tu] * B, nb.cuda(), decode_pixels=False, n_groups=1)
    assert rel_err(outb["latents"][0], ref["latents"][:, 0]) < TOL
    assert rel_err(outb["last_hidden"][:R], ref["last_hidden"][:, 0]) < TOL
    # the same images on the WIDE route (> 64 rows: every Linear a gemm256 launch on hi/lo operands, grouped-GEMM experts,
    # one lock-step group): image 0 against the oracle, the images both runs share against the <= 64-row route
    Bw = 3 * B // 2                                               # 96 / 93 rows (rows2 / rows3)
    decw = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=32, n_seq=R * Bw)
    for i in range(Bw):
        decw.prefill(decw.embed(ids[0].cuda()), seq=i * R, past=0)
    nw = torch.cat([nb, torch.randn(Bw - B, cfg.num_image_tokens_for_gen + 1, 32, generator=g)])
    outw = generate_images(decw, rf, tok, decw.embed(torch.tensor([cfg.image_start_token]).cuda()), [T] * Bw, [am] * Bw, [un] * Bw,
                           [tu] * Bw, nw.cuda(), decode_pixels=False, n_groups=1)
    assert torch.isfinite(outw["latents"]).all()
    print("wide route vs oracle:", rows_tag, "latents %.2e sem %.2e hidden %.2e" % (
        rel_err(outw["latents"][0], ref["latents"][:, 0]), rel_err(outw["sem"][0], ref["sem"][0]),
        rel_err(outw["last_hidden"][:R], ref["last_hidden"][:, 0])), "| narrow: latents %.2e hidden %.2e" % (
        rel_err(outb["latents"][0], ref["latents"][:, 0]), rel_err(outb["last_hidden"][:R], ref["last_hidden"][:, 0])))
    assert rel_err(outw["latents"][0], ref["latents"][:, 0]) < TOL
    assert rel_err(outw["sem"][0], ref["sem"][0]) < TOL
    assert rel_err(outw["last_hidden"][:R], ref["last_hidden"][:, 0]) < TOL
    rows_w = (rel_err_rows(outw["latents"][0], ref["latents"][:, 0]), rel_err_rows(outw["sem"][0], ref["sem"][0]),
              rel_err_rows(outw["last_hidden"][:R], ref["last_hidden"][:, 0]),
              rel_err_rows(outb["latents"][0], ref["latents"][:, 0]), rel_err_rows(outb["last_hidden"][:R], ref["last_hidden"][:, 0]))
    print("worst single row — wide: latents %.2e sem %.2e hidden %.2e | <= 64-row route: latents %.2e hidden %.2e" % rows_w)
    assert max(rows_w) < ROW_TOL, rows_w
    per_img = torch.stack([(outw["latents"][i] - outb["latents"][i]).abs().max() / outb["latents"][i].abs().max() for i in range(B)])
    # two HIP routes with different fp32 summation orders: they agree to ~2e-4 image by image, except where a 2^-17 difference
    # flips a near-tie of the (random-init, N(0, 0.006)) router, after which that image is a different sample (top-k is
    # discontinuous; the oracle comparison above is the parity gate) - at most one image in 16 may do that
    n_flip = int((per_img >= 5e-3).sum())
    print("wide vs <= 64-row route, per image: median %.2e, %d of %d above 5e-3" % (float(per_img.median()), n_flip, B))
    assert float(per_img.median()) < 5e-4 and n_flip <= max(1, B // 16), (float(per_img.median()), per_img.tolist())


def test_full_width_facade_batch_on_wide_route(tmp_path):
    """MingUniVisionForConditionalGeneration.generate_image_batch with 40 ragged requests at the production width (80 CFG rows:
    wide route, lm_head through gemm256, KV arena grown on demand): images 0 and 39 equal what generate() produces for those
    requests alone with the same noise."""
    from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration
    d = C.BailingMoeConfig.ming_univision_16b_a3b().to_dict()
    d.pop("model_type", None)
    d.update(num_hidden_layers=2, vocab_size=1024, num_image_tokens_for_gen=4, image_start_token=1000, pad_token_id=0, eos_token_id=1)
    cfg = C.MingUniVisionConfig(mlp_depth=2, llm_config=d, vishead_diffloss_config=dict(C.DEFAULT_VISHEAD_DIFFLOSS))
    model = MingUniVisionForConditionalGeneration(cfg, state_dict=None, seed=9, t_max=48)
    g = torch.Generator().manual_seed(2)
    B = 40
    reqs = []
    for b in range(B):
        n = 10 + (b % 5)
        unc = torch.ones(1, n, dtype=torch.long); unc[0, 2:n - 2] = 0
        reqs.append(dict(input_ids=torch.randint(2, 900, (1, n), generator=g), attention_mask=torch.ones(1, n, dtype=torch.long),
                         uncond_attention_mask=unc, text_uncond_attention_mask=unc.clone()))
    noises = torch.randn(B, 5, 32, generator=g)
    out = model.generate_image_batch(reqs, forced_first_token=1000, noises=noises, save=False)
    assert out["images"].shape == (B, 3, 64, 64) and torch.isfinite(out["images"]).all()
    for b in (0, B - 1):
        model.reset_inner_state()
        one = model.generate_image_batch([reqs[b]], forced_first_token=1000, noises=noises[b:b + 1], save=False)
        assert rel_err(out["latents"][b], one["latents"][0]) < 5e-3      # two HIP routes (80 rows wide vs 2 rows), chaotic random model
        assert rel_err(out["images"][b], one["images"][0]) < 5e-2


def test_full_width_text_batch_on_wide_route():
    """generate_text_batch with 70 conversations at the production width (decode on the wide route: grouped-GEMM experts,
    lm_head through gemm256; prompts of 6..11 tokens through the decode kernels, two of 80 / 97 tokens stacked on the bf16 MFMA
    prefill path): conversations 0, 35, 68 and 69 get the greedy tokens generate() gives them alone."""
    from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration
    d = C.BailingMoeConfig.ming_univision_16b_a3b().to_dict()
    d.pop("model_type", None)
    d.update(num_hidden_layers=2, vocab_size=1024, num_image_tokens_for_gen=4, image_start_token=1000, pad_token_id=0, eos_token_id=1)
    cfg = C.MingUniVisionConfig(mlp_depth=2, llm_config=d, vishead_diffloss_config=dict(C.DEFAULT_VISHEAD_DIFFLOSS))
    model = MingUniVisionForConditionalGeneration(cfg, state_dict=None, seed=11, t_max=128)
    g = torch.Generator().manual_seed(4)
    lens = [6 + (b % 6) for b in range(68)] + [80, 97]
    reqs = [dict(input_ids=torch.randint(2, 900, (1, n), generator=g)) for n in lens]
    tm = {}
    batch = model.generate_text_batch(reqs, max_new_tokens=6, sync_every=4, timings=tm)
    assert len(batch) == 70 and tm["steps"] <= 5
    for b in (0, 35, 68, 69):
        model.reset_inner_state()
        seq = model.generate(input_ids=reqs[b]["input_ids"], max_new_tokens=6)[0, lens[b]:].tolist()
        if 1000 in seq:                   # generate() starts an image there; the text batch returns the token
            seq = seq[:seq.index(1000) + 1]
        assert seq == batch[b][:len(seq)], (b, seq, batch[b])


@pytest.mark.parametrize("n_images,rpi", [(33, 2), (22, 3), (65, 2), (129, 2), (200, 2)])
def test_wide_rf_sampler_matches_narrow_route(n_images, rpi):
    """RectifiedFlowLoss.sample at production width on the wide route (66 / 66 / 130 / 258 / 400 rows: partial row tiles, 2 and 3 CFG
    rows; below 385 rows w12 runs split-K + slab SwiGLU, at 400 rows with the SwiGLU epilogue) against the <= 64-row route of the same
    library, image by image (both are fp32-class: 2^-17 operands)."""
    from ming_univision_amd.rf_head import RectifiedFlowHead
    from ming_univision_amd.synth import synth_tensor
    cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
    rf_cfg = dict(C.DEFAULT_VISHEAD_DIFFLOSS)
    shapes = C.llm_param_shapes(cfg, rf_cfg, 32)
    sd = {k: synth_tensor(k, s_, 7, "cuda", torch.bfloat16) for k, s_ in shapes.items() if k.startswith("vis_head") or k.startswith("diffloss")}
    rf = RectifiedFlowHead(sd, cfg.hidden_size, rf_cfg)
    g = torch.Generator().manual_seed(n_images)
    hidden = torch.randn(n_images * rpi, cfg.hidden_size, generator=g).cuda()
    noise = torch.randn(n_images, 32, generator=g).cuda()
    wide = rf.sample(hidden, noise, n_images=n_images)
    per = 64 // rpi
    nar = torch.cat([rf.sample(hidden[i * rpi:(i + per) * rpi].contiguous(), noise[i:i + per].contiguous(),
                               n_images=min(per, n_images - i)) for i in range(0, n_images, per)])
    assert torch.isfinite(wide).all()
    err = (wide - nar).abs().amax(dim=1) / nar.abs().amax(dim=1)
    assert float(err.max()) < 2e-4, float(err.max())


def test_prefill_mfma_many_matches_one_sequence_at_a_time():
    """bf16 MFMA prefill of three prompts of different lengths stacked into one row block (production layer shapes, 2 layers,
    one prompt with image-token rows): last-token hidden states and KV entries against the same prompts prefilled one by one
    (same kernels; only the GEMM row-block composition differs)."""
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
    cfg.num_hidden_layers = 2
    dec = BailingMoeDecoder.synthetic(cfg, torch.device("cuda"), seed=6, with_vocab=False, t_max=160, n_seq=6)
    g = torch.Generator().manual_seed(1)
    lens = [70, 131, 100]
    embeds = [torch.randn(n, cfg.hidden_size, generator=g).cuda() * 0.3 for n in lens]
    masks = [None, (torch.arange(131) % 3 == 0).cuda(), None]
    one = torch.cat([dec.prefill_mfma(e, seq=i, past=0, image_mask=m) for i, (e, m) in enumerate(zip(embeds, masks))])
    many = dec.prefill_mfma_many(embeds, [3, 4, 5], past=0, image_masks=masks)
    assert torch.isfinite(many).all()
    assert rel_err(many, one) < 2e-3, rel_err(many, one)
    for i, n in enumerate(lens):
        a, b = dec.kv_cache[:, i, ..., :n, :], dec.kv_cache[:, 3 + i, ..., :n, :]
        assert rel_err(b, a) < 2e-3


def test_wide_llm_step_matches_narrow_route_ragged_rows():
    """One decoder-stack step at production layer shapes for 150 rows with ragged cache lengths, holey key masks and shared
    embeddings (x_row_div = 2): wide route against the same rows pushed through the <= 64-row route in three calls."""
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
    cfg.num_hidden_layers = 2
    dec = BailingMoeDecoder.synthetic(cfg, torch.device("cuda"), seed=4, with_vocab=False, t_max=48, n_seq=150)
    g = torch.Generator().manual_seed(0)
    M = 150
    dec.kv_cache.copy_(torch.randn(dec.kv_cache.shape, generator=g).cuda() * 0.5)      # a populated cache
    kv0 = dec.kv_cache.clone()
    x = torch.randn(M // 2, cfg.hidden_size, generator=g).cuda()
    slot = torch.randint(5, 40, (M,), generator=g).to(torch.int32).cuda()
    seq = torch.arange(M, dtype=torch.int32).cuda()
    km = (torch.rand(M, 48, generator=g) > 0.2).to(torch.uint8)
    km[torch.arange(M), slot.cpu().long()] = 1                                         # the current token is always attended
    km = km.cuda()
    wide = dec.step(x, seq, slot, slot, slot + 1, km, None, rows=M, x_row_div=2)
    kv_w = dec.kv_cache.clone()
    dec.kv_cache.copy_(kv0)
    xr = x.repeat_interleave(2, dim=0)
    nar = torch.cat([dec.step(xr[i:i + 50].contiguous(), seq[i:i + 50].contiguous(), slot[i:i + 50].contiguous(),
                              slot[i:i + 50].contiguous(), (slot[i:i + 50] + 1).contiguous(), km[i:i + 50].contiguous(), None)
                     for i in range(0, M, 50)])
    assert rel_err(wide, nar) < 2e-4
    assert rel_err(kv_w, dec.kv_cache) < 1e-5                                           # same K / V rows appended


def test_span_prefill_across_pass_boundaries_vs_oracle_and_row_kernels(full):
    """mn_llm_step_spans (tiled hi/lo flash attention on the wide route): three prompts of different lengths prefilled (a) in ONE pass,
    (b) in 128-row passes that cut the sequences (spans that continue with past > 0, a last pass of <= 64 rows on the row kernels),
    (c) with a non-zero `past` (second round of a conversation) — against the fp32 oracle (last hidden state of every prompt), and the
    K / V lines they write against the 64-row decode-kernel prefill (the path without any tiling)."""
    from oracle import bailing_ref
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    d, rf_cfg, tcfg, sd, ocfg, seed = full
    cfg = C.BailingMoeConfig(**d)
    dec = BailingMoeDecoder.from_state_dict(cfg, _dev(sd), t_max=320, n_seq=6)
    assert dec.max_rows() == 2048
    g = torch.Generator().manual_seed(11)
    lens = [150, 90, 200]
    ids = [torch.randint(0, 900, (n,), generator=g) for n in lens]
    embs = [dec.embed(i.cuda()) for i in ids]
    ref_last, ref_mid = [], []
    for i in ids:                                                   # oracle: each prompt alone, then 40 more tokens on its cache
        kvs = bailing_ref.new_kv(ocfg)
        h = bailing_ref.model_forward(sd["model.word_embeddings.weight"][i[None]], sd, ocfg, torch.ones(1, len(i), dtype=torch.long), None, kvs)
        ref_last.append(h[0, -1:])
    h1 = dec.prefill_ragged(embs, [0, 1, 2], past=0)                # (a) one 440-row pass, three spans
    for b in range(3):
        assert rel_err(h1[b:b + 1], ref_last[b]) < TOL, (b, rel_err(h1[b:b + 1], ref_last[b]))
    kv_a = dec.kv_cache[:, :3].clone()
    real = dec.max_rows
    try:
        dec.max_rows = lambda: 128                                  # (b) passes of 128 rows: 128 + 128 + 128 + 56
        h2 = dec.prefill_ragged(embs, [3, 4, 5], past=0)
    finally:
        dec.max_rows = real
    assert rel_err(h2, h1) < 1e-4, rel_err(h2, h1)                  # (other tile / split-K compositions, the last pass on the row kernels)
    for b, n in enumerate(lens):
        assert rel_err(dec.kv_cache[:, 3 + b, :, :, :n], kv_a[:, b, :, :, :n]) < 1e-4
    # the row kernels (no tiling anywhere): same K / V lines and hidden states to summation order
    dec2 = BailingMoeDecoder.from_state_dict(cfg, _dev(sd), t_max=320, n_seq=1)
    hr = dec2.prefill(embs[2], seq=0, past=0)
    assert rel_err(hr[-1:], h1[2:3]) < 2e-4
    assert rel_err(dec2.kv_cache[:, 0, :, :, :200], kv_a[:, 2, :, :, :200]) < 2e-4
    # (c) a second round on sequence 2: 100 more tokens from past = 200 through prefill_wide (one span with past > 0)
    more = torch.randint(0, 900, (100,), generator=g)
    h3 = dec.prefill_wide(dec.embed(more.cuda()), seq=2, past=200)
    kvs = bailing_ref.new_kv(ocfg)
    both = torch.cat((ids[2], more))
    ho = bailing_ref.model_forward(sd["model.word_embeddings.weight"][both[None]], sd, ocfg, torch.ones(1, 300, dtype=torch.long), None, kvs)
    assert rel_err(h3[-1:], ho[0, -1:]) < TOL, rel_err(h3[-1:], ho[0, -1:])
