"""fp8 weight mode (mingnative.h section 7; BASELINE configs[4] "fp8") on the GPU.

Parity definition (written here, as SURVEY.md §2.2 asks: "a separately stated, looser tolerance" applies to the MODEL, not to
the kernels): the fp8 model is the bf16 model with the RF ResBlock matrices and the decoder stack's experts replaced by
e4m3(W / s) * s, s one power-of-two scale per output row (oracle/fp8_ref.py).  The HIP path streams the e4m3 bytes and is held to
the fp32 oracle FED THOSE DEQUANTISED WEIGHTS at the same 1e-3 as the bf16 path (kernel parity, TOL).  How far the quantised
model's outputs are from the bf16 model's — the price of the format, not of the implementation — is reported, not gated: on the
random-init full-width model (a chaotic map: a discontinuous router, 28 x 16 residual updates) an e4m3 weight perturbation
of 2^-4 relative moves the sampled latents by O(1) (measured 0.75 / 1.0 / 0.8 at latents / sem / hidden) — the fp8 model is a
DIFFERENT model of the same architecture, which is why parity is defined on its own weights.  On a trained checkpoint the
usual weight-only-quantisation picture applies (not measurable here: no weights).  Reference surface this mode stands in for:
the weight-only `dtype` switch of mingunivisioninfer.py:46-70."""
import numpy as np
import pytest
import torch

from ming_univision_amd import configuration as C
from ming_univision_amd.synth import synth_state_dict
from tests.util import llm_sd, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-3          # HIP fp8 path vs the oracle on the dequantised weights


def _dev(sd):
    return {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}


def test_e4m3_to_bf16_conversion_of_every_byte_in_both_kernels():
    """v_cvt_scalef32_pk_bf16_fp8 (streaming kernels) and v_cvt_pk_f32_fp8 (dequantiser) decode all 254 finite e4m3 bytes to the
    OCP values: W[n, :] = byte n everywhere, x = one 1.0 per row -> out[m, n] = value(n) * scale[n]."""
    from oracle import fp8_ref
    from ming_univision_amd import ops
    tab = torch.from_numpy(fp8_ref.decode_e4m3_table()).float()
    bytes_ = torch.arange(256, dtype=torch.uint8)
    bytes_[0x7F] = 0; bytes_[0xFF] = 0x80                                        # the two NaN codes never occur in a weight
    want = tab[bytes_.long()]
    for K, M in ((16, 16), (256, 3), (32, 20), (128, 40)):                               # K-slice kernel (both ring paths) and K-loop kernel
        q = bytes_.unsqueeze(1).repeat(1, K).contiguous().cuda()
        scale = torch.ones(256, device="cuda")
        scale[1::2] = 0.5
        x = torch.zeros(M, K)
        for m in range(M):
            x[m, (7 * m + 3) % K] = 1.0
        y2 = torch.stack([x.to(torch.bfloat16), torch.zeros(M, K, dtype=torch.bfloat16)]).cuda().contiguous()
        out = ops.stream_mfma_w8(y2, q, scale).cpu()
        assert torch.equal(out, (want * scale.cpu()).unsqueeze(0).expand(M, 256)), (K, M)
    dq = ops.dequant_fp8_rows(bytes_.unsqueeze(1).repeat(1, 16).contiguous().cuda(), torch.ones(256, device="cuda")).float().cpu()
    assert torch.equal(dq, want.unsqueeze(1).expand(256, 16))


def test_quantiser_is_bit_identical_to_the_oracle():
    from oracle import fp8_ref
    from ming_univision_amd import ops
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(300, 1408, generator=g) * torch.logspace(-7, 2, 300).unsqueeze(1)).to(torch.bfloat16)
    w[3] = 0
    w[5, :] = 0; w[5, 7] = 448.0
    w[6, :] = 0; w[6, 9] = 1.75 * 2.0 ** -20
    w[7, :] = 0; w[7, 9] = 1.7578125 * 2.0 ** 5
    tab = fp8_ref.decode_e4m3_table()[:127]
    mids = torch.from_numpy((tab[:-1] + tab[1:]) / 2).float()                    # round-to-nearest-even ties (bf16-exact: 5 bits)
    w[8, :] = 0; w[8, :126] = mids.to(torch.bfloat16); w[8, 200] = 448.0
    q, s = ops.quant_fp8_rows(w.cuda().contiguous())
    qo, so = fp8_ref.quantize_rows(w)
    assert torch.equal(s.cpu(), so)
    same = q.cpu() == qo
    zero = (w.float() / so.unsqueeze(1)) == 0                                    # +-0 may differ in sign only
    assert bool((same | zero).all()), int((~(same | zero)).sum())
    dq = ops.dequant_fp8_rows(q, s)
    assert torch.equal(dq.float().cpu(), fp8_ref.dequantize_rows(q.cpu(), s.cpu()))
    # 3-D (packed experts) form
    w3 = torch.randn(5, 64, 96, generator=g).to(torch.bfloat16)
    q3, s3 = ops.quant_fp8_rows(w3.cuda().contiguous())
    qo3, so3 = fp8_ref.quantize_rows(w3)
    assert s3.shape == (5, 64) and torch.equal(s3.cpu(), so3) and torch.equal(q3.cpu(), qo3)


@pytest.mark.parametrize("M", [1, 2, 3, 16, 17, 32, 33, 48, 64])
def test_stream_mfma_w8_against_float64(M):
    """Dense fp8 launches of every kernel form (K-slice <= 32 rows with one / two row tiles, K-loop above) at the RF head's shapes
    and at ragged ones (N not a multiple of 16, K not a multiple of 256 / 128) against the float64 product of the SAME operands."""
    from ming_univision_amd import ops
    g = torch.Generator().manual_seed(100 + M)
    shapes = [(2 * 8192, 3072), (3072, 8192), (1000, 1408), (40, 16), (2816, 2048), (330, 464), (72, 176), (50, 1232)]
    for N, K in shapes:
        w = (torch.randn(N, K, generator=g) * K ** -0.5 * torch.logspace(-1, 1, N).unsqueeze(1)).to(torch.bfloat16).cuda()
        q, s = ops.quant_fp8_rows(w)
        x = torch.randn(M, K, generator=g)
        a2 = ops.split_hilo(x.cuda())
        out = ops.stream_mfma_w8(a2.contiguous(), q, s)
        ref = (a2[0].double() + a2[1].double()) @ ops.dequant_fp8_rows(q, s).double().T
        e = rel_err(out, ref)
        assert e < 3e-5, (M, N, K, e)


def test_skinny_gemm_fp8_route_with_fused_prologue_and_epilogue():
    """mn_skinny_gemm with wfmt = fp8: every row count (1 included) takes prologue -> fp8 streaming launch -> epilogue."""
    from ming_univision_amd import ops
    g = torch.Generator().manual_seed(9)
    K, N = 3072, 8192
    w = (torch.randn(2 * N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    b = (torch.randn(2 * N, generator=g) * 0.1).to(torch.bfloat16).cuda()
    q, s = ops.quant_fp8_rows(w)
    wd = ops.dequant_fp8_rows(q, s).double()
    for M in (1, 2, 8, 40):
        x = torch.randn(M, K, generator=g).cuda()
        sh, sc = (torch.randn(M, K, generator=g) * 0.1).cuda(), (torch.randn(M, K, generator=g) * 0.1).cuda()
        y = ops.skinny_gemm(x, q, b, prologue="ln_mod", epilogue="swiglu", eps=1e-6, pro_a=sh, pro_b=sc, wscale=s)
        xn = torch.nn.functional.layer_norm(x.double(), (K,), eps=1e-6) * (1 + sc.double()) + sh.double()
        r = xn @ wd.T + b.double()
        ref = torch.nn.functional.silu(r[:, :N]) * r[:, N:]
        assert rel_err(y, ref) < 2e-5, (M, rel_err(y, ref))


@pytest.mark.parametrize("M", [1, 2, 3, 5])
def test_expert_pair_kernels_on_fp8_weights_against_float64(M):
    """The one-row fp8 kernel (skinny_w8.hip) on the expert launches of a 1- / 2-row step — and of the 3..5-row steps that take the pair
    launches in e4m3 since round 6 (24 / 40 pairs: their workgroups are planned into one round over the CUs) — at the 16B-A3B shapes: (row, expert) pairs
    with the SwiGLU epilogue, then the down projection as 8 K-segments (6 routed + 2 shared) with per-segment row scales, router
    weights and the residual — against the float64 formula on the dequantised weights."""
    from ming_univision_amd import ops
    g = torch.Generator().manual_seed(40 + M)
    E, S, I, H, top = 64, 2, 1408, 2048, 6
    gu = (torch.randn(E + S, 2 * I, H, generator=g) * H ** -0.5).to(torch.bfloat16).cuda()
    dn = (torch.randn(E + S, H, I, generator=g) * I ** -0.5 * torch.logspace(-1, 1, H).reshape(1, H, 1)).to(torch.bfloat16).cuda()
    gq, gs = ops.quant_fp8_rows(gu)
    dq, ds = ops.quant_fp8_rows(dn)
    xn = torch.randn(M, H, generator=g).cuda()
    res = torch.randn(M, H, generator=g).cuda()
    idx = torch.stack([torch.cat((torch.randperm(E, generator=g)[:top], torch.tensor([E, E + 1]))) for _ in range(M)]).to(torch.int32).cuda()
    w = torch.cat((torch.rand(M, top, generator=g), torch.ones(M, S)), 1).cuda()
    out = ops.moe_experts(xn, idx, w, gq, dq, res, gate_up_scale=gs, down_scale=ds)
    gd, dd = ops.dequant_fp8_rows(gq, gs).double(), ops.dequant_fp8_rows(dq, ds).double()
    ref = res.double().clone()
    for m in range(M):
        for s_ in range(top + S):
            e = int(idx[m, s_])
            r = gd[e] @ xn[m].double()
            hmid = torch.nn.functional.silu(r[:I]) * r[I:]
            ref[m] += float(w[m, s_]) * (dd[e] @ hmid.float().double())      # the kernel stores the SwiGLU output as fp32
    assert rel_err(out, ref) < 2e-5, rel_err(out, ref)


@pytest.fixture(scope="module")
def full():
    from oracle import bailing_ref
    seed = 5
    d = C.BailingMoeConfig.ming_univision_16b_a3b().to_dict()
    d.pop("model_type", None)
    d.update(num_hidden_layers=2, vocab_size=1024, num_image_tokens_for_gen=3, image_start_token=1000, pad_token_id=0)
    rf_cfg = dict(C.DEFAULT_VISHEAD_DIFFLOSS)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    sd = llm_sd(d, rf_cfg, seed)
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in d.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    return d, rf_cfg, sd, ocfg, seed


def _fp8_models(full, n_seq, fmt="fp8"):
    """HIP models of the full-width 2-layer configuration in the weight-only mode `fmt` ("fp8" | "int8") + the oracle's state dict
    of that model (dequantised weights).  "int8" (the reference's quanto mode) converts EVERY nn.Linear — attention, lm_head, the small
    RF Linears, MingTok, linear_proj — "fp8" (this library's own format) the streamed tensors only."""
    from ming_univision_amd import _lib, ops
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    from ming_univision_amd.mingtok import MingTok
    from ming_univision_amd.rf_head import RectifiedFlowHead
    d, rf_cfg, sd, ocfg, seed = full
    cfg = C.BailingMoeConfig(**d)
    dsd = _dev(sd)
    dec8 = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=32, n_seq=n_seq, weights=fmt)
    rf8 = RectifiedFlowHead(dsd, cfg.hidden_size, rf_cfg, weights=fmt)
    assert dec8.layers[0]["w_gate_up"].dtype == torch.uint8 and rf8.lists["w12"][0].dtype == torch.uint8
    assert dec8.max_rows() == 2048 and rf8.max_rows() == 2048        # (round 5: the wide route de-quantises into a scratch)
    sd8 = dict(sd)                                                   # the oracle's weights of the quantised model
    for k, v in dec8.dequantized_state_dict().items():
        sd8[k] = v.float().cpu()
    for k, v in rf8.dequantized_blocks().items():
        sd8[k] = v.float().cpu()
    # ... and they are what the oracle's own quantiser produces from the bf16 weights
    from oracle import fp8_ref, int8_ref
    full_model = fmt in _lib.FULL_MODEL
    if fmt == "int8":
        fp8_ref = int8_ref                                            # (same function names)
    keys = ["model.layers.1.mlp.experts.5.up_proj.weight", "model.layers.0.mlp.shared_experts.gate_proj.weight",
            "diffloss.net.res_blocks.3.mlp.w12.weight", "diffloss.net.res_blocks.11.mlp.w3.weight"]
    if full_model:
        keys += ["model.layers.1.attention.query_key_value.weight", "model.layers.0.attention.dense.weight", "lm_head.weight",
                 "vis_head.0.weight", "diffloss.net.cond_embed.weight", "diffloss.net.input_proj.weight", "diffloss.net.time_embed.mlp.2.weight",
                 "diffloss.net.final_layer.linear.weight", "diffloss.net.res_blocks.7.adaLN_modulation.1.weight"]
    for k in keys:
        assert torch.equal(sd8[k], fp8_ref.fake_quant_rows(sd[k])), k
    assert torch.equal(sd8["model.layers.0.mlp.gate.weight"], sd["model.layers.0.mlp.gate.weight"])        # the router is not an nn.Linear
    k = "model.layers.0.mlp.shared_experts.down_proj.weight"
    I = cfg.moe_intermediate_size
    if fmt == "int8":    # quanto: one scale per output row of the WHOLE Linear — the two pseudo-experts of the packed layout share it
        assert torch.equal(sd8[k], int8_ref.fake_quant_rows(sd[k]))
    else:                # e4m3 (own format): quantised per pseudo-expert (column block) in the packed layout
        assert torch.equal(sd8[k], torch.cat([fp8_ref.fake_quant_rows(sd[k][:, s * I:(s + 1) * I].contiguous()) for s in range(2)], 1))
    lsd = synth_state_dict(C.linear_proj_param_shapes(1024, cfg.hidden_size, 2), seed)
    dl = ops.convert_linears(_dev(lsd), fmt)
    lsd = {k_: v.float().cpu() for k_, v in dl.items()}
    tok = MingTok(C.MingTokConfig(), device="cuda", seed=seed, weights=fmt if full_model else "bf16",
                  linear_proj=[(dl["linear_proj.0.weight"], dl["linear_proj.0.bias"]), (dl["linear_proj.2.weight"], dl["linear_proj.2.bias"])])
    return cfg, dsd, dec8, rf8, sd8, lsd, tok


@pytest.mark.parametrize("rows_tag", ["rows2", "rows3"])
def test_fp8_full_width_generate_image_vs_oracle_on_dequantised_weights(full, rows_tag):
    """Full width (16B-A3B layer shapes, full RF head, full semantic decoder; 2 LLM layers, 3 visual tokens), 2 and 3 CFG rows:
    batch 1 (the reference's call shape), then the same image inside 64- / 63-row lock-step groups, then TP = 8 (all 8 shards on
    this GPU) — all in fp8 weight mode against the fp32 oracle on the dequantised weights; drift from the bf16 model reported."""
    from oracle import bailing_ref, mingtok_ref
    from ming_univision_amd.bailing_moe import BailingMoeDecoder, generate_image, generate_images
    from ming_univision_amd.rf_head import RectifiedFlowHead
    from ming_univision_amd.tp import TpSimGroup
    d, rf_cfg, sd, ocfg, seed = full
    B = 32 if rows_tag == "rows2" else 21
    cfg, dsd, dec8, rf8, sd8, lsd, tok = _fp8_models(full, 3 * B)
    tsd = {k: v.float().cpu() for k, v in tok.sd.items()}
    g = torch.Generator().manual_seed(1)
    T = 12
    ids = torch.randint(0, 900, (1, T), generator=g)
    noises = torch.randn(cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    am = torch.ones(1, T + 1, dtype=torch.long)
    un = am.clone(); un[0, 2:T - 2] = 0
    tu = am.clone(); tu[0, 2:5] = 0
    if rows_tag == "rows2":
        tu = un.clone()

    def oracle(weights):
        kvs = bailing_ref.new_kv(ocfg)
        bailing_ref.model_forward(weights["model.word_embeddings.weight"][ids], weights, ocfg, torch.ones(1, T, dtype=torch.long), None, kvs)
        caches = mingtok_ref.semdec_new_cache(tsd)
        return bailing_ref.generate_image(
            weights["model.word_embeddings.weight"][torch.tensor([[cfg.image_start_token]])], kvs, am, un, tu, weights, ocfg, noises,
            latent_to_sem=lambda lat: mingtok_ref.mingtok_feature_decoder_step(lat, tsd, caches),
            linear_proj=lambda s: bailing_ref.linear_proj(s, lsd), sem_to_pix=lambda s: None, steps=int(rf_cfg["num_sampling_steps"]))
    ref8 = oracle(sd8)
    R = ref8["last_hidden"].shape[0]
    assert R == (2 if rows_tag == "rows2" else 3)
    start = dec8.embed(torch.tensor([cfg.image_start_token]).cuda())
    # ---- batch 1
    dec8.prefill(dec8.embed(ids[0].cuda()), seq=0, past=0)
    out = generate_image(dec8, rf8, tok, start, T, am, un, tu, noises.cuda(), decode_pixels=False)
    errs = (rel_err(out["latents"], ref8["latents"][:, 0]), rel_err(out["sem"], ref8["sem"][0]), rel_err(out["last_hidden"], ref8["last_hidden"][:, 0]))
    print("fp8 batch 1 (%s) vs oracle on dequantised weights: latents %.2e sem %.2e hidden %.2e" % ((rows_tag,) + errs))
    assert max(errs) < TOL, errs
    # ---- drift of the fp8 model from the bf16 model (both oracles; the HIP bf16 path is tested elsewhere)
    ref16 = oracle(sd)
    drift = (rel_err(ref8["latents"], ref16["latents"]), rel_err(ref8["sem"], ref16["sem"]), rel_err(ref8["last_hidden"], ref16["last_hidden"]))
    print("fp8 MODEL drift from the bf16 model (%s, oracle vs oracle): latents %.2e sem %.2e hidden %.2e" % ((rows_tag,) + drift))
    assert all(np.isfinite(drift))
    # ---- the same image inside a full lock-step group (K-loop kernel with four row tiles, every expert active)
    for i in range(B):
        dec8.prefill(dec8.embed(ids[0].cuda()), seq=i * R, past=0)
    nb = torch.randn(B, cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    nb[0] = noises
    outb = generate_images(dec8, rf8, tok, start, [T] * B, [am] * B, [un] * B, [tu] * B, nb.cuda(), decode_pixels=False, n_groups=1)
    errb = (rel_err(outb["latents"][0], ref8["latents"][:, 0]), rel_err(outb["last_hidden"][:R], ref8["last_hidden"][:, 0]))
    print("fp8 %d rows in one group: image 0 latents %.2e hidden %.2e" % ((B * R,) + errb))
    assert max(errb) < TOL, errb
    # ---- TP = 8 + EP = 8 with fp8 shards (sliced from the quantised model: same values), all on this GPU
    dec1 = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=32, n_seq=3, weights="fp8")
    grp = TpSimGroup(dec1, rf8, 8, rows_cap=16)
    assert grp.max_rows() == 16 and grp.shards[0].weights == "fp8" and grp.rf_shards[0].weights == "fp8"
    grp.prefill(dec1.embed(ids[0].cuda()), seq=0, past=0)
    outt = generate_image(grp, grp.sampler(), tok, start, T, am, un, tu, noises.cuda(), decode_pixels=False)
    grp.check_err()
    errt = (rel_err(outt["latents"], ref8["latents"][:, 0]), rel_err(outt["sem"], ref8["sem"][0]), rel_err(outt["last_hidden"], ref8["last_hidden"][:, 0]))
    print("fp8 TP = 8 (simulated) vs oracle: latents %.2e sem %.2e hidden %.2e" % errt)
    assert max(errt) < TOL, errt
    full_b = sum(t.numel() * t.element_size() for ly in dec1.layers for k, t in ly.items() if t is not None and k not in ("ln1", "ln2"))
    print("decoder-stack weight bytes: fp8 %.2f GB (per TP rank %.2f GB)" % (full_b / 1e9, grp.shards[0].weight_bytes() / 1e9))


def test_fp8_text_steps_and_long_prompt(full):
    """One-row steps (text decode: grouped expert kernels at 1 row) and a 150-token prompt (64-row passes) in fp8 mode vs the oracle on the
    dequantised weights; greedy tokens equal."""
    from oracle import bailing_ref
    d, rf_cfg, sd, ocfg, seed = full
    cfg, dsd, dec8, rf8, sd8, lsd, tok = _fp8_models(full, 3)
    g = torch.Generator().manual_seed(4)
    T = 150
    ids = torch.randint(0, 900, (1, T), generator=g)
    dec8t = type(dec8).from_state_dict(cfg, dsd, t_max=T + 8, n_seq=1, weights="fp8")
    kvs = bailing_ref.new_kv(ocfg)
    ref = bailing_ref.model_forward(sd8["model.word_embeddings.weight"][ids], sd8, ocfg, torch.ones(1, T, dtype=torch.long), None, kvs)
    hid = dec8t.prefill_mfma(dec8t.embed(ids[0].cuda()), seq=0, past=0)             # fp8 mode: delegates to 64-row passes
    ref_last = ref[0, -1:]
    e = rel_err(hid, ref_last)
    print("fp8 150-token prompt, last hidden vs oracle: %.2e" % e)
    assert e < TOL
    # four greedy one-row steps
    tok_hip = int(dec8t.greedy(hid)[0])
    assert tok_hip == int(bailing_ref.lm_logits(ref_last, sd8).argmax(-1)[0])
    cur, past = tok_hip, T
    for _ in range(4):
        x = dec8t.embed(torch.tensor([cur]).cuda())
        slot = torch.tensor([past], dtype=torch.int32, device="cuda")
        h = dec8t.step(x, torch.zeros(1, dtype=torch.int32, device="cuda"), slot, slot, slot + 1)
        r = bailing_ref.model_forward(sd8["model.word_embeddings.weight"][torch.tensor([[cur]])], sd8, ocfg,
                                      torch.ones(1, past + 1, dtype=torch.long), None, kvs)
        r_last = r[0, -1:]
        assert rel_err(h, r_last) < TOL, rel_err(h, r_last)
        cur = int(dec8t.greedy(h)[0])
        assert cur == int(bailing_ref.lm_logits(r_last, sd8).argmax(-1)[0])
        past += 1


def test_fp8_facade_dtype_switch(tmp_path):
    """MingUniVisionInfer(dtype="fp8") (the reference's `dtype` argument, mingunivisioninfer.py:46-70): a checkpoint directory of
    bf16 safetensors loads in fp8 mode, generates an image + text, and rejects unknown dtypes."""
    from ming_univision_amd.infer import MingUniVisionInfer
    d = dict(vocab_size=512, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128, use_bias=False,
             rope_theta=600000.0, num_experts=8, num_shared_experts=2, num_experts_per_tok=3, moe_intermediate_size=64, multi_gate=True,
             num_image_tokens_for_gen=4, image_start_token=500, eos_token_id=1, pad_token_id=0)
    rf_cfg = dict(diffloss_w=64, diffloss_d=2, num_sampling_steps="4", gen_method="flow_matching_swiglu-4")
    tcfg = dict(low_level_encoder=dict(img_size=64, patch_size=32, depth=2, embed_dim=128, ffn_layer="swiglufused", out_dim=32),
                semantic_decoder=dict(in_dim=32, patch_size=32, embed_dim=128, decoder_depth=2, ffn_layer="swiglufused"),
                pixel_decoder=dict(patch_size=16, decoder_depth=2, embed_dim=128))
    cfg = C.MingUniVisionConfig(mlp_depth=2, llm_config=d, vishead_diffloss_config=rf_cfg, mingtok_config=tcfg)
    with pytest.raises(ValueError):
        MingUniVisionInfer(None, dtype="fp4", config=cfg)
    inf8 = MingUniVisionInfer(None, dtype="fp8", config=cfg, seed=3, t_max=128)
    assert inf8.model.model.weights == "fp8" and inf8.model.rf.weights == "fp8"
    inf16 = MingUniVisionInfer(None, dtype="bf16", config=cfg, seed=3, t_max=128)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(2, 400, (1, 9), generator=g)
    unc = torch.ones(1, 9, dtype=torch.long); unc[0, 2:7] = 0
    req = dict(input_ids=ids, attention_mask=torch.ones(1, 9, dtype=torch.long), uncond_attention_mask=unc, text_uncond_attention_mask=unc.clone())
    noises = torch.randn(1, 5, 32, generator=g)
    o8 = inf8.model.generate_image_batch([req], forced_first_token=500, noises=noises, save=False)
    o16 = inf16.model.generate_image_batch([req], forced_first_token=500, noises=noises, save=False)
    assert torch.isfinite(o8["images"]).all() and o8["images"].shape == o16["images"].shape
    psnr = 10 * np.log10(4.0 / float(((o8["images"] - o16["images"]) ** 2).mean()))
    print("tiny random-init model: PSNR(fp8 image, bf16 image) = %.1f dB (reported, not gated: see the module docstring)" % psnr)
    assert float((o8["images"] - o16["images"]).abs().max()) > 0          # the two formats really are different models


def test_fp8_checkpoint_directory(tmp_path):
    """The loader / repacker in fp8 mode: a checkpoint directory of bf16 safetensors shards keyed by the reference's
    parameter names loads with dtype="fp8" (experts packed, then quantised; RF ResBlock matrices quantised), holds exactly the bytes /
    scales the oracle's quantiser gives for those tensors, and generates the same tokens as the state-dict constructor in fp8 mode."""
    from safetensors.torch import save_file
    from oracle import fp8_ref
    from ming_univision_amd.infer import MingUniVisionInfer
    from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration
    from tests.util import load_golden, mingtok_sd
    g = load_golden("genimg_tiny")
    llm_cfg = dict(g["llm_config"]); llm_cfg["eos_token_id"] = 1
    cfg = C.MingUniVisionConfig(mlp_depth=2, llm_config=llm_cfg, vishead_diffloss_config=g["rf_config"], mingtok_config=g["mingtok_config"])
    sd = llm_sd(g["llm_config"], g["rf_config"], g["seed"])
    ckpt = {"model." + k: v for k, v in sd.items()}
    ckpt.update({"vision." + k: v for k, v in mingtok_sd(g["mingtok_config"], g["seed"]).items()})
    ckpt.update(synth_state_dict(C.linear_proj_param_shapes(128, 256, 2), g["seed"]))
    d = tmp_path / "ckpt"
    d.mkdir()
    (d / "config.json").write_text(cfg.to_json_string())
    keys = sorted(ckpt)
    for i, part in enumerate((keys[:len(keys) // 2], keys[len(keys) // 2:])):
        save_file({k: ckpt[k].to(torch.bfloat16).contiguous() for k in part}, str(d / f"model-0000{i + 1}-of-00002.safetensors"))
    infer = MingUniVisionInfer(str(d), dtype="fp8", t_max=64)
    dec, rf = infer.model.model, infer.model.rf
    assert dec.weights == "fp8" and rf.weights == "fp8"
    E, I = llm_cfg["num_experts"], llm_cfg["moe_intermediate_size"]
    bf = lambda k: sd[k].to(torch.bfloat16).float()
    q, s_ = fp8_ref.quantize_rows(torch.cat((bf("model.layers.1.mlp.experts.3.gate_proj.weight"), bf("model.layers.1.mlp.experts.3.up_proj.weight"))))
    assert torch.equal(dec.layers[1]["w_gate_up"][3].cpu(), q) and torch.equal(dec.layers[1]["w_gate_up_scale"][3].cpu(), s_)
    q, s_ = fp8_ref.quantize_rows(bf("model.layers.0.mlp.shared_experts.down_proj.weight")[:, I:2 * I].contiguous())
    assert torch.equal(dec.layers[0]["w_down"][E + 1].cpu(), q) and torch.equal(dec.layers[0]["w_down_scale"][E + 1].cpu(), s_)
    q, s_ = fp8_ref.quantize_rows(bf("diffloss.net.res_blocks.1.mlp.w3.weight"))
    assert torch.equal(rf.lists["w3"][1].cpu(), q) and torch.equal(rf.scales["w3"][1].cpu(), s_)
    direct = MingUniVisionForConditionalGeneration(cfg, state_dict=ckpt, seed=g["seed"], t_max=64, weights="fp8")
    ids = g["ids"]
    a = infer.model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=5)
    b = direct.generate(input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=5)
    assert a.tolist() == b.tolist()
    text = infer.generate([{"role": "HUMAN", "content": [{"type": "text", "text": "hi"}]}], max_new_tokens=3)
    assert isinstance(text, str)
