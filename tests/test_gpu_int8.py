"""int8 weight mode (mingnative.h section 7, MN_W_INT8; the reference's `dtype="int8"` surface, mingunivisioninfer.py:59-68: HF
QuantoConfig(weights="int8") = optimum-quanto qint8 weights) on the GPU.

Definition of parity as for the other weight-only modes: the int8 MODEL is the bf16 model with every converted nn.Linear weight W
replaced by W' = bf16(scale * q), scale = bf16(amax / 127) per output row, q = clamp(round(bf16(W / scale)), -128, 127)
(oracle/int8_ref.py: quanto's rule restated); the HIP path streams (q, scale) of the RF ResBlock / adaLN matrices and of the experts
through the weight-streaming kernels — the row scale rides the byte conversion, the product is rounded to bf16 per element —, holds W'
as bf16 for the other Linears, and is held to the fp32 oracle FED W' at 1e-3."""
import numpy as np
import pytest
import torch

from ming_univision_amd import configuration as C
from tests.util import rel_err
from tests.test_gpu_fp8 import _fp8_models, full, TOL      # noqa: F401  (`full` is a fixture)

pytestmark = pytest.mark.gpu


def test_int8_bytes_decode_exactly_in_every_kernel():
    """All 256 byte values (-128 included: the bf16 quotient can reach it) against a spread of row scales through the streaming
    kernels' conversion (K-slice and K-loop forms) and the dequantiser: bf16_rne(q * scale), bit for bit."""
    from ming_univision_amd import ops
    vals = torch.arange(-128, 128, dtype=torch.int16)
    bytes_ = vals.to(torch.int8).view(torch.uint8)
    s = torch.logspace(-3, 3, 256).to(torch.bfloat16).float()
    s[7], s[200] = 0.0123456, 3.1415926                                      # (the kernels do not need a bf16-valued scale)
    want = (vals.float() * s).to(torch.bfloat16).float()
    for K, M in ((16, 16), (256, 3), (32, 20), (128, 40), (1024, 1), (64, 64)):
        q = bytes_.unsqueeze(1).repeat(1, K).contiguous().cuda()
        x = torch.zeros(M, K)
        x[torch.arange(M), torch.arange(M) % K] = 1.0
        out = ops.stream_mfma_w8(ops.split_hilo(x.cuda()).contiguous(), q, s.cuda().contiguous(), wfmt="int8")
        assert torch.equal(out.cpu(), want.unsqueeze(0).expand(M, 256)), (K, M)
    q = bytes_.unsqueeze(1).repeat(1, 8).contiguous().cuda()
    dq = ops.dequant_rows(q, s.cuda().contiguous(), "int8")
    assert torch.equal(dq.float().cpu(), want.unsqueeze(1).expand(256, 8))


def test_int8_quantiser_is_bit_identical_to_the_oracle():
    from oracle import int8_ref
    from ming_univision_amd import ops
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(300, 1408, generator=g) * torch.logspace(-7, 2, 300).unsqueeze(1)).to(torch.bfloat16)
    w[3] = 0
    w[5, :] = 0; w[5, 7] = 127.0
    w[6, :] = 0; w[6, 9] = 254.0; w[6, 10] = 1.0; w[6, 11] = 3.0; w[6, 12] = -5.0            # scale 2: ties 0.5, 1.5, -2.5
    w[8, :] = 0; w[8, :256] = (torch.arange(256).float() - 127.5).to(torch.bfloat16); w[8, 300] = 127.0      # every tie, both signs
    q, s = ops.quant_rows(w.cuda().contiguous(), "int8")
    qo, so = int8_ref.quantize_rows(w)
    assert torch.equal(s.cpu(), so)
    assert torch.equal(q.cpu(), qo), int((q.cpu() != qo).sum())
    assert torch.equal(ops.dequant_rows(q, s, "int8").float().cpu(), int8_ref.dequantize_rows(qo, so))
    assert torch.equal(ops.fake_quant(w.cuda(), "int8").float().cpu(), int8_ref.fake_quant_rows(w))
    w3 = torch.randn(5, 64, 96, generator=g).to(torch.bfloat16)
    q3, s3 = ops.quant_rows(w3.cuda().contiguous(), "int8")
    qo3, so3 = int8_ref.quantize_rows(w3)
    assert s3.shape == (5, 64) and torch.equal(s3.cpu(), so3) and torch.equal(q3.cpu(), qo3)


@pytest.mark.parametrize("M", [1, 2, 3, 16, 17, 32, 33, 48, 64])
def test_stream_mfma_int8_against_float64(M):
    """Dense int8 launches of every kernel form (K-slice <= 32 rows with one / two row tiles, K-loop above) at the RF head's shapes and
    at ragged ones against the float64 product of the SAME operands."""
    from ming_univision_amd import ops
    g = torch.Generator().manual_seed(200 + M)
    for N, K in [(2 * 8192, 3072), (3072, 8192), (1000, 1408), (40, 16), (2816, 2048), (330, 464), (72, 176), (50, 1232)]:
        w = (torch.randn(N, K, generator=g) * K ** -0.5 * torch.logspace(-1, 1, N).unsqueeze(1)).to(torch.bfloat16).cuda()
        q, s = ops.quant_rows(w, "int8")
        x = torch.randn(M, K, generator=g)
        a2 = ops.split_hilo(x.cuda())
        out = ops.stream_mfma_w8(a2.contiguous(), q, s, wfmt="int8")
        ref = (a2[0].double() + a2[1].double()) @ ops.dequant_rows(q, s, "int8").double().T
        e = rel_err(out, ref)
        assert e < 3e-5, (M, N, K, e)


def test_int8_has_no_batch_or_segment_form_of_the_one_row_kernel():
    """The fp32-FMA pair kernels (skinny_w8.hip) multiply exact byte values and scale the sums afterwards — e4m3's rule, not quanto's
    per-element bf16 product — so int8 experts run the grouped streaming launch from one row on and the batch / K-segment forms of
    mn_skinny_gemm refuse int8 instead of computing another model."""
    from ming_univision_amd import ops
    g = torch.Generator().manual_seed(141)
    E, S, I, H, top = 8, 2, 64, 256, 3
    gu = (torch.randn(E + S, 2 * I, H, generator=g) * H ** -0.5).to(torch.bfloat16).cuda()
    dn = (torch.randn(E + S, H, I, generator=g) * I ** -0.5).to(torch.bfloat16).cuda()
    gq, gs = ops.quant_rows(gu, "int8")
    dq, ds = ops.quant_rows(dn, "int8")
    xn, res = torch.randn(1, H, generator=g).cuda(), torch.randn(1, H, generator=g).cuda()
    idx = torch.tensor([[1, 4, 6, E, E + 1]], dtype=torch.int32).cuda()
    w = torch.ones(1, top + S).cuda()
    with pytest.raises(RuntimeError):
        ops.moe_experts(xn, idx, w, gq, dq, res, gate_up_scale=gs, down_scale=ds, wfmt="int8")


def test_int8_full_width_generate_image_vs_oracle_on_dequantised_weights(full):
    """Full width (16B-A3B layer shapes, full RF head, full semantic decoder; 2 LLM layers, 3 visual tokens), 2 CFG rows: batch 1 (the
    reference's call shape), the same image inside a 64-row lock-step group, TP = 8 (all shards on this GPU) — int8 weight mode against
    the fp32 oracle on the dequantised weights; drift from the bf16 model reported next to the fp8 mode's."""
    from oracle import bailing_ref, mingtok_ref
    from ming_univision_amd.bailing_moe import BailingMoeDecoder, generate_image, generate_images
    from ming_univision_amd.tp import TpSimGroup
    d, rf_cfg, sd, ocfg, seed = full
    B = 32
    cfg, dsd, dec8, rf8, sd8, lsd, tok = _fp8_models(full, 3 * B, "int8")
    assert dec8.weights == "int8" and rf8.weights == "int8" and dec8.struct.wfmt == 2 and rf8.struct.wfmt == 2
    tsd = {k: v.float().cpu() for k, v in tok.sd.items()}                       # (MingTok's Linears are converted too: tok.weights == "int8")
    assert tok.weights == "int8"
    g = torch.Generator().manual_seed(1)
    T = 12
    ids = torch.randint(0, 900, (1, T), generator=g)
    noises = torch.randn(cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    am = torch.ones(1, T + 1, dtype=torch.long)
    un = am.clone(); un[0, 2:T - 2] = 0
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
    start = dec8.embed(torch.tensor([cfg.image_start_token]).cuda())
    dec8.prefill(dec8.embed(ids[0].cuda()), seq=0, past=0)
    out = generate_image(dec8, rf8, tok, start, T, am, un, tu, noises.cuda(), decode_pixels=False)
    errs = (rel_err(out["latents"], ref8["latents"][:, 0]), rel_err(out["sem"], ref8["sem"][0]), rel_err(out["last_hidden"], ref8["last_hidden"][:, 0]))
    print("int8 batch 1 vs oracle on dequantised weights: latents %.2e sem %.2e hidden %.2e" % errs)
    assert max(errs) < TOL, errs
    ref16 = oracle(sd)
    drift = (rel_err(ref8["latents"], ref16["latents"]), rel_err(ref8["sem"], ref16["sem"]), rel_err(ref8["last_hidden"], ref16["last_hidden"]))
    print("int8 MODEL drift from the bf16 model (oracle vs oracle): latents %.2e sem %.2e hidden %.2e" % drift)
    assert all(np.isfinite(drift))
    for i in range(B):
        dec8.prefill(dec8.embed(ids[0].cuda()), seq=i * R, past=0)
    nb = torch.randn(B, cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    nb[0] = noises
    outb = generate_images(dec8, rf8, tok, start, [T] * B, [am] * B, [un] * B, [tu] * B, nb.cuda(), decode_pixels=False, n_groups=1)
    errb = (rel_err(outb["latents"][0], ref8["latents"][:, 0]), rel_err(outb["last_hidden"][:R], ref8["last_hidden"][:, 0]))
    print("int8 %d rows in one group: image 0 latents %.2e hidden %.2e" % ((B * R,) + errb))
    assert max(errb) < TOL, errb
    dec1 = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=32, n_seq=3, weights="int8")
    grp = TpSimGroup(dec1, rf8, 8, rows_cap=16)
    assert grp.max_rows() == 16 and grp.shards[0].weights == "int8" and grp.rf_shards[0].weights == "int8"
    grp.prefill(dec1.embed(ids[0].cuda()), seq=0, past=0)
    outt = generate_image(grp, grp.sampler(), tok, start, T, am, un, tu, noises.cuda(), decode_pixels=False)
    grp.check_err()
    errt = (rel_err(outt["latents"], ref8["latents"][:, 0]), rel_err(outt["sem"], ref8["sem"][0]), rel_err(outt["last_hidden"], ref8["last_hidden"][:, 0]))
    print("int8 TP = 8 (simulated) vs oracle: latents %.2e sem %.2e hidden %.2e" % errt)
    assert max(errt) < TOL, errt


def test_int8_text_steps_and_long_prompt(full):
    """One-row steps and a 150-token prompt (64-row passes) in int8 mode vs the oracle on the dequantised weights; greedy tokens equal."""
    from oracle import bailing_ref
    d, rf_cfg, sd, ocfg, seed = full
    cfg, dsd, dec8, rf8, sd8, lsd, tok = _fp8_models(full, 3, "int8")
    g = torch.Generator().manual_seed(4)
    T = 150
    ids = torch.randint(0, 900, (1, T), generator=g)
    dec8t = type(dec8).from_state_dict(cfg, dsd, t_max=T + 8, n_seq=1, weights="int8")
    kvs = bailing_ref.new_kv(ocfg)
    ref = bailing_ref.model_forward(sd8["model.word_embeddings.weight"][ids], sd8, ocfg, torch.ones(1, T, dtype=torch.long), None, kvs)
    hid = dec8t.prefill_mfma(dec8t.embed(ids[0].cuda()), seq=0, past=0)
    ref_last = ref[0, -1:]
    e = rel_err(hid, ref_last)
    print("int8 150-token prompt, last hidden vs oracle: %.2e" % e)
    assert e < TOL
    cur, past = int(dec8t.greedy(hid)[0]), T
    assert cur == int(bailing_ref.lm_logits(ref_last, sd8).argmax(-1)[0])
    for _ in range(3):
        x = dec8t.embed(torch.tensor([cur]).cuda())
        slot = torch.tensor([past], dtype=torch.int32, device="cuda")
        h = dec8t.step(x, torch.zeros(1, dtype=torch.int32, device="cuda"), slot, slot, slot + 1, distinct_sequences=True)
        r = bailing_ref.model_forward(sd8["model.word_embeddings.weight"][torch.tensor([[cur]])], sd8, ocfg,
                                      torch.ones(1, past + 1, dtype=torch.long), None, kvs)
        assert rel_err(h, r[0, -1:]) < TOL
        cur = int(dec8t.greedy(h)[0])
        assert cur == int(bailing_ref.lm_logits(r[0, -1:], sd8).argmax(-1)[0])
        past += 1


def test_int8_facade_dtype_switch():
    """MingUniVisionInfer(dtype="int8") — the reference's own `dtype` value (mingunivisioninfer.py:59): builds in int8 mode and
    generates an image; its drift from the bf16 model is reported next to the fp8 mode's."""
    from ming_univision_amd.infer import MingUniVisionInfer
    d = dict(vocab_size=512, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128, use_bias=False,
             rope_theta=600000.0, num_experts=8, num_shared_experts=2, num_experts_per_tok=3, moe_intermediate_size=64, multi_gate=True,
             num_image_tokens_for_gen=4, image_start_token=500, eos_token_id=1, pad_token_id=0)
    rf_cfg = dict(diffloss_w=64, diffloss_d=2, num_sampling_steps="4", gen_method="flow_matching_swiglu-4")
    tcfg = dict(low_level_encoder=dict(img_size=64, patch_size=32, depth=2, embed_dim=128, ffn_layer="swiglufused", out_dim=32),
                semantic_decoder=dict(in_dim=32, patch_size=32, embed_dim=128, decoder_depth=2, ffn_layer="swiglufused"),
                pixel_decoder=dict(patch_size=16, decoder_depth=2, embed_dim=128))
    cfg = C.MingUniVisionConfig(mlp_depth=2, llm_config=d, vishead_diffloss_config=rf_cfg, mingtok_config=tcfg)
    models = {dt: MingUniVisionInfer(None, dtype=dt, config=cfg, seed=3, t_max=128) for dt in ("bf16", "fp8", "int8")}
    assert models["int8"].model.model.weights == "int8" and models["int8"].model.rf.weights == "int8"
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(2, 400, (1, 9), generator=g)
    unc = torch.ones(1, 9, dtype=torch.long); unc[0, 2:7] = 0
    req = dict(input_ids=ids, attention_mask=torch.ones(1, 9, dtype=torch.long), uncond_attention_mask=unc, text_uncond_attention_mask=unc.clone())
    noises = torch.randn(1, 5, 32, generator=g)
    o = {dt: m.model.generate_image_batch([req], forced_first_token=500, noises=noises, save=False) for dt, m in models.items()}
    assert torch.isfinite(o["int8"]["images"]).all() and o["int8"]["images"].shape == o["bf16"]["images"].shape
    psnr = {dt: 10 * np.log10(4.0 / float(((o[dt]["images"] - o["bf16"]["images"]) ** 2).mean())) for dt in ("fp8", "int8")}
    print("tiny random-init model: PSNR(int8 image, bf16 image) = %.1f dB, fp8 %.1f dB (reported, not gated)" % (psnr["int8"], psnr["fp8"]))
    assert float((o["int8"]["images"] - o["bf16"]["images"]).abs().max()) > 0
