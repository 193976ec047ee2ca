"""int4 weight mode (mingnative.h section 7, MN_W_NF4; the reference's `dtype="int4"` surface, mingunivisioninfer.py:46-58: bitsandbytes
NF4 with blockwise absmax, bf16 compute) on the GPU.

Definition of parity as for the other weight-only modes: the int4 MODEL is the bf16 model with every converted nn.Linear weight W
replaced by W' = bf16(NF4[code] * absmax) (oracle/int4_ref.py: bitsandbytes' dequantize_4bit); the HIP path streams the 4-bit codes of
the RF ResBlock / adaLN matrices and of the experts through the weight-streaming kernels (decoded per 64-element block with v_perm_b32
lookups, w8_codec.h), holds W' as bf16 for the other Linears, and is held to the fp32 oracle FED W' at 1e-3."""
import numpy as np
import pytest
import torch

from ming_univision_amd import configuration as C
from ming_univision_amd.synth import synth_state_dict
from tests.util import llm_sd, rel_err
from tests.test_gpu_fp8 import full, TOL      # noqa: F401  (`full` is a fixture)

pytestmark = pytest.mark.gpu


def _dev(sd):
    return {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}


def test_nf4_quantiser_and_decoder_are_bit_identical_to_the_oracle():
    from oracle import int4_ref
    from ming_univision_amd import ops
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(300, 1408, generator=g) * torch.logspace(-7, 2, 300).unsqueeze(1)).to(torch.bfloat16)
    w[3] = 0                                                             # all-zero blocks: absmax 0
    t = int4_ref.table()
    w[5, :16] = t.to(torch.bfloat16); w[5, 16:64] = 0; w[5, 0] = -1.0    # absmax 1: (bf16-rounded) table entries map to themselves or a neighbour
    mids = int4_ref.midpoints()
    w[6, :64] = 0; w[6, 0] = 1.0; w[6, 1:16] = mids.to(torch.bfloat16)   # values on / next to the comparison tree's thresholds
    w[7, 64:128] = 0; w[7, 64] = -3.0; w[7, 65] = 3.0 * 0.86
    q, a = ops.quant_rows(w.cuda().contiguous(), "int4")
    co, ao = int4_ref.quantize_blocks(w)
    assert q.shape == (300, 704) and a.shape == (300, 22)
    assert torch.equal(a.cpu(), ao)
    assert torch.equal(int4_ref.unpack_kernel(q.cpu()), co), int((int4_ref.unpack_kernel(q.cpu()) != co).sum())
    assert torch.equal(q.cpu(), int4_ref.pack_kernel(co))
    dq = ops.dequant_rows(q, a, "int4")
    assert torch.equal(dq.float().cpu(), int4_ref.dequantize_blocks(co, ao))
    # every code value at every nibble position, against a spread of absmax values (the per-block table + v_perm lookups)
    codes = torch.stack([torch.roll(torch.arange(16).repeat(8), s) for s in range(64)]).to(torch.uint8)          # [64, 128]
    am = torch.logspace(-4, 3, 128).reshape(64, 2).contiguous()
    got = ops.dequant_rows(int4_ref.pack_kernel(codes).cuda().contiguous(), am.cuda(), "int4")
    assert torch.equal(got.float().cpu(), int4_ref.dequantize_blocks(codes, am))
    # tensors whose rows are not a multiple of 64: blocks over the flattened tensor (bitsandbytes' rule)
    for shape in ((3072, 32), (50, 72), (7, 9)):
        w2 = torch.randn(*shape, generator=g).to(torch.bfloat16)
        assert torch.equal(ops.fake_quant(w2.cuda(), "int4").float().cpu(), int4_ref.fake_quant(w2)), shape
    w3 = torch.randn(5, 64, 192, generator=g).to(torch.bfloat16)
    q3, a3 = ops.quant_rows(w3.cuda().contiguous(), "int4")
    assert q3.shape == (5, 64, 96) and a3.shape == (5, 64, 3)
    assert torch.equal(ops.dequant_rows(q3, a3, "int4").float().cpu(), int4_ref.fake_quant(w3))


@pytest.mark.parametrize("M", [1, 2, 3, 16, 17, 32, 33, 48, 64])
def test_stream_mfma_nf4_against_float64(M):
    """Dense NF4 launches of every kernel form (K-slice <= 32 rows with one / two row tiles, K-loop above: 256-k pieces) at the RF head's
    shapes and at ragged ones (N not a multiple of 16, K = 64 .. not a multiple of 256) against the float64 product of the SAME operands
    (the hi/lo activation pair times the dequantised weights)."""
    from ming_univision_amd import ops
    g = torch.Generator().manual_seed(200 + M)
    shapes = [(2 * 8192, 3072), (3072, 8192), (1000, 1408), (40, 64), (2816, 2048), (330, 448), (72, 192), (50, 1216)]
    for N, K in shapes:
        w = (torch.randn(N, K, generator=g) * K ** -0.5 * torch.logspace(-1, 1, N).unsqueeze(1)).to(torch.bfloat16).cuda()
        q, a = ops.quant_rows(w, "int4")
        x = torch.randn(M, K, generator=g)
        a2 = ops.split_hilo(x.cuda())
        out = ops.stream_mfma_w8(a2.contiguous(), q, a, wfmt="int4")
        ref = (a2[0].double() + a2[1].double()) @ ops.dequant_rows(q, a, "int4").double().T
        e = rel_err(out, ref)
        assert e < 3e-5, (M, N, K, e)


def test_skinny_gemm_nf4_route_with_fused_prologue_and_epilogue():
    from ming_univision_amd import ops
    g = torch.Generator().manual_seed(9)
    K, N = 3072, 8192
    w = (torch.randn(2 * N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    b = (torch.randn(2 * N, generator=g) * 0.1).to(torch.bfloat16).cuda()
    q, a = ops.quant_rows(w, "int4")
    wd = ops.dequant_rows(q, a, "int4").double()
    for M in (1, 2, 8, 40):
        x = torch.randn(M, K, generator=g).cuda()
        sh, sc = (torch.randn(M, K, generator=g) * 0.1).cuda(), (torch.randn(M, K, generator=g) * 0.1).cuda()
        y = ops.skinny_gemm(x, q, b, prologue="ln_mod", epilogue="swiglu", eps=1e-6, pro_a=sh, pro_b=sc, wscale=a, wfmt="int4")
        xn = torch.nn.functional.layer_norm(x.double(), (K,), eps=1e-6) * (1 + sc.double()) + sh.double()
        r = xn @ wd.T + b.double()
        ref = torch.nn.functional.silu(r[:, :N]) * r[:, N:]
        assert rel_err(y, ref) < 2e-5, (M, rel_err(y, ref))


def _int4_models(full, n_seq):
    """HIP models of the full-width 2-layer configuration in int4 mode + the oracle's state dict of that model (W' everywhere)."""
    from oracle import int4_ref
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    from ming_univision_amd.mingtok import MingTok
    from ming_univision_amd.rf_head import RectifiedFlowHead
    from ming_univision_amd import ops
    d, rf_cfg, sd, ocfg, seed = full
    cfg = C.BailingMoeConfig(**d)
    dsd = _dev(sd)
    dec4 = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=32, n_seq=n_seq, weights="int4")
    rf4 = RectifiedFlowHead(dsd, cfg.hidden_size, rf_cfg, weights="int4")
    assert dec4.layers[0]["w_gate_up"].dtype == torch.uint8 and dec4.layers[0]["w_gate_up"].shape[-1] == cfg.hidden_size // 2
    assert rf4.lists["w12"][0].dtype == torch.uint8 and dec4.max_rows() == 2048 and rf4.max_rows() == 2048
    sd4 = dict(sd)
    for k, v in dec4.dequantized_state_dict().items():
        sd4[k] = v.float().cpu()
    for k, v in rf4.dequantized_blocks().items():
        sd4[k] = v.float().cpu()
    # ... every nn.Linear of the decoder stack and the head is the oracle's fake-quantised weight; nothing else changed
    converted = 0
    for k, v in sd.items():
        is_linear = k.endswith(".weight") and v.dim() == 2 and "word_embeddings" not in k and ".gate.weight" not in k and "_gate.weight" not in k
        if is_linear and "shared_experts.down_proj" not in k:
            assert torch.equal(sd4[k], int4_ref.fake_quant(sd[k])), k
            converted += 1
        elif not is_linear:
            assert torch.equal(sd4[k], sd[k]), k
    assert converted == 2 * (2 + 3 * 64 + 2) + 1 + 1 + 4 + 2 + 3 * 12          # qkv, dense, experts, shared gate/up | lm_head | vis_head, RF
    # the shared expert's down projection is quantised per pseudo-expert (column block of the packed layout): blocks stay inside it
    k = "model.layers.0.mlp.shared_experts.down_proj.weight"
    I = cfg.moe_intermediate_size
    assert torch.equal(sd4[k], torch.cat([int4_ref.fake_quant(sd[k][:, s * I:(s + 1) * I].contiguous()) for s in range(2)], 1))
    lsd = synth_state_dict(C.linear_proj_param_shapes(1024, cfg.hidden_size, 2), seed)
    dl = ops.convert_linears(_dev(lsd), "int4")
    lsd4 = {k: v.float().cpu() for k, v in dl.items()}
    assert torch.equal(lsd4["linear_proj.0.weight"], int4_ref.fake_quant(lsd["linear_proj.0.weight"]))
    tok = MingTok(C.MingTokConfig(), device="cuda", seed=seed, weights="int4",
                  linear_proj=[(dl["linear_proj.0.weight"], dl["linear_proj.0.bias"]), (dl["linear_proj.2.weight"], dl["linear_proj.2.bias"])])
    return cfg, dsd, dec4, rf4, sd4, lsd4, tok


@pytest.mark.parametrize("rows_tag", ["rows2", "rows3"])
def test_int4_full_width_generate_image_vs_oracle_on_dequantised_weights(full, rows_tag):
    """Full width (16B-A3B layer shapes, full RF head, full semantic decoder; 2 LLM layers, 3 visual tokens), 2 and 3 CFG rows:
    batch 1 (the reference's call shape: fused RF chain, grouped expert launches at 2 / 3 rows), then the same image inside 64- /
    63-row lock-step groups (K-loop form, every expert active) — int4 mode against the fp32 oracle on W'."""
    from oracle import bailing_ref, mingtok_ref
    from ming_univision_amd.bailing_moe import generate_image, generate_images
    d, rf_cfg, sd, ocfg, seed = full
    B = 32 if rows_tag == "rows2" else 21
    cfg, dsd, dec4, rf4, sd4, lsd4, tok = _int4_models(full, 3 * B)
    tsd = {k: v.float().cpu() for k, v in tok.sd.items()}               # (already the int4 model's values)
    g = torch.Generator().manual_seed(1)
    T = 12
    ids = torch.randint(0, 900, (1, T), generator=g)
    noises = torch.randn(cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    am = torch.ones(1, T + 1, dtype=torch.long)
    un = am.clone(); un[0, 2:T - 2] = 0
    tu = am.clone(); tu[0, 2:5] = 0
    if rows_tag == "rows2":
        tu = un.clone()

    def oracle(weights, lw):
        kvs = bailing_ref.new_kv(ocfg)
        bailing_ref.model_forward(weights["model.word_embeddings.weight"][ids], weights, ocfg, torch.ones(1, T, dtype=torch.long), None, kvs)
        caches = mingtok_ref.semdec_new_cache(tsd)
        return bailing_ref.generate_image(
            weights["model.word_embeddings.weight"][torch.tensor([[cfg.image_start_token]])], kvs, am, un, tu, weights, ocfg, noises,
            latent_to_sem=lambda lat: mingtok_ref.mingtok_feature_decoder_step(lat, tsd, caches),
            linear_proj=lambda s: bailing_ref.linear_proj(s, lw), sem_to_pix=lambda s: None, steps=int(rf_cfg["num_sampling_steps"]))
    ref4 = oracle(sd4, lsd4)
    R = ref4["last_hidden"].shape[0]
    start = dec4.embed(torch.tensor([cfg.image_start_token]).cuda())
    dec4.prefill(dec4.embed(ids[0].cuda()), seq=0, past=0)
    out = generate_image(dec4, rf4, tok, start, T, am, un, tu, noises.cuda(), decode_pixels=False)
    errs = (rel_err(out["latents"], ref4["latents"][:, 0]), rel_err(out["sem"], ref4["sem"][0]), rel_err(out["last_hidden"], ref4["last_hidden"][:, 0]))
    print("int4 batch 1 (%s) vs oracle on dequantised weights: latents %.2e sem %.2e hidden %.2e" % ((rows_tag,) + errs))
    assert max(errs) < TOL, errs
    for i in range(B):
        dec4.prefill(dec4.embed(ids[0].cuda()), seq=i * R, past=0)
    nb = torch.randn(B, cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    nb[0] = noises
    outb = generate_images(dec4, rf4, tok, start, [T] * B, [am] * B, [un] * B, [tu] * B, nb.cuda(), decode_pixels=False, n_groups=1)
    errb = (rel_err(outb["latents"][0], ref4["latents"][:, 0]), rel_err(outb["last_hidden"][:R], ref4["last_hidden"][:, 0]))
    print("int4 %d rows in one group: image 0 latents %.2e hidden %.2e" % ((B * R,) + errb))
    assert max(errb) < TOL, errb
    full_b = sum(t.numel() * t.element_size() for ly in dec4.layers for k, t in ly.items() if torch.is_tensor(t) and k not in ("ln1", "ln2"))
    print("decoder-stack weight bytes per layer: int4 %.3f GB" % (full_b / 1e9 / cfg.num_hidden_layers))
    # ---- TP = 8 + EP = 8 with NF4 shards, all on this GPU: the shared expert's 44 absmax blocks are dealt out whole (6, 6, 6, 6, 5, 5, 5, 5),
    # so the shards hold the unsharded int4 model's own codes — the same oracle
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    from ming_univision_amd.tp import TpSimGroup, nf4_shared_units
    assert [nf4_shared_units(cfg, r, 8)[1] for r in range(8)] == [384] * 4 + [320] * 4
    assert sum(nf4_shared_units(cfg, r, 8)[1] for r in range(8)) == cfg.num_shared_experts * cfg.moe_intermediate_size
    dec1 = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=32, n_seq=3, weights="int4")
    grp = TpSimGroup(dec1, rf4, 8, rows_cap=16)
    assert grp.shards[0].weights == "int4" and grp.rf_shards[0].weights == "int4" and grp.shards[0].struct.wfmt == 3
    grp.prefill(dec1.embed(ids[0].cuda()), seq=0, past=0)
    outt = generate_image(grp, grp.sampler(), tok, start, T, am, un, tu, noises.cuda(), decode_pixels=False)
    grp.check_err()
    errt = (rel_err(outt["latents"], ref4["latents"][:, 0]), rel_err(outt["sem"], ref4["sem"][0]), rel_err(outt["last_hidden"], ref4["last_hidden"][:, 0]))
    print("int4 TP = 8 (simulated, %s) vs oracle: latents %.2e sem %.2e hidden %.2e; weight bytes per TP rank %.3f GB" % (
        (rows_tag,) + errt + (grp.shards[0].weight_bytes() / 1e9,)))
    assert max(errt) < TOL, errt


def test_int4_text_steps_and_long_prompt(full):
    """One-row steps (text decode: grouped NF4 expert launches at 1 row) and a 150-token prompt (64-row passes), greedy tokens through the
    converted lm_head — against the oracle on W'."""
    from oracle import bailing_ref
    cfg, dsd, dec4, rf4, sd4, lsd4, tok = _int4_models(full, 3)
    d, rf_cfg, sd, ocfg, seed = full
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 900, (150,), generator=g)
    kvs = bailing_ref.new_kv(ocfg)
    h_ref = bailing_ref.model_forward(sd4["model.word_embeddings.weight"][ids[None]], sd4, ocfg, None, None, kvs)[:, -1:]
    dec4b = dec4.view(t_max=192, n_seq=1)
    h = dec4b.prefill(dec4b.embed(ids.cuda()), seq=0, past=0)[-1:]
    e0 = rel_err(h, h_ref[0])
    toks_ref, toks = [], []
    slot = torch.tensor([150], dtype=torch.int32, device="cuda")
    seq0 = torch.zeros(1, dtype=torch.int32, device="cuda")
    for i in range(4):
        t_ref = int(bailing_ref.lm_logits(h_ref, sd4).argmax())
        t_dev = int(dec4b.greedy(h)[0])
        toks_ref.append(t_ref); toks.append(t_dev)
        h_ref = bailing_ref.model_forward(sd4["model.word_embeddings.weight"][torch.tensor([[t_ref]])], sd4, ocfg, None, None, kvs)[:, -1:]
        h = dec4b.step(dec4b.embed(torch.tensor([t_ref]).cuda()), seq0, slot + i, slot + i, slot + i + 1)
        assert rel_err(h, h_ref[0]) < TOL, (i, rel_err(h, h_ref[0]))
    print("int4 150-token prompt: last hidden %.2e; greedy tokens %s vs %s" % (e0, toks, toks_ref))
    assert e0 < TOL and toks == toks_ref


def test_int4_facade_dtype_switch(tmp_path):
    """MingUniVisionInfer(dtype="int4") (mingunivisioninfer.py:46-58) builds the int4 model end to end (tiny synthetic configuration):
    4-bit experts / RF matrices, every other nn.Linear — vision tower, linear_proj, attention, lm_head — converted, images finite."""
    from ming_univision_amd.infer import MingUniVisionInfer
    from ming_univision_amd import _lib
    from tests.util import load_golden
    g = load_golden("genimg_tiny")
    llm_cfg = dict(g["llm_config"]); llm_cfg["eos_token_id"] = 1
    rf_cfg = dict(g["rf_config"]); rf_cfg["diffloss_w"] = 192            # SwiGLU hidden 512: rows of whole 64-element blocks
    cfg = C.MingUniVisionConfig(mlp_depth=2, llm_config=llm_cfg, vishead_diffloss_config=rf_cfg, mingtok_config=g["mingtok_config"])
    inf4 = MingUniVisionInfer(None, dtype="int4", config=cfg, seed=g["seed"], t_max=64)
    inf16 = MingUniVisionInfer(None, dtype="bf16", config=cfg, seed=g["seed"], t_max=64)
    assert inf4.model.weights == "int4" and inf4.model.model.layers[0]["w_gate_up"].dtype == torch.uint8
    assert inf4.model.rf.stream_fmt == "int4" and inf4.model.rf.lists["w3"][0].dtype == torch.uint8
    # a head whose rows are not whole blocks (the golden tiny one: SwiGLU hidden 176) keeps the int4 model's VALUES as bf16 tensors
    cfg_odd = C.MingUniVisionConfig(mlp_depth=2, llm_config=llm_cfg, vishead_diffloss_config=dict(g["rf_config"]), mingtok_config=g["mingtok_config"])
    odd = MingUniVisionInfer(None, dtype="int4", config=cfg_odd, seed=g["seed"], t_max=64)
    assert odd.model.rf.weights == "int4" and odd.model.rf.stream_fmt == "bf16" and odd.model.rf.lists["w3"][0].dtype == torch.bfloat16
    from oracle import int4_ref
    odd16 = MingUniVisionInfer(None, dtype="bf16", config=cfg_odd, seed=g["seed"], t_max=64)
    raw = odd16.model.rf.lists["w3"][1].float().cpu()
    assert raw.shape == (64, 176) and torch.equal(odd.model.rf.lists["w3"][1].float().cpu(), int4_ref.fake_quant(raw))
    out_odd = odd.model.generate(input_ids=g["ids"], attention_mask=torch.ones_like(g["ids"]), max_new_tokens=2,
                                 forced_first_token=llm_cfg["image_start_token"], output_image_prefix=str(tmp_path / "i4o"))
    assert torch.isfinite(odd.model.last_generation["latents"]).all() and out_odd.shape[1] == g["ids"].shape[1] + 2
    assert not torch.equal(inf4.model.model.lm_head, inf16.model.model.lm_head)          # lm_head is converted too
    assert not torch.equal(inf4.model.vision.sd["semantic_decoder.in_proj.weight"], inf16.model.vision.sd["semantic_decoder.in_proj.weight"])
    assert torch.equal(inf4.model.model.layers[0]["gate"], inf16.model.model.layers[0]["gate"])     # the router gate is not an nn.Linear
    ids = g["ids"]
    out = inf4.model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=2, forced_first_token=llm_cfg["image_start_token"],
                              output_image_prefix=str(tmp_path / "i4"))
    assert out.shape[1] == ids.shape[1] + 2 and torch.isfinite(inf4.model.last_generation["latents"]).all()
    with pytest.raises(ValueError):
        MingUniVisionInfer(None, dtype="int3", config=cfg)


def test_quantised_head_does_not_pin_the_bf16_matrices():
    """a head BUILT in a weight-only mode (the model load path) must not keep the bf16 ResBlock matrices it was quantised
    from — `torch.cuda.memory_allocated` after the caller drops its state dict is the codes + the small tensors, not codes + originals."""
    import gc
    from ming_univision_amd.rf_head import RectifiedFlowHead
    rf_cfg = dict(diffloss_w=768, diffloss_d=4, num_sampling_steps="4", gen_method="flow_matching_swiglu-4")      # SwiGLU hidden 2048: whole NF4 blocks
    cfg = C.BailingMoeConfig(vocab_size=64, hidden_size=256, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                             head_dim=128, num_experts=4, num_shared_experts=1, num_experts_per_tok=2, moe_intermediate_size=64)
    shapes = {k: s for k, s in C.llm_param_shapes(cfg, rf_cfg, 32).items() if k.startswith("vis_head") or k.startswith("diffloss")}

    def footprint(mode):
        gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
        base = torch.cuda.memory_allocated()
        sd = _dev(synth_state_dict(shapes, 3))
        head = RectifiedFlowHead(sd, cfg.hidden_size, rf_cfg, weights=mode)
        del sd
        gc.collect(); torch.cuda.synchronize()
        used = torch.cuda.memory_allocated() - base
        assert (head._raw_t is None) == (mode != "bf16")
        return used, head
    b16, h16 = footprint("bf16")
    blocks_bf16 = 2 * sum(w_.numel() for k in ("w12", "w3") for w_ in h16.lists[k])
    del h16
    for mode, codes_over_bf16 in (("int4", 0.25 + 4 / 128.0), ("int8", 0.5), ("fp8", 0.5)):
        used, head = footprint(mode)
        assert head.stream_fmt == mode
        saved = b16 - used
        # the ResBlock matrices shrink to their codes; nothing else grows by more than the adaLN codes (+ row scales)
        ada = head.t["ada_q"].numel() + 4 * head.t["ada_scale"].numel()
        assert saved >= blocks_bf16 * (1 - codes_over_bf16) - ada - (2 << 20), (mode, b16, used, blocks_bf16)
        del head
