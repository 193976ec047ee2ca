"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (CPU, fp32).

Run in the build container only (needs /root/reference):

    python -m oracle.gen_golden            # writes tests/golden/*.npz

The fixtures are data: configs, seeds, inputs and the reference's outputs.  The
weights are NOT stored; they are re-derived from (parameter name, seed) by
ming_univision_amd.synth, and a checksum per fixture guards against RNG drift.
The reference's source never leaves /root/reference.

Reference entry points exercised (paths relative to /root/reference):
  mingtok/modeling_mingtok.py      MingTok.forward / forward_enc_dec /
                                   forward_feature_decoder / forward_pixel_decoder
  mingunivision/diff_loss_rf_swiglu.py   SimpleMLPAdaLN.forward, RectifiedFlowLoss.sample
  mingunivision/modeling_bailing_moe.py  BailingMoeModel.forward (eager attention),
                                   BailingMoeForCausalLM.generate_image, vis_head
  mingunivision/modeling_bailingmm.py    MingUniVisionForConditionalGeneration.generate over three rounds, PAST_MODE KEEP and DROP
                                   (round 6: gen_multiround -> multiround_tiny.npz; with BailingMoeForCausalLM.forward's `<image>`
                                   branch and prepare_inputs_for_generation, driven by the installed transformers' greedy loop
                                   through ref_shim.cache_position_452)
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from ming_univision_amd import configuration as C  # noqa: E402
from ming_univision_amd.synth import synth_tensor  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

TINY_MINGTOK = dict(
    low_level_encoder=dict(img_size=64, patch_size=32, depth=2, embed_dim=128, ffn_layer="swiglufused",
                           out_dim=32, fa_enable=False),
    semantic_decoder=dict(in_dim=32, patch_size=32, embed_dim=128, decoder_depth=2, ffn_layer="swiglufused",
                          fa_enable=False),
    pixel_decoder=dict(patch_size=16, decoder_depth=2, norm_pix_loss=True, embed_dim=128, loss_type="L1-plain",
                       fa_enable=False),
    scaling_factor=8.09449291, mean=1.46817409)

TINY_LLM = dict(vocab_size=512, hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, head_dim=128, use_qkv_bias=False, use_bias=False, rms_norm_eps=1e-5,
                rope_theta=600000.0, num_experts=8, num_shared_experts=2, num_experts_per_tok=3,
                moe_intermediate_size=64, norm_topk_prob=True, multi_gate=True, first_k_dense_replace=0,
                num_image_tokens_for_gen=4, image_start_token=500, image_patch_token=499,
                embedding_dropout=0.0, attention_dropout=0.0, output_dropout=0.0, pad_token_id=0)

TINY_RF = dict(diffloss_w=64, diffloss_d=2, num_sampling_steps="4", gen_method="flow_matching_swiglu-4",
               vis_head_arch="linear2-norm")


def fill_from_synth(module, seed, skip=("rotary_emb",)):
    """Overwrite every parameter of a reference module with synth(name, seed)."""
    sd = module.state_dict()
    new, shapes = {}, {}
    for k, v in sd.items():
        if any(s in k for s in skip):
            continue
        new[k] = synth_tensor(k, tuple(v.shape), seed)
        shapes[k] = tuple(v.shape)
    module.load_state_dict(new, strict=False)
    return shapes


def checksum(module):
    return float(sum(p.double().abs().sum() for p in module.parameters()))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        elif isinstance(v, (dict, list)):
            v = np.array(json.dumps(v))
        out[k] = v
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


class ReplayRandn:
    """Replace torch.randn inside RectifiedFlowLoss.sample by a recorded noise sequence."""

    def __init__(self, noises):
        self.noises, self.i, self._orig = noises, 0, None

    def __enter__(self):
        self._orig = torch.randn

        def fake(*shape, **kw):
            n = self.noises[self.i:self.i + 1].clone()
            self.i += 1
            assert tuple(n.shape) == tuple(shape), (n.shape, shape)
            return n
        torch.randn = fake
        return self

    def __exit__(self, *a):
        torch.randn = self._orig


def build_ref_mingtok(cfg_dict, seed):
    from mingtok.modeling_mingtok import MingTok, MingTokConfig
    m = MingTok(MingTokConfig(**cfg_dict)).eval()
    shapes = fill_from_synth(m, seed)
    mine = C.mingtok_param_shapes(C.MingTokConfig(**cfg_dict))
    assert mine == shapes, set(mine.items()) ^ set(shapes.items())
    return m


def gen_mingtok():
    seed = 11
    m = build_ref_mingtok(TINY_MINGTOK, seed)
    g = torch.Generator().manual_seed(1234)
    img = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1          # native resolution
    img2 = torch.rand(1, 3, 128, 128, generator=g) * 2 - 1       # pos-embed interpolation path
    with torch.no_grad():
        enc = m.low_level_encoder
        tok = enc.prepare_tokens(img)
        tok2 = enc.prepare_tokens(img2)
        blk0 = enc.blocks[0][0](tok)
        latent_raw = enc(img)
        out = m.forward(img)
        out2 = m.forward(img2)
        recon = m.forward_enc_dec(img)
        recon2 = m.forward_enc_dec(img2)
        pix_in = m.sem_to_pix(out["x_norm_patchtokens"])
        # cached causal decode, one token at a time (CausalAttention eager, attention.py:138-163)
        lat_norm = out2["latent"][:, :5]
        cache = ref_shim.make_legacy_cache()
        steps = []
        for i in range(lat_norm.shape[1]):
            r = m.forward_feature_decoder(lat_norm[:, i:i + 1], past_key_values=cache)
            cache = r["past_key_values"]
            steps.append(r["x_norm_patchtokens"])
        dec_steps = torch.cat(steps, dim=1)
        full = m.forward_feature_decoder_wo_cache(lat_norm * m.scaling_factor + m.mean)["x_norm"]
    assert torch.allclose(dec_steps, full, atol=1e-5), (dec_steps - full).abs().max()
    save("mingtok_tiny", config=TINY_MINGTOK, seed=seed, checksum=checksum(m), img=img, img2=img2,
         enc_tokens=tok, enc_tokens2=tok2, enc_block0=blk0, latent_raw=latent_raw,
         latent=out["latent"], sem=out["x_norm_patchtokens"], latent2=out2["latent"],
         sem2=out2["x_norm_patchtokens"], recon=recon, recon2=recon2, sem_to_pix=pix_in,
         dec_latent_norm=lat_norm, dec_steps=dec_steps)


def build_ref_llm(llm_dict, rf_dict, seed, latent_dim=32):
    import modeling_bailing_moe as mbm
    from configuration_bailing_moe import BailingMoeConfig
    cfg = BailingMoeConfig(**llm_dict, _attn_implementation="eager")
    cfg.rope_scaling = None          # transformers 5.x rewrites None -> dict (SURVEY §8c shim item 3)
    cfg._attn_implementation = "eager"
    cfg.image_start_token = llm_dict["image_start_token"]
    m = mbm.BailingMoeForCausalLM(cfg)
    if rf_dict is not None:
        m.setup_vishead_diffloss(**rf_dict, hidden_size=cfg.hidden_size, image_emb_dim_for_gen=latent_dim)
    m = m.eval()
    shapes = fill_from_synth(m, seed)
    mine = C.llm_param_shapes(C.BailingMoeConfig(**llm_dict), rf_dict, latent_dim)
    assert mine == shapes, sorted(set(mine.items()) ^ set(shapes.items()))[:10]
    return m, cfg


def gen_rf():
    import diff_loss_rf_swiglu as rf
    seed = 12
    for tag, steps in (("rf_tiny", 4), ("rf_tiny16", 16)):
        head = rf.RectifiedFlowLoss(target_channels=32, z_channels=64, depth=2, width=64,
                                    num_sampling_steps=str(steps), mlp_mult=4).eval()
        shapes = {}
        new = {}
        for k, v in head.state_dict().items():
            new[k] = synth_tensor("diffloss." + k, tuple(v.shape), seed)
            shapes["diffloss." + k] = tuple(v.shape)
        head.load_state_dict(new)
        assert shapes == C.rf_param_shapes(64, 2, 64, 32, 4)
        g = torch.Generator().manual_seed(99)
        z3 = torch.randn(3, 64, generator=g)
        x3 = torch.randn(3, 32, generator=g)
        t3 = torch.tensor([0.75, 0.75, 0.75])
        noise = torch.randn(4, 32, generator=g)
        with torch.no_grad():
            v3 = head.net(x3, t3, z3)
            temb = head.net.time_embed(torch.tensor([1000.0, 937.5, 62.5]))
            with ReplayRandn(noise):
                s3 = head.sample(z3, temperature=1.0, text_cfg=3.0, image_cfg=1.1)
                s2 = head.sample(z3[:2], temperature=0.9, text_cfg=3.0, image_cfg=1.1)
                s1 = head.sample(z3[:1], temperature=1.0, text_cfg=1.0, image_cfg=1.0)
        save(tag, seed=seed, steps=steps, checksum=checksum(head), z=z3, x=x3, t=t3, v=v3, temb=temb,
             noise=noise, sample3=s3, sample2=s2, sample2_temperature=0.9, sample1=s1)


def gen_llm():
    seed = 13
    m, cfg = build_ref_llm(TINY_LLM, TINY_RF, seed)
    g = torch.Generator().manual_seed(7)
    T = 12
    ids = torch.randint(0, 400, (1, T), generator=g)
    ids[0, 3:8] = TINY_LLM["image_patch_token"]
    image_mask = ids == TINY_LLM["image_patch_token"]
    emb = m.model.word_embeddings(ids).detach()
    emb = emb + 0.1 * torch.randn(emb.shape, generator=g) * image_mask.unsqueeze(-1)
    am = torch.ones(1, T, dtype=torch.long)
    with torch.no_grad():
        cache = ref_shim.make_legacy_cache()
        out = m.model(inputs_embeds=emb, attention_mask=am, past_key_values=cache, use_cache=True,
                      image_mask=image_mask)
        hidden = out.last_hidden_state
        logits = m.compute_logit(hidden[:, -1:]).float()
        k0 = cache.key_cache[0].clone()
        v1 = cache.value_cache[1].clone()
        # single-layer pieces
        lyr = m.model.layers[0]
        x_norm = lyr.input_layernorm(emb)
        moe_in = torch.randn(1, 5, cfg.hidden_size, generator=g)
        imask5 = torch.tensor([[True, False, True, False, False]])
        moe_out, (router_logits, topk_idx) = lyr.mlp(moe_in, imask5, None)
        gate_idx, gate_w, _ = lyr.mlp.gate(moe_in)
        igate_idx, igate_w, _ = lyr.mlp.image_gate(moe_in)
        # decode: 3 CFG rows sharing the prefilled cache, different masks -> different positions
        rows = 3
        for li in range(len(cache.key_cache)):
            cache.key_cache[li] = cache.key_cache[li].repeat(rows, 1, 1, 1)
            cache.value_cache[li] = cache.value_cache[li].repeat(rows, 1, 1, 1)
        am3 = torch.ones(rows, T + 1, dtype=torch.long)
        am3[1, 2:9] = 0          # uncond: hole in the middle
        am3[2, 2:3] = 0
        am3[2, 8:9] = 0          # text-uncond: keeps the image span
        dec_hidden = []
        dec_in = []
        for step in range(3):
            x = torch.randn(rows, 1, cfg.hidden_size, generator=g)
            pos = (am3.cumsum(-1) - 1)[:, -1:]
            o = m.model(inputs_embeds=x, attention_mask=am3, position_ids=pos, past_key_values=cache,
                        use_cache=True)
            dec_in.append(x)
            dec_hidden.append(o.last_hidden_state)
            am3 = torch.cat([am3, torch.ones(rows, 1, dtype=torch.long)], dim=1)
        z = m.vis_head(dec_hidden[-1][:, -1:]).reshape(rows, -1)
    save("llm_tiny", config=TINY_LLM, rf_config=TINY_RF, seed=seed, checksum=checksum(m), ids=ids,
         image_mask=image_mask, emb=emb, hidden=hidden, logits=logits, k0=k0, v1=v1, x_norm=x_norm,
         moe_in=moe_in, moe_image_mask=imask5, moe_out=moe_out, moe_topk_idx=topk_idx.reshape(5, -1),
         gate_idx=gate_idx, gate_w=gate_w, igate_idx=igate_idx, igate_w=igate_w,
         dec_mask0=torch.ones(rows, T + 1, dtype=torch.long) * 0 + am3[:, :T + 1],
         dec_in=torch.stack(dec_in), dec_hidden=torch.stack(dec_hidden), vis_z=z)


def gen_rope3d():
    """The 3D rotary branch (rope_scaling.type == "3D"): BailingMoe3DRotaryEmbedding + apply_multimodal_rotary_pos_emb
    on random q / k with DIFFERENT t / h / w position streams, and with equal streams (== Legacy)."""
    import modeling_bailing_moe as mbm
    g = torch.Generator().manual_seed(21)
    B, T, nq, nkv, hd = 2, 5, 4, 2, 128
    q = torch.randn(B, nq, T, hd, generator=g)
    k = torch.randn(B, nkv, T, hd, generator=g)
    pos3 = torch.stack([torch.randint(0, 40, (B, T), generator=g) for _ in range(3)])      # [3,B,T]
    rot = mbm.BailingMoe3DRotaryEmbedding(hd, max_position_embeddings=64, base=600000.0)
    with torch.no_grad():
        cos, sin = rot(k, position_ids=pos3)
        q3, k3 = mbm.apply_multimodal_rotary_pos_emb(q, k, cos, sin)
        same = pos3[:1].expand(3, -1, -1)
        cos_s, sin_s = rot(k, position_ids=same)
        qs, ks = mbm.apply_multimodal_rotary_pos_emb(q, k, cos_s, sin_s)
        legacy = mbm.BailingMoeRotaryEmbeddingLegacy(hd, base=600000.0)
        cl, sl = legacy(k, seq_len=64)
        ql, kl = mbm.apply_rotary_pos_emb(q, k, cl, sl, pos3[0])
    assert torch.equal(qs, ql) and torch.equal(ks, kl)          # equal streams are bit-identical to Legacy
    save("rope3d", q=q, k=k, pos3=pos3, q3=q3, k3=k3, q_same=qs, k_same=ks, base=600000.0)


def gen_genimg():
    """End-to-end BailingMoeForCausalLM.generate_image on tiny configs (3-row and 2-row CFG)."""
    import torch.nn as nn
    seed = 14
    m, cfg = build_ref_llm(TINY_LLM, TINY_RF, seed)
    tok = build_ref_mingtok(TINY_MINGTOK, seed)
    lp_shapes = C.linear_proj_param_shapes(128, cfg.hidden_size, 2)
    linear_proj = nn.Sequential(nn.Linear(128, cfg.hidden_size), nn.GELU(), nn.Linear(cfg.hidden_size, cfg.hidden_size))
    linear_proj.load_state_dict({k[len("linear_proj."):]: synth_tensor(k, s, seed) for k, s in lp_shapes.items()})

    def latent_to_sem(latent, past_key_values=None):
        if past_key_values is None:
            past_key_values = ref_shim.make_legacy_cache()
        return tok.forward_feature_decoder(latent, past_key_values=past_key_values)

    g = torch.Generator().manual_seed(21)
    T = 10
    ids = torch.randint(0, 400, (1, T), generator=g)
    noises = torch.randn(cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    am = torch.ones(1, T + 1, dtype=torch.long)
    uncond = torch.ones(1, T + 1, dtype=torch.long)
    uncond[0, 2:8] = 0
    tuncond = torch.ones(1, T + 1, dtype=torch.long)
    tuncond[0, 2:4] = 0
    tuncond[0, 7:8] = 0
    res = {}
    for tag, tu in (("rows3", tuncond), ("rows2", uncond.clone())):
        with torch.no_grad():
            cache = ref_shim.make_legacy_cache()
            m.model(input_ids=ids, attention_mask=am[:, :T], past_key_values=cache, use_cache=True)
            start = m.model.word_embeddings(torch.tensor([[cfg.image_start_token]]))
            with ReplayRandn(noises):
                image, model_out, am_out = m.generate_image(
                    input_embeds=start, past_key_values=cache, attention_mask=am.clone(),
                    uncond_attention_mask=uncond.clone(), text_uncond_attention_mask=tu.clone(),
                    latent_to_sem_func=latent_to_sem, linear_proj=linear_proj,
                    sem_to_pix_func=tok.forward_pixel_decoder)
            logits = m.compute_logit(model_out[0][0:1]).float()
        res[tag + "_image"] = image
        res[tag + "_last_hidden"] = model_out[0]
        res[tag + "_mask_out"] = am_out
        res[tag + "_logits"] = logits
        res[tag + "_cache_len"] = np.array(cache.key_cache[0].shape[2])
        res[tag + "_k0"] = cache.key_cache[0]
        res[tag + "_tuncond"] = tu
    save("genimg_tiny", llm_config=TINY_LLM, rf_config=TINY_RF, mingtok_config=TINY_MINGTOK, seed=seed,
         ids=ids, noises=noises, mask=am, uncond=uncond, checksum=checksum(m) + checksum(tok), **res)


class _FakeTok:
    """Deterministic stand-in tokenizer used to drive the REFERENCE BailingMMProcessor methods
    (the real tokenizer.json is not shipped): same contract as ming_univision_amd.processing.SpecialTokenTokenizer."""

    def __init__(self):
        from ming_univision_amd.processing import SpecialTokenTokenizer
        self._t = SpecialTokenTokenizer()
        self.chat_template = None
        self.init_kwargs = {}

    def __call__(self, text, **kw):
        return self._t(text)

    def encode(self, text, add_special_tokens=False):
        return self._t.encode(text)

    def convert_tokens_to_ids(self, t):
        return self._t.convert_tokens_to_ids(t)


def gen_processor():
    """Reference BailingMMProcessor.apply_chat_template / _expand_image_tokens / tokenize
    (processing_bailingmm.py:282-464) driven with a stand-in tokenizer."""
    import types
    import processing_bailingmm as pb
    fake = types.SimpleNamespace(tokenizer=_FakeTok())
    fake._find_all_subsequences = lambda seq, sub: pb.BailingMMProcessor._find_all_subsequences(fake, seq, sub)
    fake.apply_system_template = lambda text: pb.USER_PREFIX
    convs = {
        "t2i": [{"role": "HUMAN", "content": [{"type": "text", "text": "Please draw a red cube."}]}],
        "edit": [{"role": "HUMAN", "content": [{"type": "image", "image": "a.png"}, {"type": "text", "text": "make it blue"}]}],
        "multi": [{"role": "HUMAN", "content": [{"type": "text", "text": "hi"}]},
                  {"role": "ASSISTANT", "content": [{"type": "text", "text": "hello there"}]},
                  {"role": "HUMAN", "content": [{"type": "image", "image": "b.png"}, {"type": "text", "text": "what is this?"}]}],
    }
    out = {}
    for name, conv in convs.items():
        text = pb.BailingMMProcessor.apply_chat_template(fake, conv, add_generation_prompt=True)
        n_img = text.count("<IMAGE>")
        if n_img:
            grid = torch.tensor([[1, 4, 4]] * n_img)
            text = pb.BailingMMProcessor._expand_image_tokens(fake, [text], grid)[0]
        enc = pb.BailingMMProcessor.tokenize(fake, [text])
        out[name] = dict(text=text, input_ids=enc["input_ids"][0].tolist(),
                         uncond=enc["uncond_attention_mask"][0].tolist(),
                         text_uncond=enc["text_uncond_attention_mask"][0].tolist())
    # an incomplete dialogue (no ASSISTANT tag after the last HUMAN tag)
    text = pb.USER_PREFIX + "<image><imagePatch><imagePatch></image>\ndescribe"
    enc = pb.BailingMMProcessor.tokenize(fake, [text])
    out["no_assistant"] = dict(text=text, input_ids=enc["input_ids"][0].tolist(),
                               uncond=enc["uncond_attention_mask"][0].tolist(),
                               text_uncond=enc["text_uncond_attention_mask"][0].tolist())
    with open(os.path.join(GOLDEN, "processor.json"), "w") as f:
        json.dump(out, f)
    print("wrote", os.path.join(GOLDEN, "processor.json"))


def gen_multiround():
    """The multi-round state machine captured from the REFERENCE's own
    MingUniVisionForConditionalGeneration.generate (modeling_bailingmm.py:206-301) -> BailingMoeForCausalLM.generate / forward's
    `<image>` branch (modeling_bailing_moe.py:1769-1796) / prepare_inputs_for_generation (:1968-2080), driven by the installed
    transformers' greedy loop through ref_shim.cache_position_452 (the 4.52.4 `cache_position` contract; nothing of the reference is
    patched except I/O: MingTok.from_pretrained returns the tiny seeded tokenizer instead of reading ./models/MingTok-Vision, and
    tensor_to_pil — which hard-codes .cuda() and writes a PNG — hands the image tensor back).

    Conversation, the shape of test_infer_unified.py's editing block, under PAST_MODE = KEEP and = DROP:
      round 0  image (64 x 64 -> 4 <imagePatch> tokens) + instruction, 3 distinct CFG masks, `<image>` forced as the first new token
               (a LogitsProcessor: the reference forwards generate kwargs to HF), then two greedy text tokens;
      round 1  text-only instruction on the carried cache / masks, `<image>` forced again, two greedy tokens;
      round 2  text-only, nothing forced: four greedy text tokens on top of two generated images.
    Stored per mode and round: input ids and the three masks handed in, the returned sequences, the three carried masks and the cache
    length afterwards, the generated image (all CFG rows), every prepare_inputs_for_generation call's bookkeeping
    (ids seen, cache length, mask length out, new ids, first position id), and layer 0's K cache after the last round."""
    import torch.nn as nn  # noqa: F401
    from transformers import LogitsProcessor, LogitsProcessorList
    ref_shim.install_multimodal()
    import modeling_bailing_moe as mbm
    import modeling_bailingmm as mm
    from configuration_bailing_moe import BailingMoeConfig
    from configuration_bailingmm import MingUniVisionConfig
    from mingtok.modeling_mingtok import MingTok
    from ming_univision_amd.processing import cfg_attention_masks
    seed = 14
    llm_dict = dict(TINY_LLM, image_patch_token=498, eos_token_id=1)
    ROLE, ROLE_E, HUMAN, ASSIST, IMG, IMG_E, PATCH = 490, 491, 300, 301, 496, 497, 498
    tok = build_ref_mingtok(TINY_MINGTOK, seed)
    orig_fp, orig_pil = MingTok.from_pretrained, mbm.tensor_to_pil
    images = []

    class _NoFile:
        def save(self, *a, **k):
            pass
    MingTok.from_pretrained = classmethod(lambda cls, *a, **k: tok)
    mbm.tensor_to_pil = lambda t: (images.append(t.detach().float().clone()), _NoFile())[1]

    class Force(LogitsProcessor):
        def __init__(self, at, token):
            self.at, self.token = at, token

        def __call__(self, input_ids, scores):
            if input_ids.shape[1] == self.at:
                scores = torch.full_like(scores, float("-inf"))
                scores[:, self.token] = 0.0
            return scores
    g = torch.Generator().manual_seed(31)
    px = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
    noises = torch.randn(16, 32, generator=g)
    rounds_in = [
        dict(ids=[ROLE, HUMAN, ROLE_E, IMG, PATCH, PATCH, PATCH, PATCH, IMG_E, 21, 22, 23, 24, ROLE, ASSIST, ROLE_E], px=True, force=True, n_new=3),
        dict(ids=[ROLE, HUMAN, ROLE_E, 31, 32, 33, ROLE, ASSIST, ROLE_E], px=False, force=True, n_new=3),
        dict(ids=[ROLE, HUMAN, ROLE_E, 41, 42, ROLE, ASSIST, ROLE_E], px=False, force=False, n_new=4),
    ]
    res = {}
    old_mode = os.environ.get("PAST_MODE")
    try:
        for mode in ("KEEP", "DROP"):
            os.environ["PAST_MODE"] = mode
            llm = BailingMoeConfig(**llm_dict, _attn_implementation="eager")
            llm.rope_scaling = None
            llm._attn_implementation = "eager"
            llm.image_start_token = llm_dict["image_start_token"]
            cfg = MingUniVisionConfig(mlp_depth=2, llm_config=llm, vishead_diffloss_config=dict(TINY_RF))
            model = mm.MingUniVisionForConditionalGeneration(cfg).eval()
            shapes = fill_from_synth(model.model, seed)
            assert shapes == C.llm_param_shapes(C.BailingMoeConfig(**llm_dict), TINY_RF, 32)
            lp_shapes = C.linear_proj_param_shapes(128, llm.hidden_size, 2)
            model.linear_proj.load_state_dict({k[len("linear_proj."):]: synth_tensor(k, s_, seed) for k, s_ in lp_shapes.items()})
            trace = []
            start = ref_shim.cache_position_452(model.model, trace)
            cache = ref_shim.make_legacy_cache()
            with torch.no_grad(), ReplayRandn(noises) as rr:
                for r, spec in enumerate(rounds_in):
                    ids = torch.tensor([spec["ids"]])
                    unc, tunc = cfg_attention_masks(spec["ids"], [ROLE, HUMAN, ROLE_E], [ROLE, ASSIST, ROLE_E], {IMG, IMG_E, PATCH})
                    am, unc, tunc = torch.ones_like(ids), torch.tensor([unc]), torch.tensor([tunc])
                    n_img, n_tr, n_noise = len(images), len(trace), rr.i
                    start()
                    kw = dict(logits_processor=LogitsProcessorList([Force(ids.shape[1], llm_dict["image_start_token"])])) if spec["force"] else {}
                    seq = model.generate(input_ids=ids, attention_mask=am, uncond_attention_mask=unc, text_uncond_attention_mask=tunc,
                                         pixel_values=px if spec["px"] else None, past_key_values=cache if r == 0 else None,
                                         max_new_tokens=spec["n_new"], use_cache=True, do_sample=False, pad_token_id=0, eos_token_id=None,
                                         output_image_prefix=os.path.join(GOLDEN, "_never_written"), **kw)
                    t = f"{mode}_r{r}_"
                    res[t + "ids"], res[t + "unc"], res[t + "tunc"] = ids, unc, tunc
                    res[t + "seq"] = seq
                    res[t + "past_am"] = model.past_attention_mask
                    res[t + "past_unc"] = model.past_uncond_attention_mask
                    res[t + "past_tunc"] = model.past_text_uncond_attention_mask
                    res[t + "cache_len"] = np.array(model.past_key_values.get_seq_length())
                    res[t + "trace"] = np.array(trace[n_tr:], dtype=np.int64).reshape(-1, 5)
                    res[t + "noise0"] = np.array(n_noise)
                    if len(images) > n_img:
                        assert len(images) == n_img + 1
                        res[t + "image"] = images[-1]
            res[mode + "_k0"] = model.past_key_values.key_cache[0]
            res[mode + "_checksum"] = np.array(checksum(model.model))
    finally:
        MingTok.from_pretrained, mbm.tensor_to_pil = orig_fp, orig_pil
        if old_mode is None:
            os.environ.pop("PAST_MODE", None)
        else:
            os.environ["PAST_MODE"] = old_mode
    save("multiround_tiny", llm_config=llm_dict, rf_config=TINY_RF, mingtok_config=TINY_MINGTOK, seed=seed, pixel_values=px, noises=noises,
         special_ids=dict(ROLE=ROLE, ROLE_E=ROLE_E, HUMAN=HUMAN, ASSIST=ASSIST, IMG=IMG, IMG_E=IMG_E, PATCH=PATCH),
         rounds=[dict(force=s_["force"], n_new=s_["n_new"], px=s_["px"]) for s_ in rounds_in], **res)


def main():
    ref_shim.install()
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    gen_mingtok()
    gen_rf()
    gen_llm()
    gen_genimg()
    gen_processor()
    gen_rope3d()
    gen_multiround()


if __name__ == "__main__":
    main()
