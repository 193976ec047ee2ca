"""Depth and footprint parity (VERDICT r3 "weak" #1, #2): what the 2-layer / 32-slot full-width cases cannot see.

  * the full DEPTH: a 28-layer stack (16 q / 4 KV heads x 128, 64 experts top-6 + 2 shared — the 16B-A3B structure at hidden 256 so
    that the fp32 CPU oracle finishes in seconds) against the oracle at 2 rows (fused chain route), 130 and 1536 rows (wide route,
    the bench's row count), every row its own sequence with its own cache length, holey key masks on a third of them;
  * the bench's exact KV ARENA [28, 1536, 2, 4, 304, 128] fp32 (53.6 GB; layer 27 starts 12.9 G floats in: offsets beyond 2^32):
    steps that touch sequences 0 and 1535 give bit-identical results and cache lines to the same steps on a 2-sequence arena;
  * BASELINE configs[2] at the reference's own understanding shape: a 1024 x 1024 image = 1 024 `<imagePatch>` tokens through the
    full-size MingTok (32 x 32 interpolated pos-embed), a 1 058-token prompt through `prefill_wide`, greedy tokens, full width.
"""
import pytest
import torch

from ming_univision_amd import configuration as C
from ming_univision_amd.synth import synth_state_dict
from tests.util import llm_sd, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-3


def _dev(sd):
    return {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}


def _narrow28():
    d = C.BailingMoeConfig.ming_univision_16b_a3b().to_dict()
    d.pop("model_type", None)
    d.update(hidden_size=256, moe_intermediate_size=64, vocab_size=512, num_hidden_layers=28, num_image_tokens_for_gen=3,
             image_start_token=500, pad_token_id=0)
    return d


@pytest.fixture(scope="module")
def deep():
    from oracle import bailing_ref
    d = _narrow28()
    rf_cfg = dict(diffloss_w=64, diffloss_d=2, num_sampling_steps="4", gen_method="flow_matching_swiglu-4")
    torch.set_num_threads(min(32, torch.get_num_threads()))
    sd = llm_sd(d, rf_cfg, 11)
    # N(0, 0.006) weights at hidden 256 leave the residual stream almost untouched; scale the projections so that 28 layers matter
    for k in sd:
        if k.endswith("weight") and sd[k].dim() == 2 and "word_embeddings" not in k and "lm_head" not in k:
            sd[k] = (sd[k] * 4.0).to(torch.bfloat16).float()
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in d.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    return d, sd, ocfg


@pytest.mark.parametrize("M", [2, 130, 1536])
def test_depth28_step_vs_oracle(deep, M):
    from oracle import bailing_ref
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    d, sd, ocfg = deep
    cfg = C.BailingMoeConfig(**d)
    t_max = 24
    dec = BailingMoeDecoder.from_state_dict(cfg, _dev(sd), t_max=t_max, n_seq=M)
    assert cfg.num_hidden_layers == 28 and dec.max_rows() == 2048
    g = torch.Generator().manual_seed(M)
    L, nkv, hd, H = 28, cfg.num_key_value_heads, cfg.head_dim, cfg.hidden_size
    lens = torch.randint(3, t_max - 1, (M,), generator=g)                 # tokens already cached per row (= this step's slot)
    kv = torch.randn(L, M, 2, nkv, t_max, hd, generator=g) * 0.5
    dec.kv_cache.copy_(kv.cuda())
    x = torch.randn(M, H, generator=g) * 0.5
    km = torch.ones(M, t_max, dtype=torch.uint8)
    for m in range(0, M, 3):                                               # CFG-style holes in the prefix, last key always attended
        km[m, 1:max(2, int(lens[m]) - 1)] = 0
    pos = torch.stack([(km[m, :int(lens[m]) + 1].long().cumsum(0) - 1)[-1] for m in range(M)])   # modeling_bailing_moe.py:1905-1907
    slot = lens.to(torch.int32).cuda()
    from ming_univision_amd._lib import check, lib, ptr
    routes = torch.full((L, M, cfg.num_experts_per_tok + dec.n_shared), -1, dtype=torch.int32, device="cuda")
    check(lib().mn_llm_route_capture(ptr(routes)), "mn_llm_route_capture")      # every layer's expert choice of the HIP path
    try:
        out = dec.step(x.cuda(), torch.arange(M, dtype=torch.int32).cuda(), slot, pos.to(torch.int32).cuda(), slot + 1, km.cuda())
        torch.cuda.synchronize()
    finally:
        check(lib().mn_llm_route_capture(None), "mn_llm_route_capture")
    routes = routes.cpu()[:, :, :cfg.num_experts_per_tok].long()
    assert int(routes.min()) >= 0 and int(routes.max()) < cfg.num_experts
    # oracle: rows are batch entries with different cache lengths -> group by length (the oracle's caches are dense tensors).
    # A 28-layer top-6-of-64 router makes 43 008 discontinuous decisions at 1536 rows: a row whose 6th and 7th logits are closer
    # than the path's rounding (2^-17-class operands, amplified layer by layer) legitimately lands on another expert and is then a
    # different sample (measured: 44 of 1536 rows, 2.9 %).  Such NEAR-TIE rows are identified on the oracle itself — the smallest
    # (6th - 7th) logit gap over the 28 layers, relative to the row's largest |logit|, is under 1e-3 — and held to a loose bound;
    # every other row is held to 1e-3, and they must be a large part of the batch.
    import torch.nn.functional as F
    k_top = cfg.num_experts_per_tok
    ref = torch.empty(M, H)
    margin = torch.full((M,), float("inf"))
    new_k = {}
    orig_gate = bailing_ref.gate
    for n in sorted(set(lens.tolist())):
        idx = (lens == n).nonzero().flatten()
        gaps = []

        def gate_rec(x2d, w, c, gaps=gaps):
            lg = F.linear(x2d, w).float()
            srt = lg.sort(dim=-1, descending=True).values
            gaps.append((srt[:, k_top - 1] - srt[:, k_top]) / lg.abs().amax(-1))
            return orig_gate(x2d, w, c)
        bailing_ref.gate = gate_rec
        try:
            kvs = [dict(k=kv[l, idx, 0, :, :n].clone(), v=kv[l, idx, 1, :, :n].clone()) for l in range(L)]
            h = bailing_ref.model_forward(x[idx].unsqueeze(1), sd, ocfg, km[idx, :n + 1].long(), pos[idx].unsqueeze(1), kvs)
        finally:
            bailing_ref.gate = orig_gate
        assert len(gaps) == L
        ref[idx] = h[:, 0]
        margin[idx] = torch.stack(gaps).amin(0)
        new_k[n] = (idx, torch.stack([kvs[l]["k"][:, :, n] for l in range(L)]), torch.stack([kvs[l]["v"][:, :, n] for l in range(L)]))
    # TEACHER-FORCED routing: the oracle takes the HIP path's experts in every layer (its own softmax scores at
    # them, renormalised — BailingMoeGate.forward minus the arg-top-k), so near-tie rows are no longer different samples: ALL rows are
    # held to the bars the clear-routing rows meet below (28 layers of x 4 weights amplify rounding ~25x: 1e-3 at the 90th percentile,
    # one decade above for the tail); flips (forced set != the oracle's own top-k of the same state) are counted.
    ref_f = torch.empty(M, H)
    flips = torch.zeros(M)
    for n in sorted(set(lens.tolist())):
        idx = (lens == n).nonzero().flatten()
        calls = []

        def gate_forced(x2d, w, c, calls=calls, idx=idx):
            lg = F.linear(x2d, w).float()
            scores = lg.softmax(dim=-1, dtype=torch.float32)
            ti = routes[len(calls)][idx]
            calls.append(1)
            tw = scores.gather(1, ti)
            tw = tw / tw.sum(dim=-1, keepdim=True) if (k_top > 1 and c.norm_topk_prob) else tw
            flips[idx] += (ti.sort(-1).values != torch.topk(scores, k=k_top, dim=-1).indices.sort(-1).values).any(-1).float()
            return ti, tw, lg
        bailing_ref.gate = gate_forced
        try:
            kvs = [dict(k=kv[l, idx, 0, :, :n].clone(), v=kv[l, idx, 1, :, :n].clone()) for l in range(L)]
            ref_f[idx] = bailing_ref.model_forward(x[idx].unsqueeze(1), sd, ocfg, km[idx, :n + 1].long(), pos[idx].unsqueeze(1), kvs)[:, 0]
        finally:
            bailing_ref.gate = orig_gate
        assert len(calls) == L
    rows_f = (out.cpu().double() - ref_f.double()).abs().amax(1) / ref_f.double().abs().amax(1)
    print("28 layers, %d rows, teacher-forced routing: ALL rows: median %.2e, 90 %% %.2e, max %.2e; rows that flipped in some layer: %d" % (
        M, float(rows_f.median()), float(rows_f.quantile(0.9)), float(rows_f.max()), int((flips > 0).sum())))
    assert float(rows_f.median()) < TOL / 3 and float(rows_f.quantile(0.9)) < TOL and float(rows_f.max()) < 1e-2
    per_row = (out.cpu().double() - ref.double()).abs().amax(1) / ref.double().abs().amax(1)
    stable = margin >= 1e-3
    unstable = ~stable
    mx = lambda t: float(t.max()) if t.numel() else 0.0
    q = lambda t, f: float(t.quantile(f)) if t.numel() else 0.0
    clear = per_row[stable]
    print("28 layers, %d rows vs oracle: %d rows with clear routing: median %.2e, 90 %% %.2e, 98 %% %.2e, max %.2e | %d near-tie rows: max %.2e, "
          "%d of them above 1e-3" % (M, int(stable.sum()), q(clear, 0.5), q(clear, 0.9), q(clear, 0.98), mx(clear), int(unstable.sum()),
                                     mx(per_row[unstable]), int((per_row[unstable] > TOL).sum())))
    # 28 layers amplify the per-operation rounding (2^-17-class operands) ~25x at the median and, on a few rows, 10x more (measured:
    # median 1.9e-4, max 2.2e-3 at 1536 rows): the bar is 1e-3 for the bulk of the well-conditioned rows and one decade above for
    # their tail; a wrong layer stride / cache line / expert would put whole rows at O(1).
    assert int(stable.sum()) >= 0.4 * M
    assert q(clear, 0.5) < TOL / 3 and q(clear, 0.9) < TOL and mx(clear) < 1e-2, (q(clear, 0.5), q(clear, 0.9), mx(clear))
    assert mx(per_row[unstable]) < 0.3 and int((per_row[unstable] > TOL).sum()) <= max(1, M // 10)
    # the K / V rows this step appended, layer by layer (layer 27 included), against the oracle's
    kc = dec.kv_cache.cpu()
    kv_err = torch.zeros(M, dtype=torch.float64)
    for n, (idx, k_ref, v_ref) in new_k.items():
        for which, r in ((0, k_ref), (1, v_ref)):                         # r [L, rows, nkv, hd]; the row's worst layer
            d = (kc[:, idx, which, :, n].double() - r.double()).abs().amax(dim=(0, 2, 3)) / r.double().abs().amax(dim=(0, 2, 3))
            kv_err[idx] = torch.maximum(kv_err[idx], d)
    print("appended K / V lines of all 28 layers, clear-routing rows: median %.2e, 90 %% %.2e, max %.2e" % (
        q(kv_err[stable], 0.5), q(kv_err[stable], 0.9), mx(kv_err[stable])))
    assert q(kv_err[stable], 0.9) < TOL and mx(kv_err[stable]) < 1e-2
    untouched = kc[:, 0, :, :, int(lens[0]) + 1:]                           # nothing beyond the appended slot was written
    assert torch.equal(untouched, kv[:, 0, :, :, int(lens[0]) + 1:])


def test_bench_arena_addressing_layer27_sequence1535():
    """The bench's arena geometry: steps on rows that live in sequences {0, 1535} (2-row chain route) and in 130 sequences spread over
    [0, 65) + [1471, 1536) (wide route) must be BIT-IDENTICAL — hidden states and every appended K / V line of all 28 layers — to the
    same steps on a compact arena holding the same cache contents: only addresses differ (layer stride 478 M floats, layer 27 at
    12.9 G floats = 51.6 GB)."""
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    d = _narrow28()
    cfg = C.BailingMoeConfig(**d)
    t_max, n_seq = 304, 1536
    big = BailingMoeDecoder.synthetic(cfg, "cuda", seed=3, with_vocab=False, t_max=t_max, n_seq=n_seq)
    assert big.kv_cache.numel() * 4 > 53e9 and big.kv_cache[27].data_ptr() - big.kv_cache.data_ptr() > (1 << 35)
    g = torch.Generator().manual_seed(9)
    for M, seqs in ((2, [0, 1535]), (130, list(range(65)) + list(range(1471, 1536)))):
        small = big.view(t_max=t_max, n_seq=M)
        past = 170                                                        # the bench's mid-generation cache length
        fill = torch.randn(28, M, 2, cfg.num_key_value_heads, past, cfg.head_dim, generator=g).cuda() * 0.5
        small.kv_cache[:, :, :, :, :past] = fill
        sq = torch.tensor(seqs, dtype=torch.long, device="cuda")
        big.kv_cache[:, sq, :, :, :past] = fill
        x = (torch.randn(M, cfg.hidden_size, generator=g) * 0.5).cuda()
        km = torch.ones(M, t_max, dtype=torch.uint8)
        km[1::2, 2:38] = 0
        km = km.cuda()
        slot = torch.full((M,), past, dtype=torch.int32, device="cuda")
        pos = torch.where(torch.arange(M, device="cuda") % 2 == 1, slot - 36, slot).to(torch.int32)
        outs = []
        for dec, rs in ((small, torch.arange(M, dtype=torch.int32, device="cuda")), (big, sq.to(torch.int32))):
            for step in range(2):                                         # two consecutive tokens: the second reads the first's lines
                h = dec.step(x if step == 0 else outs_prev, rs, slot + step, pos + step, slot + step + 1, km)
                outs_prev = h
            outs.append(h)
        assert torch.equal(outs[0], outs[1]), (M, float((outs[0] - outs[1]).abs().max()))
        assert torch.equal(small.kv_cache[:, :, :, :, past:past + 2], big.kv_cache[:, sq, :, :, past:past + 2])
        assert float(small.kv_cache[27, :, :, :, past + 1].abs().max()) > 0                  # layer 27 really appended something
        # neighbours of the touched sequences stay untouched (zero)
        assert float(big.kv_cache[:, 700:710].abs().max()) == 0.0 and float(big.kv_cache[27, 1534 if M == 2 else 1470].abs().max()) == 0.0
        del small


def test_image_to_text_1024_full_width_vs_oracle():
    """BASELINE configs[2] as the reference runs it (processing_bailingmm.py:175, 256-268: understanding images are resized to
    1024 x 1024 = 32 x 32 patches): 1 024 `<imagePatch>` tokens, bicubic-interpolated position embedding
    (vision_transformer.py:183-215), 1 058-token prompt through the wide route's prefill with the image-gate rows, full width
    (2 LLM layers), fp32-class regime, against the oracle <= 1e-3; greedy continuation equal."""
    from oracle import bailing_ref
    from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration
    from tests.test_gpu_parity_r3 import _oracle_image_to_text
    seed = 23
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
    cfg = C.MingUniVisionConfig(mlp_depth=2, llm_config=d, vishead_diffloss_config=rf_cfg)
    ckpt = {"model." + k: v for k, v in sd.items()}
    ckpt.update({"vision." + k: v for k, v in tsd.items()})
    ckpt.update(lsd)
    model = MingUniVisionForConditionalGeneration(cfg, state_dict=ckpt, seed=seed, t_max=1088)
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in d.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    gen = torch.Generator().manual_seed(8)
    px = torch.rand(1, 3, 1024, 1024, generator=gen) * 2 - 1
    ids = torch.cat([torch.randint(2, 900, (1, 12), generator=gen), torch.full((1, 1024), 1001), torch.randint(2, 900, (1, 22), generator=gen)], 1)
    assert ids.shape[1] == 1058
    n_new = 3
    ref_toks, ref_h, ref_img = _oracle_image_to_text(sd_r, lsd_r, tsd_r, ocfg, ids, px, 1001, n_new)
    feats = model.extract_image_feature(px.cuda())
    assert feats.shape == (1024, d["hidden_size"])
    e_feat = rel_err(feats, ref_img)
    emb, im, _ = model.prompt_wrap_navit(ids, feats)
    assert int(im.sum()) == 1024
    h32 = model.model.prefill_wide(emb, seq=0, past=0, image_mask=im.reshape(-1))[-1:]
    e_h = rel_err(h32, ref_h)
    print("image(1024^2) -> text, full width: features %.2e, prompt hidden (1058 tokens, wide prefill) %.2e" % (e_feat, e_h))
    assert e_feat < TOL and e_h < TOL
    seqs = model.generate(input_ids=ids, pixel_values=px, max_new_tokens=n_new)
    assert seqs[0, ids.shape[1]:].tolist() == ref_toks
