"""Parity at the benchmarked model's REAL size (VERDICT r4 "weak" #1): the exact `ming_univision_16b_a3b()` decoder — 28 layers,
H = 2048, 16 q / 4 KV heads x 128, 64 routed experts top-6 + 2 shared, I = 1408, weights drawn by ming_univision_amd/synth.py's rule
and NOT rescaled — against the fp32 oracle, which walks the 32 GB stack layer by layer on the host (`tests/util.DecoderBackedSD`:
the oracle's state dict pulls each layer's tensors from the GPU decoder on first use and drops them when the next layer is asked
for; the expert-sorted form of `oracle/bailing_ref.moe_block` keeps the 1 536-row case to seconds per layer).

  * one decode step (modeling_bailing_moe.py:1391-1540) at 2 rows (chain route), 130 and 1 536 rows (wide route; the bench's row
    count) on a pre-filled random KV arena, every row its own cache sequence with its own length, holey key masks on a third of
    them: hidden states and the K / V lines all 28 layers append;
  * prefill(12 tokens) + `generate_image` (:1623-1673, :1844-1965) of 2 visual tokens at 2 CFG rows through the full RF head
    (w = 3072, depth 12, 16 Euler steps) and the full MingTok semantic decoder.

Rows are judged one by one (per-row max-norm).  A top-6-of-64 router is discontinuous: a row whose 6th and 7th logits are closer
than the path's rounding legitimately lands on another expert and is a different sample from there on; such NEAR-TIE rows are
identified on the oracle (smallest 6th - 7th logit gap over the 28 layers, relative to the row's largest |logit|, under 1e-3) and
reported separately with the count of rows that did flip; rows with clear routing are held to 1e-3 at the 90th percentile.

Round 6: that classification left 54 % of the 1 536 rows without an arithmetic bar.  The step now ALSO runs against
the oracle with TEACHER-FORCED routing — the HIP path's own expert choice of every layer (mn_llm_route_capture) handed to the oracle's
gate, which keeps its own softmax scores at those experts — and EVERY row is held to 1e-3; router flips are a reported count.
"""
import pytest
import torch
import torch.nn.functional as F

from ming_univision_amd import configuration as C
from ming_univision_amd.synth import synth_state_dict, synth_tensor
from tests.util import DecoderBackedSD, rel_err, row_errs

pytestmark = pytest.mark.gpu

TOL = 1e-3
T_MAX = 24
LENS = (4, 7, 11, 15, 18, 21)          # cache lengths the rows draw from (the oracle's caches are dense: one group per length)


@pytest.fixture(scope="module")
def fullsize():
    from oracle import bailing_ref
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
    cfg.num_image_tokens_for_gen = 2
    assert (cfg.num_hidden_layers, cfg.hidden_size, cfg.num_experts, cfg.num_shared_experts, cfg.moe_intermediate_size) == (28, 2048, 64, 2, 1408)
    torch.set_num_threads(min(64, max(torch.get_num_threads(), 16)))
    dec = BailingMoeDecoder.synthetic(cfg, torch.device("cuda"), seed=21, with_vocab=False, t_max=T_MAX, n_seq=2)
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in cfg.to_dict().items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    return cfg, dec, ocfg


def _oracle_step_streamed(sd, ocfg, x, lens, km, pos, kv, forced=None):
    """One decode step of the oracle with the LAYER loop outermost (each layer's weights are pulled once): rows are batch entries
    grouped by cache length.  -> (hidden [M, H] after the final norm, smallest relative 6th-7th logit gap per row, new K / V lines
    [L, M, nkv, hd] x 2).
    forced [L, M, k] (expert ids): TEACHER-FORCED routing — every layer's gate takes these experts (the HIP path's own choice, captured
    through mn_llm_route_capture) with the ORACLE's softmax scores at them, renormalised (BailingMoeGate.forward's arithmetic,
    modeling_bailing_moe.py:505-520, minus the arg-top-k); `margin` then holds, per row, in how many layers the forced set differs from
    the oracle's own top-k of the same state (router flips)."""
    from oracle import bailing_ref
    M, H = x.shape
    L, k_top = ocfg.num_hidden_layers, ocfg.num_experts_per_tok
    groups = [(n, (lens == n).nonzero().flatten()) for n in sorted(set(lens.tolist()))]
    state = {n: x[idx].unsqueeze(1).float() for n, idx in groups}
    m4 = {n: bailing_ref.build_4d_mask(km[idx, :n + 1].long(), 1, n) for n, idx in groups}
    margin = torch.full((M,), float("inf")) if forced is None else torch.zeros(M)
    new_k = torch.empty(L, M, ocfg.num_key_value_heads, ocfg.head_dim)
    new_v = torch.empty_like(new_k)
    orig_gate = bailing_ref.gate
    cur = {}

    def gate_rec(x2d, w, c):
        lg = F.linear(x2d, w).float()
        if forced is not None:
            ti = forced[cur["li"]][cur["idx"]].long()
            scores = lg.softmax(dim=-1, dtype=torch.float32)
            tw = scores.gather(1, ti)
            if k_top > 1 and c.norm_topk_prob:
                tw = tw / tw.sum(dim=-1, keepdim=True)
            own = torch.topk(scores, k=k_top, dim=-1).indices
            margin[cur["idx"]] += (ti.sort(-1).values != own.sort(-1).values).any(-1).float()
            return ti, tw, lg
        srt = lg.sort(dim=-1, descending=True).values
        margin[cur["idx"]] = torch.minimum(margin[cur["idx"]], (srt[:, k_top - 1] - srt[:, k_top]) / lg.abs().amax(-1))
        return orig_gate(x2d, w, c)
    bailing_ref.gate = gate_rec
    try:
        for li in range(L):
            for n, idx in groups:
                cur["idx"], cur["li"] = idx, li
                kvl = dict(k=kv[li, idx, 0, :, :n].clone(), v=kv[li, idx, 1, :, :n].clone())
                state[n] = bailing_ref.decoder_layer(state[n], sd, li, ocfg, m4[n], pos[idx].unsqueeze(1), kvl)
                new_k[li, idx], new_v[li, idx] = kvl["k"][:, :, n], kvl["v"][:, :, n]
    finally:
        bailing_ref.gate = orig_gate
    ref = torch.empty(M, H)
    for n, idx in groups:
        ref[idx] = bailing_ref.rmsnorm(state[n], sd["model.norm.weight"], ocfg.rms_norm_eps)[:, 0]
    return ref, margin, new_k, new_v


@pytest.mark.parametrize("M", [2, 130, 1536])
def test_fullsize_step_vs_streamed_oracle(fullsize, M):
    cfg, dec0, ocfg = fullsize
    dec = dec0.view(t_max=T_MAX, n_seq=M)
    assert dec.max_rows() == 2048
    g = torch.Generator().manual_seed(100 + M)
    L, nkv, hd, H = cfg.num_hidden_layers, cfg.num_key_value_heads, cfg.head_dim, cfg.hidden_size
    lens = torch.tensor(LENS)[torch.randint(0, len(LENS), (M,), generator=g)]
    kv = torch.randn(L, M, 2, nkv, T_MAX, hd, generator=g) * 0.5
    dec.kv_cache.copy_(kv.cuda())
    x = torch.randn(M, H, generator=g) * 0.5
    km = torch.ones(M, T_MAX, dtype=torch.uint8)
    for m in range(0, M, 3):                                               # CFG-style holes in the prefix, last key always attended
        km[m, 1:max(2, int(lens[m]) - 1)] = 0
    pos = torch.stack([(km[m, :int(lens[m]) + 1].long().cumsum(0) - 1)[-1] for m in range(M)])   # modeling_bailing_moe.py:1905-1907
    slot = lens.to(torch.int32).cuda()
    from ming_univision_amd._lib import check, lib, ptr
    n_slot = cfg.num_experts_per_tok + dec.n_shared
    routes = torch.full((L, M, n_slot), -1, dtype=torch.int32, device="cuda")
    check(lib().mn_llm_route_capture(ptr(routes)), "mn_llm_route_capture")
    try:
        out = dec.step(x.cuda(), torch.arange(M, dtype=torch.int32).cuda(), slot, pos.to(torch.int32).cuda(), slot + 1, km.cuda())
        torch.cuda.synchronize()
    finally:
        check(lib().mn_llm_route_capture(None), "mn_llm_route_capture")
    routes = routes.cpu()
    k_top = cfg.num_experts_per_tok
    assert int(routes.min()) >= 0 and int(routes[:, :, :k_top].max()) < cfg.num_experts          # every layer's router was captured
    assert bool((routes[:, :, k_top:] == cfg.num_experts + torch.arange(dec.n_shared)).all())      # the shared pseudo-experts follow
    sd = DecoderBackedSD(dec)
    # (1) TEACHER-FORCED routing: the oracle takes the HIP path's expert choice in every layer, so a near-tie
    # that lands on the other side is no longer a different sample — EVERY row is held to 1e-3 on arithmetic; flips are counted
    ref_f, n_flip, fk, fv = _oracle_step_streamed(sd, ocfg, x, lens, km, pos, kv, forced=routes[:, :, :k_top])
    rows_f = row_errs(out, ref_f)
    print("FULL SIZE, %d rows, teacher-forced routing: ALL rows vs the oracle on the HIP path's experts: median %.2e, 99 %% %.2e, max %.2e; "
          "rows whose routing differs from the oracle's own top-k in some layer: %d (layer-row flips: %d)" % (
              M, float(rows_f.median()), float(rows_f.quantile(0.99)), float(rows_f.max()), int((n_flip > 0).sum()), int(n_flip.sum())))
    assert float(rows_f.max()) < TOL, float(rows_f.max())
    assert int((n_flip > 0).sum()) <= max(1, M // 20)
    # (2) free routing: the oracle's own top-k, rows classified by their smallest 6th - 7th logit gap
    ref, margin, new_k, new_v = _oracle_step_streamed(sd, ocfg, x, lens, km, pos, kv)
    per_row = row_errs(out, ref)
    stable = margin >= 1e-3
    near = ~stable
    mx = lambda t: float(t.max()) if t.numel() else 0.0
    q = lambda t, f: float(t.quantile(f)) if t.numel() else 0.0
    clear = per_row[stable]
    flips = int((per_row[near] > 1e-2).sum())
    print("FULL SIZE (28 layers x H 2048 x 64+2 experts), %d rows vs the streamed oracle: global %.2e | %d clear-routing rows: median %.2e, "
          "90 %% %.2e, 98 %% %.2e, max %.2e | %d near-tie rows: median %.2e, max %.2e, %d above 1e-3, %d flipped (> 1e-2)" % (
              M, rel_err(out, ref), int(stable.sum()), q(clear, 0.5), q(clear, 0.9), q(clear, 0.98), mx(clear), int(near.sum()),
              q(per_row[near], 0.5), mx(per_row[near]), int((per_row[near] > TOL).sum()), flips))
    assert int(stable.sum()) >= 0.4 * M or M == 2
    if int(stable.sum()):
        assert q(clear, 0.5) < TOL / 3 and q(clear, 0.9) < TOL and mx(clear) < 1e-2, (q(clear, 0.5), q(clear, 0.9), mx(clear))
    assert mx(per_row[near]) < 0.5 and int((per_row[near] > TOL).sum()) <= max(1, M // 10)
    # the K / V lines this step appended, all 28 layers, per row (the row's worst layer)
    kc = dec.kv_cache.cpu()
    rows = torch.arange(M)
    kv_err = torch.zeros(M, dtype=torch.float64)
    for which, r in ((0, new_k), (1, new_v)):
        got = kc[:, :, which][:, rows, :, lens].permute(1, 0, 2, 3)        # [L, M, nkv, hd]: row m's line at slot lens[m]
        d = (got.double() - r.double()).abs().amax(dim=(0, 2, 3)) / r.double().abs().amax(dim=(0, 2, 3))
        kv_err = torch.maximum(kv_err, d)
    print("   appended K / V lines of all 28 layers, clear-routing rows: median %.2e, 90 %% %.2e, max %.2e" % (
        q(kv_err[stable], 0.5), q(kv_err[stable], 0.9), mx(kv_err[stable])))
    if int(stable.sum()):
        assert q(kv_err[stable], 0.9) < TOL and mx(kv_err[stable]) < 1e-2
    assert torch.equal(kc[:, 0, :, :, int(lens[0]) + 1:], kv[:, 0, :, :, int(lens[0]) + 1:])       # nothing beyond the appended slot


def test_fullsize_prefill_and_generate_vs_streamed_oracle(fullsize):
    """prefill(12) + 2 visual tokens at 2 CFG rows: the full decoder, the full RF head and the full semantic decoder against
    `oracle/bailing_ref.generate_image` on the layer-streamed state dict."""
    from oracle import bailing_ref, mingtok_ref
    from ming_univision_amd.bailing_moe import generate_image
    from ming_univision_amd.mingtok import MingTok
    from ming_univision_amd.rf_head import RectifiedFlowHead
    cfg, dec0, ocfg = fullsize
    seed = 21
    rf_cfg = dict(C.DEFAULT_VISHEAD_DIFFLOSS)
    shapes = C.llm_param_shapes(cfg, rf_cfg, 32)
    rsd = {k: synth_tensor(k, s_, seed, "cuda", torch.bfloat16) for k, s_ in shapes.items() if k.startswith("vis_head") or k.startswith("diffloss")}
    rf = RectifiedFlowHead(rsd, cfg.hidden_size, rf_cfg)
    lsd = synth_state_dict(C.linear_proj_param_shapes(1024, cfg.hidden_size, 2), seed)
    dl = {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in lsd.items()}
    tok = MingTok(C.MingTokConfig(), device="cuda", seed=seed,
                  linear_proj=[(dl["linear_proj.0.weight"], dl["linear_proj.0.bias"]), (dl["linear_proj.2.weight"], dl["linear_proj.2.bias"])])
    tsd = {k: v.float().cpu() for k, v in tok.sd.items()}
    dec = dec0.view(t_max=32, n_seq=3)
    sd = DecoderBackedSD(dec, extra={k: v.float().cpu() for k, v in rsd.items()})
    g = torch.Generator().manual_seed(3)
    T = 12
    emb = torch.randn(1, T, cfg.hidden_size, generator=g) * 0.02          # word-embedding-sized rows (no vocabulary tensors on this decoder)
    start = torch.randn(1, 1, cfg.hidden_size, generator=g) * 0.02
    noises = torch.randn(cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    am = torch.ones(1, T + 1, dtype=torch.long)
    un = am.clone(); un[0, 2:T - 2] = 0
    # oracle (three walks over the 28 layers after the prefill's)
    kvs = bailing_ref.new_kv(ocfg)
    h_pre = bailing_ref.model_forward(emb, sd, ocfg, torch.ones(1, T, dtype=torch.long), None, kvs)
    caches = mingtok_ref.semdec_new_cache(tsd)
    ref = bailing_ref.generate_image(
        start, kvs, am, un, un.clone(), sd, ocfg, noises,
        latent_to_sem=lambda lat: mingtok_ref.mingtok_feature_decoder_step(lat, tsd, caches),
        linear_proj=lambda s: bailing_ref.linear_proj(s, lsd), sem_to_pix=lambda s: None,
        steps=int(rf_cfg["num_sampling_steps"]))
    assert ref["last_hidden"].shape[0] == 2
    # HIP path
    h_dev = dec.prefill(emb[0].cuda(), seq=0, past=0)
    out = generate_image(dec, rf, tok, start[0].cuda(), T, am, un, un.clone(), noises.cuda(), decode_pixels=False)
    e_pre, e_pre_rows = rel_err(h_dev, h_pre[0]), float(row_errs(h_dev, h_pre[0]).max())
    figs = {k: (rel_err(a, b), float(row_errs(a, b).max())) for k, (a, b) in dict(
        latents=(out["latents"], ref["latents"][:, 0]), sem=(out["sem"], ref["sem"][0]), hidden=(out["last_hidden"], ref["last_hidden"][:, 0])).items()}
    print("FULL SIZE prefill(12): global %.2e, worst row %.2e | generate_image, 2 CFG rows x 2 visual tokens (global / worst row): "
          "latents %.2e / %.2e, sem %.2e / %.2e, hidden %.2e / %.2e" % (e_pre, e_pre_rows, *figs["latents"], *figs["sem"], *figs["hidden"]))
    assert e_pre < TOL and e_pre_rows < 3 * TOL
    for k, (e, er) in figs.items():
        assert e < TOL and er < 3 * TOL, (k, e, er)
