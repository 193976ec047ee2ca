"""Forced ties in the MoE router ("add one forced-tie case documenting `torch.topk` order as unspecified").

`BailingMoeGate.forward` (modeling_bailing_moe.py:505-520) picks the experts with `torch.topk(scores, k)`, whose order among EQUAL
scores is unspecified (it differs between torch's CPU and CUDA kernels and between sizes).  With fp32 logits of real weights an exact
tie has measure zero; it only happens when two experts carry the same gate row.  The device's rule, on every route, is:

    among equal scores the LOWEST expert id wins   (expert e = lane e; a round picks `__ffsll(__ballot(cur == wave_max(cur)))`)

— decode_ops.hip `moe_topk_kernel` / `moe_router_row_kernel` (the rule is written down there), engine.hip `moe_route_group_kernel`,
prefill_ops.hip `moe_topk_logits_kernel`, the wide route's top-k in wide_llm.inl.  These tests pin that rule: experts 3 and 40 (and 11, 52) share a gate row,
so every row ties exactly, and the tie sits at the top-k boundary for many rows; the oracle is run with a stable descending sort
(= lowest id first among equals), and the device must agree on all three routes (2 rows: chain; 40 rows: unfused <= 64-row route;
130 rows: wide route), through the prefill's top-k over precomputed logits, and through the stand-alone router entry point."""
import pytest
import torch
import torch.nn.functional as F

from ming_univision_amd import configuration as C
from tests.util import llm_sd, rel_err, row_errs

pytestmark = pytest.mark.gpu

TIES = ((3, 40), (11, 52))


def _lowest_id_gate(x2d, w, cfg):
    """oracle/bailing_ref.gate with the tie rule made explicit: stable descending sort = lowest expert id first among equals."""
    logits = F.linear(x2d, w)
    scores = logits.softmax(dim=-1, dtype=torch.float32)
    order = torch.sort(scores, dim=-1, descending=True, stable=True).indices[:, :cfg.num_experts_per_tok]
    tw = scores.gather(-1, order)
    if cfg.num_experts_per_tok > 1 and cfg.norm_topk_prob:
        tw = tw / tw.sum(dim=-1, keepdim=True)
    return order, tw, logits


def _tied_model(hidden=256):
    from oracle import bailing_ref
    d = C.BailingMoeConfig.ming_univision_16b_a3b().to_dict()
    d.pop("model_type", None)
    d.update(hidden_size=hidden, moe_intermediate_size=64, vocab_size=512, num_hidden_layers=2, num_image_tokens_for_gen=3,
             image_start_token=500, pad_token_id=0)
    rf_cfg = dict(diffloss_w=64, diffloss_d=2, num_sampling_steps="4", gen_method="flow_matching_swiglu-4")
    sd = llm_sd(d, rf_cfg, 17)
    for li in range(d["num_hidden_layers"]):
        for name in ("gate", "image_gate"):
            g = sd[f"model.layers.{li}.mlp.{name}.weight"]
            g.copy_((g * 6.0).to(torch.bfloat16).float())      # spread the scores so that the tied pair is not always far from the cut
            for a, b in TIES:
                g[b] = g[a]
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in d.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    return d, sd, ocfg


@pytest.mark.parametrize("M", [2, 40, 130])
def test_decoder_step_with_tied_gate_rows_follows_lowest_expert_id(M):
    from oracle import bailing_ref
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    d, sd, ocfg = _tied_model()
    cfg = C.BailingMoeConfig(**d)
    t_max = 8
    dec = BailingMoeDecoder.from_state_dict(cfg, {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}, t_max=t_max, n_seq=M)
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, cfg.hidden_size, generator=g) * 0.5
    kv = torch.randn(cfg.num_hidden_layers, M, 2, cfg.num_key_value_heads, t_max, cfg.head_dim, generator=g) * 0.5
    dec.kv_cache.copy_(kv.cuda())
    n = 4
    slot = torch.full((M,), n, dtype=torch.int32, device="cuda")
    image_mask = (torch.arange(M) % 3 == 1)
    out = dec.step(x.cuda(), torch.arange(M, dtype=torch.int32).cuda(), slot, slot, slot + 1, None, image_mask.to(torch.uint8).cuda())
    boundary = []
    orig = bailing_ref.gate

    def gate_rec(x2d, w, c):
        ti, tw, lg = _lowest_id_gate(x2d, w, c)
        srt = torch.sort(lg, dim=-1, descending=True, stable=True)
        k = c.num_experts_per_tok
        boundary.append(srt.values[:, k - 1] == srt.values[:, k])           # an exact tie ACROSS the cut: the rule decides the set
        return ti, tw, lg
    bailing_ref.gate = gate_rec
    try:
        kvs = [dict(k=kv[l, :, 0, :, :n].clone(), v=kv[l, :, 1, :, :n].clone()) for l in range(cfg.num_hidden_layers)]
        ref = bailing_ref.model_forward(x[:, None], sd, ocfg, torch.ones(M, n + 1, dtype=torch.long),
                                        torch.full((M, 1), n, dtype=torch.long), kvs, image_mask=image_mask[:, None])[:, 0]
    finally:
        bailing_ref.gate = orig
    cut = torch.stack(boundary).any(0)
    errs = row_errs(out, ref)
    print("tied gate rows, %d rows: %d rows with an exact tie across the top-k cut; worst row %.2e (rows with such a tie: %.2e)" % (
        M, int(cut.sum()), float(errs.max()), float(errs[cut].max()) if int(cut.sum()) else 0.0))
    assert M < 40 or int(cut.sum()) >= 1                     # the case is exercised (the other expert of the pair would give O(1e-1))
    assert float(errs.max()) < 1e-3, errs


def test_one_row_steps_with_tied_gate_rows_follow_lowest_expert_id():
    """ONE row, hidden 512, no image gate: the step takes the one-launch router + gate/up kernel (moe_gate_up.hip), whose every
    workgroup routes the row itself — the same rule must hold there.  24 one-row steps; at least one has its tie across the top-k cut."""
    from oracle import bailing_ref
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    d, sd, ocfg = _tied_model(hidden=512)
    cfg = C.BailingMoeConfig(**d)
    t_max, n = 8, 4
    dec = BailingMoeDecoder.from_state_dict(cfg, {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}, t_max=t_max, n_seq=1)
    cuts, worst = 0, 0.0
    orig = bailing_ref.gate
    for seed in range(24):
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.randn(1, cfg.hidden_size, generator=g) * 0.5
        kv = torch.randn(cfg.num_hidden_layers, 1, 2, cfg.num_key_value_heads, t_max, cfg.head_dim, generator=g) * 0.5
        dec.kv_cache.copy_(kv.cuda())
        slot = torch.full((1,), n, dtype=torch.int32, device="cuda")
        out = dec.step(x.cuda(), torch.zeros(1, dtype=torch.int32).cuda(), slot, slot, slot + 1, distinct_sequences=True)
        boundary = []

        def gate_rec(x2d, w, c):
            ti, tw, lg = _lowest_id_gate(x2d, w, c)
            srt = torch.sort(lg, dim=-1, descending=True, stable=True)
            k = c.num_experts_per_tok
            boundary.append(srt.values[:, k - 1] == srt.values[:, k])
            return ti, tw, lg
        bailing_ref.gate = gate_rec
        try:
            kvs = [dict(k=kv[l, :, 0, :, :n].clone(), v=kv[l, :, 1, :, :n].clone()) for l in range(cfg.num_hidden_layers)]
            ref = bailing_ref.model_forward(x[:, None], sd, ocfg, torch.ones(1, n + 1, dtype=torch.long),
                                            torch.full((1, 1), n, dtype=torch.long), kvs)[:, 0]
        finally:
            bailing_ref.gate = orig
        cuts += int(torch.stack(boundary).any())
        e = float(row_errs(out, ref).max())
        worst = max(worst, e)
        assert e < 1e-3, (seed, e)
    print("one-row steps with tied gate rows: %d of 24 with an exact tie across the top-k cut; worst %.2e" % (cuts, worst))
    assert cuts >= 1


def test_router_entry_points_with_exactly_tied_logits():
    """mn_moe_router (gate GEMV + top-k) with duplicated gate rows and mn_moe_topk_logits (the prefill's top-k over precomputed logits)
    with hand-made equal logits: lowest id among equals, equal weights for the tied pair."""
    from ming_univision_amd import ops
    from ming_univision_amd._lib import check, current_stream, lib, ptr
    M, H, E, k, S = 6, 256, 64, 6, 2
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, H, generator=g)
    nw = torch.ones(H, dtype=torch.bfloat16)
    gw = (torch.randn(E, H, generator=g) * 0.3).to(torch.bfloat16)
    for a, b in TIES:
        gw[b] = gw[a]
    mask = torch.zeros(M, dtype=torch.uint8)
    xn, idx, w = ops.moe_router(x.cuda(), nw.cuda(), 1e-5, gw.cuda(), gw.cuda(), mask.cuda(), k, True, S)
    xr = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5)
    cfgk = type("c", (), dict(num_experts_per_tok=k, norm_topk_prob=True))
    ti, tw, _ = _lowest_id_gate(xr, gw.float(), cfgk)
    assert idx[:, :k].cpu().tolist() == ti.tolist()
    assert rel_err(w[:, :k], tw) < 1e-5
    for m in range(M):                                       # whenever the higher id of a pair is picked, the lower one is too, just before it
        row = idx[m, :k].cpu().tolist()
        for a, b in TIES:
            assert b not in row or (a in row and row.index(a) + 1 == row.index(b))
    # hand-made logits: ids 5, 9, 20 tie for ranks 5-7 of a top-6 -> 5 and 9 are taken, 20 is not; a full-row tie takes ids 0..5
    T = 3
    lg = torch.full((T, E), -4.0)
    lg[0, [1, 2, 3, 4]] = torch.tensor([3.0, 2.5, 2.0, 1.5]); lg[0, [5, 9, 20]] = 1.0
    lg[1] = 0.25
    lg[2, 63] = 2.0; lg[2, [62, 0]] = 1.0; lg[2, [7, 8, 30, 31]] = 0.5
    ti_d = torch.empty(T, k + S, dtype=torch.int32, device="cuda")
    tw_d = torch.empty(T, k + S, dtype=torch.float32, device="cuda")
    lgd = lg.cuda()
    check(lib().mn_moe_topk_logits(ptr(lgd), None, None, T, E, k, 1, S, ptr(ti_d), ptr(tw_d), current_stream()), "mn_moe_topk_logits")
    got = ti_d[:, :k].cpu().tolist()
    assert got[0] == [1, 2, 3, 4, 5, 9] and got[1] == [0, 1, 2, 3, 4, 5] and got[2] == [63, 0, 62, 7, 8, 30], got
    assert ti_d[:, k:].cpu().tolist() == [[E, E + 1]] * T
    assert rel_err(tw_d[1, :k], torch.full((k,), 1.0 / k)) < 1e-6
