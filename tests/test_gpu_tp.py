"""Tensor / expert parallel decode path (BASELINE configs[4]) on ONE GPU: every rank's shard, KV arena, workspace and communicator
live in one process and advance segment by segment (ming_univision_amd.tp.TpSimGroup) — the same kernels, the same push / arrival-flag
protocol as one process per GPU; the cross-device memory path itself needs the 8-GPU node (unmeasured on hardware).
Parity: the sum over the shards must be the unsharded path (<= 2e-4: summation order) and the fp32 oracle (<= 1e-3)."""
import pytest
import torch

from ming_univision_amd import configuration as C
from ming_univision_amd.synth import synth_state_dict
from tests.util import llm_sd, load_golden, mingtok_sd, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-3


def _dev(sd):
    return {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}


@pytest.mark.parametrize("world", [2, 4, 8])
def test_allreduce_oneshot_phases(world):
    """mn_allreduce_oneshot: every rank pushes, then every rank reduces — rounds on one communicator (epoch parity, flags never reset),
    ragged row counts; the result is the fp32 sum over the ranks, bit-identical on every rank (same order of additions).  Above 16 rows
    the all-reduce is TWO-SHOT (round 5): reduce-scatter push, the owners' reduce + all-gather push (phase GATHER), consume — two
    epochs; one-shot and two-shot rounds alternate on the same communicator."""
    from ming_univision_amd.tp import TpCommunicator
    comms = TpCommunicator.simulated(world, rows_cap=40, width=512)
    g = torch.Generator().manual_seed(world)
    epoch = 0
    for rnd, (M, D) in enumerate([(3, 512), (40, 256), (1, 64), (17, 512), (40, 512), (16, 512)]):
        two = comms[0].allreduce_segments(M, D) == 2
        assert two == (M > 16 and D % (4 * world) == 0)
        xs = [torch.randn(M, D, generator=g).cuda() for _ in range(world)]
        for r in range(world):
            comms[r].all_reduce(xs[r], phase=TpCommunicator.PUSH)
        for r in range(world):
            comms[r].all_reduce(xs[r], phase=TpCommunicator.GATHER)          # (nothing to do for a one-shot round)
        outs = [comms[r].all_reduce(xs[r], phase=TpCommunicator.REDUCE) for r in range(world)]
        ref = torch.stack(xs).double().sum(0)
        epoch += 2 if two else 1
        for r in range(world):
            assert rel_err(outs[r], ref) < 1e-6, (rnd, r)
            assert torch.equal(outs[r], outs[0])                  # same order of additions on every rank: bit-identical
            assert comms[r].struct.epoch == epoch
            comms[r].check_err()
    # the same 40 rows through the two-shot (default threshold) and the one-shot form (threshold raised): the same sum to fp32 rounding
    # (the owners add the senders one by one, the one-shot consumer pairs its slabs)
    xs = [torch.randn(40, 512, generator=g).cuda() for _ in range(world)]
    res = []
    for lim in (0, 1 << 20):                                       # default threshold (two-shot at 40 rows) / never
        for c in comms:
            c.struct.two_shot_rows = lim
        for ph in (TpCommunicator.PUSH, TpCommunicator.GATHER):
            for r in range(world):
                comms[r].all_reduce(xs[r], phase=ph)
        res.append([comms[r].all_reduce(xs[r], phase=TpCommunicator.REDUCE) for r in range(world)][0])
    assert rel_err(res[0], res[1]) < 1e-6 and rel_err(res[0], torch.stack(xs).double().sum(0)) < 1e-6


def test_allreduce_oneshot_concurrent_ranks_really_wait():
    """Two ranks on two HIP streams, each enqueuing push + reduce in ONE call: rank 0's reduce is on the GPU before rank 1's push
    exists, so its rows spin on their arrival flags until the other stream delivers.  (A wait that never completes sets the
    communicator's error word after a bounded spin instead of hanging the GPU.)"""
    from ming_univision_amd.tp import TpCommunicator
    comms = TpCommunicator.simulated(2, rows_cap=8, width=256)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    g = torch.Generator().manual_seed(3)
    for _ in range(5):
        xs = [torch.randn(8, 256, generator=g).cuda() for _ in range(2)]
        torch.cuda.synchronize()
        outs = [None, None]
        for r in (0, 1):
            with torch.cuda.stream(streams[r]):
                outs[r] = comms[r].all_reduce(xs[r])
        torch.cuda.synchronize()
        for r in (0, 1):
            comms[r].check_err()
            assert torch.equal(outs[r], xs[0] + xs[1])
    # the two-shot form (40 rows): scatter push, owner reduce + all-gather, consume — all three enqueued per rank in ONE call, so the
    # owners' kernels and the consumers really wait on the other stream's pushes
    comms = TpCommunicator.simulated(2, rows_cap=40, width=256)
    assert comms[0].allreduce_segments(40, 256) == 2
    for _ in range(5):
        xs = [torch.randn(40, 256, generator=g).cuda() for _ in range(2)]
        torch.cuda.synchronize()
        outs = [None, None]
        for r in (0, 1):
            with torch.cuda.stream(streams[r]):
                outs[r] = comms[r].all_reduce(xs[r])
        torch.cuda.synchronize()
        for r in (0, 1):
            comms[r].check_err()
            assert torch.equal(outs[r], xs[0] + xs[1])


def test_allreduce_phase3_above_the_one_shot_limit_and_strided_out():
    """(1) phase = PUSH | REDUCE (3) was the documented whole all-reduce before the two-shot form existed: above 16 rows it
    must still complete (the owners' gather step is implied), not wait for an epoch nobody publishes; (2) the two-shot consumer clears
    `out` before adding the pieces — only the payload columns: a strided `out` keeps what the caller holds in columns [D, ldo) and
    nothing is written past the last row."""
    import ctypes as C
    from ming_univision_amd._lib import check, current_stream, lib, ptr
    from ming_univision_amd.tp import TpCommunicator
    comms = TpCommunicator.simulated(2, rows_cap=40, width=256)
    assert comms[0].allreduce_segments(40, 256) == 2
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(40, 256, generator=g).cuda() for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ldo = 256 + 64
    bufs = [torch.full((40 * ldo + 64,), 7.0, device="cuda") for _ in range(2)]        # 64 guard floats behind the last row
    torch.cuda.synchronize()
    for r in (0, 1):
        comms[r].struct.wait_ms = 2000
        with torch.cuda.stream(streams[r]):
            check(lib().mn_allreduce_oneshot(C.byref(comms[r].struct), ptr(xs[r]), 256, ptr(bufs[r]), ldo, 40, 256,
                                             TpCommunicator.PUSH | TpCommunicator.REDUCE, current_stream()), "mn_allreduce_oneshot")
    torch.cuda.synchronize()
    for r in (0, 1):
        comms[r].check_err()
        out = bufs[r][:40 * ldo].view(40, ldo)
        assert torch.equal(out[:, :256], xs[0] + xs[1])
        assert bool((out[:, 256:] == 7.0).all()) and bool((bufs[r][40 * ldo:] == 7.0).all())
        assert comms[r].struct.epoch == 2


def test_allreduce_wait_that_expires_poisons_the_rows_and_raises():
    """A sender that never pushes (dead peer / diverged launch order): the consumer's wall-time-bounded wait (200 ms here, 30 s by
    default) expires, the rows come out as NaN — never as a sum of stale slabs — and check_err() raises naming the sender."""
    from ming_univision_amd.tp import TpCommunicator
    comms = TpCommunicator.simulated(2, rows_cap=8, width=256)
    x = torch.ones(8, 256, device="cuda")
    for c in comms:                                                    # a completed round first: the inbox holds stale finite data
        c.all_reduce(x, phase=TpCommunicator.PUSH)
    outs = [c.all_reduce(x, phase=TpCommunicator.REDUCE) for c in comms]
    torch.cuda.synchronize()
    assert all(torch.equal(o, 2 * x) for o in outs)
    comms[0].struct.wait_ms = 200
    comms[0].all_reduce(3 * x, phase=TpCommunicator.PUSH)              # rank 1 never pushes round 2
    import time
    t0 = time.perf_counter()
    out = comms[0].all_reduce(3 * x, phase=TpCommunicator.REDUCE)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert 0.15 < dt < 5.0, dt
    assert bool(torch.isnan(out).all())
    with pytest.raises(RuntimeError, match="gave up waiting for sender 1"):
        comms[0].check_err()


def test_lmhead_argmax_on_nan_rows_returns_a_valid_id():
    """torch.argmax treats NaN as the maximum: a row of NaN logits (e.g. after a poisoned all-reduce) must still give an id inside
    the vocabulary (the next step gathers word_embeddings[id])."""
    from ming_univision_amd import ops
    from ming_univision_amd.tp import pick_best
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(1000, 256, generator=g) * 0.05).to(torch.bfloat16).cuda()
    h = torch.randn(3, 256, generator=g).cuda()
    h[1] = float("nan")
    idx, val = ops.lmhead_argmax(h, w, vocab_offset=5000)
    ref = (h.double() @ w.double().T)
    assert int(idx[0]) == 5000 + int(ref[0].argmax()) and int(idx[2]) == 5000 + int(ref[2].argmax())
    assert int(idx[1]) == 5000 and bool(torch.isnan(val[1]))           # all NaN: the first one, like torch.argmax
    best = pick_best(torch.tensor([[7, 3], [9, 1]]), torch.tensor([[float("nan"), 1.0], [float("nan"), 2.0]]))
    assert best.tolist() == [7, 1]


def test_ep_dispatch_and_combine_match_the_full_expert_sum():
    """mn_ep_dispatch / mn_ep_combine: 4 ranks x 2 experts; the tile list of a rank names only its experts, the combine over the
    ranks is the weighted un-permute of ALL picks."""
    import ctypes as Ct
    from ming_univision_amd._lib import check, current_stream, lib, ptr
    from ming_univision_amd.tp import TpCommunicator
    world, E, k, M, D = 4, 8, 3, 37, 128
    g = torch.Generator().manual_seed(0)
    ti = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(M)]).to(torch.int32).cuda()
    tw = torch.rand(M, k, generator=g).cuda()
    h = torch.randn(M, D, generator=g).cuda()
    comms = TpCommunicator.simulated(world, rows_cap=64, width=D)
    L = lib()
    # expert e's "MLP output" for sorted position p is a known function of (e, p): yg[p] = (e + 1) * base[p]
    base = torch.randn(M * k, D, generator=g).cuda()
    outs, P = [], M * k
    state = []
    for r in range(world):
        cnt = torch.empty(E, dtype=torch.int32, device="cuda"); off = torch.empty(E + 1, dtype=torch.int32, device="cuda")
        perm = torch.empty(P, dtype=torch.int32, device="cuda"); slot_of = torch.empty(P, dtype=torch.int32, device="cuda")
        tg = torch.full((P // 16 + E,), -1, dtype=torch.int32, device="cuda"); tm = torch.empty_like(tg)
        nt = torch.zeros(1, dtype=torch.int32, device="cuda")
        check(L.mn_ep_dispatch(ptr(ti), M, k, E, 2 * r, 2, ptr(cnt), ptr(off), ptr(perm), ptr(slot_of), 16, ptr(tg), ptr(tm), ptr(nt),
                               current_stream()), "mn_ep_dispatch")
        n = int(nt.item())
        assert set(tg[:n].tolist()) <= {2 * r, 2 * r + 1}
        assert n == sum(-(-int(cnt[e]) // 16) for e in (2 * r, 2 * r + 1))
        assert int(cnt.sum()) == P and sorted(slot_of.tolist()) == list(range(P))
        assert torch.equal(perm[slot_of.long()].reshape(M, k), torch.arange(M, device="cuda", dtype=torch.int32).unsqueeze(1).expand(M, k))
        exp_of_pos = torch.empty(P, dtype=torch.long, device="cuda")
        exp_of_pos[slot_of.long()] = ti.reshape(-1).long()
        yg = base * (exp_of_pos + 1).unsqueeze(1).float()
        local = (exp_of_pos >= 2 * r) & (exp_of_pos < 2 * r + 2)
        yg[~local] = float("nan")                                   # rows of other ranks' experts are never written: must not be read
        state.append((slot_of, yg))
    extra = [torch.randn(2, M, D, generator=g).cuda() for _ in range(world)]     # each rank's "shared slice" slabs
    # mn_ep_combine is push + reduce in one call; with all ranks in one process the local weighted sums are taken through one-rank
    # communicators and summed across ranks with the two phases of the public all-reduce
    parts = []
    zero = torch.zeros(M, D, device="cuda")
    for r in range(world):
        one = TpCommunicator.simulated(1, rows_cap=64, width=D)[0]
        part = torch.empty(M, D, device="cuda")
        slot_of, yg = state[r]
        check(L.mn_ep_combine(Ct.byref(one.struct), ptr(yg), ptr(slot_of), ptr(ti), ptr(tw), k, 2 * r, 2, ptr(extra[r]), 2, M * D,
                              ptr(zero), D, ptr(part), D, M, D, current_stream()), "mn_ep_combine")
        one.check_err()
        parts.append(part)
    for r in range(world):
        comms[r].all_reduce(parts[r], phase=TpCommunicator.PUSH)
    for r in range(world):                                              # (37 rows: the two-shot form — every owner reduces and publishes)
        comms[r].all_reduce(parts[r], phase=TpCommunicator.GATHER)
    total = comms[0].all_reduce(parts[0], phase=TpCommunicator.REDUCE) + h
    ref = h.double().cpu().clone()
    for r in range(world):
        ref += extra[r].double().cpu().sum(0)
    tic, twc, basec = ti.cpu(), tw.double().cpu(), base.double().cpu()
    for m in range(M):
        for s_ in range(k):
            e = int(tic[m, s_])
            pos = int(state[e // 2][0][m * k + s_])                 # the owner rank's sorted position of this pick
            ref[m] += twc[m, s_] * (e + 1) * basec[pos]
    assert torch.isfinite(total).all()
    assert rel_err(total, ref) < 1e-5


@pytest.mark.parametrize("world", [2, 4])
def test_tp_llm_step_tiny_matches_unsharded_and_reference(world):
    """Tiny decoder (4 q / 2 KV heads, 8 experts + 2 shared, multi-gate) split 2 and 4 ways: a 12-token prefill with image-gate rows,
    then 3-row CFG decode steps with holey masks — against the unsharded path, the reference's golden hidden states, and rank against
    rank (bit-identical: every rank sums the same slabs in the same order)."""
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    from ming_univision_amd.tp import TpSimGroup
    g = load_golden("llm_tiny")
    sd = llm_sd(g["config"], g["rf_config"], g["seed"])
    cfg = C.BailingMoeConfig(**g["config"])
    if cfg.num_key_value_heads % world and world % cfg.num_key_value_heads:
        pytest.skip("KV heads do not split")
    dec = BailingMoeDecoder.from_state_dict(cfg, _dev(sd), t_max=64, n_seq=3)
    grp = TpSimGroup(dec, None, world, rows_cap=64)
    emb = g["emb"][0].cuda()
    T = emb.shape[0]
    h_tp = grp.prefill(emb, seq=0, past=0, image_mask=g["image_mask"][0])
    h_full = dec.prefill(emb, seq=0, past=0, image_mask=g["image_mask"][0])
    grp.check_err()
    assert grp.greedy(h_tp).tolist() == dec.greedy(h_full).tolist()          # lm_head split over the vocabulary: same greedy ids
    assert rel_err(h_tp, g["hidden"][0]) < TOL
    assert rel_err(h_tp, h_full) < 2e-4
    for o in grp.last_rank_outputs[1:]:
        assert torch.equal(o, grp.last_rank_outputs[0])
    # every rank's KV arena holds its KV heads of the full arena
    for r, sh in enumerate(grp.shards):
        kv0 = (r * sh.plan["n_q"] * cfg.num_key_value_heads) // cfg.num_attention_heads
        assert rel_err(sh.kv_cache[:, 0, :, :, :T], dec.kv_cache[:, 0, :, kv0:kv0 + sh.plan["n_kv"], :T]) < 2e-4
    # CFG decode: replicate the prompt to 3 rows, then the golden decode inputs
    for s in (1, 2):
        grp.copy_sequence(0, s, T); dec.copy_sequence(0, s, T)
    from ming_univision_amd.bailing_moe import ImageGenState
    st = ImageGenState(grp, [g["dec_mask0"]], [T])
    for i in range(g["dec_in"].shape[0]):
        x = g["dec_in"][i][:, 0].cuda().contiguous()
        h1 = grp.step(x, st.row_seq, st.row_slot, st.row_pos, st.row_len, st.key_mask)
        h2 = dec.step(x, st.row_seq, st.row_slot, st.row_pos, st.row_len, st.key_mask)
        assert rel_err(h1, g["dec_hidden"][i][:, 0]) < TOL, i
        assert rel_err(h1, h2) < 2e-4
        st.advance()
    grp.check_err()


def test_tp_group_token_sampling_and_rf_sampling_are_two_methods(tmp_path):
    """TpSimGroup / TpRank defined `sample` twice, so the TOKEN sampler (hidden, uniforms, temperature, top_k,
    top_p) was shadowed by the RF sampler and `generate(do_sample=True)` on a TP decoder ran the RF head on the uniforms.  The RF
    sampler is `rf_sample` now (reached through `sampler()`); a sampled decode through a TP = 2 group behind the façade draws the
    same tokens as the unsharded model at the same generator seed, and the group's `sample` IS the unsharded token sampler."""
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    from ming_univision_amd.mingtok import MingTok
    from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration
    from ming_univision_amd.rf_head import RectifiedFlowHead
    from ming_univision_amd.tp import TpRank, TpSimGroup
    g = load_golden("genimg_tiny")
    llm_cfg = dict(g["llm_config"]); llm_cfg["eos_token_id"] = 1
    cfg = C.BailingMoeConfig(**llm_cfg)
    mcfg = C.MingUniVisionConfig(mlp_depth=2, llm_config=llm_cfg, vishead_diffloss_config=g["rf_config"], mingtok_config=g["mingtok_config"])
    dsd = _dev(llm_sd(g["llm_config"], g["rf_config"], g["seed"]))
    dl = _dev(synth_state_dict(C.linear_proj_param_shapes(128, 256, 2), g["seed"]))
    lp = [(dl["linear_proj.0.weight"], dl["linear_proj.0.bias"]), (dl["linear_proj.2.weight"], dl["linear_proj.2.bias"])]
    tok = MingTok(C.MingTokConfig(**g["mingtok_config"]), state_dict=mingtok_sd(g["mingtok_config"], g["seed"]), linear_proj=lp)
    rf = RectifiedFlowHead(dsd, cfg.hidden_size, g["rf_config"])
    dec = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=64, n_seq=3)
    grp = TpSimGroup(BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=64, n_seq=3), None, 2, rows_cap=64)   # (the tiny RF width does not split)
    for klass in (TpSimGroup, TpRank):
        assert klass.sample.__code__.co_varnames[:3] == ("self", "hidden", "u") and hasattr(klass, "rf_sample")
    hidden = torch.randn(3, cfg.hidden_size, generator=torch.Generator().manual_seed(1)).cuda()
    u = torch.tensor([0.11, 0.52, 0.93], device="cuda")
    assert grp.sample(hidden, u, 0.8, 20, 0.9).tolist() == dec.sample(hidden, u, 0.8, 20, 0.9).tolist()
    single = MingUniVisionForConditionalGeneration.from_parts(mcfg, tok, dec, rf, lp, seed=g["seed"])
    shard = MingUniVisionForConditionalGeneration.from_parts(mcfg, tok, grp, rf, lp, seed=g["seed"])
    ids = g["ids"]
    T = ids.shape[1]
    for temperature, top_k, top_p, seed in ((1.0, 50, 1.0, 1), (0.7, 10, 0.9, 2), (1.3, 0, 0.8, 3)):
        outs = []
        for model in (single, shard):
            model.reset_inner_state()
            gen = torch.Generator(device="cuda").manual_seed(seed)
            outs.append(model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=6, do_sample=True,
                                       temperature=temperature, top_k=top_k, top_p=top_p, generator=gen,
                                       output_image_prefix=str(tmp_path / "s"))[0, T:].tolist())
        assert outs[0] == outs[1], (temperature, top_k, top_p, outs)
    grp.check_err()


def test_tp_wide_rows_match_unsharded():
    """Above 64 rows the TP composites run on gemm256 (the wide route's kernels) instead of the weight-streaming ones: 70 rows of
    the tiny decoder (ragged cache lengths, holey masks, image-gate rows) on 2 ranks, and 66 rows of a small RF head on 4 ranks,
    against the unsharded path."""
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    from ming_univision_amd.rf_head import RectifiedFlowHead
    from ming_univision_amd.synth import synth_tensor
    from ming_univision_amd.tp import TpCommunicator, TpRfShard, TpSimGroup
    g = load_golden("llm_tiny")
    cfg = C.BailingMoeConfig(**g["config"])
    dec = BailingMoeDecoder.from_state_dict(cfg, _dev(llm_sd(g["config"], g["rf_config"], g["seed"])), t_max=48, n_seq=70)
    grp = TpSimGroup(dec, None, 2, rows_cap=128)
    gen = torch.Generator().manual_seed(0)
    M = 70
    kv0 = torch.randn(dec.kv_cache.shape, generator=gen).cuda() * 0.5
    dec.kv_cache.copy_(kv0)
    for r, sh in enumerate(grp.shards):
        k0 = (r * sh.plan["n_q"] * cfg.num_key_value_heads) // cfg.num_attention_heads
        sh.kv_cache.copy_(kv0[:, :, :, k0:k0 + sh.plan["n_kv"]])
    x = torch.randn(M, cfg.hidden_size, generator=gen).cuda()
    slot = torch.randint(5, 40, (M,), generator=gen).to(torch.int32).cuda()
    seq = torch.arange(M, dtype=torch.int32).cuda()
    km = (torch.rand(M, 48, generator=gen) > 0.2).to(torch.uint8)
    km[torch.arange(M), slot.cpu().long()] = 1
    km = km.cuda()
    im = (torch.arange(M) % 3 == 0).to(torch.uint8).cuda()
    h_tp = grp.step(x, seq, slot, slot, slot + 1, km, im)
    h_full = dec.step(x, seq, slot, slot, slot + 1, km, im)
    grp.check_err()
    assert rel_err(h_tp, h_full) < 2e-4
    for o in grp.last_rank_outputs[1:]:
        assert torch.equal(o, grp.last_rank_outputs[0])
    # RF head: w = 384, hidden 1024 -> 256 per rank, 66 rows (33 images x 2 CFG rows)
    rf_cfg = dict(diffloss_w=384, diffloss_d=2, num_sampling_steps="3", gen_method="flow_matching_swiglu-4")
    shapes = C.rf_param_shapes(384, 2, 384, 32, 4)
    shapes.update({"vis_head.0.weight": (384, 256), "vis_head.0.bias": (384,), "vis_head.1.weight": (384,), "vis_head.1.bias": (384,)})
    sd = {k: synth_tensor(k, s_, 3, "cuda", torch.bfloat16) for k, s_ in shapes.items()}
    rf = RectifiedFlowHead(sd, 256, rf_cfg)
    world = 4
    shards = [TpRfShard(rf, r, world) for r in range(world)]
    comms = TpCommunicator.simulated(world, rows_cap=128, width=rf.w)
    hidden = torch.randn(66, 256, generator=gen).cuda()
    noise = torch.randn(33, 32, generator=gen).cuda()
    ref = rf.sample(hidden, noise, n_images=33)
    outs = [torch.empty_like(ref) for _ in range(world)]
    assert shards[0].n_segments(comms[0], 66) == 2 * (shards[0].n_segments() - 1) + 1      # 66 rows: every all-reduce two-shot
    for seg in range(shards[0].n_segments(comms[0], 66)):
        for r in range(world):
            shards[r].sample_tp(comms[r], hidden, noise, out=outs[r], n_images=33, seg_begin=seg, seg_end=seg + 1)
    for r in range(world):
        comms[r].check_err()
    err = (outs[0] - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)
    assert float(err.max()) < 2e-4, float(err.max())


def test_tp_relayed_transport_matches_unsharded():
    """The RCCL fallback transport (TpCommunicator.relayed: pushes stay local, an all-gather over the process group delivers them
    between two segments) with a loop-back process group standing in for RCCL: two ranks of the tiny decoder, prefill + decode."""
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    from ming_univision_amd.tp import TpCommunicator, TpDecoderShard
    g = load_golden("llm_tiny")
    cfg = C.BailingMoeConfig(**g["config"])
    dec = BailingMoeDecoder.from_state_dict(cfg, _dev(llm_sd(g["config"], g["rf_config"], g["seed"])), t_max=64, n_seq=3)
    world = 2
    comms = []

    class Loopback:                                   # all_gather of the in-process "ranks": slab p comes from rank p's own inbox
        def __init__(self, rank):
            self.rank = rank

        def get_rank(self):
            return self.rank

        def get_world_size(self):
            return world

        def all_gather(self, outs, mine):
            for p_, o in enumerate(outs):
                src = comms[p_]._relay[1].reshape(-1)[o.storage_offset():o.storage_offset() + o.numel()]
                if p_ != self.rank:
                    o.copy_(src)

    comms.extend(TpCommunicator.relayed(Loopback(r), 64, cfg.hidden_size, "cuda") for r in range(world))
    shards = [TpDecoderShard(dec, r, world) for r in range(world)]
    emb = g["emb"][0].cuda()
    T = emb.shape[0]
    slot = torch.arange(T, dtype=torch.int32).cuda()
    seq = torch.zeros(T, dtype=torch.int32).cuda()
    im = g["image_mask"][0].to(torch.uint8).cuda()
    outs = [torch.empty(T, cfg.hidden_size, device="cuda") for _ in range(world)]
    n_seg = shards[0].n_segments(comms[0], T)
    assert n_seg == shards[0].n_segments()               # the relayed transport delivers whole rows: one-shot only
    base = comms[0].struct.epoch
    for s_ in range(n_seg):
        for r in range(world):
            shards[r].step_tp(comms[r], emb, seq, slot, slot, slot + 1, None, im, out=outs[r], seg_begin=s_, seg_end=s_ + 1)
        if s_ + 1 < n_seg:
            for r in range(world):
                comms[r].relay(base + s_ + 1, T * cfg.hidden_size)
    for c in comms:
        c.check_err()
    assert rel_err(outs[0], g["hidden"][0]) < TOL and torch.equal(outs[0], outs[1])


def test_tp_rf_sampler_full_size_matches_unsharded():
    """The full RF head (w = 3072, 12 blocks, hidden 8192, 16 Euler steps: 192 all-reduces per token) split 8 ways, 2 and 6 CFG rows."""
    from ming_univision_amd.rf_head import RectifiedFlowHead
    from ming_univision_amd.synth import synth_tensor
    from ming_univision_amd.tp import TpCommunicator, TpRfShard
    cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
    rf_cfg = dict(C.DEFAULT_VISHEAD_DIFFLOSS)
    shapes = C.llm_param_shapes(cfg, rf_cfg, 32)
    sd = {k: synth_tensor(k, s_, 7, "cuda", torch.bfloat16) for k, s_ in shapes.items() if k.startswith("vis_head") or k.startswith("diffloss")}
    rf = RectifiedFlowHead(sd, cfg.hidden_size, rf_cfg)
    world = 8
    shards = [TpRfShard(rf, r, world) for r in range(world)]
    assert shards[0].hidden == 1024 and shards[0].n_segments() == 193
    comms = TpCommunicator.simulated(world, rows_cap=8, width=rf.w)
    g = torch.Generator().manual_seed(5)
    for n_images, rpi in ((1, 2), (2, 3)):
        hidden = torch.randn(n_images * rpi, cfg.hidden_size, generator=g).cuda()
        noise = torch.randn(n_images, 32, generator=g).cuda()
        ref = rf.sample(hidden, noise, n_images=n_images)
        outs = [torch.empty_like(ref) for _ in range(world)]
        for seg in range(193):
            for r in range(world):
                shards[r].sample_tp(comms[r], hidden, noise, out=outs[r], n_images=n_images, seg_begin=seg, seg_end=seg + 1)
        for r in range(world):
            comms[r].check_err()
            assert torch.equal(outs[r], outs[0])
        err = (outs[0] - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)
        assert float(err.max()) < 2e-4, float(err.max())
    assert comms[0].struct.epoch == 2 * 192


def test_tp8_full_width_generate_image_vs_oracle():
    """BASELINE configs[4] shapes on one GPU: the 16B-A3B layer shapes (16 q / 4 KV heads -> 2 + 1 per rank, 64 experts -> 8 per rank,
    shared width 2816 -> 352 (+32 zero units) per rank), the full RF head (hidden 8192 -> 1024 per rank), 2 LLM layers, 3 visual tokens,
    3 CFG rows, TP = 8: prompt prefill, the AR loop with the sharded decoder stack and sampler, the replicated semantic decoder —
    image 0 against the fp32 oracle <= 1e-3 and against the unsharded HIP path."""
    from oracle import bailing_ref, mingtok_ref
    from ming_univision_amd.bailing_moe import BailingMoeDecoder, generate_image
    from ming_univision_amd.mingtok import MingTok
    from ming_univision_amd.rf_head import RectifiedFlowHead
    from ming_univision_amd.tp import TpSimGroup, shard_plan
    seed = 5
    d = C.BailingMoeConfig.ming_univision_16b_a3b().to_dict()
    d.pop("model_type", None)
    d.update(num_hidden_layers=2, vocab_size=1024, num_image_tokens_for_gen=3, image_start_token=1000, pad_token_id=0)
    rf_cfg = dict(C.DEFAULT_VISHEAD_DIFFLOSS)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    sd = llm_sd(d, rf_cfg, seed)
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in d.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    cfg = C.BailingMoeConfig(**d)
    pl = shard_plan(cfg, 8, rf_hidden=8192)
    assert (pl["n_q"], pl["n_kv"], pl["n_experts"], pl["shared"], pl["shared_pad"], pl["rf_hidden"]) == (2, 1, 8, 352, 384, 1024)
    dsd = _dev(sd)
    dec = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=32, n_seq=3)
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
    tu = am.clone(); tu[0, 2:5] = 0
    kvs = bailing_ref.new_kv(ocfg)
    bailing_ref.model_forward(sd["model.word_embeddings.weight"][ids], sd, ocfg, torch.ones(1, T, dtype=torch.long), None, kvs)
    caches = mingtok_ref.semdec_new_cache(tsd)
    ref = bailing_ref.generate_image(
        sd["model.word_embeddings.weight"][torch.tensor([[cfg.image_start_token]])], kvs, am, un, tu, sd, ocfg, noises,
        latent_to_sem=lambda lat: mingtok_ref.mingtok_feature_decoder_step(lat, tsd, caches),
        linear_proj=lambda s: bailing_ref.linear_proj(s, lsd), sem_to_pix=lambda s: None, steps=16)
    assert ref["last_hidden"].shape[0] == 3
    grp = TpSimGroup(dec, rf, 8, rows_cap=16)
    grp.prefill(dec.embed(ids[0].cuda()), seq=0, past=0)
    start = dec.embed(torch.tensor([cfg.image_start_token]).cuda())
    out = generate_image(grp, grp.sampler(), tok, start, T, am, un, tu, noises.cuda(), decode_pixels=False)
    grp.check_err()
    errs = (rel_err(out["latents"], ref["latents"][:, 0]), rel_err(out["sem"], ref["sem"][0]), rel_err(out["last_hidden"], ref["last_hidden"][:, 0]))
    print("TP = 8 (simulated on one GPU) vs oracle: latents %.2e sem %.2e hidden %.2e" % errs)
    assert max(errs) < TOL, errs
    dec.prefill(dec.embed(ids[0].cuda()), seq=0, past=0)
    one = generate_image(dec, rf, tok, start, T, am, un, tu, noises.cuda(), decode_pixels=False)
    assert rel_err(out["latents"], one["latents"]) < 5e-4
    # bytes a rank streams per visual token at TP = 8 vs the unsharded path (the point of TP: batch-1 latency)
    print("decoder-stack weight bytes per rank: %.2f GB of %.2f GB" % (grp.shards[0].weight_bytes() / 1e9,
          sum(t.numel() * 2 for ly in dec.layers for k, t in ly.items() if t is not None and k not in ("ln1", "ln2")) / 1e9))


def test_lmhead_argmax_matches_logits_argmax():
    """mn_lmhead_argmax (1, 7, 40 and 130 rows: skinny and MFMA routes) against torch.argmax of mn logits; vocabulary slices with an
    offset (the TP form) reduce to the same pick; ties go to the lowest index."""
    import ctypes as Ct
    from ming_univision_amd import ops
    from ming_univision_amd._lib import check, current_stream, lib, ptr
    g = torch.Generator().manual_seed(0)
    V, H = 5000, 256
    W = (torch.randn(V, H, generator=g) * H ** -0.5).to(torch.bfloat16).cuda()
    for M in (1, 7, 40, 130):
        x = torch.randn(M, H, generator=g).cuda()
        idx, val = ops.lmhead_argmax(x, W)
        ref_logits = x.double().cpu() @ W.double().cpu().T
        assert idx.tolist() == ref_logits.argmax(-1).tolist()
        assert rel_err(val, ref_logits.max(-1).values) < 1e-4
        # two vocabulary slices (TP = 2): the better (val, idx) pair is the global pick
        i0, v0 = ops.lmhead_argmax(x, W[:2500], vocab_offset=0)
        i1, v1 = ops.lmhead_argmax(x, W[2500:], vocab_offset=2500)
        pick = torch.where(v1 > v0, i1, i0)
        assert pick.tolist() == idx.tolist()
    W2 = W.clone(); W2[77] = W2[4000]                              # a tie: rows 77 and 4000 are the same vector
    x = W2[4000].float().unsqueeze(0) * 10
    idx, _ = ops.lmhead_argmax(x.contiguous(), W2)
    assert int(idx[0]) == 77


IPC_WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from ming_univision_amd import configuration as C
from ming_univision_amd.bailing_moe import BailingMoeDecoder, ImageGenState
from ming_univision_amd.tp import TpRank
from tests.util import llm_sd, load_golden, rel_err

dist.init_process_group("gloo")                      # setup traffic only (IPC handles, barriers); both ranks drive cuda:0
rank = dist.get_rank()
torch.cuda.set_device(0)
g = load_golden("llm_tiny")
sd = llm_sd(g["config"], g["rf_config"], g["seed"])
cfg = C.BailingMoeConfig(**g["config"])
dec = BailingMoeDecoder.from_state_dict(cfg, {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}, t_max=64, n_seq=3)
tp = TpRank(dec, None, dist, rows_cap=64, transport="xgmi")
emb = g["emb"][0].cuda()
T = emb.shape[0]
torch.cuda.synchronize(); dist.barrier()
h = tp.prefill(emb, seq=0, past=0, image_mask=g["image_mask"][0])           # ONE call per pass: all segments, the waits are real
torch.cuda.synchronize(); tp.check_err(); dist.barrier()
for s in (1, 2):
    tp.copy_sequence(0, s, T)
st = ImageGenState(tp, [g["dec_mask0"]], [T])
errs = [rel_err(h, g["hidden"][0])]
for i in range(g["dec_in"].shape[0]):
    hd = tp.step(g["dec_in"][i][:, 0].cuda().contiguous(), st.row_seq, st.row_slot, st.row_pos, st.row_len, st.key_mask)
    errs.append(rel_err(hd, g["dec_hidden"][i][:, 0]))
    st.advance()
torch.cuda.synchronize(); tp.check_err()
t = torch.tensor([max(errs)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"max_err": float(t.item()), "epoch": int(tp.comm.struct.epoch)}))
tp.comm.close()                                      # unmap the peer's inbox / flags, free this rank's
dist.barrier()
dist.destroy_process_group()
"""


def test_tp_two_processes_one_gpu_over_ipc(tmp_path):
    """TP = 2 as it runs in production — one PROCESS per rank, fine-grained inboxes exchanged as IPC handles (torch.distributed for
    the setup), every composite one call with real flag waits across the processes — except that both ranks drive this box's single
    GPU (the IPC mapping and the system-scope release / acquire are exercised; the xGMI hop is not).  Tiny decoder: prefill with
    image-gate rows + CFG decode steps against the reference's golden hidden states on BOTH ranks."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ipc_worker.py"
    script.write_text(IPC_WORKER % root)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", port, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    g = load_golden("llm_tiny")
    n_ar = 2 * g["config"]["num_hidden_layers"]
    assert res["max_err"] < TOL, res
    assert res["epoch"] == n_ar * (1 + g["dec_in"].shape[0]), res      # one prefill pass + the decode steps, 2 all-reduces per layer


@pytest.mark.parametrize("mode", ["tp-xgmi", "tp-rccl", "replicas"])
def test_bench_two_ranks_on_one_gpu(mode):
    """bench.py's multi-rank paths with TWO real ranks on this box's single GPU (MING_BENCH_SINGLE_GPU=1: gloo process group, both
    ranks on cuda:0; a plumbing check, not a result): `--gpus 2` starts the ranks itself; `--tp` builds each rank's shard of a
    2-layer 16B-A3B stack + the RF head, maps the inboxes over IPC (or relays them through the process group) and generates the
    images as one TP group (`scaling` strong); the default is two replicas (`scaling` weak, twice the tokens)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MING_BENCH_SINGLE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--layers", "2", "--tokens", "16", "--images", "2", "--steps", "1",
           "--warmup", "1", "--no-cpu-baseline", "--no-batch1"]
    if mode.startswith("tp"):
        cmd += ["--tp", "--tp-transport", mode.split("-")[1]]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["outputs_finite"] is True and res["value"] > 0
    assert res["config"]["single_gpu_plumbing_check"] is True
    if mode.startswith("tp"):
        assert res["scaling"] == "strong" and res["config"]["parallelism"].startswith("tp2+ep2")
        assert abs(res["value"] - 16 * 2 / (res["ms_per_step"] / 1e3)) < 1e-6 * res["value"]          # the group's images, counted once
    else:
        assert res["scaling"] == "weak" and res["config"]["parallelism"] == "replicas x2"
        assert abs(res["value"] - 2 * 16 * 2 / (res["ms_per_step"] / 1e3)) < 1e-6 * res["value"]      # both replicas' images


@pytest.mark.parametrize("past_mode", ["KEEP", "DROP"])
def test_tp8_multi_round_edit_text_image_vs_oracle(past_mode, tmp_path, monkeypatch):
    """BASELINE configs[4]'s workload under TP = 8 + EP = 8 (all shards on this GPU), full width (2 LLM layers, full RF head, full
    MingTok): round 1 EDITS — an input image (512 x 512 -> 256 `<imagePatch>` tokens through MingTok + linear_proj, image-gate rows
    in the sharded prefill) plus an instruction, processor-style masks -> THREE CFG rows -> image out; round 2 is text on top of
    the cache; round 3 asks for a SECOND image with its own holes, its CFG rows built from the masks the PAST_MODE policy carried
    over (modeling_bailingmm.py:229-234, 273-299; KEEP keeps round 1's holes, DROP replaces them by the cond mask).  Every round
    against the oracle driven the same way (tests/util.OracleConversation): image latents / semantic tokens / last hidden states
    <= 1e-3, greedy tokens equal."""
    from oracle import bailing_ref
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    from ming_univision_amd.mingtok import MingTok
    from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration
    from ming_univision_amd.processing import cfg_attention_masks
    from ming_univision_amd.rf_head import RectifiedFlowHead
    from ming_univision_amd.tp import TpSimGroup
    from tests.util import OracleConversation
    monkeypatch.setenv("PAST_MODE", past_mode)
    seed = 31
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
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in d.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    cfg = C.BailingMoeConfig(**d)
    dsd, dl = _dev(sd), _dev(lsd)
    dec = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=320, n_seq=3)
    rf = RectifiedFlowHead(dsd, cfg.hidden_size, rf_cfg)
    lp = [(dl["linear_proj.0.weight"], dl["linear_proj.0.bias"]), (dl["linear_proj.2.weight"], dl["linear_proj.2.bias"])]
    tok = MingTok(tcfg, state_dict=tsd, device="cuda", seed=seed, linear_proj=lp)
    grp = TpSimGroup(dec, rf, 8, rows_cap=128)                       # the 275-token prompt prefills in 128-row sharded passes
    mcfg = C.MingUniVisionConfig(mlp_depth=2, llm_config=d, vishead_diffloss_config=rf_cfg)
    model = MingUniVisionForConditionalGeneration.from_parts(mcfg, tok, grp, grp.sampler(), lp, seed=seed)
    ROLE, ROLE_E, HUMAN, ASSIST, IMG, IMG_E, PATCH = 990, 991, 300, 301, 996, 997, 1001
    gen = torch.Generator().manual_seed(12)
    px = torch.rand(1, 3, 512, 512, generator=gen) * 2 - 1
    ids1 = [ROLE, HUMAN, ROLE_E, IMG] + [PATCH] * 256 + [IMG_E] + torch.randint(2, 900, (11,), generator=gen).tolist() + [ROLE, ASSIST, ROLE_E]
    unc1, tunc1 = cfg_attention_masks(ids1, [ROLE, HUMAN, ROLE_E], [ROLE, ASSIST, ROLE_E], {IMG, IMG_E, PATCH})
    assert unc1 != tunc1
    ids2 = [ROLE, HUMAN, ROLE_E] + torch.randint(2, 900, (5,), generator=gen).tolist() + [ROLE, ASSIST, ROLE_E]
    ids3 = [ROLE, HUMAN, ROLE_E] + torch.randint(2, 900, (7,), generator=gen).tolist() + [ROLE, ASSIST, ROLE_E]
    unc3, tunc3 = cfg_attention_masks(ids3, [ROLE, HUMAN, ROLE_E], [ROLE, ASSIST, ROLE_E], {IMG, IMG_E, PATCH})
    n_tok = d["num_image_tokens_for_gen"]
    # the noise generate() will draw for an image (torch.randn on its generator, diff_loss_rf_swiglu.py:117-122) is replayed for the oracle
    noises = []
    for sd_ in (123, 456):
        model.noise_generator.manual_seed(sd_)
        noises.append(torch.randn(n_tok + 1, 32, generator=model.noise_generator, device="cuda").cpu())
    t = lambda v: torch.tensor([v])
    # ---- oracle conversation
    oc = OracleConversation(sd_r, lsd_r, tsd_r, ocfg, steps=int(rf_cfg["num_sampling_steps"]), past_mode=past_mode, eos_token_id=1)
    o1 = oc.round(t(ids1), t(unc1), t(tunc1), pixel_values=px, patch_id=PATCH, max_new_tokens=2, forced_first_token=1000, noises=noises[:1])
    o2 = oc.round(t(ids2), max_new_tokens=3)
    o3 = oc.round(t(ids3), t(unc3), t(tunc3), max_new_tokens=2, forced_first_token=1000, noises=noises[1:])
    assert o1["images"][0]["last_hidden"].shape[0] == 3 and len(o3["images"]) == 1

    # ---- the TP group through the facade
    def run_round(ids, unc=None, tunc=None, noise_seed=None, **kw):
        if noise_seed is not None:
            model.noise_generator.manual_seed(noise_seed)
        return model.generate(input_ids=t(ids), attention_mask=torch.ones(1, len(ids), dtype=torch.long),
                              uncond_attention_mask=None if unc is None else t(unc),
                              text_uncond_attention_mask=None if tunc is None else t(tunc), **kw)
    s1 = run_round(ids1, unc1, tunc1, 123, pixel_values=px, max_new_tokens=2, forced_first_token=1000,
                   output_image_prefix=str(tmp_path / "r1"))
    g1 = model.last_generation
    r1 = o1["images"][0]
    e1 = (rel_err(g1["latents"], r1["latents"][:, 0]), rel_err(g1["sem"], r1["sem"][0]), rel_err(g1["last_hidden"], r1["last_hidden"][:, 0]))
    print("TP = 8 %s round 1 (edit: image in, 3 CFG rows, image out) vs oracle: latents %.2e sem %.2e hidden %.2e" % ((past_mode,) + e1))
    assert g1["last_hidden"].shape[0] == 3 and max(e1) < TOL, e1
    assert s1[0, len(ids1):].tolist() == o1["tokens"]
    s2 = run_round(ids2, max_new_tokens=3)
    assert s2[0, len(ids2):].tolist() == o2["tokens"], (s2[0, len(ids2):].tolist(), o2["tokens"])
    s3 = run_round(ids3, unc3, tunc3, 456, max_new_tokens=2, forced_first_token=1000, output_image_prefix=str(tmp_path / "r3"))
    g3 = model.last_generation
    r3 = o3["images"][0]
    assert g3["last_hidden"].shape[0] == r3["last_hidden"].shape[0]
    e3 = (rel_err(g3["latents"], r3["latents"][:, 0]), rel_err(g3["sem"], r3["sem"][0]), rel_err(g3["last_hidden"], r3["last_hidden"][:, 0]))
    print("TP = 8 %s round 3 (second image on the carried-over cache + masks, %d CFG rows) vs oracle: latents %.2e sem %.2e hidden %.2e"
          % ((past_mode, g3["last_hidden"].shape[0]) + e3))
    assert max(e3) < TOL, e3
    assert s3[0, len(ids3):].tolist() == o3["tokens"]
    assert model.past_len == oc.cache_len
    for a, b in zip((model.past_attention_mask, model.past_uncond_attention_mask, model.past_text_uncond_attention_mask), oc.past):
        assert torch.equal(a, b)
    grp.check_err()
