"""Sampled text decoding (mn_sample_logits; the `do_sample` kwargs the reference forwards to HF generate, modeling_bailingmm.py:
249-262) on the GPU: the kernel against oracle/sample_ref.py (itself pinned to the installed transformers' warpers by
tests/test_sample_oracle.py) draw by draw, the draws against the warped distribution, and the façade's sampled decode against
the oracle model sampled with the same uniforms."""
import numpy as np
import pytest
import torch

from ming_univision_amd import configuration as C
from oracle import sample_ref
from tests.util import llm_sd, load_golden, mingtok_sd

pytestmark = pytest.mark.gpu

EDGE = 2e-6       # a uniform this close to a CDF step may land on the neighbouring token in fp32 (the oracle sums in fp64)


def _check_row(x, us, got, temperature, top_k, top_p):
    """Every draw equals the oracle's; a draw within EDGE of a CDF step may land on a token whose oracle CDF interval lies within EDGE
    of the uniform (fp32 prefix sums cannot resolve less; with tiny probabilities that can be several tokens away)."""
    if top_k <= 0 and top_p >= 1.0:
        ranked = np.arange(len(x))
        p = np.exp((x.astype(np.float64) - x.max()) / temperature)
        p /= p.sum()
    else:
        ranked, p = sample_ref.warped_distribution(x, temperature, top_k, top_p, sample_ref.CANDIDATE_CAP)
    pos = {int(t): i for i, t in enumerate(ranked)}
    cdf = np.cumsum(p)
    n_edge = 0
    for u, t in zip(us, got):
        want, margin = sample_ref.sample_token(x, float(u), temperature, top_k, top_p)
        if margin > EDGE:
            assert int(t) == want, (float(u), int(t), want, temperature, top_k, top_p)
        else:
            n_edge += 1
            assert int(t) in pos
            r = pos[int(t)]
            lo, hi = (cdf[r - 1] if r else 0.0), cdf[r]
            assert lo - EDGE <= float(u) <= hi + EDGE, (float(u), int(t), want, lo, hi)
    return n_edge


@pytest.mark.parametrize("temperature,top_k,top_p", [(1.0, 50, 1.0), (0.7, 50, 0.9), (1.3, 0, 0.8), (1.0, 1, 1.0), (0.6, 0, 1.0),
                                                     (1.0, 5, 0.3), (2.0, 200, 0.95), (1.0, 2048, 1.0), (1.0, 2048, 0.999), (1.5, 0, 0.999)])
def test_sample_logits_vs_oracle_draw_by_draw(temperature, top_k, top_p):
    from ming_univision_amd import ops
    rng = np.random.default_rng(17)
    n_edge = n = 0
    for V, scale in ((126464, 2.5), (4099, 1.0), (1000, 6.0), (37, 1.0), (1, 1.0)):
        x = (rng.standard_normal(V) * scale).astype(np.float32)
        if V == 4099:
            x[rng.integers(0, V, 60)] = x.max()                            # ties at the top and (top_k = 50) at the threshold
        if sample_ref.top_p_margin(x, temperature, top_k, top_p, sample_ref.CANDIDATE_CAP) < 1e-6:
            continue
        M = 48
        us = np.concatenate((rng.random(M - 4), [0.0, 1.0 - 2.0 ** -24, 0.5, 2.0 ** -30])).astype(np.float32)
        logits = torch.from_numpy(x).cuda().repeat(M, 1).contiguous()
        got = ops.sample_logits(logits, torch.from_numpy(us).cuda(), temperature, top_k, top_p).cpu().numpy()
        n_edge += _check_row(x, us, got, temperature, top_k, top_p)
        n += M
        # a strided view (the logits in a wider workspace) and a vocabulary offset
        wide = torch.zeros(3, V + 5, device="cuda")
        wide[:, :V] = logits[:3]
        got2 = ops.sample_logits(wide[:, :V], torch.from_numpy(us[:3]).cuda(), temperature, top_k, top_p, vocab_offset=1000).cpu().numpy()
        np.testing.assert_array_equal(got2, got[:3] + 1000)
    assert n_edge <= n // 20


def test_sample_logits_distinct_rows_and_determinism():
    """Every row is warped on its own; the same uniforms give the same tokens (no dependence on atomics' arrival order)."""
    from ming_univision_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    logits = torch.randn(64, 126464, device="cuda", generator=g) * 3
    u = torch.rand(64, device="cuda", generator=g)
    a = ops.sample_logits(logits, u, 0.9, 50, 0.95)
    for _ in range(3):
        assert torch.equal(ops.sample_logits(logits, u, 0.9, 50, 0.95), a)
    x, us, got = logits.cpu().numpy(), u.cpu().numpy(), a.cpu().numpy()
    for m in range(0, 64, 7):
        _check_row(x[m], us[m:m + 1], got[m:m + 1], 0.9, 50, 0.95)
    # temperature -> 0 is the arg-max
    assert torch.equal(ops.sample_logits(logits, u, 1e-3, 50, 1.0), logits.argmax(-1))


@pytest.mark.parametrize("temperature,top_k,top_p", [(1.0, 8, 1.0), (0.8, 0, 0.7), (1.2, 0, 1.0)])
def test_draws_follow_the_warped_distribution(temperature, top_k, top_p):
    """200 000 draws at torch.rand uniforms: chi-square of the token counts against the oracle's probabilities."""
    from ming_univision_amd import ops
    rng = np.random.default_rng(2)
    V = 300
    x = (rng.standard_normal(V) * 1.5).astype(np.float32)
    N = 200_000
    g = torch.Generator(device="cuda").manual_seed(11)
    logits = torch.from_numpy(x).cuda().repeat(2000, 1).contiguous()
    counts = np.zeros(V)
    for _ in range(N // 2000):
        t = ops.sample_logits(logits, torch.rand(2000, device="cuda", generator=g), temperature, top_k, top_p)
        counts += np.bincount(t.cpu().numpy(), minlength=V)
    pure = top_k <= 0 and top_p >= 1.0
    if pure:
        p = np.exp(x.astype(np.float64) / temperature); p /= p.sum()
    else:
        ranked, pr = sample_ref.warped_distribution(x, temperature, top_k, top_p)
        p = np.zeros(V); p[ranked] = pr
    assert counts[p == 0].sum() == 0                                       # nothing outside the kept set, ever
    keep = p * N >= 5
    chi2 = float((((counts - p * N) ** 2)[keep] / (p * N)[keep]).sum())
    dof = int(keep.sum()) - 1
    assert chi2 < dof + 5 * np.sqrt(2 * dof) + 10, (chi2, dof)


def test_sample_logits_candidate_capacity_status_and_full_vocabulary_nucleus():
    """(1) top-p without top-k cuts at top_p of the FULL vocabulary's mass — a peaked row whose tail beyond rank 2048 holds
    16 % of the mass keeps HF's nucleus, which the 2048-candidate normalisation would have shrunk; (2) top_k above the capacity is an
    error on both sides of the C ABI; (3) a nucleus / tie set that does not fit is cut deterministically (lowest ids among ties) and
    reported through the status word."""
    from ming_univision_amd import _lib, ops
    rng = np.random.default_rng(3)
    cap = sample_ref.CANDIDATE_CAP
    y = np.concatenate([np.array([13.0, 12.5, 12.0], np.float32), rng.standard_normal(120000).astype(np.float32)])
    e = np.exp(y.astype(np.float64) - 13.0)
    top_p = float(0.5 * (e[0] / e.sum() + e[0] / np.sort(e)[-cap:].sum()))
    M = 32
    us = rng.random(M).astype(np.float32)
    st = torch.full((M,), -1, dtype=torch.int32, device="cuda")
    got = ops.sample_logits(torch.from_numpy(y).cuda().repeat(M, 1).contiguous(), torch.from_numpy(us).cuda(), 1.0, 0, top_p, status=st).cpu().numpy()
    assert st.tolist() == [0] * M
    assert set(got.tolist()) == {0, 1}                                     # token 1 is in HF's nucleus (mass above it < top_p of the full mass)
    _check_row(y, us, got, 1.0, 0, top_p)
    # near-uniform row, top_p ~ 1: the nucleus wants ~ all 6000 tokens -> the 2048 best, flagged
    x = rng.standard_normal(6000).astype(np.float32)
    st.fill_(-1)
    got = ops.sample_logits(torch.from_numpy(x).cuda().repeat(M, 1).contiguous(), torch.from_numpy(us).cuda(), 1.0, 0, 0.999999, status=st).cpu().numpy()
    assert st.tolist() == [ops.SAMPLE_NUCLEUS_TRUNCATED] * M == [sample_ref.truncated(x, 1.0, 0, 0.999999)] * M
    _check_row(x, us, got, 1.0, 0, 0.999999)
    # 3000 ties at the 10th score: 9 better tokens + the 2039 lowest tied ids, whatever order the atomics arrive in
    z = np.full(4000, 1.0, np.float32)
    z[100:109] = 5.0
    z[3500:] = 0.0
    st.fill_(-1)
    got = ops.sample_logits(torch.from_numpy(z).cuda().repeat(M, 1).contiguous(), torch.from_numpy(us).cuda(), 1.0, 10, 1.0, status=st).cpu().numpy()
    assert st.tolist() == [ops.SAMPLE_TIES_TRUNCATED] * M == [sample_ref.truncated(z, 1.0, 10, 1.0)] * M
    _check_row(z, us, got, 1.0, 10, 1.0)
    again = ops.sample_logits(torch.from_numpy(z).cuda().repeat(M, 1).contiguous(), torch.from_numpy(us).cuda(), 1.0, 10, 1.0).cpu().numpy()
    assert np.array_equal(got, again)
    with pytest.raises(ValueError):
        ops.sample_logits(torch.zeros(1, 6000, device="cuda"), torch.zeros(1, device="cuda"), 1.0, cap + 1, 1.0)
    idx = torch.empty(1, dtype=torch.int64, device="cuda")
    lg, u1 = torch.zeros(1, 6000, device="cuda"), torch.zeros(1, device="cuda")
    rc = _lib.lib().mn_sample_logits(_lib.ptr(lg), 6000, 1, 6000, 1.0, cap + 1, 1.0, _lib.ptr(u1), 0, _lib.ptr(idx), None, None)
    assert rc != 0 and b"top_k" in _lib.lib().mn_last_error()


def test_generate_do_sample_vs_oracle_model(tmp_path):
    """The façade's sampled decode (tiny golden model): every new token is the oracle model's logits -> oracle draw at the same
    uniform; the same generator seed reproduces the run; greedy stays the default; unknown generate kwargs raise."""
    from oracle import bailing_ref
    from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration
    from ming_univision_amd.synth import synth_state_dict
    g = load_golden("genimg_tiny")
    llm_cfg = dict(g["llm_config"]); llm_cfg["eos_token_id"] = 1
    cfg = C.MingUniVisionConfig(mlp_depth=2, llm_config=llm_cfg, vishead_diffloss_config=g["rf_config"], mingtok_config=g["mingtok_config"])
    sd = llm_sd(g["llm_config"], g["rf_config"], g["seed"])
    ckpt = {"model." + k: v for k, v in sd.items()}
    ckpt.update({"vision." + k: v for k, v in mingtok_sd(g["mingtok_config"], g["seed"]).items()})
    ckpt.update(synth_state_dict(C.linear_proj_param_shapes(128, 256, 2), g["seed"]))
    model = MingUniVisionForConditionalGeneration(cfg, state_dict=ckpt, seed=g["seed"], t_max=64)
    ids = g["ids"]
    T = ids.shape[1]
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in g["llm_config"].items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    img_tok = cfg.llm_config.image_start_token
    n_checked = 0
    for temperature, top_k, top_p, seed in ((1.0, 50, 1.0, 1), (0.7, 10, 0.9, 2), (1.5, 0, 1.0, 3), (1.0, 0, 0.6, 4)):
        n_new = 8
        gen = torch.Generator(device="cuda").manual_seed(seed)
        us = torch.rand(n_new, device="cuda", generator=gen).cpu().numpy()
        gen.manual_seed(seed)
        model.reset_inner_state()
        seqs = model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=n_new, do_sample=True,
                              temperature=temperature, top_k=top_k, top_p=top_p, generator=gen, output_image_prefix=str(tmp_path / "s"))
        new = seqs[0, T:].tolist()
        gen.manual_seed(seed)
        model.reset_inner_state()
        again = model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=n_new, do_sample=True,
                               temperature=temperature, top_k=top_k, top_p=top_p, generator=gen, output_image_prefix=str(tmp_path / "s"))
        assert again[0, T:].tolist() == new
        # teacher-forced oracle: logits of the oracle model on the tokens the device chose so far -> the oracle's draw
        kvs = bailing_ref.new_kv(ocfg)
        h = bailing_ref.model_forward(sd["model.word_embeddings.weight"][ids], sd, ocfg, None, None, kvs)
        for i, tok in enumerate(new):
            lg = bailing_ref.lm_logits(h[:, -1:], sd).reshape(-1).double().numpy()
            want, margin = sample_ref.sample_token(lg, float(us[i]), temperature, top_k, top_p)
            # the device logits carry the decode path's 1e-4-class error: skip draws near a CDF step or a top-p cut
            if margin > 2e-3 and sample_ref.top_p_margin(lg, temperature, top_k, top_p, sample_ref.CANDIDATE_CAP) > 2e-3:
                assert tok == want, (i, tok, want)
                n_checked += 1
            if tok in (1, img_tok) or i + 1 == len(new):
                break
            h = bailing_ref.model_forward(sd["model.word_embeddings.weight"][torch.tensor([[tok]])], sd, ocfg, None, None, kvs)
    assert n_checked >= 12
    model.reset_inner_state()
    greedy = model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=4)[0, T:].tolist()
    model.reset_inner_state()
    cold = model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=4, do_sample=True, temperature=1e-3,
                          top_k=50)[0, T:].tolist()
    assert cold == greedy
    with pytest.raises(TypeError):
        model.generate(input_ids=ids, max_new_tokens=2, num_beams=4)
    with pytest.raises(ValueError):
        model.generate(input_ids=ids, max_new_tokens=2, do_sample=True, temperature=0.0)


def test_generate_text_batch_do_sample_matches_single_rows(tmp_path):
    """Lock-step sampled decode of B conversations: sequence b with uniforms [:, b] equals `generate` alone with those uniforms."""
    from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration
    from ming_univision_amd.synth import synth_state_dict
    g = load_golden("genimg_tiny")
    llm_cfg = dict(g["llm_config"]); llm_cfg["eos_token_id"] = 1
    cfg = C.MingUniVisionConfig(mlp_depth=2, llm_config=llm_cfg, vishead_diffloss_config=g["rf_config"], mingtok_config=g["mingtok_config"])
    sd = llm_sd(g["llm_config"], g["rf_config"], g["seed"])
    ckpt = {"model." + k: v for k, v in sd.items()}
    ckpt.update({"vision." + k: v for k, v in mingtok_sd(g["mingtok_config"], g["seed"]).items()})
    ckpt.update(synth_state_dict(C.linear_proj_param_shapes(128, 256, 2), g["seed"]))
    model = MingUniVisionForConditionalGeneration(cfg, state_dict=ckpt, seed=g["seed"], t_max=64)
    ids = g["ids"]
    B, n_new = 3, 6
    reqs = [{"input_ids": torch.roll(ids, b, 1)} for b in range(B)]
    gen = torch.Generator(device="cuda").manual_seed(9)
    us = torch.rand(n_new, B, device="cuda", generator=gen)
    gen.manual_seed(9)
    out = model.generate_text_batch(reqs, max_new_tokens=n_new, do_sample=True, temperature=0.9, top_k=20, top_p=0.95, generator=gen)

    import ming_univision_amd.modeling as M
    real_rand = torch.rand
    for b in range(B):
        col = us[:, b].contiguous()
        try:
            M.torch.rand = lambda n, device=None, generator=None, _c=col: _c[:n].clone()
            model.reset_inner_state()
            one = model.generate(input_ids=reqs[b]["input_ids"], max_new_tokens=n_new, do_sample=True, temperature=0.9, top_k=20, top_p=0.95,
                                 output_image_prefix=str(tmp_path / "b"))
        finally:
            M.torch.rand = real_rand
        one = one[0, ids.shape[1]:].tolist()
        stop = [i for i, t in enumerate(out[b]) if t in (1, cfg.llm_config.image_start_token)]    # `<image>` is only a token in the batch
        cut = stop[0] + 1 if stop else len(out[b])
        assert one[:cut] == out[b][:cut], (b, one, out[b])


def test_sample_logits_nan_row_returns_a_valid_id():
    """A poisoned row (the TP wait-expiry path writes NaN) must never turn into an out-of-range embedding index."""
    from ming_univision_amd import ops
    V = 5000
    logits = torch.full((3, V), float("nan"), device="cuda")
    logits[1] = torch.randn(V, device="cuda")
    logits[1, 17] = float("nan")
    u = torch.tensor([0.3, 0.7, 0.999], device="cuda")
    for temperature, top_k, top_p in ((1.0, 50, 1.0), (1.0, 0, 0.9), (0.8, 0, 1.0)):
        t = ops.sample_logits(logits, u, temperature, top_k, top_p)
        assert bool(((t >= 0) & (t < V)).all()), t.tolist()
