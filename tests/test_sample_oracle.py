"""The sampling oracle (oracle/sample_ref.py) pinned to the installed transformers' logits warpers — CPU only."""
import numpy as np
import pytest
import torch

from oracle import sample_ref

lp = pytest.importorskip("transformers.generation.logits_process")


def hf_distribution(logits, temperature, top_k, top_p):
    """What HF generate samples from: TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> softmax."""
    s = torch.tensor(logits, dtype=torch.float64)[None]
    ids = torch.zeros(1, 1, dtype=torch.long)
    if temperature != 1.0:
        s = lp.TemperatureLogitsWarper(float(temperature))(ids, s)
    if top_k > 0:
        s = lp.TopKLogitsWarper(top_k=top_k)(ids, s)
    if top_p < 1.0:
        s = lp.TopPLogitsWarper(top_p=float(top_p))(ids, s)
    return torch.softmax(s, -1)[0].numpy()


@pytest.mark.parametrize("temperature,top_k,top_p", [(1.0, 50, 1.0), (0.7, 50, 0.9), (1.3, 0, 0.8), (1.0, 1, 1.0), (0.5, 0, 1.0),
                                                     (1.0, 5, 0.3), (2.0, 200, 0.95), (1.0, 3000, 1.0)])
def test_warped_distribution_matches_transformers(temperature, top_k, top_p):
    rng = np.random.default_rng(5)
    for case in range(6):
        V = [1000, 4099, 126464][case % 3]
        x = rng.standard_normal(V).astype(np.float32) * [1.0, 4.0][case % 2]
        if case == 3:
            x[rng.integers(0, V, 40)] = x.max()                       # ties at the top (and possibly at the top-k threshold)
        if sample_ref.top_p_margin(x, temperature, top_k, top_p) < 1e-9:
            continue
        want = hf_distribution(x, temperature, top_k, top_p)
        ranked, p = sample_ref.warped_distribution(x, temperature, top_k, top_p)
        got = np.zeros(V)
        got[ranked] = p
        if case == 3:
            # equal scores: torch.sort's order among them is arbitrary, so WHICH of the tied tokens top-p cuts is too (the oracle
            # cuts the higher ids) — the kept scores and probabilities must still agree as multisets
            np.testing.assert_array_equal(np.sort(x[want > 0]), np.sort(x[ranked]))
            np.testing.assert_allclose(np.sort(got[got > 0]), np.sort(want[want > 0]), rtol=1e-9)
        else:
            assert set(np.flatnonzero(want > 0)) == set(ranked.tolist())
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-15)
        xs = x[ranked].astype(np.float64) / temperature
        assert np.all(np.diff(xs) <= 0)                                # ranked by descending score
        same = np.diff(xs) == 0
        assert np.all(np.diff(ranked)[same] > 0)                       # ties by ascending id


def test_sample_token_inverse_cdf():
    x = np.log(np.array([0.1, 0.4, 0.2, 0.3], dtype=np.float64)).astype(np.float32)
    # ranking: id 1 (0.4), id 3 (0.3), id 2 (0.2), id 0 (0.1)
    assert [sample_ref.sample_token(x, u, top_k=4)[0] for u in (0.0, 0.39, 0.41, 0.69, 0.71, 0.89, 0.91, 0.999)] == [1, 1, 3, 3, 2, 2, 0, 0]
    # top_p = 0.5: id 1 (mass above 0) and id 3 (mass above 0.4 < 0.5) stay: renormalised 4/7, 3/7
    assert [sample_ref.sample_token(x, u, top_k=4, top_p=0.5)[0] for u in (0.0, 0.57, 0.58, 0.99)] == [1, 1, 3, 3]
    # pure temperature sampling walks the vocabulary in id order
    assert [sample_ref.sample_token(x, u)[0] for u in (0.05, 0.15, 0.55, 0.75)] == [0, 1, 2, 3]
    # temperature -> 0 concentrates on the arg-max
    assert sample_ref.sample_token(x, 0.999, temperature=0.01, top_k=4)[0] == 1


def test_candidate_cap_restates_the_kernel():
    """What the kernel's 2048-candidate capacity changes, and only then: a nucleus that fits is HF's nucleus exactly — its
    limit is top_p of the FULL vocabulary's mass, not of the 2048 best —; a larger one is cut and flagged; top_k above the capacity is
    refused; ties that overflow keep the lowest ids."""
    rng = np.random.default_rng(1)
    cap = sample_ref.CANDIDATE_CAP
    x = rng.standard_normal(6000).astype(np.float32)
    ranked, _ = sample_ref.warped_distribution(x, 1.0, 0, 0.999999, cap=cap)            # near-uniform: the nucleus wants ~all 6000
    assert len(ranked) == cap and sample_ref.truncated(x, 1.0, 0, 0.999999) == 1
    # a peaked row with a long tail: HF's nucleus is small although the tail beyond rank 2048 holds real mass
    y = np.concatenate([np.array([13.0, 12.5, 12.0], np.float32), rng.standard_normal(120000).astype(np.float32)])
    e = np.exp(y.astype(np.float64) - 13.0)
    z_full, z_cap = e.sum(), np.sort(e)[-cap:].sum()
    assert 1.0 - z_cap / z_full > 0.1                                                     # the mass the old Z_2048 limit ignored
    top_p = 0.5 * (e[0] / z_full + e[0] / z_cap)            # between the mass above token 1 under the two normalisations
    hf, p_hf = sample_ref.warped_distribution(y, 1.0, 0, top_p, cap=None)
    got, p_got = sample_ref.warped_distribution(y, 1.0, 0, top_p, cap=cap)
    assert hf.tolist() == [0, 1] and np.array_equal(hf, got) and np.allclose(p_hf, p_got) and sample_ref.truncated(y, 1.0, 0, top_p) == 0
    assert e[0] / z_cap >= top_p                            # ... the 2048-candidate normalisation would have dropped token 1
    with pytest.raises(ValueError):
        sample_ref.warped_distribution(x, 1.0, 5000, 1.0, cap=cap)
    # 3000 ties at the 10th score: the 9 better tokens + the 2039 lowest tied ids
    z = np.full(4000, 1.0, np.float32)
    z[100:109] = 5.0
    z[3500:] = 0.0
    ranked, _ = sample_ref.warped_distribution(z, 1.0, 10, 1.0, cap=cap)
    assert len(ranked) == cap and sample_ref.truncated(z, 1.0, 10, 1.0) == 2
    assert ranked[:9].tolist() == list(range(100, 109)) and ranked[9:].tolist() == [i for i in range(4000) if z[i] == 1.0][:cap - 9]
