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
    rng = np.random.default_rng(1)
    x = rng.standard_normal(6000).astype(np.float32)
    ranked, _ = sample_ref.warped_distribution(x, 1.0, 0, 0.999999, cap=sample_ref.CANDIDATE_CAP)
    assert len(ranked) <= sample_ref.CANDIDATE_CAP
    ranked, _ = sample_ref.warped_distribution(x, 1.0, 5000, 1.0, cap=sample_ref.CANDIDATE_CAP)
    assert len(ranked) == sample_ref.CANDIDATE_CAP
