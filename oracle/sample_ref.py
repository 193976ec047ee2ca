"""CPU oracle of the sampled next-token pick (the `do_sample` branch of text decoding).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference does not implement sampling itself: BailingMMNativeForConditionalGeneration.generate forwards its generate
kwargs to HF `GenerationMixin.generate` (mingunivision/modeling_bailingmm.py:249-262 -> modeling_bailing_moe.py:1769-1796), whose
generation config (mingunivision/config.json:30,103-109) carries do_sample=false, temperature=1.0, top_k=50, top_p=1.0.  The
algorithm therefore lives in the third-party dependency `transformers` (installed here: 4.x/5.x `generation/logits_process.py`),
restated below in numpy float64:

    TemperatureLogitsWarper   scores / temperature
    TopKLogitsWarper          remove scores < the k-th largest score (ties at the threshold stay)
    TopPLogitsWarper          ascending sort, softmax, cumsum; remove where cumsum <= 1 - top_p; the best token always stays
    softmax -> one multinomial draw

PARITY PIN: tests/test_sample_oracle.py runs the installed transformers' three warper classes on the same logits and checks
that `warped_distribution` keeps the same token set with the same probabilities.  The draw itself (torch.multinomial's random
stream) is not portable across devices, so the drawn token is defined here as the inverse CDF over the kept tokens ranked by
descending score (ties: ascending id) at a caller-supplied uniform — the HIP kernel is held to that, and a chi-square test
holds the draws to the warped distribution.
"""
import numpy as np

CANDIDATE_CAP = 2048      # mn_sample_logits ranks at most this many candidates


def warped_distribution(logits, temperature=1.0, top_k=0, top_p=1.0, cap=None):
    """One row of logits -> (ids ranked by descending score / ascending id, their probabilities), after HF's warpers.
    `cap` (None for the plain HF semantics) restates the kernel's candidate capacity: top_k above it is refused; a kept set larger
    than it (a top-p nucleus of more than `cap` tokens without top-k, or ties at the k-th score) is cut to the first `cap` tokens of
    the ranking — the rows the kernel reports through its status word; `truncated(…)` below tells which."""
    x = np.asarray(logits, dtype=np.float64) / float(temperature)
    V = x.shape[0]
    keep = np.ones(V, dtype=bool)
    k = int(top_k)
    if cap is not None and k > cap:
        raise ValueError(f"top_k = {k} exceeds the {cap} candidates the kernel ranks")
    if k > 0:
        k = min(k, V)
        kth = np.sort(x)[V - k]
        keep &= x >= kth                                     # `scores < topk[-1]` is removed: ties with the k-th stay
        if cap is not None and keep.sum() > cap:              # more ties than candidates: the lowest ids among the ties stay
            ties = np.flatnonzero(keep & (x == kth))
            keep[ties[cap - int((x > kth).sum()):]] = False
    if top_p < 1.0:
        ids = np.flatnonzero(keep)
        order = ids[np.lexsort((-ids, x[ids]))]              # ascending score (ties: descending id = reverse of the ranking)
        p = np.exp(x[order] - x[order].max())
        p /= p.sum()
        cum = np.cumsum(p)
        remove = cum <= 1.0 - top_p
        remove[-1] = False                                    # min_tokens_to_keep = 1
        keep[order[remove]] = False
    ids = np.flatnonzero(keep)
    ranked = ids[np.lexsort((ids, -x[ids]))]                 # descending score, ties by ascending id
    if cap is not None and len(ranked) > cap:
        ranked = ranked[:cap]
    p = np.exp(x[ranked] - x[ranked].max())
    return ranked, p / p.sum()


def truncated(logits, temperature=1.0, top_k=0, top_p=1.0, cap=CANDIDATE_CAP):
    """The status word mn_sample_logits reports for this row: 1 = the nucleus holds more than `cap` tokens (top_k = 0), 2 = more
    ties at the k-th score than candidates."""
    x = np.asarray(logits, dtype=np.float64) / float(temperature)
    bits = 0
    if top_k <= 0 and top_p < 1.0 and len(warped_distribution(logits, temperature, 0, top_p, None)[0]) > cap:
        bits |= 1
    if top_k > 0:
        kth = np.sort(x)[x.shape[0] - min(int(top_k), x.shape[0])]
        if int((x >= kth).sum()) > cap:
            bits |= 2
    return bits


def top_p_margin(logits, temperature=1.0, top_k=0, top_p=1.0, cap=None):
    """Distance of the top-p cut from the nearest cumulative-probability boundary (a row whose margin is ~1e-7 may keep one
    token more or less in fp32 than in fp64; tests skip such rows).  inf when top-p is off."""
    if top_p >= 1.0:
        return np.inf
    _, p = warped_distribution(logits, temperature, top_k, 1.0, None)
    above = np.concatenate(([0.0], np.cumsum(p)[:-1]))       # mass ranked above each token
    return float(np.min(np.abs(above[1:] - top_p))) if len(p) > 1 else np.inf


def sample_token(logits, u, temperature=1.0, top_k=0, top_p=1.0, cap=CANDIDATE_CAP):
    """The token mn_sample_logits picks: inverse CDF of the warped distribution at u in [0, 1).  Also returns the distance of
    u from the nearest CDF step (draws within ~1e-6 of a step may fall on the neighbouring token in fp32)."""
    pure = top_k <= 0 and top_p >= 1.0
    if pure:                                                  # whole vocabulary, id order
        x = np.asarray(logits, dtype=np.float64) / float(temperature)
        p = np.exp(x - x.max())
        p /= p.sum()
        ids = np.arange(x.shape[0])
    else:
        ids, p = warped_distribution(logits, temperature, top_k, top_p, cap)
    cdf = np.cumsum(p)
    r = int(np.searchsorted(cdf, u, side="right"))
    r = min(r, len(ids) - 1)
    return int(ids[r]), float(np.min(np.abs(cdf - u)))
