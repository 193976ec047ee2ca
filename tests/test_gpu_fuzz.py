"""Seeded random-shape sweep of the GEMM / attention entry points against float64 statements of the same ops: the shapes the product
calls are covered one by one in test_gpu_kernels.py; this file walks the argument space around them (ragged M / N, K from one K-tile
up, every epilogue, hi/lo and plain operands, random expert splits, random cache lengths and masks).  Tolerances as there: fp32-class
results 3e-5, bf16 results 1 ulp (2^-8 of the row scale)."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    from ming_univision_amd import ops as o
    o.lib()
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def bw(*shape, seed=0, scale=1.0):
    w = rnd(*shape, seed=seed, scale=scale).to(torch.bfloat16)
    return w.cuda(), w.double()


def _shapes(n, seed, m_hi, n_hi, k_tiles_hi):
    r = random.Random(seed)
    out = []
    for _ in range(n):
        M = r.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, r.randint(1, m_hi)])
        N = 4 * r.choice([1, 2, 16, 31, 32, 33, 63, 64, 65, r.randint(1, n_hi // 4)])
        K = 64 * r.choice([1, 2, 3, r.randint(1, k_tiles_hi)])
        out.append((M, N, K))
    return out


@pytest.mark.parametrize("M,N,K", _shapes(14, 101, 700, 1100, 24))
def test_gemm256_random_shapes(ops, M, N, K):
    x = rnd(M, K, seed=M * 7 + N)
    w, wf = bw(N, K, seed=K + 1, scale=K ** -0.5)
    b, bf = bw(N, seed=K + 2)
    a2 = ops.split_hilo(x.cuda())
    xr = a2[0].double().cpu() + a2[1].double().cpu()
    ref = xr @ wf.T + bf
    assert rel(ops.gemm256(a2, w, b, "f32"), ref) < 3e-5                       # hi/lo operand, fp32 result
    acc = rnd(M, N, seed=5).cuda()
    base = acc.double().cpu()
    ops.gemm256(a2, w, b, "f32_resid", out=acc)
    assert rel(acc, base + ref) < 3e-5
    xb = x.to(torch.bfloat16)
    refb = xb.double() @ wf.T + bf
    got = ops.gemm256(xb.cuda(), w, b, "bf16")                                  # plain bf16 operand, bf16 result
    assert rel(got, refb) < 2 ** -8
    assert rel(ops.gemm256(xb.cuda(), w, b, "bf16_gelu"), F.gelu(refb)) < 2 ** -7
    for ks in (2, 3, 5):
        if K // 64 >= 2 * ks:
            P = ops.gemm256_splitk(a2, w, b, ks)
            assert rel(P.sum(0), ref) < 3e-5, ks


@pytest.mark.parametrize("M,N,K", _shapes(8, 202, 520, 700, 16))
def test_gemm256_swiglu_random_shapes(ops, M, N, K):
    x = rnd(M, K, seed=M + 3 * N)
    w, wf = bw(2 * N, K, seed=K + 11, scale=K ** -0.5)
    b, bf = bw(2 * N, seed=K + 12)
    a2 = ops.split_hilo(x.cuda())
    xr = a2[0].double().cpu() + a2[1].double().cpu()
    r = xr @ wf.T + bf
    y = ops.gemm256_swiglu_split(a2, w, b)
    assert rel(y[0].double() + y[1].double(), F.silu(r[:, :N]) * r[:, N:]) < 3e-5
    xb = x.to(torch.bfloat16)
    rb = xb.double() @ wf.T + bf
    assert rel(ops.gemm256_swiglu(xb.cuda(), w, b), F.silu(rb[:, :N]) * rb[:, N:]) < 2 ** -7


@pytest.mark.parametrize("seed", range(6))
def test_grouped_tile_list_random_splits(ops, seed):
    r = random.Random(300 + seed)
    G = r.randint(1, 24)
    T = r.choice([1, 5, 130, r.randint(1, 900)])
    n_slot = r.choice([1, 2, 3])
    K = 64 * r.randint(1, 6)
    N = 4 * r.randint(1, 80)
    g = torch.Generator().manual_seed(seed)
    ids = torch.stack([torch.randperm(max(G, n_slot), generator=g)[:n_slot] % G for _ in range(T)]).to(torch.int32)   # skewed for small G
    if r.random() < 0.5:
        ids[: T // 2] = ids[0]                                                  # a few very large groups
    x = rnd(T, K, seed=seed + 40)
    a2 = ops.split_hilo(x.cuda())
    xr = a2[0].double().cpu() + a2[1].double().cpu()
    for swiglu in (False, True):
        w, wf = bw(G, 2 * N if swiglu else N, K, seed=seed + 50 + swiglu, scale=K ** -0.5)
        out, off, cnt, perm = ops.gemm256_grouped_tiles(a2, ids.cuda(), w, G, swiglu)
        off, cnt, perm = off.cpu(), cnt.cpu(), perm.cpu().long()
        assert int(cnt.sum()) == T * n_slot and cnt.tolist() == torch.bincount(ids.flatten().long(), minlength=G).tolist()
        grp = torch.repeat_interleave(torch.arange(G), cnt.long())
        rr = torch.einsum("pk,pnk->pn", xr[perm], wf[grp])
        ref = F.silu(rr[:, :N]) * rr[:, N:] if swiglu else rr
        got = out[0].double() + out[1].double() if swiglu else out
        assert rel(got, ref) < 3e-5


@pytest.mark.parametrize("M,N,K", [(1, 4, 64), (3, 1000, 1984), (17, 260, 704), (33, 72, 4160), (64, 516, 1408), (48, 3076, 512)])
def test_skinny_random_shapes(ops, M, N, K):
    """<= 64 rows: the weight-streaming route (one-row FMA kernel, MFMA K-slice and K-loop forms by row count)."""
    x = rnd(M, K, seed=M + N)
    w, wf = bw(N, K, seed=K + 21, scale=K ** -0.5)
    b, bf = bw(N, seed=K + 22)
    y = ops.skinny_gemm(x.cuda(), w, b)
    assert rel(y, x.double() @ wf.T + bf) < 3e-5


@pytest.mark.parametrize("seed", range(5))
def test_attn_decode_random_lengths_and_masks(ops, seed):
    r = random.Random(500 + seed)
    hd, nq, nkv = r.choice([(128, 16, 4), (64, 16, 16), (128, 8, 8)])
    M = r.choice([1, 2, 7, 66, 130])                                   # per-head kernels and the GQA kernel of the wide route
    t_max = r.choice([8, 130, 515])
    n_seq = M
    kv = rnd(n_seq, 2, nkv, t_max, hd, seed=seed + 60)
    q = rnd(M, nq * hd, seed=seed + 61)
    lens = torch.tensor([r.randint(1, t_max) for _ in range(M)], dtype=torch.int32)
    seqs = torch.randperm(n_seq, generator=torch.Generator().manual_seed(seed))[:M].to(torch.int32)
    mask = (torch.rand(M, t_max, generator=torch.Generator().manual_seed(seed + 62)) > 0.4).to(torch.uint8)
    for m in range(M):
        mask[m, int(lens[m]) - 1] = 1                                    # at least one live key per row
    use_mask = seed % 2 == 0
    out = ops.attn_decode(q.cuda(), nq, nkv, hd, kv.cuda(), seqs.cuda(), lens.cuda(), mask.cuda() if use_mask else None)
    rep = nq // nkv
    qd = q.double().view(M, nq, hd)
    for m in range(M):
        L, s = int(lens[m]), int(seqs[m])
        K_, V_ = kv[s, 0, :, :L].double(), kv[s, 1, :, :L].double()
        sc = torch.einsum("hd,htd->ht", qd[m], K_.repeat_interleave(rep, 0))
        if use_mask:
            sc = sc.masked_fill(mask[m, :L] == 0, float("-inf"))
        ref = torch.einsum("ht,htd->hd", sc.softmax(-1), V_.repeat_interleave(rep, 0))
        assert rel(out[m].view(nq, hd), ref) < 3e-5, (m, L)


@pytest.mark.parametrize("M,V,H,off", [(1, 1000, 256, 0), (3, 126464, 2048, 0), (2, 4099, 512, 70000), (64, 777, 128, 5)])
def test_lmhead_argmax_random(ops, M, V, H, off):
    h = rnd(M, H, seed=V)
    w, wf = bw(V, H, seed=H, scale=H ** -0.5)
    idx, val = ops.lmhead_argmax(h.cuda(), w, off)
    logits = h.double() @ wf.T
    best = logits.max(-1)
    assert rel(val, best.values) < 3e-5
    # the index may differ from float64's only where two logits tie within the fp32 error of the product
    picked = logits.gather(1, (idx.cpu().long() - off).view(-1, 1)).squeeze(1)
    assert torch.all(best.values - picked <= 3e-5 * best.values.abs().max())


# ---- the decoder step over random small configurations (both routes) ---------------------------------------------------------
def _llm_cases():
    r = random.Random(900)
    cases = []
    for i in range(6):
        hd = r.choice([64, 128])
        n_kv = r.choice([1, 2, 4])
        n_q = n_kv * r.choice([1, 2, 4])
        cases.append(dict(hidden_size=64 * r.randint(1, 5), num_attention_heads=n_q, num_key_value_heads=n_kv, head_dim=hd,
                          num_experts=r.choice([4, 8, 16]), num_experts_per_tok=r.choice([1, 2, 4]), num_shared_experts=r.choice([0, 1, 2]),
                          moe_intermediate_size=64 * r.randint(1, 3), norm_topk_prob=r.random() < 0.7, multi_gate=r.random() < 0.5,
                          num_hidden_layers=2, rows=r.choice([1, 2, 5, 33, 70, 130]), seed=i))
    return cases


@pytest.mark.parametrize("case", _llm_cases(), ids=lambda c: "h%d_q%dkv%dx%d_e%dk%ds%d_rows%d" % (
    c["hidden_size"], c["num_attention_heads"], c["num_key_value_heads"], c["head_dim"], c["num_experts"], c["num_experts_per_tok"],
    c["num_shared_experts"], c["rows"]))
def test_decoder_step_random_configs_vs_oracle(case):
    """Cached decode steps (modeling_bailing_moe.py:1214-1218, 656-829, 505-639) of random small Bailing-MoE configurations — head
    counts, GQA ratios, expert counts, top-k, 0 / 1 / 2 shared experts, image gate on or off — at row counts on the weight-streaming
    route (<= 64) and on the wide route, with holey key masks, against the fp32 oracle."""
    from ming_univision_amd import configuration as C
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    from oracle import bailing_ref
    from tests.util import llm_sd, rel_err
    rows, seed = case["rows"], case["seed"]
    d = dict(vocab_size=128, use_qkv_bias=False, use_bias=False, rms_norm_eps=1e-5, rope_theta=600000.0, first_k_dense_replace=0,
             num_image_tokens_for_gen=4, image_start_token=100, image_patch_token=99, embedding_dropout=0.0, attention_dropout=0.0,
             output_dropout=0.0, pad_token_id=0)
    d.update({k: v for k, v in case.items() if k not in ("rows", "seed")})
    rf_cfg = dict(diffloss_w=64, diffloss_d=1, num_sampling_steps="2", gen_method="flow_matching_swiglu-4", vis_head_arch="linear2-norm")
    sd = llm_sd(d, rf_cfg, 40 + seed)
    cfg = C.BailingMoeConfig(**d)
    dec = BailingMoeDecoder.from_state_dict(cfg, {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}, t_max=8, n_seq=rows)
    sdr = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}          # the oracle multiplies the same bf16-rounded weights
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in d.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    kvs = bailing_ref.new_kv(ocfg)
    gen = torch.Generator().manual_seed(seed)
    seq = torch.arange(rows, dtype=torch.int32).cuda()
    keep = torch.ones(rows, 8, dtype=torch.long)
    for step in range(3):
        x = torch.randn(rows, 1, cfg.hidden_size, generator=gen)
        img = (torch.rand(rows, 1, generator=gen) < 0.5) if case["multi_gate"] else None
        if step == 1:
            keep[::2, 0] = 0                                                # from step 1 on every other row stops seeing key 0
        am = keep[:, :step + 1]
        pos = torch.full((rows, 1), step, dtype=torch.long)
        ref = bailing_ref.model_forward(x, sdr, ocfg, am, pos, kvs, image_mask=img)
        slot = torch.full((rows,), step, dtype=torch.int32).cuda()
        out = dec.step(x[:, 0].cuda().contiguous(), seq, slot, slot, slot + 1, keep.to(torch.uint8).cuda(),
                       image_mask=None if img is None else img[:, 0].to(torch.uint8).cuda())
        assert rel_err(out, ref[:, 0]) < 1e-3, step
