"""GPU parity of every primitive C-ABI operator against a plain fp32 PyTorch statement of the
same op (computed on CPU in float64/float32 from the SAME bf16-rounded weights).
Tolerance: 1e-3 relative (north_star) unless a kernel's output type is bf16 (1 ulp = 2^-8)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-3


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    from ming_univision_amd import ops as o
    o.lib()
    return o


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def bw(*shape, seed=0, scale=1.0):
    """bf16 weight on GPU + its exact fp32 value on CPU"""
    w = rnd(*shape, seed=seed, scale=scale).to(torch.bfloat16)
    return w.cuda(), w.float()


@pytest.mark.parametrize("M,N,K", [(1, 64, 32), (2, 3072, 3072), (3, 1000, 1408), (4, 37, 344), (8, 515, 2048),
                                   (2, 3072, 8192), (5, 129, 8192), (7, 9, 520)])
def test_skinny_plain_bias(ops, M, N, K):
    x = rnd(M, K, seed=1)
    w, wf = bw(N, K, seed=2, scale=K ** -0.5)
    b, bf = bw(N, seed=3)
    y = ops.skinny_gemm(x.cuda(), w, b)
    assert rel(y, x.double() @ wf.double().T + bf.double()) < 1e-5
    for epi, fn in (("silu", F.silu), ("gelu", F.gelu)):
        y = ops.skinny_gemm(x.cuda(), w, b, epilogue=epi)
        assert rel(y, fn(x.double() @ wf.double().T + bf.double())) < 1e-5


@pytest.mark.parametrize("M,N,K", [(2, 8192, 3072), (3, 344, 128), (1, 1408, 2048), (8, 100, 1024)])
def test_skinny_swiglu(ops, M, N, K):
    x = rnd(M, K, seed=4)
    w, wf = bw(2 * N, K, seed=5, scale=K ** -0.5)
    b, bf = bw(2 * N, seed=6)
    y = ops.skinny_gemm(x.cuda(), w, b, epilogue="swiglu")
    ref = x.double() @ wf.double().T + bf.double()
    assert rel(y, F.silu(ref[:, :N]) * ref[:, N:]) < 1e-5


def test_skinny_prologues_and_resid(ops):
    M, N, K = 3, 777, 3072
    x = rnd(M, K, seed=7) * 2 + 0.3
    w, wf = bw(N, K, seed=8, scale=K ** -0.5)
    b, bf = bw(N, seed=9)
    g, gf = bw(K, seed=10); g2 = (gf * 0.1 + 1).to(torch.bfloat16); g, gf = g2.cuda(), g2.float()
    be, bef = bw(K, seed=11, scale=0.1)
    shift, scale, res, gate = rnd(M, K, seed=12), rnd(M, K, seed=13), rnd(M, N, seed=14), rnd(M, N, seed=15)
    xd, wd = x.double(), wf.double()
    lin = lambda h: h @ wd.T + bf.double()
    y = ops.skinny_gemm(x.cuda(), w, b, prologue="silu")
    assert rel(y, lin(F.silu(xd))) < 1e-5
    y = ops.skinny_gemm(x.cuda(), w, b, prologue="add_silu", pro_a=shift[0].contiguous().cuda())
    assert rel(y, lin(F.silu(xd + shift[0].double()))) < 1e-5
    y = ops.skinny_gemm(x.cuda(), w, b, prologue="rmsnorm", ln_g=g, eps=1e-5)
    rms = xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-5) * gf.double()
    assert rel(y, lin(rms)) < 1e-5
    ln = F.layer_norm(xd, (K,), gf.double(), bef.double(), 1e-6)
    y = ops.skinny_gemm(x.cuda(), w, b, prologue="ln", ln_g=g, ln_b=be, eps=1e-6)
    assert rel(y, lin(ln)) < 1e-5
    y = ops.skinny_gemm(x.cuda(), w, b, prologue="ln_mod", ln_g=g, ln_b=be, eps=1e-6, pro_a=shift.cuda(), pro_b=scale.cuda())
    assert rel(y, lin(ln * (1 + scale.double()) + shift.double())) < 1e-5
    y = ops.skinny_gemm(x.cuda(), w, b, prologue="ln_mod", eps=1e-6, pro_a=shift.cuda(), pro_b=scale.cuda())
    assert rel(y, lin(F.layer_norm(xd, (K,), None, None, 1e-6) * (1 + scale.double()) + shift.double())) < 1e-5
    y = ops.skinny_gemm(x.cuda(), w, b, epilogue="resid", res=res.cuda())
    assert rel(y, res.double() + lin(xd)) < 1e-5
    r = res.cuda().clone()
    ops.skinny_gemm(x.cuda(), w, b, epilogue="resid_gate", res=r, gate=gate.cuda(), out=r)   # in place
    assert rel(r, res.double() + gate.double() * lin(xd)) < 1e-5


@pytest.mark.parametrize("M", [2, 3, 4, 7])
def test_expert_pair_launches_at_full_shapes(ops, M):
    """(row, expert) pair launches of a few-row decode step at the 16B-A3B expert shapes, bf16 weights: 16 / 24 / 32 / 56 pairs x the launch
    plan that keeps all pairs' workgroups in one round over the CUs (skinny_gemm.hip: 24 pairs used to get 11 x 24 = 264 workgroups)
    — SwiGLU epilogue, down projection as K-segments with router weights and residual — against float64."""
    g = torch.Generator().manual_seed(60 + M)
    E, S, I, H, top = 64, 2, 1408, 2048, 6
    gu = (torch.randn(E + S, 2 * I, H, generator=g) * H ** -0.5).to(torch.bfloat16)
    dn = (torch.randn(E + S, H, I, generator=g) * I ** -0.5).to(torch.bfloat16)
    xn = torch.randn(M, H, generator=g)
    res = torch.randn(M, H, generator=g)
    idx = torch.stack([torch.cat((torch.randperm(E, generator=g)[:top], torch.tensor([E, E + 1]))) for _ in range(M)]).to(torch.int32)
    w = torch.cat((torch.rand(M, top, generator=g), torch.ones(M, S)), 1)
    out = ops.moe_experts(xn.cuda(), idx.cuda(), w.cuda(), gu.cuda(), dn.cuda(), res.cuda())
    ref = res.double().clone()
    for m in range(M):
        for s_ in range(top + S):
            e = int(idx[m, s_])
            r = gu[e].double() @ xn[m].double()
            hmid = F.silu(r[:I]) * r[I:]
            ref[m] += float(w[m, s_]) * (dn[e].double() @ hmid.float().double())      # the kernel stores the SwiGLU output as fp32
    assert rel(out, ref) < 1e-5, rel(out, ref)


def test_router_and_experts(ops):
    M, H, E, k, I, S = 5, 256, 8, 3, 64, 2
    x = rnd(M, H, seed=20)
    nw, nwf = bw(H, seed=21); nw2 = (nwf * 0.1 + 1).to(torch.bfloat16); nw, nwf = nw2.cuda(), nw2.float()
    gw, gwf = bw(E, H, seed=22, scale=0.3)
    iw, iwf = bw(E, H, seed=23, scale=0.3)
    mask = torch.tensor([1, 0, 1, 0, 0], dtype=torch.uint8)
    xn, idx, w = ops.moe_router(x.cuda(), nw, 1e-5, gw, iw, mask.cuda(), k, True, S)
    xr = (x.double() * torch.rsqrt(x.double().pow(2).mean(-1, keepdim=True) + 1e-5) * nwf.double())
    assert rel(xn, xr) < 1e-5
    for m in range(M):
        gsel = iwf if mask[m] else gwf
        p = (xr[m] @ gsel.double().T).softmax(-1)
        tw, ti = torch.topk(p, k)
        assert ti.tolist() == idx[m, :k].cpu().tolist()
        assert rel(w[m, :k], tw / tw.sum()) < 1e-5
        assert idx[m, k:].cpu().tolist() == [E, E + 1] and w[m, k:].cpu().tolist() == [1.0, 1.0]
    gu, guf = bw(E + S, 2 * I, H, seed=24, scale=H ** -0.5)
    dn, dnf = bw(E + S, H, I, seed=25, scale=I ** -0.5)
    res = rnd(M, H, seed=26)
    y = ops.moe_experts(xn, idx, w, gu, dn, res.cuda())
    ref = res.double().clone()
    for m in range(M):
        for s in range(k + S):
            e, ww = int(idx[m, s]), float(w[m, s])
            h = guf[e].double() @ xr[m]
            ref[m] += ww * (dnf[e].double() @ (F.silu(h[:I]) * h[I:]))
    assert rel(y, ref) < 1e-5


@pytest.mark.parametrize("hd,nq,nkv,rope", [(128, 16, 4, True), (64, 16, 16, False), (128, 4, 2, True)])
def test_rope_kv_attn_decode(ops, hd, nq, nkv, rope):
    n_seq, t_max, M = 3, 300, 3
    kv = rnd(n_seq, 2, nkv, t_max, hd, seed=30)
    kvd = kv.cuda()
    qkv = rnd(M, (nq + 2 * nkv) * hd, seed=31)
    lens = torch.tensor([200, 157, 1], dtype=torch.int32)
    slot = lens - 1
    pos = torch.tensor([120, 99, 0], dtype=torch.int32)
    seqs = torch.tensor([0, 1, 2], dtype=torch.int32)
    mask = (torch.rand(M, t_max, generator=torch.Generator().manual_seed(32)) > 0.3).to(torch.uint8)
    for m in range(M):
        mask[m, slot[m]] = 1
    from oracle.bailing_ref import rope_cos_sin, rotate_half
    cos, sin = rope_cos_sin(hd, 600000.0, 256)
    half = hd // 2
    qs = 1.0 / math.sqrt(hd)
    q = ops.rope_kv_append(qkv.cuda(), nq, nkv, hd, kvd, seqs.cuda(), slot.cuda(), pos.cuda() if rope else None,
                           cos[:, :half].contiguous().cuda() if rope else None,
                           sin[:, :half].contiguous().cuda() if rope else None, qs)
    out = ops.attn_decode(q, nq, nkv, hd, kvd, seqs.cuda(), lens.cuda(), mask.cuda())
    x = qkv.double().view(M, nq + 2 * nkv, hd)
    for m in range(M):
        qq, kk, vv = x[m, :nq], x[m, nq:nq + nkv], x[m, nq + nkv:]
        if rope:
            c, s = cos[pos[m]].double(), sin[pos[m]].double()
            qq = qq * c + rotate_half(qq) * s
            kk = kk * c + rotate_half(kk) * s
        assert rel(q[m].view(nq, hd), qq * qs) < 1e-5
        L = int(lens[m])
        K = kv[m, 0, :, :L].double().clone(); V = kv[m, 1, :, :L].double().clone()
        K[:, L - 1], V[:, L - 1] = kk, vv
        assert rel(kvd[m, 0, :, L - 1], kk) < 1e-6 and rel(kvd[m, 1, :, L - 1], vv) < 1e-6
        rep = nq // nkv
        for h in range(nq):
            sc = (K[h // rep] @ (qq[h] * qs))
            sc = sc.masked_fill(mask[m, :L] == 0, float("-inf")).softmax(-1)
            assert rel(out[m].view(nq, hd)[h], sc @ V[h // rep]) < 1e-5, (m, h)


@pytest.mark.parametrize("M,hd,nq,nkv", [(3, 128, 16, 4), (70, 128, 16, 4), (5, 64, 16, 16)])
def test_attn_decode_fully_masked_row_follows_the_reference_mask(ops, M, hd, nq, nkv):
    """A row whose keys are ALL masked: the reference's additive finfo.min mask (modeling_bailing_moe.py:1466) absorbs every score in
    fp32, the softmax degenerates to uniform over the row's keys and the output is their mean V — restated exactly as the reference
    computes it (scores + finfo.min, fp32 softmax).  Rows 0 and M-1 are fully masked, the others keep their holey masks; 3 and 5
    rows take the per-head kernel, 70 rows the GQA kernel of the wide route."""
    n_seq, t_max = M, 96
    kv = rnd(n_seq, 2, nkv, t_max, hd, seed=40)
    q = rnd(M, nq * hd, seed=41) * (1.0 / math.sqrt(hd))
    g = torch.Generator().manual_seed(42)
    lens = torch.randint(1, t_max + 1, (M,), generator=g).to(torch.int32)
    mask = (torch.rand(M, t_max, generator=g) > 0.4).to(torch.uint8)
    for m in range(M):
        mask[m, lens[m] - 1] = 1
    mask[0] = 0
    mask[M - 1] = 0
    seqs = torch.arange(M, dtype=torch.int32)
    out = ops.attn_decode(q.cuda(), nq, nkv, hd, kv.cuda(), seqs.cuda(), lens.cuda(), mask.cuda())
    assert torch.isfinite(out).all()
    neg = torch.finfo(torch.float32).min
    rep = nq // nkv
    for m in range(M):
        L = int(lens[m])
        K, V = kv[m, 0, :, :L].float(), kv[m, 1, :, :L].float()
        add = torch.where(mask[m, :L] == 0, torch.tensor(neg), torch.tensor(0.0))
        for h in (0, nq // 2, nq - 1):
            w = (K[h // rep] @ q[m].view(nq, hd)[h]) + add                       # fp32, like the reference's eager path
            ref = torch.softmax(w, dim=-1, dtype=torch.float32) @ V[h // rep]
            assert rel(out[m].view(nq, hd)[h], ref) < 2e-5, (m, h)
    assert rel(out[0].view(nq, hd)[0], kv[0, 1, 0, :int(lens[0])].mean(0)) < 2e-5   # = the mean of V over the row's keys


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 344), (4160, 2304, 768), (96, 1000, 3072), (70, 32, 32)])
def test_gemm_bf16(ops, M, N, K):
    a = rnd(M, K, seed=40).to(torch.bfloat16)
    w, wf = bw(N, K, seed=41, scale=K ** -0.5)
    b, bf = bw(N, seed=42)
    ref = a.double() @ wf.double().T + bf.double()
    y = ops.gemm_bf16(a.cuda(), w, b, "f32")
    assert rel(y, ref) < 1e-5
    y = ops.gemm_bf16(a.cuda(), w, b, "bf16")
    assert rel(y, ref) < 2 ** -8
    y = ops.gemm_bf16(a.cuda(), w, b, "bf16_gelu")
    assert rel(y, F.gelu(ref)) < 2 ** -8
    acc = rnd(M, N, seed=43)
    accd = acc.cuda()
    ops.gemm_bf16(a.cuda(), w, b, "f32_resid", out=accd)
    assert rel(accd, acc.double() + ref) < 1e-5


def test_layernorm_swiglu_convert(ops):
    x = rnd(37, 768, seed=50) * 3 + 1
    g, gf = bw(768, seed=51); be, bef = bw(768, seed=52)
    ref = F.layer_norm(x.double(), (768,), gf.double(), bef.double(), 1e-6)
    assert rel(ops.layernorm_bf16(x.cuda(), g, be), ref) < 2 ** -8
    assert rel(ops.layernorm_bf16(x.cuda(), g, be, gelu=True), F.gelu(ref)) < 2 ** -8
    x12 = rnd(9, 688, seed=53).to(torch.bfloat16)
    assert rel(ops.swiglu_bf16(x12.cuda()), F.silu(x12[:, :344].double()) * x12[:, 344:].double()) < 2 ** -8
    v = rnd(1000, seed=54)
    assert torch.equal(ops.f32_to_bf16(v.cuda()).cpu(), v.to(torch.bfloat16))
    hi, lo = ops.f32_split_bf16(v.cuda())
    assert rel(hi.float() + lo.float(), v) < 2 ** -15


@pytest.mark.parametrize("B,T,nh,causal", [(2, 65, 2, False), (1, 257, 12, False), (2, 5, 2, True), (1, 300, 16, True),
                                           (1, 1024, 16, False)])
def test_attn_prefill_hd64(ops, B, T, nh, causal):
    qkv = rnd(B * T, 3 * nh * 64, seed=60).to(torch.bfloat16)
    out = ops.attn_prefill_hd64(qkv.cuda(), B, T, nh, causal)
    x = qkv.double().view(B, T, 3, nh, 64).permute(2, 0, 3, 1, 4)
    q, k, v = x[0] * 0.125, x[1], x[2]
    att = q @ k.transpose(-1, -2)
    if causal:
        att = att.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool), 1), float("-inf"))
    ref = (att.softmax(-1) @ v).transpose(1, 2).reshape(B * T, nh * 64)
    # P is rounded to bf16 before P.V (as flash-attn does) and the output is bf16
    assert rel(out, ref) < 3 * 2 ** -8


@pytest.mark.parametrize("B,T,nh,causal", [(2, 257, 12, False), (1, 300, 4, True), (3, 64, 2, True), (1, 1025, 16, False), (2, 130, 3, True)])
def test_attn_prefill_hd64_f32_hilo(ops, B, T, nh, causal):
    """The fp32-class flash attention (operands as bf16 hi + lo pairs, three MFMAs per product) against fp64 softmax attention on
    fp32 inputs with a wide dynamic range: 1e-4-class error (the plain bf16 kernel above: 1e-2), fp32 and hi/lo outputs agree."""
    g = torch.Generator().manual_seed(62)
    qkv = torch.randn(B * T, 3 * nh * 64, generator=g) * (1.0 + 3.0 * torch.rand(B * T, 1, generator=g))
    split, out = ops.attn_prefill_hd64_f32(qkv.cuda(), B, T, nh, causal, want_f32=True)
    x = qkv.double().view(B, T, 3, nh, 64).permute(2, 0, 3, 1, 4)
    q, k, v = x[0] * 0.125, x[1], x[2]
    att = q @ k.transpose(-1, -2)
    if causal:
        att = att.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool), 1), float("-inf"))
    ref = (att.softmax(-1) @ v).transpose(1, 2).reshape(B * T, nh * 64)
    assert rel(out, ref) < 1e-4
    rec = split[0].float() + split[1].float()
    assert rel(rec, out) < 2 ** -15
    only_split = ops.attn_prefill_hd64_f32(qkv.cuda(), B, T, nh, causal)
    assert torch.equal(only_split, split)


def test_flash_prefill_gqa_hd128_f32_hilo_spans(ops):
    """The fp32-class GQA flash attention of mn_llm_step_spans (hd 128, 16:4, q fp32, K / V from the fp32 arena, operands as bf16 hi/lo
    pairs): four spans with their OWN past (a sequence cut by a pass boundary continues with past > 0), against fp64 causal softmax
    attention on the unrounded fp32 values: 1e-4-class; fp32 output and hi/lo output agree."""
    from ming_univision_amd._lib import lib, ptr, check, current_stream
    L = lib()
    nq, nkv, t_max, hd = 16, 4, 260, 128
    spans = [(2, 0, 70, 0), (0, 70, 33, 21), (3, 103, 129, 100), (1, 232, 17, 0)]        # (seq, r0, len, past)
    M = sum(n for _, _, n, _ in spans)
    g = torch.Generator().manual_seed(63)
    kv = torch.randn(4, 2, nkv, t_max, hd, generator=g) * (0.5 + torch.rand(4, 2, nkv, t_max, 1, generator=g) * 2)
    q = torch.randn(M, nq, hd, generator=g) * 0.2
    tab = torch.tensor(spans, dtype=torch.int32).cuda()
    out = torch.zeros(M, nq * hd, device="cuda")
    split = torch.zeros(2, M, nq * hd, dtype=torch.bfloat16, device="cuda")
    qd, kvd = q.cuda(), kv.cuda()                    # (named: a temporary's block would be reused by the next allocation)
    check(L.mn_flash_prefill_gqa_hd128_f32(ptr(qd), ptr(kvd), t_max, nq, nkv, ptr(tab), len(spans), max(n for _, _, n, _ in spans),
                                           ptr(out), ptr(split), M * nq * hd, current_stream()), "flash f32")
    kd = kv.double()
    for s_, r0, n, past in spans:
        T = past + n
        K = kd[s_, 0, :, :T].repeat_interleave(nq // nkv, 0)
        V = kd[s_, 1, :, :T].repeat_interleave(nq // nkv, 0)
        Q = q[r0:r0 + n].double().permute(1, 0, 2)
        att = Q @ K.transpose(-1, -2)
        ok = torch.arange(T)[None, :] <= (past + torch.arange(n))[:, None]
        att = att.masked_fill(~ok[None], float("-inf"))
        ref = (att.softmax(-1) @ V).permute(1, 0, 2).reshape(n, nq * hd)
        assert rel(out[r0:r0 + n], ref) < 1e-4, (s_, r0, n, past)
    assert rel(split[0].float() + split[1].float(), out) < 2 ** -15


@pytest.mark.parametrize("past,with_mask", [(0, False), (37, False), (5, True)])
def test_flash_prefill_gqa_hd128_spans(ops, past, with_mask):
    """GQA 16:4 flash attention (hd 128, bottom-right causal) of three prompt spans of different lengths in ONE launch, K / V
    read from the fp32 arena (sequences 2, 0, 3 of 4), with keys before the span (past) and holey key masks; and the
    one-span entry mn_attn_prefill_gqa_hd128, against fp64 softmax attention."""
    from ming_univision_amd._lib import lib, ptr, check, current_stream
    L = lib()
    nq, nkv, t_max, hd = 16, 4, 200, 128
    lens, seqs = [70, 33, 129], [2, 0, 3]
    g = torch.Generator().manual_seed(61)
    kv = torch.randn(4, 2, nkv, t_max, hd, generator=g)
    q = (torch.randn(sum(lens), nq, hd, generator=g) * 0.2).to(torch.bfloat16)
    km = None
    if with_mask:
        km = (torch.rand(len(lens), t_max, generator=g) > 0.3).to(torch.uint8)
        for i, n in enumerate(lens):
            km[i, past:past + n] |= (torch.arange(n) % 7 == 0).to(torch.uint8)     # a few guaranteed keys (incl. query 0's own)
    r0s = [sum(lens[:i]) for i in range(len(lens))]
    tab = torch.tensor([[s_, r0, n] for s_, r0, n in zip(seqs, r0s, lens)], dtype=torch.int32).cuda()
    kvd, qd = kv.cuda(), q.cuda()
    kmd = km.cuda() if with_mask else None
    out = torch.zeros(sum(lens), nq * hd, dtype=torch.bfloat16, device="cuda")
    check(L.mn_flash_prefill_gqa_hd128(ptr(qd), ptr(kvd), t_max, nq, nkv, past, ptr(tab), len(lens), max(lens),
                                       ptr(kmd), t_max, ptr(out), current_stream()), "flash")
    kb = kv.to(torch.bfloat16).double()                      # the kernel multiplies bf16 K / V
    for i, (s_, r0, n) in enumerate(zip(seqs, r0s, lens)):
        T = past + n
        K = kb[s_, 0, :, :T].repeat_interleave(nq // nkv, 0)  # [nq, T, hd]
        V = kb[s_, 1, :, :T].repeat_interleave(nq // nkv, 0)
        Q = q[r0:r0 + n].double().permute(1, 0, 2)            # [nq, n, hd]
        att = Q @ K.transpose(-1, -2)
        ok = torch.arange(T)[None, :] <= (past + torch.arange(n))[:, None]
        if with_mask:
            ok = ok & km[i, :T].bool()[None, :]
        att = att.masked_fill(~ok[None], float("-inf"))
        ref = (att.softmax(-1) @ V).permute(1, 0, 2).reshape(n, nq * hd)
        assert rel(out[r0:r0 + n], ref) < 3 * 2 ** -8, (i, rel(out[r0:r0 + n], ref))
        one = torch.zeros(n, nq * hd, dtype=torch.bfloat16, device="cuda")
        check(L.mn_attn_prefill_gqa_hd128(ptr(qd[r0:]), ptr(kvd[s_]), t_max, nq, nkv, past, n,
                                          ptr(kmd[i]) if with_mask else None, ptr(one), current_stream()), "attn")
        assert torch.equal(one, out[r0:r0 + n])


@pytest.mark.parametrize("M,N,K,epi,pro", [(9, 515, 2048, "none", "rmsnorm"), (16, 8192, 3072, "swiglu", "ln_mod"),
                                           (12, 3072, 8192, "resid_gate", "none"), (16, 32, 3072, "none", "ln_mod"),
                                           (10, 1000, 32, "silu", "none"), (16, 3072, 2048, "resid", "add_silu"),
                                           (32, 8192, 3072, "swiglu", "ln_mod"), (24, 3072, 2048, "resid", "rmsnorm"),
                                           (17, 100, 1408, "none", "none"), (31, 3072, 8192, "resid_gate", "none"),
                                           (64, 8192, 3072, "swiglu", "ln_mod"), (48, 3072, 8192, "resid_gate", "none"),
                                           (33, 100, 264, "none", "rmsnorm"), (64, 2048, 2048, "resid", "rmsnorm")])
def test_skinny_medium_rows(ops, M, N, K, epi, pro):
    """5..32 rows take the split-bf16 MFMA route (prologue -> K-sliced streaming GEMM -> reduce+epilogue);
    > 16 rows use two 16-row MFMA tiles per weight tile, > 32 rows the K-loop form (four row tiles, two weight tiles
    per wave)."""
    x = rnd(M, K, seed=70) * 1.5 + 0.2
    rows = 2 * N if epi == "swiglu" else N
    w, wf = bw(rows, K, seed=71, scale=K ** -0.5)
    b, bf = bw(rows, seed=72)
    g, gf = bw(K, seed=73); g2 = (gf * 0.1 + 1).to(torch.bfloat16); g, gf = g2.cuda(), g2.float()
    be, bef = bw(K, seed=74, scale=0.1)
    sh, sc = rnd(M, K, seed=75), rnd(M, K, seed=76)
    res, gate = rnd(M, N, seed=77), rnd(M, N, seed=78)
    xd = x.double()
    kw = {}
    if pro == "rmsnorm":
        xp = xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-5) * gf.double(); kw = dict(prologue=pro, ln_g=g, eps=1e-5)
    elif pro == "ln_mod":
        xp = F.layer_norm(xd, (K,), gf.double(), bef.double(), 1e-6) * (1 + sc.double()) + sh.double()
        kw = dict(prologue=pro, ln_g=g, ln_b=be, eps=1e-6, pro_a=sh.cuda(), pro_b=sc.cuda())
    elif pro == "add_silu":
        xp = F.silu(xd + sh[0].double()); kw = dict(prologue=pro, pro_a=sh[0].contiguous().cuda())
    else:
        xp = xd
    y = xp @ wf.double().T + bf.double()
    if epi == "swiglu":
        ref = F.silu(y[:, :N]) * y[:, N:]
    elif epi == "silu":
        ref = F.silu(y)
    elif epi == "resid":
        ref = res.double() + y; kw["res"] = res.cuda()
    elif epi == "resid_gate":
        ref = res.double() + gate.double() * y; kw.update(res=res.cuda(), gate=gate.cuda())
    else:
        ref = y
    out = ops.skinny_gemm(x.cuda(), w, b, epilogue=epi, **kw)
    assert rel(out, ref) < 2e-5     # hi+lo bf16 split: 2^-17 relative per activation


@pytest.mark.parametrize("G,max_rows,N,K,gather", [(10, 16, 2816, 2048, True), (7, 32, 2048, 1408, False),
                                                   (3, 5, 100, 264, True), (9, 64, 2816, 2048, True),
                                                   (5, 50, 2048, 1408, False)])
def test_stream_mfma_grouped(ops, G, max_rows, N, K, gather):
    """Grouped weight-streaming kernel (MoE experts): ragged groups incl. empty ones, gathered or contiguous x rows,
    K not a multiple of the 256-k chunk; partial slabs summed here in float64."""
    import ctypes as C
    from ming_univision_amd._lib import lib, ptr, current_stream, check
    g = torch.Generator().manual_seed(5)
    cnt = torch.randint(0, max_rows + 1, (G,), generator=g)
    cnt[1] = 0
    cnt[G - 1] = max_rows
    off = torch.zeros(G + 1, dtype=torch.int32)
    off[1:] = cnt.cumsum(0)
    total = int(off[-1])
    n_x = (9 if max_rows <= 32 else 70) if gather else total
    xrows = torch.randint(0, n_x, (total,), generator=g, dtype=torch.int32) if gather else None
    x = rnd(n_x, K, seed=90)
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    Y = torch.cat([hi, lo], 0).contiguous().cuda()
    w, wf = bw(G, N, K, seed=91, scale=K ** -0.5)
    nz = lib().mn_stream_mfma_grouped_slices(G, max_rows, N, K)
    P = torch.full((nz, total, N), float("nan"), device="cuda")
    offd = off.cuda()
    xr = xrows.cuda() if gather else None
    rc = lib().mn_stream_mfma_grouped(ptr(Y), n_x, ptr(w), N * K, ptr(P), total, ptr(offd), ptr(xr), G, max_rows, N, K,
                                      current_stream())
    assert rc == nz, rc
    out = P.double().sum(0).cpu()
    xe = (hi.double() + lo.double())
    for gi in range(G):
        for r in range(int(off[gi]), int(off[gi + 1])):
            src = int(xrows[r]) if gather else r
            ref = wf[gi].double() @ xe[src]
            assert rel(out[r], ref) < 1e-5, (gi, r)


def test_rope3d_kv_append(ops):
    """mn_rope_kv_append_3d against the reference's 3D rotary (golden from BailingMoe3DRotaryEmbedding +
    apply_multimodal_rotary_pos_emb): distinct t / h / w streams, sections 16/24/24; equal streams == Legacy."""
    import os
    import numpy as np
    from ming_univision_amd.bailing_moe import rope_tables
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rope3d.npz"))
    q, k, pos3 = torch.from_numpy(z["q"]), torch.from_numpy(z["k"]), torch.from_numpy(z["pos3"])
    B, nq, T, hd = q.shape
    nkv = k.shape[1]
    M = B * T
    v = rnd(B, nkv, T, hd, seed=5)
    qkv = torch.cat([q, k, v], dim=1).permute(0, 2, 1, 3).reshape(M, (nq + 2 * nkv) * hd).contiguous()   # rows = (b, t)
    cos, sin = rope_tables(hd, float(z["base"]), 64, torch.device("cuda"))
    kv = torch.zeros(B, 2, nkv, T, hd, device="cuda")
    row_seq = torch.arange(B).repeat_interleave(T).to(torch.int32).cuda()
    row_slot = torch.arange(T).repeat(B).to(torch.int32).cuda()
    for name, p3 in (("q3", pos3), ("q_same", pos3[:1].expand(3, -1, -1))):
        rp = p3.reshape(3, M).to(torch.int32).contiguous().cuda()
        qo = ops.rope_kv_append(qkv.cuda(), nq, nkv, hd, kv, row_seq, row_slot, rp, cos, sin, q_scale=0.5,
                                mrope_section=[16, 24, 24])
        qr = torch.from_numpy(z[name]).permute(0, 2, 1, 3).reshape(M, nq * hd) * 0.5
        kr = torch.from_numpy(z["k3" if name == "q3" else "k_same"])
        assert rel(qo, qr) < 1e-6, name
        assert rel(kv[:, 0], kr) < 1e-6 and rel(kv[:, 1], v) == 0.0, name
    # Legacy entry point with the t stream alone == equal streams
    qo = ops.rope_kv_append(qkv.cuda(), nq, nkv, hd, kv, row_seq, row_slot, pos3[0].reshape(M).to(torch.int32).cuda(), cos, sin,
                            q_scale=0.5)
    assert rel(qo, torch.from_numpy(z["q_same"]).permute(0, 2, 1, 3).reshape(M, nq * hd) * 0.5) < 1e-6


# ---- wide-row GEMM (gemm256.hip): 256 x 256 tiles, hi/lo rows, fused SwiGLU + split, split-K, grouped + gathered ----
@pytest.mark.parametrize("M,N,K", [(300, 1000, 192), (1, 4, 64), (257, 260, 128), (4160, 2304, 768)])
def test_gemm256_plain_epilogues(ops, M, N, K):
    a = rnd(M, K, seed=21).to(torch.bfloat16)
    w, wf = bw(N, K, seed=22, scale=K ** -0.5)
    b, bf = bw(N, seed=23)
    ref = a.double() @ wf.double().T + bf.double()
    assert rel(ops.gemm256(a.cuda(), w, b, "f32"), ref) < 1e-5
    assert rel(ops.gemm256(a.cuda(), w, b, "bf16"), ref) < 2 ** -8
    assert rel(ops.gemm256(a.cuda(), w, b, "bf16_gelu"), F.gelu(ref)) < 2 ** -8
    r0 = rnd(M, N, seed=24)
    out = r0.cuda().clone()
    ops.gemm256(a.cuda(), w, b, "f32_resid", out=out)
    assert rel(out, r0.double() + ref) < 1e-5


@pytest.mark.parametrize("M,N,K", [(200, 516, 1024), (130, 64, 3072), (1100, 260, 192)])
def test_gemm256_hilo_and_splitk(ops, M, N, K):
    x = rnd(M, K, seed=25)
    w, wf = bw(N, K, seed=26, scale=K ** -0.5)
    b, bf = bw(N, seed=27)
    a2 = ops.split_hilo(x.cuda())
    xr = a2[0].double().cpu() + a2[1].double().cpu()
    assert rel(xr, x) < 2 ** -16                                      # the operand itself: x = hi + lo to 2^-17
    ref = xr @ wf.double().T + bf.double()
    assert rel(ops.gemm256(a2, w, b, "f32"), ref) < 1e-5
    for ks in (1, 2, 3, 8):
        P = ops.gemm256_splitk(a2, w, b, ks)
        assert 1 <= P.shape[0] <= ks
        assert rel(P.double().sum(0), ref) < 1e-5


@pytest.mark.parametrize("M,hidden,K", [(130, 200, 256), (512, 2752, 1024), (96, 1408, 2048)])
def test_gemm256_swiglu_split(ops, M, hidden, K):
    x = rnd(M, K, seed=28)
    w, wf = bw(2 * hidden, K, seed=29, scale=K ** -0.5)
    b, bf = bw(2 * hidden, seed=30, scale=0.1)
    a2 = ops.split_hilo(x.cuda())
    xr = a2[0].double().cpu() + a2[1].double().cpu()
    r = xr @ wf.double().T + bf.double()
    ref = F.silu(r[:, :hidden]) * r[:, hidden:]
    y = ops.gemm256_swiglu_split(a2, w, b)
    assert rel(y[0].double() + y[1].double(), ref) < 3e-5            # the result is itself a hi/lo pair (2^-17)
    y = ops.gemm256_swiglu_split(a2, w, None)
    r = xr @ wf.double().T
    assert rel(y[0].double() + y[1].double(), F.silu(r[:, :hidden]) * r[:, hidden:]) < 3e-5


@pytest.mark.parametrize("gather", [True, False])
def test_gemm256_grouped(ops, gather):
    G, K, N, R = 6, 256, 132, 150
    counts = [0, 7, 128, 300, 1, 129]                               # empty group, one row, exact tile, several tiles, tile + 1
    n_pos = sum(counts)
    off = torch.tensor([sum(counts[:g]) for g in range(G + 1)], dtype=torch.int32)
    cnt = torch.tensor(counts, dtype=torch.int32)
    x = rnd(R if gather else n_pos, K, seed=31)
    g = torch.Generator().manual_seed(32)
    rows = torch.randint(0, R, (n_pos,), generator=g, dtype=torch.int32) if gather else None
    a2 = ops.split_hilo(x.cuda())
    xr = a2[0].double().cpu() + a2[1].double().cpu()
    xs = xr[rows.long()] if gather else xr
    for swiglu in (False, True):
        w, wf = bw(G, 2 * N if swiglu else N, K, seed=33 + swiglu, scale=K ** -0.5)
        out = ops.gemm256_grouped(a2, None if rows is None else rows.cuda(), w, off.cuda(), cnt.cuda(), n_pos, max(counts), swiglu)
        ref = torch.zeros(n_pos, N, dtype=torch.float64)
        for gi in range(G):
            lo, hi = int(off[gi]), int(off[gi]) + counts[gi]
            r = xs[lo:hi] @ wf[gi].double().T
            ref[lo:hi] = F.silu(r[:, :N]) * r[:, N:] if swiglu else r
        got = out[0].double() + out[1].double() if swiglu else out
        assert rel(got, ref) < 3e-5


def test_gemm256_grouped_tile_list(ops):
    """The expert GEMMs of the lock-step decoder step: device-built row-tile list (hi/lo rows: 128-row tiles), rows gathered while
    staging, the XCD-aware order ranging over the live tiles of a grid sized for the worst case, dead M-fragments skipped."""
    G, K, N = 13, 320, 136
    # empty, one row, one fragment, +1, every live-fragment count of either wave row (33 / 40: 3 of 4; 70 / 90 / 110: second wave row
    # with 1 / 2 / 3), tile edges, the bench's size, 4 tiles — the kernel runs a different copy of its K loop per live-fragment count
    counts = [0, 1, 16, 17, 40, 70, 90, 110, 127, 128, 129, 144, 400]
    T = sum(counts)
    g = torch.Generator().manual_seed(51)
    ids = torch.cat([torch.full((c,), gi, dtype=torch.int32) for gi, c in enumerate(counts)])[torch.randperm(T, generator=g)].reshape(T, 1)
    x = rnd(T, K, seed=52)
    a2 = ops.split_hilo(x.cuda())
    xr = a2[0].double().cpu() + a2[1].double().cpu()
    for swiglu in (False, True):
        w, wf = bw(G, 2 * N if swiglu else N, K, seed=53 + swiglu, scale=K ** -0.5)
        out, off, cnt, perm = ops.gemm256_grouped_tiles(a2, ids.cuda(), w, G, swiglu)
        assert cnt.cpu().tolist() == counts
        off, perm = off.cpu(), perm.cpu().long()
        assert torch.equal(ids[perm, 0].long(), torch.repeat_interleave(torch.arange(G), torch.tensor(counts)))
        ref = torch.zeros(T, N, dtype=torch.float64)
        for gi in range(G):
            lo, hi = int(off[gi]), int(off[gi]) + counts[gi]
            r = xr[perm[lo:hi]] @ wf[gi].double().T
            ref[lo:hi] = F.silu(r[:, :N]) * r[:, N:] if swiglu else r
        got = out[0].double() + out[1].double() if swiglu else out
        assert rel(got, ref) < 3e-5
        # the order inside a group is arbitrary (LDS atomics of the sort), a row's result does not depend on its position
        out2, _, _, perm2 = ops.gemm256_grouped_tiles(a2, ids.cuda(), w, G, swiglu)
        by_row, by_row2 = torch.empty_like(out), torch.empty_like(out2)
        by_row[..., perm.cuda(), :] = out
        by_row2[..., perm2.long(), :] = out2
        assert torch.equal(by_row, by_row2)


def test_gemm256_race_screen(ops):
    """The counted-vmcnt pipeline has no hardware interlock between LDS-DMA writes and fragment reads: a schedule error shows
    as RARE wrong tiles.  Screen: many launches of production shapes under load must be bitwise identical and right."""
    x = rnd(1024, 3072, seed=41)
    w, wf = bw(2 * 4096, 3072, seed=42, scale=3072 ** -0.5)
    a2 = ops.split_hilo(x.cuda())
    xr = a2[0].double().cpu() + a2[1].double().cpu()
    r = xr @ wf.double().T
    ref_sw = F.silu(r[:, :4096]) * r[:, 4096:]
    first_sw = first_f = None
    for it in range(40):
        y = ops.gemm256_swiglu_split(a2, w, None)
        o = ops.gemm256(a2, w, None, "f32")
        if first_sw is None:
            first_sw, first_f = y.clone(), o.clone()
            assert rel(y[0].double() + y[1].double(), ref_sw) < 3e-5 and rel(o, r) < 1e-5
        else:
            assert torch.equal(y, first_sw) and torch.equal(o, first_f), it
    big = rnd(4096, 4096, seed=43).to(torch.bfloat16).cuda()            # many K-tiles, many tiles per CU
    wb, wbf = bw(4096, 4096, seed=44, scale=4096 ** -0.5)
    o0 = ops.gemm256(big, wb, None, "f32")
    assert rel(o0, big.double().cpu() @ wbf.double().T) < 1e-5
    for it in range(20):
        assert torch.equal(ops.gemm256(big, wb, None, "f32"), o0), it


def test_mingtok_layout_passes_match_the_views_they_replace():
    """layout_ops.hip: im2col + cast / split, cls append + pos-embed add, the sub-token rearrange and unpatchify + clamp are bit-identical
    to the reshape / permute / copy formulations of the reference (patch_embed.py:76-78, vision_transformer.py:218-223, 515-527;
    modeling_mingtok.py:184-188, 195)."""
    from ming_univision_amd import ops
    g = torch.Generator().manual_seed(5)
    for B, Hi, Wi, P in ((2, 64, 96, 32), (3, 256, 256, 32), (1, 32, 32, 16)):
        x = (torch.rand(B, 3, Hi, Wi, generator=g) * 2 - 1).cuda()
        gh, gw = Hi // P, Wi // P
        cols = x.reshape(B, 3, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, 3 * P * P).contiguous()
        assert torch.equal(ops.patchify_operand(x, P), ops.f32_to_bf16(cols))
        assert torch.equal(ops.patchify_operand(x, P, hilo=True), ops.split_hilo(cols))
    B, N, D = 3, 64, 768
    tok = torch.randn(B * N, D, generator=g).cuda()
    cls = torch.randn(D, generator=g).to(torch.bfloat16).cuda()
    pos = torch.randn(N + 1, D, generator=g).cuda()
    want = torch.cat((tok.reshape(B, N, D), cls.float().reshape(1, 1, D).expand(B, 1, D)), 1) + pos.unsqueeze(0)
    assert torch.equal(ops.tokens_assemble(tok, cls, pos, B, N), want)
    B, h, w, r, Dp = 2, 4, 4, 2, 1024
    y = torch.randn(B * h * w, r * r * Dp, generator=g).cuda()
    want = y.reshape(B, h, w, r, r, Dp).permute(0, 1, 3, 2, 4, 5).reshape(B * h * r * w * r, Dp).contiguous()
    assert torch.equal(ops.subtoken_rearrange(y, B, h, w, r, Dp), want)
    B, hh, ww, p = 2, 8, 8, 16
    o = (torch.randn(B * hh * ww, p * p * 3, generator=g) * 0.8).cuda()
    want = o.reshape(B, hh, ww, p, p, 3).permute(0, 5, 1, 3, 2, 4).reshape(B, 3, hh * p, ww * p).clamp(-1.0, 1.0).contiguous()
    assert torch.equal(ops.unpatchify_clamp(o, B, hh, ww, p), want)
