"""CPU-only tests: host logic (processor, CFG masks, configs), the C-ABI surface and the
world_size-2 replica harness over gloo.  No compute call into the HIP library is made here."""
import ctypes
import json
import os
import re
import subprocess
import sys

import pytest
import torch

from ming_univision_amd import configuration as C
from ming_univision_amd.processing import BailingMMProcessor, SpecialTokenTokenizer, cfg_attention_masks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_processor_matches_reference():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "processor.json")))
    proc = BailingMMProcessor(tokenizer=SpecialTokenTokenizer())
    convs = {
        "t2i": [{"role": "HUMAN", "content": [{"type": "text", "text": "Please draw a red cube."}]}],
        "edit": [{"role": "HUMAN", "content": [{"type": "image", "image": "a.png"}, {"type": "text", "text": "make it blue"}]}],
        "multi": [{"role": "HUMAN", "content": [{"type": "text", "text": "hi"}]},
                  {"role": "ASSISTANT", "content": [{"type": "text", "text": "hello there"}]},
                  {"role": "HUMAN", "content": [{"type": "image", "image": "b.png"}, {"type": "text", "text": "what is this?"}]}],
    }
    for name, conv in convs.items():
        text = proc.apply_chat_template(conv, add_generation_prompt=True)
        n_img = text.count("<IMAGE>")
        if n_img:
            text = proc._expand_image_tokens([text], torch.tensor([[1, 4, 4]] * n_img))[0]
        assert text == g[name]["text"], name
        enc = proc.tokenize([text])
        assert enc["input_ids"][0].tolist() == g[name]["input_ids"]
        assert enc["uncond_attention_mask"][0].tolist() == g[name]["uncond"], name
        assert enc["text_uncond_attention_mask"][0].tolist() == g[name]["text_uncond"], name
    enc = proc.tokenize([g["no_assistant"]["text"]])
    assert enc["uncond_attention_mask"][0].tolist() == g["no_assistant"]["uncond"]
    assert enc["text_uncond_attention_mask"][0].tolist() == g["no_assistant"]["text_uncond"]


def test_cfg_masks_edge_cases():
    u, a = [1, 2], [1, 3]
    assert cfg_attention_masks([], u, a, set()) == ([], [])                       # empty
    assert cfg_attention_masks([5, 6, 7], u, a, set()) == ([1, 1, 1], [1, 1, 1])  # no role tags
    m, t = cfg_attention_masks([1, 2, 9, 8, 1, 3], u, a, {8})
    assert m == [1, 1, 0, 0, 1, 1] and t == [1, 1, 0, 1, 1, 1]


def test_processor_call_with_tensor_image():
    proc = BailingMMProcessor()
    img = torch.zeros(3, 512, 512)
    text = proc.apply_chat_template([{"role": "HUMAN", "content": [{"type": "image", "image": "x"}, {"type": "text", "text": "edit"}]}])
    out = proc(images=[img], text=[text], for_edit=True, image_patch_size=32)
    assert out["pixel_values"].shape == (1, 3, 512, 512) and out["image_grid_thw"].tolist() == [[1, 16, 16]]
    assert int((out["input_ids"] == 126346).sum()) == 256
    assert out["uncond_attention_mask"].shape == out["input_ids"].shape


def test_image_transforms():
    from PIL import Image
    from ming_univision_amd.processing import MingTokCenterCropProcessor, MingTokUndProcessor
    im = Image.new("RGB", (300, 200), (255, 0, 0))
    t = MingTokCenterCropProcessor(64)(im)
    assert t.shape == (3, 64, 64) and abs(float(t[0].mean()) - 1.0) < 1e-5 and abs(float(t[1].mean()) + 1.0) < 1e-5
    assert MingTokUndProcessor(32)(im).shape == (3, 32, 32)


def test_image_transforms_geometry_and_values():
    """MingTokCenterCropProcessor / MingTokUndProcessor (processing_bailingmm.py:80-123; mingtok/utils/processor.py:8-30) on a
    non-trivial image: torchvision's Resize(int) maps the SHORTER side to S keeping the aspect (long side int(S * long / short)),
    CenterCrop rounds the offset half-away-from-zero, ToTensor scales by 1/255, Normalize(0.5, 0.5) maps to [-1, 1].  torchvision is
    not installed here, so the expected tensor is assembled from PIL's own resize (what torchvision calls for PIL inputs) with the
    documented geometry; odd sizes exercise both rounding rules."""
    import numpy as np
    from PIL import Image
    from ming_univision_amd.processing import MingTokCenterCropProcessor, MingTokUndProcessor
    rng = np.random.RandomState(0)
    for (w, h, S) in ((301, 200, 64), (97, 233, 32), (64, 64, 48)):
        arr = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        im = Image.fromarray(arr)
        t = MingTokCenterCropProcessor(S)(im)
        nw, nh = (S, int(S * h / w)) if w <= h else (int(S * w / h), S)
        rs = np.asarray(im.resize((nw, nh), Image.BICUBIC), dtype=np.float32)
        left, top = int(round((nw - S) / 2.0)), int(round((nh - S) / 2.0))
        exp = torch.from_numpy(rs[top:top + S, left:left + S]).permute(2, 0, 1) / 255.0
        exp = (exp - 0.5) / 0.5
        assert t.shape == (3, S, S) and t.dtype == torch.float32
        assert torch.equal(t, exp), (w, h, S)
        u = MingTokUndProcessor(S)(im)
        expu = (torch.from_numpy(np.asarray(im.resize((S, S), Image.BICUBIC), dtype=np.float32)).permute(2, 0, 1) / 255.0 - 0.5) / 0.5
        assert torch.equal(u, expu)
        assert float(t.min()) >= -1.0 and float(t.max()) <= 1.0


def test_build_cfg_rows_matches_reference_masks():
    from ming_univision_amd.bailing_moe import build_cfg_rows
    from tests.util import load_golden
    g = load_golden("genimg_tiny")
    for tag, rows in (("rows3", 3), ("rows2", 2)):
        am = build_cfg_rows(g["mask"], g["uncond"], g[tag + "_tuncond"])
        assert am.shape[0] == rows
        n = g["llm_config"]["num_image_tokens_for_gen"]
        assert torch.equal(torch.cat([am, torch.ones(rows, n, dtype=am.dtype)], 1), g[tag + "_mask_out"])
    short = g["uncond"][:, :6]   # uncond mask shorter than the cond mask is padded with the cond tail (:1870-1874)
    am = build_cfg_rows(g["mask"], short, None)
    assert am.shape == (2, g["mask"].shape[1]) and torch.equal(am[1, 6:], g["mask"][0, 6:])


def test_lockstep_group_split_has_no_empty_group():
    from ming_univision_amd.bailing_moe import split_groups
    for B in range(1, 40):
        for n in range(1, B + 1):
            gs = split_groups(B, n)
            assert len(gs) <= n and gs[0][0] == 0 and gs[-1][1] == B
            assert all(lo < hi for lo, hi in gs) and all(a[1] == b[0] for a, b in zip(gs, gs[1:]))
            assert max(hi - lo for lo, hi in gs) == -(-B // n)
    assert split_groups(5, 4) == [(0, 2), (2, 4), (4, 5)] and split_groups(9, 4) == [(0, 3), (3, 6), (6, 9)]


def test_splitk_plan_for_few_row_residual_gemms():
    """ops.splitk_plan: split only when 256 x 256 tiles would cover under half the chip, keep >= 256 k per slice."""
    from ming_univision_amd.ops import splitk_plan
    assert splitk_plan(16384, 1024, 1024) == 0            # 64 x 4 tiles: a full round already
    assert splitk_plan(4160, 1024, 256) == 0              # K too short to split
    for M, N, K in ((4160, 1024, 1024), (4160, 1024, 2752), (4160, 768, 768), (65, 1024, 1024), (65, 768, 3072), (1024, 1024, 4096)):
        ks = splitk_plan(M, N, K)
        tiles = -(-M // 256) * -(-N // 256)
        assert 2 <= ks <= 8 and K // ks >= 256 and tiles * ks <= 512, (M, N, K, ks)


def test_config_shapes_and_sizes():
    cfg = C.MingUniVisionConfig.ming_univision_16b_a3b()
    n_llm = sum(int(torch.tensor(s).prod()) for s in C.llm_param_shapes(cfg.llm_config).values())
    assert abs(n_llm / 1e9 - 16.8) < 0.15                       # 16.8 B (Ling-lite), SURVEY.md
    n_rf = sum(int(torch.tensor(s).prod()) for k, s in
               C.llm_param_shapes(cfg.llm_config, cfg.vishead_diffloss_config).items() if k.startswith("diffloss"))
    assert abs(n_rf / 1e9 - 1.285) < 0.01                       # RF head 1.285 B
    n_tok = sum(int(torch.tensor(s).prod()) for s in C.mingtok_param_shapes(cfg.mingtok_config).values())
    assert abs(n_tok / 1e6 - 697.7) < 0.5                       # MingTok-Vision 697.7 M
    assert C.swiglu_hidden(3072, 4) == 8192 and C.swiglu_hidden(1024) == 2736 and C.swiglu_hidden(768) == 2048
    d = C.MingUniVisionConfig.from_dict(json.loads(cfg.to_json_string()))
    assert d.llm_config.num_experts == 64 and d.vishead_diffloss_config["diffloss_w"] == 3072


def _header_functions():
    src = open(os.path.join(ROOT, "include", "mingnative.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mn_[a-z0-9_]+)\s*\(", src)))


def test_c_abi_exports_every_declared_symbol():
    from ming_univision_amd import _lib
    so = _lib.LIB_PATH
    if not os.path.exists(so):
        pytest.fail("libmingnative.so is not built (run __graft_entry__.build())")
    handle = ctypes.CDLL(so)                      # loads without a GPU
    declared = _header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in mingnative.h but not exported"
        assert name in _lib.SYMBOLS, f"{name} has no ctypes prototype"
    assert handle.mn_version() >= 110
    for name in _lib.SYMBOLS:
        assert name in declared, f"{name} bound in _lib.py but not declared in the header"
    # ... and nothing else: the library is built with -fvisibility=hidden, the A/B hooks of tools/ live in libmingnative_dev.so
    nm = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in nm.splitlines() if " T " in l)
    assert exported == declared, (sorted(set(exported) - set(declared)), sorted(set(declared) - set(exported)))
    assert not any("tune" in n for n in exported)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ming_univision_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("the oracle", "").replace("CPU oracle", ""), fn


def test_ops_fail_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ming_univision_amd import ops
    with pytest.raises(RuntimeError):
        ops.skinny_gemm(torch.zeros(1, 8), torch.zeros(4, 8, dtype=torch.bfloat16))


WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch
from ming_univision_amd.dist_util import ReplicaGroup
g = ReplicaGroup(backend="gloo")
assert g.world == 2
seeds = g.seed(1000)
import time
def work():
    time.sleep(0.05 * (g.rank + 1))       # rank 1 is slower: the reported time must be ITS time
    return g.rank
dt, out = g.timed(work, steps=2)
assert out == g.rank and dt >= 0.19, dt
t = torch.tensor([float(seeds)])
g.dist.all_reduce(t)
assert t.item() == 2001.0                  # distinct per-rank seeds 1000, 1001
if g.rank == 0:
    print(json.dumps({"dt": dt, "total": g.total(256 * 2)}))
g.close()
"""


def test_replica_group_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    import socket
    with socket.socket() as sk:                    # a free port, so concurrent jobs / stale workers cannot collide
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["total"] == 1024 and 0.19 <= res["dt"] < 5.0


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment starts 2 ranks itself (child torch.distributed.run, gloo in the
    --dry-run plumbing mode) and rank 0's line reports them; with a launcher present the flag must agree with WORLD_SIZE."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--tiny", "--dry-run", "--steps", "2"],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # ONE JSON line, from rank 0
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["dry_run"] is True and res["value"] is None
    assert res["ms_per_step"] >= 19.0                            # MAX over ranks: rank 1 sleeps 20 ms per step
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True,
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


TP_WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
import torch.nn.functional as F
from ming_univision_amd import configuration as C
from ming_univision_amd.bailing_moe import pack_experts
from ming_univision_amd.tp import shard_attention, shard_experts, shard_plan, shard_rf_block
from oracle import bailing_ref, rf_ref
from tests.util import llm_sd, load_golden

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = load_golden("llm_tiny")
sd = llm_sd(g["config"], g["rf_config"], g["seed"])            # every rank synthesises the same full weights, keeps its shard
cfg = C.BailingMoeConfig(**g["config"])
ocfg = bailing_ref.LLMConfig(**{k: v for k, v in g["config"].items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
pl = shard_plan(cfg, world)
H, hd, nq, nkv = cfg.hidden_size, cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads
gen = torch.Generator().manual_seed(7)
M = 5
x = torch.randn(1, M, H, generator=gen)
p = "model.layers.0"
err = {}

# ---- attention: this rank's q heads + its KV head, dense over its columns; all-reduce = BailingMoeAttention.forward
wqkv_l, wdense_l = shard_attention(sd[p + ".attention.query_key_value.weight"], sd[p + ".attention.dense.weight"], cfg, rank, world)
assert wqkv_l.shape == ((pl["n_q"] + 2 * pl["n_kv"]) * hd, H) and wdense_l.shape == (H, pl["n_q"] * hd)
qkv = F.linear(x, wqkv_l).view(1, M, pl["n_q"] + 2 * pl["n_kv"], hd)
q, k, v = (t.transpose(1, 2) for t in qkv.split([pl["n_q"], pl["n_kv"], pl["n_kv"]], dim=-2))
pos = torch.arange(M).unsqueeze(0)
cos, sin = bailing_ref.rope_cos_sin(hd, cfg.rope_theta, M)
q, k = bailing_ref.apply_rope(q, k, cos, sin, pos)
rep = pl["n_q"] // pl["n_kv"]
kk, vv = k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1)
w = torch.matmul(q / hd ** 0.5, kk.transpose(2, 3)) + bailing_ref.build_4d_mask(torch.ones(1, M, dtype=torch.long), M, 0)
o = torch.matmul(F.softmax(w, dim=-1, dtype=torch.float32), vv).transpose(1, 2).reshape(1, M, pl["n_q"] * hd)
part = F.linear(o, wdense_l)
dist.all_reduce(part)
full = bailing_ref.attention(x, sd, p + ".attention", ocfg, bailing_ref.build_4d_mask(torch.ones(1, M, dtype=torch.long), M, 0), pos,
                             dict(k=None, v=None))
err["attention"] = float((part - full).abs().max() / full.abs().max())

# ---- experts: its E / world routed experts on the rows routed to them + its slice of the shared expert; all-reduce = moe block
gu, dn = (t.float() for t in pack_experts(sd, p + ".mlp", cfg))      # packs into bf16: lossless, the values are bf16-rounded
gu_l, dn_l, ws_gu, ws_dn = shard_experts(gu, dn, cfg, rank, world)
x2 = x.reshape(M, H)
ti, tw, _ = bailing_ref.gate(x2, sd[p + ".mlp.gate.weight"], ocfg)       # routing replicated on every rank
e0, I = rank * pl["n_experts"], cfg.moe_intermediate_size
part = torch.zeros(M, H)
n_local = 0
for m in range(M):
    for s in range(cfg.num_experts_per_tok):
        e = int(ti[m, s])
        if e0 <= e < e0 + pl["n_experts"]:
            n_local += 1
            y = F.linear(x2[m:m + 1], gu_l[e - e0])
            part[m] += tw[m, s] * F.linear(F.silu(y[:, :I]) * y[:, I:], dn_l[e - e0])[0]
pad = pl["shared_pad"]
y = F.linear(x2, ws_gu)
part += F.linear(F.silu(y[:, :pad]) * y[:, pad:], ws_dn)
dist.all_reduce(part)
full, _ = bailing_ref.moe_block(x, sd, p + ".mlp", ocfg, None)
err["moe"] = float((part - full[0]).abs().max() / full.abs().max())
cnt = torch.tensor([float(n_local)])
dist.all_reduce(cnt)
assert int(cnt.item()) == M * cfg.num_experts_per_tok          # every pick has exactly one owner

# ---- RF ResBlock MLP: its hidden units; all-reduce (+ b3 once) = SwiGLUFFNFused
gen2 = torch.Generator().manual_seed(11)
wd, hid = 64, 128
w12, b12 = torch.randn(2 * hid, wd, generator=gen2) * 0.1, torch.randn(2 * hid, generator=gen2) * 0.1
w3, b3 = torch.randn(wd, hid, generator=gen2) * 0.1, torch.randn(wd, generator=gen2) * 0.1
xr = torch.randn(3, wd, generator=gen2)
a, bb, c = shard_rf_block(w12, b12, w3, rank, world)
n = hid // world
y = F.linear(xr, a, bb)
part = F.linear(F.silu(y[:, :n]) * y[:, n:], c)
dist.all_reduce(part)
yf = F.linear(xr, w12, b12)
full = F.linear(F.silu(yf[:, :hid]) * yf[:, hid:], w3, b3)
err["rf_block"] = float((part + b3 - full).abs().max() / full.abs().max())
if rank == 0:
    print(json.dumps(err))
dist.barrier()
dist.destroy_process_group()
"""


def test_tp_partitioning_world2_gloo(tmp_path):
    """The tensor / expert-parallel partitioning (ming_univision_amd.tp.shard_*: heads, KV replication, expert window, shared-expert
    slice with zero padding, RF hidden split) on two gloo ranks: each rank computes its partial with plain fp32 math on ITS shard,
    the all-reduce over the process group must equal the oracle's unsharded attention / MoE block / SwiGLU block."""
    script = tmp_path / "tp_worker.py"
    script.write_text(TP_WORKER % ROOT)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    err = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert set(err) == {"attention", "moe", "rf_block"} and max(err.values()) < 1e-5, err


def test_tp_shard_plan_16b_a3b():
    from ming_univision_amd.tp import shard_plan
    cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
    assert shard_plan(cfg, 8, rf_hidden=8192) == dict(n_q=2, n_kv=1, n_experts=8, shared=352, shared_pad=384, rf_hidden=1024)
    assert shard_plan(cfg, 4, rf_hidden=8192) == dict(n_q=4, n_kv=1, n_experts=16, shared=704, shared_pad=704, rf_hidden=2048)
    assert shard_plan(cfg, 2)["n_kv"] == 2 and shard_plan(cfg, 1)["shared_pad"] == 2816
    with pytest.raises(ValueError):
        shard_plan(cfg, 3)
    with pytest.raises(ValueError):
        shard_plan(cfg, 32)                                        # 16 q heads do not split 32 ways


@pytest.mark.parametrize("world", [2, 4, 8])
def test_tp_nf4_shards_are_pure_indexing(world):
    """int4 (NF4) TP shards on CPU tensors: the shared expert's 64-unit absmax blocks are dealt out whole, so the ranks' slices —
    codes (two per byte) and absmax values — tile the packed expert tensors exactly (nothing re-quantised, nothing lost), padding
    carries zero absmax, and the routed experts are split by expert."""
    from ming_univision_amd.tp import nf4_shared_units, shard_experts_nf4, shard_plan
    cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
    E, S, I, H = 4 * world, cfg.num_shared_experts, cfg.moe_intermediate_size, 256          # (a narrow H: the split runs over units, not over H)
    cfg.num_experts = E
    g = torch.Generator().manual_seed(world)
    gu = torch.randint(0, 256, (E + S, 2 * I, H // 2), generator=g, dtype=torch.uint8)
    dn = torch.randint(0, 256, (E + S, H, I // 2), generator=g, dtype=torch.uint8)
    gus = torch.rand(E + S, 2 * I, H // 64, generator=g) + 0.5
    dns = torch.rand(E + S, H, I // 64, generator=g) + 0.5
    pad = shard_plan(cfg, world)["shared_pad"]
    units = [nf4_shared_units(cfg, r, world) for r in range(world)]
    assert units[0][0] == 0 and all(units[r][0] + units[r][1] == units[r + 1][0] for r in range(world - 1))
    assert units[-1][0] + units[-1][1] == S * I and all(n % 64 == 0 and u0 % 64 == 0 and n <= pad for u0, n in units)
    if world == 8:
        assert [n for _, n in units] == [384] * 4 + [320] * 4
    sg_full = torch.cat([gu[E + s, :I] for s in range(S)], 0); su_full = torch.cat([gu[E + s, I:] for s in range(S)], 0)
    sd_full = torch.cat([dn[E + s] for s in range(S)], 1); sa_full = torch.cat([dns[E + s] for s in range(S)], 1)
    for r in range(world):
        o = shard_experts_nf4(gu, dn, gus, dns, cfg, r, world)
        u0, n = units[r]
        ne = E // world
        assert torch.equal(o[0], gu[r * ne:(r + 1) * ne]) and torch.equal(o[5], dns[r * ne:(r + 1) * ne])
        assert o[2].shape == (2 * pad, H // 2) and o[6].shape == (2 * pad, H // 64) and o[3].shape == (H, pad // 2) and o[7].shape == (H, pad // 64)
        assert torch.equal(o[2][:n], sg_full[u0:u0 + n]) and torch.equal(o[2][pad:pad + n], su_full[u0:u0 + n])
        assert torch.equal(o[3][:, :n // 2], sd_full[:, u0 // 2:(u0 + n) // 2]) and torch.equal(o[7][:, :n // 64], sa_full[:, u0 // 64:(u0 + n) // 64])
        assert float(o[6][n:pad].abs().sum()) == 0 and float(o[6][pad + n:].abs().sum()) == 0 and float(o[7][:, n // 64:].abs().sum()) == 0


def test_tp_vocab_parallel_pick_rule():
    """The reduce of the ranks' (logit, id) pairs of a vocabulary-split lm_head: largest logit, lowest id among equals."""
    from ming_univision_amd.tp import pick_best
    vals = torch.tensor([[1.0, 5.0, 2.0], [3.0, 5.0, 2.0], [3.0, 4.0, -1.0]])
    idxs = torch.tensor([[7, 900, 11], [1000, 20, 12], [40, 30, 13]])
    assert pick_best(idxs, vals).tolist() == [40, 20, 11]


def test_hf_tokenizer_adapter_with_a_real_tokenizer_json(tmp_path):
    """infer.HFTokenizerAdapter over a `tokenizers` tokenizer.json (what MingUniVisionInfer loads next to a checkpoint): a small
    WordLevel tokenizer that carries the reference's special tokens at the reference's ids (tokenizer_config.json: 126340 / 126341
    `<role>` / `</role>`, 126346-126348).  The processor's chat template, image-token expansion and CFG masks
    (processing_bailingmm.py:282-361) must behave through it exactly as through the stand-in tokenizer: the uncond mask is zero
    strictly between the last HUMAN tag and the next ASSISTANT tag, the text-uncond mask keeps the image tokens there."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from ming_univision_amd.infer import HFTokenizerAdapter
    from ming_univision_amd.processing import SPECIAL_TOKEN_IDS
    words = ["[UNK]", "HUMAN", "ASSISTANT", "SYSTEM", "describe", "this", "picture", "please", "a", "cat", "hello", "You", "are", "."]
    vocab = {w: i for i, w in enumerate(words)}
    vocab.update(SPECIAL_TOKEN_IDS)
    tk = Tokenizer(models.WordLevel(vocab=vocab, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    tk.add_special_tokens(list(SPECIAL_TOKEN_IDS))
    path = tmp_path / "tokenizer.json"
    tk.save(str(path))
    ad = HFTokenizerAdapter(str(path))
    assert ad.convert_tokens_to_ids("<imagePatch>") == 126346 and ad.convert_tokens_to_ids("<role>") == 126340
    assert ad.encode("<role>HUMAN</role>") == [126340, vocab["HUMAN"], 126341]
    proc = BailingMMProcessor(tokenizer=ad)
    conv = [{"role": "HUMAN", "content": [{"type": "image", "image": "x.png"}, {"type": "text", "text": "describe this picture please"}]}]
    text = proc.apply_chat_template(conv, add_generation_prompt=True)
    out = proc(images=[torch.zeros(3, 64, 64)], text=[text], image_patch_size=32)
    ids = out["input_ids"][0].tolist()
    assert ids.count(126346) == 4 and ids.count(126347) == 1 and ids.count(126348) == 1
    h = [i for i in range(len(ids) - 2) if ids[i:i + 3] == [126340, vocab["HUMAN"], 126341]][-1]
    a = [i for i in range(len(ids) - 2) if ids[i:i + 3] == [126340, vocab["ASSISTANT"], 126341]][-1]
    unc, tunc = out["uncond_attention_mask"][0].tolist(), out["text_uncond_attention_mask"][0].tolist()
    for i, t in enumerate(ids):
        inside = h + 3 <= i < a
        assert unc[i] == (0 if inside else 1), i
        assert tunc[i] == (0 if inside and t not in (126346, 126347, 126348) else 1), i
    assert any(inside for inside in (h + 3 <= i < a for i in range(len(ids))))
    assert ad.decode([vocab["a"], vocab["cat"], 126081], skip_special_tokens=True).split() == ["a", "cat"]
    assert ad.batch_decode([[vocab["hello"]], [vocab["cat"], vocab["."]]]) == ["hello", "cat ."]


def test_prefill_pass_spans_partition_the_stacked_sequences():
    """pass_spans (the span tables of mn_llm_step_spans): whatever the pass size, the spans of all passes tile every sequence exactly
    once, in order, with `past` = the slots the sequence already wrote; rows inside a pass are contiguous and in stack order."""
    from ming_univision_amd.bailing_moe import pass_spans
    import random
    rnd = random.Random(0)
    for trial in range(200):
        n = rnd.randint(1, 6)
        lens = [rnd.randint(1, 300) for _ in range(n)]
        seqs = rnd.sample(range(20), n)
        past = rnd.choice([0, 0, 17, 200])
        step = rnd.choice([1, 7, 64, 128, 2048])
        starts = [0]
        for L in lens:
            starts.append(starts[-1] + L)
        covered = {s: past for s in seqs}                        # next expected slot of every sequence
        for r0 in range(0, starts[-1], step):
            r1 = min(starts[-1], r0 + step)
            spans = pass_spans(starts, seqs, past, r0, r1)
            assert sum(sp[2] for sp in spans) == r1 - r0
            row = 0
            for seq, first, rows, p in spans:
                assert first == row and rows >= 1
                assert p == covered[seq], (trial, seq, p, covered[seq])
                covered[seq] += rows
                row += rows
        assert all(covered[s] == past + L for s, L in zip(seqs, lens))


def test_sampling_arguments_are_validated_on_the_host():
    """generate()'s sampling kwargs (the ones the reference forwards to HF generate) are checked before anything touches the GPU."""
    import inspect
    from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration as M
    sig = inspect.signature(M.generate)
    for k, d in (("do_sample", False), ("temperature", 1.0), ("top_k", 50), ("top_p", 1.0), ("generator", None)):
        assert sig.parameters[k].default == d               # HF's defaults (mingunivision/config.json:30,103-109)
    sig = inspect.signature(M.generate_text_batch)
    assert sig.parameters["do_sample"].default is False and sig.parameters["top_k"].default == 50
    # HF kwargs the reference would forward (modeling_bailingmm.py:249-262): neutral values pass, result-changing ones raise
    from ming_univision_amd.modeling import check_sampling_args, filter_generate_kwargs
    assert filter_generate_kwargs({}, 7, "t") == {7}
    assert filter_generate_kwargs(dict(pad_token_id=0, use_cache=True, num_beams=1, repetition_penalty=1.0, eos_token_id=[3, 9]), 7, "t") == {3, 9}
    assert filter_generate_kwargs(dict(logits_processor=[], streamer=None), 7, "t") == {7}
    for bad in (dict(num_beams=4), dict(repetition_penalty=1.2), dict(min_new_tokens=5), dict(frobnicate=1), dict(return_dict_in_generate=True)):
        with pytest.raises(TypeError):
            filter_generate_kwargs(bad, 7, "t")
    check_sampling_args(True, 0.7, 2048, 0.9, "t")
    check_sampling_args(False, 0.0, 5000, 2.0, "t")            # greedy: the sampling values are not looked at
    for bad in ((0.0, 50, 1.0), (1.0, -1, 1.0), (1.0, 50, 0.0), (1.0, 50, 1.5), (1.0, 2049, 1.0)):
        with pytest.raises(ValueError):
            check_sampling_args(True, *bad, "t")


def test_ctypes_structures_match_the_library_layout():
    """`_lib.py`'s ctypes Structures against the C structs of the built library — sizeof through mn_sizeof_*, every
    field offset through mn_struct_layout (no GPU needed) — and the guard itself: a Structure with one more field must be refused."""
    from ming_univision_amd import _lib
    handle = ctypes.CDLL(_lib.LIB_PATH)
    _lib.check_struct_layouts(handle)
    for sid, sizer, klass in _lib.STRUCTS:
        fn = getattr(handle, sizer)
        fn.restype = ctypes.c_size_t
        assert fn() == ctypes.sizeof(klass) and fn() % 8 == 0, (sizer, fn(), ctypes.sizeof(klass))
    handle.mn_struct_layout.restype = ctypes.c_int
    handle.mn_struct_layout.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.c_int]
    assert handle.mn_struct_layout(99, (ctypes.c_size_t * 4)(), 4) < 0

    class Grown(ctypes.Structure):                     # a binding that added a field the library does not know
        _fields_ = list(_lib.TpComm._fields_) + [("extra", ctypes.c_int64)]
    real = _lib.STRUCTS
    try:
        _lib.STRUCTS = ((4, "mn_sizeof_tp_comm", Grown),)
        with pytest.raises(RuntimeError):
            _lib.check_struct_layouts(handle)

        class Swapped(ctypes.Structure):               # same size, two fields in the other order
            _fields_ = [("world", ctypes.c_int32), ("rank", ctypes.c_int32)] + list(_lib.TpComm._fields_[2:5]) + \
                       [("epoch", ctypes.c_uint32), ("rows_cap", ctypes.c_int32)] + list(_lib.TpComm._fields_[7:])
        assert ctypes.sizeof(Swapped) == ctypes.sizeof(_lib.TpComm)
        _lib.STRUCTS = ((4, "mn_sizeof_tp_comm", Swapped),)
        _lib.check_struct_layouts(handle)              # offsets equal: field NAMES are not part of a C layout ...
    finally:
        _lib.STRUCTS = real
