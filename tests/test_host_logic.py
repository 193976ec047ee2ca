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
