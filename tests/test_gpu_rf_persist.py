"""The persistent per-Euler-step launch of the RF head at the reference's call shape (stream_kc.hip: every ResBlock of a step in one
launch, grid barriers between the phases).  Parity with the oracle is covered where the other routes are (test_gpu_fullwidth.py,
test_gpu_rf_shapes.py, test_gpu_fp8.py run this launch at 2 rows); here: what only a resident, barrier-synchronised launch can get
wrong — state left in the barrier words between calls, and two such launches meeting on the device from two streams."""
import pytest
import torch

from ming_univision_amd import configuration as C
from ming_univision_amd.synth import synth_tensor

pytestmark = pytest.mark.gpu

LLM_HIDDEN = 2048


def _head(weights="bf16", seed=7):
    from ming_univision_amd.rf_head import RectifiedFlowHead
    w, d, steps, mult = 3072, 12, 16, 4                               # the 16B-A3B head: 256 + 192 workgroups per phase pair
    rf_cfg = dict(diffloss_w=w, diffloss_d=d, num_sampling_steps=str(steps), gen_method=f"flow_matching_swiglu-{mult}",
                  vis_head_arch="linear2-norm")
    shapes = {"vis_head.0.weight": (w, LLM_HIDDEN), "vis_head.0.bias": (w,), "vis_head.1.weight": (w,), "vis_head.1.bias": (w,)}
    shapes.update(C.rf_param_shapes(w, d, w, 32, mult))
    sd = {k: synth_tensor(k, s, seed, "cuda", torch.bfloat16) for k, s in shapes.items()}
    rf = RectifiedFlowHead(sd, LLM_HIDDEN, rf_cfg, weights=weights)
    rf._sd = sd
    return rf


@pytest.mark.parametrize("weights", ["bf16", "fp8"])
def test_repeated_calls_and_two_streams_give_the_same_bits(weights):
    rf = _head(weights)
    g = torch.Generator(device="cuda").manual_seed(3)
    cases = [(torch.randn(2, LLM_HIDDEN, device="cuda", generator=g), torch.randn(1, 32, device="cuda", generator=g)) for _ in range(4)]
    ref = [rf.sample(h, n, n_images=1).clone() for h, n in cases]
    for r in ref:
        assert torch.isfinite(r).all()
    # the same workspace (barrier words included) again and again
    for _ in range(3):
        for (h, n), r in zip(cases, ref):
            assert torch.equal(rf.sample(h, n, n_images=1), r)
    # two streams, each with its own workspace, enqueued back to back: the launches must be ordered on the device, not interleaved
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = {}
    for rep in range(3):
        for i, (h, n) in enumerate(cases):
            st = s1 if i % 2 == 0 else s2
            with torch.cuda.stream(st):
                outs[(rep, i)] = rf.sample(h, n, n_images=1)
    torch.cuda.synchronize()
    for (rep, i), o in outs.items():
        assert torch.equal(o, ref[i]), (rep, i)


def test_one_row_and_three_rows_keep_their_routes():
    """1 row (no CFG) and 3 rows (image editing: 3 CFG rows) around the 2-row launch: finite, deterministic results."""
    rf = _head("fp8")
    g = torch.Generator(device="cuda").manual_seed(5)
    for rows in (1, 3):
        h = torch.randn(rows, LLM_HIDDEN, device="cuda", generator=g)
        n = torch.randn(1, 32, device="cuda", generator=g)
        a = rf.sample(h, n, n_images=1).clone()
        assert torch.isfinite(a).all() and torch.equal(rf.sample(h, n, n_images=1), a)


def test_bf16_one_row_takes_the_matrix_core_launches_and_matches_the_oracle():
    """bf16 at ONE row used the fp32-FMA kernels; where the shape runs K-complete it now takes the same launches as 2 rows."""
    from oracle import rf_ref
    from tests.util import rel_err
    rf = _head("bf16")
    osd = {k: v.float().cpu() for k, v in rf._sd.items()}
    rsd = {k[len("diffloss."):]: v for k, v in osd.items() if k.startswith("diffloss.")}
    g = torch.Generator().manual_seed(11)
    h, n = torch.randn(1, LLM_HIDDEN, generator=g), torch.randn(1, 32, generator=g)
    torch.set_num_threads(min(64, max(torch.get_num_threads(), 16)))
    ref = rf_ref.sample(rf_ref.vis_head(h, osd), n, rsd, steps=16)[0]
    got = rf.sample(h.cuda(), n.cuda(), n_images=1)
    e = rel_err(got, ref)
    print(f"bf16 head, 1 row: {e:.2e}")
    assert e < 1e-3


def test_env_switch_keeps_the_launches_and_agrees():
    """MINGNATIVE_RF_PERSIST=0 (two processes on one GPU) keeps the two launches per ResBlock: same latents to 1e-5 (the whole-sampler
    launch computes the final layer in fp32 FMAs, the launches in hi/lo MFMA slabs)."""
    import json
    import os
    import subprocess
    import sys
    code = (
        "import json, torch, sys\n"
        "sys.path.insert(0, %r)\n"
        "from tests.test_gpu_rf_persist import _head, LLM_HIDDEN\n"
        "rf = _head('bf16')\n"
        "g = torch.Generator(device='cuda').manual_seed(3)\n"
        "h = torch.randn(2, LLM_HIDDEN, device='cuda', generator=g); n = torch.randn(1, 32, device='cuda', generator=g)\n"
        "print('LAT', json.dumps(rf.sample(h, n, n_images=1).flatten().tolist()))\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, MINGNATIVE_RF_PERSIST="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lat0 = torch.tensor(json.loads([l for l in out.stdout.splitlines() if l.startswith("LAT ")][0][4:]))
    rf = _head("bf16")
    g = torch.Generator(device="cuda").manual_seed(3)
    h = torch.randn(2, LLM_HIDDEN, device="cuda", generator=g); n = torch.randn(1, 32, device="cuda", generator=g)
    lat1 = rf.sample(h, n, n_images=1).flatten().cpu()
    d = float((lat1 - lat0).abs().max() / lat0.abs().max())
    print("persistent launch vs MINGNATIVE_RF_PERSIST=0: %.2e" % d)
    assert d < 1e-5 and d > 0.0          # (> 0: the other process really took the other route)


def test_sampler_inside_a_graph_capture_takes_the_launches():
    """A captured stream gets the two launches per ResBlock (the persistent launch orders itself against other streams with event
    calls, which a capture must not see): capture + replay give the eager result to 1e-5."""
    rf = _head("bf16")
    g = torch.Generator(device="cuda").manual_seed(9)
    h = torch.randn(2, LLM_HIDDEN, device="cuda", generator=g)
    n = torch.randn(1, 32, device="cuda", generator=g)
    out = torch.empty(1, 32, device="cuda")
    eager = rf.sample(h, n, n_images=1).clone()
    rf.sample(h, n, n_images=1, out=out)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        rf.sample(h, n, n_images=1, out=out)
    out.zero_()
    gr.replay()
    torch.cuda.synchronize()
    d = float((out - eager).abs().max() / eager.abs().max())
    print("captured sampler vs eager persistent launch: %.2e" % d)
    assert torch.isfinite(out).all() and d < 1e-5


def test_a_late_resident_workgroup_poisons_the_result_and_raises_on_the_host():
    """one workgroup of the persistent launch arrives after the others' barrier patience has run out (what a co-tenant
    process or a CU mask does).  Every other workgroup times out, the straggler itself does NOT — it must still see the error word:
    the latents are NaN (never finite-but-wrong), the caller's status word is raised and RectifiedFlowHead.check_err() names the
    cause; the next call (no fault) is bit-identical to an undisturbed one.  Runs on libmingnative_dev.so (the fault hook)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "import tools.devlib\n"
        "from ming_univision_amd._lib import lib\n"
        "from tests.test_gpu_rf_persist import _head, LLM_HIDDEN\n"
        "rf = _head('bf16')\n"
        "g = torch.Generator(device='cuda').manual_seed(3)\n"
        "h = torch.randn(2, LLM_HIDDEN, device='cuda', generator=g); n = torch.randn(1, 32, device='cuda', generator=g)\n"
        "good = rf.sample(h, n, n_images=1).clone(); rf.check_err()\n"
        "assert torch.isfinite(good).all()\n"
        "for wg in (0, 5, 255):\n"
        "    lib().mn_rf_kc_fault(wg, 3)\n"
        "    bad = rf.sample(h, n, n_images=1).clone()\n"
        "    torch.cuda.synchronize()\n"
        "    assert torch.isnan(bad).all(), (wg, bad)\n"
        "    try:\n"
        "        rf.check_err(); raise SystemExit('check_err did not raise for wg %%d' %% wg)\n"
        "    except RuntimeError as e:\n"
        "        assert 'MINGNATIVE_RF_PERSIST=0' in str(e)\n"
        "    lib().mn_rf_kc_fault(-1, 0)\n"
        "    again = rf.sample(h, n, n_images=1); rf.check_err()\n"
        "    assert torch.equal(again, good), wg\n"
        "print('FAULT_OK')\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "FAULT_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])


@pytest.mark.parametrize("weights", ["bf16", "fp8", "int8"])
def test_three_rows_take_the_k_complete_launches_and_match_the_oracle(weights):
    """Round 6: 3 rows — the editing shape, RectifiedFlowLoss.sample's 3-way CFG (diff_loss_rf_swiglu.py:143-150) — ran
    the three-launch chain because w3's operand image + 8 KiB weight tiles per wave exceed the 160 KiB of LDS; with half tiles (4 KiB per
    wave, chunks parked in two halves) the K-complete launches and the whole-sampler persistent launch take 3 rows too — bf16 heads:
    the byte formats keep the chain at 3 rows (a masked half-tile park measured slower than it).  Every mode against the oracle's
    sample() fed the mode's de-quantised weights, 1e-3; the persistent launch (default) and the launch form of the same bodies (what
    MINGNATIVE_RF_PERSIST=0 gives; here: inside a graph capture) agree to 1e-5."""
    from oracle import rf_ref
    from tests.util import rel_err
    rf = _head(weights)
    assert rf.stream_fmt == weights
    osd = {k: v.float().cpu() for k, v in rf._sd.items()}
    if weights != "bf16":
        osd.update({k: v.float().cpu() for k, v in rf.dequantized_blocks().items()})
    rsd = {k[len("diffloss."):]: v for k, v in osd.items() if k.startswith("diffloss.")}
    g = torch.Generator().manual_seed(23)
    h, n = torch.randn(3, LLM_HIDDEN, generator=g), torch.randn(1, 32, generator=g)
    torch.set_num_threads(min(64, max(torch.get_num_threads(), 16)))
    ref = rf_ref.sample(rf_ref.vis_head(h, osd), n, rsd, steps=16)[0]
    hd, nd = h.cuda(), n.cuda()
    got = rf.sample(hd, nd, n_images=1).clone()
    rf.check_err()
    e = rel_err(got, ref)
    assert torch.equal(rf.sample(hd, nd, n_images=1), got)              # deterministic, barrier words reusable
    # the launch form of the same bodies: a captured stream takes the two launches per ResBlock (no event calls inside a capture)
    out = torch.empty(1, 32, device="cuda")
    rf.sample(hd, nd, n_images=1, out=out)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        rf.sample(hd, nd, n_images=1, out=out)
    out.zero_()
    gr.replay()
    torch.cuda.synchronize()
    d = rel_err(out, got)
    print(f"{weights} head, 3 rows: vs oracle {e:.2e}; launches vs the persistent launch {d:.2e}")
    assert e < 1e-3 and d < 1e-5
    # 2 images x 3 rows stays on the chain (6 rows) and gives the same latents for image 0
    two = rf.sample(torch.cat([hd, hd]), torch.cat([nd, nd]), n_images=2)
    assert rel_err(two[0], ref) < 1e-3 and rel_err(two[1], ref) < 1e-3
