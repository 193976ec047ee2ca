"""Shape robustness of the rectified-flow head.  `vishead_diffloss_config` of the real checkpoint is not known
here — `setup_vishead_diffloss` is parametrised (modeling_bailing_moe.py:1559-1584: diffloss_w, diffloss_d, num_sampling_steps,
gen_method "flow_matching_swiglu-<mlp_mult>") and the builder only ever ran w in {64, 384, 3072}, d in {1, 2, 12}.  First contact with
real weights at another width / depth / step count must not mis-launch: every (w, d, steps, mlp_mult) below runs `RectifiedFlowHead.
sample` against `oracle/rf_ref.sample` at 2 and 3 rows (the fused chain of the reference's call shape), 64 rows (weight-streaming
route, K-loop form) and 130 rows (wide route) — or, where a route does not exist for the shape (the wide route needs every width to be
a multiple of 64; `max_rows()` says so), the call must raise a clear error instead of launching."""
import pytest
import torch

from ming_univision_amd import configuration as C
from ming_univision_amd.synth import synth_tensor
from tests.util import rel_err, row_errs

pytestmark = pytest.mark.gpu

TOL = 1e-3
LLM_HIDDEN = 2048
SHAPES = [(1024, 6, 8, 4), (1536, 8, 25, 4), (2048, 8, 16, 4), (4096, 16, 16, 4), (3072, 12, 50, 2)]


def _head(w, d, steps, mult, seed):
    from ming_univision_amd.rf_head import RectifiedFlowHead
    rf_cfg = dict(diffloss_w=w, diffloss_d=d, num_sampling_steps=str(steps), gen_method=f"flow_matching_swiglu-{mult}",
                  vis_head_arch="linear2-norm")
    shapes = {"vis_head.0.weight": (w, LLM_HIDDEN), "vis_head.0.bias": (w,), "vis_head.1.weight": (w,), "vis_head.1.bias": (w,)}
    shapes.update(C.rf_param_shapes(w, d, w, 32, mult))
    sd = {k: synth_tensor(k, s, seed, "cuda", torch.bfloat16) for k, s in shapes.items()}
    return RectifiedFlowHead(sd, LLM_HIDDEN, rf_cfg), sd


@pytest.mark.parametrize("w,d,steps,mult", SHAPES)
def test_rf_head_shapes_on_every_route_vs_oracle(w, d, steps, mult):
    from oracle import rf_ref
    torch.set_num_threads(min(64, max(torch.get_num_threads(), 16)))
    rf, sd = _head(w, d, steps, mult, seed=41)
    assert rf.hidden == C.swiglu_hidden(w, mult) and rf.steps == steps
    osd = {k: v.float().cpu() for k, v in sd.items()}
    rsd = {k[len("diffloss."):]: v for k, v in osd.items() if k.startswith("diffloss.")}
    g = torch.Generator().manual_seed(w + d)
    n_img = 65
    hidden = torch.randn(2 * n_img, LLM_HIDDEN, generator=g)
    noise = torch.randn(n_img, 32, generator=g)
    hidden3 = torch.randn(3, LLM_HIDDEN, generator=g)
    noise3 = torch.randn(1, 32, generator=g)

    def oracle(h_rows, nz):
        return rf_ref.sample(rf_ref.vis_head(h_rows, osd), nz, rsd, steps=steps)[0]
    wide_ok = rf.max_rows() >= 130
    assert wide_ok == (w % 64 == 0 and rf.hidden % 64 == 0), (rf.max_rows(), w, rf.hidden)
    ref = {i: oracle(hidden[2 * i:2 * i + 2], noise[i:i + 1]) for i in ((0, 31, 64) if wide_ok else (0, 31))}   # images are independent
    ref3 = oracle(hidden3, noise3)
    hd, nd = hidden.cuda(), noise.cuda()
    report = []
    for rows, imgs in ((2, (0,)), (64, (0, 31)), (130, (0, 31, 64))):
        n = rows // 2
        if rows > rf.max_rows():
            with pytest.raises(RuntimeError):                    # a route that does not exist for this shape refuses, loudly
                rf.sample(hd[:rows].contiguous(), nd[:n].contiguous(), n_images=n)
            report.append(f"{rows} rows: refused (no wide route: hidden {rf.hidden} % 64 = {rf.hidden % 64})")
            continue
        out = rf.sample(hd[:rows].contiguous(), nd[:n].contiguous(), n_images=n)
        assert torch.isfinite(out).all()
        e = max(rel_err(out[i], ref[i]) for i in imgs)
        report.append(f"{rows} rows: {e:.2e}")
        assert e < TOL, (rows, e)
    out3 = rf.sample(hidden3.cuda(), noise3.cuda(), n_images=1)
    e3 = rel_err(out3[0], ref3)
    report.append(f"3 rows: {e3:.2e}")
    print(f"RF head w={w} d={d} steps={steps} mlp_mult={mult} (hidden {rf.hidden}) vs oracle — " + " | ".join(report))
    assert e3 < TOL
