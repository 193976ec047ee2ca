"""C3 prefill cost split: MingTok (encoder + semantic decoder + linear_proj) on n x 1024^2 images, and the bf16 MFMA prefill of
n stacked 1058-token prompts (16B-A3B shapes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd import configuration as C
from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration
cfg = C.MingUniVisionConfig.ming_univision_16b_a3b()
model = MingUniVisionForConditionalGeneration(cfg, device="cuda", seed=0, t_max=1152)
model.model.ensure_sequences(8)
def wall(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for nb in (1, 2, 4, 8):
    px = (torch.rand(nb, 3, 1024, 1024) * 2 - 1).cuda()
    t = wall(lambda: model.extract_image_feature(px))
    print(f"vision tower, {nb} x 1024^2: {t:7.2f} ms = {t / nb:6.2f} ms per image", flush=True)
for nb in (1, 2, 4, 7):
    e = [torch.randn(1058, 2048, device="cuda") * 0.02 for _ in range(nb)]
    t = wall(lambda: model.model.prefill_mfma_many(e, list(range(nb)), past=0))
    print(f"prefill_mfma_many, {nb} x 1058 tokens: {t:7.2f} ms = {t / nb:6.2f} ms per prompt", flush=True)
