"""BASELINE configs[2] at the reference's shape — 1024^2 image -> 1 024 image tokens, 1 058-token prompt — prefill only (generate with
max_new_tokens = 1), fp32-class regime (default), 3 timed runs.  Meant to run under rocprofv3 --kernel-trace (tools/prof_c3_prefill.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd import configuration as C
from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration
cfg = C.MingUniVisionConfig.ming_univision_16b_a3b()
model = MingUniVisionForConditionalGeneration(cfg, device="cuda", seed=0, t_max=1400)
ids = torch.randint(0, 100000, (1, 12 + 1026 + 20))
ids[0, 12] = cfg.llm_config.image_start_token
ids[0, 13:13 + 1024] = cfg.llm_config.image_patch_token
ids[0, 13 + 1024] = 126348
px = torch.rand(1, 3, 1024, 1024) * 2 - 1
model.understanding_precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
for it in range(4):
    model.reset_inner_state()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=px, max_new_tokens=1)
    torch.cuda.synchronize()
    print(f"prefill incl. vision: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
