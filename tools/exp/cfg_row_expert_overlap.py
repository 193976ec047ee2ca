"""How many DISTINCT routed experts the CFG rows of one image select per layer during image generation (bench.py's batch-1 call shape,
synthetic 16B-A3B weights): the last decode step's routing of every layer through mn_llm_route_capture.  The (row, expert) pair launches
read rows x top_k expert matrices per layer; a route that reads every distinct expert once reads `distinct` of them.
    python tools/exp/cfg_row_expert_overlap.py"""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from ming_univision_amd._lib import lib, ptr, check
dev = torch.device("cuda", 0)
args = argparse.Namespace(tiny=False, tokens=64, layers=None, prompt_len=40, images=1, cfg_rows=3, weights="bf16")
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
g = torch.Generator(device=dev).manual_seed(1)
prompt = torch.randint(0, 100000, (1, 40), device=dev, generator=g)
noises = torch.randn(1, 65, 32, device=dev, generator=g)
L, K = cfg.num_hidden_layers, cfg.num_experts_per_tok
for rows in (2, 3):
    n_slot = K + 2
    routes = torch.full((L, rows, n_slot), -1, dtype=torch.int32, device=dev)
    check(lib().mn_llm_route_capture(ptr(routes)), "mn_llm_route_capture")
    try:
        bench.one_image(cfg, dec, rf, tok, prompt, noises, 1, rows)
        torch.cuda.synchronize()
    finally:
        check(lib().mn_llm_route_capture(None), "mn_llm_route_capture")
    r = routes.cpu()
    distinct = [len(set(int(e) for e in r[l, :, :].flatten().tolist() if 0 <= e < cfg.num_experts)) for l in range(L)]
    print(f"{rows} CFG rows: routed pairs per layer {rows * K}, distinct routed experts per layer: mean {sum(distinct) / L:.2f}, min {min(distinct)}, max {max(distinct)}  {distinct}", flush=True)
