import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_models import *
from ming_univision_amd.bailing_moe import BailingMoeDecoder, generate_image, generate_images
from ming_univision_amd.mingtok import MingTok
from ming_univision_amd.rf_head import RectifiedFlowHead
g = load_golden("genimg_tiny")
sd = llm_sd(g["llm_config"], g["rf_config"], g["seed"])
dsd = to_dev(sd)
cfg = C.BailingMoeConfig(**g["llm_config"])
B, R = 21, 3
dec = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=32, n_seq=B * R)
rf = RectifiedFlowHead(dsd, cfg.hidden_size, g["rf_config"])
lsd = to_dev(synth_state_dict(C.linear_proj_param_shapes(128, 256, 2), g["seed"]))
tok = MingTok(C.MingTokConfig(**g["mingtok_config"]), state_dict=mingtok_sd(g["mingtok_config"], g["seed"]),
              linear_proj=[(lsd["linear_proj.0.weight"], lsd["linear_proj.0.bias"]), (lsd["linear_proj.2.weight"], lsd["linear_proj.2.bias"])])
gen = torch.Generator().manual_seed(3)
T = g["ids"].shape[1]
prompts = [g["ids"][0]] + [torch.randint(0, 400, (T - i % 4,), generator=gen) for i in range(1, B)]
noises = torch.randn(B, cfg.num_image_tokens_for_gen + 1, 32, generator=gen)
noises[0] = g["noises"]
start = dec.embed(torch.tensor([cfg.image_start_token]).cuda())
def masks(n):
    am = torch.ones(1, n + 1, dtype=torch.long)
    un = am.clone(); un[0, 2:n - 2] = 0
    tu = am.clone(); tu[0, 2:4] = 0
    return am, un, tu
singles = []
for i in range(B):
    dec.prefill(dec.embed(prompts[i].cuda()), seq=0, past=0)
    am, un, tu = (g["mask"], g["uncond"], g["rows3_tuncond"]) if i == 0 else masks(prompts[i].numel())
    o = generate_image(dec, rf, tok, start, prompts[i].numel(), am, un, tu, noises[i].cuda())
    singles.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()})
for nb in (21, 10, 5):
    ams, uns, tus = [], [], []
    for i in range(nb):
        dec.prefill(dec.embed(prompts[i].cuda()), seq=i * R, past=0)
        am, un, tu = (g["mask"], g["uncond"], g["rows3_tuncond"]) if i == 0 else masks(prompts[i].numel())
        ams.append(am); uns.append(un); tus.append(tu)
    out = generate_images(dec, rf, tok, start, [p.numel() for p in prompts[:nb]], ams, uns, tus, noises[:nb].cuda())
    print("batch", nb, " ".join(f"{i}:{rel_err(out['latents'][i], singles[i]['latents']):.1e}/{rel_err(out['last_hidden'][i*R:(i+1)*R], singles[i]['last_hidden']):.1e}" for i in range(nb)))
