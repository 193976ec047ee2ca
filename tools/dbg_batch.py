import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd import configuration as C
from ming_univision_amd.synth import synth_state_dict
from tests.util import llm_sd, load_golden, mingtok_sd
from ming_univision_amd.bailing_moe import BailingMoeDecoder, generate_images
from ming_univision_amd.mingtok import MingTok
from ming_univision_amd.rf_head import RectifiedFlowHead
B = int(sys.argv[1]); R = int(sys.argv[2])
g = load_golden("genimg_tiny")
to_dev = lambda sd: {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}
dsd = to_dev(llm_sd(g["llm_config"], g["rf_config"], g["seed"]))
cfg = C.BailingMoeConfig(**g["llm_config"])
dec = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=32, n_seq=B * R)
rf = RectifiedFlowHead(dsd, cfg.hidden_size, g["rf_config"])
lsd = to_dev(synth_state_dict(C.linear_proj_param_shapes(128, 256, 2), g["seed"]))
tok = MingTok(C.MingTokConfig(**g["mingtok_config"]), state_dict=mingtok_sd(g["mingtok_config"], g["seed"]),
              linear_proj=[(lsd["linear_proj.0.weight"], lsd["linear_proj.0.bias"]), (lsd["linear_proj.2.weight"], lsd["linear_proj.2.bias"])])
T = 10
ams, uns, tus = [], [], []
for i in range(B):
    dec.prefill(dec.embed(torch.randint(0, 400, (T,)).cuda()), seq=i * R, past=0)
    am = torch.ones(1, T + 1, dtype=torch.long); un = am.clone(); un[0, 2:8] = 0; tu = am.clone()
    if R == 3: tu[0, 2:4] = 0
    else: tu = un.clone()
    ams.append(am); uns.append(un); tus.append(tu)
torch.cuda.synchronize(); print("prefill ok", flush=True)
start = dec.embed(torch.tensor([cfg.image_start_token]).cuda())
which = sys.argv[3] if len(sys.argv) > 3 else "all"
if which == "llm":
    from ming_univision_amd.bailing_moe import ImageGenState, build_cfg_rows
    st = ImageGenState(dec, [build_cfg_rows(a, u, t) for a, u, t in zip(ams, uns, tus)], [T] * B)
    h = dec.step(start, st.row_seq, st.row_slot, st.row_pos, st.row_len, st.key_mask, None, rows=B * R)
    torch.cuda.synchronize(); print("llm ok", float(h.abs().sum()), flush=True)
    lat = rf.sample(h, torch.randn(B, 32, device="cuda"), n_images=B)
    torch.cuda.synchronize(); print("rf ok", float(lat.abs().sum()), flush=True)
else:
    out = generate_images(dec, rf, tok, start, [T] * B, ams, uns, tus, torch.randn(B, 5, 32).cuda())
    torch.cuda.synchronize(); print("gen ok", float(out["image"].abs().sum()), flush=True)
