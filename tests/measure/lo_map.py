"""Per-site lo-pass map: the full-width generation case of tests/test_gpu_fullwidth.py (16B-A3B widths, 2 LLM layers,
3 visual tokens, 2 CFG rows, 96 rows on the wide route) with ONE Linear site at a time multiplying plain bf16 activations instead of
the hi/lo pair; error of image 0 against the fp32 oracle.  Uses the dev library (mn_lo_drop_mask).  Prints one line per site.
usage: lo_map.py [images (48; the semantic decoder takes the wide route — where the switch acts — from 65 images on)] [site prefix]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
from ming_univision_amd import configuration as C
from ming_univision_amd._lib import lib
from ming_univision_amd.bailing_moe import BailingMoeDecoder, generate_images
from ming_univision_amd.mingtok import MingTok
from ming_univision_amd.rf_head import RectifiedFlowHead
from ming_univision_amd.synth import synth_state_dict
from oracle import bailing_ref, mingtok_ref
from tests.util import llm_sd, rel_err

SITES = ["rf.vis_head", "rf.cond_embed", "rf.adaLN", "rf.w12", "rf.w3", "rf.final", "llm.qkv", "llm.dense", "llm.gate", "llm.experts",
         "sem.qkv", "sem.proj", "sem.w12", "sem.w3", "sem.linear_proj"]
L = lib()
L.mn_lo_drop_mask.argtypes = [ctypes.c_uint]; L.mn_lo_drop_mask.restype = None
seed = 5
d = C.BailingMoeConfig.ming_univision_16b_a3b().to_dict(); d.pop("model_type", None)
d.update(num_hidden_layers=2, vocab_size=1024, num_image_tokens_for_gen=3, image_start_token=1000, pad_token_id=0)
rf_cfg = dict(C.DEFAULT_VISHEAD_DIFFLOSS)
torch.set_num_threads(min(32, torch.get_num_threads()))
sd = llm_sd(d, rf_cfg, seed)
ocfg = bailing_ref.LLMConfig(**{k: v for k, v in d.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
cfg = C.BailingMoeConfig(**d)
dsd = {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}
rf = RectifiedFlowHead(dsd, cfg.hidden_size, rf_cfg)
lsd = synth_state_dict(C.linear_proj_param_shapes(1024, cfg.hidden_size, 2), seed)
dl = {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in lsd.items()}
tok = MingTok(C.MingTokConfig(), device="cuda", seed=seed,
              linear_proj=[(dl["linear_proj.0.weight"], dl["linear_proj.0.bias"]), (dl["linear_proj.2.weight"], dl["linear_proj.2.bias"])])
tsd = {k: v.float().cpu() for k, v in tok.sd.items()}
g = torch.Generator().manual_seed(1)
T = 12
ids = torch.randint(0, 900, (1, T), generator=g)
noises = torch.randn(cfg.num_image_tokens_for_gen + 1, 32, generator=g)
am = torch.ones(1, T + 1, dtype=torch.long)
un = am.clone(); un[0, 2:T - 2] = 0
kvs = bailing_ref.new_kv(ocfg)
bailing_ref.model_forward(sd["model.word_embeddings.weight"][ids], sd, ocfg, torch.ones(1, T, dtype=torch.long), None, kvs)
caches = mingtok_ref.semdec_new_cache(tsd)
ref = bailing_ref.generate_image(sd["model.word_embeddings.weight"][torch.tensor([[cfg.image_start_token]])], kvs, am, un, un.clone(), sd, ocfg,
                                 noises, latent_to_sem=lambda lat: mingtok_ref.mingtok_feature_decoder_step(lat, tsd, caches),
                                 linear_proj=lambda s: bailing_ref.linear_proj(s, lsd), sem_to_pix=lambda s: None, steps=16)
B, R = (int(sys.argv[1]) if len(sys.argv) > 1 else 48), 2
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""
dec = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=32, n_seq=R * B)
nb = torch.randn(B, cfg.num_image_tokens_for_gen + 1, 32, generator=g); nb[0] = noises
emb = dec.embed(ids[0].cuda())


def run(mask):
    L.mn_lo_drop_mask(mask)
    dec.kv_cache.zero_()
    dec.prefill_many(emb.unsqueeze(0).expand(B, T, emb.shape[1]).contiguous(), [R * i for i in range(B)])
    out = generate_images(dec, rf, tok, dec.embed(torch.tensor([cfg.image_start_token]).cuda()), [T] * B, [am] * B, [un] * B, [un.clone()] * B,
                          nb.cuda(), decode_pixels=False, n_groups=1)
    return (rel_err(out["latents"][0], ref["latents"][:, 0]), rel_err(out["sem"][0], ref["sem"][0]),
            rel_err(out["last_hidden"][:R], ref["last_hidden"][:, 0]))


print("site                 latents    sem        hidden   (rel. to the fp32 oracle; bar 1e-3, shipping bar 5e-4)")
print("%-20s %.2e   %.2e   %.2e" % (("none (all hi/lo)",) + run(0)))
for i, name in enumerate(SITES):
    if not name.startswith(ONLY): continue
    print("%-20s %.2e   %.2e   %.2e" % ((name,) + run(1 << i)), flush=True)
print("%-20s %.2e   %.2e   %.2e" % (("ALL sites",) + run((1 << len(SITES)) - 1)))
L.mn_lo_drop_mask(0)
