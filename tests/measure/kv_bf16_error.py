"""bf16 KV cache, priced (VERDICT r2 #6 suggestion): the full-width generation case of tests/measure/lo_map.py with every K / V row
rounded to bf16 as it is appended (LLM arena and the semantic decoder's cache; the attention arithmetic stays fp32) — the values a
bf16 cache would hold.  Error of image 0 against the fp32 oracle.  Uses the dev library (mn_kv_round_bf16)."""
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

L = lib()
L.mn_kv_round_bf16.argtypes = [ctypes.c_int]; L.mn_kv_round_bf16.restype = None
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
B, R = 48, 2
dec = BailingMoeDecoder.from_state_dict(cfg, dsd, t_max=32, n_seq=R * B)
nb = torch.randn(B, cfg.num_image_tokens_for_gen + 1, 32, generator=g); nb[0] = noises
emb = dec.embed(ids[0].cuda())


def run(on):
    L.mn_kv_round_bf16(on)
    dec.kv_cache.zero_()
    dec.prefill_many(emb.unsqueeze(0).expand(B, T, emb.shape[1]).contiguous(), [R * i for i in range(B)])
    out = generate_images(dec, rf, tok, dec.embed(torch.tensor([cfg.image_start_token]).cuda()), [T] * B, [am] * B, [un] * B, [un.clone()] * B,
                          nb.cuda(), decode_pixels=False, n_groups=1)
    return (rel_err(out["latents"][0], ref["latents"][:, 0]), rel_err(out["sem"][0], ref["sem"][0]),
            rel_err(out["last_hidden"][:R], ref["last_hidden"][:, 0]))


print("KV cache             latents    sem        hidden   (rel. to the fp32 oracle; bar 1e-3, shipping bar 5e-4)")
print("%-20s %.2e   %.2e   %.2e" % (("fp32 (shipped)",) + run(0)))
print("%-20s %.2e   %.2e   %.2e" % (("rounded to bf16",) + run(1)))
L.mn_kv_round_bf16(0)
