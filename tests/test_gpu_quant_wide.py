"""The weight-only modes above 64 rows (round 5): the wide route expands the byte codes into a bf16 scratch — the RF head's blocks once
per sampler call, a decoder layer's experts once per layer and step (`wide_dequant_rows`, wide_rf.inl / wide_llm.inl) — and runs the
bf16 GEMMs on it, so `dtype="int8"` / `"int4"` (and fp8) generate more than 32 images in lock-step like the bf16 model does.  Full
width (16B-A3B layer shapes, full RF head, full semantic decoder; 2 LLM layers, 3 visual tokens): 65 images x 2 CFG rows = 130 rows
in one group, image 0 against the fp32 oracle fed the model's own de-quantised weights, 1e-3."""
import pytest
import torch

from ming_univision_amd import configuration as C
from tests.util import rel_err
from tests.test_gpu_fp8 import full, _fp8_models           # noqa: F401  (module-scoped fixture + model builder)

pytestmark = pytest.mark.gpu

TOL = 1e-3


@pytest.mark.parametrize("fmt", ["fp8", "int8", "int4"])
def test_quantised_modes_take_the_wide_route_above_64_rows(full, fmt):
    from oracle import bailing_ref, mingtok_ref
    from ming_univision_amd.bailing_moe import generate_images
    d, rf_cfg, sd, ocfg, seed = full
    B, R = 65, 2
    if fmt == "int4":
        from tests.test_gpu_int4 import _int4_models
        cfg, dsd, dec, rf, sdq, lsd, tok = _int4_models(full, B * R)
    else:
        cfg, dsd, dec, rf, sdq, lsd, tok = _fp8_models(full, B * R, fmt)
    assert dec.max_rows() == 2048 and rf.max_rows() == 2048
    tsd = {k: v.float().cpu() for k, v in tok.sd.items()}
    g = torch.Generator().manual_seed(1)
    T = 12
    ids = torch.randint(0, 900, (1, T), generator=g)
    noises = torch.randn(cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    am = torch.ones(1, T + 1, dtype=torch.long)
    un = am.clone(); un[0, 2:T - 2] = 0
    kvs = bailing_ref.new_kv(ocfg)
    bailing_ref.model_forward(sdq["model.word_embeddings.weight"][ids], sdq, ocfg, torch.ones(1, T, dtype=torch.long), None, kvs)
    caches = mingtok_ref.semdec_new_cache(tsd)
    ref = bailing_ref.generate_image(
        sdq["model.word_embeddings.weight"][torch.tensor([[cfg.image_start_token]])], kvs, am, un, un.clone(), sdq, ocfg, noises,
        latent_to_sem=lambda lat: mingtok_ref.mingtok_feature_decoder_step(lat, tsd, caches),
        linear_proj=lambda s: bailing_ref.linear_proj(s, lsd), sem_to_pix=lambda s: None, steps=int(rf_cfg["num_sampling_steps"]))
    assert ref["last_hidden"].shape[0] == R
    start = dec.embed(torch.tensor([cfg.image_start_token]).cuda())
    for i in range(B):
        dec.prefill(dec.embed(ids[0].cuda()), seq=i * R, past=0)
    nb = torch.randn(B, cfg.num_image_tokens_for_gen + 1, 32, generator=g)
    nb[0] = noises
    out = generate_images(dec, rf, tok, start, [T] * B, [am] * B, [un] * B, [un.clone()] * B, nb.cuda(), decode_pixels=False, n_groups=1)
    errs = (rel_err(out["latents"][0], ref["latents"][:, 0]), rel_err(out["last_hidden"][:R], ref["last_hidden"][:, 0]))
    print("%s, %d rows in one group (wide route on de-quantised scratch): image 0 latents %.2e hidden %.2e" % ((fmt, B * R) + errs))
    assert all(torch.isfinite(out["latents"]).all() for _ in (0,)) and max(errs) < TOL, errs
