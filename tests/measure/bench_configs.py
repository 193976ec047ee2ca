"""Secondary configurations of BASELINE.json (configs[0..2]) measured on the HIP path:
  C1  MingTok enc->dec, 1 x 256^2, full-size synthetic weights: PSNR vs the fp32 CPU oracle + latency
  C2  MingTok enc->dec, batch 64 x 256^2: images/s and achieved TFLOP/s (213 GFLOP per image)
  C3  16B-A3B image->text understanding: 1024^2 image -> 1024 image tokens, ~1060-token prefill + 64 greedy decode steps
Prints one JSON object per config."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd import configuration as C
from ming_univision_amd.mingtok import MingTok


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def c1_c2(args):
    tok = MingTok(C.MingTokConfig(), device="cuda", seed=0)
    g = torch.Generator().manual_seed(1234)
    img = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    rec = tok.forward_enc_dec(img.cuda())
    out = {"config": "C1 MingTok enc->dec 1x256^2", "ms": timeit(lambda: tok.forward_enc_dec(img.cuda()), 5) * 1e3}
    if not args.no_oracle:
        from oracle import mingtok_ref
        sd = {k: v.float().cpu() for k, v in tok.sd.items()}
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        t0 = time.perf_counter()
        with torch.no_grad():
            ref = mingtok_ref.mingtok_forward_enc_dec(img, sd)
        out["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
        mse = float(((rec.cpu().double() - ref.double()) ** 2).mean())
        out["psnr_vs_fp32_oracle_db"] = 10 * torch.log10(torch.tensor(4.0 / mse)).item()
        mse_in = lambda x: float(((x.double().cpu() - img.double()) ** 2).mean())
        out["psnr_vs_input_db"] = {"hip": 10 * torch.log10(torch.tensor(4.0 / mse_in(rec))).item(),
                                   "oracle": 10 * torch.log10(torch.tensor(4.0 / mse_in(ref))).item()}
    out["ms_fp32_class"] = timeit(lambda: tok.forward_enc_dec(img.cuda(), precision="fp32"), 3) * 1e3
    print(json.dumps(out), flush=True)
    B = 64
    imgs = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).cuda()
    dt = timeit(lambda: tok.forward_enc_dec(imgs), 3)
    print(json.dumps({"config": "C2 MingTok enc->dec 64x256^2", "ms_per_batch": dt * 1e3, "images_per_s": B / dt,
                      "achieved_TFLOPs": 213e9 * B / dt / 1e12, "mfma_bf16_peak_TFLOPs": 2500.0,
                      "frac_of_mfma_peak": 213e9 * B / dt / 2.5e15}), flush=True)
    dt32 = timeit(lambda: tok.forward_enc_dec(imgs, precision="fp32"), 1)
    print(json.dumps({"config": "C2 fp32-class regime (hi/lo GEMMs, fp32 attention)", "ms_per_batch": dt32 * 1e3, "images_per_s": B / dt32}), flush=True)
    dt = timeit(lambda: tok.forward(imgs), 3)
    dtp = timeit(lambda: tok.forward_pixel_decoder(tok.forward(imgs)["x_norm_patchtokens"]), 1)
    print(json.dumps({"config": "C2 split", "enc+sem_ms": dt * 1e3, "enc+sem+pix_ms": dtp * 1e3}), flush=True)


def c3(args):
    from ming_univision_amd.modeling import MingUniVisionForConditionalGeneration
    cfg = C.MingUniVisionConfig.ming_univision_16b_a3b()
    model = MingUniVisionForConditionalGeneration(cfg, device="cuda", seed=0, t_max=1400)
    ids = torch.randint(0, 100000, (1, 12 + 1026 + 20))
    ids[0, 12] = cfg.llm_config.image_start_token
    ids[0, 13:13 + 1024] = cfg.llm_config.image_patch_token
    ids[0, 13 + 1024] = 126348
    px = torch.rand(1, 3, 1024, 1024) * 2 - 1
    for regime in ("fp32", "bf16"):
        model.understanding_precision = regime
        model.reset_inner_state()
        model.extract_image_feature(px.cuda())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.extract_image_feature(px.cuda())
        torch.cuda.synchronize(); t_img = time.perf_counter() - t0
        model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=px, max_new_tokens=3)   # warm-up
        torch.cuda.synchronize(); model.reset_inner_state()
        t0 = time.perf_counter()
        seq = model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=px, max_new_tokens=1)
        torch.cuda.synchronize(); t_prefill = time.perf_counter() - t0
        model.reset_inner_state()
        t0 = time.perf_counter()
        seq = model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=px, max_new_tokens=65)
        torch.cuda.synchronize(); t_all = time.perf_counter() - t0
        print(json.dumps({"config": "C3 16B-A3B image(1024^2)->text, %s regime" % ("fp32-class (default)" if regime == "fp32" else "bf16 (autocast-like)"),
                          "prompt_tokens": ids.shape[1], "mingtok_1024_plus_linear_proj_ms": t_img * 1e3, "prefill_incl_vision_s": t_prefill,
                          "decode_tokens_per_s": 64 / max(1e-9, t_all - t_prefill), "new_tokens": int(seq.shape[1] - ids.shape[1])}), flush=True)
    model.understanding_precision = "fp32"
    # B conversations in lock-step (generate_text_batch): same prompt shape per conversation, different token ids
    for B in args.c3_batches:
        del model
        torch.cuda.empty_cache()
        model = MingUniVisionForConditionalGeneration(cfg, device="cuda", seed=0, t_max=1152)
        reqs = []
        for b in range(B):
            r = ids.clone()
            r[0, :12] = torch.randint(0, 100000, (12,))
            reqs.append(dict(input_ids=r, pixel_values=px))
        model.generate_text_batch(reqs[:2], max_new_tokens=2)
        tm = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model.generate_text_batch(reqs, max_new_tokens=65, sync_every=1 << 30, timings=tm)
        torch.cuda.synchronize(); t_all = time.perf_counter() - t0
        print(json.dumps({"config": f"C3 x{B} conversations in lock-step", "prefill_incl_vision_s_per_conversation": tm["prefill_s"] / B,
                          "decode_tokens_per_s": tm["steps"] * B / tm["decode_s"], "ms_per_decode_step": tm["decode_s"] / tm["steps"] * 1e3,
                          "end_to_end_tokens_per_s": 65 * B / t_all, "new_tokens": len(out[0])}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--c3", action="store_true")
    ap.add_argument("--c3-batches", type=int, nargs="*", default=[16, 64, 256])
    a = ap.parse_args()
    c1_c2(a)
    if a.c3:
        c3(a)
