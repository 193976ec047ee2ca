#!/usr/bin/env python3
"""bench.py — visual tokens/s of Ming-UniVision-16B-A3B text->512^2 image generation on MI355X.

Workload (BASELINE.json configs[3], the configuration the metric is quoted on; fits one GPU):
  random-init weights of the exact 16B-A3B architecture (bf16, 16.8 B params + 1.29 B RF head +
  0.70 B MingTok), a 40-token text prompt, forced `<image>`, 2 CFG rows (text->image), 256 visual
  tokens: per token one 28-layer MoE step over the CFG rows, the 16-step rectified-flow SwiGLU
  sampler, one cached semantic-decoder step + linear_proj; then the 24-layer pixel decoder.
  A "step" = one batch of images generated in lock-step (prompt prefill + 257 LLM steps + 256 samplers + pixel
  decode); default 768 images = 1536 CFG rows in one group, the wide route.

One process per GPU (independent prompts per rank = replicas, no data-path collective: the path
is a strictly sequential AR chain per image).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The lock-step groups run on separate HIP streams; the ROCm runtime multiplexes streams onto 4 hardware queues by
# default, which serialises two of four groups (1447 vs 1686 tokens/s).  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak (same guide; the 2:1-sparsity figure is never used)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tokens", type=int, default=256, help="visual tokens per image (256 = 512^2)")
    ap.add_argument("--prompt-len", type=int, default=40)
    ap.add_argument("--images", type=int, default=768,
                    help="images generated in lock-step per GPU (an image batch; 1 = the reference's batch-size-1 call). More than "
                         "32 images per group (64 CFG rows) take the wide route: every Linear a 256x256-tile MFMA GEMM")
    ap.add_argument("--groups", type=int, default=1,
                    help="split the image batch into this many lock-step groups on separate HIP streams (pays below 64 rows per "
                         "group, where launches are HBM/latency-bound; the wide route fills the chip from one group)")
    ap.add_argument("--no-batch1", action="store_true", help="skip the extra batch-size-1 measurement")
    ap.add_argument("--tiny", action="store_true", help="tiny architecture (plumbing check only; INVALID as a result)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / JSON plumbing only (gloo, no GPU, no kernels; `value` is null): the CPU test of --gpus N")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py starts the ranks itself (0 = a free one)")
    ap.add_argument("--layers", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cfg-rows", type=int, default=2, help=argparse.SUPPRESS)       # 3 = editing-style CFG (experiments)
    args = ap.parse_args()
    args.groups = max(1, min(args.groups, args.images))
    return args


def build_models(args, device, seed):
    from ming_univision_amd import configuration as C
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    from ming_univision_amd.mingtok import MingTok
    from ming_univision_amd.rf_head import RectifiedFlowHead
    from ming_univision_amd.synth import synth_tensor
    if args.tiny:
        cfg = C.BailingMoeConfig(vocab_size=512, hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
                                 num_key_value_heads=2, head_dim=128, use_bias=False, rope_theta=600000.0, num_experts=8,
                                 num_shared_experts=2, num_experts_per_tok=3, moe_intermediate_size=64, multi_gate=True,
                                 num_image_tokens_for_gen=args.tokens, image_start_token=500)
        rf_cfg = dict(diffloss_w=64, diffloss_d=2, num_sampling_steps="4", gen_method="flow_matching_swiglu-4")
        tcfg = C.MingTokConfig(
            low_level_encoder=dict(img_size=64, patch_size=32, depth=2, embed_dim=128, ffn_layer="swiglufused", out_dim=32),
            semantic_decoder=dict(in_dim=32, patch_size=32, embed_dim=128, decoder_depth=2, ffn_layer="swiglufused"),
            pixel_decoder=dict(patch_size=16, decoder_depth=2, embed_dim=128))
    else:
        cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
        cfg.num_image_tokens_for_gen = args.tokens
        if args.layers:
            cfg.num_hidden_layers = args.layers
        rf_cfg = dict(C.DEFAULT_VISHEAD_DIFFLOSS)
        tcfg = C.MingTokConfig()
    t_max = args.prompt_len + args.tokens + 8
    dec = BailingMoeDecoder.synthetic(cfg, device, seed=seed, t_max=t_max, n_seq=max(2, args.cfg_rows) * args.images)
    full = C.llm_param_shapes(cfg, rf_cfg, 32)
    rf_sd = {k: synth_tensor(k, s, seed, device, torch.bfloat16) for k, s in full.items()
             if k.startswith("vis_head") or k.startswith("diffloss")}
    rf = RectifiedFlowHead(rf_sd, cfg.hidden_size, rf_cfg)
    D = tcfg.semantic_decoder["embed_dim"]
    lp = {k: synth_tensor(k, s, seed, device, torch.bfloat16)
          for k, s in C.linear_proj_param_shapes(D, cfg.hidden_size, 2).items()}
    tok = MingTok(tcfg, device=device, seed=seed,
                  linear_proj=[(lp["linear_proj.0.weight"], lp["linear_proj.0.bias"]),
                               (lp["linear_proj.2.weight"], lp["linear_proj.2.bias"])])
    return cfg, dec, rf, tok


def one_image(cfg, dec, rf, tok, prompts, noises, groups=1, cfg_rows=2):
    """prefill -> forced <image> -> generate_images (2 CFG rows per image) -> pixel decode.
    prompts [B, T] ids, noises [B, n+1, 32]; B images advance in lock-step (B = 1: the reference's call)."""
    from ming_univision_amd.bailing_moe import generate_images
    B, T = prompts.shape
    if B == 1:
        dec.prefill(dec.embed(prompts[0]), seq=0, past=0)
    else:   # the prompts of the batch prefill in lock-step, many rows per pass through the stack
        dec.prefill_many(dec.embed(prompts).reshape(B, T, -1), [cfg_rows * i for i in range(B)])
    start = dec.embed(torch.tensor([cfg.image_start_token], device=prompts.device))
    am = torch.ones(1, T + 1, dtype=torch.long)
    unc = torch.ones(1, T + 1, dtype=torch.long)
    unc[0, 2:T - 2] = 0                     # uncond row: the user's text span is masked out
    tunc = unc.clone()
    if cfg_rows == 3:
        tunc = am.clone()
        tunc[0, 2:6] = 0                    # text-uncond row of the editing path: a different hole
    return generate_images(dec, rf, tok, start, [T] * B, [am] * B, [unc] * B, [tunc] * B, noises, n_groups=groups)


def dominant_kernel_roofline(rf, rows, iters=48):
    """Average duration of the dominant kernel — the RF head's w12 weight-streaming launch
    (skinny GEMM, N = 2 x hidden, K = w, fused LN-modulate prologue + SwiGLU epilogue) — timed with
    HIP events on the stream it is launched on, cycling through the real per-block weights."""
    from ming_univision_amd import ops
    dev = rf.t["vis_w"].device
    w, hid = rf.w, rf.hidden
    x = torch.randn(rows, w, device=dev)
    sh, sc = torch.randn(rows, w, device=dev) * 0.1, torch.randn(rows, w, device=dev) * 0.1
    out = torch.empty(rows, hid, device=dev)

    def launch(b):
        ops.skinny_gemm(x, rf.lists["w12"][b], rf.lists["b12"][b], prologue="ln_mod", epilogue="swiglu", out=out,
                        ln_g=rf.lists["ln_g"][b], ln_b=rf.lists["ln_b"][b], eps=1e-6, pro_a=sh, pro_b=sc)
    for b in range(rf.depth):
        launch(b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        launch(i % rf.depth)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    nbytes = 2 * hid * w * 2          # algorithmic bytes: the bf16 weight matrix, read once
    traffic = None                    # HBM bytes per launch from the committed PMC pass (same kernel, same shape)
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_skinny_w12_rows2.json")
    if os.path.exists(pmc) and hid == 8192 and w == 3072 and rows == 2:
        traffic = json.load(open(pmc)).get("traffic_bytes_per_launch")
    return dict(traffic=traffic, kernel="skinny_kernel<rows,R,SWIGLU>(RF w12: N=2x%d, K=%d)" % (hid, w), us=us, bytes=nbytes,
                gbs=nbytes / us * 1e-3)


def dominant_kernel_roofline_stream(rf, rows, iters=48):
    """Batched generation (rows >= 5): the dominant kernel is stream_mfma_kernel on the RF head's w12 matrices
    (Ntot = 2 x hidden, K = w).  Timed alone with HIP events on the launch stream, cycling the 12 real matrices."""
    import ctypes as C
    from ming_univision_amd._lib import lib, ptr, current_stream, check
    dev = rf.t["vis_w"].device
    w, hid = rf.w, rf.hidden
    nz = lib().mn_stream_mfma_slices(rows, 2 * hid, w)
    Y = (torch.randn(2 * rows, w, device=dev) * 0.5).to(torch.bfloat16)
    P = torch.empty(nz * rows * 2 * hid, dtype=torch.float32, device=dev)

    def launch(b):
        rc = lib().mn_stream_mfma(ptr(Y), ptr(rf.lists["w12"][b]), ptr(P), rows, 2 * hid, w, current_stream())
        if rc < 0:
            check(rc, "mn_stream_mfma")
    for b in range(rf.depth):
        launch(b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        launch(i % rf.depth)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    nbytes = 2 * hid * w * 2
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_stream_rows%d.json" % rows)   # PMC passes of this kernel at this row count
    if os.path.exists(pmc) and hid == 8192 and w == 3072:
        traffic = json.load(open(pmc)).get("traffic_bytes_per_launch")
    name = "stream_kloop_kernel<4,2,2,64>" if rows > 32 else "stream_mfma_lds_kernel<%d,1,512>" % (2 if rows > 16 else 1)
    return dict(traffic=traffic, kernel="%s (RF w12: Ntot=2x%d, K=%d, rows=%d)" % (name, hid, w, rows), us=us,
                bytes=nbytes, gbs=nbytes / us * 1e-3)


def dominant_kernel_roofline_wide(rf, rows, iters=48):
    """Wide route (> 64 rows in lock-step): the dominant kernel is gemm256_kernel<SWIGLU_SPLIT, hi/lo rows> on the RF head's
    w12 matrices — MFMA-bound.  Timed alone with HIP events on the launch stream, cycling the 12 real matrices.
    Algorithmic flops = 2 * rows * (2 * hidden) * w (the Linear itself); the kernel issues twice that on the matrix cores
    because the fp32 activations enter as bf16 hi + lo halves (numerics policy)."""
    from ming_univision_amd._lib import lib, ptr, current_stream, check
    dev = rf.t["vis_w"].device
    w, hid = rf.w, rf.hidden
    A = (torch.randn(2, rows, w, device=dev) * 0.5).to(torch.bfloat16)
    A[1] *= 2.0 ** -9
    Y = torch.empty(2, rows, hid, dtype=torch.bfloat16, device=dev)

    def launch(b):
        check(lib().mn_gemm256_swiglu_split(ptr(A), w, A.stride(0), ptr(rf.lists["w12"][b]), w, ptr(rf.lists["b12"][b]), ptr(Y),
                                            hid, Y.stride(0), rows, hid, w, current_stream()), "mn_gemm256_swiglu_split")
    for b in range(rf.depth):
        launch(b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        launch(i % rf.depth)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    flops = 2.0 * rows * 2 * hid * w
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r02_pmc_gemm256_w12_rows%d.json" % rows)
    if os.path.exists(pmc) and hid == 8192 and w == 3072:
        traffic = json.load(open(pmc)).get("traffic_bytes_per_launch")
    # the same call site inside a whole bench step, from the committed rocprofv3 kernel trace of `bench.py --steps 1` reduced per
    # (kernel, grid) by tools/site_stats.py (profiles/r03_bench_default_site_stats.csv): what the launch costs IN the step
    in_step = None
    site = os.path.join(ROOT, "profiles", "r03_bench_default_site_stats.csv")
    if os.path.exists(site) and hid == 8192 and w == 3072 and rows == 1536:
        import csv
        for r in csv.DictReader(open(site)):
            if r["site"].startswith("RF w12"):
                in_step = dict(us=float(r["avg_us"]), calls=int(r["calls"]))
    return dict(traffic=traffic, kernel="gemm256_kernel<SWIGLU_SPLIT,2-phase,hi/lo> (RF w12: N=2x%d, K=%d, rows=%d)" % (hid, w, rows),
                us=us, flops=flops, tflops=flops / us * 1e-6, in_step=in_step)


def cpu_baseline(args, rows=2):
    """The oracle (CPU restatement, fp32 PyTorch) on a bounded sample of the same workload:
    one visual token = RF sampler (full size, `rows` CFG rows) + 28 x one full-shape MoE decoder layer
    (decode step against a prompt-length cache) + one semantic-decoder step.  ~10-30 s of CPU work."""
    from ming_univision_amd import configuration as C
    from oracle import bailing_ref, mingtok_ref, rf_ref
    # GEMV-sized fp32 work does not scale past a few tens of threads (256 threads made the RF sampler
    # 100x slower than 8 threads in a first run); use what the reference's CPU path would sensibly get.
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    cores = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)

    def fill(shapes):
        return {k: (torch.rand(s, generator=g) - 0.5) * (2.0 / max(1, s[-1]) ** 0.5) if len(s) > 1
                else torch.ones(s) for k, s in shapes.items()}
    cfgC = C.BailingMoeConfig.ming_univision_16b_a3b()
    rf_shapes = {k: s for k, s in C.llm_param_shapes(cfgC, dict(C.DEFAULT_VISHEAD_DIFFLOSS), 32).items()
                 if k.startswith("vis_head") or k.startswith("diffloss")}
    sd = fill(rf_shapes)
    rf_sd = {k[len("diffloss."):]: v for k, v in sd.items() if k.startswith("diffloss.")}
    hid = torch.randn(rows, cfgC.hidden_size, generator=g)
    noise = torch.randn(1, 32, generator=g)
    with torch.no_grad():
        t0 = time.perf_counter()
        z = rf_ref.vis_head(hid, sd)
        rf_ref.sample(z, noise, rf_sd, steps=16)
        t_rf = time.perf_counter() - t0
    del sd, rf_sd
    ocfg = bailing_ref.LLMConfig(num_hidden_layers=1)
    lsd = fill(C.llm_layer_param_shapes(cfgC, 0))
    T = args.prompt_len + 128
    kv = [dict(k=torch.randn(rows, 4, T, 128, generator=g), v=torch.randn(rows, 4, T, 128, generator=g))]
    x = torch.randn(rows, 1, cfgC.hidden_size, generator=g)
    am = torch.ones(rows, T + 1, dtype=torch.long)
    pos = torch.full((rows, 1), T, dtype=torch.long)
    with torch.no_grad():
        m4 = bailing_ref.build_4d_mask(am, 1, T)
        bailing_ref.decoder_layer(x, lsd, 0, ocfg, m4, pos, dict(kv[0]))      # warm
        t0 = time.perf_counter()
        bailing_ref.decoder_layer(x, lsd, 0, ocfg, m4, pos, dict(kv[0]))
        t_layer = time.perf_counter() - t0
    del lsd
    tcfg = C.MingTokConfig()
    msd = fill({k: s for k, s in C.mingtok_param_shapes(tcfg).items() if k.startswith("semantic_decoder")})
    caches = mingtok_ref.semdec_new_cache(msd)
    with torch.no_grad():
        for _ in range(2):
            t0 = time.perf_counter()
            mingtok_ref.semdec_forward(torch.randn(1, 1, 32, generator=g), msd, kv_caches=caches)
            t_sem = time.perf_counter() - t0
    per_tok = t_rf + cfgC.num_hidden_layers * t_layer + t_sem
    return dict(value=1.0 / per_tok, unit="visual_tokens/s", cores=cores, kind="port",
                sample=("1 visual token at full 16B-A3B shapes, fp32 oracle: RF sampler rows=%d %.2fs + 28 x one MoE "
                        "decoder layer (decode step, %d-token cache) %.3fs + semantic-decoder step %.3fs; pixel decoder "
                        "and prompt prefill not included (favours the CPU)" % (rows, t_rf, T, t_layer, t_sem)))


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) under torch.distributed.run as a CHILD
    process — this process has not touched the GPU (no torch.cuda call yet) and never execs —, relay their output (rank 0
    prints the JSON line) and return the child's exit code."""
    import socket
    import subprocess
    port = args.master_port
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    sys.stderr.write("[bench] starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    return subprocess.run(cmd, env=env).returncode


def dry_run(args):
    """Plumbing check of the multi-rank harness on CPU (gloo): rendezvous, barrier-bracketed timing, MAX over ranks, one JSON
    line on rank 0.  No kernel runs and no throughput is claimed (`value` null)."""
    from ming_univision_amd.dist_util import ReplicaGroup
    grp = ReplicaGroup(backend="gloo")
    dt, _ = grp.timed(lambda: time.sleep(0.01 * (grp.rank + 1)), args.steps)
    ranks = grp.dist.get_world_size() if grp.dist is not None else 1
    if grp.rank == 0:
        print(json.dumps({"metric": "visual tokens/sec (16B-A3B 512^2 gen)", "value": None, "unit": "visual_tokens/s",
                          "n_gpus": grp.world, "rccl_ranks": ranks, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "dry_run": True,
                          "config": {"workload": "dry run: no kernels", "parallelism": "replicas x%d" % grp.world}}), flush=True)
    grp.close()


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        raise SystemExit(launch_ranks(args))        # before anything initialises the GPU in this process
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, env_world))
    if args.dry_run:
        return dry_run(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    from ming_univision_amd.dist_util import ReplicaGroup
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    grp = ReplicaGroup(backend="nccl", device=device)   # "nccl" is RCCL on ROCm
    world, rank = grp.world, grp.rank
    rccl_ranks = grp.dist.get_world_size() if grp.dist is not None else 1

    cfg, dec, rf, tok = build_models(args, device, seed=0)
    g = torch.Generator(device=device).manual_seed(grp.seed(1000))     # independent prompt / noise per replica
    prompt = torch.randint(0, min(cfg.vocab_size, 100000), (args.images, args.prompt_len), generator=g, device=device)
    noises = torch.randn(args.images, args.tokens + 1, 32, generator=g, device=device)

    for _ in range(args.warmup):
        one_image(cfg, dec, rf, tok, prompt, noises, args.groups, args.cfg_rows)
    dt, out = grp.timed(lambda: one_image(cfg, dec, rf, tok, prompt, noises, args.groups, args.cfg_rows), args.steps)
    finite = bool(torch.isfinite(out["image"]).all()) and bool(torch.isfinite(out["latents"]).all())

    batch1 = None
    if args.images > 1 and not args.no_batch1:
        # the reference's own call shape: one image at a time (batch-size 1), same weights, same prompt
        one_image(cfg, dec, rf, tok, prompt[:1], noises[:1])
        dt1, _ = grp.timed(lambda: one_image(cfg, dec, rf, tok, prompt[:1], noises[:1]), 1)
        batch1 = {"images_per_step_per_gpu": 1, "value": args.tokens * world / dt1, "unit": "visual_tokens/s",
                  "ms_per_image": dt1 * 1e3}

    if rank == 0:
        rows = args.cfg_rows
        per_group = (args.images + args.groups - 1) // args.groups
        sys.stderr.write("[bench] %d images x %d tokens x %d steps in %.2f s = %.1f visual tokens/s\n"
                         % (args.images, args.tokens, args.steps, dt, args.tokens * args.steps * world * args.images / dt))
        if rows * per_group > 64:            # wide route: MFMA-bound GEMMs
            dom = dominant_kernel_roofline_wide(rf, rows * per_group)
            roof = {"bound": "mfma", "achieved": dom["tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": dom["tflops"] / MFMA_PEAK_TFLOPS, "traffic": dom["traffic"], "kernel": dom["kernel"],
                    "flops_per_launch": dom["flops"], "us_per_launch": dom["us"],
                    "mfma_issued_tflops": 2 * dom["tflops"],
                    "timing": "HIP events over 48 back-to-back launches of this call site alone, on the launch stream"}
            if dom.get("in_step"):      # the same call site inside a full step (rocprofv3 kernel trace, per-site reduction)
                roof["us_per_launch_in_step_rocprof"] = dom["in_step"]["us"]
                roof["launches_in_step_rocprof"] = dom["in_step"]["calls"]
                roof["frac_in_step_rocprof"] = dom["flops"] / dom["in_step"]["us"] * 1e-6 / MFMA_PEAK_TFLOPS
        else:
            dom = (dominant_kernel_roofline_stream(rf, rows * per_group) if rows * per_group >= 2   # matrix-core route
                   else dominant_kernel_roofline(rf, rows))
            roof = {"bound": "hbm", "achieved": dom["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": dom["gbs"] / HBM_PEAK_GBS, "traffic": dom["traffic"], "kernel": dom["kernel"],
                    "bytes_per_launch": dom["bytes"], "us_per_launch": dom["us"]}
        total_tokens = args.tokens * args.steps * world * args.images
        res = {
            "metric": "visual tokens/sec (16B-A3B 512^2 gen)", "value": total_tokens / dt, "unit": "visual_tokens/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Ming-UniVision-16B-A3B text->image 512^2 (BASELINE configs[3]): %d-token prompt, "
                                   "%d CFG rows, %d visual tokens/image, RF head w=%d d=%d steps=%d, MingTok pixel decode; "
                                   "random-init bf16 weights" % (args.prompt_len, args.cfg_rows, args.tokens, rf.w, rf.depth, rf.steps),
                       "images_per_step_per_gpu": args.images, "stream_groups": args.groups, "parallelism": "replicas x%d" % world, "tiny": bool(args.tiny)},
            "roofline": roof,
            "outputs_finite": finite,
        }
        if batch1 is not None:
            res["batch1"] = batch1
        # whole-token accounting
        ada_bytes = rf.t["ada_w"].numel() * 2          # read once per token (all steps in one GEMM)
        sec_per_token = dt / (args.tokens * args.steps)
        if rows * per_group > 64:
            # wide route (MFMA-bound): algorithmic flops of one image-token = 2 x (parameters each row multiplies), CFG rows
            # included; the kernels issue twice that on the matrix cores (bf16 hi + lo activation halves)
            A = rf.t["ada_w"].shape[0]
            rf_row = 2.0 * (rf.steps * rf.depth * 3 * rf.hidden * rf.w + rf.steps * A * rf.w + rf.w * (cfg.hidden_size + rf.w)
                            + rf.steps * 2 * rf.target * rf.w)
            c = dec.cfg
            llm_row = 2.0 * c.num_hidden_layers * ((c.num_attention_heads + 2 * c.num_key_value_heads) * c.head_dim * c.hidden_size
                                                   + c.num_attention_heads * c.head_dim * c.hidden_size + c.num_experts * c.hidden_size
                                                   + (c.num_experts_per_tok + dec.n_shared) * 3 * c.moe_intermediate_size * c.hidden_size)
            sem_img = 2.0 * 303.0e6
            per_image_token = rows * (rf_row + llm_row) + sem_img
            res["token_level"] = {"algorithmic_GFLOP_per_image_token": per_image_token / 1e9,
                                  "achieved_TFLOPs": per_image_token * args.images / sec_per_token / 1e12,
                                  "mfma_issued_TFLOPs": 2 * per_image_token * args.images / sec_per_token / 1e12,
                                  "ms_per_lockstep_token": sec_per_token * 1e3}
        else:
            tok_bytes = (rf.steps * (rf.weight_bytes_per_step() - ada_bytes) + ada_bytes
                         + dec.weight_bytes_active(min(64, 6 * rows * per_group)) + 0.61e9)
            # every group streams the weights once per lock-step token
            res["token_level"] = {"algorithmic_GB_per_group_token": tok_bytes / 1e9, "groups": args.groups,
                                  "achieved_GBs": args.groups * tok_bytes / sec_per_token / 1e9}
        if not args.no_cpu_baseline and not args.tiny:
            try:
                res["cpu_baseline"] = cpu_baseline(args, rows)
            except Exception as ex:  # the baseline must never kill the bench line
                res["cpu_baseline"] = {"value": None, "unit": "visual_tokens/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (ex,)}
        print(json.dumps(res), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
