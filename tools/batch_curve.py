"""Throughput-vs-batch curve of the headline workload: visual tokens/s of BASELINE configs[3] (16B-A3B text -> 512^2,
2 CFG rows per image) over the number of images generated in lock-step, across the three routes — fused chain (<= 4 rows), weight-streaming
kernels (<= 64 rows), wide MFMA route (65+ rows) — in bf16 and in the fp8 / int8 / int4 weight modes (whose wide route expands the codes
into a bf16 scratch: the RF head once per sampler call, a decoder layer's experts once per layer and step).

A point = one whole-image run of `--tokens` visual tokens (default 64 = an 8 x 8 grid: the per-token time is flat over an image's 256 tokens
up to the cache length) after one untimed run, same prompts and noise for every weight mode.  Writes ONE JSON (stdout, or --out) with the curve and a check
that no point falls more than 5 % under the linear interpolation of its neighbours — a cliff at a route switch would show there.

    python tools/batch_curve.py --out profiles/r05_batch_curve.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", default="1,2,4,8,16,32,33,48,64,128,256,512,768,1024")
    ap.add_argument("--tokens", type=int, default=64, help="visual tokens per image: a square grid (64 = 8 x 8 -> a 256^2 image)")
    ap.add_argument("--prompt-len", type=int, default=40)
    ap.add_argument("--modes", default="bf16,fp8,int8,int4")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    images = [int(x) for x in a.images.split(",")]
    dev = torch.device("cuda:0")
    args = argparse.Namespace(tiny=False, tokens=a.tokens, layers=None, prompt_len=a.prompt_len, images=max(images), cfg_rows=2, weights="bf16")
    cfg, dec, rf, tok = bench.build_models(args, dev, seed=0)
    g = torch.Generator(device=dev).manual_seed(1)
    prompts = torch.randint(0, cfg.vocab_size - 1000, (max(images), a.prompt_len), generator=g, device=dev)
    noises = torch.randn(max(images), a.tokens + 1, 32, generator=g, device=dev)
    curves = {}
    for mode in a.modes.split(","):
        if mode == "bf16":
            d, r, lim = dec, rf, max(images)
        else:
            d, r, lim = dec.to_fp8(n_seq=2 * max(images), weights=mode), rf.to_fp8(mode), max(images)   # (above 64 rows: the wide route on a de-quantised scratch)
        pts = []
        for B in images:
            if B > lim:
                continue
            run = lambda: bench.one_image(cfg, d, r, tok, prompts[:B], noises[:B], groups=1)
            run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            rows = 2 * B
            route = "fused chain (<= 4 rows)" if rows <= 4 else ("weight-streaming (<= 64 rows)" if rows <= 64 else "wide MFMA route")
            pts.append(dict(images=B, rows=rows, route=route, tokens_per_s=B * a.tokens / dt, ms_per_token=dt / a.tokens * 1e3))
            sys.stderr.write("%s %4d images: %8.1f tokens/s\n" % (mode, B, pts[-1]["tokens_per_s"]))
        # no point more than 5 % under the interpolation of its neighbours (in images)
        dips = []
        for i in range(1, len(pts) - 1):
            x0, x1, x2 = (pts[j]["images"] for j in (i - 1, i, i + 1))
            y0, y1, y2 = (pts[j]["tokens_per_s"] for j in (i - 1, i, i + 1))
            interp = y0 + (y2 - y0) * (x1 - x0) / (x2 - x0)
            pts[i]["vs_neighbour_interpolation"] = y1 / interp
            if y1 < 0.95 * interp:
                dips.append(x1)
        curves[mode] = dict(points=pts, dips_over_5_percent=dips)
        if mode != "bf16":
            del d, r
            torch.cuda.empty_cache()
    out = dict(workload="Ming-UniVision-16B-A3B text->image 512^2 (BASELINE configs[3]), 2 CFG rows per image, %d-token prompt, %d visual tokens "
                        "per point, one lock-step group" % (a.prompt_len, a.tokens),
               unit="visual_tokens/s", gpu=torch.cuda.get_device_name(0), curves=curves,
               route_thresholds="rows <= 4: fused RF chain; <= 64: weight-streaming kernels (RF head switches to the wide route from 41 rows); "
                                "above: every Linear a 256 x 256-tile MFMA GEMM")
    txt = json.dumps(out, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
