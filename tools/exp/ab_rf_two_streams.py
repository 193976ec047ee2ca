"""RF sampler at the bench's operating point: 768 images (1536 rows) in one mn_rf_sample call vs halves (and quarters) of the images
on concurrent HIP streams — does the dispatcher fill one launch's partial rounds (w3: 432 workgroups on 256 CUs) with the other's?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import argparse
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
args = argparse.Namespace(tiny=False, tokens=4, layers=1, prompt_len=40, images=B, cfg_rows=2)
dev = torch.device("cuda")
cfg, dec, rf, tok = bench.build_models(args, dev, seed=0)
g = torch.Generator(device=dev).manual_seed(0)
hid = torch.randn(2 * B, cfg.hidden_size, device=dev, generator=g)
noise = torch.randn(B, rf.target, device=dev, generator=g)
def run(parts):
    n = B // parts
    streams = [torch.cuda.Stream() for _ in range(parts)] if parts > 1 else [torch.cuda.current_stream()]
    outs = [torch.empty(n, rf.target, device=dev) for _ in range(parts)]
    def once():
        ev = torch.cuda.Event(); ev.record()
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                s.wait_event(ev)
                rf.sample(hid[2 * n * i: 2 * n * (i + 1)], noise[n * i: n * (i + 1)], out=outs[i], n_images=n)
        for s in streams: torch.cuda.current_stream().wait_stream(s)
    once(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3, torch.cat(outs)
ref = None
for rnd in range(2):
    for parts in (1, 2, 4):
        ms, out = run(parts)
        if ref is None: ref = out
        print(f"round {rnd}: {parts} stream(s) x {B // parts} images: {ms:.2f} ms per sampler call, identical to one call: {torch.equal(out, ref)}", flush=True)
