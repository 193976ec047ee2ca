"""Shader clock and socket power while the roofline kernel (gemm256 SwiGLU-split on RF w12, 1536 rows) runs back to back, on random and
on zero operands, against an HBM-bound kernel (a device copy): `rocm-smi` sampled from a thread while the launches are queued.
What "power-limited" means for the 0.26 / 0.29 roofline fraction.  `clock_under_load.py loop`: the same sampling over the
lock-step token loop (768 images, 16 tokens)."""
import os, sys, re, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd._lib import lib, ptr, check, current_stream
L = lib()
rows, K, hid = 1536, 3072, 8192
dev = "cuda"
g = torch.Generator().manual_seed(0)
def split(x):
    hi = x.to(torch.bfloat16); lo = (x - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo]).contiguous()
a_rand = split(torch.randn(rows, K, generator=g).to(dev))
w_rand = [(torch.randn(2 * hid, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev) for _ in range(4)]
a_zero, w_zero = torch.zeros_like(a_rand), [torch.zeros_like(w_rand[0])]
y = torch.empty(2, rows, hid, dtype=torch.bfloat16, device=dev)
big = torch.empty(1 << 28, dtype=torch.float32, device=dev); big2 = torch.empty_like(big)
def w12(a2, ws):
    return lambda i: check(L.mn_gemm256_swiglu_split(ptr(a2), K, a2.stride(0), ptr(ws[i % len(ws)]), K, None, ptr(y), hid, y.stride(0), rows, hid, K, current_stream()), "x")
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:
        return None, None
    sclk = re.search(r"sclk clock level.*?\((\d+)Mhz\)", out)
    pw = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([\d.]+)", out)
    return (int(sclk.group(1)) if sclk else None), (float(pw.group(1)) if pw else None)
def measure(name, fn, seconds=6.0, per_launch_flop=None):
    samples, stop = [], False
    def sampler():
        while not stop:
            samples.append(smi()); time.sleep(0.15)
    for i in range(20): fn(i)
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for i in range(200): fn(n + i)
        n += 200
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop = True; th.join()
    s = [x for x in samples[2:] if x[0]]
    clk = sum(x[0] for x in s) / max(1, len(s)); pw = [x[1] for x in s if x[1]]
    extra = f", {per_launch_flop * n / dt / 1e12:.0f} TFLOP/s algorithmic" if per_launch_flop else ""
    print(f"{name}: {dt / n * 1e6:.1f} us per launch{extra}; sclk {clk:.0f} MHz over {len(s)} samples" +
          (f", power {sum(pw) / len(pw):.0f} W" if pw else ""), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "loop":
    # the lock-step token loop itself (768 images, 16 visual tokens): average clock and power over the whole run
    import argparse
    import bench
    B, NT = 768, 16
    a = argparse.Namespace(tiny=False, tokens=NT, layers=None, prompt_len=40, images=B, cfg_rows=2)
    cfg, dec, rf, tok = bench.build_models(a, torch.device("cuda"), 0)
    gg = torch.Generator(device="cuda").manual_seed(0)
    prompt = torch.randint(0, 100000, (B, 40), generator=gg, device="cuda")
    noises = torch.randn(B, NT + 1, 32, generator=gg, device="cuda")
    bench.one_image(cfg, dec, rf, tok, prompt, noises, 1, 2); torch.cuda.synchronize()
    samples, stop = [], False
    def sampler():
        while not stop:
            samples.append(smi()); time.sleep(0.1)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter()
    for _ in range(3): bench.one_image(cfg, dec, rf, tok, prompt, noises, 1, 2)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    stop = True; th.join()
    sm = [x for x in samples[2:] if x[0]]
    pw = [x[1] for x in sm if x[1]]
    print(f"token loop, {B} images x {NT} tokens: {dt:.3f} s = {B * NT / dt:.0f} tokens/s; sclk mean {sum(x[0] for x in sm) / len(sm):.0f} MHz "
          f"(min {min(x[0] for x in sm)}, max {max(x[0] for x in sm)}), power mean {sum(pw) / len(pw):.0f} W (max {max(pw):.0f}) over {len(sm)} samples", flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "energy":
    # energy per launch (mean socket power x time, static power included) of the token loop's main call shapes at 1536 rows
    from ming_univision_amd import ops
    W_, HID = 3072, 8192
    yb = split(torch.randn(rows, HID, generator=g).to(dev))
    w3s = [(torch.randn(W_, HID, generator=g) * HID ** -0.5).to(torch.bfloat16).to(dev) for _ in range(4)]
    P3 = torch.empty(3, rows, W_, dtype=torch.float32, device=dev)
    def w3(i): L.mn_gemm256_splitk(ptr(yb), HID, yb.stride(0), ptr(w3s[i % 4]), HID, None, ptr(P3), rows, W_, HID, 3, current_stream())
    xg = torch.randn(rows, W_, device=dev); gg_ = torch.ones(W_, dtype=torch.bfloat16, device=dev); bb_ = torch.zeros(W_, dtype=torch.bfloat16, device=dev)
    def glue(i): ops.slab_resid_norm(P3, xg, gg_, bb_)
    M_, nq, nkv, hd, t_max = rows, 16, 4, 128, 304
    kvc = torch.randn(M_, 2, nkv, t_max, hd, device=dev); qq = torch.randn(M_, nq * hd, device=dev)
    seqs = torch.arange(M_, dtype=torch.int32, device=dev); lens = torch.full((M_,), 170, dtype=torch.int32, device=dev)
    def attn(i): ops.attn_decode(qq, nq, nkv, hd, kvc, seqs, lens)
    def energy(name, fn, flop=None, seconds=5.0):
        samples, stop = [], False
        def sampler():
            while not stop:
                samples.append(smi()); time.sleep(0.15)
        for i in range(20): fn(i)
        torch.cuda.synchronize()
        th = threading.Thread(target=sampler); th.start()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < seconds:
            for i in range(100): fn(n + i)
            n += 100; torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        stop = True; th.join()
        s_ = [x for x in samples[2:] if x[0] and x[1]]
        pw = sum(x[1] for x in s_) / len(s_); clk = sum(x[0] for x in s_) / len(s_)
        print(f"{name}: {dt / n * 1e6:.1f} us, {pw:.0f} W at {clk:.0f} MHz -> {pw * dt / n * 1e3:.1f} mJ per launch", flush=True)
    energy("RF w12 (SwiGLU-split hi/lo)", w12(a_rand, w_rand))
    energy("RF w3 (split-K 3 slabs)", w3)
    energy("row kernel (3 slabs + residual + LN, bf16 out: mn_slab_resid_norm)", glue)
    energy("GQA decode attention, 170 keys (split + combine)", attn)
    sys.exit(0)
print("idle:", smi(), flush=True)
fl = 2.0 * rows * 2 * hid * K
measure("w12 on random operands, back to back", w12(a_rand, w_rand), per_launch_flop=fl)
measure("w12 on zero operands, back to back", w12(a_zero, w_zero), per_launch_flop=fl)
measure("1 GiB device copy (HBM-bound)", lambda i: big2.copy_(big))
