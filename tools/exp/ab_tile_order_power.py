"""gemm256 hi/lo on RF w12 at 1536 rows: band height of the XCD-aware tile order (mn_gemm256_tune_order group_m) against time, shader clock
and socket power (sysfs) — on a power-limited kernel fabric traffic costs clock.  Interleaved rounds, 3000 launches per arm."""
import sys, os, ctypes, glob, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
from ming_univision_amd._lib import lib, ptr, check, current_stream
L = lib()
L.mn_gemm256_tune_order.argtypes = [ctypes.c_int, ctypes.c_int]; L.mn_gemm256_tune_order.restype = None
rows, K, hid = int(sys.argv[1]) if len(sys.argv) > 1 else 1536, 3072, 8192
dev = "cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(rows, K, generator=g).to(dev)
hi = x.to(torch.bfloat16); a2 = torch.stack([hi, (x - hi.float()).to(torch.bfloat16)]).contiguous()
ws = [(torch.randn(2 * hid, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev) for _ in range(6)]
y = torch.empty(2, rows, hid, dtype=torch.bfloat16, device=dev)
def launch(i):
    check(L.mn_gemm256_swiglu_split(ptr(a2), K, a2.stride(0), ptr(ws[i % 6]), K, None, ptr(y), hid, y.stride(0), rows, hid, K, current_stream()), "w12")
def smi():
    out = []
    for d in glob.glob("/sys/class/drm/card[0-9]*/device"):
        try:
            clk = [float(l.split(":")[1].lower().replace("mhz", "").replace("*", "")) for l in open(d + "/pp_dpm_sclk") if "*" in l][0]
            pw = [float(open(h + "/power1_input").read()) * 1e-6 for h in glob.glob(d + "/hwmon/hwmon*") if os.path.exists(h + "/power1_input")]
            out.append((clk, pw[0] if pw else 0.0))
        except Exception:
            pass
    return max(out, key=lambda t: t[1]) if out else (0.0, 0.0)
for rnd in range(3):
    for gm in (4, 1, 2, 3, 6, 12):
        L.mn_gemm256_tune_order(gm, 1)
        launch(0); torch.cuda.synchronize()
        samples, stop = [], [False]
        def poll():
            while not stop[0]:
                samples.append(smi()); time.sleep(0.05)
        th = threading.Thread(target=poll)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        th.start(); s.record()
        for i in range(3000): launch(i)
        e.record(); torch.cuda.synchronize(); stop[0] = True; th.join()
        us = s.elapsed_time(e) * 1e3 / 3000
        sm = samples[len(samples) // 4:]
        print(f"round {rnd} group_m {gm:2d}: {us:7.2f} us per launch = {2.0 * rows * 2 * hid * K / us * 1e-6:6.0f} TFLOP/s; sclk {sum(c for c, _ in sm) / max(1, len(sm)):6.0f} MHz, "
              f"{sum(p for _, p in sm) / max(1, len(sm)):6.0f} W ({len(sm)} samples)", flush=True)
L.mn_gemm256_tune_order(4, 1)
