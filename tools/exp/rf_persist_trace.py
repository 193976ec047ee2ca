"""Timeline of the persistent per-Euler-step launch (stream_kc.hip): workgroup 0's clock stamps per phase, averaged over the blocks of
the LAST Euler step of one sampler call.  Columns: us from the phase's start to: operand image in LDS, weight stream done, epilogue
stored, arrival flag raised, next phase's first chunks parked / requested, barrier passed (= the next phase's start).
    python tools/exp/rf_persist_trace.py [bf16|fp8|int8|int4] [rows]"""
import sys, os, argparse, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_rf_kc_trace.argtypes = [ctypes.c_void_p]; L.mn_rf_kc_trace.restype = None
L.mn_rf_tune_fuse.argtypes = [ctypes.c_int]; L.mn_rf_tune_fuse.restype = None
L.mn_rf_tune_fuse(3 | 32)                            # one launch per Euler step (the whole-sampler launch carries no stamps)
L.mn_rf_kc_persist_all.argtypes = [ctypes.c_int]; L.mn_rf_kc_persist_all.restype = None
L.mn_rf_kc_persist_all(1)                          # (A/B every format; the product enables bf16 and e4m3)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
args = argparse.Namespace(tiny=False, tokens=256, layers=2, prompt_len=40, images=1, cfg_rows=2, weights="bf16")
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
del dec, tok
weights = sys.argv[1] if len(sys.argv) > 1 else "bf16"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if weights != "bf16":
    rf = rf.to_fp8(weights)
hid = torch.randn(rows, cfg.hidden_size, device=dev, generator=g)
noise = torch.randn(1, 32, device=dev, generator=g)
lat = torch.empty(1, 32, device=dev)
for _ in range(3):
    rf.sample(hid, noise, n_images=1, out=lat)
torch.cuda.synchronize()
nph = 2 * rf.depth
G = 256
buf = torch.zeros(G * nph * 8, dtype=torch.int64, device=dev)
L.mn_rf_kc_trace(buf.data_ptr())
rf.sample(hid, noise, n_images=1, out=lat)
torch.cuda.synchronize()
L.mn_rf_kc_trace(None)
T = buf.cpu().reshape(G, nph, 8).double() / 100.0         # us
names = ["operand image", "stream done", "epilogue", "arrived", "prefetched", "barrier passed"]
for ph, label, n_wg in ((0, "w12'", 256), (1, "w3'", 192)):
    t = T[0]
    sel = t[ph::2]
    sel = sel[1:-1] if len(sel) > 2 else sel               # (the first / last phases of a launch have no barrier on one side)
    d = sel[:, 1:7] - sel[:, 0:1]
    print(f"{weights} rows {rows} {label} workgroup 0: " + ", ".join(f"{n} {d[:, i].mean():.2f}" for i, n in enumerate(names)) + f"   (us from phase start; {len(sel)} phases)")
    # every workgroup: its times relative to the EARLIEST phase start of the phase
    ph_ids = list(range(ph, nph, 2))[1:-1]
    for k, n in ((0, "phase start"), (1, "operand image"), (2, "stream done"), (4, "arrived"), (5, "prefetched"), (6, "barrier passed")):
        x = torch.stack([T[:n_wg, q, k] - T[:n_wg, q, 0].min() for q in ph_ids])       # [phases, wg]
        print(f"    all {n_wg} workgroups, {n:15s}: min {x.min(1).values.mean():6.2f}  median {x.median(1).values.mean():6.2f}  90% {x.quantile(0.9, 1).mean():6.2f}  max {x.max(1).values.mean():6.2f}")
    # who is last?  by XCD (wg % 8)
    q = ph_ids[len(ph_ids) // 2]
    arr = T[:n_wg, q, 4] - T[:n_wg, q, 0].min()
    print("    arrival by XCD (wg % 8), one phase: " + " ".join(f"{arr[x::8].mean():.1f}" for x in range(8)) + "   latest workgroups: " + " ".join(str(int(i)) for i in arr.argsort(descending=True)[:8]))
whole = (T[0, -1, 3] - T[0, 0, 0]).item()
print(f"launch body {whole:.1f} us for {rf.depth} blocks = {whole / rf.depth:.2f} us per block")
