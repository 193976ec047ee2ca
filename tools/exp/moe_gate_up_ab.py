"""A/B of the one-launch router + expert gate/up of 1-row decode steps (moe_gate_up.hip: every workgroup routes its row itself) against the
router launch + pair launch, in one process on the full 28-layer 16B-A3B stack: one-row text decode (tokens/s), ms per 1-row step, and the
difference of the hidden states of one step (same input)."""
import sys, os, argparse, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_moe_tune_gate_up.argtypes = [ctypes.c_int, ctypes.c_int]; L.mn_moe_tune_gate_up.restype = None
L.mn_moe_tune_down.argtypes = [ctypes.c_int]; L.mn_moe_tune_down.restype = None
dev = torch.device("cuda", 0)
args = argparse.Namespace(tiny=False, tokens=256, layers=None, prompt_len=40, images=1, cfg_rows=2, weights="bf16")
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
weights = sys.argv[1] if len(sys.argv) > 1 else "bf16"
if weights != "bf16":
    dec = dec.to_fp8(n_seq=2, weights=weights)
g = torch.Generator(device=dev).manual_seed(1)
prompt = torch.randint(0, cfg.vocab_size - 1000, (40,), generator=g, device=dev)
small = dec.view(t_max=200, n_seq=2)
x = torch.randn(1, cfg.hidden_size, device=dev, generator=g)
seq = torch.zeros(1, dtype=torch.int32, device=dev); slot = torch.full((1,), 60, dtype=torch.int32, device=dev)
def step1():
    def run(): return small.step(x, seq, slot, slot, slot + 1, distinct_sequences=True)
    run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): out = run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 20, out.clone()
ARMS = (("router launch + pair launch + K-segment down launch", 0, 0), ("+ moe_down.hip", 0, 1), ("+ one-launch router + gate/up", 1, 1))
for rnd in range(3):
    res = []
    for name, gu, dn in ARMS:
        L.mn_moe_tune_gate_up(gu, 1)
        L.mn_moe_tune_down(dn)
        res.append((bench.text_decode_rate(small, prompt), ) + step1())
    d = (res[2][2] - res[0][2]).abs().max().item() / res[0][2].abs().max().item()
    print("%s round %d: " % (weights, rnd) + "; ".join("%s: %.1f tokens/s, %.3f ms per 1-row step" % (ARMS[i][0], res[i][0], res[i][1]) for i in range(3)) +
          "; hidden states of the first and last form differ by %.2e" % d, flush=True)
L.mn_moe_tune_gate_up(1, 1)
L.mn_moe_tune_down(1)
