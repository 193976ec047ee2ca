"""A/B of the 1- / 2-row MoE down projection (moe_down.hip: segments over the waves, one HBM round trip) against round 4's K-segment
skinny launch, in one process on the full 28-layer 16B-A3B stack: one-row text decode (tokens/s) and the 2-row decoder step (ms)."""
import sys, os, argparse, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_moe_tune_down.argtypes = [ctypes.c_int]; L.mn_moe_tune_down.restype = None
dev = torch.device("cuda", 0)
args = argparse.Namespace(tiny=False, tokens=256, layers=None, prompt_len=40, images=1, cfg_rows=2, weights="bf16")
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
g = torch.Generator(device=dev).manual_seed(1)
prompt = torch.randint(0, cfg.vocab_size - 1000, (40,), generator=g, device=dev)
small = dec.view(t_max=200, n_seq=2)
def step2():
    x = torch.randn(2, cfg.hidden_size, device=dev, generator=g)
    seq = torch.arange(2, dtype=torch.int32, device=dev); slot = torch.full((2,), 60, dtype=torch.int32, device=dev)
    def run(): small.step(x, seq, slot, slot, slot + 1, distinct_sequences=True)
    run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 20, small.step(x, seq, slot, slot, slot + 1, distinct_sequences=True).clone()
for rnd in range(2):
    res = {}
    for on in (0, 1):
        L.mn_moe_tune_down(on)
        res[on] = (bench.text_decode_rate(small, prompt), ) + step2()
    d = (res[1][2] - res[0][2]).abs().max().item() / res[0][2].abs().max().item()
    print("round %d: one-row text decode %.1f -> %.1f tokens/s; 2-row 28-layer step %.3f -> %.3f ms; hidden states differ by %.2e" % (
        rnd, res[0][0], res[1][0], res[0][1], res[1][1], d), flush=True)
L.mn_moe_tune_down(1)
