"""One-row greedy text decode of the 16B-A3B stack against a 1 058-token cache: ms per token (decoder step + lm_head / arg-max), with
the library given as argv[1] (same-box A/B of two builds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ming_univision_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from ming_univision_amd import configuration as C
from ming_univision_amd.bailing_moe import BailingMoeDecoder
from ming_univision_amd._lib import check, current_stream, lib, ptr
cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
dec = BailingMoeDecoder.synthetic(cfg, torch.device("cuda"), seed=0, with_vocab=True, t_max=1400, n_seq=1)
dec.kv_cache.normal_(0, 0.5)
seq = torch.zeros(1, dtype=torch.int32, device="cuda")


def run(n):
    slot = torch.full((1,), 1058, dtype=torch.int32, device="cuda")
    ln = slot + 1
    tok = torch.tensor([17], device="cuda")
    for _ in range(n):
        h = dec.step(dec.embed(tok), seq, slot, slot, ln)
        tok = dec.greedy(h)
        check(lib().mn_rows_advance(ptr(slot), ptr(ln), None, 1, 1, current_stream()), "adv")
    torch.cuda.synchronize()


run(8)
for _ in range(3):
    t0 = time.perf_counter()
    run(64)
    dt = time.perf_counter() - t0
    print("%s: %.3f ms per token = %.1f tokens/s" % (os.path.basename(_lib.LIB_PATH), dt / 64 * 1e3, 64 / dt), flush=True)
