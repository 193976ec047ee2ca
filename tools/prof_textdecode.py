"""16B-A3B text decode, batch 1: 24 greedy steps after a short prompt (for rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd import configuration as C
from ming_univision_amd.bailing_moe import BailingMoeDecoder
cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
dec = BailingMoeDecoder.synthetic(cfg, torch.device("cuda"), seed=0, t_max=256, n_seq=1)
ids = torch.randint(0, 100000, (16,), device="cuda")
h = dec.prefill(dec.embed(ids), seq=0, past=0)[-1:]
seq0 = torch.zeros(1, dtype=torch.int32, device="cuda")
n = 16
for step in range(24):
    tok = torch.argmax(dec.logits(h)[0]).reshape(1)
    slot = torch.tensor([n], dtype=torch.int32, device="cuda")
    h = dec.step(dec.embed(tok), seq0, slot, slot, slot + 1)
    n += 1
torch.cuda.synchronize()
