"""Greedy text decode at one row (16B-A3B shapes, 1058-token cache): the launch sequence of one token, for rocprofv3.
    python tools/exp/prof_text_decode.py [steps] [bf16|fp8|int8|int4]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd import configuration as C
from ming_univision_amd.bailing_moe import BailingMoeDecoder
cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
dec = BailingMoeDecoder.synthetic(cfg, torch.device("cuda"), seed=0, with_vocab=True, t_max=1200, n_seq=1,
                                  weights=sys.argv[2] if len(sys.argv) > 2 else "bf16")
x = torch.randn(1, cfg.hidden_size, device="cuda") * 0.02
seq = torch.zeros(1, dtype=torch.int32, device="cuda")
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    slot = torch.full((1,), 1058 + t, dtype=torch.int32, device="cuda")
    h = dec.step(x, seq, slot, slot, slot + 1)
    lg = dec.logits(h)
torch.cuda.synchronize()
