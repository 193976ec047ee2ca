import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd import configuration as C
from ming_univision_amd.bailing_moe import BailingMoeDecoder
cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
cfg.num_hidden_layers = 4
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dec = BailingMoeDecoder.synthetic(cfg, torch.device("cuda"), seed=0, with_vocab=False, t_max=1152, n_seq=nb)
e = [torch.randn(1058, 2048, device="cuda") * 0.02 for _ in range(nb)]
for _ in range(3):
    dec.prefill_mfma_many(e, list(range(nb)), past=0)
torch.cuda.synchronize()
