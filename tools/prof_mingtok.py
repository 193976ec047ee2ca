import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ming_univision_amd import configuration as C
from ming_univision_amd.mingtok import MingTok
tok = MingTok(C.MingTokConfig(), device="cuda", seed=0)
imgs = (torch.rand(64, 3, 256, 256) * 2 - 1).cuda()
for _ in range(2):
    tok.forward_enc_dec(imgs)
torch.cuda.synchronize()
