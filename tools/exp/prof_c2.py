"""C2 (MingTok enc -> dec, 64 x 256^2) a few passes, for rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd import configuration as C
from ming_univision_amd.mingtok import MingTok
tok = MingTok(C.MingTokConfig(), device="cuda", seed=0)
g = torch.Generator().manual_seed(1234)
imgs = (torch.rand(64, 3, 256, 256, generator=g) * 2 - 1).cuda()
for _ in range(3):
    tok.forward_enc_dec(imgs)
torch.cuda.synchronize()
