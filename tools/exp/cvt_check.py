import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
bits = torch.randint(-2**31, 2**31 - 1, (1 << 24,), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
x = bits.view(torch.float32)
x = x[torch.isfinite(x)].contiguous()
a = ops.f32_to_bf16(x.reshape(1, -1)).reshape(-1).view(torch.int16)
b = x.to(torch.bfloat16).view(torch.int16)
d = (a != b)
print("finite values:", x.numel(), "mismatches:", int(d.sum()))
if d.any():
    i = d.nonzero()[:10, 0]
    print(x[i], a[i], b[i])
    den = (x[d].abs() < 1.2e-38)
    print("of which f32-denormal inputs:", int(den.sum()))
