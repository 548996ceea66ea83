import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops
torch.manual_seed(0)
N, Dh = 16, 16
q = torch.zeros(1, N, Dh, device='cuda'); k = torch.zeros(1, N, Dh, device='cuda')
v = torch.eye(N, device='cuda')[None].contiguous()
mask = torch.zeros(1, N, N, dtype=torch.uint8, device='cuda')
mask[0, :, 12:] = 1          # nobody sees keys 12..15
mask[0, 3, 5] = 1            # query 3 does not see key 5
out, lse = ops.mha_train_fwd(q, k, v, 1, mask, None, 1.0)
torch.set_printoptions(precision=3, linewidth=200)
print(out[0].cpu())
print(lse.cpu())
