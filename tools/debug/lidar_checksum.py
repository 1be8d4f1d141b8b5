"""Bit-level fingerprint of the sparse LiDAR encoder's output on a fixed synthetic cloud (compare across kernel switches)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from co_occ_amd import lidar as L  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(8)
n = 280000
r = torch.rand(n, generator=g) ** 0.5 * 50
th = torch.rand(n, generator=g) * 6.2832
pts = torch.stack([r * torch.cos(th), r * torch.sin(th), torch.randn(n, generator=g) * 0.8 - 1.5, torch.rand(n, generator=g)], 1).to(dev)
vox = L.Voxelization([0.125] * 3, [-50, -50, -5, 50, 50, 3], 10, (90000, 120000)).eval()
vfe = L.HardSimpleVFE(5)
torch.manual_seed(3)
enc = L.SparseLiDAREnc8x(4, dict(type="BN1d"), 16, 128, [800, 800, 64]).to(dev).eval()
gw = torch.Generator().manual_seed(11)
with torch.no_grad():
    for n_, q in enc.named_parameters():                      # non-trivial BN / GN affine terms, weights with gain ~1
        if q.dim() == 1:
            q.copy_((torch.rand(q.shape, generator=gw) * 0.5 + (0.75 if "weight" in n_ else -0.25)).to(dev))
        else:
            fan = q[0].numel()
            q.copy_((torch.randn(q.shape, generator=gw) * (2.0 / fan) ** 0.5).to(dev))
    v, c, k = vox(pts)
    res = enc(vfe(v, k, c), c, 1)
out, feats = res["x"], res["pts_feats"][0].feats
torch.cuda.synchronize()
for name, t in (("x", out), ("pts_feats[0].feats", feats)):
    b = t.contiguous().cpu().numpy().tobytes()
    print("lidar encoder %s %s sha1 %s  abs-sum %.6f" % (name, tuple(t.shape), hashlib.sha1(b).hexdigest(), float(t.abs().double().sum())))
