"""``ViewTransformerLiftSplatShootVoxel`` -- the geometry + pooling half of
P/coocc/image2bev/ViewTransformerLSSVoxel.py / ViewTransformerLSSBEVDepth.py (P1, P2).
DepthNet / lift (the image branch upstream of the path) are out of scope:
``forward`` takes the already-lifted volume.
"""
import torch
from torch import nn

from ._lib import call, host_f32, ptr
from .ops import _pool_workspace
from .registry import NECKS

_F32 = torch.float32


def gen_dx_bx(xbound, ybound, zbound):
    """ViewTransformerLSSBEVDepth.py:21-25."""
    dx = torch.Tensor([row[2] for row in [xbound, ybound, zbound]])
    bx = torch.Tensor([row[0] + row[2] / 2.0 for row in [xbound, ybound, zbound]])
    nx = torch.Tensor([(row[1] - row[0]) / row[2] for row in [xbound, ybound, zbound]])
    return dx, bx, nx


@NECKS.register_module()
class ViewTransformerLiftSplatShootVoxel(nn.Module):
    def __init__(self, loss_depth_weight=1.0, scale=16, point_cloud_range=None, loss_depth_type='bce',
                 grid_config=None, data_config=None, numC_input=512, numC_Trans=64, downsample=16,
                 accelerate=False, use_bev_pool=True, vp_megvii=False, vp_stero=False, cam_channels=27,
                 loss_depth_reg_weight=0.0, use_voxel_net=False, **kwargs):
        super().__init__()
        if grid_config is None:
            grid_config = {'xbound': [-51.2, 51.2, 0.8], 'ybound': [-51.2, 51.2, 0.8],
                           'zbound': [-10.0, 10.0, 20.0], 'dbound': [1.0, 60.0, 1.0]}
        if vp_megvii:
            raise NotImplementedError("vp_megvii pooling is not used by the coocc_nusc configs")
        self.grid_config = grid_config
        dx, bx, nx = gen_dx_bx(grid_config['xbound'], grid_config['ybound'], grid_config['zbound'])
        self.dx = nn.Parameter(dx, requires_grad=False)
        self.bx = nn.Parameter(bx, requires_grad=False)
        self.nx = nn.Parameter(nx, requires_grad=False)
        self.data_config = data_config or {'input_size': (256, 704)}
        self.downsample, self.scale = downsample, scale
        self.numC_input, self.numC_Trans = numC_input, numC_Trans
        self.loss_depth_weight, self.loss_depth_type = loss_depth_weight, loss_depth_type
        self.cam_depth_range = grid_config['dbound']
        self.point_cloud_range = point_cloud_range
        self.frustum = self.create_frustum()
        self.D = self.frustum.shape[0]

    def create_frustum(self):
        """ViewTransformerLSSBEVDepth.py:104-115 (kept as the reference's [D,fH,fW,3] Parameter)."""
        ogfH, ogfW = self.data_config['input_size']
        fH, fW = ogfH // self.downsample, ogfW // self.downsample
        ds = torch.arange(*self.grid_config['dbound'], dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
        D = ds.shape[0]
        xs = torch.linspace(0, ogfW - 1, fW, dtype=torch.float).view(1, 1, fW).expand(D, fH, fW)
        ys = torch.linspace(0, ogfH - 1, fH, dtype=torch.float).view(1, fH, 1).expand(D, fH, fW)
        return nn.Parameter(torch.stack((xs, ys, ds), -1), requires_grad=False)

    def _camera_mats(self, rots, trans, intrins, post_rots, post_trans, bda):
        """Per-camera constants of the geometry chain (33 floats each) + the frustum axes."""
        B, N, _ = trans.shape
        if intrins.shape[3] == 4 or bda.shape[-1] == 4:
            raise NotImplementedError("KITTI 3x4 intrinsics / 4x4 bda are not on the nuScenes path")
        fr = self.frustum.to(trans.device)
        xs, ys, ds = fr[0, 0, :, 0].contiguous(), fr[0, :, 0, 1].contiguous(), fr[:, 0, 0, 2].contiguous()
        mats = torch.cat([torch.inverse(post_rots).reshape(B * N, 9), post_trans.reshape(B * N, 3),
                          rots.matmul(torch.inverse(intrins)).reshape(B * N, 9), trans.reshape(B * N, 3),
                          bda.view(B, 1, 9).expand(B, N, 9).reshape(B * N, 9)], 1).float().contiguous()
        return mats, xs, ys, ds

    def get_geometry(self, rots, trans, intrins, post_rots, post_trans, bda):
        """ViewTransformerLSSBEVDepth.py:117-150 -> [B,N,D,fH,fW,3].  The 3x3 inverses and the
        rots @ inv(intrins) product are host-side torch (6 tiny matrices); the per-point chain
        runs in one HIP kernel."""
        B, N, _ = trans.shape
        mats, xs, ys, ds = self._camera_mats(rots, trans, intrins, post_rots, post_trans, bda)
        D, fH, fW = ds.numel(), ys.numel(), xs.numel()
        geom = torch.empty(B, N, D, fH, fW, 3, device=trans.device, dtype=_F32)
        call("coocc_get_geometry", ptr(mats), ptr(xs), ptr(ys), ptr(ds), B * N, D, fH, fW, ptr(geom))
        return geom

    def voxel_pooling(self, geom_feats, x):
        """ViewTransformerLSSVoxel.py:100-123: geom [B,N,D,H,W,3], x [B,N,D,H,W,C] -> [B,C,X,Y,Z]
        (a channels-last view).  Truncate-then-filter quantisation and the pooling run in the
        library without argsort / boolean-mask compaction."""
        B, N, D, H, W, C = x.shape
        Nprime = B * N * D * H * W
        X, Y, Z = (int(v) for v in self.nx.tolist())
        xf = x.reshape(Nprime, C).float().contiguous()
        g = geom_feats.reshape(Nprime, 3).float().contiguous()
        lo = (self.bx - self.dx / 2.).tolist()
        out = torch.empty(B * X * Y * Z, C, device=x.device, dtype=_F32)
        ws = _pool_workspace(x.device, Nprime, B * X * Y * Z)
        call("coocc_voxel_pool", ptr(xf), ptr(g), Nprime, Nprime // B, C, host_f32(lo + self.dx.tolist()), B, X, Y, Z,
             ptr(out), C, ptr(ws), ws.numel())
        return out.view(B, X, Y, Z, C).permute(0, 4, 1, 2, 3)

    def lift_splat(self, depth_prob, img_feat, geom_feats=None, cams=None):
        """Fused Lift + Splat (ViewTransformerLSSVoxel.py:135-145 without the [B,N,D,H,W,C] volume):
        depth_prob [B*N,D,H,W], img_feat [B*N,C,H,W] -> [B,C,X,Y,Z].  Geometry is either the tensor
        get_geometry returned ([B,N,D,H,W,3]) or, with cams=(rots, trans, intrins, post_rots, post_trans,
        bda), computed inside the key kernel so it never exists in HBM."""
        BN, C, H, W = img_feat.shape
        D = depth_prob.shape[1]
        assert tuple(depth_prob.shape) == (BN, D, H, W) and (geom_feats is None) != (cams is None)
        B = geom_feats.shape[0] if cams is None else cams[1].shape[0]
        X, Y, Z = (int(v) for v in self.nx.tolist())
        dev = img_feat.device
        feat = torch.empty(BN * H * W, C, device=dev, dtype=_F32)
        call("coocc_ncdhw_to_ndhwc", ptr(img_feat.float().contiguous()), ptr(feat), BN, C, H * W, C, 0)
        lo = host_f32((self.bx - self.dx / 2.).tolist() + self.dx.tolist())
        npts = BN * D * H * W
        out = torch.empty(B * X * Y * Z, C, device=dev, dtype=_F32)
        ws = _pool_workspace(dev, npts, B * X * Y * Z)
        dp = depth_prob.float().contiguous()
        if cams is None:
            assert geom_feats.numel() == npts * 3
            g = geom_feats.reshape(-1, 3).float().contiguous()
            call("coocc_lift_splat", ptr(dp), ptr(feat), ptr(g), BN, D, H, W, C, npts // B, lo, B, X, Y, Z, ptr(out), C,
                 ptr(ws), ws.numel())
        else:
            mats, xs, ys, ds = self._camera_mats(*cams)
            assert (ds.numel(), ys.numel(), xs.numel()) == (D, H, W)
            call("coocc_lift_splat_cams", ptr(dp), ptr(feat), ptr(mats), ptr(xs), ptr(ys), ptr(ds), BN, D, H, W, C,
                 npts // B, lo, B, X, Y, Z, ptr(out), C, ptr(ws), ws.numel())
        return out.view(B, X, Y, Z, C).permute(0, 4, 1, 2, 3)

    def forward(self, input):
        """(volume [B,N,D,H,W,C], rots, trans, intrins, post_rots, post_trans, bda, ...) ->
        (bev_feat, geom): the Splat half of ViewTransformerLSSVoxel.py:125-145."""
        volume, rots, trans, intrins, post_rots, post_trans, bda = input[:7]
        geom = self.get_geometry(rots, trans, intrins, post_rots, post_trans, bda)
        return self.voxel_pooling(geom, volume), geom
