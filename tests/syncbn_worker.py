"""Worker of test_gpu_backward.test_syncbn_two_ranks_one_gpu: two processes on ONE GPU (gloo carries the two small
all-reduces), each holding one sample of a batch of two; every rank checks its share against torch's BatchNorm3d in
training mode over the whole batch on the CPU."""
import os
import sys

import torch
import torch.distributed as dist
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from co_occ_amd import autograd as ag
        from util import assert_close
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        g = torch.Generator().manual_seed(21)
        Cin, Cout, X, Y, Z = 8, 12, 6, 5, 4
        x = torch.randn(world, Cin, X, Y, Z, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.1
        res = torch.randn(world, Cout, X, Y, Z, generator=g)
        gout = torch.randn(world, Cout, X, Y, Z, generator=g)
        gam, bet = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
        # reference: plain BatchNorm3d in training mode over the whole batch
        bn_r = torch.nn.BatchNorm3d(Cout, eps=1e-3, momentum=0.1).train()
        bn_r.weight.data.copy_(gam); bn_r.bias.data.copy_(bet)
        xr, wr, rr = x.clone().requires_grad_(), w.clone().requires_grad_(), res.clone().requires_grad_()
        yr = F.relu(bn_r(F.conv3d(xr, wr, padding=1)) + rr)
        yr.backward(gout)
        # ours: this rank's sample
        bn = torch.nn.SyncBatchNorm(Cout, eps=1e-3, momentum=0.1).to(dev).train()
        bn.weight.data.copy_(gam); bn.bias.data.copy_(bet)
        rows = lambda t: t.permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1]).contiguous()
        vol = lambda r: r.view(1, X, Y, Z, -1).permute(0, 4, 1, 2, 3)
        xd = rows(x[rank:rank + 1]).to(dev).requires_grad_()
        wd = w.to(dev).requires_grad_()
        rd = rows(res[rank:rank + 1]).to(dev).requires_grad_()
        out, _ = ag.conv3d_bn_train_rows(xd, wd, (1, X, Y, Z), bn, relu=True, res2d=rd)      # SyncBatchNorm -> synchronised
        assert_close(vol(out.detach().cpu()), yr.detach()[rank:rank + 1], what="syncbn forward")
        out.backward(rows(gout[rank:rank + 1]).to(dev))
        assert_close(vol(xd.grad.cpu()), xr.grad[rank:rank + 1], what="syncbn dx")
        assert_close(vol(rd.grad.cpu()), rr.grad[rank:rank + 1], what="syncbn dres")
        for ours, ref, what in ((wd.grad, wr.grad, "dw"), (bn.weight.grad, bn_r.weight.grad, "dgamma"), (bn.bias.grad, bn_r.bias.grad, "dbeta")):
            tot = ours.detach().clone()
            dist.all_reduce(tot)                      # what DDP does with the per-rank shares
            assert_close(tot.cpu(), ref, what="syncbn " + what)
        assert_close(bn.running_mean.cpu(), bn_r.running_mean, what="running_mean")
        assert_close(bn.running_var.cpu(), bn_r.running_var, what="running_var")
        # a single rank without synchronisation must NOT match (the check above is not vacuous)
        bn2 = torch.nn.BatchNorm3d(Cout, eps=1e-3).to(dev).train()
        bn2.weight.data.copy_(gam); bn2.bias.data.copy_(bet)
        out2, _ = ag.conv3d_bn_train_rows(xd.detach(), wd.detach(), (1, X, Y, Z), bn2, relu=True, res2d=rd.detach())
        assert float((vol(out2.cpu()) - yr.detach()[rank:rank + 1]).abs().max()) > 1e-3
    finally:
        dist.destroy_process_group()
