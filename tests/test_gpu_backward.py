"""backward kernels vs torch autograd of the same op in plain fp32 on the CPU
(the floating-point reference the brief allows for float kernels; the reference repo has no backward
kernels of its own -- it relies on torch autograd through cudnn)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import co_occ_amd as pkg
from co_occ_amd import autograd as ag
from util import assert_close

pytestmark = pytest.mark.gpu


def _rows(x):       # [B,C,X,Y,Z] -> [B*X*Y*Z, C]
    return x.permute(0, 2, 3, 4, 1).reshape(-1, x.shape[1]).contiguous()


def _vol(r, B, X, Y, Z):
    return r.view(B, X, Y, Z, -1).permute(0, 4, 1, 2, 3)


CONV_CASES = [
    # Cin, Cout, grid, k, stride, bn, bias, res, relu
    (8, 12, (6, 5, 4), 3, 1, True, False, True, True),
    (16, 8, (7, 6, 5), 3, 2, True, False, False, True),
    (12, 20, (5, 4, 3), 1, 1, False, True, False, False),
    (64, 17, (4, 4, 2), 1, 1, False, True, False, False),      # class logits: Cout % 4 != 0
    (132, 136, (9, 8, 4), 3, 1, True, True, True, True),       # crosses the 128-wide tiles
    (32, 160, (10, 9, 3), 3, 2, False, False, False, True),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "c%d-%d_k%d_s%d" % (c[0], c[1], c[3], c[4]))
def test_conv_backward_vs_torch_autograd(dev, case):
    Cin, Cout, (X, Y, Z), k, stride, use_bn, use_bias, use_res, relu = case
    g = torch.Generator().manual_seed(Cin * 131 + Cout)
    B = 2
    x = torch.randn(B, Cin, X, Y, Z, generator=g)
    w = torch.randn(Cout, Cin, k, k, k, generator=g) / (Cin * k ** 3) ** 0.5
    b = torch.randn(Cout, generator=g) if use_bias else None
    bn = torch.nn.BatchNorm3d(Cout).eval()
    bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.1)
    bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
    bn.weight.data.copy_(torch.rand(Cout, generator=g) + 0.5)
    bn.bias.data.copy_(torch.randn(Cout, generator=g) * 0.1)
    pad = k // 2
    Xo, Yo, Zo = ((n + 2 * pad - k) // stride + 1 for n in (X, Y, Z))
    res = torch.randn(B, Cout, Xo, Yo, Zo, generator=g) if use_res else None
    gout = torch.randn(B, Cout, Xo, Yo, Zo, generator=g)

    # fp32 reference on the CPU
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if use_bias else None
    rr = res.clone().requires_grad_() if use_res else None
    y = F.conv3d(xr, wr, br, stride=stride, padding=pad)
    if use_bn:
        y = bn(y)
    if use_res:
        y = y + rr
    if relu:
        y = F.relu(y)
    y.backward(gout)

    xd = _rows(x).to(dev).requires_grad_()
    wd = w.to(dev).requires_grad_()
    bd = b.to(dev).requires_grad_() if use_bias else None
    rd = _rows(res).to(dev).requires_grad_() if use_res else None
    out, geom = ag.conv3d_rows(xd, wd, (B, X, Y, Z), bias=bd, bn=bn if use_bn else None, stride=stride, relu=relu, res2d=rd)
    assert geom == (B, Xo, Yo, Zo)
    assert_close(_vol(out.detach().cpu(), B, Xo, Yo, Zo), y.detach(), what="forward")
    out.backward(_rows(gout).to(dev))
    assert_close(_vol(xd.grad.cpu(), B, X, Y, Z), xr.grad, what="dgrad")
    assert_close(wd.grad.cpu(), wr.grad, what="wgrad")
    if use_bias:
        assert_close(bd.grad.cpu(), br.grad, what="dbias")
    if use_res:
        assert_close(_vol(rd.grad.cpu(), B, Xo, Yo, Zo), rr.grad, what="dres")


def test_linear_backward_and_chain(dev):
    """Two stacked layers (Linear+ReLU -> Linear): gradients flow through both Functions."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(300, 128, generator=g)
    w1, b1 = torch.randn(64, 128, generator=g) / 11, torch.randn(64, generator=g)
    w2, b2 = torch.randn(4, 64, generator=g) / 8, torch.randn(4, generator=g)
    ref = [t.clone().requires_grad_() for t in (x, w1, b1, w2, b2)]
    yr = F.linear(F.relu(F.linear(ref[0], ref[1], ref[2])), ref[3], ref[4])
    yr.square().sum().backward()
    got = [t.to(dev).requires_grad_() for t in (x, w1, b1, w2, b2)]
    y = ag.linear_rows(ag.linear_rows(got[0], got[1], got[2], relu=True), got[3], got[4])
    assert_close(y.detach().cpu(), yr.detach(), what="forward")
    y.square().sum().backward()
    for a, b, name in zip(got, ref, ("dx", "dw1", "db1", "dw2", "db2")):
        assert_close(a.grad.cpu(), b.grad, what=name)
