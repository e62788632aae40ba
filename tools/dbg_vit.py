"""dev: piecewise checks of the ViT projector path (gelu, unpatchify, bilinear at a non-integer ratio, dgrad GEMM)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from joligen_amd import _lib
from joligen_amd._lib import check
from joligen_amd.ops import _dt, _st

D0 = "cuda:0"
L = _lib.lib()
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
g = torch.Generator().manual_seed(5)
for dtype in (torch.float16, torch.bfloat16):
    x = (torch.randn(3, 37, 1536, generator=g) * 2).to(dtype)
    dy = torch.randn(3, 37, 1536, generator=g).to(dtype)
    xr = x.float().requires_grad_(True)
    yr = F.gelu(xr); yr.backward(dy.float())
    xd, dyd = x.to(D0), dy.to(D0)
    y, dx = torch.empty_like(xd), torch.empty_like(xd)
    check(L.jg_gelu_fwd(_dt(xd), xd.data_ptr(), y.data_ptr(), xd.numel(), _st()), "gelu")
    check(L.jg_gelu_bwd(_dt(xd), xd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), xd.numel(), _st()), "gelu_bwd")
    print(dtype, "gelu fwd", rel(y, yr.detach()), "bwd", rel(dx, xr.grad))
    print(" x", x.flatten()[:4].tolist(), "dy", dy.flatten()[:4].tolist(), "dx", dx.flatten()[:4].tolist(), "ref", xr.grad.flatten()[:4].tolist())
    B, Hp, P = 2, 3, 16
    dcol = torch.randn(B * Hp * Hp, 8 * P * P, generator=g).to(dtype)
    dimg = torch.empty(B, Hp * P, Hp * P, 8, device=D0, dtype=dtype)
    dcd = dcol.to(D0)
    check(L.jg_unpatchify(_dt(dcd), dcd.data_ptr(), dimg.data_ptr(), B, Hp, Hp, P, _st()), "unpatchify")
    want = dcol.view(B, Hp, Hp, 8, P, P).flip(4, 5).permute(0, 1, 4, 2, 5, 3).reshape(B, Hp * P, Hp * P, 8)
    print(" unpatchify equal", torch.equal(dimg.cpu(), want))
    from joligen_amd.modules.projected_d import bilinear
    for (Hi, Ho) in ((64, 96), (64, 256), (64, 100)):
        x = torch.randn(2, 8, Hi, Hi, generator=g).to(dtype)
        gy = torch.randn(2, 8, Ho, Ho, generator=g).to(dtype)
        xr = x.float().requires_grad_(True)
        yr = F.interpolate(xr, size=(Ho, Ho), mode="bilinear", align_corners=False)
        yr.backward(gy.float())
        xd = x.permute(0, 2, 3, 1).contiguous().to(D0).requires_grad_(True)
        y = bilinear(xd, Ho, Ho, False)
        y.backward(gy.permute(0, 2, 3, 1).contiguous().to(D0))
        print(" bilinear", Hi, Ho, "fwd", rel(y.permute(0, 3, 1, 2), yr.detach()), "bwd", rel(xd.grad.permute(0, 3, 1, 2), xr.grad))
    # patch-embed conv fwd + my dgrad path vs torch
    import torch.nn as nn
    from joligen_amd.arena import ParamArena
    from joligen_amd.modules.layers import JGConv2d
    from joligen_amd import ops
    from joligen_amd.ops import conv_nt, _gemm_geom
    for S in (96, 256):
        class M(nn.Module):
            def __init__(s):
                super().__init__(); s.proj = JGConv2d(3, 384, 16, padding=0, stride=16)
        torch.manual_seed(1)
        m = M()
        with torch.no_grad():
            m.proj.weight.copy_(m.proj.weight.to(dtype).float())
        wref, bref = m.proj.weight.detach().clone(), m.proj.bias.detach().clone()
        ParamArena(m, D0, dtype, priority=()).refresh()
        x = torch.randn(2, 3, S, S, generator=g).to(dtype)
        xr = x.float().requires_grad_(True)
        yr = F.conv2d(xr, wref, bref, stride=16)
        gy = torch.randn(yr.shape, generator=g).to(dtype)
        yr.backward(gy.float())
        xd = ops.to_nhwc(x.float().to(D0), dtype, 8)
        pm = m.proj.meta
        y = ops.conv2d_forward(xd, pm)
        print(" patch conv", S, "fwd", rel(y.permute(0, 3, 1, 2), yr.detach()))
        gr = S // 16
        N, C = gr * gr, 384
        dpe = gy.permute(0, 2, 3, 1).contiguous().to(D0).view(2 * N, C)
        K = pm.Cin * 256
        dcol = torch.empty((2 * N, K), device=D0, dtype=dtype)
        conv_nt(dpe, pm.w16T.view(K, C), dcol, **_gemm_geom(2 * N, K, C), ldx=C, ldw=C, ldy=K)
        dxi = torch.empty((2, S, S, 8), device=D0, dtype=dtype)
        check(L.jg_unpatchify(_dt(dcol), dcol.data_ptr(), dxi.data_ptr(), 2, gr, gr, 16, _st()), "unpatchify")
        print(" patch conv", S, "dgrad", rel(dxi.permute(0, 3, 1, 2)[:, :3], xr.grad), "pad ch max", float(dxi[..., 3:].abs().max()))
