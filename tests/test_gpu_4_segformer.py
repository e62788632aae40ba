"""GPU parity tests of the SegFormer attention generator: every new kernel against an fp32 torch reference of the
same op, then the whole generator (forward in train mode with the recorded DropPath / Dropout2d draws, feature taps, backward, eval
mode, BatchNorm running statistics) against fixtures from the unmodified reference (oracle/make_golden_segformer.py)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

import jg_oracle as O

pytestmark = pytest.mark.gpu
D0 = "cuda:0"
DTYPES = [torch.float16, torch.bfloat16]
TOL = {torch.float16: 3e-3, torch.bfloat16: 2e-2}


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def param(t):
    p = torch.nn.Parameter(t.to(D0).float().contiguous())
    p.grad = torch.zeros_like(p)
    return p


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("R,C", [(4096, 32), (1000, 64), (77, 160), (512, 256), (3, 8)])
def test_layernorm(R, C, dtype):
    from joligen_amd import ops_segformer as S
    x, gy = rnd((2, R, C), dtype, 1), rnd((2, R, C), dtype, 2)
    w, b = 1 + 0.2 * rnd((C,), torch.float32, 3), 0.1 * rnd((C,), torch.float32, 4)
    xr, wr, br = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), wr, br, 1e-6)
    yr.backward(gy.float())
    xd, wd, bd = x.to(D0).requires_grad_(True), param(w), param(b)
    y = S.layer_norm(xd, wd, bd, 1e-6)
    y.backward(gy.to(D0))
    torch.cuda.synchronize()
    assert relerr(y, yr) < TOL[dtype] and relerr(xd.grad, xr.grad) < TOL[dtype], (relerr(y, yr), relerr(xd.grad, xr.grad))
    assert relerr(wd.grad, wr.grad) < 1e-3 and relerr(bd.grad, br.grad) < 1e-3, (relerr(wd.grad, wr.grad), relerr(bd.grad, br.grad))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("R,C", [(4096, 32), (77, 160), (3, 8)])
def test_layernorm_with_identity_gradient(R, C, dtype):
    """round 5: `layer_norm_id` = (x, LayerNorm(x)) of a pre-norm block x + f(norm(x)); the identity output's gradient is added inside the
    LayerNorm-backward launch (jg_layernorm_bwd_add) -- against fp32 torch autograd on the same little block, and with either output unused."""
    from joligen_amd import ops_segformer as S
    x, gy = rnd((2, R, C), dtype, 1), rnd((2, R, C), dtype, 2)
    w, b = 1 + 0.2 * rnd((C,), torch.float32, 3), 0.1 * rnd((C,), torch.float32, 4)
    xr, wr, br = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = 0.5 * xr + 2.0 * F.layer_norm(xr, (C,), wr, br, 1e-6)
    yr.backward(gy.float())
    xd, wd, bd = x.to(D0).requires_grad_(True), param(w), param(b)
    xi, h = S.layer_norm_id(xd, wd, bd, 1e-6)
    y = 0.5 * xi + 2.0 * h
    y.backward(gy.to(D0))
    torch.cuda.synchronize()
    assert relerr(y, yr) < TOL[dtype] and relerr(xd.grad, xr.grad) < TOL[dtype], (relerr(y, yr), relerr(xd.grad, xr.grad))
    assert relerr(wd.grad, wr.grad) < 1e-3 and relerr(bd.grad, br.grad) < 1e-3
    # only the normalised branch used / only the identity used
    x2 = x.to(D0).requires_grad_(True)
    S.layer_norm_id(x2, param(w), param(b), 1e-6)[1].backward(gy.to(D0))
    x3 = x.to(D0).requires_grad_(True)
    S.layer_norm(x3, param(w), param(b), 1e-6).backward(gy.to(D0))
    assert torch.equal(x2.grad, x3.grad)
    x4 = x.to(D0).requires_grad_(True)
    S.layer_norm_id(x4, param(w), param(b), 1e-6)[0].backward(gy.to(D0))
    assert torch.equal(x4.grad, gy.to(D0))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,C,scaled", [(4, 1024, 32, True), (3, 77, 160, True), (2, 64, 256, False), (5, 3, 8, True)])
def test_add_layer_norm_id(B, N, C, scaled, dtype):
    """round 6 (`jg_layernorm_fwd_add`): y = identity + branch * scale[image] and LayerNorm(y) in one pass, both outputs and every gradient
    (identity, branch, gamma, beta) against torch autograd of the same expression on the 16-bit-rounded sum."""
    from joligen_amd import ops_segformer as S
    ident, br = rnd((B, N, C), dtype, 90), rnd((B, N, C), dtype, 91)
    sc = (torch.rand(B, generator=torch.Generator().manual_seed(3)) > 0.3).float() / 0.7 if scaled else None
    g_, b_ = rnd((C,), torch.float32, 92) * 0.3 + 1.0, rnd((C,), torch.float32, 93) * 0.1
    gy, gh = rnd((B, N, C), dtype, 94), rnd((B, N, C), dtype, 95)
    ir, brr = ident.float().requires_grad_(True), br.float().requires_grad_(True)
    gr, btr = g_.clone().requires_grad_(True), b_.clone().requires_grad_(True)
    yr = ir + brr * (sc.view(B, 1, 1) if scaled else 1.0)
    hr = F.layer_norm(yr, (C,), gr, btr, 1e-6)
    (yr * gy.float()).sum().backward(retain_graph=True)
    (hr * gh.float()).sum().backward()
    idv, bd = ident.to(D0).requires_grad_(True), br.to(D0).requires_grad_(True)
    gp, bp = param(g_), param(b_)
    y, h = S.add_layer_norm_id(idv, bd, None if sc is None else sc.to(D0), gp, bp, 1e-6)
    torch.autograd.backward([y, h], [gy.to(D0), gh.to(D0)])
    torch.cuda.synchronize()
    assert relerr(y, yr) < TOL[dtype] and relerr(h, hr) < TOL[dtype]
    assert relerr(idv.grad, ir.grad) < TOL[dtype] and relerr(bd.grad, brr.grad) < TOL[dtype]
    assert relerr(gp.grad, gr.grad) < 2 * TOL[dtype] and relerr(bp.grad, btr.grad) < 2 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("two_phase", [True, False])
@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 128), (1, 9, 7, 256), (2, 4, 4, 1024), (1, 32, 32, 64), (8, 64, 64, 128), (4, 16, 16, 640)])
def test_dwconv3x3_gelu(B, H, W, C, dtype, two_phase, monkeypatch):
    """two_phase: the weight / bias partials of the blocks through a workspace and a summing launch (the default), or atomics from
    every block; (8, 64, 64, 128) runs 1024 first-phase blocks = 16 row groups of the summing grid."""
    from joligen_amd import ops_segformer as S
    monkeypatch.setattr(S, "DW_TWO_PHASE", two_phase)
    x, gy = rnd((B, C, H, W), dtype, 5), rnd((B, C, H, W), dtype, 6)
    w, b = rnd((C, 1, 3, 3), torch.float32, 7, 0.4), rnd((C,), torch.float32, 8, 0.1)
    xr, wr, br = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.gelu(F.conv2d(xr, wr, br, padding=1, groups=C))
    yr.backward(gy.float())
    xd = x.permute(0, 2, 3, 1).contiguous().to(D0).requires_grad_(True)
    wd, bd = param(w), param(b)            # [C,1,3,3] contiguous == the arena's [C][3][3][1] layout
    y = S.dwconv3x3(xd, wd, bd, gelu=True)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(D0))
    torch.cuda.synchronize()
    nchw = lambda t: t.permute(0, 3, 1, 2)
    assert relerr(nchw(y), yr) < TOL[dtype] and relerr(nchw(xd.grad), xr.grad) < 2 * TOL[dtype], (relerr(nchw(y), yr), relerr(nchw(xd.grad), xr.grad))
    assert relerr(wd.grad, wr.grad) < 2 * TOL[dtype] and relerr(bd.grad, br.grad) < 2 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 128), (1, 9, 7, 256), (2, 4, 5, 64), (4, 64, 64, 256)])
def test_dwconv3x3_reflect_padding_in_one_launch(B, H, W, C, dtype, monkeypatch):
    """Round 6: nn.Conv2d(C, C, 3, padding=1, padding_mode='reflect', groups=C) -- the depth-wise convolution of SeparableConv2d in the mobile
    ResNet blocks (mobile_modules.py:4-40) -- with the mirror inside the kernels (jg_dwconv3x3_fwd_pad / _bwd_ws_pad, pad_mode 1): forward,
    input gradient (the border-adjacent rows / columns collect their mirror images' terms; the corners four windows), weight and bias gradient
    against fp32 autograd of F.pad(mode='reflect') + F.conv2d; the border ring of the input gradient separately; and against the
    reflect-pad -> zero-padded kernel -> crop composition of rounds 3-5 (JG_DW_REFLECT=0)."""
    from joligen_amd import ops_segformer as S
    x, gy = rnd((B, C, H, W), dtype, 15), rnd((B, C, H, W), dtype, 16)
    w, b = rnd((C, 1, 3, 3), torch.float32, 17, 0.4), rnd((C,), torch.float32, 18, 0.1)
    xr, wr, br = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), wr, br, groups=C)
    yr.backward(gy.float())
    nchw = lambda t: t.permute(0, 3, 1, 2)
    got = {}
    for one in (True, False):
        monkeypatch.setattr(S, "DW_REFLECT", one)
        xd = x.permute(0, 2, 3, 1).contiguous().to(D0).requires_grad_(True)
        wd, bd = param(w), param(b)
        y = S.dwconv3x3(xd, wd, bd, gelu=False, reflect=True)
        y.backward(gy.permute(0, 2, 3, 1).contiguous().to(D0))
        torch.cuda.synchronize()
        got[one] = (nchw(y).float().cpu(), nchw(xd.grad).float().cpu(), wd.grad.float().cpu(), bd.grad.float().cpu())
    y, dx, dw, db = got[True]
    assert relerr(y, yr) < TOL[dtype] and relerr(dx, xr.grad) < 2 * TOL[dtype], (relerr(y, yr), relerr(dx, xr.grad))
    assert relerr(dw, wr.grad) < 2 * TOL[dtype] and relerr(db, br.grad) < 2 * TOL[dtype]
    ring = torch.zeros(H, W, dtype=torch.bool)
    ring[[0, 1, H - 2, H - 1], :] = True
    ring[:, [0, 1, W - 2, W - 1]] = True
    assert relerr(dx[:, :, ring], xr.grad[:, :, ring]) < 2 * TOL[dtype], relerr(dx[:, :, ring], xr.grad[:, :, ring])
    for a, c in zip(got[True], got[False]):
        assert relerr(a, c) < 2 * TOL[dtype], relerr(a, c)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Tq,Tkv,heads", [(2, 4096, 64, 1), (2, 1024, 64, 2), (1, 100, 16, 5), (2, 64, 64, 8), (1, 256, 256, 2), (1, 70, 130, 1)])
def test_attention_smallkv(B, Tq, Tkv, heads, dtype):
    from joligen_amd import ops_segformer as S
    C = heads * 32
    q, kv, go = rnd((B, Tq, C), dtype, 9), rnd((B, Tkv, 2 * C), dtype, 10), rnd((B, Tq, C), dtype, 11)
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    qh = qr.view(B, Tq, heads, 32).transpose(1, 2)
    kh = kvr[..., :C].reshape(B, Tkv, heads, 32).transpose(1, 2)
    vh = kvr[..., C:].reshape(B, Tkv, heads, 32).transpose(1, 2)
    orf = (torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(32), -1) @ vh).transpose(1, 2).reshape(B, Tq, C)
    orf.backward(go.float())
    qd, kvd = q.to(D0).requires_grad_(True), kv.to(D0).requires_grad_(True)
    o = S.attention_smallkv(qd, kvd, heads)
    o.backward(go.to(D0))
    torch.cuda.synchronize()
    assert relerr(o, orf) < TOL[dtype], relerr(o, orf)
    assert relerr(qd.grad, qr.grad) < 2 * TOL[dtype] and relerr(kvd.grad, kvr.grad) < 2 * TOL[dtype], (relerr(qd.grad, qr.grad), relerr(kvd.grad, kvr.grad))


@pytest.mark.parametrize("dtype", DTYPES)
def test_resize_concat(dtype):
    from joligen_amd import ops_segformer as S
    B, Ho = 2, 16
    shapes = [(Ho, 32), (8, 64), (4, 40), (2, 16)]
    xs = [rnd((B, c, s, s), dtype, 20 + i) for i, (s, c) in enumerate(shapes)]
    gy = rnd((B, sum(c for _, c in shapes), Ho, Ho), dtype, 30)
    xr = [x.float().requires_grad_(True) for x in xs]
    yr = torch.cat([F.interpolate(x, size=(Ho, Ho), mode="bilinear", align_corners=False) for x in xr], 1)
    yr.backward(gy.float())
    xd = [x.permute(0, 2, 3, 1).contiguous().to(D0).requires_grad_(True) for x in xs]
    y = S.resize_concat(xd, Ho, Ho)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(D0))
    torch.cuda.synchronize()
    assert relerr(y.permute(0, 3, 1, 2), yr) < TOL[dtype]
    for a, b in zip(xd, xr):
        assert relerr(a.grad.permute(0, 3, 1, 2), b.grad) < TOL[dtype], relerr(a.grad.permute(0, 3, 1, 2), b.grad)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", [0, 2])
@pytest.mark.parametrize("separable", [True, False])
def test_resize_sum(dtype, act, separable, monkeypatch):
    """round 6 (`jg_resize_sum`): act(x0 + sum_i F.interpolate(x_i)) and its adjoint against torch, 1 .. 3 resized terms, ragged sizes;
    the adjoint in its separable two-launch form (`jg_resize_sum_bwd`, default) and as activation gradient + one gather launch per term"""
    from joligen_amd import ops_segformer as S
    monkeypatch.setattr(S, "RESIZE_BWD_SEPARABLE", separable)
    B, Ho, Wo, C = 2, 16, 24, 40
    for sizes in ([(8, 12)], [(8, 12), (4, 6), (2, 3)], [(5, 7), (16, 24)], [(2, 3), (1, 1)]):
        x0 = rnd((B, C, Ho, Wo), dtype, 40)
        xs = [rnd((B, C, h, w), dtype, 41 + i) for i, (h, w) in enumerate(sizes)]
        gy = rnd((B, C, Ho, Wo), dtype, 50)
        r0, rs = x0.float().requires_grad_(True), [x.float().requires_grad_(True) for x in xs]
        yr = r0 + sum(F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False) for x in rs)
        yr = torch.relu(yr) if act == 2 else yr
        yr.backward(gy.float())
        d0 = x0.permute(0, 2, 3, 1).contiguous().to(D0).requires_grad_(True)
        ds = [x.permute(0, 2, 3, 1).contiguous().to(D0).requires_grad_(True) for x in xs]
        y = S.resize_sum(d0, ds, act)
        y.backward(gy.permute(0, 2, 3, 1).contiguous().to(D0))
        torch.cuda.synchronize()
        assert relerr(y.permute(0, 3, 1, 2), yr) < TOL[dtype]
        for a, b in zip([d0] + ds, [r0] + rs):
            assert relerr(a.grad.permute(0, 3, 1, 2), b.grad) < TOL[dtype], (sizes, relerr(a.grad.permute(0, 3, 1, 2), b.grad))


@pytest.mark.parametrize("dtype", DTYPES)
def test_resize_sum_with_channel_dropout(dtype):
    """`resize_sum(..., chscale)`: relu(x0 + sum_i resize(x_i)) * s[b, c] with s = Dropout2d's (U >= p) / (1 - p) (zeros included) and its adjoint
    against torch"""
    from joligen_amd import ops_segformer as S
    B, Ho, Wo, C = 3, 16, 24, 40
    x0 = rnd((B, C, Ho, Wo), dtype, 40)
    xs = [rnd((B, C, h, w), dtype, 41 + i) for i, (h, w) in enumerate([(8, 12), (4, 6)])]
    gy = rnd((B, C, Ho, Wo), dtype, 50)
    sc = (torch.rand(B, C, generator=torch.Generator().manual_seed(3)) >= 0.3).float() / 0.7
    assert float(sc.min()) == 0.0
    r0, rs = x0.float().requires_grad_(True), [x.float().requires_grad_(True) for x in xs]
    yr = torch.relu(r0 + sum(F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False) for x in rs)) * sc[:, :, None, None]
    yr.backward(gy.float())
    d0 = x0.permute(0, 2, 3, 1).contiguous().to(D0).requires_grad_(True)
    ds = [x.permute(0, 2, 3, 1).contiguous().to(D0).requires_grad_(True) for x in xs]
    y = S.resize_sum(d0, ds, 2, sc.to(D0))
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(D0))
    torch.cuda.synchronize()
    assert relerr(y.permute(0, 3, 1, 2), yr) < TOL[dtype]
    for a, b in zip([d0] + ds, [r0] + rs):
        assert relerr(a.grad.permute(0, 3, 1, 2), b.grad) < TOL[dtype], relerr(a.grad.permute(0, 3, 1, 2), b.grad)


def test_resize_sum_backward_forms_agree_at_the_head_shape():
    """the two adjoint forms of `resize_sum` on the SegformerHead shape of BASELINE configs[2] (64 x 64 x 256 against 32 / 16 / 8, ReLU): same
    g = dy relu'(y), the term gradients agree to fp32 summation order before the 16-bit store; a term that wants no gradient is skipped"""
    from joligen_amd import ops_segformer as S
    B, Ho, C = 4, 64, 256
    x0 = rnd((B, Ho, Ho, C), torch.bfloat16, 80).to(D0)
    xs = [rnd((B, Ho >> (i + 1), Ho >> (i + 1), C), torch.bfloat16, 81 + i).to(D0) for i in range(3)]
    gy = rnd((B, Ho, Ho, C), torch.bfloat16, 90).to(D0)
    res = {}
    for sep in (True, False):
        S.RESIZE_BWD_SEPARABLE = sep
        try:
            d0 = x0.clone().requires_grad_(True)
            ds = [x.clone().requires_grad_(i != 1) for i, x in enumerate(xs)]
            S.resize_sum(d0, ds, 2).backward(gy)
            torch.cuda.synchronize()
        finally:
            S.RESIZE_BWD_SEPARABLE = True
        assert ds[1].grad is None
        res[sep] = [d0.grad.float(), ds[0].grad.float(), ds[2].grad.float()]
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1:], res[False][1:]):
        assert relerr(a, b) < 2e-3, relerr(a, b)


@pytest.mark.parametrize("dtype", DTYPES)
def test_segformer_head_fusion_before_resize(dtype, monkeypatch):
    """round 6 (`JG_HEAD_COMMUTE`, default on): SegformerHead with its 1x1 fusion convolution applied to every stage's map BEFORE the bilinear
    resize (sum of four resized C-channel maps) against the reference's structure in fp32 torch -- resize, concatenate, convolve (mmseg
    SegformerHead.forward as used by segformer_generator.py) -- output, input gradients and the gradients of every weight and bias; and against
    the concatenating path of rounds 3-5 (`HEAD_COMMUTE = False`)."""
    from joligen_amd.arena import ParamArena
    from joligen_amd.modules import segformer as M

    torch.manual_seed(11)
    in_ch, ch, ncls, B, Ho = [32, 64, 160, 256], 64, 16, 2, 16
    head = M.SegformerHead(None, in_ch, [0, 1, 2, 3], ch, dropout_ratio=0.0, num_classes=ncls)
    with torch.no_grad():
        for p_ in head.parameters():
            p_.copy_(p_.to(dtype).float())
    ref = {k: v.detach().clone().requires_grad_(True) for k, v in head.named_parameters()}
    ParamArena(head, torch.device(D0), dtype, priority=()).refresh()
    xs = [rnd((B, c, Ho >> i, Ho >> i), dtype, 60 + i) for i, c in enumerate(in_ch)]
    xr = [x.float().requires_grad_(True) for x in xs]
    outs = [F.interpolate(torch.relu(F.conv2d(x, ref["convs.%d.conv.weight" % i], ref["convs.%d.conv.bias" % i])), size=(Ho, Ho), mode="bilinear",
                          align_corners=False) for i, x in enumerate(xr)]
    hr = torch.relu(F.conv2d(torch.cat(outs, 1), ref["fusion_conv.conv.weight"], ref["fusion_conv.conv.bias"]))
    yr = F.conv2d(hr, ref["conv_seg.weight"], ref["conv_seg.bias"])
    gy = rnd(tuple(yr.shape), dtype, 70)
    yr.backward(gy.float())
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(M, "HEAD_COMMUTE", mode)
        for p_ in head.parameters():
            p_.grad.zero_()
        xd = [x.permute(0, 2, 3, 1).contiguous().to(D0).requires_grad_(True) for x in xs]
        y = head(xd)
        y.backward(gy.permute(0, 2, 3, 1).contiguous().to(D0)[..., :y.shape[-1]] if y.shape[-1] == ncls else
                   F.pad(gy.permute(0, 2, 3, 1), (0, y.shape[-1] - ncls)).contiguous().to(D0))
        torch.cuda.synchronize()
        res[mode] = (y[..., :ncls].permute(0, 3, 1, 2).float().cpu(), [x.grad.permute(0, 3, 1, 2).float().cpu() for x in xd],
                     {k: v.grad.detach().float().cpu().clone() for k, v in head.named_parameters()})
    for mode in (True, False):
        y, gx, gp = res[mode]
        assert relerr(y, yr) < TOL[dtype], (mode, relerr(y, yr))
        for a, b in zip(gx, xr):
            assert relerr(a, b.grad) < 2 * TOL[dtype], (mode, relerr(a, b.grad))
        for k in ref:
            assert relerr(gp[k], ref[k].grad) < 3 * TOL[dtype], (mode, k, relerr(gp[k], ref[k].grad))      # (bias gradients: sums with cancellation)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("training", [True, False])
def test_batch_norm_relu(training, dtype):
    from joligen_amd import ops_segformer as S
    B, C, H = 3, 64, 12
    x, gy = rnd((B, C, H, H), dtype, 40, 2.0), rnd((B, C, H, H), dtype, 41)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * rnd((C,), torch.float32, 42))
        bn.bias.copy_(0.2 * rnd((C,), torch.float32, 43))
        bn.running_mean.copy_(0.1 * rnd((C,), torch.float32, 44))
        bn.running_var.copy_(0.5 + rnd((C,), torch.float32, 45).abs())
    bn.train(training)
    import copy
    bd = copy.deepcopy(bn).to(D0)
    for p in bd.parameters():
        p.grad = torch.zeros_like(p)
    xr = x.float().requires_grad_(True)
    yr = F.relu(bn(xr))
    yr.backward(gy.float())
    xd = x.permute(0, 2, 3, 1).contiguous().to(D0).requires_grad_(True)
    y = S.batch_norm(xd, bd, S.JG_ACT_RELU)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(D0))
    torch.cuda.synchronize()
    assert relerr(y.permute(0, 3, 1, 2), yr) < TOL[dtype] and relerr(xd.grad.permute(0, 3, 1, 2), xr.grad) < 2 * TOL[dtype]
    assert relerr(bd.weight.grad, bn.weight.grad) < 2 * TOL[dtype] and relerr(bd.bias.grad, bn.bias.grad) < 2 * TOL[dtype]
    assert relerr(bd.running_mean, bn.running_mean) < 1e-3 and relerr(bd.running_var, bn.running_var) < 1e-3
    assert int(bd.num_batches_tracked) == int(bn.num_batches_tracked)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_compose(dtype):
    from joligen_amd import ops_segformer as S
    B, Sz, f, na, ni, nc = 2, 32, 4, 10, 9, 3
    img, logits, xin, go = rnd((B, 27, Sz, Sz), dtype, 50), rnd((B, na, Sz // f, Sz // f), dtype, 51, 2.0), rnd((B, 3, Sz, Sz), dtype, 52), rnd((B, 3, Sz, Sz), dtype, 53)
    ir, lr, xr = img.float().requires_grad_(True), logits.float().requires_grad_(True), xin.float().requires_grad_(True)
    att = F.interpolate(torch.softmax(lr, 1), size=(Sz, Sz))
    outr = sum(ir[:, 3 * i:3 * i + 3] * att[:, i:i + 1] for i in range(ni)) + xr * att[:, 9:10]
    outr.backward(go.float())

    def pad(t, c):
        t = t.permute(0, 2, 3, 1)
        return torch.cat([t, torch.zeros(*t.shape[:3], c - t.shape[-1], dtype=t.dtype)], -1).contiguous().to(D0)

    idv, ld, xd = pad(img, 32).requires_grad_(True), pad(logits, 16).requires_grad_(True), pad(xin, 8).requires_grad_(True)
    out = S.attention_compose(idv, ld, xd, na, ni, nc)
    out.backward(pad(go, 8))
    torch.cuda.synchronize()
    assert relerr(out[..., :3].permute(0, 3, 1, 2), outr) < TOL[dtype] and float(out[..., 3:].float().abs().max()) == 0
    assert relerr(idv.grad[..., :27].permute(0, 3, 1, 2), ir.grad) < TOL[dtype]
    assert relerr(ld.grad[..., :na].permute(0, 3, 1, 2), lr.grad) < 2 * TOL[dtype], relerr(ld.grad[..., :na].permute(0, 3, 1, 2), lr.grad)
    assert relerr(xd.grad[..., :3].permute(0, 3, 1, 2), xr.grad) < TOL[dtype]
    assert float(idv.grad[..., 27:].float().abs().max()) == 0 and float(ld.grad[..., na:].float().abs().max()) == 0


def load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", ["s64", "s128"])
def test_segformer_generator_vs_reference_golden(golden_dir, name, dtype):
    from joligen_amd import ops
    from joligen_amd.modules.segformer import SegformerGenerator_attn

    g = load(golden_dir, f"segformer_{name}.pt")
    c = g["cfg"]
    net = SegformerGenerator_attn(None, None, 3, c["S"], 10, 1)
    assert list(net.state_dict().keys()) == g["keys"]
    net.load_state_dict(O.synth_state_dict(net.state_dict(), seed=4))
    net.jg_finalize(torch.device(D0), dtype)
    net.train()
    it = iter(g["rands"])
    net.rand.source = lambda shape: next(it)
    x = ops.to_nhwc(g["x"].to(D0), dtype, 8).requires_grad_(True)
    out = net(x)
    e = relerr(out.permute(0, 3, 1, 2)[:, :3], g["out"])
    assert e < 3 * TOL[dtype], e
    out.backward(ops.to_nhwc(g["R"].to(D0), dtype, 8))
    torch.cuda.synchronize()
    for k, v in g["bn_after"].items():
        mine = dict(net.state_dict())[k]
        assert relerr(mine.float(), v.float()) < 5e-3, (k, relerr(mine.float(), v.float()))
    e = relerr(x.grad.permute(0, 3, 1, 2)[:, :3], g["dx"])
    assert e < (0.1 if dtype == torch.float16 else 0.35), e
    bad = []
    P = dict(net.named_parameters())
    for k, ref in g["grad_checks"].items():
        v = P[k].grad.detach().float().cpu()
        # 16-bit activations through 8 attention / MixFFN blocks + the BatchNorm tail: the gradient norms of the early layers
        # carry a few per cent of rounding noise (the per-kernel tests above hold every op to 3e-3 / 2e-2)
        tol = (0.12 if dtype == torch.float16 else 0.3) * float(ref[0]) + 1e-6
        if abs(float(v.norm()) - float(ref[0])) > tol:
            bad.append((k, round(float(v.norm()) / float(ref[0]), 3)))
    assert not bad, (len(bad), bad[:12])
    with torch.no_grad():
        feats = net.get_feats(x.detach(), [0, 1, 2, 3])          # consumes the remaining recorded draws
    for f, ref in zip(feats, g["feats"]):
        assert relerr(f.permute(0, 3, 1, 2), ref) < 3 * TOL[dtype], (tuple(ref.shape), relerr(f.permute(0, 3, 1, 2), ref))
    net.eval()
    with torch.no_grad():
        oe = net(x.detach())
    assert relerr(oe.permute(0, 3, 1, 2)[:, :3], g["out_eval"]) < 3 * TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_sync_batchnorm_path_matches_local_path(dtype):
    """under data parallelism the reference's BatchNorm layers are SyncBatchNorm (base_model.py:725-737).  The synchronised
    path (local sums -> [C, 2] all-reduce -> coefficient kernels over "one image of world * B * HW pixels"; local dgamma / dbeta, global
    dx reductions) run with a single rank must reproduce the plain path: output, input gradient, affine gradients, running statistics."""
    import torch.nn as nn

    from joligen_amd import ops_segformer as S
    from joligen_amd.ops import JG_ACT_RELU

    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(8)
    x = (torch.randn(3, 12, 10, 64, generator=g) * 1.5 + 0.3).to(dtype)
    gy = torch.randn(3, 12, 10, 64, generator=g).to(dtype)
    res = {}
    for sync in (False, True):
        bn = nn.BatchNorm2d(64).to(d)
        with torch.no_grad():
            bn.weight.copy_(1 + 0.1 * torch.randn(64, generator=g.manual_seed(9)))
            bn.bias.copy_(0.1 * torch.randn(64, generator=g))
        bn.weight.grad, bn.bias.grad = torch.zeros_like(bn.weight), torch.zeros_like(bn.bias)
        bn.train()
        S.FORCE_SYNC_BN = sync
        try:
            xd = x.to(d).requires_grad_(True)
            y = S.batch_norm(xd, bn, JG_ACT_RELU)
            y.backward(gy.to(d))
            torch.cuda.synchronize()
        finally:
            S.FORCE_SYNC_BN = False
        res[sync] = (y.detach().float(), xd.grad.float(), bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_mean.clone(), bn.running_var.clone())
    for a, b in zip(res[True], res[False]):
        assert relerr(a, b) < 1e-5, relerr(a, b)
