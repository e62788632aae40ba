"""ring (NST-stage) form of the generic LDS-DMA GEMM against the double-buffered form: the ViT projector's token GEMMs, SegFormer linears, the
UNet's 1x1 layers at 32 x 32 and a few im2col shapes; outputs must be bit-equal (same accumulation order).  Dev tool (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from joligen_amd import _lib, ops
from tools.conv_bench import timeit

d, dt = torch.device("cuda:0"), torch.bfloat16
ws = torch.empty(64 << 20, device=d, dtype=torch.uint8)
SHAPES = [  # B, H, W, Cin, Cout, ks, stride
    (16, 1, 257, 384, 1536, 1, 1), (16, 1, 257, 1536, 384, 1, 1), (16, 1, 257, 384, 1152, 1, 1), (16, 1, 257, 384, 384, 1, 1), (16, 1, 257, 1152, 384, 1, 1),
    (32, 32, 32, 512, 1024, 1, 1), (32, 32, 32, 1024, 512, 1, 1), (32, 1, 1024, 512, 1536, 1, 1), (32, 1, 1024, 512, 512, 1, 1), (32, 64, 64, 768, 256, 1, 1),
    (64, 16, 16, 640, 160, 1, 1), (64, 16, 16, 160, 640, 1, 1), (64, 1, 256, 160, 160, 1, 1), (32, 1, 64, 256, 256, 1, 1), (64, 8, 8, 1024, 256, 1, 1),
    (16, 32, 32, 512, 256, 4, 1), (16, 64, 64, 256, 128, 4, 1), (16, 31, 31, 256, 512, 4, 1), (32, 64, 64, 128, 256, 3, 1), (16, 32, 32, 128, 256, 4, 2),
]
for (B, H, W, Cin, Cout, ks, stride) in SHAPES:
    pad = 0 if ks == 1 else (1 if ks in (3, 4) else 0)
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    x = torch.randn(B, H, W, Cin, device=d).to(dt)
    w = (torch.randn(Cout, ks, ks, Cin, device=d) / (ks * ks * Cin) ** 0.5).to(dt)
    bias = torch.randn(Cout, device=d)
    geo = dict(B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=ks, S=ks, pad=pad, stride=stride, Ho=Ho, Wo=Wo, ldx=Cin, ldw=ks * ks * Cin, ldy=Cout)
    line = f"{str((B, H, W, Cin, Cout, ks, stride)):38s}"
    outs = []
    for ring in (0, 1, 2):
        _lib.set_tuning("JG_CONV_RING", ring)
        y = torch.empty(B, Ho, Wo, Cout, device=d, dtype=dt)
        ops.conv_nt(x, w, y, bias=bias, **geo)
        torch.cuda.synchronize()
        t = timeit(lambda: ops.conv_nt(x, w, y, bias=bias, **geo), reps=30)
        outs.append(y)
        line += f"  ring{ring} {(_lib.lib().jg_last_kernel().decode() or '?')[-26:]:>26s} {t * 1e6:6.1f} us {2.0 * B * Ho * Wo * Cin * ks * ks * Cout / t / 1e12:5.0f} TF"
    _lib.set_tuning("JG_CONV_RING", 1)
    line += "  equal" if all(torch.equal(outs[0], o) for o in outs[1:]) else f"  DIFFER {max((outs[0].float() - o.float()).abs().max().item() for o in outs[1:]):.3g}"
    print(line, flush=True)
