"""wgrad_tn_tr_kernel with one / two register stages of global loads in flight (JG_WGRAD_DEEP 0 / 3) on the UNet's 1x1 weight gradients and a few
CUT shapes.  Dev tool (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from joligen_amd import _lib, ops
from joligen_amd.ops import JG_OUT_ATOMIC_F32
from tools.conv_bench import timeit

d, dt = torch.device("cuda:0"), torch.bfloat16
SH = [(32, 256, 256, 128, 64, 1, 512), (32, 256, 256, 192, 64, 1, 512), (32, 256, 256, 8, 64, 3, 512), (32, 128, 128, 384, 128, 1, 512), (32, 128, 128, 256, 128, 1, 512),
      (32, 64, 64, 768, 256, 1, 128), (32, 64, 64, 512, 256, 1, 192), (32, 32, 32, 1024, 512, 1, 48), (32, 1, 1024, 512, 1536, 1, 32), (32, 64, 64, 1024, 256, 1, 96),
      (64, 1, 4096, 32, 32, 1, 512), (64, 16, 16, 640, 160, 1, 64), (16, 1, 257, 384, 1536, 1, 16)]
for (B, H, W, Cin, Cout, ks, splitk) in SH:
    pad = ks // 2
    x = torch.randn(B, H, W, Cin, device=d).to(dt)
    dy = torch.randn(B, H, W, Cout, device=d).to(dt)
    line = f"{str((B, H, W, Cin, Cout, ks, splitk)):40s}"
    outs = []
    for deep in (0, 3):
        _lib.set_tuning("JG_WGRAD_DEEP", deep)
        prev = _lib.set_tuning("JG_WGRAD_VARIANT", 2)
        dw = torch.zeros(Cout, ks, ks, Cin, device=d)
        db = torch.zeros(Cout, device=d)
        kw = dict(B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=ks, S=ks, pad=pad, stride=1, Ho=H, Wo=W, lddy=Cout, ldx=Cin, lddw=ks * ks * Cin, dbias=db, splitk=splitk,
                  out_mode=JG_OUT_ATOMIC_F32)
        ops.wgrad_tn(dy, x, dw, **kw)
        torch.cuda.synchronize()
        outs.append((dw.clone(), db.clone()))
        t = timeit(lambda: ops.wgrad_tn(dy, x, dw, **kw), reps=20)
        _lib.set_tuning("JG_WGRAD_VARIANT", prev)
        gb = 2.0 * B * H * W * (Cin + Cout) / 1e9
        line += f"  deep{deep} {(_lib.lib().jg_last_kernel().decode() or '?')[-22:]:>22s} {t * 1e6:7.1f} us {gb / t / 1e3:5.2f} TB/s"
    _lib.set_tuning("JG_WGRAD_DEEP", 0)
    e = float((outs[0][0] - outs[1][0]).norm() / outs[0][0].norm())
    eb = float((outs[0][1] - outs[1][1]).norm() / outs[0][1].norm())
    print(line + f"  rel diff {e:.1e} / bias {eb:.1e}", flush=True)
