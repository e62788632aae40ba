"""dev: device time of the small-KV attention (T_kv = 64, head dim 32) forward / backward at the MiT-B0 stage shapes of the CUT configs[2]
step (batch 32 = generator pass, 64 = batched NCE encoder pass) against the bytes each launch has to move.
usage (GPU box): python tools/attn_kv64_probe.py > gpurun_out/attn_kv64_probe.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joligen_amd import ops_segformer as S  # noqa: E402

D = "cuda:0"


def timed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / reps


for B in (32, 64):
    for Tq, heads in ((4096, 1), (1024, 2), (256, 5), (64, 8)):
        C = heads * 32
        q = torch.randn(B, Tq, C, device=D, dtype=torch.bfloat16, requires_grad=True)
        kv = torch.randn(B, 64, 2 * C, device=D, dtype=torch.bfloat16, requires_grad=True)
        o = S.attention_smallkv(q, kv, heads)
        do = torch.randn_like(o)
        fwd = timed(lambda: S.attention_smallkv(q, kv, heads))
        bwd = timed(lambda: o.backward(do, retain_graph=True))
        mb_f = 2 * B * Tq * C * 2 / 1e6
        mb_b = 4 * B * Tq * C * 2 / 1e6
        print(f"B {B} Tq {Tq} heads {heads}: fwd {fwd:7.1f} us ({mb_f:6.1f} MB = {mb_f / 5e6 * 1e6:5.1f} us at 5 TB/s)   "
              f"bwd (memset + kernel + convert) {bwd:7.1f} us ({mb_b:6.1f} MB = {mb_b / 5e6 * 1e6:5.1f} us)", flush=True)
