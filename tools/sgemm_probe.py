import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from joligen_amd import ops, _lib
d = "cuda:0"
M, N, K = 16384, 256, 256
A = torch.randn(M, K, device=d); B = torch.randn(N, K, device=d); C = torch.empty(M, N, device=d)
At = A.t().contiguous(); 
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for split in (0, 1):
    _lib.set_tuning("JG_SGEMM_SPLIT", split)
    a = t(lambda: ops.sgemm(A, B, C, M, N, K, (K, 1), (K, 1), (N, 1)))
    # dW form: [N x K'] = dy^T x : A = dy viewed [N][M] (stride (1, N)), B = x [K][M] (stride (1,K))
    dW = torch.empty(N, K, device=d)
    b = t(lambda: ops.sgemm(C, A, dW, N, K, M, (1, N), (1, K), (K, 1)))
    print("split", split, "fwd %.1f us, dW %.1f us" % (a, b))
