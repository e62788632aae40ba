"""Dev tool (GPU box): per-parameter gradient error of the first CUT G-group backward against the CPU oracle.
usage: python tools/dbg_cut_grads.py [config0|monce|patchnce] [fp16|bf16] [nce_T override]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import jg_oracle as O
from test_gpu_5_cutloss import build_cut_model, relerr
from test_oracle_golden import cut_ids, cut_trainer_for

name = sys.argv[1] if len(sys.argv) > 1 else "config0"
dtype = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "fp16") else torch.bfloat16
if name == "custom":       # synthetic configuration from the environment: NGF NB NDF S B P LAYERS NCE
    E = os.environ.get
    c = dict(ngf=int(E("NGF", 64)), n_blocks=int(E("NB", 9)), ndf=int(E("NDF", 64)), S=int(E("S", 128)), B=int(E("B", 1)),
             nce_layers=E("LAYERS", "0,4,8,12,16"), num_patches=int(E("P", 256)), nce_loss=E("NCE", "monce"), pool=50, iters=1)
    from joligen_amd.modules.discriminators import NLayerDiscriminator
    from joligen_amd.modules.resnet_generator import ResnetGenerator
    gen = torch.Generator().manual_seed(7)
    A0 = torch.rand(c["B"], 3, c["S"], c["S"], generator=gen) * 2 - 1
    B0 = torch.rand(c["B"], 3, c["S"], c["S"], generator=gen) * 2 - 1
    nG, nD = ResnetGenerator(3, 3, c["ngf"], n_blocks=c["n_blocks"]), NLayerDiscriminator(3, c["ndf"])
    layers = [int(i) for i in c["nce_layers"].split(",")]
    chans = nG.feat_channels(layers)
    sdF = {}
    for i, ch in enumerate(chans):
        sdF.update({f"mlp_{i}.0.weight": torch.empty(256, ch), f"mlp_{i}.0.bias": torch.empty(256), f"mlp_{i}.2.weight": torch.empty(256, 256),
                    f"mlp_{i}.2.bias": torch.empty(256)})
    with torch.no_grad():
        fs = O.resnet_encoder(O.synth_state_dict(nG.state_dict(), 0), A0, c["n_blocks"], layers)[1]
    perms = [torch.randperm(f.shape[2] * f.shape[3], generator=gen)[: c["num_patches"]] for f in fs] * 2
    g = dict(cfg=c, hp=dict(lr_G=2e-4, lr_D=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, ema_beta=0.999, T=0.07, lambda_NCE=1.0, lambda_GAN=1.0),
             steps=[dict(A=A0, B=B0, perms=perms, pool_draws=[])], keysG=list(nG.state_dict().keys()),
             shapesG={k: tuple(v.shape) for k, v in nG.state_dict().items()}, keysD=list(nD.state_dict().keys()),
             shapesD={k: tuple(v.shape) for k, v in nD.state_dict().items()}, keysF=list(sdF.keys()), shapesF={k: tuple(v.shape) for k, v in sdF.items()})
else:
    g = torch.load(os.path.join(ROOT, "tests", "golden", f"cutstep_{name}.pt"), weights_only=False)
    c = g["cfg"]
T = float(sys.argv[3]) if len(sys.argv) > 3 else g["hp"]["T"]
model = build_cut_model(g, dtype)
model.opt.alg_cut_nce_T = T
if len(sys.argv) > 4:
    model.loss_scale = float(sys.argv[4])
LG = float(os.environ.get("LAMBDA_GAN", "1"))
LN = float(os.environ.get("LAMBDA_NCE", "1"))
model.opt.alg_gan_lambda, model.opt.alg_cut_lambda_NCE = LG, LN
s = g["steps"][0]
model.data_dependent_initialize({"A": s["A"], "B": s["B"]})
rd = (lambda v: v.to(dtype).float())
sdG = {k: rd(v) for k, v in O.synth_state_dict(model.netG_A.state_dict(), seed=0).items()}
sdD = {k: rd(v) for k, v in O.synth_state_dict(model.netD_B_basic.state_dict(), seed=1).items()}
sdF = O.synth_state_dict(model.netF.state_dict(), seed=3)
model.netG_A.load_state_dict(sdG)
model.netD_B_basic.load_state_dict(sdD)
model.netF.load_state_dict(sdF)
nl = len(c["nce_layers"].split(","))
ids_ab, ids_idt = cut_ids(s, nl, c["num_patches"])
model.patch_ids_injection = lambda call, shapes: [i.to("cuda:0") for i in (ids_ab if call == 0 else ids_idt)]
A, Bi = rd(s["A"]), rd(s["B"])
DETACH = os.environ.get("DETACH", "")
if DETACH:
    # both sides call the sampler for k first, then for q (cut_model.py:873-885)
    netF_fwd, cnt = model.netF.forward, [0, 0]

    def f_mine(*a, **k):
        feats, ids = netF_fwd(*a, **k)
        which = "kq"[cnt[0] % 2]
        cnt[0] += 1
        return ([f.detach() for f in feats] if which == DETACH else feats), ids
    model.netF.forward = f_mine
    psf = O.patch_sample_f

    def f_oracle(*a, **k):
        feats = psf(*a, **k)
        which = "kq"[cnt[1] % 2]
        cnt[1] += 1
        return [f.detach() for f in feats] if which == DETACH else feats
    O.patch_sample_f = f_oracle
model.set_input({"A": A, "B": Bi})
for net in ("G_A", "F", "D_B_basic"):
    model.set_requires_grad(getattr(model, "net" + net), net != "D_B_basic")
model.forward()
model.compute_G_loss()
model.loss_G_tot.backward()
torch.cuda.synchronize()
tr, _ = cut_trainer_for(g)
tr.cfg["T"] = T
tr.cfg["lambda_GAN"], tr.cfg["lambda_NCE"] = LG, LN
tr.G, tr.D = {k: v.clone() for k, v in sdG.items()}, {k: v.clone() for k, v in sdD.items()}
tr.pool.rng = tr.real_pools[0].rng = tr.real_pools[1].rng = random.Random(0)
tr.step(A, Bi, ids_ab, ids_idt)
print("losses mine", {k: round(float(getattr(model, "loss_" + k)), 5) for k in model.loss_names_G}, "oracle", tr.losses)
print("fake_B relerr", relerr(model.fake_B.permute(0, 3, 1, 2)[:, :3].float(), tr.fake_B))
rows = []
for net, key in ((model.netG_A, "G"), (model.netF, "F")):
    for k, p in net.named_parameters():
        ref = tr.last_grads[key][k]
        mine = p.grad.detach().float().cpu() / model.loss_scale
        rows.append((relerr(mine, ref), key, k, float(ref.norm()), float(mine.norm())))
rows.sort(reverse=True)
for r in (rows if os.environ.get("ALL") else rows[:25]):
    print("%.4f %s %-40s ref %.3e mine %.3e" % r)
for r in [r for r in rows if r[1] == "F"]:
    print("%.4f %s %-40s ref %.3e mine %.3e" % r)
print("median relerr", sorted(r[0] for r in rows)[len(rows) // 2])
