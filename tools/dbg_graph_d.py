"""Dev tool (GPU box): the CUT discriminator half replayed from a hipGraph (`jg_graph_D`) against the eager half on the side stream, step by
step -- D loss, gradient norm and parameter norm of every discriminator arena.  This is the tool the two findings of DESIGN.md 11.2
came from; `profiles/r04_graph_replay_probe.txt` is its DBG_POKE output with the runtime's AQL-packet capture ON.

    python tools/dbg_graph_d.py [train_iter_size, default 4]          # 2 * iter_size + 1 calls of optimize_parameters() per driver

environment:
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=1   the ROCm 7.2 default that corrupts interleaved replays (bench.py / joligen_amd set 0 when unset)
    DBG_SMALL=1        64 x 64 model with frozen parameters (resnet G, ngf 32) instead of the configs[2] shape at batch 4
    DBG_NETDS=...      discriminators, default "projected_d,basic"
    DBG_POKE=1         after the capturing step: replay the untouched graph after each of a list of eager activities
    DBG_REPLAY=1       after the capturing step: three extra replays in a row
    DBG_GRADS=1        per-parameter gradient norms of the discriminators at call iter_size + 1
    DBG_INTER=1        norms of the graph's static inputs, D(real) and the 16-bit weight copies after every call
    JG_DBG_EARLY_D_SYNC=1   (cut_model) device synchronisation between the discriminator half and the generator's backward
"""
import argparse
import gc
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

SMALL = bool(os.environ.get("DBG_SMALL"))
NETDS = os.environ.get("DBG_NETDS", "projected_d,basic")
ITS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
DEV = torch.device("cuda", 0)
_g = torch.Generator().manual_seed(77)
_B, _S = (2, 64) if SMALL else (4, 256)
BATCH = {"A": (torch.rand(_B, 3, _S, _S, generator=_g) * 2 - 1).to(DEV), "B": (torch.rand(_B, 3, _S, _S, generator=_g) * 2 - 1).to(DEV)}


def make(graph):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if SMALL:
            from joligen_amd.models import create_model
            from joligen_amd.options import opt_from_json

            cfg = {"model_type": "cut", "G": {"netG": "resnet", "ngf": 32, "nblocks": 2}, "D": {"netDs": NETDS.split(","), "ndf": 32, "proj_interp": 128},
                   "alg": {"cut": {"nce_layers": "0,4,8"}}, "data": {"crop_size": 64, "load_size": 64},
                   "train": {"batch_size": 2, "G_ema": True, "iter_size": ITS, "pool_size": 0, "G_lr": 0.0, "D_lr": 0.0}}
            model = create_model(opt_from_json(cfg, overrides={"jg_act_dtype": "bf16", "gpu_ids": "0", "jg_graph_D": graph}), 0)
            model.data_dependent_initialize(BATCH)
            model.setup(model.opt)
            model.single_gpu()
        else:
            ns = argparse.Namespace(model="cut", netG="segformer_attn_conv", netDs=NETDS, batch=4, size=256, dtype="bf16", efficient=1, force_exchange=False)
            model, _ = bench.build_model(ns, 0, 0, 1)
            model.opt.jg_graph_D = graph
            model.opt.train_iter_size = ITS
            if ITS > 1:
                model.iter_calculator_init()
    for p in (model.real_A_pool, model.real_B_pool, model.fake_B_pool):
        p.pool_size = 0
    return model


def poke(m):
    """replay the captured, untouched graph after each eager activity: every line should print the same loss"""
    st, dn = m._dg, m.discriminators_names[0]

    def replay(tag):
        st["graph"].replay()
        torch.cuda.synchronize()
        print("   ", tag, [float(v) for v in st["vals"]])

    replay("replay")
    with torch.no_grad():
        m._net(dn)(st["real"])
    torch.cuda.synchronize()
    replay("after an eager no-grad D forward on the default stream")
    y = m._net(dn)(st["real"])
    torch.cuda.synchronize()
    replay("after an eager D forward with autograd")
    y.float().sum().backward()
    torch.cuda.synchronize()
    replay("after its backward")
    junk = [torch.empty(1 << 20, device=DEV) for _ in range(64)]
    del junk
    torch.cuda.synchronize()
    replay("after eager allocations")
    gc.collect()
    torch.cuda.synchronize()
    replay("after gc.collect()")
    t = torch.zeros(1024, device=DEV)
    for n in (1000, 3000, 10000):
        for _ in range(n):
            t.add_(1.0)
        torch.cuda.synchronize()
        replay("after %d tiny eager launches on the default stream" % n)
    m.set_input(BATCH)
    m._group_flags(m.group_G)
    m.forward()
    torch.cuda.synchronize()
    replay("after an eager G forward")
    m.compute_G_loss()
    torch.cuda.synchronize()
    replay("after compute_G_loss")
    m.loss_G_tot.backward()
    torch.cuda.synchronize()
    replay("after the G backward")


def main():
    for graph in (False, True):
        os.environ["JG_GRAPH_D"] = "1" if graph else "0"
        torch.manual_seed(0)
        m = make(graph)
        print("graph" if graph else "eager")
        for i in range(1, 2 * ITS + 2):
            m.set_input(BATCH)
            m.optimize_parameters()
            torch.cuda.synchronize()
            row = ["%.5f" % float(m.loss_D_tot.detach())]
            for dn in m.discriminators_names:
                a = m._net(dn).arena
                row.append("%s: |g| %.4e |p| %.6e" % (dn, float(a.g.norm()), float(a.p.norm())))
            print(i, " | ".join(row))
            st = getattr(m, "_dg", None) if graph else None
            if st is None:
                continue
            if i == 3 and os.environ.get("DBG_REPLAY"):
                for r in range(3):
                    st["graph"].replay()
                    torch.cuda.synchronize()
                    print("   extra replay", r, float(st["tot"]), [float(v) for v in st["vals"]])
            if i == ITS + 1 and os.environ.get("DBG_GRADS"):
                for dn in m.discriminators_names:
                    for name, prm in m._net(dn).named_parameters():
                        print("   ", dn, name, tuple(prm.shape), "%.4e" % float(prm.grad.float().norm()))
            if os.environ.get("DBG_INTER"):
                c = getattr(m, m.discriminators_names[0] + "_loss_calculator")
                a = m._net(m.discriminators_names[0]).arena
                print("    real %.5e fake %.5e pred_real %.5e w16 %.6e w16T %.6e vals %s" % (
                    float(st["real"].float().norm()), float(st["fakes"][0].float().norm()), float(c.pred_real.float().norm()),
                    float(a.w16.float().norm()), float(a.w16T.float().norm()), [float(v) for v in st["vals"]]))
            if i == 3 and os.environ.get("DBG_POKE"):
                poke(m)


if __name__ == "__main__":
    main()
