"""dev: D half from a hipGraph vs eager, step by step (losses, gradient norms of the discriminator arenas)"""
import argparse, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def make(graph, its):
    if os.environ.get("DBG_SMALL"):
        from joligen_amd.models import create_model
        from joligen_amd.options import opt_from_json
        cfg = {"model_type": "cut", "G": {"netG": "resnet", "ngf": 32, "nblocks": 2}, "D": {"netDs": os.environ.get("DBG_NETDS", "projected_d,basic").split(","), "ndf": 32, "proj_interp": 128},
               "alg": {"cut": {"nce_layers": "0,4,8"}}, "data": {"crop_size": 64, "load_size": 64},
               "train": {"batch_size": 2, "G_ema": True, "iter_size": its, "pool_size": 0, "G_lr": 0.0, "D_lr": 0.0}}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = create_model(opt_from_json(cfg, overrides={"jg_act_dtype": "bf16", "gpu_ids": "0", "jg_graph_D": graph}), 0)
            model.data_dependent_initialize(batch)
            model.setup(model.opt)
            model.single_gpu()
        return model
    ns = argparse.Namespace(model="cut", netG="segformer_attn_conv", netDs=os.environ.get("DBG_NETDS", "projected_d,basic"), batch=4, size=256, dtype="bf16", efficient=1, force_exchange=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, _ = bench.build_model(ns, 0, 0, 1)
    model.opt.jg_graph_D = graph
    model.opt.train_iter_size = its
    if its > 1:
        model.iter_calculator_init()
    return model

its = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(77)
SZ = (2, 64) if os.environ.get("DBG_SMALL") else (4, 256)
batch = {"A": (torch.rand(SZ[0], 3, SZ[1], SZ[1], generator=g) * 2 - 1).to(dev), "B": (torch.rand(SZ[0], 3, SZ[1], SZ[1], generator=g) * 2 - 1).to(dev)}
for graph in (False, True):
    os.environ["JG_GRAPH_D"] = "1" if graph else "0"
    torch.manual_seed(0)
    m = make(graph, its)
    import random
    for p in (m.real_A_pool, m.real_B_pool, m.fake_B_pool):
        p.pool_size = 0
    print("graph" if graph else "eager")
    for i in range(2 * its + 1):
        m.set_input(batch)
        m.optimize_parameters()
        torch.cuda.synchronize()
        row = [f"{float(getattr(m, 'loss_D_tot', float('nan'))):.5f}"]
        for dn in m.discriminators_names:
            a = m._net(dn).arena
            row.append(f"{dn}: |g| {float(a.g.norm()):.4e} |p| {float(a.p.norm()):.6e}")
        print(i + 1, " | ".join(row))
        if graph and i + 1 == 3 and os.environ.get("DBG_REPLAY"):
            st = m._dg
            for r in range(3):
                st["graph"].replay()
                torch.cuda.synchronize()
                print("   extra replay", r, float(st["tot"]), [float(v) for v in st["vals"]])
        if graph and i + 1 == its + 1 and os.environ.get("DBG_GRADS"):
            for dn in m.discriminators_names:
                for name, prm in m._net(dn).named_parameters():
                    print("   ", dn, name, tuple(prm.shape), "%.4e" % float(prm.grad.float().norm()))
        if graph and os.environ.get("DBG_INTER") and getattr(m, "_dg", None):
            st = m._dg
            c = getattr(m, m.discriminators_names[0] + "_loss_calculator")
            a = m._net(m.discriminators_names[0]).arena
            print("    real %.5e fake %.5e pred_real %.5e w16 %.6e w16T %.6e vals %s" % (float(st["real"].float().norm()), float(st["fakes"][0].float().norm()),
                  float(c.pred_real.float().norm()), float(a.w16.float().norm()), float(a.w16T.float().norm()), [float(v) for v in st["vals"]]))
        if graph and i + 1 == 3 and os.environ.get("DBG_POKE"):
            st = m._dg
            dn = m.discriminators_names[0]
            def rp(tag):
                st["graph"].replay(); torch.cuda.synchronize()
                print("   ", tag, [float(v) for v in st["vals"]])
            rp("replay")
            with torch.no_grad():
                y = m._net(dn)(st["real"]); torch.cuda.synchronize()
            rp("after an eager no-grad D forward on the default stream")
            y = m._net(dn)(st["real"]); torch.cuda.synchronize()
            rp("after an eager D forward with autograd")
            y.float().sum().backward(); torch.cuda.synchronize()
            rp("after its backward")
            junk = [torch.empty(1 << 20, device=dev) for _ in range(64)]; del junk; torch.cuda.synchronize()
            rp("after eager allocations")
            import gc; gc.collect(); torch.cuda.synchronize()
            rp("after gc.collect()")
            t = torch.zeros(1024, device=dev)
            for n in (1000, 3000, 10000):
                for _ in range(n):
                    t.add_(1.0)
                torch.cuda.synchronize()
                rp("after %d tiny eager launches on the default stream" % n)
            m.set_input(batch); m._group_flags(m.group_G); m.forward(); torch.cuda.synchronize()
            rp("after an eager G forward")
            m.compute_G_loss(); torch.cuda.synchronize()
            rp("after compute_G_loss")
            m.loss_G_tot.backward(); torch.cuda.synchronize()
            rp("after the G backward")
