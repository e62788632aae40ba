"""Generate tests/golden/sampling_*.pt: DiffusionGenerator.restoration (DDPM sampler) of the UNMODIFIED reference on
CPU with a short test schedule.  TEST INFRASTRUCTURE ONLY.
   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_sampling.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402

import jg_oracle as O  # noqa: E402
import make_golden as MG  # noqa: E402

OUT = MG.OUT
T_TEST = 6
CFGS = {"tiny_eff": MG.TINY["tiny_eff"], "tiny_attn": MG.TINY["tiny_attn"]}


def main():
    os.chdir("/tmp")
    from models import create_model

    for name, c in CFGS.items():
        opt = MG.build_opt(c)
        opt.G_diff_n_timestep_test = T_TEST
        torch.manual_seed(0)
        model = create_model(opt, 0)
        netG = model.netG_A
        ref_sd = netG.state_dict()
        netG.load_state_dict(O.synth_state_dict(ref_sd, seed=0))
        netG.eval()
        B, S = c["B"], c["S"]
        data = MG.synth_batch(B, S, seed=777)
        y_cond, y_0, mask = data["A"], data["B"], data["B_label_mask"]
        g = torch.Generator().manual_seed(31)
        y_t0 = torch.randn(B, 3, S, S, generator=g)
        gen = torch.Generator().manual_seed(32)
        noises = [torch.randn(B, 3, S, S, generator=gen) for _ in range(T_TEST - 1)]   # t = T-1 .. 1
        torch.manual_seed(32)   # the reference draws randn_like(y_t) from the default generator once per step with t > 0
        with torch.no_grad():
            y_out, ret = netG.restoration(y_cond, y_t=y_t0.clone(), y_0=y_0, mask=mask, sample_num=2)
        netG.set_new_sampling_method("ddim")
        with torch.no_grad():
            y_ddim, ret_ddim = netG.restoration(y_cond, y_t=y_t0.clone(), y_0=y_0, mask=mask, sample_num=2, ddim_num_steps=4,
                                                ddim_eta=0.5)
        sched = {k.split(".")[-1]: v.clone() for k, v in netG.state_dict().items() if O._is_buffer(k) and k.endswith("_test")}
        torch.save(dict(cfg=c, T=T_TEST, A=y_cond, B=y_0, mask=mask, y_t0=y_t0, noises=noises, y_out=y_out, ret=ret, y_ddim=y_ddim, ret_ddim=ret_ddim,
                        sched_test=sched, keys=list(ref_sd.keys()), shapes={k: tuple(v.shape) for k, v in ref_sd.items()}),
                   os.path.join(OUT, f"sampling_{name}.pt"))
        print(name, "sampled", float(y_out.abs().mean()), tuple(ret.shape))


if __name__ == "__main__":
    main()
