"""Time the UNMODIFIED reference's palette_model training step on this container's host cores next to the CPU oracle port
(oracle/jg_oracle.py), same weights, same batch, BASELINE configs[1] shape at batch 1 (256x256, efficient UNet, AdamW + EMA).

TEST INFRASTRUCTURE ONLY (build container; the reference does not exist on the GPU box).
    PYTHONDONTWRITEBYTECODE=1 python oracle/measure_reference_cpu.py  ->  profiles/r02_cpu_reference_vs_port.json
bench.py's `cpu_baseline` leg times the PORT on the GPU box's cores; this file is the once-measured ratio between the port and the
reference itself that the bench line quotes."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402

import jg_oracle as O  # noqa: E402
from make_golden import build_opt, synth_batch  # noqa: E402


def main():
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    c = dict(ngf=64, mults=[1, 2, 4, 8], res_blocks=[2, 2, 2, 2], attn_res=[16], efficient=True, S=256, B=1)
    os.chdir("/tmp")
    from models import create_model

    opt = build_opt(c)
    torch.manual_seed(0)
    model = create_model(opt, 0)
    model.setup(opt)
    model.use_temporal = False
    sd = {k: v.detach().clone() for k, v in model.netG_A.state_dict().items()}
    data = synth_batch(c["B"], c["S"], seed=99)
    n = 6

    def run_ref():
        model.set_input(data)
        model.optimize_parameters()

    run_ref()
    t0 = time.perf_counter()
    for _ in range(n):
        run_ref()
    t_ref = (time.perf_counter() - t0) / n
    tr = O.OraclePaletteTrainer(sd, O.UNetCfg(efficient=True))
    gen = torch.Generator().manual_seed(3)

    def run_port():
        t, u, noise = O.draw_step_randomness(gen, data["B"], 2000)
        tr.optimize_parameters(data["B"], data["A"], data["B_label_mask"], noise, t, u)

    run_port()
    t0 = time.perf_counter()
    for _ in range(n):
        run_port()
    t_port = (time.perf_counter() - t0) / n
    out = dict(config="palette_model DDPM, efficient UNet ngf 64 mults [1,2,4,8], 256x256, batch 1, AdamW + EMA, fp32", cores=cores,
               torch=torch.__version__, steps_timed=n, reference_s_per_step=round(t_ref, 4), port_s_per_step=round(t_port, 4),
               reference_img_per_s=round(c["B"] / t_ref, 4), port_img_per_s=round(c["B"] / t_port, 4),
               port_over_reference=round(t_ref / t_port, 3))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r02_cpu_reference_vs_port.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out)


if __name__ == "__main__":
    main()
