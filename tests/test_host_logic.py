"""CPU: host-side logic that needs no GPU -- the C ABI library loads and exports every symbol
include/jg355.h declares, module trees reproduce the reference's state_dict keys/shapes, the flat
parameter arena's views/layout, options flattening, checkpoint hooks."""
import os
import re
import subprocess

import pytest
import torch

import jg_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from joligen_amd import _lib

    path = _lib.build()
    header = open(os.path.join(ROOT, "include", "jg355.h")).read()
    declared = set(re.findall(r"\b(jg_[a-z0-9_]+)\s*\(", header))
    declared -= {"jg_stream_t"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    nm = subprocess.run(["nm", "-D", path], check=True, capture_output=True, text=True).stdout
    for name in declared:
        assert f" T {name}" in nm, name
    L = _lib.lib()  # dlopen + argtypes; no compute call (no GPU here)
    assert L.jg_version() >= 100
    assert b"bad argument" in L.jg_strerror(-1)


def test_struct_layout_matches_header(tmp_path):
    """ctypes mirrors of jg_conv_args / jg_wgrad_args: field order as in the header, and sizes / offsets equal
    to what a C compiler makes of include/jg355.h (gcc)."""
    import ctypes as C
    import subprocess

    from joligen_amd._lib import ConvArgs, WgradArgs

    hdr = open(os.path.join(ROOT, "include", "jg355.h")).read()
    nocomment = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    body = nocomment[nocomment.index("typedef struct {"):nocomment.index("} jg_conv_args;")]
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*[;,]", body)
    assert names == [f[0] for f in ConvArgs._fields_], names
    body = nocomment[nocomment.index("} jg_conv_args;"):]
    body = body[body.index("typedef struct {"):body.index("} jg_wgrad_args;")]
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*[;,]", body)
    assert names == [f[0] for f in WgradArgs._fields_], names
    # the compiler's view of the two structs
    probe = ["stats", "ldstats", "gn_x", "gn_act", "ldx", "sxb", "alpha"]
    wprobe = ["lddy", "sdyb", "alpha", "dbias_scale"]
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "jg355.h"\nint main(void) {\n'
        '  printf("%zu %zu", sizeof(jg_conv_args), sizeof(jg_wgrad_args));\n'
        + "".join(f'  printf(" %zu", offsetof(jg_conv_args, {n}));\n' for n in probe)
        + "".join(f'  printf(" %zu", offsetof(jg_wgrad_args, {n}));\n' for n in wprobe)
        + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    vals = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    mine = [C.sizeof(ConvArgs), C.sizeof(WgradArgs)] + [getattr(ConvArgs, n).offset for n in probe] \
        + [getattr(WgradArgs, n).offset for n in wprobe]
    assert vals == mine, (vals, mine)


@pytest.mark.parametrize("name", ["tiny_eff", "tiny_noeff", "tiny_attn"])
def test_module_tree_has_reference_state_dict(golden_dir, name):
    from joligen_amd.models.palette_model import define_G
    from joligen_amd.options import opt_from_json

    g = torch.load(os.path.join(golden_dir, f"palette_step_{name}.pt"), weights_only=False)
    c = g["cfg"]
    opt = opt_from_json({}, dict(G_ngf=c["ngf"], G_unet_mha_channel_mults=c["mults"], G_unet_mha_res_blocks=c["res_blocks"],
                                 G_unet_mha_attn_res=c["attn_res"], G_unet_mha_vit_efficient=c["efficient"],
                                 data_crop_size=c["S"]))
    net = define_G(**vars(opt))
    sd = net.state_dict()
    assert list(sd.keys()) == g["keys"]
    for k, v in sd.items():
        assert tuple(v.shape) == g["shapes"][k], k
    sched = torch.load(os.path.join(golden_dir, "schedule.pt"), weights_only=False)
    for k, v in sd.items():
        if O._is_buffer(k):
            assert torch.equal(v, sched[k.split(".")[-1]]), k
    # zero-initialised layers of the reference (zero_module) are zero here too
    for k, v in sd.items():
        if k.endswith("out_layers.3.weight") or k.endswith("proj_out.weight") or k.endswith("out.2.weight"):
            assert float(v.abs().max()) == 0.0, k


def test_full_size_network_matches_survey_counts():
    from joligen_amd.models.palette_model import define_G
    from joligen_amd.options import opt_from_json

    net = define_G(**vars(opt_from_json({}, dict(data_crop_size=256))))
    assert sum(p.numel() for p in net.parameters()) == 59346627 + 2112
    assert len(net.state_dict()) == 338 and len(list(net.parameters())) == 324


def test_options_flatten_reference_json_layout():
    from joligen_amd.options import opt_from_json

    cfg = {"G": {"netG": "unet_mha", "ngf": 32, "unet_mha_vit_efficient": True},
           "alg": {"diffusion": {"lambda_G": 2.0, "task": "inpainting"}, "palette": {"loss": "MSE"}},
           "train": {"batch_size": 4, "iter_size": 16, "G_ema": True, "optim": "adamw"},
           "data": {"crop_size": 128, "online_creation": {"crop_size_A": 128}}, "gpu_ids": "0,1", "model_type": "palette"}
    opt = opt_from_json(cfg, {"train_iter_size": 1})
    assert opt.G_ngf == 32 and opt.G_unet_mha_vit_efficient is True and opt.alg_diffusion_lambda_G == 2.0
    assert opt.train_batch_size == 4 and opt.train_iter_size == 1 and opt.train_G_ema is True
    assert opt.data_online_creation_crop_size_A == 128 and opt.gpu_ids == [0, 1] and opt.isTrain
    assert opt_from_json({}, {"gpu_ids": "-1"}).gpu_ids == []


def test_train_continue_from_option_contract(tmp_path):
    """The host half of /root/reference/tests/test_train_continue_from.py: `train_continue` and `train_continue_from` exclude each other
    (options/train_options.py:694-697), the load suffix is `iter_<n>` when `train_load_iter > 0` (train.py:92-95), and
    `finetune_source.json` records the source checkpoints with the reference's keys (train.py:98-120).  The device half (setup() loads
    `<source>/<suffix>_net_<name>.pth`, save_dir stays the target run's) is tests/test_gpu_1_model.py::test_train_continue_from_loads_source_run."""
    import json

    import pytest

    from joligen_amd.options import get_train_load_suffix, opt_from_json, save_finetune_source_metadata

    with pytest.raises(ValueError, match="mutually exclusive"):
        opt_from_json({}, {"train_continue": True, "train_continue_from": "/tmp/source_run"})
    opt = opt_from_json({}, {"train_continue_from": "source_run", "train_load_iter": 123, "checkpoints_dir": str(tmp_path), "name": "target_run"})
    assert get_train_load_suffix(opt) == "iter_123"
    assert get_train_load_suffix(opt_from_json({}, {})) == "latest"
    path = save_finetune_source_metadata(opt, "python train.py ...", ["G_A"])
    meta = json.loads(open(path).read())
    assert path == str(tmp_path / "target_run" / "finetune_source.json")
    assert meta["train_continue_from"] == "source_run" and meta["load_suffix"] == "iter_123"
    assert meta["checkpoint_files"] == ["source_run/iter_123_net_G_A.pth"] and meta["command_line"] == "python train.py ..."
    assert save_finetune_source_metadata(opt_from_json({}, {}), "x", ["G_A"]) is None


def test_param_arena_layout_on_cpu():
    """The arena itself is plain tensor-view bookkeeping and can be exercised on the CPU device
    (only refresh()/adamw_step() launch kernels)."""
    from joligen_amd.arena import ParamArena
    from joligen_amd.models.palette_model import define_G
    from joligen_amd.options import opt_from_json

    opt = opt_from_json({}, dict(G_ngf=32, G_unet_mha_channel_mults=[1, 2], G_unet_mha_res_blocks=[1, 1], data_crop_size=16))
    net = define_G(**vars(opt))
    ref = {k: v.clone() for k, v in net.state_dict().items()}
    arena = ParamArena(net, "cpu", torch.bfloat16)
    # values preserved, logical shapes unchanged, conv weights physically KRSC
    for k, p in net.named_parameters():
        assert torch.equal(p.detach(), ref[k]), k
        assert p.grad is not None and p.grad.shape == p.shape
        off, n = arena.slices[k]
        assert p.data_ptr() == arena.p.data_ptr() + 4 * off
        assert p.grad.data_ptr() == arena.g.data_ptr() + 4 * off
        if p.dim() == 4:
            O_, I, R, S = p.shape
            assert p.stride() == (R * S * I, 1, S * I, I), (k, p.stride())
            flat = arena.p[off:off + n].view(O_, R, S, I)
            assert torch.equal(flat, ref[k].permute(0, 2, 3, 1))
    # the stacked embedding projection is one contiguous [sum(2C), emb] block in module order
    unet = net.denoise_fn.model
    Wall = arena.group_view("emb_layers.1.weight").view(unet.emb_total, unet.cond_embed_dim)
    off = 0
    for m in unet.modules():
        if hasattr(m, "emb_slice") and m.emb_slice is not None:
            assert m.emb_slice[0] == off
            assert torch.equal(Wall[off:off + m.emb_slice[1]], m.emb_layers[1].weight.detach())
            off += m.emb_slice[1]
    assert off == unet.emb_total
    # state_dict() hands out plain contiguous tensors in the reference layout
    sd = net.state_dict()
    for k, v in sd.items():
        assert v.is_contiguous() and torch.equal(v, ref[k]), k
        assert v.untyped_storage().nbytes() <= max(v.numel(), 1) * v.element_size() + 64
    # load_state_dict writes through the views and marks the 16-bit copies stale
    arena.dirty = False
    net.load_state_dict(O.synth_state_dict(sd, seed=3))
    assert arena.dirty
    k0 = "denoise_fn.model.input_blocks.1.0.in_layers.2.weight"
    off, n = arena.slices[k0]
    assert torch.equal(arena.p[off:off + n].view(32, 3, 3, 32), O.synth_state_dict(sd, seed=3)[k0].permute(0, 2, 3, 1))
    # working-copy descriptors: padded shapes and offsets are consistent
    desc = arena.desc.tolist()
    tot = 0
    for src, dst, dstT, cout, rs, cin, coutp, cinp in desc:
        assert dst == tot and coutp % 8 == 0 and cinp % 8 == 0 and coutp >= cout and cinp >= cin
        tot += coutp * rs * cinp
    assert tot <= arena.w16.numel()
    stem = net.denoise_fn.model.input_blocks[0][0].meta
    assert (stem.Cin, stem.Cin_real, stem.Cout) == (8, 6, 32)
    head = net.denoise_fn.model.out[2].meta
    assert (head.Cout, head.Cout_real) == (8, 3) and head.bias_pad is not None


def test_ops_refuse_cpu_tensors():
    """No CPU / eager fallback: the product path fails loudly without a GPU."""
    from joligen_amd import ops
    from joligen_amd.models import create_model
    from joligen_amd.options import opt_from_json

    with pytest.raises(RuntimeError, match="GPU only|no CPU"):
        ops.group_norm(torch.zeros(1, 4, 4, 8, dtype=torch.bfloat16), 1)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            create_model(opt_from_json({}, {"gpu_ids": "0"}), 0)


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    paths = [os.path.join(dirpath, f) for top in ("joligen_amd", "tools") for dirpath, _, files in os.walk(os.path.join(ROOT, top))
             for f in files if f.endswith(".py")]
    assert len(paths) > 20
    for path in paths:
        src = open(path).read()
        assert "jg_oracle" not in src and "ref_shim" not in src and "/root/reference" not in src.replace(
            "/root/reference/models", "").replace("/root/reference/", "REF/"), path


def test_hot_kernels_do_not_spill(tmp_path):
    """The halo-resident convolution / weight-gradient kernels sit at the VGPR limit of their occupancy target: any scratch spill
    costs 2-3x (a reflect-padding edit once pushed the 8-row weight-gradient configuration into 95 spilled registers).  Compile
    the two translation units with -save-temps and require zero spills and zero private segment for every kernel in them."""
    import re
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(ROOT, "joligen_amd", "csrc")
    for src in ("conv_halo.hip", "wgrad_halo.hip", "conv_p64.hip"):
        out = tmp_path / src.replace(".hip", "")
        out.mkdir()
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-c", os.path.join(csrc, src), "-o",
                            str(out / "o.o"), "-save-temps=obj"], capture_output=True, text=True, cwd=str(out))
        assert r.returncode == 0, r.stderr[-2000:]
        asm = [f for f in os.listdir(out) if f.endswith("gfx950.s")]
        assert asm, os.listdir(out)
        text = open(out / asm[0]).read()
        names = re.findall(r"\.name:\s+(\S+)", text)
        spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)]
        priv = [int(v) for v in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)]
        assert names and len(spills) >= len([n for n in names if "halo_kernel" in n or "p64_kernel" in n])
        if src == "conv_p64.hip":
            # the persistent kernel sits at the 256-register limit of two waves per SIMD: a handful of loop-invariant scalars are
            # allowed to live in scratch (reloaded twice per tile); anything more means a whole array went there
            assert max(spills) <= 8, list(zip(names, spills, priv))
            continue
        assert max(spills) == 0 and max(priv) == 0, list(zip(names, spills, priv))
        # occupancy budgets (512 VGPRs per SIMD lane): LDS (66 KB per workgroup) holds the 64-wide tiles at two workgroups per CU =
        # two waves per SIMD, the 128x512 8-wave tile at one workgroup: both stay clear of the 256-register line with room to spare
        vg = [int(v) for v in re.findall(r"\.vgpr_count:\s+(\d+)", text)]
        for nm, v in zip(names, vg):
            if "conv3x3_halo_kernel" in nm and "Li64ELi256E" in nm:
                assert v <= 168, (nm, v)
            if "conv3x3_halo_kernel" in nm and "Li128ELi512E" in nm:
                assert v <= 168, (nm, v)


def test_image_pool_matches_reference_semantics():
    """util/image_pool.py:19-56 behaviour of joligen_amd.util.image_pool.ImagePool (host logic, CPU tensors): fill phase, 50 % swap
    with a random stored image, draw order uniform -> randint, detached returns; against the oracle restatement with the same RNG."""
    import random

    import torch

    import jg_oracle as O
    from joligen_amd.util.image_pool import ImagePool

    for size in (0, 1, 3):
        ref, mine = O.OracleImagePool(size, random.Random(7)), ImagePool(size, random.Random(7))
        for it in range(15):
            x = (torch.full((2, 4, 4, 8), float(it)) + torch.arange(2).view(2, 1, 1, 1) * 0.5).requires_grad_(size > 0)
            a, b = ref.query(x), mine.query(x)
            assert torch.equal(a, b)
            assert size == 0 or not b.requires_grad
        assert len(mine) == min(size, 30) if size else True


def test_cut_options_and_forced_segformer_settings():
    """CUT defaults of cut_model.py (:39-137) and the settings the reference enforces for a SegFormer generator (:205-210); the model
    class itself refuses to build without a GPU (no CPU fallback in the product path)."""
    import pytest
    import torch

    from joligen_amd.models.cut_model import CUT_DEFAULTS
    from joligen_amd.options import opt_from_json

    assert CUT_DEFAULTS["alg_cut_nce_T"] == 0.07 and CUT_DEFAULTS["alg_cut_num_patches"] == 256
    assert CUT_DEFAULTS["alg_cut_nce_layers"] == "0,4,8,12,16" and CUT_DEFAULTS["alg_cut_nce_loss"] == "monce"
    opt = opt_from_json({"model_type": "cut", "G": {"netG": "segformer_attn_conv"}, "alg": {"cut": {"nce_T": 0.07}}}, {"gpu_ids": "0"})
    assert opt.alg_cut_nce_T == 0.07 and opt.G_netG == "segformer_attn_conv"
    if not torch.cuda.is_available():
        from joligen_amd.models import create_model

        with pytest.raises(RuntimeError, match="MI355X"):
            create_model(opt, 0)


def test_early_exchange_parameter_coverage_of_the_fused_backward():
    """Data parallel (parallel.EarlyExchange): walking the UNet the way unet_exec's forward builds its tape, the records' own
    parameters (`_own_params`) cover every UNet parameter EXACTLY once except the ResBlock embedding projections (their gradient
    comes from the stacked linear behind the node); with the conditioning MLP those are the parameters that go out with the
    optimizer step.  Checked on the CPU-built module tree + arena of an efficient and a non-efficient UNet, then driven through
    EarlyExchange's bookkeeping (tail-first readiness, nothing reported twice, every chunk accounted for)."""
    import torch
    import torch.nn as nn

    from joligen_amd import parallel
    from joligen_amd.arena import ParamArena
    from joligen_amd.models.palette_model import define_G
    from joligen_amd.modules import unet_exec as ue
    from joligen_amd.modules.unet_generator_attn import AttentionBlock, ResBlock
    from joligen_amd.options import opt_from_json

    for efficient, mults, nres in ((True, [1, 2], [1, 1]), (False, [1, 2, 2], [2, 1, 1])):
        opt = opt_from_json({}, dict(G_ngf=32, G_unet_mha_channel_mults=mults, G_unet_mha_res_blocks=nres, G_unet_mha_attn_res=[16],
                                     G_unet_mha_vit_efficient=efficient, data_crop_size=32, gpu_ids="0"))
        net = define_G(**vars(opt))
        arena = ParamArena(net, torch.device("cpu"), torch.bfloat16)
        u = net.denoise_fn.model
        recs = [dict(kind="stem", m=list(u.input_blocks[0])[0].meta)]
        blocks = [list(b) for b in list(u.input_blocks)[1:]] + [list(u.middle_block)] + [list(b) for b in u.output_blocks]
        for layers in blocks:
            for layer in layers:
                if isinstance(layer, ResBlock):
                    recs.append(dict(kind="res", rb=layer, identity=isinstance(layer.skip_connection, nn.Identity)))
                else:
                    assert isinstance(layer, AttentionBlock), type(layer)
                    recs.append(dict(kind="attn", blk=layer))
        recs.append(dict(kind="head", gn=u.out[0].norm, m=u.out[2].meta))
        name_of = {id(p): n for n, p in net.named_parameters()}
        reported = [id(p) for rec in recs for p in ue._own_params(rec)]
        assert len(reported) == len(set(reported))                                    # nothing twice
        assert all(i in name_of for i in reported)
        left = sorted(n for i, n in name_of.items() if i not in set(reported))
        assert left and all(("emb_layers" in n) or ("cond_embed" in n) for n in left), left   # only the embedding path is outside the node
        # the bookkeeping: report in backward (reverse tape) order
        ex = parallel.EarlyExchange.__new__(parallel.EarlyExchange)
        launched = []
        ex._launch = lambda c, ex=ex: (ex.sent.__setitem__(c, True), launched.append(c))
        parallel.EarlyExchange.__init__(ex, arena, list(net.named_parameters()), n_chunks=8)
        for rec in reversed(recs):
            for p in ue._own_params(rec):
                ex.mark(p)
        early = list(launched)
        assert len(early) >= len(ex.bounds) - 3, (early, len(ex.bounds))              # the priority (embedding) group sits at the head
        assert early == sorted(early, reverse=True) or len(set(early)) == len(early)  # tail-first, each chunk once
        assert all(ex.left[c] == 0 for c in early) and all(ex.left[c] > 0 or ex.total[c] == 0 for c in range(len(ex.bounds)) if c not in early)
        for k in list(parallel._EARLY):                                                # keep the module-level registry clean for other tests
            if parallel._EARLY[k] is ex:
                del parallel._EARLY[k]


def test_fp16_loss_scale_policy_polls_every_step_after_growth():
    """BaseModel.poll_overflow (host half of GradScaler, reference base_model.py:89-90,1268-1274): the dropped-step counter is
    read every `jg_overflow_poll` optimizer steps, and on EVERY step of the window that follows a growth event (an overflowing probe costs
    one update, not fifty); `jg_loss_scale_growth_interval <= 0` never grows; the step count of a dropped step is given back."""
    from types import SimpleNamespace

    from joligen_amd.models.base_model import BaseModel

    class Arena:
        def __init__(self):
            self.overflow, self.step = torch.zeros(2, dtype=torch.int64), 0

    def model(growth, every=10):
        m = object.__new__(BaseModel)
        m.opt = SimpleNamespace(jg_overflow_poll=every, train_iter_size=1, jg_loss_scale_growth_interval=growth)
        m.act_dtype, m.niter, m.loss_scale = torch.float16, 0, 1024.0
        a = Arena()
        m.optimizers = [SimpleNamespace(arena=a, grad_scale=1.0 / 1024.0)]
        return m, a

    m, a = model(growth=20)
    for _ in range(20):                       # 20 clean steps: polled at 10 and 20, grows at 20
        m.niter += 1; a.step += 1
        m.poll_overflow()
    assert m.loss_scale == 2048.0 and m.optimizers[0].grad_scale == 1.0 / 2048.0
    m.niter += 1; a.step += 1
    a.overflow[1] += 1                        # the very next step overflows on the doubled scale ...
    m.poll_overflow()                         # ... and is seen at once (step 21 is not a multiple of 10)
    assert m.loss_scale == 1024.0 and a.step == 20, (m.loss_scale, a.step)
    for _ in range(8):                        # back to the sparse schedule: an overflow at step 22 waits for the poll at step 30
        m.niter += 1; a.step += 1
        if m.niter == 22:
            a.overflow[1] += 1
        m.poll_overflow()
        assert m.loss_scale == 1024.0
    m.niter += 1; a.step += 1
    m.poll_overflow()
    assert m.niter == 30 and m.loss_scale == 512.0
    m, a = model(growth=0)                    # static scale: never grows
    for _ in range(200):
        m.niter += 1; a.step += 1
        m.poll_overflow()
    assert m.loss_scale == 1024.0


def test_hip_graphs_safe_flag():
    """joligen_amd/__init__.py: the package switches ROCm's graph AQL-packet capture off when it still can (variable unset, HIP runtime not yet
    initialised) and records whether replays can be trusted; an explicit user setting is respected either way."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os, joligen_amd; print(joligen_amd.HIP_GRAPHS_SAFE, os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'))"

    def run(value):
        env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
        if value is not None:
            env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = value
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout.split()

    assert run(None) == ["True", "0"]
    assert run("0") == ["True", "0"]
    assert run("1") == ["False", "1"]
