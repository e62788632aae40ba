"""Step driver mirroring /root/reference/models/base_model.py for the training hot path:
`optimize_parameters` (:1302-1377), `compute_step` (:1250-1282), `ema_step` (:1284-1297),
`set_requires_grad` (:1196-1217), `get_current_losses` (:815-822), `save_networks` (:824-868),
`load_networks` (:957-1103, plain path), `setup` (:694-723), `parallelize`/`single_gpu`
(:725-745), `update_learning_rate` (:770-779).

Differences that are the point of this build:
  * networks keep their parameters in a flat arena; `parallelize()` wraps them in
    FlatDataParallel (one RCCL all-reduce of the flat gradient per step) instead of DDP;
  * the optimizer is the fused AdamW(+EMA+zero_grad) kernel; the EMA copy is a flat buffer and
    `net<name>_ema` is a light module view of it (same state_dict keys);
  * no GradScaler: bf16 activations need none, fp16 uses a static loss scale.
Evaluation / metrics (reference :1637-2287) are out of scope; `export_networks` (:870-938) traces plain-torch mirrors (util/export.py).
"""
from __future__ import annotations

import copy
import os
from collections import OrderedDict
from contextlib import ExitStack

import torch

from .. import parallel
from ..optim import FusedAdamW


class NetworkGroup:
    """util/network_group.py of the reference."""

    def __init__(self, networks_to_optimize, forward_functions, backward_functions, loss_names_list, optimizer,
                 loss_backward, networks_to_ema=()):
        self.networks_to_optimize = networks_to_optimize
        self.forward_functions = forward_functions
        self.backward_functions = backward_functions
        self.loss_names_list = loss_names_list
        self.optimizer = optimizer
        self.loss_backward = loss_backward
        self.networks_to_ema = list(networks_to_ema)


class IterCalculator:
    """util/iter_calculator.py of the reference (loss averaging over train_iter_size)."""

    def __init__(self, loss_names):
        self.loss_names = loss_names
        for n in loss_names:
            setattr(self, "loss_" + n, 0)
            setattr(self, "loss_" + n + "_cur", 0)

    def compute_last_step(self, loss_names):
        for n in loss_names:
            setattr(self, "loss_" + n, getattr(self, "loss_" + n + "_cur"))
            setattr(self, "loss_" + n + "_cur", 0)

    def compute_step(self, loss_name, value):
        setattr(self, "loss_" + loss_name + "_cur", getattr(self, "loss_" + loss_name + "_cur") + value)


def get_scheduler(optimizer, opt):
    """models/modules/utils.py:115-157."""
    from torch.optim import lr_scheduler

    if opt.train_lr_policy == "linear":
        def lambda_rule(epoch):
            return 1.0 - max(0, epoch + opt.train_epoch_count - opt.train_n_epochs) / float(opt.train_n_epochs_decay + 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    if opt.train_lr_policy == "step":
        return lr_scheduler.StepLR(optimizer, step_size=opt.train_lr_decay_iters, gamma=0.1)
    if opt.train_lr_policy == "multistep":
        return lr_scheduler.MultiStepLR(optimizer, milestones=opt.train_lr_steps, gamma=0.1)
    if opt.train_lr_policy == "cosine":
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.train_n_epochs, eta_min=0)
    raise NotImplementedError(f"learning rate policy [{opt.train_lr_policy}] is not implemented")


ACT_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16}


class BaseModel:
    # True: the model's step runs exactly one backward per network and optimizer step, so the data-parallel gradient exchange
    # may start from inside the backward (parallel.EarlyExchange)
    overlap_exchange = False

    def __init__(self, opt, rank):
        self.rank = rank
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        self.with_amp = opt.with_amp
        if not torch.cuda.is_available() or not self.gpu_ids:
            raise RuntimeError("joligen_amd runs on MI355X GPUs only: no CPU fallback (gpu_ids=%r)" % (opt.gpu_ids,))
        self.use_cuda = True
        self.device = torch.device("cuda:{}".format(self.gpu_ids[rank]))
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.loss_names, self.model_names, self.visual_names, self.optimizers = [], [], [], []
        self.niter = 0
        self.objects_to_update = []
        self.act_dtype = ACT_DTYPES[getattr(opt, "jg_act_dtype", "bf16")]
        ls = float(getattr(opt, "jg_loss_scale", 0.0) or 0.0)
        self.loss_scale = ls if ls > 0 else (65536.0 if self.act_dtype == torch.float16 else 1.0)
        self._ema_fused_this_iter = set()

    # ---- optimizer factory (train.py:51-62) ------------------------------------------------
    def make_optimizer(self, net, lr, betas, weight_decay, eps):
        name = self.opt.train_optim
        if name not in ("adam", "adamw", "radam", "lion"):
            raise NotImplementedError(f"train_optim={name!r}: adam / adamw / radam / lion have a fused MI355X kernel (adam8bit needs "
                                      "bitsandbytes' blockwise quantisation maps, unavailable offline)")
        arena = net.jg_finalize(self.device, self.act_dtype)
        if weight_decay != 0.0:
            # the fused kernel walks the whole arena: torch.optim skips parameters without a gradient, so a frozen slice
            # ('freeze' / 'cv_ensemble' names, set_requires_grad) must not silently decay
            frozen = [n for n, prm in net.named_parameters() if "freeze" in n or "cv_ensemble" in n or not prm.requires_grad]
            if frozen:
                raise NotImplementedError(f"weight decay with frozen parameters ({frozen[:3]}...) is not supported by the flat-arena optimizer")
        opt = FusedAdamW(arena, net.parameters(), lr, betas, weight_decay, eps, decoupled=(name == "adamw"), kind=name)
        opt.grad_scale = 1.0 / self.loss_scale
        if self.act_dtype == torch.float16:
            arena.enable_overflow_check()
        return opt

    def poll_overflow(self):
        """fp16 loss-scale maintenance, the host half of torch.cuda.amp.GradScaler (reference base_model.py:89-90,1268-1274): the
        kernels drop a step with non-finite gradients on their own (no host sync per step); every `jg_overflow_poll` OPTIMIZER steps
        (default 50) the dropped-step counters are read back.  A dropped step halves the loss scale (backoff_factor 0.5) and is taken
        out of the optimizers' bias-correction step count; `jg_loss_scale_growth_interval` clean optimizer steps in a row (default
        2000, GradScaler's growth_interval; <= 0 = never) double it again (growth_factor 2); for `jg_overflow_poll` steps after a growth
        event the counters are read on EVERY optimizer step, so an overflowing probe loses one update, not fifty.  The scale only changes on an accumulation boundary
        (niter % train_iter_size == 0): gradients already in the arena were scaled with the old value.
        Approximation, documented: between a dropped step and the poll that sees it (< jg_overflow_poll steps) the bias correction runs
        one step ahead per drop (torch steps `step` only on applied updates)."""
        every = int(getattr(self.opt, "jg_overflow_poll", 50) or 50)
        iter_size = max(1, int(getattr(self.opt, "train_iter_size", 1) or 1))
        if self.act_dtype != torch.float16 or self.niter % iter_size != 0:
            return
        ostep = self.niter // iter_size
        # right after a growth event the doubled scale is a PROBE: poll on every optimizer step of the next window, so that an
        # overflowing probe costs one dropped update (as with GradScaler) instead of up to `every`
        probing = ostep <= getattr(self, "_probe_until", 0)
        if ostep % every != 0 and not probing:
            return
        dropped = 0
        for o in self.optimizers:
            a = getattr(o, "arena", None)
            if a is None or a.overflow is None:
                continue
            n = int(a.overflow[1])
            new = n - getattr(a, "_dropped_seen", 0)
            a._dropped_seen = n
            a.step -= new            # a dropped step must not advance the bias correction
            dropped += new
        g = getattr(self.opt, "jg_loss_scale_growth_interval", 2000)
        growth = 2000 if g is None else int(g)            # <= 0: never grow (a static jg_loss_scale stays static until an overflow)
        since = ostep - getattr(self, "_last_poll_step", 0)
        self._last_poll_step = ostep
        if dropped:
            self._clean_steps = 0
            self._probe_until = 0
            self.loss_scale = max(1.0, self.loss_scale / 2.0)
            print(f"[joligen_amd] {dropped} optimizer step(s) dropped on non-finite fp16 gradients: loss scale -> {self.loss_scale:g}")
        else:
            self._clean_steps = getattr(self, "_clean_steps", 0) + since
            if growth > 0 and self._clean_steps >= growth:
                self._clean_steps = 0
                self.loss_scale = min(self.loss_scale * 2.0, 2.0 ** 24)
                self._probe_until = ostep + every
        for o in self.optimizers:
            o.grad_scale = 1.0 / self.loss_scale

    # ---- setup / parallel --------------------------------------------------------------------
    def setup(self, opt):
        if self.isTrain:
            self.schedulers = [get_scheduler(o, opt) for o in self.optimizers]
        if not self.isTrain or opt.train_continue or getattr(opt, "train_continue_from", ""):
            suffix = "iter_%d" % opt.train_load_iter if opt.train_load_iter > 0 else opt.train_epoch
            self.load_networks(suffix, load_dir=os.path.expanduser(opt.train_continue_from) or None)

    def _net(self, name):
        net = getattr(self, "net" + name)
        return net.module if isinstance(net, parallel.FlatDataParallel) else net

    def single_gpu(self):
        for name in self.model_names:
            self._net(name).jg_finalize(self.device, self.act_dtype)

    def parallelize(self, rank):
        """One process per GPU; parameters broadcast from rank 0 (DDP constructor semantics)."""
        for name in self.model_names:
            net = self._net(name)
            arena = net.jg_finalize(self.device, self.act_dtype)
            self.set_requires_grad(net, True)
            parallel.broadcast_params(arena, 0)
            if self.overlap_exchange and os.environ.get("JG_OVERLAP_EXCHANGE", "1") != "0":
                arena.early_exchange = parallel.EarlyExchange(arena, list(net.named_parameters()))
            setattr(self, "net" + name, parallel.FlatDataParallel(net))

    def eval(self):
        for name in self.model_names:
            getattr(self, "net" + name).eval()

    def update_learning_rate(self):
        for s in self.schedulers:
            s.step()

    # ---- requires_grad toggling (:1196-1217) -------------------------------------------------
    def set_requires_grad(self, nets, requires_grad=False, _frozen_structure=False):
        if not isinstance(nets, list):
            nets = [nets]
        cache = self.__dict__.setdefault("_requires_grad_lists", {})
        for net in nets:
            if net is None:
                continue
            # the (parameter, frozen-by-name) pairs of a network, listed once: the walk over named_parameters() cost ~1 ms of host time
            # per step in the CUT step (eight calls), which is enqueue-bound to within 3 %
            # (only for calls out of the step driver: by then every parameter lives in its arena and the module tree no longer changes --
            # PatchSampleF's MLPs, created on the first forward, exist before the arenas are built)
            ent = cache.get(id(net)) if _frozen_structure else None
            if ent is None or ent[0] is not net:
                ent = cache[id(net)] = (net, [(prm, "freeze" in name or "cv_ensemble" in name) for name, prm in net.named_parameters()])
            for param, frozen in ent[1]:
                param.requires_grad = requires_grad and not frozen

    # ---- step driver (:1302-1377) --------------------------------------------------------------
    def optimize_parameters(self):
        self.niter += 1
        self._ema_fused_this_iter = set()
        with ExitStack() as stack:
            if len(self.opt.gpu_ids) > 1 and self.niter % self.opt.train_iter_size != 0:
                stack.enter_context(parallel.no_sync())
            for group in self.networks_groups:
                for network in self.model_names:
                    self.set_requires_grad(getattr(self, "net" + network), network in group.networks_to_optimize, _frozen_structure=True)
                for forward in group.forward_functions or []:
                    getattr(self, forward)()
                for backward in group.backward_functions:
                    getattr(self, backward)()
                for loss in group.loss_backward:
                    ll = getattr(self, loss) / self.opt.train_iter_size
                    ll.backward()
                loss_names = []
                for temp in group.loss_names_list:
                    loss_names += getattr(self, temp)
                self.compute_step(group.optimizer, loss_names, group)
                if self.opt.train_G_ema:
                    for network in self.model_names:
                        if network in group.networks_to_ema:
                            self.ema_step(network)
            for obj in self.objects_to_update:
                obj.update(self.niter)
        self.poll_overflow()

    def compute_step(self, optimizers_names, loss_names, group=None):
        """:1250-1282.  The EMA update of the group's networks is fused into the optimizer launch
        when the EMA copy already exists (always, except on the very first step)."""
        optimizers = [getattr(self, n) for n in optimizers_names]
        if self.opt.train_iter_size > 1:
            for n in loss_names:
                value = getattr(self, "loss_" + n).clone() / self.opt.train_iter_size
                self.iter_calculator.compute_step(n, value.detach() if torch.is_tensor(value) else value)
        if self.niter % self.opt.train_iter_size == 0:
            for optimizer in optimizers:
                ema_beta = None
                if self.opt.train_G_ema and group is not None and isinstance(optimizer, FusedAdamW):
                    owners = [n for n in group.networks_to_ema if self._net(n).arena is optimizer.arena]
                    if owners and optimizer.arena.ema is not None:
                        ema_beta = self.opt.train_G_ema_beta
                        self._ema_fused_this_iter.update(owners)
                optimizer.step(ema_beta=ema_beta) if isinstance(optimizer, FusedAdamW) else optimizer.step()
                optimizer.zero_grad() if not isinstance(optimizer, FusedAdamW) else None
            if self.opt.train_iter_size > 1:
                self.iter_calculator.compute_last_step(loss_names)
                for n in loss_names:
                    setattr(self, "loss_" + n + "_avg", getattr(self.iter_calculator, "loss_" + n))

    def ema_step(self, network_name):
        """:1284-1297.  First call: the EMA copy is created from the current parameters."""
        net = self._net(network_name)
        arena = net.arena
        if arena.ema is None:
            arena.ema_create()
            setattr(self, "net" + network_name + "_ema", _EmaView(net, arena))
            if network_name in self._ema_fused_this_iter:
                self._ema_fused_this_iter.discard(network_name)
        if network_name in self._ema_fused_this_iter:
            return  # already updated inside the optimizer launch of this iteration
        arena.ema_update(self.opt.train_G_ema_beta)

    def iter_calculator_init(self):
        if self.opt.train_iter_size > 1:
            self.iter_calculator = IterCalculator(self.loss_names)
            for i, cur in enumerate(self.loss_names):
                self.loss_names[i] = cur + "_avg"
                setattr(self, "loss_" + self.loss_names[i], 0)

    def get_current_losses(self):
        out = OrderedDict()
        for name in self.loss_names:
            if isinstance(name, str):
                out[name] = getattr(self, "loss_" + name)
        return out

    def get_current_losses_reduced(self):
        """the logging path of train.py:288-303: the current losses averaged over the ranks (one small all-reduce)"""
        return parallel.reduce_losses(self.get_current_losses())

    def get_current_batch_size(self):
        return self.real_A.shape[0]

    # ---- visuals (:762-764, :782-806): what train.py's display / test loops call ---------------------------------------
    def compute_visuals(self, nb_imgs):
        """base: nothing; the diffusion models run their sampler here (`inference`)"""
        pass

    def get_current_visuals(self, nb_imgs, phase="train", test_name=""):
        visual_ret = []
        for i, group in enumerate(self.visual_names):
            cur_visual = OrderedDict()
            for name in group:
                if phase == "test":
                    name = name + "_test_" + test_name
                if isinstance(name, str) and hasattr(self, name):
                    cur_visual[name] = getattr(self, name)
            visual_ret.append(cur_visual)
            if self.opt.model_type not in ("cut", "cycle_gan") and i == nb_imgs - 1:      # GANs have more outputs in practice
                break
        return visual_ret

    def _publish_visuals(self, nb_imgs, offset=0):
        """`<name><k>` attributes of the first nb_imgs images for every entry of gen_visual_names (palette_model.py:852-862)"""
        for name in self.gen_visual_names:
            whole = getattr(self, name[:-1], None)
            if whole is None:
                continue
            for k in range(min(nb_imgs, self.get_current_batch_size())):
                cur = whole[k:k + 1]
                if "mask" in name:
                    cur = cur.squeeze(0)
                setattr(self, name + str(offset + k), cur)

    # ---- checkpoints (:824-868, :957-1103) -------------------------------------------------------
    def save_networks(self, epoch, blocking=None):
        """`<epoch>_net_<name>.pth` (+ `_ema.pth`) with the reference's keys / layout.
        blocking=False (or opt.jg_async_checkpoint): the step loop only pays for one device -> pinned-host copy of each flat arena,
        enqueued on a side stream; a writer thread waits for the copy, rebuilds the reference-layout state_dict from the host
        snapshot and writes the file (tmp + rename).  `wait_checkpoints()` joins the writers (called by the next save / load)."""
        if blocking is None:
            blocking = not getattr(self.opt, "jg_async_checkpoint", False)
        os.makedirs(self.save_dir, exist_ok=True)
        self.wait_checkpoints()
        for name in self.model_names:
            net = self._net(name)
            path = os.path.join(self.save_dir, "%s_net_%s.pth" % (epoch, name))
            ema = getattr(self, "net" + name + "_ema", None) if self.opt.train_G_ema else None
            ema_path = os.path.join(self.save_dir, "%s_net_%s_ema.pth" % (epoch, name))
            if blocking or getattr(net, "arena", None) is None:
                torch.save(net.state_dict(), path)
                if ema is not None:
                    torch.save(ema.state_dict(), ema_path)
                continue
            self._save_async(net, path, None)
            if ema is not None:
                self._save_async(net, ema_path, net.arena.ema)

    def _save_async(self, net, path, flat):
        import threading

        arena = net.arena
        flat = arena.p if flat is None else flat
        if not hasattr(self, "_ckpt_stream"):
            self._ckpt_stream, self._ckpt_threads = torch.cuda.Stream(device=self.device), []
        keys = getattr(arena, "_sd_keys", None)
        if keys is None:      # key order of the reference layout (parameters and buffers interleaved in module order), computed once
            keys = arena._sd_keys = list(net.state_dict(keep_vars=True).keys())
        shapes = {n: tuple(prm.shape) for n, prm in net.named_parameters()}
        host = torch.empty(flat.shape, dtype=flat.dtype, pin_memory=True)
        side = self._ckpt_stream
        # The snapshot is taken ON THE COMPUTE STREAM (one device-to-device copy of the flat arena, ~0.1 ms for 237 MB): the next
        # optimize_parameters() updates arena.p / arena.ema in place and is ordered behind this clone, so the file can never mix
        # step N and step N + 1.  Only the slow device -> host copy (and the file write) leave the critical path.
        snap = flat.clone()
        bsnap = {k: v.detach().clone() for k, v in net.named_buffers()}
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            host.copy_(snap, non_blocking=True)
            bufs = {k: v.to("cpu", non_blocking=True) for k, v in bsnap.items()}
            done = torch.cuda.Event()
            done.record(side)
        snap.record_stream(side)
        for v in bsnap.values():
            v.record_stream(side)

        def write():
            done.synchronize()
            sd = OrderedDict()
            for k in keys:
                if k in arena.slices:
                    off, _ = arena.slices[k]
                    sd[k] = arena._views(host, off, shapes[k]).clone(memory_format=torch.contiguous_format)
                else:
                    sd[k] = bufs[k].clone()
            tmp = path + ".tmp"
            torch.save(sd, tmp)
            os.replace(tmp, path)

        t = threading.Thread(target=write, daemon=False)
        t.start()
        self._ckpt_threads.append(t)

    def wait_checkpoints(self):
        for t in getattr(self, "_ckpt_threads", []):
            t.join()
        if hasattr(self, "_ckpt_threads"):
            self._ckpt_threads = []

    def export_networks(self, epoch):
        """:870-938.  For every network of `model_names_export` (the GAN generators; palette / cm are skipped by the reference too) the
        `<epoch>_net_<name>.pth` written by save_networks is loaded into a CPU generator and traced to `<epoch>_net_<name>.onnx`
        (always, except for the generators the reference excludes) and `<epoch>_net_<name>.pt` (TorchScript, `train_export_jit`).
        util/export.py holds the plain-torch generators the tracing runs on (the HIP modules are not traceable); an export this
        installation cannot produce (no `onnx` package, no mirror of the generator) is reported and skipped -- the training loop
        (train.py:351-357 calls this after every save) goes on.  Returns the list of files written."""
        if self.opt.model_type in ("palette", "cm", "cm_gan", "sc", "b2b"):
            return []
        from ..util.export import export

        self.wait_checkpoints()          # an asynchronous save of this epoch must be on disk before it is read back
        written = []
        netG = self.opt.G_netG
        for name in getattr(self, "model_names_export", ["G_A"]):
            save_path = os.path.join(self.save_dir, "%s_net_%s.pth" % (epoch, name))
            onnx_ok = (not getattr(self.opt, "train_feat_wavelet", False) and not any(t in netG for t in ("ittr", "hdit", "img2img_turbo"))
                       and netG != "hat" and not (torch.__version__[0] == "2" and "segformer" in netG))
            if onnx_ok:
                written.append(export(self.opt, save_path, save_path.replace(".pth", ".onnx"), getattr(self, "onnx_opset_version", 12), "onnx"))
            if getattr(self.opt, "train_export_jit", False) and not any(t in netG for t in ("uvit", "hdit", "img2img_turbo")):
                written.append(export(self.opt, save_path, save_path.replace(".pth", ".pt"), getattr(self, "onnx_opset_version", 12), "jit"))
        return [w for w in written if w]

    def load_networks(self, epoch, load_dir=None):
        self.wait_checkpoints()
        load_dir = load_dir or self.save_dir
        for name in self.model_names:
            path = os.path.join(load_dir, "%s_net_%s.pth" % (epoch, name))
            state_dict = torch.load(path, map_location="cpu")
            if hasattr(state_dict, "_metadata"):
                del state_dict._metadata
            # :1098-1103: the reference passes `strict=self.opt.model_load_no_strictness` (non-strict unless the flag is set -- the
            # flag's name says the opposite, the call is what reference checkpoints rely on)
            net = self._net(name)
            res = net.load_state_dict(state_dict, strict=bool(getattr(self.opt, "model_load_no_strictness", False)))
            if hasattr(net, "check_loaded_backbone"):      # projected discriminator: a checkpoint without the frozen backbone must not pass silently
                net.check_loaded_backbone(res, path)


class _EmaView:
    """`net<name>_ema` of the reference is a deepcopy of the network; here it is a view of the
    arena's flat EMA buffer that serves `state_dict()` (checkpoint `<suffix>_net_<name>_ema.pth`),
    `parameters()` and `named_parameters()` with the reference's keys."""

    def __init__(self, net, arena):
        self._net, self._arena = net, arena

    def named_parameters(self):
        return list(self._arena.named_views(self._arena.ema).items())

    def parameters(self):
        return [v for _, v in self.named_parameters()]

    def state_dict(self):
        sd = self._net.state_dict()
        for k, v in self._arena.named_views(self._arena.ema).items():
            sd[k] = v.detach().clone(memory_format=torch.contiguous_format)
        return sd

    def eval(self):
        return self
