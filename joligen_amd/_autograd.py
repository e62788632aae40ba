"""autograd.Function base of every op wrapper in the package."""
import torch

_Function = torch.autograd.Function


class JGFunction(_Function):
    """`torch.autograd.Function` whose `apply()` goes straight to the C++ entry.  The stock classmethod first probes `setup_context`, asks
    whether functorch transforms are active and unwraps dead functorch wrappers: ~1.7 us per call (measured, CPU), 1100 calls per CUT
    configs[2] step, whose wall time is the host's enqueue time to within 3 % (tools/cut_host_probe.py).  None of the ops defines
    `setup_context` or is meant to run under vmap / jvp transforms; keyword arguments are not accepted (the C++ entry takes positionals)."""

    @classmethod
    def apply(cls, *args):
        return super(_Function, cls).apply(*args)
