"""joligen_amd: MI355X-native (gfx950) training step for joliGEN's palette_model hot path.

Hand-written HIP kernels (csrc/, C ABI in include/jg355.h) behind a Python host that mirrors the
reference's module / model interface for this path.  GPU only: there is no CPU or eager fallback.
"""
__version__ = "0.1.0"
