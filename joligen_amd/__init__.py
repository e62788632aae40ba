"""joligen_amd: MI355X-native (gfx950) training step for joliGEN's palette_model hot path.

Hand-written HIP kernels (csrc/, C ABI in include/jg355.h) behind a Python host that mirrors the
reference's module / model interface for this path.  GPU only: there is no CPU or eager fallback.
"""
import os as _os
import sys as _sys

# ROCm 7.2: hipGraph replays are corrupted by eager launches in between unless the runtime's AQL-packet capture is off (models/cut_model.py,
# profiles/r04_graph_replay_probe.txt).  The HIP runtime reads the variable when it INITIALISES (first HIP call, not `import torch`:
# checked), so setting it here works whenever the package is imported before the first GPU call; HIP_GRAPHS_SAFE says whether it did.
_v = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")
if _v is None:
    _t = _sys.modules.get("torch")
    if _t is not None and _t.cuda.is_initialized():
        HIP_GRAPHS_SAFE = False          # too late for this process: the graph paths stay off
    else:
        _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
        HIP_GRAPHS_SAFE = True
else:
    HIP_GRAPHS_SAFE = _v == "0"

__version__ = "0.1.0"
