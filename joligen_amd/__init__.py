"""joligen_amd: MI355X-native (gfx950) training step for joliGEN's palette_model hot path.

Hand-written HIP kernels (csrc/, C ABI in include/jg355.h) behind a Python host that mirrors the
reference's module / model interface for this path.  GPU only: there is no CPU or eager fallback.
"""
import os as _os

# ROCm 7.2: hipGraph replays are corrupted by eager launches in between unless the runtime's AQL-packet capture is off (models/cut_model.py,
# profiles/r04_graph_replay_probe.txt).  Read by the HIP runtime when it initialises, so this only helps when the package is imported
# before the first HIP call; the graph path checks itself with a canary either way.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

__version__ = "0.1.0"
