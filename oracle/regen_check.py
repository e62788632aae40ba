"""TEST INFRASTRUCTURE (not product code): proof that tests/golden/ is what the committed recipes write from the UNMODIFIED reference.

Regenerating all fixtures takes ~7 minutes of CPU (17 recipes, each builds reference models).  The outcome can only change when one
of its inputs changes, so the full run is keyed on a digest of those inputs:

    every oracle/*.py (recipes, ref_shim.py, jg_oracle.py and the helpers they import), every *.py of /root/reference, and the
    torch / numpy versions.

`python oracle/regen_check.py --write` runs every recipe into a scratch directory, requires every regenerated fixture to equal the
committed one bit for bit, and records {inputs digest, sha256 of every fixture} in tests/golden/REGENERATED.json.
tests/test_oracle_golden.py::test_fixtures_regenerate recomputes the digests: when the inputs digest still matches the recorded one it
only checks the fixture hashes (seconds); when anything changed -- or with JG_FULL_REGEN=1 -- it runs the full regeneration again.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MANIFEST = os.path.join(GOLDEN, "REGENERATED.json")
REFERENCE = "/root/reference"

RECIPES = ["make_golden_resize.py", "make_golden_pix2pix.py", "make_golden_accum.py", "make_golden_cutaccum.py", "make_golden_minsnr.py",
           "make_golden_heads16.py", "make_golden.py", "make_golden_cm.py", "make_golden_cond.py", "make_golden_cut.py",
           "make_golden_cutstep.py", "make_golden_palette_loss.py", "make_golden_projd.py", "make_golden_projd_vit.py", "make_golden_resattn.py",
           "make_golden_sampling.py", "make_golden_segformer.py"]


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def inputs_digest():
    """sha256 over (relative path, content hash) of every regeneration input, plus the library versions the arithmetic depends on"""
    import numpy
    import torch

    h = hashlib.sha256()
    h.update(("torch %s numpy %s\n" % (torch.__version__, numpy.__version__)).encode())
    odir = os.path.join(ROOT, "oracle")
    for f in sorted(os.listdir(odir)):
        if f.endswith(".py") and f != os.path.basename(__file__):
            h.update(("oracle/%s %s\n" % (f, _sha(os.path.join(odir, f)))).encode())
    for d, dirs, files in os.walk(REFERENCE):
        dirs[:] = sorted(x for x in dirs if not x.startswith("."))
        for f in sorted(files):
            if f.endswith(".py"):
                p = os.path.join(d, f)
                h.update(("%s %s\n" % (os.path.relpath(p, REFERENCE), _sha(p))).encode())
    return h.hexdigest()


def fixture_digests(directory=GOLDEN):
    return {f: _sha(os.path.join(directory, f)) for f in sorted(os.listdir(directory)) if f.endswith(".pt")}


def regenerate(out_dir):
    """run every recipe with JG_GOLDEN_OUT=out_dir; returns [(script, returncode, output tail)]"""
    env = dict(os.environ, JG_GOLDEN_OUT=str(out_dir), PYTHONDONTWRITEBYTECODE="1")

    def run(script):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", script)], env=env, cwd="/tmp", capture_output=True, text=True)
        return script, r.returncode, (r.stdout + r.stderr)[-1500:]

    # torch-CPU reductions are bit-reproducible only at the thread count the fixtures were written with (the default), so the recipes
    # are not pinned to fewer threads; three at a time
    with ThreadPoolExecutor(max_workers=3) as ex:
        return list(ex.map(run, RECIPES))


def same(a, b, path):
    """recursive equality of two loaded fixtures: tensors bit-exact, containers element-wise, floats exactly"""
    import torch

    if isinstance(a, torch.Tensor):
        assert isinstance(b, torch.Tensor) and a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), path
    elif isinstance(a, dict):
        assert isinstance(b, dict) and list(a.keys()) == list(b.keys()), (path, list(a.keys()), list(b.keys()) if isinstance(b, dict) else b)
        for k in a:
            same(a[k], b[k], f"{path}[{k!r}]")
    elif isinstance(a, (list, tuple)):
        assert type(a) is type(b) and len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            same(x, y, f"{path}[{i}]")
    else:
        assert a == b or (a != a and b != b), (path, a, b)


def full_check(scratch):
    """regenerate into `scratch` and compare with the committed fixtures; raises AssertionError on any difference"""
    import torch

    for script, rc, tail in regenerate(scratch):
        assert rc == 0, (script, tail)
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".pt"))
    made = sorted(f for f in os.listdir(scratch) if f.endswith(".pt"))
    assert made == committed, (set(committed) ^ set(made))
    for f in committed:
        same(torch.load(os.path.join(GOLDEN, f), weights_only=False), torch.load(os.path.join(scratch, f), weights_only=False), f)


def main():
    if "--write" not in sys.argv:
        print(__doc__)
        return
    with tempfile.TemporaryDirectory() as scratch:
        full_check(scratch)
    with open(MANIFEST, "w") as f:
        json.dump({"inputs_sha256": inputs_digest(), "fixtures": fixture_digests()}, f, indent=1, sort_keys=True)
        f.write("\n")
    print("all %d fixtures regenerate bit for bit; wrote %s" % (len(fixture_digests()), MANIFEST))


if __name__ == "__main__":
    main()
