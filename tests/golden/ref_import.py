"""Import harness for the upstream reference (build container only).

The reference lives read-only at /root/reference and never travels to the GPU
box; this module is used ONLY by tests/golden/make_golden.py to generate the
fixtures committed next to it.  Three of the reference's third-party imports
are absent in this image (torchattacks, hostlist, and torch's long-removed
zero_gradients pulled in through attacks/__init__.py), so minimal stand-ins for
those *third-party* modules are registered before import.  No reference source
is copied: the modules are executed from where they lie.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("ADIL_REFERENCE_ROOT", "/root/reference")


class _AttackBase:
    """Surface of torchattacks.attack.Attack that adil.py relies on
    (adil.py:38,68,109; performance.py:159)."""

    def __init__(self, name, model):
        self.attack = name
        self.model = model
        self.device = next(model.parameters()).device
        self._targeted = False

    def __call__(self, *args, **kwargs):
        self.model.eval()
        return self.forward(*args, **kwargs)


def load_reference():
    """Returns (utils, adil, adil_regularized, performance) reference modules."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    ta = types.ModuleType("torchattacks")
    ta_attack = types.ModuleType("torchattacks.attack")
    ta_attack.Attack = ta.Attack = _AttackBase
    ta.attack = ta_attack
    sys.modules["torchattacks"] = ta
    sys.modules["torchattacks.attack"] = ta_attack

    hl = types.ModuleType("hostlist")
    hl.expand_hostlist = lambda s: [s]
    sys.modules["hostlist"] = hl
    for k, v in dict(SLURM_JOB_NODELIST="localhost", SLURM_STEP_GPUS="0", SLURM_NTASKS="1",
                     SLURM_JOB_NUM_NODES="1", SLURM_PROCID="0", SLURM_LOCALID="0").items():
        os.environ.setdefault(k, v)

    # The repo under test ships its own top-level `attacks` / `performance`
    # modules with the same names; make sure the reference's win here.
    for name in [m for m in sys.modules if m == "attacks" or m.startswith("attacks.")
                 or m in ("performance", "env_setting")]:
        del sys.modules[name]
    sys.path.insert(0, REFERENCE_ROOT)
    for name, path in [("attacks", os.path.join(REFERENCE_ROOT, "attacks")),
                       ("attacks.attacks_classes", os.path.join(REFERENCE_ROOT, "attacks", "attacks_classes"))]:
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    import importlib
    U = importlib.import_module("attacks.utils")
    A = importlib.import_module("attacks.attacks_classes.adil")
    R = importlib.import_module("attacks.attacks_classes.adil_regularized")
    PERF = importlib.import_module("performance")
    for mod in (U, A, R, PERF):
        assert os.path.realpath(mod.__file__).startswith(os.path.realpath(REFERENCE_ROOT)), mod.__file__
    return U, A, R, PERF
