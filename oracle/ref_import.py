"""TEST INFRASTRUCTURE ONLY — never imported by the product path.

Imports the read-only reference tree (/root/reference) under the stub packages in oracle/stubs so
that its agents / plugins can be driven directly on CPU to (a) validate the restatement in
oracle/ocl_oracle.py and (b) generate the golden fixtures under tests/golden/.  /root/reference does
not exist on the GPU box, so nothing that runs there may import this module; `available()` is the
guard.

Recipe follows SURVEY.md §8(c): stubs first on sys.path, PYTHONDONTWRITEBYTECODE so the reference
mount is not littered, GPU hidden is unnecessary here (no GPU in the build container).
"""
import contextlib
import io
import os
import sys
from types import SimpleNamespace

REF_ROOT = os.environ.get("OCL_REFERENCE_ROOT", "/root/reference")
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "utils", "name_match.py"))


def activate():
    """Put stubs + reference on sys.path (idempotent). Returns the reference `utils.name_match`."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    for p in (REF_ROOT, _STUBS):
        if p in sys.path:
            sys.path.remove(p)
    sys.path[:0] = [_STUBS, REF_ROOT]
    import utils.name_match as nm  # noqa: E402  (reference module)
    return nm


def default_params(**over):
    """The SimpleNamespace the reference's agents read."""
    p = dict(agent="ER", retrieve="random", update="random", data="cifar100", mem_size=1000,
             eps_mem_batch=10, cuda=False, epoch=1, batch=10, test_batch=128, verbose=False,
             optimizer="SGD", learning_rate=0.1, weight_decay=0, mem_iters=1, subsample=50, k=3,
             aser_type="asvm", n_smp_cls=1.5, num_tasks=10, temp=0.07, head="mlp",
             buffer_tracker=False, warmup=4, error_analysis=False, seed=0, fix_order=False,
             trick={"labels_trick": False, "kd_trick": False, "separated_softmax": False,
                    "review_trick": False, "ncm_trick": False, "kd_trick_star": False})
    p.update(over)
    return SimpleNamespace(**p)


@contextlib.contextmanager
def quiet():
    """The reference prints a lot (buffer size, accuracies)."""
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield buf


def build_agent(params):
    """model/opt/agent exactly as experiment/run.py:38-41 does."""
    nm = activate()
    from utils.setup_elements import setup_architecture, setup_opt
    with quiet():
        model = setup_architecture(params)
        opt = setup_opt(params.optimizer, model, params.learning_rate, params.weight_decay)
        agent = nm.agents[params.agent](model, opt, params)
    return model, opt, agent
