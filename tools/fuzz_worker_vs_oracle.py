"""Random (solver, wavenumber, frequency band) tasks: the GPU worker (es_worker_run) against the oracle's state machine
(oracle/workers.py) driven by the CPU port's determinant -- root lists and evaluation counts must be identical.
GPU box; not a test (tests/test_workers_gpu.py holds the fixed cases)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eigensolver_amd as E  # noqa: E402
from eigensolver_amd import _lib  # noqa: E402
from oracle import workers as OW  # noqa: E402
from tests.test_workers_gpu import _port_evaluator  # noqa: E402

ctx = _lib.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
SPEC_KEY = {E.CylinderNonUniformFlow: "CF", E.SlabNonUniformFlow: "SF-G", E.SlabUniformFlow: "SF-U"}


def make(i):
    c = i % 5
    if c == 0:
        return E.CylinderNonUniformFlow(U_i0=float(rng.uniform(-0.8, 0.8)), width=float(rng.choice([0.6, 0.9, 1.5, 1e5])), ctx=ctx), "CF"
    if c == 1:
        ph = bool(rng.integers(2))
        return E.CylinderNonUniformDensity(width=float(rng.choice([0.9, 1.25, 3.0])), photospheric=ph, ctx=ctx), ("CD-P" if ph else "CD-C")
    if c == 2:
        v = str(rng.choice(["kink_fast", "kink_slow", "sausage", "sausage_slow"]))
        key = {"kink_fast": "CR-KF", "kink_slow": "CR-KS", "sausage": "CR-SF", "sausage_slow": "CR-SS"}[v]
        return E.CylinderRotationalFlow(v_twist=float(rng.choice([0.05, 0.1, 0.25])), power=float(rng.choice([0.8, 1.0, 1.25])), variant=v, ctx=ctx), key
    if c == 3:
        return E.SlabNonUniformFlow(U_i0=float(rng.uniform(0.1, 0.9)), width=float(rng.choice([1.0, 1.5, 1e5])), ctx=ctx), "SF-G"
    co = bool(rng.integers(2))
    return E.SlabNonUniformDensity(width=float(rng.choice([0.9, 1.5, 3.0])), coronal=co, ctx=ctx), ("SD-C" if co else "SD-P")


n = bad = 0
for i in range(40):
    s, key = make(i)
    for mode in s.modes:
        k = float(rng.uniform(0.2, 3.8))
        bands = s.bands(k, int(rng.integers(10, 40)))
        freq = bands[int(rng.integers(len(bands)))]
        got, nev = s.run_batch(mode, [k], freq[None, :], return_evals=True)
        spec = OW.SPECS[(key, mode)]
        roots, _, req = OW.run_worker(spec, _port_evaluator(s, mode), k, freq)
        n += 1
        if list(got[0]) != list(roots) or int(nev[0]) != len(req):
            bad += 1
            print("MISMATCH", type(s).__name__, key, mode, k, len(got[0]), len(roots), int(nev[0]), len(req))
    s.close()
print("tasks", n, "mismatches", bad)
