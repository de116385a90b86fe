"""BASELINE.json configs[0]: "Slab / non-uniform density, kink mode, 64 k-points, fp64 NumPy on CPU (plumbing, no
GPU)": the whole CPU oracle chain end to end -- equilibrium, determinant (C port), the SD-P kink worker state
machine over the reference's band, and a DOP853 cross-check of what it reports."""
import numpy as np

from eigensolver_amd import equilibrium as q
from oracle import workers as OW
from tests import cases


def test_slab_density_kink_64_k_points():
    eq = q.SlabDensity(width=1.5, n_nodes=1001)
    port = cases.port_problem(eq, "kink")
    truth = cases.truth_problem(eq, "kink")
    import ctypes as C
    from oracle.port import lib
    L = lib()

    def evaluate(k, w):
        d, rel = C.c_double(), C.c_double()
        st = L.port_eval(port.h, k, w, C.byref(d), C.byref(rel))
        if st == 1:
            return OW.ST_LEAKY, float("nan"), float("nan"), float("nan")
        norm = abs(d.value) * 100.0 / rel.value if rel.value == rel.value and rel.value != 0 else float("nan")
        return st, d.value, norm, 0.0

    spec = OW.SPECS[("SD-P", "kink")]
    ks = np.linspace(0.2, 3.5, 64)
    n_roots, checked = 0, 0
    for k in ks:
        # fast-kink band above the interior sound speed, below the exterior one (where the stored width15 kink roots
        # lie: W in [0.89, 1.15]); end points kept off the singular speeds
        freq = np.linspace(eq.c_i0 * k * 1.001, 1.25 * k, 24)
        roots, kk, req = OW.run_worker(spec, evaluate, k, freq)
        assert len(roots) == len(kk)
        n_roots += len(roots)
        for w in roots[:1]:
            d, a, b, st = truth.mismatch(k, w)
            if st == 0:
                checked += 1
                assert abs(d) * 100 / max(abs(a), abs(b)) < spec.tol * 1.01      # accepted under the worker's own rule
    assert n_roots >= 32 and checked >= 16
