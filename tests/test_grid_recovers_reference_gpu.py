"""The grid path (es_shoot_eval_grid + es_shoot_find_roots: the bench's path) against the roots the REFERENCE found in
its own driver sweeps and traced worker calls (tests/golden/roots_*.json, trace_*.json, executed in the build container), for the calls whose reference run
was clean (classified "identical" by tests/test_reference_agreement.py: no silent fsolve failure, no singular point).
The reference reports two kinds of "roots":
  * refined ones -- a sign change between two samples of its frequency band, narrowed by locate_* until the mismatch is
    below tol: on a dense omega-grid over the band the grid search must report an accepted root inside the reference's
    own sampling interval around each of them;
  * sampled ones -- a sample of the band itself at which the mismatch is already below its (loose: 1-6 %) tolerance,
    accepted without any bracket: there need not be a zero of D next to them, so the grid search is not asked for one;
    the GPU evaluation at that very frequency must satisfy the reference's acceptance rule."""
import json
import os

import numpy as np
import pytest

from tests import refcases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
AGREEMENT = json.load(open(os.path.join(G, "agreement_table.json")))
NW = 1536


@pytest.mark.parametrize("kind,name", [("roots", n) for n in refcases.rootset_names()] +
                         [("trace", n) for n in refcases.trace_names()])
def test_grid_search_finds_the_reference_roots(es_ctx, kind, name):
    key, factory = refcases.solver_factories()[name]
    solver = factory(es_ctx)
    calls = refcases.load_calls(kind, name)
    cats = AGREEMENT[f"{kind}:{name}"]
    n_refined = n_found = n_sampled = n_sampled_ok = 0
    missing = []
    for mode in sorted({c["fn"] for c in calls}):
        idx = [i for i, c in enumerate(calls) if c["fn"] == mode and cats[i] == "identical" and len(c["roots_w"])]
        if not idx:
            continue
        prob = solver.problem(mode)
        tol = float(solver.WORKER[mode][0])
        k = np.array([calls[i]["k"] for i in idx])
        wq = np.empty((len(idx), NW))
        for r, i in enumerate(idx):
            f = np.sort(calls[i]["freq"])
            wq[r] = np.linspace(f[0], f[-1], NW)
        D, st = prob.eval_grid(k, wq, w_mode=2)
        roots, cnt = prob.find_roots(k, wq, D, st, w_mode=2, n_bisect=16, tol_percent=tol)
        row = roots["row"].cpu().numpy(); w = roots["w"].cpu().numpy(); ok = roots["flag"].cpu().numpy() == 1
        sampled = []
        for r, i in enumerate(idx):
            f = np.sort(calls[i]["freq"])
            mine = w[(row == r) & ok]
            for wr in calls[i]["roots_w"]:
                if np.min(np.abs(f - wr)) <= 1e-12 * abs(wr):
                    sampled.append((calls[i]["k"], wr))
                    continue
                j = int(np.clip(np.searchsorted(f, wr), 1, len(f) - 1))
                lo, hi = f[j - 1], f[j]                      # the reference's own sampling interval around its root
                n_refined += 1
                if np.any((mine >= lo) & (mine <= hi)):
                    n_found += 1
                else:
                    missing.append((mode, calls[i]["k"], wr, lo, hi))
        if sampled:
            Ds, sts, rels = prob.eval_points([a for a, _ in sampled], [b for _, b in sampled], want_rel=True)
            good = (sts.cpu().numpy() == 0) & (rels.cpu().numpy() < tol)
            n_sampled += len(sampled)
            n_sampled_ok += int(good.sum())
    print(f"{kind}:{name}: refined reference roots recovered by the grid search {n_found} / {n_refined}; sampled ones accepted by the GPU "
          f"evaluation {n_sampled_ok} / {n_sampled}; missing: {missing[:4]}")
    assert n_found == n_refined, (name, missing[:6])
    assert n_sampled_ok == n_sampled, name
    solver.close()
