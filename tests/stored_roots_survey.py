"""Evaluate every stored reference root (tests/golden/stored_roots.npz) with the CPU port under the parameters
tests/stored_sets.describe() infers, print the accepted fraction per file / mode and write
tests/golden/stored_roots_floors.json: the MEASURED accepted fraction per file and mode (4 decimals).  The test
(tests/test_stored_roots.py) fails when a fraction falls more than 0.05 below the committed value, so a regression that
lowers the acceptance rate of any stored set by 5 points is caught whatever the absolute level of that set.

    python tests/stored_roots_survey.py [--write]
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # repo root (this file lives in tests/)
sys.path.insert(0, ROOT)
from tests import cases, stored_sets as S  # noqa: E402


def main():
    floors, rows = {}, []
    tot = acc = 0
    for tag in S.TAGS:
        eq, tol = S.describe(tag)
        for mode, w, k in S.pairs(tag):
            if len(w) == 0:
                continue
            D, rel, st = cases.port_problem(eq, mode).eval_points(k, w, nthreads=8)
            ok = rel < tol
            frac = float(np.mean(ok))
            rows.append((tag, mode, len(w), frac, int((st == 3).sum())))
            print(f"{tag:45s} {mode:8s} n={len(w):4d} accepted={frac:5.2f} continuum={int((st == 3).sum()):4d}"
                  + ("   [unpinned]" if tag in S.UNPINNED else ""))
            if tag not in S.UNPINNED:
                floors.setdefault(tag, {})[mode] = round(frac, 4)
                tot += len(w)
                acc += int(ok.sum())
    print(f"pinned files: {len(floors)}  stored roots: {tot}  accepted by the port: {acc} ({100.0 * acc / tot:.1f} %)")
    if "--write" in sys.argv:
        with open(os.path.join(ROOT, "tests", "golden", "stored_roots_floors.json"), "w") as f:
            json.dump(floors, f, indent=1, sort_keys=True)
        print("wrote measured fractions")


if __name__ == "__main__":
    main()
