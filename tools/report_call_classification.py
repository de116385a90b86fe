"""Classify every recorded reference worker call against the algorithm the GPU runs (CPU only: oracle state machine +
C port, tests/agreement.py) and write tests/golden/agreement_table.json, the committed per-call category table that
tests/test_reference_agreement.py reproduces; prints the markdown table of DESIGN.md section 6.

    python tools/report_call_classification.py [--no-write] [-v] [--converged]

--converged: the second fixture set (tests/golden/*_conv*, tools/gen_golden.py --converge: the reference run again with
its unconverged fsolve calls re-solved to convergence on its own objective) -> tests/golden/agreement_table_conv.json.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import agreement as A, refcases as R  # noqa: E402

CATS = ["identical", "fsolve", "singular", "exterior", "interior_noise", "objective_noise", "unexplained", "no_trace"]


def main():
    verbose = "-v" in sys.argv
    conv = "--converged" in sys.argv
    table, rows, tot = {}, [], {}
    for kind, names in (("trace", R.trace_names(conv)), ("roots", R.rootset_names(conv))):
        for name in names:
            t = time.time()
            res = A.classify_fixture(kind, name, conv=conv)
            sm = A.summarize(res)
            table[f"{kind}:{name}"] = [r["category"] for r in res]
            rows.append((kind, name, R.solver_factories()[name][0], len(res), sm, sum(r["same_roots"] for r in res)))
            for c, v in sm.items():
                tot[c] = tot.get(c, 0) + v
            if verbose:
                print(f"# {kind}:{name} {time.time() - t:.1f} s", file=sys.stderr)
                for r in res:
                    if r["category"] != "identical":
                        print("   ", {k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items()
                                      if k != "ours_roots"}, file=sys.stderr)
    nc = 6 if conv else 5
    print("| fixture | script | calls | " + " | ".join(CATS[:nc]) + " | root list identical |")
    print("|---|---|---|" + "---|" * (nc + 1))
    for kind, name, key, n, sm, same in rows:
        print(f"| {kind}:{name} | {key} | {n} | " + " | ".join(str(sm.get(c, 0)) for c in CATS[:nc]) + f" | {same} |")
    n_all = sum(r[3] for r in rows)
    print(f"| **total** | | {n_all} | " + " | ".join(str(tot.get(c, 0)) for c in CATS[:nc]) +
          f" | {sum(r[5] for r in rows)} |")
    bad = tot.get("unexplained", 0) + tot.get("no_trace", 0)
    print(f"\nunexplained: {tot.get('unexplained', 0)}, without evaluation records: {tot.get('no_trace', 0)}")
    if "--no-write" not in sys.argv:
        with open(os.path.join(R.G, "agreement_table_conv.json" if conv else "agreement_table.json"), "w") as f:
            json.dump(table, f, indent=0, sort_keys=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
