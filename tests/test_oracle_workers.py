"""The worker state machine (oracle/workers.py: accept / bracket / recursive 3-point refinement)
replayed on the reference's own numbers: feeding it the mismatch values the reference computed (golden traces),
it must evaluate exactly the same sequence of frequencies and `put` exactly the same roots."""
import json
import math
import os

import numpy as np
import pytest

from oracle import workers as W
from tests.test_oracle_golden import PROBLEMS, G

FILE_KEY = {"CF_uniform": "CF", "CF_flow": "CF", "CDC_w095": "CD-C", "CDC_uniform": "CD-C", "CDP": "CD-P",
            "CRSF": "CR-SF", "CRKS": "CR-KS", "CRKF": "CR-KF", "SFU": "SF-U", "SFG_uniform": "SF-G",
            "SFG_flow": "SF-G", "SDP_uniform": "SD-P", "SDP_w15": "SD-P", "SDC_w09": "SD-C", "SDC_uniform": "SD-C",
            "CRSS": "CR-SS"}


def _replay_evaluator(prob, call):
    # the reference's evaluation is not a pure function of omega next to a pole (ill-conditioned shoot + fsolve):
    # the same frequency can return different mismatches on re-evaluation, so values are replayed in order
    table = {}
    for ev in call["evals"]:
        if ev["omega"] is None or ev["d"] is None:
            continue
        ext = ev["ext_end"]
        slope = ext[2] if len(ext) == 4 else ext[1]
        table.setdefault(ev["omega"], []).append((ev["d"], slope))

    def evaluate(k, w, w_cst=None):
        m_e, cst = prob.exterior(k, w)[:2]
        if w_cst is not None:                      # CR-SF: xi_e_const of the grid point that opened the refinement
            cst = prob.exterior(k, w_cst)[1]
        if m_e < 0:
            return W.ST_LEAKY, float("nan"), float("nan"), float("nan")
        q = table[w]                             # KeyError = the state machine asked for a point the reference never evaluated
        d, slope = q.pop(0) if len(q) > 1 else q[0]
        outer = cst * slope
        return W.ST_OK, d, outer, outer - d
    return evaluate


@pytest.mark.parametrize("case", sorted(FILE_KEY))
def test_replay_reference_trace(case):
    tr = json.load(open(os.path.join(G, f"trace_{case}.json")))
    for call in tr["calls"]:
        spec = W.SPECS[(FILE_KEY[case], call["fn"])]
        # the tolerance in the trace is the one the reference ran with
        tol = tr.get("xi_tol") if call["fn"] == "kink" and "xi_tol" in tr else tr.get("p_tol", tr.get("P_tol"))
        if FILE_KEY[case].startswith(("CD-", "CF")) or FILE_KEY[case] in ("CR-SF", "CR-SS"):
            tol = tr["xi_tol"]               # both cylinder workers test against xi_tol (CD-C:809, :1106)
        assert tol == spec.tol, (case, tol, spec.tol)
        prob = PROBLEMS[case](call["fn"])
        roots, ks, requested = W.run_worker(spec, _replay_evaluator(prob, call), call["k"], call["freq"])
        ref_seq = [(e["where"], e["omega"]) for e in call["evals"] if e["omega"] is not None and e["d"] is not None]
        assert requested == ref_seq, (case, call["fn"], call["k"])
        assert roots == call["roots_w"], (case, call["fn"], call["k"], roots, call["roots_w"])
        assert ks == call["roots_k"]


def test_band_builder_matches_reference_driver():
    # CD-C:225-228 / :1142-1145 -- note the missing comma `cT_e -c_e` in the reference's speeds list
    c_i0, c_e, vA_i0, vA_e = 1.0, 0.5, 2.0, 5.0
    cT_i0 = math.sqrt(c_i0 ** 2 * vA_i0 ** 2 / (c_i0 ** 2 + vA_i0 ** 2))
    cT_e = math.sqrt(c_e ** 2 * vA_e ** 2 / (c_e ** 2 + vA_e ** 2))
    speeds = [c_i0, c_e, vA_i0, vA_e, cT_i0, cT_e - c_e, -c_i0, -vA_i0, -vA_e, -cT_i0, -cT_e]
    bands = W.band_frequencies(speeds, 2.0, 90)
    assert len(bands) == 10 and all(len(b) == 90 for b in bands)
    assert bands[0][0] == -vA_e * 2.0 and bands[-1][-1] == vA_e * 2.0
    f = W.sfu_frequencies(1.0, 0.5547, 0.75, -0.15)
    assert len(f[0]) == 80 and len(f[1]) == 100 and abs(f[0][0] - (10 ** 0.001 - 1)) < 1e-15
