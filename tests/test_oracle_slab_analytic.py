"""Oracle (oracle/slab.py part A) pinned against values produced by the reference's own functions
(flow_multiprocessor.py:107-127, scan :166-303), stored in tests/golden/slab_analytic.npz by tools/gen_golden.py."""
import os

import numpy as np

from oracle.slab import SlabAnalytic, SAUSAGE, KINK, SAUSAGE_BODY, KINK_BODY

NAMES = {SAUSAGE: "disp_rel_sausage", KINK: "disp_rel_kink", SAUSAGE_BODY: "disp_rel_sausage_body",
         KINK_BODY: "disp_rel_kink_body"}


def _gold(golden_dir):
    return np.load(os.path.join(golden_dir, "slab_analytic.npz"))


def test_constants(golden_dir):
    g = _gold(golden_dir)
    o = SlabAnalytic()
    assert o.R1 == float(g["R1"]) == 2.271604938271605
    assert o.cT_i == float(g["cT_i"])
    assert o.cT_e == float(g["cT_e"]) == 0.0


def test_known_answers():
    o = SlabAnalytic()
    assert o.disp(SAUSAGE, 0.5, 1.0) == -1.9436261147208946
    assert o.disp(KINK, 0.5, 1.0) == -2.9115276150635645


def test_values_bit_exact(golden_dir):
    g = _gold(golden_dir)
    o = SlabAnalytic()
    K, W = g["K"], g["W"]
    for mode, name in NAMES.items():
        D = o.disp(mode, W[None, :], K[:, None])
        ref = g[name]
        assert np.array_equal(np.isnan(D), np.isnan(ref))
        m = ~np.isnan(ref)
        assert np.array_equal(D[m], ref[m]), name            # same numpy calls -> bit identical


def test_scan_matches_reference_loops(golden_dir):
    g = _gold(golden_dir)
    o = SlabAnalytic()
    step = float(g["step"])
    ks, ws = o.scan(SAUSAGE, g["scan_D_range"], g["scan_W_range"], step)
    assert np.array_equal(ks, g["scan_x_out_sausage"]) and np.array_equal(ws, g["scan_W_array_sausage"])
    assert len(ks) == 70 and ks[0] == 0.05 and ws[0] == 0.5415000000000001
    ks, ws = o.scan(KINK, g["scan_D_range"], g["scan_W_range"], step)
    assert np.array_equal(ks, g["scan_x_out_kink"]) and np.array_equal(ws, g["scan_W_array_kink"])
    # body modes: the reference scans W_body_range then W_body_range2 for every K (SF-U:208-240)
    for mode, kx, kw in ((SAUSAGE_BODY, "scan_x_out_sausage_body", "scan_W_array_sausage_body"),
                         (KINK_BODY, "scan_x_out_kink_body", "scan_W_array_kink_body")):
        allk, allw = [], []
        for x in g["scan_D_range"]:
            for rng in (g["scan_W_body_range"], g["scan_W_body_range2"]):
                k1, w1 = o.scan(mode, [x], rng, step)
                allk.append(k1), allw.append(w1)
        allk, allw = np.concatenate(allk), np.concatenate(allw)
        assert np.array_equal(allk, g[kx]) and np.array_equal(allw, g[kw])
        fk, fw = o.pole_filter(mode, allk, allw)
        assert np.array_equal(fk, g[kx + "1"]) and np.array_equal(fw, g[kw + "1"])
