"""root post-processing and the pickle layout, against the literal expressions of the
reference's analysis scripts applied to the reference's own stored roots."""
import os
import pickle
import pickletools

import numpy as np
import pytest

from eigensolver_amd import postprocess as pp
from tests import stored_sets as S


def test_sort_matches_reference_expression():
    for tag in ("cyl_flow_coronal_noflow", "slab_flow_coronal_w15"):
        for mode, w, k in S.pairs(tag):
            ref_w = np.array([x for _, x in sorted(zip(k, w))])          # analysis_cylinder_flow_coronal.py:224
            ref_k = np.sort(k)
            sw, sk = pp.sort_by_wavenumber(w, k)
            assert np.array_equal(sw, ref_w) and np.array_equal(sk, ref_k)


def test_split_and_fit():
    mode, w, k = S.pairs("cyl_flow_coronal_noflow")[1]                    # kink roots of the uniform cylinder
    w, k = pp.sort_by_wavenumber(w, k)
    bands = {"fast": (2.0, 5.0), "slow": (0.8944271909999159, 1.0), "backward_fast": (-5.0, -2.0)}
    br = pp.split_branches(w, k, bands)
    assert sum(len(v[0]) for v in br.values()) <= len(w)
    fw, fk = br["fast"]
    assert len(fw) > 50 and np.all((fw / fk > 2.0) & (fw / fk < 5.0))
    # the band holds several radial harmonics; the fundamental is the lowest phase speed at each k
    uk = np.unique(fk[fk > 0.5])
    W0 = np.array([np.min(fw[fk == x] / x) for x in uk])
    poly = pp.fit_branch(uk, W0, deg=4)
    assert np.median(np.abs(poly(uk) - W0)) < 0.05        # a few k lack the fundamental in the stored set (jumps)


def test_pickle_layout_roundtrip(tmp_path):
    res = {"sausage": (np.array([1.0, 2.0]), np.array([0.5, 0.5])), "kink": (np.array([3.0]), np.array([0.7]))}
    path = os.path.join(tmp_path, "out.pickle")
    pp.save_pickle(path, res)
    # the file has the structure of the reference's pickles: a list of four 1-D float64 arrays
    ops = [op.name for op, arg, pos in pickletools.genops(open(path, "rb").read())]
    assert "EMPTY_LIST" in ops or "LIST" in ops
    a = pickle.load(open(path, "rb"))                                     # our own file
    assert len(a) == 4 and all(isinstance(x, np.ndarray) and x.dtype == np.float64 and x.ndim == 1 for x in a)
    assert np.array_equal(a[0], res["sausage"][0]) and np.array_equal(a[3], res["kink"][1])
    assert len(pp.pickle_layout({"kink": (np.zeros(2), np.ones(2))})) == 2       # rotational scripts: [w, k]


def test_vtk_dump_layout(tmp_path):
    """Legacy-VTK structured grid as Export_vtk.py:70-112 writes it: header text, big-endian float32, (x, y, z)
    interleaved with the first index fastest, one SCALARS block per variable -- checked against an element-by-element
    packer."""
    import struct
    rng = np.random.default_rng(0)
    shape = (3, 4, 2)
    x, y, z = (rng.normal(size=shape) for _ in range(3))
    v1, v2 = rng.normal(size=shape), rng.normal(size=shape)
    path = pp.write_vtk(tmp_path / "dump", x, y, z, [v1, v2], ["P", "xi_r"])
    got = open(path, "rb").read()
    ax, ay, az = shape
    want = b"# vtk DataFile Version 3.0 \nvtk output \nBINARY \nDATASET STRUCTURED_GRID \n"
    want += b"DIMENSIONS  3 4 2  \nPOINTS 24 float  \n"
    for k in range(az):
        for j in range(ay):
            for i in range(ax):
                want += struct.pack(">f", x[i, j, k]) + struct.pack(">f", y[i, j, k]) + struct.pack(">f", z[i, j, k])
    want += b"\nPOINT_DATA 24  "
    for name, v in (("P", v1), ("xi_r", v2)):
        want += b"\nSCALARS " + name.encode() + b" float \nLOOKUP_TABLE default \n"
        for k in range(az):
            for j in range(ay):
                for i in range(ax):
                    want += struct.pack(">f", v[i, j, k])
    assert got == want
    with pytest.raises(ValueError):
        pp.write_vtk(tmp_path / "bad", x, y, z, [v1[:2]], ["P"])
