"""Root-table post-processing and result formats -- what every analysis script of the
reference does first with a result pickle (e.g. Cylinder/Non-uniform flow/Coronal/Eigenfunctions/
analysis_cylinder_flow_coronal.py:216-250 and the commented classifiers :282-372): sort the (k, omega) pairs by
wavenumber, split them into branches by phase-speed bands, fit a polynomial per branch; plus the pickle layout
`[omega_sausage, k_sausage, omega_kink, k_kink]` the solver scripts dump (Density_cylinder.py:1182-1183) so that the
reference's own analysis / movie scripts can consume the GPU results unchanged.  Host-side NumPy: the tables hold a
few thousand roots."""
import pickle

import numpy as np


def sort_by_wavenumber(omegas, ks):
    """`[x for _, x in sorted(zip(ks, omegas))]`, `np.sort(ks)` (analysis_cylinder_flow_coronal.py:224-228): pairs
    ordered by k, ties by omega."""
    omegas, ks = np.asarray(omegas, dtype=np.float64), np.asarray(ks, dtype=np.float64)
    order = np.lexsort((omegas, ks))
    return omegas[order], ks[order]


def split_branches(omegas, ks, bands):
    """Classify roots by phase speed: bands = {name: (lo, hi)} -> {name: (omegas, ks)} with lo < omega/k < hi
    (the commented classifier blocks, e.g. fast body vA_i < w/k < vA_e, :244-372)."""
    omegas, ks = np.asarray(omegas, dtype=np.float64), np.asarray(ks, dtype=np.float64)
    with np.errstate(all="ignore"):
        W = omegas / ks
    out = {}
    for name, (lo, hi) in bands.items():
        sel = (W > lo) & (W < hi)
        out[name] = (omegas[sel], ks[sel])
    return out


def fit_branch(ks, omegas, deg=6):
    """Polynomial fit omega(k) of one branch (np.polyfit / np.poly1d as in the analysis scripts)."""
    if len(ks) <= deg:
        raise ValueError("not enough points for the requested degree")
    return np.poly1d(np.polyfit(np.asarray(ks, dtype=np.float64), np.asarray(omegas, dtype=np.float64), deg))


def pickle_layout(result):
    """`solve()` output {"sausage": (w, k), "kink": (w, k)} -> the list the reference dumps:
    [sol_omegas1, sol_ks1, sol_omegas_kink1, sol_ks_kink1]; single-mode scripts (rotational) dump [w, k]."""
    if "sausage" in result and "kink" in result:
        return [np.asarray(result["sausage"][0]), np.asarray(result["sausage"][1]),
                np.asarray(result["kink"][0]), np.asarray(result["kink"][1])]
    (w, k), = result.values()
    return [np.asarray(w), np.asarray(k)]


def save_pickle(path, result):
    """Write the result in the reference's pickle layout (Density_cylinder.py:1182-1183)."""
    with open(path, "wb") as f:
        pickle.dump(pickle_layout(result), f)


def from_root_table(roots, accepted_only=True):
    """Grid-mode root table (ShootProblem.find_roots) -> (omegas, ks) NumPy arrays."""
    w = roots["w"].detach().cpu().numpy()
    k = roots["k"].detach().cpu().numpy()
    if accepted_only:
        sel = roots["flag"].detach().cpu().numpy() == 1
        w, k = w[sel], k[sel]
    return w, k


def write_vtk(dump_file, x, y, z, variables, names):
    """Legacy-VTK structured-grid dump of scalar fields on an irregular grid, byte for byte what the reference's
    `makeDumpVTK(x, y, z, variables, varList, dumpFile)` writes (Cylinder/Non-uniform density/Coronal/Movies/
    Export_vtk.py:70-112): header lines as there (trailing blanks included), BINARY, big-endian float32, points as
    interleaved (x, y, z) with the first index fastest, one `SCALARS <name> float` block per variable.
    x, y, z and every variable are arrays of the same shape (ax, ay, az).  Writes `<dump_file>.vtk`."""
    x, y, z = (np.asarray(a, dtype=np.float64) for a in (x, y, z))
    if not (x.ndim == 3 and x.shape == y.shape == z.shape):
        raise ValueError("x, y, z must be 3-D arrays of one shape")
    if len(variables) != len(names):
        raise ValueError("one name per variable")
    ax, ay, az = x.shape
    n = ax * ay * az

    def packed(a):                     # loop order k, j, i with i fastest = Fortran order of an (i, j, k) array
        return np.asarray(a, dtype=np.float64).astype(">f4").ravel(order="F")

    with open(str(dump_file) + ".vtk", "wb") as f:
        f.write(b"# vtk DataFile Version 3.0 \n")
        f.write(b"vtk output \n")
        f.write(b"BINARY \n")
        f.write(b"DATASET STRUCTURED_GRID \n")
        f.write(("DIMENSIONS  %s %s %s  \n" % (ax, ay, az)).encode())
        f.write(("POINTS %s float  \n" % n).encode())
        pts = np.empty((n, 3), dtype=">f4")
        pts[:, 0], pts[:, 1], pts[:, 2] = packed(x), packed(y), packed(z)
        f.write(pts.tobytes())
        f.write(("\nPOINT_DATA %s  " % n).encode())
        for name, var in zip(names, variables):
            var = np.asarray(var)
            if var.shape != x.shape:
                raise ValueError(f"variable {name!r} has shape {var.shape}, grid is {x.shape}")
            f.write(("\nSCALARS %s float \n" % name).encode())
            f.write(b"LOOKUP_TABLE default \n")
            f.write(packed(var).tobytes())
    return str(dump_file) + ".vtk"
