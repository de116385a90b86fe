"""Known-answer sets: roots stored by the reference authors (`*/Example data/*.pickle`, converted without
unpickling by tools/pickle_to_npz.py) with the parameter set each file was generated with.
A stored root (k, omega) must satisfy the acceptance measure rel = 100|d|/max(|outer|,|inner|) < tol of its worker."""
import os

import numpy as np

from eigensolver_amd import equilibrium as q

NPZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stored_roots.npz")

PHOTO = dict(c_e=1.5, vA_e=0.5, r_sign=1.0, n_nodes=1000, ic=(1e-8, 1e-8))
# tag -> (equilibrium, tolerance in percent to test with, minimum fraction (sausage, kink) that must pass)
SETS = {
    "slab_density_photospheric_w1e5": (q.SlabDensity(width=1e5, n_nodes=1001), 1.0, (0.97, 0.97)),
    "slab_density_photospheric_w15": (q.SlabDensity(width=1.5, n_nodes=1001), 3.0, (0.55, 0.90)),   # 116/214 sausage roots lie in a continuum band
    "slab_density_coronal_w1e5": (q.SlabDensity(width=1e5, vA_i0=1.2, vA_e=3.0, c_e=0.4, L_factor=3.0, n_nodes=1001), 1.0, (0.85, 0.88)),
    "slab_flow_coronal_w1e5": (q.SlabFlow(U_i0=0.35, width=1e5), 1.0, (0.85, 0.85)),
    "slab_flow_coronal_w15": (q.SlabFlow(U_i0=0.35, width=1.5), 1.0, (0.82, 0.85)),
    "cyl_density_coronal_w1e5": (q.CylinderDensity(width=1e5), 1.0, (0.97, 0.97)),
    "cyl_density_coronal_w09": (q.CylinderDensity(width=0.9), 1.0, (0.90, 0.85)),
    "cyl_density_coronal_w15": (q.CylinderDensity(width=1.5), 1.0, (0.93, 0.93)),
    "cyl_density_photospheric_w1e5": (q.CylinderDensity(width=1e5, **PHOTO), 3.0, (0.93, 0.97)),
    "cyl_flow_coronal_noflow": (q.CylinderFlow(), 6.0, (0.97, 0.97)),
    "cyl_rot_v01_p1_fund_kink": (q.CylinderRotation(v_twist=0.1, power=1.0), 3.0, (None, 0.95)),
    "cyl_rot_v01_p08_sausage_fast": (q.CylinderRotation(v_twist=0.1, power=0.8, r_axis=0.01), 6.0, (0.93, None)),
}


def pairs(tag):
    g = np.load(NPZ)
    n = len([k for k in g.files if k.startswith(tag + "/")])
    if n == 4:
        return [("sausage", g[tag + "/0"], g[tag + "/1"]), ("kink", g[tag + "/2"], g[tag + "/3"])]
    return [("kink" if "kink" in tag else "sausage", g[tag + "/0"], g[tag + "/1"])]
