"""Convert the reference's stored result pickles (`*/Example data/*.pickle`) into plain .npz fixtures WITHOUT
unpickling: the files are walked with pickletools.genops (a disassembler that executes nothing) and the float64
payload of every numpy array is taken from its raw-bytes argument.

    python tools/pickle_to_npz.py     ->  tests/golden/stored_roots.npz  (keys "<tag>/<i>")

Layout of a reference pickle (Density_cylinder.py:1182-1183): a list of 4 float64 1-D arrays
[omega_sausage, k_sausage, omega_kink, k_kink]; rotational files hold 2 arrays [omega, k].
"""
import os
import pickletools
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "stored_roots.npz")

FILES = {
    "slab_density_photospheric_w1e5": "Slab/Non uniform density/Photospheric/Example data/width1e5.pickle",
    "slab_density_photospheric_w15": "Slab/Non uniform density/Photospheric/Example data/width15.pickle",
    "slab_density_coronal_w1e5": "Slab/Non uniform density/Coronal/Example data/width1e5_coronal.pickle",
    "slab_flow_coronal_w1e5": "Slab/Non uniform flow/Example data/flow_width1e5_coronal.pickle",
    "slab_flow_coronal_w15": "Slab/Non uniform flow/Example data/flow_width15_coronal.pickle",
    "cyl_density_coronal_w1e5": "Cylinder/Non-uniform density/Coronal/Example data/Cylindrical_coronal_width1e5.pickle",
    "cyl_density_coronal_w09": "Cylinder/Non-uniform density/Coronal/Example data/Cylindrical_coronal_width09.pickle",
    "cyl_density_coronal_w15": "Cylinder/Non-uniform density/Coronal/Example data/Cylindrical_coronal_width15.pickle",
    "cyl_density_photospheric_w1e5": "Cylinder/Non-uniform density/Photospheric/Example data/Cylindrical_photospheric_width_1e5.pickle",
    "cyl_flow_coronal_noflow": "Cylinder/Non-uniform flow/Coronal/Example data/Cylindrical_coronal_flow_noflow.pickle",
    "cyl_rot_v01_p1_fund_kink": "Cylinder/Rotational flow/Photospheric/Example data/Cylindrical_photospheric_vtwist01_power1_fund_kink.pickle",
    "cyl_rot_v01_p08_sausage_fast": "Cylinder/Rotational flow/Photospheric/Example data/Cylindrical_photospheric_vtwist01_power08_sausage_fast.pickle",
}

ALLOWED_GLOBALS = {("numpy.core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
                   ("numpy._core.multiarray", "_reconstruct")}


def arrays_from_pickle(path):
    """Return the list of float64 arrays in the file, in order, by static inspection of the opcode stream."""
    data = open(path, "rb").read()
    out = []
    last_strings = []
    for op, arg, pos in pickletools.genops(data):
        name = op.name
        if name in ("GLOBAL",):
            mod, attr = arg.split(" ")
            if (mod, attr) not in ALLOWED_GLOBALS:
                raise ValueError(f"unexpected global {arg} in {path}")
        elif name in ("STACK_GLOBAL",):
            if tuple(last_strings[-2:]) not in ALLOWED_GLOBALS:
                raise ValueError(f"unexpected stack global {last_strings[-2:]} in {path}")
        elif name in ("SHORT_BINUNICODE", "BINUNICODE", "UNICODE", "SHORT_BINSTRING", "BINSTRING"):
            if isinstance(arg, str):
                last_strings.append(arg)
            if isinstance(arg, (bytes, str)) and name in ("SHORT_BINSTRING", "BINSTRING") and len(arg) >= 8 and len(arg) % 8 == 0:
                raw = arg.encode("latin-1") if isinstance(arg, str) else arg
                out.append(np.frombuffer(raw, dtype="<f8").copy())
        elif name in ("BINBYTES", "SHORT_BINBYTES", "BINBYTES8"):
            if len(arg) % 8 == 0 and len(arg) > 0:
                out.append(np.frombuffer(arg, dtype="<f8").copy())
        elif name in ("REDUCE", "BUILD", "TUPLE", "TUPLE1", "TUPLE2", "TUPLE3", "EMPTY_TUPLE", "EMPTY_LIST", "MARK",
                      "APPENDS", "APPEND", "BINPUT", "LONG_BINPUT", "BINGET", "LONG_BINGET", "PROTO", "FRAME",
                      "STOP", "BININT", "BININT1", "BININT2", "NEWFALSE", "NEWTRUE", "NONE", "MEMOIZE", "LIST",
                      "PUT", "GET", "INT", "LONG1"):
            pass
        else:
            raise ValueError(f"unexpected opcode {name} in {path}")
    return out


def main():
    res = {}
    for tag, rel in FILES.items():
        arrs = arrays_from_pickle(os.path.join(REF, rel))
        # drop the tiny 'b' dtype-description strings that are not multiples of 8 (already filtered) and keep data
        arrs = [a for a in arrs if np.all(np.isfinite(a))]
        assert len(arrs) in (2, 4), (tag, len(arrs), [a.shape for a in arrs])
        for i, a in enumerate(arrs):
            res[f"{tag}/{i}"] = a
        print(tag, [a.shape for a in arrs])
    np.savez_compressed(OUT, **res)
    print("wrote", OUT)


if __name__ == "__main__":
    sys.exit(main())
