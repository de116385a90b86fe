"""Convert the reference's stored result pickles (`*/Example data/*.pickle`) into plain .npz fixtures WITHOUT
unpickling: the files are walked with pickletools.genops (a disassembler that executes nothing) and the float64
payload of every numpy array is taken from its raw-bytes argument.

    python tools/pickle_to_npz.py     ->  tests/golden/stored_roots.npz  (keys "<tag>/<i>", all 90 pickles)
                                          tests/golden/stored_roots_index.json  (tag -> reference file, array sizes)

Layout of a reference pickle (Density_cylinder.py:1182-1183): a list of 4 float64 1-D arrays
[omega_sausage, k_sausage, omega_kink, k_kink]; rotational files hold 2 arrays [omega, k].
"""
import os
import pickletools
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "stored_roots.npz")

# tags of the first fixture set keep their historical names; every other pickle gets a tag derived from its path
LEGACY_TAGS = {
    "Slab/Non uniform density/Photospheric/Example data/width1e5.pickle": "slab_density_photospheric_w1e5",
    "Slab/Non uniform density/Photospheric/Example data/width15.pickle": "slab_density_photospheric_w15",
    "Slab/Non uniform density/Coronal/Example data/width1e5_coronal.pickle": "slab_density_coronal_w1e5",
    "Slab/Non uniform flow/Example data/flow_width1e5_coronal.pickle": "slab_flow_coronal_w1e5",
    "Slab/Non uniform flow/Example data/flow_width15_coronal.pickle": "slab_flow_coronal_w15",
    "Cylinder/Non-uniform density/Coronal/Example data/Cylindrical_coronal_width1e5.pickle": "cyl_density_coronal_w1e5",
    "Cylinder/Non-uniform density/Coronal/Example data/Cylindrical_coronal_width09.pickle": "cyl_density_coronal_w09",
    "Cylinder/Non-uniform density/Coronal/Example data/Cylindrical_coronal_width15.pickle": "cyl_density_coronal_w15",
    "Cylinder/Non-uniform density/Photospheric/Example data/Cylindrical_photospheric_width_1e5.pickle": "cyl_density_photospheric_w1e5",
    "Cylinder/Non-uniform flow/Coronal/Example data/Cylindrical_coronal_flow_noflow.pickle": "cyl_flow_coronal_noflow",
    "Cylinder/Rotational flow/Photospheric/Example data/Cylindrical_photospheric_vtwist01_power1_fund_kink.pickle": "cyl_rot_v01_p1_fund_kink",
    "Cylinder/Rotational flow/Photospheric/Example data/Cylindrical_photospheric_vtwist01_power08_sausage_fast.pickle": "cyl_rot_v01_p08_sausage_fast",
}
FAMILY_PREFIX = [
    ("Slab/Non uniform density/Photospheric", "slab_density_photospheric"),
    ("Slab/Non uniform density/Coronal", "slab_density_coronal"),
    ("Slab/Non uniform flow", "slab_flow_coronal"),
    ("Cylinder/Non-uniform density/Coronal", "cyl_density_coronal"),
    ("Cylinder/Non-uniform density/Photospheric", "cyl_density_photospheric"),
    ("Cylinder/Non-uniform flow/Coronal", "cyl_flow_coronal"),
    ("Cylinder/Rotational flow/Photospheric", "cyl_rot"),
]


def tag_of(rel):
    """Stable fixture tag of a reference pickle (path relative to the reference root)."""
    if rel in LEGACY_TAGS:
        return LEGACY_TAGS[rel]
    fam = next(pre for path, pre in FAMILY_PREFIX if rel.startswith(path))
    stem = os.path.splitext(os.path.basename(rel))[0]
    for junk in ("Cylindrical_photospheric_", "Cylindrical_coronal_", "_coronal", "flow_"):
        stem = stem.replace(junk, "")
    stem = stem.replace("width_", "w").replace("width", "w").replace("vtwist", "v").replace("power", "p")
    return f"{fam}_{stem}"


def all_files():
    out = {}
    for root, _, names in os.walk(REF):
        for n in sorted(names):
            if n.endswith(".pickle"):
                rel = os.path.relpath(os.path.join(root, n), REF)
                t = tag_of(rel)
                assert t not in out, (t, rel, out[t])
                out[t] = rel
    return dict(sorted(out.items()))


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
    import json
    res, index = {}, {}
    for tag, rel in all_files().items():
        arrs = arrays_from_pickle(os.path.join(REF, rel))
        assert all(np.all(np.isfinite(a)) for a in arrs), tag
        assert len(arrs) in (2, 4), (tag, len(arrs), [a.shape for a in arrs])
        for i, a in enumerate(arrs):
            res[f"{tag}/{i}"] = a
        index[tag] = {"file": rel, "sizes": [int(a.size) for a in arrs]}
        print(tag, [a.shape for a in arrs])
    np.savez_compressed(OUT, **res)
    with open(OUT.replace(".npz", "_index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)
    print("wrote", OUT, len(index), "files")


if __name__ == "__main__":
    sys.exit(main())
