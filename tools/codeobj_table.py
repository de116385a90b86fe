"""Register / spill / LDS table of the shipped gfx950 kernels, read from the code objects themselves.

    python tools/codeobj_table.py [--filter shoot_grid] [--md | --json]

For every built object under eigensolver_amd/lib/*.o (the non-IEEE build) the .hip_fatbin section is unbundled and the
AMDGPU metadata note of the gfx950 code object is read with llvm-readelf: VGPRs, AGPRs, spilled VGPRs / SGPRs, scratch
bytes per lane (private segment), static LDS bytes.  profiles/README.md quotes THIS output (claims about spills are
generated, not typed); tests/test_codeobj.py holds the march kernels to "no spill inside a march loop".
"""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = (".vgpr_count", ".agpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
          ".group_segment_fixed_size", ".sgpr_count", ".max_flat_workgroup_size")


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.split("\n")[:len(names)]


def short(name):
    """`void (anonymous namespace)::shoot_grid_kernel<0, 4, 256, false, 3>(ShootDev, ...)` -> `shoot_grid_kernel<0,4,256,false,3>`"""
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:                       # cut at the argument list (first '(' outside template brackets)
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).replace(", ", ",").replace("(unsigned char)", "").replace("(int)", "")


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat):
            return []
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
        if r.returncode != 0:
            return []
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    out, cur = [], None
    for line in notes.split("\n"):
        s = line.strip()
        if s.startswith("- .agpr_count") or (s.startswith("- .") and cur is not None and ".name" in cur and ".vgpr_count" in cur):
            if cur and ".name" in cur:
                out.append(cur)
            cur = {}
            s = s[2:]
        m = re.match(r"(\.[a-z_]+):\s+(.*)$", s)
        if m and cur is not None:
            key, val = m.group(1), m.group(2).strip()
            if key == ".name" and ".name" not in cur:
                cur[key] = val.strip("'")
            elif key in FIELDS and key not in cur:
                cur[key] = int(val)
    if cur and ".name" in cur:
        out.append(cur)
    out = [k for k in out if ".vgpr_count" in k]
    names = demangle([k[".name"] for k in out])
    for k, n in zip(out, names):
        k["kernel"] = short(n)
        k["object"] = os.path.basename(obj)
    return out


def table(filt=None):
    rows = []
    for obj in sorted(glob.glob(os.path.join(ROOT, "eigensolver_amd", "lib", "*.o"))):
        if obj.endswith(".ieee.o"):
            continue
        rows += kernels_of(obj)
    if filt:
        rows = [r for r in rows if re.search(filt, r["kernel"])]
    rows.sort(key=lambda r: (r["object"], r["kernel"]))
    return rows


def main():
    filt = None
    if "--filter" in sys.argv:
        filt = sys.argv[sys.argv.index("--filter") + 1]
    rows = table(filt)
    if "--json" in sys.argv:
        print(json.dumps(rows, indent=1))
        return
    print("| kernel | VGPRs | AGPRs | spilled VGPRs | spilled SGPRs | scratch B/lane | LDS B |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| `{r['kernel']}` | {r['.vgpr_count']} | {r.get('.agpr_count', 0)} | {r.get('.vgpr_spill_count', 0)} | "
              f"{r.get('.sgpr_spill_count', 0)} | {r.get('.private_segment_fixed_size', 0)} | {r.get('.group_segment_fixed_size', 0)} |")


if __name__ == "__main__":
    main()
