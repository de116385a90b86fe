"""Instruction count of the march loop of every built grid kernel, read from the disassembly of the shipped code object.

    python tools/isa_loop_count.py [--write]      (--write: profiles/isa_loop_counts.json, read by bench.py)

The march loop of a kernel = the innermost backward-branch region that contains LDS reads (the node entries) and
v_rcp_f64 (ONE reciprocal per point and RK4 step -- per point and TWO steps for the untwisted cylinder since its marches
share a division between two steps: their number in the loop body gives points x steps per iteration).
Per point-step: fp64 VALU instructions (4 issue cycles per wave on gfx950: 16 lanes per cycle), v_rcp_f64 (quarter
rate: 16 cycles), other VALU (4), LDS reads (issued through the same port; counted, priced at 0 in `issue_cycles`),
scalar / wait instructions.  bench.py prices a launch with it:  issue_cycles x (points / 64) x steps / 1024 SIMDs
against the measured cycles of the launch (valu_issue), instead of numbers typed into the source.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import codeobj_table  # noqa: E402

LLVM = codeobj_table.LLVM
OUT = os.path.join(ROOT, "profiles", "isa_loop_counts.json")
INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")


def disassemble(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj], check=True,
                       capture_output=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout


def kernels(asm):
    """{mangled name: [(addr, mnemonic, operands)]}"""
    out, cur = {}, None
    for line in asm.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        m = INSN.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def classify(mn):
    if mn == "v_rcp_f64_e32" or mn.startswith("v_rcp_f64"):
        return "rcp_f64"
    if mn.startswith("v_fma_f64") or mn.startswith("v_fmac_f64"):
        return "fp64_fma"
    if mn.startswith("v_") and "_f64" in mn and not mn.startswith("v_cvt"):
        return "fp64"
    if mn.startswith("v_pk_fma_f32"):
        return "packed_f32_fma"
    if mn.startswith("v_pk_") and "_f32" in mn:
        return "packed_f32"
    if mn.startswith("v_fma_f32") or mn.startswith("v_fmac_f32"):
        return "f32_fma"
    if mn.startswith("v_") and mn.split("_e")[0].endswith("_f32") and not mn.startswith("v_cvt") and not mn.startswith("v_rcp"):
        return "f32"
    if mn.startswith("v_rcp_f32") or mn.startswith("v_rcp_iflag"):
        return "rcp_f32"
    if mn.startswith("v_"):
        return "valu_other"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith("scratch_") or mn.startswith("buffer_") or mn.startswith("global_") or mn.startswith("flat_"):
        return "vmem"
    return "scalar"


def march_loop(insns, rcp_class="rcp_f64"):
    """The step loop of the march: the innermost backward-branch region that reads the LDS node table (ds_read), holds
    reciprocals and contains no workgroup barrier (the enclosing chunk loop does; the loops of the Bessel code and the
    LDS staging read no LDS).  None if the kernel has none."""
    index = {a: i for i, (a, _, _) in enumerate(insns)}
    regions = []
    for i, (a, mn, ops) in enumerate(insns):
        if not (mn.startswith("s_cbranch") or mn == "s_branch"):
            continue
        try:
            imm = int(ops.split()[0])
        except (ValueError, IndexError):
            continue
        if imm >= 0x8000:
            imm -= 0x10000
        tgt = a + 4 + 4 * imm
        if tgt > a or tgt not in index:
            continue
        mns = [m for _, m, _ in insns[index[tgt]:i + 1]]
        if any(m.startswith("ds_read") for m in mns) and any(classify(m) == rcp_class for m in mns) and \
                not any(m.startswith("s_barrier") for m in mns):
            regions.append((index[tgt], i))
    # innermost qualifying regions (one-wave workgroups have no barrier in the chunk loop either); the largest of them
    inner = [r for r in regions if not any(o != r and o[0] >= r[0] and o[1] <= r[1] for o in regions)]
    if not inner:
        return None
    # the march loop reads a whole row of node entries per step: of the candidates, the one with the most LDS reads (a loop
    # of the exterior code that touches the LDS stash once qualifies otherwise)
    lo, hi = max(inner, key=lambda r: (sum(1 for _, m, _ in insns[r[0]:r[1] + 1] if m.startswith("ds_read")), r[1] - r[0]))
    return insns[lo:hi + 1]


def conditional_block(body):
    """Addresses of the loop instructions a forward SCALAR branch of the loop can skip (uniform condition: the
    renormalisation of the fp32 march, entered on every 16th iteration).  Lane-masked regions (s_cbranch_execz) are not
    meant: their instructions issue whether or not lanes are active, unless no lane is."""
    lo, hi = body[0][0], body[-1][0]
    skipped = set()
    for a, mn, ops in body:
        if mn not in ("s_cbranch_scc0", "s_cbranch_scc1"):
            continue
        try:
            imm = int(ops.split()[0])
        except (ValueError, IndexError):
            continue
        if imm >= 0x8000:
            imm -= 0x10000
        tgt = a + 4 + 4 * imm
        if a < tgt <= hi + 4:
            skipped.update(x for x, _, _ in body if a < x < tgt)
    return skipped


def count(body, rcp_class="rcp_f64", steps_per_rcp=1):
    """steps_per_rcp: RK4 steps of a point that share one reciprocal (2 for the fp64 marches of every family but the twisted
    cylinder, fam_rcp4 in csrc/es_shoot_device.hpp: coefficients4)."""
    c = {}
    cond = conditional_block(body)
    cond_cycles = 0.0
    for a, mn, _ in body:
        k = classify(mn)
        c[k] = c.get(k, 0) + 1
        if a in cond and (k.startswith(("fp64", "packed", "f32", "valu")) or k.startswith("rcp")):
            cond_cycles += 16.0 if k.startswith("rcp") else 4.0
    n = c.get(rcp_class, 0) * steps_per_rcp
    per = {k: v / n for k, v in c.items()}
    # issue cycles per wave and point-step: fp64 / packed fp32 / other VALU 4, quarter-rate reciprocals 16
    cyc = 4.0 * sum(per.get(k, 0) for k in ("fp64", "fp64_fma", "packed_f32", "packed_f32_fma", "f32", "f32_fma", "valu_other")) + \
        16.0 * (per.get("rcp_f64", 0) + per.get("rcp_f32", 0))
    # issue cycles of the part every iteration executes (a block behind a uniform forward branch left out: bench.py
    # prices the launch with this one and says so)
    return {"loop_instructions": len(body), "point_steps_per_iteration": n, "per_point_step": per,
            "issue_cycles_per_wave_point_step": cyc, "issue_cycles_unconditional": cyc - cond_cycles / n,
            "conditional_instructions": len(cond), "scratch_in_loop": c.get("vmem", 0)}


def table():
    obj = os.path.join(ROOT, "eigensolver_amd", "lib", "es_shoot.o")
    ks = kernels(disassemble(obj))
    names = list(ks)
    pretty = codeobj_table.demangle(names)
    out = {}
    for mangled, dem in zip(names, pretty):
        short = codeobj_table.short(dem)
        if short.startswith("shoot_grid_kernel"):
            body = march_loop(ks[mangled])
            if body:
                out[short] = count(body, steps_per_rcp=1 if short.startswith("shoot_grid_kernel<1,") else 2)
        elif short.startswith("shoot_grid_f32_kernel"):
            body = march_loop(ks[mangled], "rcp_f32")
            if body:
                r = count(body, "rcp_f32")
                # the fp32 loop works on PAIRS of points: one v_rcp_f32 per point and step all the same
                out[short] = r
    return out


def main():
    t = table()
    if "--write" in sys.argv:
        json.dump(t, open(OUT, "w"), indent=1, sort_keys=True)
    for k in sorted(t):
        v = t[k]
        p = v["per_point_step"]
        print(f"{k}: {v['point_steps_per_iteration']} point-steps/iteration, per point-step fp64 {p.get('fp64', 0) + p.get('fp64_fma', 0):.2f} "
              f"(fma {p.get('fp64_fma', 0):.2f}) rcp {p.get('rcp_f64', 0) + p.get('rcp_f32', 0):.2f} packed f32 {p.get('packed_f32', 0) + p.get('packed_f32_fma', 0):.2f} "
              f"f32 {p.get('f32', 0) + p.get('f32_fma', 0):.2f} other VALU {p.get('valu_other', 0):.2f} "
              f"lds {p.get('lds', 0):.2f} scalar {p.get('scalar', 0):.2f} -> {v['issue_cycles_per_wave_point_step']:.1f} cycles "
              f"({v['issue_cycles_unconditional']:.1f} without the {v['conditional_instructions']} instructions behind a uniform branch); "
              f"scratch in loop {v['scratch_in_loop']}")


if __name__ == "__main__":
    main()
