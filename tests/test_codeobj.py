"""CPU-side checks of the shipped gfx950 code objects (no GPU needed: hipcc cross-compiles, the metadata and the
disassembly are read with the ROCm LLVM tools): the launch shapes the per-family tables select have no scratch access
inside their march loop, and profiles/isa_loop_counts.json -- the instruction counts bench.py prices a launch with --
is the one of the library as built."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="ROCm LLVM tools not present")


@pytest.fixture(scope="module")
def built():
    from eigensolver_amd import build
    build.build()
    import codeobj_table
    import isa_loop_count
    return codeobj_table.table("shoot_grid_kernel|shoot_grid_f32_kernel|refine"), isa_loop_count.table()


def test_no_spill_inside_a_march_loop(built):
    rows, loops = built
    grid = [r for r in rows if r["kernel"].startswith("shoot_grid_kernel")]
    assert len(grid) >= 12
    for name, v in loops.items():
        if name.startswith("shoot_grid_kernel"):
            assert v["scratch_in_loop"] == 0, (name, v["scratch_in_loop"])
            assert v["point_steps_per_iteration"] in (1, 2, 3, 4, 6, 8, 12, 16), (name, v["point_steps_per_iteration"])
    # whole-kernel spill counts of the fp64 grid shapes: none since the kernels process one tile per workgroup (the tile
    # loop of rounds 2 - 3 carried hoisted tile-invariant values through the march: 168 registers + 6 spilled values for
    # the headline shape, 123 and none without it)
    for r in grid:
        assert r.get(".vgpr_spill_count", 0) == 0, r
    f32 = {r["kernel"]: r for r in rows if r["kernel"].startswith("shoot_grid_f32_kernel")}
    for name in ("shoot_grid_f32_kernel<1,4,256,true,2>", "shoot_grid_f32_kernel<0,4,256,false,3>",
                 "shoot_grid_f32_kernel<2,4,256,false,3>", "shoot_grid_f32_kernel<3,4,256,false,3>"):
        assert f32[name].get(".vgpr_spill_count", 0) == 0, f32[name]


def test_isa_loop_counts_file_is_current(built):
    _, loops = built
    committed = json.load(open(os.path.join(ROOT, "profiles", "isa_loop_counts.json")))
    if os.environ.get("ES_BUILD_ALL_SHAPES") == "1":
        pytest.skip("measuring build")
    assert set(committed) == set(loops), sorted(set(committed) ^ set(loops))
    for k in loops:
        assert committed[k]["per_point_step"] == pytest.approx(loops[k]["per_point_step"]), k
        assert committed[k]["issue_cycles_per_wave_point_step"] == pytest.approx(loops[k]["issue_cycles_per_wave_point_step"]), k
    # the headline loop: 34 fp64 instructions + HALF a v_rcp_f64 per point and RK4 step (two steps share a division)
    h = loops["shoot_grid_kernel<0,4,256,false,3>"]["per_point_step"]
    assert h["fp64"] + h["fp64_fma"] == pytest.approx(34.0) and h["rcp_f64"] == pytest.approx(0.5)
