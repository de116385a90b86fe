"""Oracle (C port) against the roots the reference authors stored with their code: known-answer test of the whole
physics chain (equilibrium, coefficient set, exterior, axis condition, mismatch) for every geometry family, over all
90 stored files (86 pinned, 4 listed in stored_sets.UNPINNED with the reason)."""

import numpy as np
import pytest

from tests import cases, stored_sets as S

# every family of stored files (prefix of the tag[:suffix]); sausage_fast: the cutoff is smeared, see below
FAMILIES = ["slab_density_coronal", "slab_density_photospheric", "slab_flow_coronal", "cyl_density_coronal",
            "cyl_density_photospheric", "cyl_flow_coronal", "cyl_rot:fund_kink", "cyl_rot:sausage_slow",
            "cyl_rot:slow_kink", "cyl_rot:sausage_fast"]
MAX_DROP = 0.05          # a stored set may lose at most this much of its committed accepted fraction


def test_every_reference_pickle_is_a_fixture():
    assert len(S.TAGS) == 90 and len(S.PINNED) == 86
    assert set(S.FLOORS) == set(S.PINNED)
    for tag in S.TAGS:
        S.describe(tag)                                          # parameters can be inferred for every file
        assert all(len(w) == len(k) for _, w, k in S.pairs(tag))


@pytest.mark.parametrize("tag", S.PINNED)
def test_port_accepts_stored_roots(tag):
    eq, tol = S.describe(tag)
    for mode, w, k in S.pairs(tag):
        if len(w) == 0:
            continue
        D, rel, st = cases.port_problem(eq, mode).eval_points(k, w, nthreads=8)
        frac = float(np.mean(rel < tol))
        committed = S.floor_of(tag, mode)             # the measured fraction at the time of the commit
        assert frac >= committed - MAX_DROP, (tag, mode, frac, committed)


def test_pooled_acceptance():
    """Over all pinned files at least 85 % of the 21 384 stored roots satisfy their worker's acceptance test under
    the oracle (tolerances as stored_sets.describe() infers them from the cutoff of each family's measures); the
    remainder are continuum-band points the reference integrates through and the tails described in stored_sets."""
    tot = acc = 0
    for tag in S.PINNED:
        eq, tol = S.describe(tag)
        for mode, w, k in S.pairs(tag):
            if len(w):
                D, rel, st = cases.port_problem(eq, mode).eval_points(k, w, nthreads=8)
                tot += len(w)
                acc += int(np.sum(rel < tol))
    assert tot == 21384 and acc / tot >= 0.85, (tot, acc)


@pytest.mark.parametrize("family", FAMILIES)
def test_acceptance_measure_cutoff(family):
    """The reference accepted a point iff ITS measure was below tol, so over many stored roots the oracle's measure
    must fill the band just below tol and be (nearly) empty just above it."""
    fam, _, kind = family.partition(":")
    ratios = []
    for tag in S.PINNED:
        if not tag.startswith(fam) or (kind and not tag.endswith(kind)):
            continue
        eq, tol = S.describe(tag)
        for mode, w, k in S.pairs(tag):
            if len(w):
                D, rel, st = cases.port_problem(eq, mode).eval_points(k, w, nthreads=8)
                ratios.append(rel[(st == 0) & np.isfinite(rel)] / tol)
    r = np.concatenate(ratios)
    below = int(np.sum((r >= 0.75) & (r < 1.0)))
    above = int(np.sum((r >= 1.0) & (r < 1.25)))
    print(f"{family}: {len(r)} stored roots, {below} with measure in [0.75, 1) tol, {above} in [1, 1.25) tol")
    if kind == "sausage_fast":
        # grid points of a 40-per-band main grid accepted at tol = 1.5 %, the size of the error of the reference's own
        # LSODA exterior (tests/stored_sets.py): its measure scatters around ours by about a tolerance, the cutoff is
        # smeared -- the step is still there (fewer above than below), the bulk is below tol
        # smeared -- no step at tol, a smooth decay (153, 174, 141, 124, 89 ... roots per 0.25 % bin around 1.5 %): asserted
        # is only that the bulk lies below tol and the density decays across it
        assert below >= 100 and above <= 0.9 * below, (family, below, above)
        assert np.mean(r < 1.0) >= 0.45, (family, float(np.mean(r < 1.0)))
        return
    assert below >= 100, (family, below)
    # photospheric density cylinder: exterior started at [1e-8, 1e-8] with a small m_e -- the reference's LSODA exterior
    # error smears the step (404 -> 76); every other family drops by more than 10x across tol
    assert above <= (0.25 if family == "cyl_density_photospheric" else 0.10) * below, (family, below, above)


def test_unpinned_files_are_listed(capsys):
    """The 4 stored files no parameter set reproduces: reported with the reason, never silently dropped."""
    assert len(S.UNPINNED) == 4
    for tag, why in S.UNPINNED.items():
        assert tag in S.TAGS
        eq, tol = S.describe(tag)
        fr = []
        for mode, w, k in S.pairs(tag):
            if len(w):
                D, rel, st = cases.port_problem(eq, mode).eval_points(k, w, nthreads=8)
                fr.append((mode, len(w), round(float(np.mean(rel < tol)), 2)))
        print(f"UNPINNED {tag}: {fr} -- {why}")
