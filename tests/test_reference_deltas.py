"""Consumes tests/golden/reference_deltas.json -- written by tools/pin_reference.jl on a machine with Julia + GPCC.jl, the only
route from "parity unpinned" to pinned.  Skipped while that file does not exist (this image has no Julia)."""
import json
import os

import pytest

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_deltas.json")


@pytest.mark.skipif(not os.path.exists(PATH), reason="tools/pin_reference.jl has not been run (no Julia in the build image): parity unpinned")
def test_fixtures_agree_with_the_real_reference():
    with open(PATH) as f:
        d = json.load(f)
    assert len(d["cases"]) >= 40
    for c in d["cases"]:
        assert c["rel"] <= 1e-10, c
    for c in d["covariances"]:
        assert c["rel_Kxy"] <= 1e-13 and c["rel_Kxx"] <= 1e-13, c
    assert d["probabilities"]["rel_flat"] <= 1e-12 and d["probabilities"]["rel_prior"] <= 1e-12
    assert d["nonpd_throws_PosDefException"] is True
    # the restated MiscUtil transforms (gpcc_fit.h): report, and fail loudly if the reading was wrong
    assert d["miscutil"]["makepositive_minus_softplus"] <= 1e-12, "makepositive is not softplus: fix gpcc_fit.h / fit.py"
    assert d["miscutil"]["transformbetween_minus_logistic"] <= 1e-12, "transformbetween is not the logistic map"


def test_pin_script_is_committed_and_names_the_reference_functions():
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pin_reference.jl")).read()
    for needle in ("GPCC.delayedCovariance", "MvNormal", "getprobabilities", "makematrixsymmetric!", "makepositive",
                   "transformbetween", "nearestposdef", "gpcc(", "reference_deltas.json"):
        assert needle in src, needle
