"""CPU: oracle/fv_oracle.py's restatement of the non-orthogonal correction (correctedSnGrad, the corrected
gaussLaplacianScheme with scalar / symmTensor / tensor diffusivity), gaussDivScheme and the patch halves of
interpolate / gaussGrad is PINNED bit for bit against the reference's own classes (tests/golden/fvnonorth_*.npz from
oracle/_ref/fv_driver mode nonorth: jittered hex boxes, one with a cyclic pair, and a prism mesh)."""
import os
import sys

import numpy as np
import pytest

import fv_oracle as fo
import nonorth_common as nc


@pytest.mark.parametrize("name", nc.CASES)
def test_nonorth_chain_oracle_matches_reference(name):
    assert nc.run_chain(name, nc.OracleBackend(fo)) == []


def test_nonorth_fixture_is_live_when_the_reference_build_exists():
    """with oracle/_ref/fv_driver present (this container) one fixture is regenerated and must be identical"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import fv_case
    import make_fv_golden
    if not fv_case.driver_available():
        pytest.skip("oracle/_ref/fv_driver not built (oracle/build_ref_fv.sh)")
    name = "fvnonorth_box_5x4x6_cyclic"
    live = make_fv_golden.generate_nonorth(name)
    g = dict(np.load(os.path.join(nc.GOLDEN, name + ".npz")))
    assert sorted(live) == sorted(g)
    for k in g:
        assert np.array_equal(np.asarray(live[k]), g[k]), k


def test_mesh_is_really_non_orthogonal():
    for name in nc.CASES:
        g, _ = nc.load(name)
        assert np.abs(g["nonOrthCorrectionVectors"]).max() > 0.1
        assert np.abs(g["ref_snGradCorrection_T"]).max() > 0.1
