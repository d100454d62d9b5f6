"""Oracle pinned against the fixtures produced by the reference's actual module files (oracle/make_golden.py)."""
import pytest
import torch

from conftest import assert_close_with_nonfinite, golden_names, load_golden
from oracle import functional as F

TOL = dict(atol=1e-6, rtol=1e-6)   # same op sequence as the reference -> expected bit-exact; slack for BLAS variation


@pytest.mark.parametrize("name", golden_names("dcrnn_"))
def test_dcrnn_cell_matches_reference_fixture(name):
    g = load_golden(name)
    X, H0, ei, ew = g["in"]["X"], g["in"]["H0"], g["in"]["edge_index"], g["in"]["edge_weight"]
    p = g["param"]
    assert_close_with_nonfinite(F.dcrnn_cell(X, ei, None, None, p), g["out"]["H_noweight"], what="no weight", **TOL)
    assert_close_with_nonfinite(F.dcrnn_cell(X, ei, ew, None, p), g["out"]["H_weight"], what="weight", **TOL)
    assert_close_with_nonfinite(F.dcrnn_cell(X, ei, ew, H0, p), g["out"]["H_weight_hidden"], what="hidden", **TOL)


def test_ws_fixture_really_contains_nonfinite_values():
    g = load_golden("dcrnn_ws_directed_K3")
    assert not torch.isfinite(g["out"]["H_weight"]).all(), "the reference's mock graph must yield inf/nan (Appendix B.4)"


def test_dconv_dense_and_scatter_forms():
    g = load_golden("dconv_sensor_asym_K3")
    X, ei, ew = g["in"]["X"], g["in"]["edge_index"], g["in"]["edge_weight"]
    W, b = g["param"]["weight"], g["param"]["bias"]
    assert_close_with_nonfinite(F.dconv(X, ei, ew, W, b), g["out"]["H"], what="DConv", **TOL)
    assert_close_with_nonfinite(F.batched_dconv(X, ei, ew, W, b), g["out"]["H_batched"], what="BatchedDConv", **TOL)
    # unique edges, non-zero weights -> both forms are the same operator
    assert_close_with_nonfinite(g["out"]["H"], g["out"]["H_batched"], atol=1e-5, rtol=1e-5, what="dense vs scatter")


@pytest.mark.parametrize("name", golden_names("batched_dcrnn_"))
def test_batched_dcrnn_matches_reference_fixture(name):
    g = load_golden(name)
    out = F.batched_dcrnn(g["in"]["X"], g["in"]["edge_index"], g["in"]["edge_weight"], g["param"])
    assert_close_with_nonfinite(out, g["out"]["out"], what=name, **TOL)


def test_tgcn_matches_reference_fixture():
    g = load_golden("tgcn_sensor")
    X, H0, ei, ew, p = g["in"]["X"], g["in"]["H0"], g["in"]["edge_index"], g["in"]["edge_weight"], g["param"]
    assert_close_with_nonfinite(F.tgcn_cell(X, ei, None, None, p), g["out"]["H_noweight"], **TOL)
    assert_close_with_nonfinite(F.tgcn_cell(X, ei, ew, None, p), g["out"]["H_weight"], **TOL)
    assert_close_with_nonfinite(F.tgcn_cell(X, ei, ew, H0, p), g["out"]["H_weight_hidden"], **TOL)
    assert_close_with_nonfinite(F.tgcn_cell(X, ei, ew, H0, p, improved=True), g["out"]["H_improved"], **TOL)


def test_tgcn2_matches_reference_fixture():
    g = load_golden("tgcn2_sensor")
    X, H0, ei, ew, p = g["in"]["X"], g["in"]["H0"], g["in"]["edge_index"], g["in"]["edge_weight"], g["param"]
    assert_close_with_nonfinite(F.tgcn_cell(X, ei, ew, None, p), g["out"]["H_weight"], **TOL)
    assert_close_with_nonfinite(F.tgcn_cell(X, ei, ew, H0, p), g["out"]["H_weight_hidden"], **TOL)


@pytest.mark.parametrize("name", ["a3tgcn_sensor", "a3tgcn2_sensor"])
def test_a3tgcn_matches_reference_fixture(name):
    g = load_golden(name)
    X, H0, ei, ew, p = g["in"]["X"], g["in"]["H0"], g["in"]["edge_index"], g["in"]["edge_weight"], g["param"]
    assert_close_with_nonfinite(F.a3tgcn(X, ei, ew, None, p), g["out"]["H_weight"], **TOL)
    assert_close_with_nonfinite(F.a3tgcn(X, ei, ew, H0, p), g["out"]["H_weight_hidden"], **TOL)


def test_chebconvattention_matches_reference_fixture():
    g = load_golden("chebconvattention_sensor")
    X, S, ei, ew = g["in"]["X"], g["in"]["S"], g["in"]["edge_index"], g["in"]["edge_weight"]
    W, b = g["param"]["_weight"], g["param"]["_bias"]
    for norm, lam in (("sym", None), ("rw", float(g["meta"]["lambda_rw"])), (None, float(g["meta"]["lambda_none"]))):
        out = F.cheb_conv_attention(X, ei, S, ew, W, b, norm, lam)
        assert_close_with_nonfinite(out, g["out"]["out_" + str(norm)], what=str(norm), **TOL)
        out = F.cheb_conv_attention(X, ei, S, None, W, b, norm, lam)
        assert_close_with_nonfinite(out, g["out"]["out_noweight_" + str(norm)], what=f"no weight {norm}", **TOL)
