"""Solver arguments the plugin accepts for compatibility with the DIFFCP plugin (diffcp_if.py:356-367) but does not act on are SAID, once per
process and topic, not swallowed: an EXPLICIT acceleration_lookback > 1 (one-pair history), mode other than lsqr / dense, solve_method, n_jobs_*.
The default configuration stays silent (a call that is valid for the reference must survive `-W error`)."""
import warnings

import pytest


def _fresh():
    from cvxpylayers_amd.interfaces import mi355_if
    mi355_if._WARNED.clear()
    return mi355_if


def test_lookback_beyond_one_pair_is_said_once_and_only_when_it_was_asked_for():
    m = _fresh()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.note_ignored_args({"acceleration_lookback": 10}, explicit_lookback=False)          # SCS's default, forwarded: silent
        assert not w
        m.note_ignored_args({"acceleration_lookback": 10}, explicit_lookback=True)
        m.note_ignored_args({"acceleration_lookback": 5}, explicit_lookback=True)
    assert len(w) == 1 and "ONE-pair" in str(w[0].message) and "SCS's default" not in str(w[0].message)


@pytest.mark.parametrize("lb", [0, 1])
def test_lookback_zero_and_one_are_exactly_what_runs(lb):
    m = _fresh()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.note_ignored_args({"acceleration_lookback": lb}, explicit_lookback=True)
    assert not w


def test_mode_lsqr_lsmr_and_dense_are_acted_on_other_modes_and_n_jobs_are_reported():
    m = _fresh()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.note_ignored_args({"acceleration_lookback": 0, "mode": "lsqr", "n_jobs_forward": 4}, explicit_lookback=True)          # mode="lsqr": diffcp's LSQR adjoint (ce_vjp_lsqr)
        m.note_ignored_args({"acceleration_lookback": 0, "mode": "dense", "n_jobs_backward": -1}, explicit_lookback=True)       # mode="dense": the direct elimination
        m.note_ignored_args({"acceleration_lookback": 0, "mode": "lsmr"}, explicit_lookback=True)                               # mode="lsmr": the iterative adjoint under LSMR's recurrences (ce_set_lsqr_variant)
        m.note_ignored_args({"acceleration_lookback": 0, "mode": "bicg"}, explicit_lookback=True)
    msgs = [str(x.message) for x in w]
    assert len(msgs) == 3 and any("'mode'" in t and "bicg" in t for t in msgs) and any("n_jobs_forward" in t for t in msgs) and any("n_jobs_backward" in t for t in msgs)
    assert m.adjoint_mode({"mode": "lsmr"}) == "lsqr" and m.lsqr_rule({"mode": "lsmr"}, 3, 4)[4] == "lsmr" and m.lsqr_rule({"mode": "lsqr"}, 3, 4)[4] == "lsqr"
    assert m.unpack_rule((1e-9, 1e-9, 7), 3, 4) == (1e-9, 1e-9, 7, "full", "lsqr") and m.unpack_rule((1e-9, 1e-9, 7, "reduced"), 3, 4)[3:] == ("reduced", "lsqr") and m.unpack_rule(None, 3, 4)[2] == 16
    assert m.adjoint_mode({"mode": "lsqr"}) == "lsqr" and m.adjoint_mode({"mode": "dense"}) == "dense" and m.adjoint_mode({}) == "direct"      # (default: elimination + device-side LSQR re-solve of rank-deficient instances; "dense": the elimination alone)
    # and they still pass validation (unknown names do not)
    m.make_settings({"mode": "lsqr", "n_jobs_forward": 4, "eps": 1e-6})
    with pytest.raises(ValueError):
        m.make_settings({"lookback": 3})


def test_status_counts_are_recomputed_when_the_device_summary_never_arrives():
    """ensure_summary (mi355_if.ConeEngine): a ready flag still clear AFTER the stream was drained means the device's stores did not reach the pinned buffer;
    the counts then come from the status vector itself (min, #inaccurate, #flagged) and the engine stops spin-polling -- said once."""
    import numpy as np
    import torch
    m = _fresh()
    eng = object.__new__(m.ConeEngine)
    eng._summary_np = np.zeros((3, 4), dtype=np.int32)
    eng._summary_vec = [torch.tensor([1, 2, 1, -2, 2], dtype=torch.int32), None, torch.tensor([0, 4, 1, 0], dtype=torch.int32)]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        eng.ensure_summary(0)
        eng.ensure_summary(2)
        eng.ensure_summary(1)            # nothing enqueued in that slot: left alone
    assert eng._summary_np[0].tolist() == [-2, 2, 5, 1]          # min status, two "solved, inaccurate", five with (v & 3) != 0
    assert eng._summary_np[2].tolist() == [0, 0, 1, 1]           # adjoint flags: bit 2 (rank-deficient, 4) is not a failure, bits 0-1 are
    assert eng._summary_np[1].tolist() == [0, 0, 0, 0]
    assert eng._summary_no_spin and len(w) == 1 and "did not arrive" in str(w[0].message)
    eng._summary_np[0] = [1, 0, 0, 1]                            # a slot whose flag IS set is never touched
    eng.ensure_summary(0)
    assert eng._summary_np[0].tolist() == [1, 0, 0, 1]


def test_constant_A_is_decided_structurally_from_the_parameter_map():
    """MI355_ctx(reduced_A_mat=...) (the keyword MOREAU_ctx takes; /root/reference moreau_if.py:234-256): the A part of the value vector is batch-invariant
    exactly when no A entry has a parameter column in the map -- decided once at construction, no per-call compare of B x nnzA values and no host sync.
    b may depend on parameters (unlike Moreau's PA_is_constant, which also wants b constant).  Without the map the answer stays None (decided per call)."""
    import numpy as np
    import scipy.sparse as sp
    m = _fresh()
    # 2 x 2 template, CSC of [A | b]: column 0 {rows 0, 1}, column 1 {row 1}, b column {rows 0, 1}  ->  nnz_aug = 5, nnzA = 3
    structure = (np.array([0, 1, 1, 0, 1]), np.array([0, 2, 3, 5]), (2, 3))
    P = 2          # two parameters + the constant column
    only_b = sp.csr_array((np.ones(5), (np.array([0, 1, 2, 3, 4]), np.array([2, 2, 2, 0, 1]))), shape=(5, P + 1))        # A entries constant, b depends on the parameters
    a_too = sp.csr_array((np.ones(5), (np.array([0, 1, 2, 3, 4]), np.array([2, 0, 2, 0, 1]))), shape=(5, P + 1))         # A entry 1 depends on parameter 0
    assert m.MI355_ctx(None, structure, {"z": 0, "l": 2, "q": []}).A_is_constant is None
    assert m.MI355_ctx(None, structure, {"z": 0, "l": 2, "q": []}, reduced_A_mat=only_b).A_is_constant is True
    assert m.MI355_ctx(None, structure, {"z": 0, "l": 2, "q": []}, reduced_A_mat=a_too).A_is_constant is False
