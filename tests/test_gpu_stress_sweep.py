"""Randomised engine-vs-oracle sweep (tests/stress_kit.py): 48 random template shapes across every kernel variant (register-tiled k_fwd2 variants, 512-thread
variants, the size-generic kernels), sparse patterns, templates with a duplicated equality row (every instance rank deficient: flagged by the elimination and
re-solved by the device-side LSQR).  Forward solutions against the oracle at eps 1e-9, the plugin's default adjoint against the oracle's dense elimination /
LSQR mode.  scripts/stress_sweep.py runs the same sweep for more shapes and seeds (round 6: 300 shapes, no failure)."""
import pytest

pytestmark = pytest.mark.gpu


def test_random_templates_forward_and_default_adjoint_against_the_oracle():
    from stress_kit import sweep
    fails, notes, lines = sweep(48, 11, 12, verbose=False)
    assert not fails, "\n".join(l for l in lines if l.startswith("FAIL"))


def test_random_templates_with_psd_exponential_and_power_cones():
    """the same sweep with PSD blocks, exponential and power triples in the mix (pivoting adjoint kernels + LSQR re-solve)"""
    from stress_kit import sweep
    fails, notes, lines = sweep(36, 5, 10, verbose=False, ext=True)
    assert not fails, "\n".join(l for l in lines if l.startswith("FAIL"))


def test_random_shared_A_templates(monkeypatch):
    """shared-A templates in the shape canonicalisation produces (dense rows + bounds + cone blocks of single-entry rows): the persistent kernels k_sa_fwd /
    k_sa_lsqr and the batch-GEMM fallback against the oracle (forward at eps 1e-9, diffcp's LSQR adjoint on the full system under a tight rule)"""
    monkeypatch.setenv("CE_CONST_A", "1")
    from stress_kit import sweep_shared
    fails, notes, lines = sweep_shared(30, 3, 10, verbose=False)
    assert not fails, "\n".join(l for l in lines if l.startswith("FAIL"))
