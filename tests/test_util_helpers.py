"""The parity helpers themselves: none of them may accept a NaN/Inf where the reference is finite
(round-1 VERDICT, weak #1: `err > tol` is False for NaN, so a kernel returning NaN passed)."""
import pytest
import torch

from _util import (assert_close, assert_close_outliers, assert_close_scaled, assert_sum_close)

BAD = [float('nan'), float('inf'), float('-inf')]


def _pair():
    ref = torch.linspace(-2.0, 2.0, 64).reshape(8, 8)
    return ref.clone(), ref


@pytest.mark.parametrize('bad', BAD)
def test_assert_close_rejects_nonfinite(bad):
    got, ref = _pair()
    assert_close(got, ref)
    got[3, 4] = bad
    with pytest.raises(AssertionError):
        assert_close(got, ref)


@pytest.mark.parametrize('bad', BAD)
def test_assert_sum_close_rejects_nonfinite(bad):
    got, ref = _pair()
    assert_sum_close(got, ref, ref.double())
    got[0, 0] = bad
    with pytest.raises(AssertionError):
        assert_sum_close(got, ref, ref.double())
    with pytest.raises(AssertionError):  # also with the condition-aware bound
        assert_sum_close(got, ref, ref.double(), abs_sum=ref.abs() * 100)


@pytest.mark.parametrize('bad', BAD)
def test_assert_close_scaled_rejects_nonfinite(bad):
    got, ref = _pair()
    assert_close_scaled(got, ref)
    got[7, 7] = bad
    with pytest.raises(AssertionError):
        assert_close_scaled(got, ref)


@pytest.mark.parametrize('bad', BAD)
def test_assert_close_outliers_rejects_nonfinite(bad):
    ref = torch.linspace(-2.0, 2.0, 4096).reshape(64, 64)
    got = ref.clone()
    assert_close_outliers(got, ref)
    got[5, 5] = bad  # ONE element of 4096 is below the outlier fraction, and must still fail
    with pytest.raises(AssertionError):
        assert_close_outliers(got, ref)


def test_nonfinite_reference_must_be_reproduced():
    ref = torch.tensor([1.0, float('inf'), float('-inf'), float('nan')])
    assert_close(ref.clone(), ref)
    for i, v in ((1, 1e30), (1, float('-inf')), (2, float('inf')), (3, 0.0)):
        got = ref.clone()
        got[i] = v
        with pytest.raises(AssertionError):
            assert_close(got, ref)


def test_finite_mismatch_still_fails_and_integers_are_exact():
    got, ref = _pair()
    got[1, 1] += 1e-3
    with pytest.raises(AssertionError):
        assert_close(got, ref)
    a = torch.arange(10)
    assert_close(a.clone(), a)
    b = a.clone()
    b[3] = 7
    with pytest.raises(AssertionError):
        assert_close(b, a)
