"""CPU suite: the two independent restatements (torch-fp64 composition vs brute-force C loops)
must agree on every op of the hot path.  This is what stands in for reference golden vectors
(parity unpinned: the reference has none)."""
import pytest

import op_cases as C

TOL = 2e-6   # the C oracle stores float32; the torch side is float64


@pytest.mark.parametrize('shape', C.CONV_SHAPES_SMALL, ids=str)
def test_conv(oracle_abi, shape):
    C.case_conv(oracle_abi, shape, TOL)


def test_deconv_pitched(oracle_abi):
    C.case_deconv_pitched(oracle_abi, TOL)


def test_conv_pitched(oracle_abi):
    C.case_conv_pitched(oracle_abi, TOL)


@pytest.mark.parametrize('shape', C.DECONV_SHAPES_SMALL, ids=str)
def test_deconv(oracle_abi, shape):
    C.case_deconv(oracle_abi, shape, TOL)


@pytest.mark.parametrize('shape', C.BN_SHAPES, ids=str)
def test_bn(oracle_abi, shape):
    C.case_bn(oracle_abi, shape, TOL)


def test_bn_large_mean(oracle_abi):
    C.case_bn_large_mean(oracle_abi, 1e-4)


def test_bias(oracle_abi):
    C.case_bias(oracle_abi, TOL)


@pytest.mark.parametrize('shape', C.DNA_SHAPES, ids=str)
def test_dna(oracle_abi, shape):
    C.case_dna(oracle_abi, shape, TOL)


@pytest.mark.parametrize('shape', [(2, 9, 7, 3, 4, 5), (1, 16, 16, 1, 1, 3), (2, 20, 17, 3, 10, 5), (1, 8, 8, 4, 3, 7)])
def test_cdna(oracle_abi, shape):
    C.case_cdna(oracle_abi, shape, TOL)


@pytest.mark.parametrize('shape,act,groups', [((4, 6, 5, 8), 'relu', 1), ((6, 3, 3, 12), 'lrelu', 2), ((2, 5, 7, 3), None, 1)])
def test_sync_bn_entries(oracle_abi, shape, act, groups):
    C.case_sync_bn_entries(oracle_abi, shape, act, groups, TOL)


def test_bwd_pair(oracle_abi):
    C.case_bwd_pair(oracle_abi, TOL, exact=False)


def test_wgrad_deferred_reduction(oracle_abi):
    C.case_wgrad_deferred(oracle_abi, TOL, exact=False)


def test_copy_many(oracle_abi):
    C.case_copy_many(oracle_abi)


@pytest.mark.parametrize('shape', [C.DNA_SHAPES[i] for i in (0, 1, 3, 4, 8)], ids=str)
def test_dna_bias(oracle_abi, shape):
    """The C restatement of softmax(logits + bias) and its dbias against the torch restatement (autograd)."""
    C.case_dna_bias(oracle_abi, shape, TOL)


def test_dna_extreme(oracle_abi):
    C.case_dna_extreme_logits(oracle_abi, TOL)


def test_plumbing(oracle_abi):
    C.case_plumbing(oracle_abi, TOL)


def test_losses(oracle_abi):
    C.case_losses(oracle_abi, TOL)


def test_optimizers(oracle_abi):
    C.case_optimizers(oracle_abi, TOL)
