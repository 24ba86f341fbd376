"""CPU: the oracle's fp32 elementary functions (restated bit-for-bit by the HIP
kernels) against float64 numpy, and the monotonicity the row-max kernel needs."""
import numpy as np


def ulp_err(y, ref):
    ref32 = ref.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(y.astype(np.float64) - ref) / np.maximum(ulp, 1e-45)


def test_exp_accuracy(oracle_lib):
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.uniform(-87, 88, 200000), rs.uniform(-5, 5, 200000),
                        [0.0, -0.0, 1.0, -1.0, 88.72, -87.3, -100.0, 4.1351666]]).astype(np.float32)
    y = oracle_lib.vec('expf', x)
    assert ulp_err(y, np.exp(x.astype(np.float64))).max() <= 2.0
    assert oracle_lib.vec('expf', np.array([89.0, -104.0, np.inf, -np.inf], np.float32)).tolist() \
        == [np.inf, 0.0, np.inf, 0.0]


def test_log_accuracy(oracle_lib):
    rs = np.random.RandomState(1)
    x = np.concatenate([np.exp(rs.uniform(-80, 80, 200000)), rs.uniform(0.5, 2.0, 200000),
                        [1.0, 2.0, 0.5, 1.17549435e-38, 1e-40]]).astype(np.float32)
    y = oracle_lib.vec('logf', x)
    ref = np.log(x.astype(np.float64))
    err = np.abs(y.astype(np.float64) - ref)
    assert (err <= 2.0 * np.spacing(np.abs(ref.astype(np.float32))).astype(np.float64) + 1e-7).all()


def test_sigmoid_accuracy(oracle_lib):
    rs = np.random.RandomState(2)
    x = (rs.standard_normal(300000) * 6).astype(np.float32)
    y = oracle_lib.vec('sigmoidf', x)
    ref = 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    assert ulp_err(y, ref).max() <= 3.0


def test_sqrt_sigmoid_is_monotone_exhaustive(oracle_lib):
    """every adjacent fp32 pair in [-110, 100]: k_rowmax reduces the class logits and applies
    sqrt(sigmoid()) once, which equals max of the scores only if this holds."""
    assert oracle_lib.sigmoid_nonmonotone_count(-110.0, 100.0) == 0
