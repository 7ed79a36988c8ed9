import numpy as np
from sklearn.metrics import average_precision_score



def test_macro_mean_conventions():
    """A label without positives counts as AP 0 in the macro mean by default (torcheval's MultilabelAUPRC convention,
    parity unpinned - see oracle/metrics_oracle.py::macro_mean) or is skipped with empty_label_ap=None."""
    from oracle import metrics_oracle as MO

    rng = np.random.RandomState(0)
    S, Y = rng.randn(40, 7).astype(np.float32), rng.rand(40, 7) < 0.3
    Y[:, 3] = False
    per = [MO.average_precision_fast(S[:, j], Y[:, j]) for j in range(7)]
    keep = [j for j in range(7) if Y[:, j].any()]
    sk = [average_precision_score(Y[:, j], S[:, j]) for j in keep]
    np.testing.assert_allclose(MO.macro_mean(per, None), np.mean(sk), rtol=1e-12)
    np.testing.assert_allclose(MO.macro_mean(per), np.sum(sk) / 7, rtol=1e-12)
    np.testing.assert_allclose(MO.macro_mean(per, 0.5), (np.sum(sk) + 0.5) / 7, rtol=1e-12)


def test_metrics_oracle_matches_sklearn():
    """The oracle the device kernels are checked against is itself pinned to sklearn's AP (same published definition)."""
    from oracle import metrics_oracle as MO

    rng = np.random.RandomState(1)
    for n, p, ties in ((30, 0.3, False), (400, 0.05, False), (300, 0.2, True), (5, 1.0, False)):
        s = rng.rand(n).astype(np.float32)
        if ties:
            s = np.round(s, 1).astype(np.float32)
        y = rng.rand(n) < p
        y[0] = True
        want = average_precision_score(y, s)
        np.testing.assert_allclose(MO.average_precision(s, y), want, rtol=1e-12)
        np.testing.assert_allclose(MO.average_precision_fast(s, y), want, rtol=1e-12)
    assert np.isnan(MO.average_precision(np.ones(4), np.zeros(4)))


def test_binned_auprc_oracle_properties():
    """Binned estimate: perfect separation -> 1; with as many thresholds as distinct scores (placed on them) the
    lower-threshold Riemann sum brackets the exact AP from the definition's other side and converges to it."""
    from oracle import metrics_oracle as MO

    thr = np.linspace(0, 1, 50, dtype=np.float32)
    s = np.r_[np.full(10, 0.9), np.full(30, 0.1)].astype(np.float32)
    y = np.r_[np.ones(10), np.zeros(30)]
    np.testing.assert_allclose(MO.binned_auprc(s, y, thr), 1.0)
    rng = np.random.RandomState(2)
    s = rng.rand(5000).astype(np.float32)
    y = rng.rand(5000) < s  # informative scores
    exact = MO.average_precision_fast(s, y)
    assert abs(MO.binned_auprc(s, y, thr) - exact) < 0.02
    assert abs(MO.binned_auprc(s, y, np.linspace(0, 1, 2000, dtype=np.float32)) - exact) < 2e-3
    assert np.isnan(MO.binned_auprc(s, np.zeros(5000), thr))


def test_device_metrics_refuse_cpu():
    import pytest
    import torch
    from protnote_amd.utils.evaluation import DeviceAveragePrecision, DeviceBinnedAUPRC

    with pytest.raises(RuntimeError, match="HIP device"):
        DeviceAveragePrecision(4, 8, torch.device("cpu"))
    with pytest.raises(RuntimeError, match="HIP device"):
        DeviceBinnedAUPRC(4, "cpu")
