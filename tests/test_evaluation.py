import numpy as np
from sklearn.metrics import average_precision_score

from protnote_amd.utils.evaluation import average_precision, map_macro, map_micro


def test_average_precision_matches_sklearn():
    rng = np.random.RandomState(0)
    for n, p, ties in ((50, 0.3, False), (2000, 0.02, False), (500, 0.2, True)):
        s = rng.randn(n)
        if ties:
            s = np.round(s, 1)
        y = rng.rand(n) < p
        y[0] = True
        np.testing.assert_allclose(average_precision(s, y), average_precision_score(y, s), rtol=1e-12)
    S, Y = rng.randn(40, 7), rng.rand(40, 7) < 0.3
    Y[:, 3] = False  # a label without positives is skipped by the macro mean
    keep = [j for j in range(7) if Y[:, j].any()]
    np.testing.assert_allclose(map_macro(S, Y), np.mean([average_precision_score(Y[:, j], S[:, j]) for j in keep]))
    np.testing.assert_allclose(map_micro(S, Y), average_precision_score(Y.ravel(), S.ravel()))
