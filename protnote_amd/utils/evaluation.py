"""Average-precision metrics for mAP parity checks (the arithmetic behind the reference's torcheval
BinaryAUPRC / MultilabelAUPRC at ProtNoteTrainer.py:477-485 and torchmetrics AveragePrecision in
utils/evaluation.py:148-169 - third-party there, restated here in numpy and cross-checked against
sklearn.metrics.average_precision_score in tests/test_evaluation.py).  Host-side, off the hot loop."""
import numpy as np


def average_precision(scores, labels) -> float:
    """AP = sum_n (R_n - R_{n-1}) P_n over the distinct score thresholds, descending (ties share a threshold)."""
    scores = np.asarray(scores, dtype=np.float64).ravel()
    labels = np.asarray(labels).ravel() > 0
    npos = int(labels.sum())
    if npos == 0:
        return float("nan")
    order = np.argsort(-scores, kind="mergesort")
    s, y = scores[order], labels[order]
    last_of_group = np.r_[s[1:] != s[:-1], True]
    tp = np.cumsum(y)[last_of_group].astype(np.float64)
    k = (np.flatnonzero(last_of_group) + 1).astype(np.float64)
    precision, recall = tp / k, tp / npos
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def map_micro(scores, labels) -> float:
    """One AP over all (sequence, label) pairs (reference: BinaryAUPRC on flattened predictions)."""
    return average_precision(scores, labels)


def map_macro(scores, labels) -> float:
    """Mean of per-label AP over labels that have at least one positive."""
    scores, labels = np.asarray(scores), np.asarray(labels)
    aps = [average_precision(scores[:, j], labels[:, j]) for j in range(scores.shape[1])]
    aps = [a for a in aps if not np.isnan(a)]
    return float(np.mean(aps)) if aps else float("nan")
