"""CPU oracle for the evaluation metrics (TEST INFRASTRUCTURE ONLY - imported by tests/, never by the product).

Restates the definition behind the reference's mAP numbers: torcheval 0.0.7 `BinaryAUPRC` / `MultilabelAUPRC`
(ProtNoteTrainer.py:477-479), `BinaryBinnedAUPRC` / `MultilabelBinnedAUPRC(threshold=50)` (:481-485) and
torchmetrics 1.2.0 `AveragePrecision` (utils/evaluation.py:148-169).  Those packages are third-party, pinned in the
reference's setup.py / environment.yml, and absent both from /root/reference and from this image, so this restatement
is **parity unpinned** against them; it is pinned instead against sklearn.metrics.average_precision_score (same
published definition: AP = sum_n (R_n - R_{n-1}) P_n over the distinct score thresholds) in
tests/test_evaluation.py.  Plain Python/numpy loops, f64."""
import numpy as np


def average_precision(scores, labels) -> float:
    """Exact AP by definition, one threshold per distinct score, descending."""
    scores = np.asarray(scores, dtype=np.float32).ravel()
    y = np.asarray(labels).ravel() > 0
    npos = int(y.sum())
    if npos == 0:
        return float("nan")
    ap, prev_recall = 0.0, 0.0
    for t in np.unique(scores)[::-1]:
        pred = scores >= t
        tp = int((pred & y).sum())
        recall, precision = tp / npos, tp / int(pred.sum())
        ap += (recall - prev_recall) * precision
        prev_recall = recall
    return ap


def average_precision_fast(scores, labels) -> float:
    """Same value via one sort (for sizes where the O(n * thresholds) definition above is too slow)."""
    scores = np.asarray(scores, dtype=np.float32).ravel()
    y = np.asarray(labels).ravel() > 0
    npos = int(y.sum())
    if npos == 0:
        return float("nan")
    order = np.argsort(-scores.astype(np.float64), kind="stable")
    s, y = scores[order], y[order]
    ends = np.r_[s[1:] != s[:-1], True]
    tp = np.cumsum(y)[ends].astype(np.float64)
    k = (np.flatnonzero(ends) + 1).astype(np.float64)
    return float(np.sum(np.diff(np.r_[0.0, tp]) * tp / k) / npos)


def binned_auprc(scores, labels, thresholds) -> float:
    """Binned-threshold AUPRC: for each threshold t_k (ascending) predictions are `p >= t_k`; precision_k = TP/(TP+FP)
    with 0/0 := 1, recall_k = TP/(TP+FN); a final point (precision 1, recall 0) is appended and the area is the
    Riemann sum  sum_k (recall_k - recall_{k+1}) * precision_k."""
    scores = np.asarray(scores, dtype=np.float32).ravel()
    y = np.asarray(labels).ravel() > 0
    thresholds = np.asarray(thresholds, dtype=np.float32)
    npos = int(y.sum())
    if npos == 0:
        return float("nan")
    prec, rec = [], []
    for t in thresholds:
        pred = scores >= t
        tp, cnt = int((pred & y).sum()), int(pred.sum())
        prec.append(tp / cnt if cnt else 1.0)
        rec.append(tp / npos)
    prec.append(1.0)
    rec.append(0.0)
    return float(sum((rec[k] - rec[k + 1]) * prec[k] for k in range(len(thresholds))))


def macro_mean(per_label_ap, empty_label_ap=0.0) -> float:
    """Macro average over the label axis.  A label with no positive has an undefined AP (NaN above).  torcheval 0.0.7's
    precision-recall curve turns the 0/0 recalls of such a label into 1.0 ("If recalls are NaNs, set NaNs to 1.0s"),
    which makes its Riemann sum the precision at the top-ranked item = 0, and `average="macro"` then takes the plain
    mean over num_labels - i.e. such labels COUNT as 0 (this is why the reference evaluates with
    `only_represented_labels`, ProtNoteTrainer.py:469-472,517-519).  torcheval is absent here, so that reading is
    **parity unpinned**; `empty_label_ap=None` gives the other convention (skip those labels)."""
    a = np.asarray(per_label_ap, dtype=np.float64)
    if empty_label_ap is None:
        return float(np.nanmean(a))
    return float(np.mean(np.where(np.isnan(a), float(empty_label_ap), a)))
