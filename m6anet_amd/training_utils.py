"""Host-side mirror of the evaluation half of m6anet/utils/training_utils.py.

Only `validate` (training_utils.py:213-268) and its metric helpers (:15-45) -- SURVEY.md section 8(f)
rank 4: the same dictionary the reference returns, with the predictions computed by
M6ANetEngine.validate_forward (every read encoded once on the GPU; the reference re-encodes the 20
sampled reads of every site in every pass).  Training itself is out of scope.
"""
import time

import numpy as np


def _curve_counts(y_true, y_score):
    """Cumulative true/false positives at every distinct score, highest first
    (sklearn.metrics._ranking._binary_clf_curve for 0/1 labels, which roc_curve and
    precision_recall_curve -- training_utils.py:26,42 -- are built on)."""
    y_true = np.asarray(y_true, np.float64).ravel()
    y_score = np.asarray(y_score, np.float64).ravel()
    order = np.argsort(y_score, kind="mergesort")[::-1]
    y_score, y_true = y_score[order], y_true[order]
    distinct = np.where(np.diff(y_score))[0]
    idx = np.r_[distinct, y_true.size - 1]
    tps = np.cumsum(y_true)[idx]
    fps = 1 + idx - tps
    return fps, tps


def _trapezoid(y, x):
    """sklearn.metrics.auc's area (np.trapz)."""
    y, x = np.asarray(y, np.float64), np.asarray(x, np.float64)
    return float(np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]) * 0.5))


def get_roc_auc(y_true, y_pred):
    """auc(roc_curve(y_true, y_pred)) (training_utils.py:26-27); collinear points are not dropped, which
    leaves the trapezoid area unchanged."""
    fps, tps = _curve_counts(y_true, y_pred)
    fps, tps = np.r_[0, fps], np.r_[0, tps]
    if fps[-1] <= 0 or tps[-1] <= 0:
        return float("nan")
    return _trapezoid(tps / tps[-1], fps / fps[-1])


def get_pr_auc(y_true, y_pred):
    """auc(recall, precision) of precision_recall_curve(..., pos_label=1) (training_utils.py:42-43)."""
    fps, tps = _curve_counts(y_true, y_pred)
    ps = tps + fps
    precision = np.where(ps > 0, tps / np.maximum(ps, 1), 0.0)
    recall = tps / tps[-1] if tps[-1] > 0 else np.ones_like(tps)
    precision, recall = np.r_[precision[::-1], 1.0], np.r_[recall[::-1], 0.0]
    return -_trapezoid(precision, recall)


def binary_cross_entropy(y_pred, y_true):
    """torch.nn.BCELoss() (mean reduction, log clamped at -100), what train configs use as criterion."""
    p = np.asarray(y_pred, np.float32)
    y = np.asarray(y_true, np.float32)
    with np.errstate(divide="ignore"):
        lp = np.maximum(np.log(p), np.float32(-100.0))
        lq = np.maximum(np.log1p(-p), np.float32(-100.0))
    return float(np.mean(-(y * lp + (1.0 - y) * lq), dtype=np.float32))


def validate(engine, X, site_kmers, off, y_true, n_iterations=1, seed=0, criterion=binary_cross_entropy):
    """The reference's `validate(model, val_dl, device, criterion, n_iterations)` for a whole validation
    split held as the flat arrays of include/m6a.h (DataLoader order = site order, num_workers=0).
    Returns the same keys: y_pred (list of passes), y_true, compute_time, roc_auc, pr_auc, avg_loss."""
    start = time.time()
    y_pred, y_pred_avg = engine.validate_forward(X, site_kmers, off, n_iterations=n_iterations, seed=seed)
    if hasattr(y_pred, "cpu"):
        y_pred, y_pred_avg = y_pred.cpu().numpy(), y_pred_avg.cpu().numpy()
    compute_time = time.time() - start
    y_true = np.asarray(y_true).flatten()
    return {
        "y_pred": [row for row in y_pred],
        "y_true": y_true,
        "compute_time": compute_time,
        "roc_auc": get_roc_auc(y_true, y_pred_avg),
        "pr_auc": get_pr_auc(y_true, y_pred_avg),
        "avg_loss": criterion(y_pred_avg, y_true),
    }
