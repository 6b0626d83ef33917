"""`metric.eval_mosei_metric` — the module the reference driver imports (`from metric import *`,
main_frame_val_text_missing.py:39) and calls at :366-367 but does not ship.

Returns a dict with at least 'mae' and 'f1' (the keys the driver reads, main :292-294, :369), using the
MOSEI conventions the toolkit itself uses: binary accuracy / weighted F1 on the sign of the sentiment with
zero-label samples removed (toolkit/dataloader/cmumosei.py:149-163).  Host-side numpy: this is
evaluation bookkeeping on [N] vectors, not part of the GPU hot path."""
import numpy as np


def eval_mosei_metric(preds, labels, names=None):
    p = np.asarray(preds, dtype=np.float64).reshape(-1)
    y = np.asarray(labels, dtype=np.float64).reshape(-1)
    if p.shape != y.shape:
        raise ValueError(f"eval_mosei_metric: {p.shape} vs {y.shape}")
    out = {'mae': float(np.mean(np.abs(p - y))), 'mse': float(np.mean((p - y) ** 2))}
    out['corr'] = float(np.corrcoef(p, y)[0, 1]) if p.size > 1 and p.std() > 0 and y.std() > 0 else 0.0
    nz = y != 0
    yt, pt = y[nz] > 0, p[nz] > 0
    out['acc2'] = float(np.mean(yt == pt)) if nz.any() else 0.0
    f1 = 0.0
    for cls in (False, True):                      # weighted F1 over the two classes
        tp = np.sum((pt == cls) & (yt == cls))
        fp = np.sum((pt == cls) & (yt != cls))
        fn = np.sum((pt != cls) & (yt == cls))
        support = np.sum(yt == cls)
        if support and (2 * tp + fp + fn):
            f1 += support * (2 * tp / (2 * tp + fp + fn))
    out['f1'] = float(f1 / max(1, nz.sum()))
    out['n'] = int(p.size)
    return out
