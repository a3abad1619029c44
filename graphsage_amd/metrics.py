"""Streaming metrics the reference's training script asks Keras for
(``train_movielens.py:187-189``): root-mean-squared error, binary accuracy at
0.5, and ``AUC()`` with its defaults -- ROC curve over 200 fixed thresholds,
area by trapezoids over the false-positive axis.  State lives on the device of
the predictions; ``result()`` is the only call that synchronises.
"""
import torch

__all__ = ['RootMeanSquaredError', 'BinaryAccuracy', 'AUC']


def _flat(y_true, y_pred):
    y_pred = y_pred.detach().reshape(-1).to(torch.float32)
    y_true = torch.as_tensor(y_true, device=y_pred.device).reshape(-1).to(torch.float32)
    return y_true, y_pred


class RootMeanSquaredError:
    name = 'root_mean_squared_error'

    def __init__(self):
        self.reset_state()

    def reset_state(self):
        self.sq, self.n = None, 0

    def update_state(self, y_true, y_pred):
        y_true, y_pred = _flat(y_true, y_pred)
        s = ((y_pred - y_true) ** 2).sum().double()
        self.sq = s if self.sq is None else self.sq + s
        self.n += y_pred.numel()

    def result(self):
        return float((self.sq / self.n).sqrt()) if self.n else 0.0


class BinaryAccuracy:
    name = 'binary_accuracy'

    def __init__(self, threshold=0.5):
        self.threshold = threshold
        self.reset_state()

    def reset_state(self):
        self.hit, self.n = None, 0

    def update_state(self, y_true, y_pred):
        y_true, y_pred = _flat(y_true, y_pred)
        h = ((y_pred > self.threshold).to(torch.float32) == y_true).sum()
        self.hit = h if self.hit is None else self.hit + h
        self.n += y_pred.numel()

    def result(self):
        return float(self.hit) / self.n if self.n else 0.0


class AUC:
    name = 'auc'

    def __init__(self, num_thresholds=200):
        if num_thresholds < 2:
            raise ValueError("num_thresholds must be > 1")
        self.num_thresholds = num_thresholds
        eps = 1e-7
        self._th = [0.0 - eps] + [(i + 1) / (num_thresholds - 1) for i in range(num_thresholds - 2)] + [1.0 + eps]
        self.reset_state()

    def reset_state(self):
        self.tp = self.fp = None
        self.pos = self.neg = 0

    def update_state(self, y_true, y_pred):
        y_true, y_pred = _flat(y_true, y_pred)
        th = torch.tensor(self._th, dtype=torch.float32, device=y_pred.device)
        above = y_pred[None, :] > th[:, None]                       # [T, n]
        is_pos = (y_true > 0.5)
        tp = (above & is_pos[None, :]).sum(1).double()
        fp = (above & ~is_pos[None, :]).sum(1).double()
        self.tp = tp if self.tp is None else self.tp + tp
        self.fp = fp if self.fp is None else self.fp + fp
        self.pos = self.pos + is_pos.sum().double()
        self.neg = self.neg + (~is_pos).sum().double()

    def result(self):
        if self.tp is None:
            return 0.0
        tpr = self.tp / torch.clamp(self.pos, min=1e-7)
        fpr = self.fp / torch.clamp(self.neg, min=1e-7)
        return float(((fpr[:-1] - fpr[1:]) * (tpr[:-1] + tpr[1:]) / 2.0).sum())
