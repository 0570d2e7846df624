"""Extra losses / metrics of the drop-in API (mirror of the reference's General/LossesMetrics.py).

These are eval-side elementwise metrics (out of scope for kernels); they stay plain
torch expressions, vectorised where the reference loops in Python (kPrecision).  Same names, call
signatures and results.
"""
import torch
import torch.nn.functional as F

from .Core import TEN, ARR

__all__ = ['MSPE_loss', 'logMSE_loss', 'expMSPE_loss', 'fbeta_loss', 'kPrecision', 'AUC']


class MSPE_loss(object):
    "mean(((yhat - y)/y)^2)  (General/LossesMetrics.py:18-23)"
    def __call__(self, preds, target):
        return ((preds - target) / target).pow(2).mean()


class logMSE_loss(object):
    "mean((log yhat - log y)^2)  (General/LossesMetrics.py:25-32)"
    def __call__(self, preds, target):
        return (torch.log(preds) - torch.log(target)).pow(2).mean()


class expMSPE_loss(object):
    "MSPE of exp(preds) vs exp(target)  (General/LossesMetrics.py:34-42)"
    def __call__(self, preds, target):
        ep, et = torch.exp(preds), torch.exp(target)
        return ((ep - et) / et).pow(2).mean()


class fbeta_loss(object):
    "F-beta score for multi-label classification (General/LossesMetrics.py:44-78)."
    def __init__(self, beta, threshold=0.5, use_thresh=True, eps=1e-9):
        self.beta, self.eps = beta, eps
        self.threshold, self.use_thresh = threshold, use_thresh

    def __call__(self, y_pred, y_true):
        b2 = self.beta ** 2
        y_pred = (y_pred.sigmoid() >= self.threshold).float() if self.use_thresh else y_pred.float()
        y_true = y_true.float()
        tp = (y_pred * y_true).sum(dim=1)
        p = tp / (y_pred.sum(dim=1) + self.eps)
        r = tp / (y_true.sum(dim=1) + self.eps)
        return torch.mean((1 + b2) * (p * r) / (b2 * p + r + self.eps))


class kPrecision(object):
    "precision@k for single-label classification (General/LossesMetrics.py:80-107), vectorised."
    def __init__(self, k):
        self.k = k

    def __call__(self, preds, target, weights=None):
        N = len(preds)
        w = torch.ones(N, dtype=torch.float64) if weights is None else torch.as_tensor(weights, dtype=torch.float64)
        top = preds.sort(dim=1, descending=True)[1][:, :self.k]
        hit = (top == target.view(-1, 1))
        rank = torch.arange(1, self.k + 1, device=preds.device, dtype=torch.float64)
        prec = (hit.double() / rank).sum(dim=1).cpu()          # at most one hit per row
        return TEN(float((prec * w).sum() / w.sum()))


class AUC(object):
    "Area under the ROC curve for 2-class logits (General/LossesMetrics.py:110-124)."
    def __call__(self, preds, target):
        import sklearn.metrics as skm
        probs = torch.exp(F.log_softmax(preds, dim=1))
        return TEN(float(skm.roc_auc_score(ARR(target), ARR(probs[:, 1]))))
