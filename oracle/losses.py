"""Oracle Dice loss: smp 0.3.3 ``losses.DiceLoss`` restated (TEST INFRASTRUCTURE).

Reference call sites: ``src/models/smp/model.py:55`` (construction,
MULTILABEL_MODE, from_logits=True) and ``:81,115`` (training/validation).
"""
import torch
import torch.nn.functional as F

MULTILABEL_MODE = 'multilabel'


def soft_dice_score(output, target, smooth=0.0, eps=1e-7, dims=None):
    if dims is not None:
        intersection = torch.sum(output * target, dim=dims)
        cardinality = torch.sum(output + target, dim=dims)
    else:
        intersection = torch.sum(output * target)
        cardinality = torch.sum(output + target)
    return (2.0 * intersection + smooth) / (cardinality + smooth).clamp_min(eps)


class DiceLoss(torch.nn.Module):
    def __init__(self, mode=MULTILABEL_MODE, from_logits=True, smooth=0.0, eps=1e-7):
        super().__init__()
        assert mode == MULTILABEL_MODE, 'the reference uses multilabel mode only'
        self.from_logits = from_logits
        self.smooth = smooth
        self.eps = eps

    def forward(self, y_pred, y_true):
        assert y_true.size(0) == y_pred.size(0)
        if self.from_logits:
            y_pred = F.logsigmoid(y_pred).exp()
        bs, c = y_true.size(0), y_pred.size(1)
        dims = (0, 2)
        y_true = y_true.view(bs, c, -1)
        y_pred = y_pred.view(bs, c, -1)
        scores = soft_dice_score(y_pred, y_true.type_as(y_pred), self.smooth, self.eps, dims)
        loss = 1.0 - scores
        mask = y_true.sum(dims) > 0
        loss = loss * mask.to(loss.dtype)
        return loss.mean()


def bce_with_logits(y_pred, y_true):
    """The BCE half of north_star's "Dice/BCE loss+grad": ``torch.nn.functional.binary_cross_entropy_with_logits`` with its defaults
    (reduction 'mean' over every element).  The reference itself builds DiceLoss only (``model.py:55``); this is the oracle of the
    engine's ``loss='bce'`` / ``'dice+bce'`` switch."""
    return F.binary_cross_entropy_with_logits(y_pred, y_true.type_as(y_pred))


class DiceBCELoss(torch.nn.Module):
    """Unweighted sum of the two (``loss='dice+bce'``)."""

    def __init__(self):
        super().__init__()
        self.dice = DiceLoss()

    def forward(self, y_pred, y_true):
        return self.dice(y_pred, y_true) + bce_with_logits(y_pred, y_true)
