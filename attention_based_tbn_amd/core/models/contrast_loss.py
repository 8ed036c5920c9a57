"""ContrastLoss on the attention weights -- mirrors reference core/models/contrast_loss.py:4-25.
A handful of elementwise torch ops on an (R, T<=25) tensor; kept as PyTorch-ROCm ops."""
import torch.nn as nn


class ContrastLoss(nn.Module):
    def __init__(self, threshold=0.5, reduction=None):
        super().__init__()
        self.threshold = threshold
        if reduction not in ("mean", "batchmean", "sum"):
            raise Exception(f"{reduction} type reduction not supported for Contrast Loss")
        self.reduction = reduction

    def forward(self, input):
        hi = input.detach() >= self.threshold          # weights already above the threshold are pushed up
        signed = input.masked_fill(hi, 0) - input.masked_fill(~hi, 0)
        loss = signed.sum(dim=1)
        if self.reduction in ("mean", "batchmean"):
            loss = loss.mean()
        return loss
