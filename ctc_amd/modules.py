"""nn.Module wrappers with the reference's constructor and forward signatures.

``models/__init__.py:82-83`` builds ``NoBlankCTC().cuda()`` / ``NoBlankBinaryCTC().cuda()``
and ``train.py:427,576`` calls ``ctc_loss(v_output, v_target_var, input_length,
v_target_length)``; switching to this engine is a one-line import change
(INTEGRATION.md).  Unlike the reference modules these are stateless and re-entrant
(the reference stashes T,B,C,S on ``self``, NoBlankCTC.py:130-131).
"""
import torch.nn as nn

from . import functional as F


class NoBlankCTC(nn.Module):
    """CTC without blank symbols (NoBlankCTC.py:22-141): T x S lattice, stay/advance."""

    def forward(self, yseq, label, input_length, target_length):
        # yseq [T,B,C] raw logits; label [B,S] class indices (-1 padded)
        return F.noblank_ctc_loss(yseq, label, input_length, target_length)[0]


class NoBlankBinaryCTC(nn.Module):
    """Multi-label sigmoid variant (NoBlankBinaryCTC.py:22-151): label [B,S,C] multi-hot."""

    def forward(self, yseq, label, input_length, target_length):
        return F.binary_ctc_loss(yseq, label, input_length, target_length)[0]


class BlankCTC(nn.Module):
    """torch.nn.CTCLoss(blank=0) as used at models/layers/AsyncTFCriterion.py:198:
    reduction='mean', zero_infinity=False, padded [N,S] targets, log-prob input."""

    def __init__(self, blank=0):
        super().__init__()
        self.blank = int(blank)

    def forward(self, log_probs, targets, input_lengths, target_lengths):
        return F.blank_ctc_loss(log_probs, targets, input_lengths, target_lengths, self.blank)[0]
