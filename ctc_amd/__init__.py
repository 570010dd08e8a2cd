"""ctc_amd -- MI355X-native CTC loss engine (hand-written HIP for gfx950).

Drop-in for the loss path of gotaku6629/CTC: ``CTCLoss.apply`` plus ``NoBlankCTC`` /
``NoBlankBinaryCTC`` / ``BlankCTC`` modules over the C ABI of include/ctc_amd.h.
"""
from ._lib import CtcAmdError, SO_PATH  # noqa: F401
from .functional import (CTCLoss, binary_best_path, binary_ctc_loss, binary_posteriors, blank_ctc_loss, check_status, collective_gate,  # noqa: F401
                         dedup_multihot_targets,
                         noblank_best_path, noblank_ctc_loss, noblank_posteriors, release_workspaces, set_blank_schedule,
                         workspace_status)
from .modules import BlankCTC, NoBlankBinaryCTC, NoBlankCTC  # noqa: F401
from .producer import LSTM_cell, head_forward, lstm_cell_step, lstm_series  # noqa: F401

__all__ = ["CTCLoss", "NoBlankCTC", "NoBlankBinaryCTC", "BlankCTC", "noblank_ctc_loss",
           "binary_ctc_loss", "blank_ctc_loss", "noblank_best_path", "noblank_posteriors", "CtcAmdError",
           "workspace_status", "release_workspaces", "check_status", "set_blank_schedule", "dedup_multihot_targets", "collective_gate", "LSTM_cell", "head_forward", "lstm_cell_step", "lstm_series", "binary_posteriors", "binary_best_path"]
