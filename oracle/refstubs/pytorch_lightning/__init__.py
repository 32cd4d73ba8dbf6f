"""Import stub for the absent `pytorch_lightning` (reference pins 1.3.5).  Only what the inference path touches:
`LightningModule` as an nn.Module with `load_from_checkpoint` reading {'state_dict': ...}
(motion_infiller/models/motion_traj_joint_model.py:44,65) and `loggers.LightningLoggerBase` (lib/utils/log_utils.py:5)."""
import torch
from torch import nn
from . import loggers  # noqa: F401


class LightningModule(nn.Module):

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **kwargs):
        ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
        model = cls(**kwargs)
        model.load_state_dict(ckpt['state_dict'], strict=strict)
        return model

    def log(self, *args, **kwargs):
        pass
