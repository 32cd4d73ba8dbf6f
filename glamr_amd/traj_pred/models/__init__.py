"""traj_pred.models of the reference (model_dict registry, traj_pred/models/__init__.py:4-6)."""
from ...models.prior_models import TrajPredVAE  # noqa: F401

model_dict = {'traj_pred_vae': TrajPredVAE}
