"""motion_infiller.models of the reference (model_dict registry, motion_infiller/models/__init__.py:5-7)."""
from ...models.prior_models import MotionInfillerVAE, MotionTrajJointModel  # noqa: F401

model_dict = {'motion_infiller_vae': MotionInfillerVAE}
