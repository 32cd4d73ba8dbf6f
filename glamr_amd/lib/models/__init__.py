from .smpl import SMPL, ModelOutput, SMPL_MODEL_DIR  # noqa: F401
