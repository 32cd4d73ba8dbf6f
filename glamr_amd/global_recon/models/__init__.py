"""global_recon.models of the reference (model_dict registry, global_recon/models/__init__.py:4-6)."""
from .global_recon_model import GlobalReconOptimizer  # noqa: F401

model_dict = {'global_recon_model': GlobalReconOptimizer}
