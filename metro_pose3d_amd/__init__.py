"""metro_pose3d_amd: MI355X (gfx950) implementation of the MeTRo inference hot path.

Public surface (mirrors reference inference.py):
    estimate_pose(images, model_path) -> (poses, joint_edges, joint_names)
plus the pieces under it: ModelSpec, Engine (plan + forward over libmetro_hip.so), the model
container (save_model / load_model) and batch sharding over the GPUs of a node (dist).
"""
from metro_pose3d_amd.spec import ModelSpec  # noqa: F401
from metro_pose3d_amd.modelfile import load_model, save_model  # noqa: F401

__all__ = ['ModelSpec', 'load_model', 'save_model', 'Engine', 'estimate_pose']


def __getattr__(name):
    # Engine / estimate_pose import torch; keep `import metro_pose3d_amd` light.
    if name == 'Engine':
        from metro_pose3d_amd.engine import Engine
        return Engine
    if name == 'estimate_pose':
        from metro_pose3d_amd.inference import estimate_pose
        return estimate_pose
    raise AttributeError(name)
