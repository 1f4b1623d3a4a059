"""On-disk model container consumed by `estimate_pose(images, model_path)`.

The reference's `model_path` is a frozen TensorFlow GraphDef (`.pb`, written by
src/main.py:143-161 and read by inference.py:31-38).  `load_model` also reads such `.pb` files directly (tfgraph.py, SURVEY.md
section 8 row f1).  The native container is a NumPy `.npz` holding
  * `__metro_spec__`: JSON of ModelSpec (the flag values the graph would have baked in), and
  * one fp32 array per TF-slim variable, under its slim name, conv kernels HWIO -- i.e. exactly
    the constants a frozen graph holds, so the importer only has to produce this dictionary.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from metro_pose3d_amd.spec import ModelSpec

SPEC_KEY = '__metro_spec__'


def save_model(path: str, spec: ModelSpec, params: Dict[str, np.ndarray]) -> None:
    arrays = {k.replace('/', '|'): np.asarray(v, dtype=np.float32) for k, v in params.items()}
    arrays[SPEC_KEY] = np.frombuffer(spec.to_json().encode(), dtype=np.uint8)
    with open(path, 'wb') as f:
        np.savez(f, **arrays)


def load_model(path: str) -> Tuple[ModelSpec, Dict[str, np.ndarray]]:
    """`.npz` container (above) or a frozen TensorFlow GraphDef `.pb` as written by the reference's
    export (src/main.py:143-161), decoded without TensorFlow by tfgraph.py."""
    with open(path, 'rb') as f:
        magic = f.read(4)
    if magic[:2] != b'PK':                      # not a zip archive -> try GraphDef
        from metro_pose3d_amd.tfgraph import load_frozen_graph
        try:
            return load_frozen_graph(path)
        except Exception as e:                   # noqa: BLE001
            raise ValueError(f'{path}: neither a metro .npz container nor a readable frozen GraphDef ({e})')
    with np.load(path, allow_pickle=False) as z:
        if SPEC_KEY not in z.files:
            raise ValueError(f'{path}: not a metro model file (no {SPEC_KEY} entry)')
        spec = ModelSpec.from_json(bytes(z[SPEC_KEY]).decode())
        params = {k.replace('|', '/'): z[k] for k in z.files if k != SPEC_KEY}
    return spec, params
