#!/usr/bin/env python3
"""CLI twin of the reference's inference.py: `python inference.py --model-path=MODEL`."""
from metro_pose3d_amd.inference import estimate_pose, main, visualize_pose  # noqa: F401

if __name__ == '__main__':
    main()
