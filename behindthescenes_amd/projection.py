"""distance_to_z (utils/projection_operations.py:4-16) through the HIP kernel ``bts_distance_to_z``."""
from . import native


def distance_to_z(depths, projs):
    return native.distance_to_z(depths.float().contiguous(), projs.float())
