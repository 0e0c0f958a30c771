"""behindthescenes_amd -- MI355X-native density-field renderer behind the reference's (Brummi/BehindTheScenes) Python
interfaces.  The computation lives in libbts_render.so (hand-written HIP for gfx950); see include/bts_render.h."""
from ._lib import BtsNativeError  # noqa: F401
from . import parallel, torch_modes  # noqa: F401
from .backbone import FeatureMapEncoder, make_backbone, register_backbone  # noqa: F401
from .code import PositionalEncoding  # noqa: F401
from .field import BTSNet  # noqa: F401
from .loss import ReconstructionLoss  # noqa: F401
from .mlp import ResnetBlockFC, ResnetFC, make_mlp  # noqa: F401
from .projection import distance_to_z  # noqa: F401
from .ray_sampler import ImageRaySampler, PatchRaySampler, RandomRaySampler, gen_rays  # noqa: F401
from .renderer import NeRFRenderer, _RenderWrapper  # noqa: F401
from .train_step import FusedEvalFrame, FusedTrainStep  # noqa: F401

__all__ = ["BTSNet", "NeRFRenderer", "PositionalEncoding", "ResnetFC", "ResnetBlockFC", "make_mlp", "make_backbone",
           "ImageRaySampler", "PatchRaySampler", "RandomRaySampler", "gen_rays", "distance_to_z", "ReconstructionLoss", "FusedTrainStep", "FusedEvalFrame", "BtsNativeError"]
