from .base_sampler import Base_sampler, sugbraph  # noqa: F401
from .dgllsampler import DGLLNeighborSampler  # noqa: F401
from .fast_sampler import FastNeighborSampler  # noqa: F401
