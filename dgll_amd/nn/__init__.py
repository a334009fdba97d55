"""dgll.nn -- conv layers of the reference (dgll/nn/Convolution/__init__.py:7 `__all__`) on the HIP engine."""
from .Convolution import *  # noqa: F401,F403
from .Convolution import __all__  # noqa: F401
