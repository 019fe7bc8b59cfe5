"""Import alias: the package directory is `pair-net_amd/` (not a valid Python
identifier), so `import pairnet_amd` resolves its submodules from there."""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.dirname(
    _os.path.abspath(__file__))), "pair-net_amd"))

from .api import *  # noqa: F401,F403,E402
