# The public surface is the module-level function API, used as `from librubiks import cube`
# (reference librubiks/cube/__init__.py:2).
from .cube import *  # noqa: F401,F403
from .cube import Cube, scramble_batch, sequence_scrambler_device  # noqa: F401
from .device import DeviceCubes  # noqa: F401
