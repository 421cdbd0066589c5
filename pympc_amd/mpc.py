"""Import-path compatibility: ``from pympc_amd.mpc import MPCController`` mirrors
``from pyMPC.mpc import MPCController``."""
from .controller import MPCController  # noqa: F401
