"""mp-gadget_amd -- MI355X (gfx950) native TreePM (+SPH) force engine behind MP-Gadget's force-step entry points.

The directory name contains a hyphen (it mirrors the reference's name), so import it with
    importlib.import_module("mp-gadget_amd")
or through the `mpgadget_amd()` helper of the repo-root conftest / bench.
"""
from . import ics  # noqa: F401
from . import engine  # noqa: F401
from . import rows  # noqa: F401
from . import domain_peano  # noqa: F401
from . import dist  # noqa: F401
from .engine import Engine, EngineError, PARTICLE_DTYPE, make_particles, SphTimes, KickFactors, DriftKickTimes  # noqa: F401
