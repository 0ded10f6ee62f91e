"""Importable alias of the `alfred-margaret_amd/` package directory (a hyphen is not a valid
Python identifier).  All code lives in `alfred-margaret_amd/`; this only extends __path__."""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "alfred-margaret_amd"))

from .api import *  # noqa: E402,F401,F403
from . import api, build  # noqa: E402,F401
