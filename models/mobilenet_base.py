"""Reference module path `models.mobilenet_base` -> MI355X implementation (atomnas_amd.models.mobilenet_base)."""
from atomnas_amd.models.mobilenet_base import *  # noqa: F401,F403
from atomnas_amd.models import mobilenet_base as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
