"""Reference module path `models.mobilenet_supernet` -> MI355X implementation (atomnas_amd.models.mobilenet_supernet)."""
from atomnas_amd.models.mobilenet_supernet import *  # noqa: F401,F403
from atomnas_amd.models import mobilenet_supernet as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
