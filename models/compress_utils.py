"""Reference module path `models.compress_utils` -> MI355X implementation (atomnas_amd.models.compress_utils)."""
from atomnas_amd.models.compress_utils import *  # noqa: F401,F403
from atomnas_amd.models import compress_utils as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
