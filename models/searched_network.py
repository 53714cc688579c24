"""Reference module path `models.searched_network` -> MI355X implementation (atomnas_amd.models.searched_network)."""
from atomnas_amd.models.searched_network import *  # noqa: F401,F403
from atomnas_amd.models import searched_network as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
