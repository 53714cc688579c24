"""Reference module path `utils.config` -> MI355X implementation (atomnas_amd.utils.config)."""
from atomnas_amd.utils import config as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
