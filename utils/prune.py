"""Reference module path `utils.prune` -> MI355X implementation (atomnas_amd.utils.prune)."""
from atomnas_amd.utils import prune as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
