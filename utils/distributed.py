"""Reference module path `utils.distributed` -> MI355X implementation (atomnas_amd.utils.distributed)."""
from atomnas_amd.utils import distributed as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
