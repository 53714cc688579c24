"""Reference module path `utils.transforms` -> MI355X implementation (atomnas_amd.utils.transforms)."""
from atomnas_amd.utils import transforms as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
