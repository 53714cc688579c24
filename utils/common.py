"""Reference module path `utils.common` -> MI355X implementation (atomnas_amd.utils.common)."""
from atomnas_amd.utils import common as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
