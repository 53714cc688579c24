"""Reference module path `utils.rmsprop` -> MI355X implementation (atomnas_amd.utils.rmsprop)."""
from atomnas_amd.utils import rmsprop as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
