"""Reference module path `utils.model_profiling` -> MI355X implementation (atomnas_amd.utils.model_profiling)."""
from atomnas_amd.utils import model_profiling as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
