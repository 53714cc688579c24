"""Reference module path `utils.optim` -> MI355X implementation (atomnas_amd.utils.optim)."""
from atomnas_amd.utils import optim as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
