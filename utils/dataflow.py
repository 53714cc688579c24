"""Reference module path `utils.dataflow` -> MI355X implementation (atomnas_amd.utils.dataflow)."""
from atomnas_amd.utils import dataflow as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
