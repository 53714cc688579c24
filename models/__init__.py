"""Reference package path `models` (see atomnas_amd.models)."""
