"""Reference package path `utils` (see atomnas_amd.utils)."""
