"""pytest plumbing of the experiment tests: the product suite's fixtures and markers (tests/conftest.py), loaded by path."""
import importlib.util
import os

_path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "conftest.py")
_spec = importlib.util.spec_from_file_location("atomnas_tests_conftest", _path)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("__")})
