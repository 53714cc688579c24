"""yaml -> FLAGS with the reference's semantics (utils/config.py), without the import-time argv parsing.

  * `!include path.yml` relative to the including file (environment variables expanded); non-yaml files are included as text
  * `${VAR}` anywhere in a scalar is expanded from the environment (recorded in ENV_EXPANDED)
  * a top-level `_default:` mapping is the base the file's own keys override (applied at every include level)
  * keys containing dots ('model_kwparams.batch_norm_momentum': 0.01) are nested assignments into existing mappings
  * `app('app:<yml>', ['--a.b', 'v', ...])`: CLI overrides must exist and are cast with the type of the current value
    (so `--flag False` on a bool yields True, as in the reference)
The reference builds a module-level singleton from sys.argv when `utils.config` is imported; here `load_app(argv)` does the
same on demand and `FLAGS` is a lazily filled proxy, so importing this module has no side effects.
"""
import os
import re

import yaml

ENV_EXPANDED = {}
_ENV_PATTERN = re.compile(r'.*\$\{([^}^{]+)\}.*')


def nested_set(dic, keys, value, existed=False):
    for key in keys[:-1]:
        dic = dic[key]
    if existed:
        if keys[-1] not in dic:
            raise RuntimeError('{} does not exist in the dict'.format(keys[-1]))
        value = type(dic[keys[-1]])(value)
    dic[keys[-1]] = value


class _Loader(yaml.SafeLoader):
    def __init__(self, stream):
        self._root = os.path.dirname(getattr(stream, 'name', '')) or os.path.curdir
        super().__init__(stream)

    def _include(self, node):
        rel = os.path.expandvars(self.construct_scalar(node))
        path = os.path.abspath(os.path.join(self._root, rel))
        with open(path, 'r') as f:
            if os.path.splitext(path)[1].lower() in ('.yml', '.yaml'):
                return yaml.load(f, _Loader)
            return f.read()

    def _envpath(self, node):
        src = node.value
        res = os.path.expandvars(src)
        ENV_EXPANDED[src] = res
        return res

    def get_single_data(self):
        data = super().get_single_data()
        if not isinstance(data, dict):
            return data
        merged = data.pop('_default', None) or {}
        merged.update(data)
        for key in [k for k in merged if isinstance(k, str) and '.' in k]:
            nested_set(merged, key.split('.'), merged.pop(key))
        return merged


_Loader.add_constructor('!include', _Loader._include)
_Loader.add_constructor('!path', _Loader._envpath)
_Loader.add_implicit_resolver('!path', _ENV_PATTERN, None)


class AttrDict(dict):
    """dict with attribute access, recursively (utils/config.py:83-140)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in list(self.items()):
            self[k] = self._wrap(v)

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, list) and v and isinstance(v[0], dict):
            return [AttrDict._wrap(i) for i in v]
        return v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def yaml(self):
        return {k: (v.yaml() if isinstance(v, AttrDict) else ([i.yaml() if isinstance(i, AttrDict) else i for i in v] if isinstance(v, list) else v))
                for k, v in self.items()}

    def __repr__(self):
        lines = []
        for k, v in self.items():
            if isinstance(v, AttrDict):
                lines.append('{}:'.format(k))
                lines += ['    ' + ln for ln in repr(v).split('\n')]
            else:
                lines.append('{}: {}'.format(k, v))
        return '\n'.join(lines)


class Config(AttrDict):
    def __init__(self, filename=None, verbose=False):
        assert filename and os.path.exists(filename), 'File {} not exist.'.format(filename)
        with open(filename, 'r') as f:
            cfg = yaml.load(f, _Loader)
        cfg['config_path'] = filename
        AttrDict.__init__(self, cfg)
        if verbose:
            print(self)


def load_app(argv):
    """argv = ['app:<yml>', '--a.b', 'v', ...] -> Config (also installed as the module-level FLAGS)."""
    if not argv or not argv[0].startswith('app:'):
        raise RuntimeError('Cfg should start with `app:`')
    flags = Config(argv[0][4:])
    opts = argv[1:]
    if len(opts) % 2 == 1:
        raise RuntimeError('Override params should be key/val')
    for key, val in zip(opts[0::2], opts[1::2]):
        if not key.startswith('--'):
            raise RuntimeError('Override key should start with `--`')
        nested_set(flags, key[2:].split('.'), val, existed=True)
    FLAGS.bind(flags)
    return flags


class _FlagsProxy(object):
    """Module-level FLAGS: behaves like the loaded Config once load_app() has run."""
    _target = None

    def bind(self, cfg):
        object.__setattr__(self, '_target', cfg)

    def _t(self):
        if self._target is None:
            raise RuntimeError('FLAGS is empty: call atomnas_amd.utils.config.load_app(["app:<yml>", ...]) first')
        return self._target

    def __getattr__(self, name):
        return getattr(self._t(), name)

    def __setattr__(self, name, value):
        setattr(self._t(), name, value)

    def __getitem__(self, k):
        return self._t()[k]

    def __setitem__(self, k, v):
        self._t()[k] = v

    def __contains__(self, k):
        return k in self._t()

    def get(self, k, d=None):
        return self._t().get(k, d)

    def __repr__(self):
        return repr(self._t())


FLAGS = _FlagsProxy()
