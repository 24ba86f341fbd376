"""Python-file configs, loaded the way the reference's tools do.

The reference reads `configs/iou_aware_single_stage_detector/*.py` with
mmcv.Config.fromfile (reference tools/test.py:134, tools/train.py:45): the file
is executed and its module-level names become an attribute-dict tree.  mmcv is
a third-party dependency that is not part of this build, so the small subset of
its behaviour the path relies on is restated here: attribute access, `.get`,
`.copy`, `.pop`, `in`, item access, nested dicts converted recursively
(reference iou_aware_retina_head.py:536,561-563; bbox_nms.py:26-28).
"""
import os


class ConfigDict(dict):
    """dict with attribute access; nested dicts are wrapped on the way in."""

    def __init__(self, *args, **kwargs):
        super(ConfigDict, self).__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, ConfigDict):
            return v
        if isinstance(v, dict):
            return ConfigDict(v)
        if isinstance(v, list):
            return [ConfigDict._wrap(x) for x in v]
        if isinstance(v, tuple):
            return tuple(ConfigDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super(ConfigDict, self).__setitem__(k, self._wrap(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'ConfigDict' object has no attribute '%s'" % name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name)

    def copy(self):
        return ConfigDict(dict.copy(self))

    def __deepcopy__(self, memo):
        import copy
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]

    def to_dict(self):
        def un(v):
            if isinstance(v, ConfigDict):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(un(x) for x in v)
            return v
        return un(self)


class Config(object):
    """Result of `Config.fromfile(path)`: top-level names as attributes."""

    def __init__(self, cfg_dict=None, filename=None, text=''):
        object.__setattr__(self, '_cfg_dict', ConfigDict(cfg_dict or {}))
        object.__setattr__(self, '_filename', filename)
        object.__setattr__(self, '_text', text)

    @staticmethod
    def fromfile(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise IOError('config file "%s" does not exist' % filename)
        if not filename.endswith('.py'):
            raise IOError('only python-file configs are supported: %s' % filename)
        with open(filename, 'r') as f:
            text = f.read()
        scope = {'__file__': filename, '__name__': '__iouaware_config__'}
        exec(compile(text, filename, 'exec'), scope)
        names = {k: v for k, v in scope.items()
                 if not k.startswith('__') and not callable(v) and not _is_module(v)}
        return Config(names, filename=filename, text=text)

    @property
    def filename(self):
        return self._filename

    @property
    def text(self):
        return self._text

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def get(self, name, default=None):
        return self._cfg_dict.get(name, default)

    def __repr__(self):
        return 'Config (path: %s): %r' % (self._filename, dict(self._cfg_dict))


def _is_module(v):
    import types
    return isinstance(v, types.ModuleType)
