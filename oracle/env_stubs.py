"""Stand-ins for THIRD-PARTY packages the reference imports but this image lacks -- TEST INFRASTRUCTURE ONLY.

The unchanged entry points (`train_maskplanner.py`, `test_maskplanner.py`) import `omegaconf`, `wandb`, `seaborn`,
`point_cloud_utils` and `pyvista` at module level (utils/__init__.py:7,10; utils/pointcloud.py:5,7; utils/disk.py:9;
utils/visualize.py:10,12).  None of them is installed here and there is no network.  To check in the build container that the
drop-in aliases (maskplanner_amd/dropin.py) really carry those entry points, `install()` registers:

  * `omegaconf`: a small YAML-backed re-implementation of the handful of calls the reference makes (utils/args.py:59-110,
    utils/config.py:5-11): OmegaConf.load / merge / from_cli / to_container / save / create, configs with attribute AND
    item access, ListConfig.  Like OmegaConf (YAML 1.2 scalars) it reads `1e-3` as a float, which plain PyYAML does not.
  * `wandb`, `seaborn`, `point_cloud_utils`, `pyvista`: inert modules (any attribute is a no-op callable).

Nothing of this is reference code and none of it is used by the product (maskplanner_amd/) or on the GPU box.
"""
import re
import sys
import types

import yaml

_FLOAT = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?$")


def _scalar(v):
    """YAML 1.2 core-schema floats that PyYAML (YAML 1.1) leaves as strings: 1e-3, 1E5, ..."""
    if isinstance(v, str) and _FLOAT.match(v) and not v.isdigit():
        try:
            return float(v)
        except ValueError:
            return v
    return v


class ListConfig(list):
    @property
    def _content(self):
        return list(self)


class DictConfig(dict):
    """dict with attribute access; nested mappings / lists are wrapped on the way in."""

    def __init__(self, data=None):
        super().__init__()
        for k, v in (data or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, _wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return self[k] if k in self else default


def _wrap(v):
    if isinstance(v, DictConfig) or isinstance(v, ListConfig):
        return v
    if isinstance(v, dict):
        return DictConfig(v)
    if isinstance(v, (list, tuple)):
        return ListConfig(_wrap(x) for x in v)
    return _scalar(v)


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_plain(x) for x in v]
    return v


def _merge_into(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge_into(dst[k], v)
        else:
            dst[k] = _wrap(_plain(v))     # a copy: later mutation of one config must not leak into another
    return dst


class OmegaConf:
    @staticmethod
    def create(data=None):
        return _wrap(data or {})

    @staticmethod
    def load(path):
        with open(path) as f:
            return _wrap(yaml.safe_load(f) or {})

    @staticmethod
    def merge(*configs):
        out = DictConfig()
        for c in configs:
            _merge_into(out, c)
        return out

    @staticmethod
    def from_cli(args_list=None):
        """key=value pairs; dotted keys nest; values are YAML scalars / flow sequences (`config=[a,b]`)."""
        out = DictConfig()
        for arg in (sys.argv[1:] if args_list is None else args_list):
            if "=" not in arg:
                continue
            key, _, val = arg.partition("=")
            node = out
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = DictConfig()
                node = node[p]
            node[parts[-1]] = yaml.safe_load(val) if val != "" else None
        return out

    @staticmethod
    def to_container(cfg, resolve=False):
        return _plain(cfg)

    @staticmethod
    def save(config, f):
        text = yaml.safe_dump(_plain(config))
        if hasattr(f, "write"):
            f.write(text)
        else:
            with open(f, "w") as fh:
                fh.write(text)


class _Inert(types.ModuleType):
    """A module whose every attribute is an inert object: callable, indexable, settable, nothing happens."""

    class _Obj:
        def __call__(self, *a, **k):
            return self

        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return self

        def __setattr__(self, k, v):
            pass

        def __getitem__(self, k):
            return self

        def __setitem__(self, k, v):
            pass

        def __iter__(self):
            return iter(())

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Inert._Obj()


def install():
    """Register the stand-ins for whichever of the packages is not importable.  Returns the names installed."""
    import importlib.util
    done = []
    if importlib.util.find_spec("omegaconf") is None and "omegaconf" not in sys.modules:
        m = types.ModuleType("omegaconf")
        m.OmegaConf, m.ListConfig, m.DictConfig = OmegaConf, ListConfig, DictConfig
        lc = types.ModuleType("omegaconf.listconfig")
        lc.ListConfig = ListConfig
        dc = types.ModuleType("omegaconf.dictconfig")
        dc.DictConfig = DictConfig
        m.listconfig, m.dictconfig = lc, dc
        sys.modules.update({"omegaconf": m, "omegaconf.listconfig": lc, "omegaconf.dictconfig": dc})
        done.append("omegaconf")
    try:   # utils/cluster.py:7 imports a class that newer networkx releases dropped (used by offline post-processing only)
        import networkx.algorithms.tree as _nxt
        if not hasattr(_nxt, "Edmonds"):
            _nxt.Edmonds = type("Edmonds", (), {})
            done.append("networkx.algorithms.tree.Edmonds")
    except ImportError:
        pass
    for name in ("wandb", "seaborn", "point_cloud_utils", "pyvista"):
        if name not in sys.modules and importlib.util.find_spec(name) is None:
            sys.modules[name] = _Inert(name)
            done.append(name)
    return done
