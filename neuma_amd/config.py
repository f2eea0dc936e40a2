"""YAML experiment configurations in the reference's schema (experiments/configs/**/*.yaml).

The reference reads them with OmegaConf into a DictConfig; omegaconf is not a dependency here - PyYAML plus a dict with
attribute access and `.get(key, default)` covers everything the drivers use (attribute reads, item assignment of the
"manually setting" overrides, nested `.get`)."""
from pathlib import Path

import yaml


class Cfg(dict):
    """dict with attribute access; nested dicts (also inside lists) are converted on construction and on assignment."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, Cfg):
            return Cfg(v)
        if isinstance(v, (list, tuple)):
            return [Cfg._wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Cfg._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        def un(v):
            if isinstance(v, dict):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, list):
                return [un(x) for x in v]
            return v
        return un(self)


def load_config(path) -> Cfg:
    with open(path) as f:
        return Cfg(yaml.safe_load(f) or {})


def save_config(cfg, path) -> None:
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with open(path, "w") as f:
        yaml.safe_dump(cfg.to_dict() if isinstance(cfg, Cfg) else dict(cfg), f, sort_keys=False)
