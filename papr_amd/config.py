"""Two-level YAML configuration with the PAPR merge semantics.

Mirrors the reference's config surface (utils.py:14-39, train.py:339-353): a scene file is
deep-merged over ``configs/default.yml``; the special list ``datasets`` is merged entry-wise by
its ``name`` field; the result is wrapped so nested keys read as attributes.
"""
import copy
import os

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")


class ConfigNode(dict):
    """dict with attribute access; nested dicts are wrapped on the way out (utils.py:14-19)."""

    def __getattr__(self, name):
        try:
            value = self[name]
        except KeyError:
            raise AttributeError(name)
        if isinstance(value, dict) and not isinstance(value, ConfigNode):
            value = ConfigNode(value)
        return value

    def __setattr__(self, name, value):
        self[name] = value


def deep_merge(base, override):
    """In-place merge of ``override`` into ``base`` (utils.py:22-39)."""
    for key, val in override.items():
        if isinstance(val, dict):
            deep_merge(base.setdefault(key, {}), val)
        elif isinstance(val, list) and key == "datasets":
            for entry in val:
                for have in base[key]:
                    if have["name"] == entry["name"]:
                        have.update(entry)
                        break
                else:
                    fresh = copy.deepcopy(base[key][0])
                    deep_merge(fresh, entry)
                    base[key].append(fresh)
        else:
            base[key] = val
    return base


def load_config(scene=None, overrides=None, default=None):
    """default.yml (+ scene yml) (+ overrides dict) -> plain nested dict."""
    with open(default or os.path.join(CONFIG_DIR, "default.yml")) as f:
        cfg = yaml.safe_load(f)
    if scene:
        path = scene if os.path.exists(scene) else os.path.join(CONFIG_DIR, scene)
        with open(path) as f:
            deep_merge(cfg, yaml.safe_load(f) or {})
    if overrides:
        deep_merge(cfg, copy.deepcopy(overrides))
    return cfg


def eval_config(train_cfg):
    """Evaluation view of a training config: ``dataset`` overridden by ``eval.dataset`` (train.py:351-353)."""
    cfg = copy.deepcopy(dict(train_cfg))
    cfg["dataset"].update(cfg["eval"]["dataset"])
    return cfg


def as_node(cfg):
    return cfg if isinstance(cfg, ConfigNode) else ConfigNode(cfg)


def parse_overrides(pairs):
    """`a.b.c=value` strings (the drivers' --set option) -> nested override dict; values are read as YAML."""
    over = {}
    for kv in pairs or []:
        key, val = kv.split("=", 1)
        node = over
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val)
    return over
