"""YAML configs with `include:` indirection, recursive override and dot access.

Host-side support code mirroring the public names of the reference's utils/config_utils.py
(`DictConfig`, `update_config`, `config_from_kwargs`, reference lines 6-141) so that the entry
scripts read the same.  Paths inside `include:` are resolved against the current directory first
(the reference's behaviour) and then against this package's `src/` parent.
"""
import argparse
import os

import yaml

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class DictConfig(dict):
    """dict whose nested dicts are reachable as attributes: cfg.model.encoder.transformer.n_layers."""

    def __getattr__(self, key):
        if key.startswith("__") and key.endswith("__"):      # pickle / copy probe dunders with getattr
            raise AttributeError(key)
        try:
            item = self[key]
        except KeyError:
            raise KeyError(key) from None
        return DictConfig(item) if isinstance(item, dict) else item

    def get_dict(self):
        return super()


def _open_config(path):
    for cand in (path, os.path.join(_PKG_ROOT, path)):
        if os.path.exists(cand):
            with open(cand, "r") as fh:
                return yaml.safe_load(fh)
    raise FileNotFoundError(path)


def _expand(node):
    """Replace every 'include:<file>' string by the parsed file, depth first."""
    if isinstance(node, str) and node.startswith("include:"):
        node = _open_config(node[len("include:"):])
    if isinstance(node, dict):
        for k in list(node):
            node[k] = _expand(node[k])
    return node


def _overlay(base, new):
    """Write the leaves of `new` over `base`, creating sub-dicts where `base` has none."""
    if not isinstance(new, dict):
        return new
    if not isinstance(base, dict):
        base = {}
    for k, v in new.items():
        base[k] = _overlay(base.get(k, {}), v)
    return base


def update_config(default_config, config=None):
    """`default_config` (dict or path) overridden by `config` (dict, path or None)."""
    if isinstance(default_config, str):
        default_config = _open_config(default_config)
    if config is None:
        config = default_config
    elif isinstance(config, str):
        config = _open_config(config)
    return DictConfig(_overlay(_expand(default_config), _expand(config)))


class ParseKwargs(argparse.Action):
    """argparse action turning `a.b=1 c=x` into {'a.b': '1', 'c': 'x'}."""

    def __call__(self, parser, namespace, values, option_string=None):
        parsed = {}
        for item in values:
            k, v = item.split("=")
            parsed[k] = v
        setattr(namespace, self.dest, parsed)


def convert_to_dtype(text):
    text = text.strip()
    if text.startswith("[") and text.endswith("]"):
        return [convert_to_dtype(x) for x in text[1:-1].split(",")]
    low = {"null": None, "None": None, "none": None, "true": True, "True": True, "false": False, "False": False}
    if text in low:
        return low[text]
    if text.replace("-", "").isdigit():
        return int(text)
    try:
        return float(text)
    except ValueError:
        return text


def config_from_kwargs(kwargs):
    """{'a.b.c': '3'} -> DictConfig({'a': {'b': {'c': 3}}})."""
    tree = {}
    for dotted, raw in (kwargs or {}).items():
        *parents, leaf = dotted.split(".")
        node = tree
        for part in parents:
            node = node.setdefault(part, {})
        node[leaf] = convert_to_dtype(raw)
    return DictConfig(tree)
