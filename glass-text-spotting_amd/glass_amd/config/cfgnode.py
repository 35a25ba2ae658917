"""A small yacs-compatible CfgNode (yacs/detectron2 are not installed on either box).

Supports what the reference's predictor does with its cfg
(reference glass/inference/glass_runner.py:31-39): attribute access, `merge_from_file`
of the reference YAMLs verbatim (including `_BASE_`), `merge_from_list(opts)`,
`clone()`, `hasattr` probing (reference glass/modeling/meta_arch/glass_rcnn.py:40-53).
"""
from __future__ import annotations

import copy
import os
from ast import literal_eval
from typing import Any, List

import yaml

_BASE_KEY = "_BASE_"


class CfgNode(dict):
    def __init__(self, init_dict=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        # keys that may be set although absent from the defaults (yacs `set_new_allowed`)
        object.__setattr__(self, "_new_allowed", True)
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # attribute access -------------------------------------------------------
    def __getattr__(self, name: str) -> Any:
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name: str, value: Any) -> None:
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"Attempted to set {name} on a frozen CfgNode")
        self[name] = value

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        return out

    def clone(self) -> "CfgNode":
        return copy.deepcopy(self)

    def freeze(self) -> None:
        object.__setattr__(self, "_frozen", True)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self) -> None:
        object.__setattr__(self, "_frozen", False)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    # merging ----------------------------------------------------------------
    @staticmethod
    def load_yaml_with_base(filename: str) -> dict:
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}
        if _BASE_KEY in cfg:
            base = cfg.pop(_BASE_KEY)
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            merged = CfgNode.load_yaml_with_base(base)
            _merge_dict(cfg, merged)
            return merged
        return cfg

    def merge_from_file(self, cfg_filename: str) -> None:
        self.merge_from_other_cfg(self.load_yaml_with_base(cfg_filename))

    def merge_from_other_cfg(self, other: dict) -> None:
        _merge_into(self, other, [])

    def merge_from_list(self, cfg_list: List[Any]) -> None:
        assert len(cfg_list) % 2 == 0, "Override list has odd length: {}".format(cfg_list)
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            d = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                if p not in d:
                    d[p] = CfgNode()
                d = d[p]
            d[parts[-1]] = _coerce(_decode(v), d.get(parts[-1]), full_key)

    def dump(self) -> str:
        return yaml.safe_dump(_to_plain(self), default_flow_style=None)


def _to_plain(node):
    if isinstance(node, dict):
        return {k: _to_plain(v) for k, v in node.items()}
    if isinstance(node, tuple):
        return [_to_plain(v) for v in node]
    return node


def _merge_dict(src: dict, dst: dict) -> None:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge_dict(v, dst[k])
        else:
            dst[k] = v


def _decode(v: Any) -> Any:
    if not isinstance(v, str):
        return v
    try:
        return literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new: Any, old: Any, key: str) -> Any:
    """yacs type reconciliation: tuple<->list, int->float; otherwise replace."""
    if old is None or type(new) is type(old):
        return new
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        return list(new)
    if isinstance(old, float) and isinstance(new, int):
        return float(new)
    return new


def _merge_into(dst: CfgNode, src: dict, path: List[str]) -> None:
    for k, v in src.items():
        if isinstance(v, dict):
            if k not in dst or not isinstance(dst[k], CfgNode):
                dst[k] = CfgNode()
            _merge_into(dst[k], v, path + [k])
        else:
            dst[k] = _coerce(_decode(v) if isinstance(v, str) and k not in _RAW_STRING_KEYS else v,
                             dst.get(k), ".".join(path + [k]))


# string-valued keys that must never be literal_eval'ed (a character set such as
# "0123..." would otherwise turn into an int)
_RAW_STRING_KEYS = {"CHARACTER_SET", "WEIGHTS", "NAME", "FORMAT", "ROOT", "CONFIG", "OUTPUT_DIR"}
