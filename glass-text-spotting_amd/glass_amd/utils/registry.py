"""Name -> class registry with the call surface the reference uses.

The reference selects every hot-path module by a name string in the cfg through
detectron2's `Registry` (`Registry("X").register()` as a decorator, `.get(name)`),
e.g. reference glass/modeling/fusion/local_feature_extraction.py:9,18 and
glass/modeling/fusion/fusion_modules.py:10,18.  detectron2 is not installed on
either box, so this is a work-alike; `mirror_into_detectron2()` additionally
registers the same objects into detectron2's own registries when it is importable.
"""
from __future__ import annotations

from typing import Any, Dict, Iterator, Optional, Tuple


class _Unbuilt:
    """placeholder of a reference name this build knows about but does not implement"""
    def __init__(self, reason: str) -> None:
        self.reason = reason


class Registry:
    def __init__(self, name: str) -> None:
        self._name = name
        self._obj_map: Dict[str, Any] = {}

    def register_unbuilt(self, name: str, reason: str) -> None:
        """A name the reference registers that this build deliberately does not implement: `get(name)` raises
        NotImplementedError with the reason instead of the KeyError of an unknown name."""
        self._do_register(name, _Unbuilt(reason))

    def _do_register(self, name: str, obj: Any) -> None:
        if name in self._obj_map:
            raise AssertionError(
                f"An object named '{name}' was already registered in '{self._name}' registry!")
        self._obj_map[name] = obj

    def register(self, obj: Any = None, *, name: Optional[str] = None) -> Any:
        if obj is None:
            def deco(func_or_class: Any) -> Any:
                self._do_register(name or func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(name or obj.__name__, obj)
        return obj

    def get(self, name: str) -> Any:
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        if isinstance(ret, _Unbuilt):
            raise NotImplementedError(f"'{name}' ({self._name} registry) is a reference name this build does not implement: {ret.reason}")
        return ret

    def __contains__(self, name: str) -> bool:
        return name in self._obj_map

    def __iter__(self) -> Iterator[Tuple[str, Any]]:
        return iter((k, v) for k, v in self._obj_map.items() if not isinstance(v, _Unbuilt))

    def keys(self):
        return self._obj_map.keys()

    def __repr__(self) -> str:
        return f"Registry({self._name}: {sorted(self._obj_map)})"


# detectron2-owned registries the reference registers into
# (reference glass/modeling/meta_arch/glass_rcnn.py:13, proposal_generator/rotated_rpn.py:16,
#  fusion/recognizers_hybrid_head.py:66).
META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")
ROI_MASK_HEAD_REGISTRY = Registry("ROI_MASK_HEAD")
ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")
RPN_HEAD_REGISTRY = Registry("RPN_HEAD")


def mirror_into_detectron2() -> bool:
    """If detectron2 is importable, register our classes under the same names in its
    registries so `detectron2.modeling.build_model(cfg)` resolves to them.  Returns
    False (and does nothing) when detectron2 is absent."""
    try:
        from detectron2.modeling import (  # type: ignore
            META_ARCH_REGISTRY as D2_META, PROPOSAL_GENERATOR_REGISTRY as D2_PG,
            ROI_HEADS_REGISTRY as D2_RH)
    except Exception:
        return False
    for ours, theirs in ((META_ARCH_REGISTRY, D2_META), (PROPOSAL_GENERATOR_REGISTRY, D2_PG),
                         (ROI_HEADS_REGISTRY, D2_RH)):
        for name, obj in ours:
            if name not in theirs:
                theirs.register(obj)
    return True
