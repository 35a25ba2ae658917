"""Work-alikes of the detectron2 v0.6 containers the hot path exchanges at its boundary
(`Instances`, `RotatedBoxes`, `ImageList`, `ShapeSpec`) [d2-recall: detectron2/structures/
{instances,rotated_boxes,image_list}.py are not vendored in the reference].

Field contract consumed downstream (reference glass/inference/glass_runner.py:100-102,
glass/postprocess/post_processor_rotated_boxes.py:66-87, glass/evaluation/text_evaluator.py:
323-348): pred_boxes (RotatedBoxes, (cx,cy,w,h,angle_deg CCW)), scores, pred_classes,
orientations, pred_text_prob, pred_polygons.

These are host containers only: elementwise box arithmetic below is plumbing on tiny
(<=100 x 5) tensors; every hot-path computation goes through the HIP library.
"""
from __future__ import annotations

import itertools
import math
from collections import namedtuple
from typing import Any, Dict, List, Sequence, Tuple

import torch


class ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    def __new__(cls, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)


class RotatedBoxes:
    """(N,5) float tensor of (x_center, y_center, width, height, angle) boxes; angle in
    degrees, counter-clockwise positive in image coordinates."""

    def __init__(self, tensor: torch.Tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, 5)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 5, tensor.size()
        self.tensor = tensor

    def clone(self) -> "RotatedBoxes":
        return RotatedBoxes(self.tensor.clone())

    def to(self, *args, **kwargs) -> "RotatedBoxes":
        return RotatedBoxes(self.tensor.to(*args, **kwargs))

    def area(self) -> torch.Tensor:
        return self.tensor[:, 2] * self.tensor[:, 3]

    def normalize_angles(self) -> None:
        self.tensor[:, 4] = (self.tensor[:, 4] + 180.0) % 360.0 - 180.0

    def clip(self, box_size: Tuple[int, int], clip_angle_threshold: float = 1.0) -> None:
        """Clip only the (almost) horizontal boxes to the image; box_size = (h, w)."""
        h, w = box_size
        self.normalize_angles()
        idx = torch.where(torch.abs(self.tensor[:, 4]) <= clip_angle_threshold)[0]
        x1 = self.tensor[idx, 0] - self.tensor[idx, 2] / 2.0
        y1 = self.tensor[idx, 1] - self.tensor[idx, 3] / 2.0
        x2 = self.tensor[idx, 0] + self.tensor[idx, 2] / 2.0
        y2 = self.tensor[idx, 1] + self.tensor[idx, 3] / 2.0
        x1.clamp_(min=0, max=w)
        y1.clamp_(min=0, max=h)
        x2.clamp_(min=0, max=w)
        y2.clamp_(min=0, max=h)
        self.tensor[idx, 0] = (x1 + x2) / 2.0
        self.tensor[idx, 1] = (y1 + y2) / 2.0
        self.tensor[idx, 2] = torch.min(self.tensor[idx, 2], x2 - x1)
        self.tensor[idx, 3] = torch.min(self.tensor[idx, 3], y2 - y1)

    def nonempty(self, threshold: float = 0.0) -> torch.Tensor:
        return (self.tensor[:, 2] > threshold) & (self.tensor[:, 3] > threshold)

    def scale(self, scale_x: float, scale_y: float) -> None:
        self.tensor[:, 0] *= scale_x
        self.tensor[:, 1] *= scale_y
        theta = self.tensor[:, 4] * math.pi / 180.0
        c, s = torch.cos(theta), torch.sin(theta)
        self.tensor[:, 2] *= torch.sqrt((scale_x * c) ** 2 + (scale_y * s) ** 2)
        self.tensor[:, 3] *= torch.sqrt((scale_x * s) ** 2 + (scale_y * c) ** 2)
        self.tensor[:, 4] = torch.atan2(scale_x * s, scale_y * c) * 180 / math.pi

    def __getitem__(self, item) -> "RotatedBoxes":
        if isinstance(item, int):
            return RotatedBoxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, f"Indexing on RotatedBoxes with {item} failed to return a matrix!"
        return RotatedBoxes(b)

    def __len__(self) -> int:
        return self.tensor.shape[0]

    def __repr__(self) -> str:
        return "RotatedBoxes(" + str(self.tensor) + ")"

    @classmethod
    def cat(cls, boxes_list: List["RotatedBoxes"]) -> "RotatedBoxes":
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    @property
    def device(self) -> torch.device:
        return self.tensor.device

    def __iter__(self):
        yield from self.tensor


class Boxes:
    """Axis-aligned (x1,y1,x2,y2) boxes; only what the hot path's conversion helpers need
    (reference glass/structures/boxes.py:51-64)."""

    def __init__(self, tensor: torch.Tensor):
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4
        self.tensor = tensor

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    def to(self, *a, **k):
        return Boxes(self.tensor.to(*a, **k))

    @property
    def device(self):
        return self.tensor.device


class Instances:
    """Per-image bag of equally long fields (d2 `Instances` semantics)."""

    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        self._image_size = image_size
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self) -> Tuple[int, int]:
        return self._image_size

    def __setattr__(self, name: str, val: Any) -> None:
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name: str) -> Any:
        if name == "_fields" or name not in self._fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return self._fields[name]

    def set(self, name: str, value: Any) -> None:
        data_len = len(value)
        if len(self._fields):
            assert len(self) == data_len, \
                "Adding a field of length {} to a Instances of length {}".format(data_len, len(self))
        self._fields[name] = value

    def has(self, name: str) -> bool:
        return name in self._fields

    def remove(self, name: str) -> None:
        del self._fields[name]

    def get(self, name: str) -> Any:
        return self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def to(self, *args: Any, **kwargs: Any) -> "Instances":
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            if hasattr(v, "to"):
                v = v.to(*args, **kwargs)
            ret.set(k, v)
        return ret

    def __getitem__(self, item) -> "Instances":
        if type(item) == int:
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            if isinstance(v, list):
                if isinstance(item, torch.Tensor):
                    idx = item.nonzero().flatten().tolist() if item.dtype == torch.bool else item.tolist()
                    ret.set(k, [v[i] for i in idx])
                else:
                    ret.set(k, v[item])
            else:
                ret.set(k, v[item])
        return ret

    def __len__(self) -> int:
        for v in self._fields.values():
            return v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    @staticmethod
    def cat(instance_lists: List["Instances"]) -> "Instances":
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        image_size = instance_lists[0].image_size
        ret = Instances(image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = list(itertools.chain(*values))
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError("Unsupported type {} for concatenation".format(type(v0)))
            ret.set(k, values)
        return ret

    def __str__(self) -> str:
        s = self.__class__.__name__ + "("
        s += "num_instances={}, ".format(len(self) if self._fields else 0)
        s += "image_height={}, image_width={}, ".format(*self._image_size)
        s += "fields=[{}])".format(", ".join((f"{k}: {v}" for k, v in self._fields.items())))
        return s

    __repr__ = __str__


class ImageList:
    """Batch of images padded bottom/right to one size; `tensor` is (N,C,H,W) like d2's.
    The HIP pipeline keeps its own NHWC copy; this container only carries sizes and the
    NCHW view the reference's ROI-heads signature expects."""

    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self) -> int:
        return len(self.image_sizes)

    def __getitem__(self, idx) -> torch.Tensor:
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    @property
    def device(self):
        return self.tensor.device

    def to(self, *a, **k) -> "ImageList":
        return ImageList(self.tensor.to(*a, **k), self.image_sizes)

    @staticmethod
    def padded_shape(sizes: Sequence[Tuple[int, int]], size_divisibility: int = 0) -> Tuple[int, int]:
        mh = max(s[0] for s in sizes)
        mw = max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            mh = (mh + d - 1) // d * d
            mw = (mw + d - 1) // d * d
        return mh, mw

    @staticmethod
    def from_tensors(tensors: List[torch.Tensor], size_divisibility: int = 0,
                     pad_value: float = 0.0) -> "ImageList":
        assert len(tensors) > 0
        image_sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in tensors]
        mh, mw = ImageList.padded_shape(image_sizes, size_divisibility)
        batch_shape = [len(tensors)] + list(tensors[0].shape[:-2]) + [mh, mw]
        batched = tensors[0].new_full(batch_shape, pad_value)
        for img, pad_img in zip(tensors, batched):
            pad_img[..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batched.contiguous(), image_sizes)
