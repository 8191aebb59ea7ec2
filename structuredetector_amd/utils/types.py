"""Host-side annotation value types: the input type of `Encode` and the output type of `Decoder`.

API-compatible with the reference's `Keypoint` / `Box` / `Object` / `ImageAnnotation`
(src/sdnet/utils/utils.py:12-308): same attribute names, same in-place `resize` / `normalize`
with copying `resized` / `normalized` twins, same JSON schema (README.md:40-71).  Plain Python,
float64 arithmetic like the reference.
"""
from __future__ import annotations

import copy
import json
import math
from pathlib import Path


class _Scalable:
    """resize/normalize are in place and return self; the -ed variants work on a deep copy."""

    def _scale(self, fx, fy):
        raise NotImplementedError

    def resize(self, in_size, out_size):
        (iw, ih), (ow, oh) = in_size, out_size
        return self._scale(ow / iw, oh / ih, divide=False)

    def normalize(self, size):
        return self._scale(size[0], size[1], divide=True)

    def clone(self):
        """Deep copy.  The value types below override it with an explicit field-by-field copy: copy.deepcopy's generic walk (memo dict,
        __reduce_ex__ per object) was 70 % of Evaluator.accumulate, which takes seven resized() copies per image."""
        return copy.deepcopy(self)

    def resized(self, in_size, out_size):
        return self.clone().resize(in_size, out_size)

    def normalized(self, size):
        return self.clone().normalize(size)


class Keypoint(_Scalable):
    def __init__(self, kind, x, y, score=None):
        self.kind, self.x, self.y, self.score = kind, x, y, score

    def _scale(self, fx, fy, divide):
        if divide:
            self.x /= fx
            self.y /= fy
        else:
            self.x *= fx
            self.y *= fy
        return self

    def clone(self):
        return Keypoint(self.kind, self.x, self.y, self.score) if type(self) is Keypoint and len(self.__dict__) == 4 else copy.deepcopy(self)

    def distance(self, other):
        import numpy as np
        return float(np.hypot(self.x - other.x, self.y - other.y))          # utils.py:31-32 (np.hypot, not math.hypot: they may differ by an ulp)

    def json_repr(self):
        return {"kind": self.kind, "location": {"x": self.x, "y": self.y}, "score": self.score}

    @staticmethod
    def from_json(d):
        loc = d["location"]
        return Keypoint(d["kind"], loc["x"], loc["y"], d.get("score"))

    def __repr__(self):
        return f"Keypoint(kind: {self.kind}, x: {self.x}, y: {self.y}, score: {self.score})"


class Box(_Scalable):
    def __init__(self, x_min, y_min, x_max, y_max):
        self.x_min, self.y_min, self.x_max, self.y_max = x_min, y_min, x_max, y_max

    x_mid = property(lambda s: (s.x_max + s.x_min) / 2)
    y_mid = property(lambda s: (s.y_max + s.y_min) / 2)
    width = property(lambda s: abs(s.x_max - s.x_min))
    height = property(lambda s: abs(s.y_max - s.y_min))

    def _scale(self, fx, fy, divide):
        if divide:
            self.x_min /= fx; self.x_max /= fx; self.y_min /= fy; self.y_max /= fy
        else:
            self.x_min *= fx; self.x_max *= fx; self.y_min *= fy; self.y_max *= fy
        return self

    def clone(self):
        return Box(self.x_min, self.y_min, self.x_max, self.y_max) if type(self) is Box and len(self.__dict__) == 4 else copy.deepcopy(self)

    def yolo_coords(self, size):
        return (self.x_mid / size[0], self.y_mid / size[1], self.width / size[0], self.height / size[1])

    def standardize(self):
        self.x_min, self.x_max = sorted((self.x_min, self.x_max))
        self.y_min, self.y_max = sorted((self.y_min, self.y_max))
        return self

    def standardized(self):
        return copy.deepcopy(self).standardize()

    def json_repr(self):
        return {"x_min": self.x_min, "y_min": self.y_min, "x_max": self.x_max, "y_max": self.y_max}

    @staticmethod
    def from_json(d):
        return None if d is None else Box(d["x_min"], d["y_min"], d["x_max"], d["y_max"])

    def __repr__(self):
        return f"Box(x_min: {self.x_min}, y_min: {self.y_min}, x_max: {self.x_max}, y_max: {self.y_max})"


class Object(_Scalable):
    def __init__(self, name, anchor, parts=None, box=None):
        self.name, self.anchor, self.parts, self.box = name, anchor, (parts or []), box

    @property
    def x(self):
        return self.anchor.x

    @x.setter
    def x(self, v):
        self.anchor.x = v

    @property
    def y(self):
        return self.anchor.y

    @y.setter
    def y(self, v):
        self.anchor.y = v

    @property
    def nb_parts(self):
        return len(self.parts)

    def _scale(self, fx, fy, divide):
        for item in (self.anchor, self.box, *self.parts):
            if item is not None:
                item._scale(fx, fy, divide)
        return self

    def clone(self):
        if type(self) is not Object or len(self.__dict__) != 4:
            return copy.deepcopy(self)
        return Object(self.name, self.anchor.clone(), [kp.clone() for kp in self.parts], None if self.box is None else self.box.clone())

    def distance(self, other):
        return self.anchor.distance(other.anchor)

    def json_repr(self):
        return {"label": self.name, "box": self.box.json_repr() if self.box else None,
                "parts": [kp.json_repr() for kp in (self.anchor, *self.parts)]}

    @staticmethod
    def from_json(d, anchor_name):
        # the reference indexes d["box"] unconditionally (utils.py:213); accept a missing key too
        anchor, parts = None, []
        for kp in map(Keypoint.from_json, d["parts"]):
            if kp.kind == anchor_name:
                assert anchor is None, "More than one anchor found for object, achor must be unique."
                anchor = kp
            else:
                parts.append(kp)
        assert anchor is not None, f"Anchor part with name '{anchor_name}' not found while decoding JSON file."
        return Object(d["label"], anchor, parts, Box.from_json(d.get("box")))

    def __repr__(self):
        return f"Object(name: {self.name}, anchor: {self.anchor}, parts: {self.parts}, box: {self.box})"


class ImageAnnotation:
    def __init__(self, image_path, objects=None, img_size=None):
        self.image_path = Path(image_path)
        self.objects = objects or []
        self.img_size = img_size

    image_name = property(lambda s: s.image_path.name)
    image_stem = property(lambda s: s.image_path.stem)
    nb_parts = property(lambda s: sum(o.nb_parts for o in s.objects))
    is_empty = property(lambda s: len(s.objects) == 0)

    def __len__(self):
        return len(self.objects)

    def resize(self, in_size, out_size):
        for o in self.objects:
            o.resize(in_size, out_size)
        return self

    def clone(self):
        if type(self) is not ImageAnnotation or any(k not in ("image_path", "objects", "img_size") for k in self.__dict__):
            return copy.deepcopy(self)
        size = self.img_size
        return ImageAnnotation(self.image_path, [o.clone() for o in self.objects], copy.copy(size) if isinstance(size, list) else size)

    def resized(self, in_size, out_size):
        return self.clone().resize(in_size, out_size)

    def normalize(self, size=None):
        size = size or self.img_size
        assert size, f"Annotation for '{self.image_path}' does not have a size."
        for o in self.objects:
            o.normalize(size)
        return self

    def normalized(self, size=None):
        return self.clone().normalize(size)

    @staticmethod
    def from_json(file, anchor_name):
        d = json.loads(Path(file).read_text())
        return ImageAnnotation(Path(d["image_path"]), [Object.from_json(o, anchor_name) for o in d["objects"]],
                               d.get("img_size"))

    def json_repr(self):
        return {"image_path": str(self.image_path.expanduser().resolve()), "img_size": self.img_size,
                "objects": [o.json_repr() for o in self.objects]}

    def save_json(self, save_dir=None):
        save_dir = Path(save_dir or "detections/")
        save_dir.mkdir(exist_ok=True)
        (save_dir / self.image_path.with_suffix(".json").name).write_text(json.dumps(self.json_repr(), indent=2))

    def __repr__(self):
        return f"ImageAnnotation(name: {self.image_name}, objects: {self.objects}, img_size: {self.img_size})"
