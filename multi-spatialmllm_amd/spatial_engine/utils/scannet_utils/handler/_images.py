"""Image I/O behind the façade: OpenCV when present (as in the reference), Pillow otherwise, and an
in-memory registry so synthetic scenes and tests need no files."""
from __future__ import annotations

import numpy as np

MEMORY = {}      # path -> ndarray (depth: [DH, DW] uint16; colour: [H, W, 3] uint8 RGB)


def register(path: str, array: np.ndarray):
    MEMORY[path] = array


def read_depth(path: str) -> np.ndarray:
    if path in MEMORY:
        return MEMORY[path]
    try:
        import cv2
        img = cv2.imread(path, -1)
        if img is None:
            raise FileNotFoundError(path)
        return img
    except ImportError:
        from PIL import Image
        return np.array(Image.open(path))


def read_color_rgb(path: str) -> np.ndarray:
    if path in MEMORY:
        return MEMORY[path]
    try:
        import cv2
        img = cv2.imread(path)
        if img is None:
            raise FileNotFoundError(path)
        return img[..., ::-1]
    except ImportError:
        from PIL import Image
        return np.array(Image.open(path).convert("RGB"))


def image_shape(path: str):
    if path in MEMORY:
        return MEMORY[path].shape[:2]
    try:
        from PIL import Image
        with Image.open(path) as im:      # header only: no decode needed for the size
            return im.size[1], im.size[0]
    except ImportError:
        return read_color_rgb(path).shape[:2]
