"""Annotated images of the "dot" task variants: filled discs and letter labels on a copy of a colour frame.

Reference: the cv2.circle / cv2.putText calls of depth_estimation_dot_engine.py:158-168, depth_comparison_dot_engine.py
:332-346, visual_correspondence_qa_engine_dot_2_multichoice.py:362-397, single_object_movement_engine_dot.py:328-339.
This is file I/O around the geometry path, not part of it: the marks (pixel, radius, colour, label) are computed by the
heads; drawing and JPEG encoding run on the host.  The filled disc is OpenCV's own integer rasteriser restated
(``filled_circle_spans``: cv2.circle(img, c, r, color, -1) with the default LINE_8 and shift 0 takes the midpoint loop of
modules/imgproc/src/drawing.cpp ``Circle(..., fill=true)``, no anti-aliasing), so the disc's pixels are the ones OpenCV
sets; the Hershey-font label and the JPEG bytes are Pillow's and are NOT claimed identical (cv2 is not installable here:
nothing could pin them) -- the records never depend on the pixels.
"""
from __future__ import annotations

import dataclasses
import os
import shutil
from typing import List, Optional, Sequence, Tuple


@dataclasses.dataclass
class Mark:
    x: int
    y: int
    radius: int
    color: Tuple[int, int, int]              # as drawn by upstream: a BGR triple handed to cv2 on a BGR image
    label: Optional[str] = None
    label_offset: Tuple[int, int] = (15, 0)


def generate_distinct_colors(n: int, rng, max_retries: int = 10) -> List[Tuple[int, int, int]]:
    """Up to ``max_retries`` random colours kept when far (L1 > 300) from those already kept, topped up from five fixed
    ones (DE_D:23-35 and its copies in the other dot scripts): same draws from ``rng`` in the same order."""
    colors: List[Tuple[int, int, int]] = []
    retries = 0
    while len(colors) < n and retries < max_retries:
        color = (rng.randint(0, 255), rng.randint(0, 255), rng.randint(0, 255))
        if all(sum(abs(a - b) for a, b in zip(color, other)) > 300 for other in colors):
            colors.append(color)
        retries += 1
    if len(colors) < n:
        fixed = [(255, 0, 0), (0, 255, 0), (0, 0, 255), (0, 0, 0), (255, 255, 255)]
        colors += rng.sample(fixed, n - len(colors))
    return colors


def filled_circle_spans(cx: int, cy: int, radius: int, width: int, height: int) -> List[Tuple[int, int, int]]:
    """Horizontal spans (y, x_first, x_last), inclusive and clipped to the image, that OpenCV's non-anti-aliased filled circle
    sets -- the integer midpoint loop of ``Circle()`` in modules/imgproc/src/drawing.cpp (the path cv2.circle takes for
    thickness < 0, LINE_8, shift 0): error term ``err``, odd increments ``plus`` / ``minus``, a row pair at +-dy with
    half-width dx and a row pair at +-dx with half-width dy per step.  Spans may repeat a row (as upstream's do)."""
    spans: List[Tuple[int, int, int]] = []

    def hline(y, x0, x1):
        if 0 <= y < height:
            x0, x1 = max(x0, 0), min(x1, width - 1)
            if x0 <= x1:
                spans.append((y, x0, x1))

    err, dx, dy, plus, minus = 0, int(radius), 0, 1, (int(radius) << 1) - 1
    while dx >= dy:
        hline(cy - dy, cx - dx, cx + dx)
        hline(cy + dy, cx - dx, cx + dx)
        hline(cy - dx, cx - dy, cx + dy)
        hline(cy + dx, cx - dy, cx + dy)
        dy += 1
        err += plus
        plus += 2
        if err > 0:                         # mask = (err <= 0) - 1: all ones exactly when err > 0
            err -= minus
            dx -= 1
            minus -= 2
    return spans


def draw_filled_circle(image, center: Tuple[int, int], radius: int, color) -> None:
    """cv2.circle(image, center, radius, color, -1) on an [H, W, C] array, in place."""
    h, w = image.shape[:2]
    for y, x0, x1 in filled_circle_spans(int(center[0]), int(center[1]), int(radius), w, h):
        image[y, x0:x1 + 1] = color


class PillowAnnotator:
    """Draw marks on ``src`` and save to ``dst`` (JPEG).  Colours arrive in upstream's channel order (BGR) and are
    flipped for Pillow's RGB canvas, so the saved picture shows what OpenCV would have shown."""

    def annotate(self, src: str, dst: str, marks: Sequence[Mark]):
        from PIL import Image, ImageDraw
        os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
        with Image.open(src) as im:
            im = im.convert("RGB")
        import numpy as np
        canvas = np.array(im)                         # [H, W, 3] RGB, writable copy
        for m in marks:                               # discs first, pixel for pixel as OpenCV rasterises them
            draw_filled_circle(canvas, (m.x, m.y), m.radius, (m.color[2], m.color[1], m.color[0]))
        im = Image.fromarray(canvas)
        draw = ImageDraw.Draw(im)
        for m in marks:
            if m.label:
                rgb = (m.color[2], m.color[1], m.color[0])
                draw.text((m.x + m.label_offset[0], m.y + m.label_offset[1] - 22), m.label, fill=rgb, font=self._font())
        im.save(dst, quality=95)

    def copy(self, src: str, dst: str):
        os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
        shutil.copy(src, dst)

    _cached_font = None

    @classmethod
    def _font(cls):
        if cls._cached_font is None:
            from PIL import ImageFont
            try:
                cls._cached_font = ImageFont.load_default(size=28)
            except TypeError:
                cls._cached_font = ImageFont.load_default()
        return cls._cached_font


class RecordingAnnotator:
    """Collects the jobs instead of touching images (tests, dry runs)."""

    def __init__(self):
        self.jobs: List[tuple] = []

    def annotate(self, src, dst, marks):
        self.jobs.append(("annotate", src, dst, [dataclasses.astuple(m) for m in marks]))

    def copy(self, src, dst):
        self.jobs.append(("copy", src, dst))
