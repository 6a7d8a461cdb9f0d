"""Annotated images of the "dot" task variants: filled discs and letter labels on a copy of a colour frame.

Reference: the cv2.circle / cv2.putText calls of depth_estimation_dot_engine.py:158-168, depth_comparison_dot_engine.py
:332-346, visual_correspondence_qa_engine_dot_2_multichoice.py:362-397, single_object_movement_engine_dot.py:328-339.
This is file I/O around the geometry path, not part of it: the marks (pixel, radius, colour, label) are computed by the
heads; drawing and JPEG encoding run on the host through Pillow.  Rendering is close to, not identical with, OpenCV's
(anti-aliasing and the Hershey font differ) -- the records never depend on the pixels.
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


class PillowAnnotator:
    """Draw marks on ``src`` and save to ``dst`` (JPEG).  Colours arrive in upstream's channel order (BGR) and are
    flipped for Pillow's RGB canvas, so the saved picture shows what OpenCV would have shown."""

    def annotate(self, src: str, dst: str, marks: Sequence[Mark]):
        from PIL import Image, ImageDraw
        os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
        with Image.open(src) as im:
            im = im.convert("RGB")
        draw = ImageDraw.Draw(im)
        for m in marks:
            rgb = (m.color[2], m.color[1], m.color[0])
            draw.ellipse((m.x - m.radius, m.y - m.radius, m.x + m.radius, m.y + m.radius), fill=rgb)
            if m.label:
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
