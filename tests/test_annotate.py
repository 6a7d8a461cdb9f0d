"""mspa/annotate.py: the filled disc of the "dot" task images is OpenCV's integer rasteriser restated (cv2.circle(..., -1),
LINE_8, shift 0 = the midpoint loop of drawing.cpp ``Circle(fill=true)``).  cv2 cannot be installed here, so the pin is the
algorithm evaluated by hand for the radii the reference uses (10; image_width // 100) plus its structural properties."""
import numpy as np

from mspa import annotate


def half_widths(radius, size=64):
    img = np.zeros((size, size), dtype=np.uint8)
    c = size // 2
    annotate.draw_filled_circle(img[..., None], (c, c), radius, 1)
    rows = {}
    for y in range(size):
        xs = np.nonzero(img[y])[0]
        if len(xs):
            assert xs[0] + xs[-1] == 2 * c and len(xs) == xs[-1] - xs[0] + 1        # one symmetric run per row
            rows[y - c] = (len(xs) - 1) // 2
    return rows


def test_radius_10_matches_the_hand_evaluated_loop():
    # err / dx after each step of Circle(): (dy, dx) = (0,10) (1,9) (2,9) (3,9) (4,9) (5,8) (6,8) (7,7); row +-dy gets half
    # width dx, row +-dx gets half width dy, the union per row:
    want = [10, 9, 9, 9, 9, 8, 8, 7, 6, 4, 0]
    rows = half_widths(10)
    assert sorted(rows) == list(range(-10, 11))
    assert [rows[k] for k in range(0, 11)] == want and [rows[-k] for k in range(0, 11)] == want


def test_small_radii_and_symmetry():
    assert half_widths(0) == {0: 0}                                  # a single pixel
    assert half_widths(1) == {-1: 0, 0: 1, 1: 0}                      # the plus shape
    for r in (2, 5, 6, 12, 19):
        rows = half_widths(r, 96)
        assert sorted(rows) == list(range(-r, r + 1)) and rows[0] == r and rows[r] <= rows[r - 1]
        assert all(rows[k] == rows[-k] for k in range(r + 1))
        # transpose symmetry of the midpoint circle: the mask equals its transpose
        img = np.zeros((96, 96, 1), dtype=np.uint8)
        annotate.draw_filled_circle(img, (48, 48), r, 1)
        assert np.array_equal(img[..., 0], img[..., 0].T)
        # never outside the true disc by more than the rasterisation's half pixel, never missing its interior
        yy, xx = np.mgrid[0:96, 0:96]
        d2 = (yy - 48) ** 2 + (xx - 48) ** 2
        assert (img[..., 0][d2 <= (r - 1) ** 2] == 1).all() and (img[..., 0][d2 > (r + 0.5) ** 2 + 1] == 0).all()


def test_clipping_at_the_image_border():
    full = np.zeros((200, 200, 3), dtype=np.uint8)
    annotate.draw_filled_circle(full, (100, 100), 10, (7, 8, 9))
    for cx, cy in ((3, 5), (38, 2), (0, 39), (39, 39), (-4, 20), (20, 45)):
        img = np.zeros((40, 40, 3), dtype=np.uint8)
        ox, oy = 100 - cx, 100 - cy                                  # the same disc, seen through a 40 x 40 window
        annotate.draw_filled_circle(img, (cx, cy), 10, (7, 8, 9))
        assert np.array_equal(img, full[oy:oy + 40, ox:ox + 40]), (cx, cy)
    assert annotate.filled_circle_spans(200, 200, 10, 40, 40) == []


def test_pillow_annotator_paints_the_same_disc(tmp_path):
    from PIL import Image
    src, dst = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    Image.fromarray(np.zeros((48, 64, 3), dtype=np.uint8)).save(src)
    ann = annotate.PillowAnnotator()
    ann.annotate(src, dst, [annotate.Mark(x=20, y=24, radius=10, color=(10, 20, 30))])     # BGR as upstream hands it over
    got = np.array(Image.open(dst).convert("RGB"))
    want = np.zeros((48, 64, 3), dtype=np.uint8)
    annotate.draw_filled_circle(want, (20, 24), 10, (30, 20, 10))
    assert np.array_equal(got, want)
