"""Depth-frame decode on the MI355X (csrc/device_ingest.hip) against zlib and against the host PNG reader, bit for bit:
mspa_inflate_blocks_device on streams of every compression level / strategy / window (stored, fixed and dynamic blocks, run-like
and far matches), its refusal of damaged streams, mspa_png_unfilter_device on all five row filters, and the composed reader on
640 x 480 depth PNGs as Pillow writes them (what the reference's cv2.imread decodes, info_handler.py:149-155)."""
import os
import struct
import sys
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _upload(streams):
    import torch
    offs, pos = [], 0
    for s in streams:
        offs.append(pos)
        pos += (len(s) + 15) // 16 * 16 + 16
    buf = np.zeros(pos, dtype=np.uint8)
    for o, s in zip(offs, streams):
        buf[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    return (torch.from_numpy(buf).cuda(), torch.tensor(offs, dtype=torch.int64).cuda(),
            torch.tensor([len(s) for s in streams], dtype=torch.int64).cuda())


def _payloads(n, rng):
    """Byte strings of length n that exercise different parts of the decoder."""
    depth_like = (np.cumsum(rng.integers(-3, 4, n // 2)) % 4000 + 500).astype(">u2").tobytes()
    block = rng.integers(0, 256, 33000, dtype=np.uint8).tobytes()
    return {
        "random": rng.integers(0, 256, n, dtype=np.uint8).tobytes(),                       # literals only, long codes
        "zeros": bytes(n),                                                                 # distance 1, length 258 chains
        "period3": (b"abc" * n)[:n],                                                       # overlapping copies, dist < 64
        "period100": (bytes(range(100)) * n)[:n],                                          # dist >= 64 inside the ring
        "far": (block * (n // len(block) + 1))[:n],                                        # distance 33 000 > the 32 K window: literals;
        "far_window": ((block[:20000]) * (n // 20000 + 1))[:n],                            # distance 20 000: beyond the ring, from HBM
        "depth_like": (depth_like + bytes(n))[:n],
        "text": (b"the quick brown fox jumps over the lazy dog. " * (n // 40 + 1))[:n],
        "skewed": rng.choice(np.arange(256, dtype=np.uint8), n, p=np.r_[[0.5], np.full(255, 0.5 / 255)]).tobytes(),
        "mixed": (rng.integers(0, 4, n // 3, dtype=np.uint8).tobytes() + block[:n // 3] + bytes(n))[:n],
    }


def test_inflate_matches_zlib_on_every_kind_of_stream():
    import torch
    from mspa import engine
    rng = np.random.default_rng(7)
    N = 70001                                                   # not a multiple of 4: the tail of the ring leaves byte by byte
    cases, streams = [], []
    for name, data in _payloads(N, rng).items():
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                for wbits in (9, 15):
                    c = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
                    streams.append(c.compress(data) + c.flush())
                    cases.append((name, level, strategy, wbits, data))
    src, off, nb = _upload(streams)
    out, status = engine.inflate_blocks_device(src, off, nb, N)
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    got = out.cpu().numpy()
    wrong = [(c[:4], int(s)) for c, s in zip(cases, st) if s != 0]
    assert wrong == [], wrong[:10]
    for k, c in enumerate(cases):
        assert got[k, :N].tobytes() == c[4], c[:4]
    assert len(streams) == 400


def test_inflate_refuses_what_it_cannot_verify():
    import torch
    from mspa import engine
    rng = np.random.default_rng(8)
    N = 50000
    data = _payloads(N, rng)["depth_like"]
    good = zlib.compress(data, 6)
    streams = [good]
    kinds = ["good"]
    for cut in (1, 4, 5, 100, len(good) // 2):                 # truncated: inside the trailer, inside the data
        streams.append(good[:-cut])
        kinds.append(f"cut{cut}")
    streams.append(good[:-1] + bytes([good[-1] ^ 1]))          # wrong Adler-32
    kinds.append("adler")
    streams.append(zlib.compress(data + b"x", 6))              # inflates to N + 1
    kinds.append("longer")
    streams.append(zlib.compress(data[:-1], 6))                # inflates to N - 1
    kinds.append("shorter")
    streams.append(b"\x78\x9c" + bytes(40))                    # stored block with LEN = NLEN = 0
    kinds.append("zeros")
    streams.append(bytes([0x78, 0xBB]) + good[2:])             # FDICT set / bad FCHECK
    kinds.append("header")
    streams.append(b"\x78\x9c\x07" + bytes(20))                # block type 3
    kinds.append("type3")
    streams.append(b"")                                        # nothing at all
    kinds.append("empty")
    flips = []
    for _ in range(150):                                       # single-bit damage anywhere in the stream
        b = bytearray(good)
        pos = int(rng.integers(0, len(b) * 8))
        b[pos >> 3] ^= 1 << (pos & 7)
        flips.append(bytes(b))
    streams += flips
    kinds += ["flip"] * len(flips)
    src, off, nb = _upload(streams)
    out, status = engine.inflate_blocks_device(src, off, nb, N)
    torch.cuda.synchronize()
    st, got = status.cpu().numpy(), out.cpu().numpy()
    assert st[0] == 0 and got[0, :N].tobytes() == data
    for k in range(1, len(kinds)):
        if kinds[k] == "flip":                                  # damage is either detected or did not change the output
            assert st[k] != 0 or got[k, :N].tobytes() == data, k
        else:
            assert st[k] != 0, kinds[k]
    assert (st[len(kinds) - len(flips):] != 0).sum() >= len(flips) - 5
    # what zlib itself says about the same streams: it accepts none of the named ones either
    for k in range(1, len(kinds) - len(flips)):
        try:
            ok = zlib.decompress(streams[k]) == data
        except zlib.error:
            ok = False
        assert not ok, kinds[k]


def test_argument_checks():
    import torch
    from mspa import _lib, engine
    src, off, nb = _upload([zlib.compress(b"abc" * 100)])
    with pytest.raises(_lib.MspaError):
        engine.inflate_blocks_device(src[1:].contiguous()[0:16].clone()[1:], off, nb, 300)      # misaligned source
    out, status = engine.inflate_blocks_device(src, off + 1, nb, 300)                           # a stream off its 8-byte unit
    torch.cuda.synchronize()
    assert int(status[0]) == 1
    out, status = engine.inflate_blocks_device(src, off, nb + 10_000, 300)                      # a stream reaching past the buffer
    torch.cuda.synchronize()
    assert int(status[0]) == 1
    out, status = engine.inflate_blocks_device(src, off[:0], nb[:0], 300)
    assert tuple(status.shape) == (0,)


def _png_rows(a, filters, level=6):
    from test_sweep_cpu import _png_gray16
    return _png_gray16(a, filters, level)


def test_unfilter_all_five_filters_and_the_composed_reader(tmp_path):
    import torch
    from PIL import Image
    from mspa import ingest
    rng = np.random.default_rng(3)
    frames, paths = [], []
    shapes = [(37, 53), (37, 53), (37, 53), (37, 53), (37, 53), (37, 53), (37, 53)]
    for k, filters in enumerate([[0], [1], [2], [3], [4], [0, 1, 2, 3, 4], [4, 3, 2, 1]]):
        a = rng.integers(0, 65536, shapes[k], dtype=np.uint16) if k % 2 else \
            (np.add.outer(np.arange(37), np.arange(53)) * 419 % 65536).astype(np.uint16)
        p = str(tmp_path / f"f{k}.png")
        open(p, "wb").write(_png_rows(a, filters))
        frames.append(a)
        paths.append(p)
    got = ingest.read_depth_frames_device(paths, "cuda", 3)
    torch.cuda.synchronize()
    assert got.dtype == torch.int16 and tuple(got.shape) == (7, 37, 53)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), np.stack(frames))
    # taller than one 64-row band, wider than a ring line; every filter, Pillow's own adaptive choice, and frames the device must
    # hand to the host path: another bit depth, a truncated file, a stream with a damaged trailer
    big = [(rng.integers(0, 4000, (150, 200)) + np.add.outer(np.arange(150), np.arange(200)) * 3).astype(np.uint16) for _ in range(4)]
    bp = []
    for k, a in enumerate(big):
        p = str(tmp_path / f"big{k}.png")
        if k < 2:
            open(p, "wb").write(_png_rows(a, [[4, 3, 1, 2, 0], [3, 4]][k], level=[1, 9][k]))
        else:
            Image.fromarray(a).save(p, compress_level=[1, 6][k - 2])
        bp.append(p)
    p8 = str(tmp_path / "eight.png")
    Image.fromarray((big[0] >> 8).astype(np.uint8)).save(p8)
    damaged = str(tmp_path / "damaged.png")
    raw = bytearray(open(bp[3], "rb").read())
    idat = raw.rfind(b"IEND") - 4 - 4 - 1                       # last byte of the last IDAT payload = the Adler-32's last byte
    raw[idat] ^= 0x40
    open(damaged, "wb").write(bytes(raw))
    host = ingest.read_depth_frames(bp, 2)
    dev = ingest.read_depth_frames_device(bp, "cuda", 2)
    assert np.array_equal(dev.cpu().numpy().view(np.uint16), host) and np.array_equal(host, np.stack(big))
    mixed = ingest.read_depth_frames_device([bp[0], p8, bp[2]], "cuda", 2, general_reader=lambda p: np.array(Image.open(p)), hw=(150, 200))
    assert np.array_equal(mixed[1].cpu().numpy().view(np.uint16), big[0] >> 8) and np.array_equal(mixed[2].cpu().numpy().view(np.uint16), big[2])
    with pytest.raises(ValueError, match="corrupt"):            # the host path's verdict on the damaged file, not garbage pixels
        ingest.read_depth_frames_device([bp[0], damaged], "cuda", 2)


@pytest.mark.parametrize("h,w", [(5, 2), (70, 130), (33, 640), (9, 1024), (9, 1026), (3, 254), (66, 126)])
def test_unfilter_row_parallel_path_shapes(h, w, tmp_path):
    """Images with None / Sub / Up rows only take the row-parallel path (lanes along the row, Sub as a wave-wide prefix sum);
    widths that leave lanes partly or wholly idle, more than 8 dwords per lane (back to the skewed pipeline), heights past one
    64-row band -- every pixel against the host reader."""
    import torch
    from mspa import ingest
    rng = np.random.default_rng(h * 10007 + w)
    frames, paths = [], []
    for k, filters in enumerate([[1], [2], [0, 1, 2], [2, 1, 1, 0], [1, 2, 4]]):      # the last one: a Paeth row -> skewed pipeline
        a = rng.integers(0, 65536, (h, w), dtype=np.uint16)
        p = str(tmp_path / f"r{k}.png")
        open(p, "wb").write(_png_rows(a, filters, level=1))
        frames.append(a)
        paths.append(p)
    got = ingest.read_depth_frames_device(paths, "cuda", 2)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint16), np.stack(frames))
    assert np.array_equal(ingest.read_depth_frames(paths, 2), np.stack(frames))


def test_full_size_depth_frames_as_the_dataset_stores_them(tmp_path):
    """640 x 480 depth frames of the synthetic room (5 mm noise, 7 % invalid pixels) written by Pillow at three
    compression levels: the device decode equals the host decode on every pixel, and no frame needed the host."""
    import torch
    from PIL import Image
    from mspa import engine, ingest, synth
    sc = synth.make_scene(77, n_points=2048, n_frames=6, color_hw=(480, 640), depth_hw=(480, 640), invalid_pose_frac=0.0, with_color=False)
    paths = []
    for k, image_id in enumerate(sc.valid_image_ids):
        p = str(tmp_path / f"{image_id}.png")
        Image.fromarray(sc.depth[image_id]).save(p, compress_level=[1, 6, 9][k % 3])
        paths.append(p)
    host = ingest.read_depth_frames(paths, 4)
    buf, offsets, nbytes, st, cap = ingest.pack_depth_pngs(paths, 480, 640, 4)
    assert (st == 0).all() and (offsets % 16 == 0).all()
    for k in range(len(paths)):                                 # the packed bytes ARE the scanlines' zlib stream
        assert len(zlib.decompress(buf[offsets[k]:offsets[k] + nbytes[k]].tobytes())) == 480 * 1281
    src = torch.from_numpy(buf[:cap]).cuda()
    raw, status = engine.inflate_blocks_device(src, torch.from_numpy(offsets).cuda(), torch.from_numpy(nbytes).cuda(), 480 * 1281)
    out = engine.png_unfilter_device(raw, 480, 640, status)
    torch.cuda.synchronize()
    assert status.cpu().numpy().tolist() == [0] * len(paths)
    assert np.array_equal(out.cpu().numpy().view(np.uint16), host)
    assert np.array_equal(host, np.stack([sc.depth[i] for i in sc.valid_image_ids]))


def test_smooth_full_size_frames_take_the_paeth_pipeline(tmp_path):
    """The same room without its sensor noise: Pillow's adaptive writer gives nearly every row the Paeth filter (what smooth real
    depth gets), so the un-filter's skewed pixel-pair pipeline -- the second launch, png_unfilter_hard_kernel -- is what decodes
    these frames.  Every pixel against the host reader and the rendered frames; heights that leave the last band ragged."""
    import torch
    from PIL import Image
    from scipy.ndimage import uniform_filter
    from mspa import engine, ingest, synth
    sc = synth.make_scene(78, n_points=2048, n_frames=4, color_hw=(480, 640), depth_hw=(480, 640), invalid_pose_frac=0.0, with_color=False)
    for h in (480, 417):
        frames, paths = [], []
        for k, image_id in enumerate(sc.valid_image_ids):
            a = uniform_filter(sc.depth[image_id].astype(np.float64), 9).astype(np.uint16)[:h]
            p = str(tmp_path / f"s{h}_{image_id}.png")
            Image.fromarray(a).save(p, compress_level=[1, 6, 9, 6][k % 4])
            frames.append(a)
            paths.append(p)
        buf, offsets, nbytes, st, cap = ingest.pack_depth_pngs(paths, h, 640, 4)
        assert (st == 0).all()
        scan = zlib.decompress(buf[offsets[0]:offsets[0] + nbytes[0]].tobytes())
        assert sum(scan[y * 1281] == 4 for y in range(h)) > 0.9 * h                 # Paeth rows
        src = torch.from_numpy(buf[:cap]).cuda()
        raw, status = engine.inflate_blocks_device(src, torch.from_numpy(offsets).cuda(), torch.from_numpy(nbytes).cuda(), h * 1281)
        out = engine.png_unfilter_device(raw, h, 640, status)
        torch.cuda.synchronize()
        assert status.cpu().numpy().tolist() == [0] * len(paths)
        assert np.array_equal(out.cpu().numpy().view(np.uint16), np.stack(frames))
        assert np.array_equal(ingest.read_depth_frames(paths, 4), np.stack(frames))


def test_sens_depth_frames_inflated_on_the_device(tmp_path):
    """A .sens stream's zlib depth payloads (extract_posed_images.py:49-57 inflates them one by one): compressed bytes over PCIe,
    one wave per frame on the device -- the same frames as the host reader's, a damaged payload handed to zlib (whose error it is)."""
    import torch
    from mspa import sens, synth
    sc = synth.make_scene(31, n_points=512, n_frames=9, color_hw=(240, 320), depth_hw=(240, 320), invalid_pose_frac=0.0, with_color=False)
    ids = sc.image_ids
    path = str(tmp_path / "scene.sens")
    sens.write_sens(path, sc.K.astype(np.float32), [sc.E[i].astype(np.float32) for i in ids], [sc.depth[i] for i in ids])
    host = sens.read_sens(path, frame_skip=2)
    dev = sens.read_sens(path, frame_skip=2, depth_to_device="cuda")
    torch.cuda.synchronize()
    assert dev.depth.shape[0] == 0 and tuple(dev.depth_device.shape) == (5, 240, 320) and dev.frame_index == host.frame_index
    assert np.array_equal(dev.depth_device.cpu().numpy().view(np.uint16), host.depth)
    assert np.array_equal(host.depth, np.stack([sc.depth[i] for i in ids[::2]]))
    # a payload whose last byte (Adler-32) is damaged: the device declines it, zlib raises
    raw = bytearray(open(path, "rb").read())
    raw[-1] ^= 0x10
    bad = str(tmp_path / "bad.sens")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(zlib.error):
        sens.read_sens(bad, depth_to_device="cuda")


@pytest.mark.parametrize("h,w", [(120, 1296), (60, 2000), (200, 896)])
def test_wide_frames_whose_row_distance_lies_beyond_the_ring(h, w, tmp_path):
    """The inflate's output ring is 2 KB: a match up to 1 790 bytes back is copied LDS to LDS, a farther one loads the flushed bytes
    from HBM (deferred, behind ``vmcnt``).  A 640-pixel row is 1 281 bytes, so ScanNet's depth never leaves the ring for the row
    above; THESE frames do with every such match -- 896 pixels: 1 793 bytes, the first distance past the ring; 1 296: 2 593 (inside
    the 4 KB ring this kernel had before); 2 000: 4 001 -- smooth content (long matches, Up / Paeth rows from an adaptive writer),
    repeated rows (matches at exactly one and two rows' distance) and noise; every pixel against the host reader and the source."""
    import torch
    from PIL import Image
    from mspa import ingest
    rng = np.random.default_rng(h + w)
    y, x = np.mgrid[0:h, 0:w]
    smooth = (1500 + 40 * np.sin(x / 37.0) + 25 * np.cos(y / 11.0) + x // 7).astype(np.uint16)
    rows = np.tile(rng.integers(0, 65536, (2, w), dtype=np.uint16), (h // 2, 1))             # row n == row n - 2: distance 2 (2 w + 1)
    noisy = (smooth + rng.integers(0, 6, (h, w))).astype(np.uint16)
    holes = smooth.copy()
    holes[rng.random((h, w)) < 0.07] = 0
    frames, paths = [], []
    for k, (a, level) in enumerate([(smooth, 6), (rows, 6), (noisy, 1), (holes, 9), (smooth, 1)]):
        p = str(tmp_path / f"w{k}.png")
        if k == 4:
            from test_sweep_cpu import _png_gray16
            open(p, "wb").write(_png_gray16(a, [2], level))                                  # Up rows only: the residue repeats row to row
        else:
            Image.fromarray(a).save(p, compress_level=level)
        frames.append(a)
        paths.append(p)
    dev = ingest.read_depth_frames_device(paths, "cuda", 2)
    torch.cuda.synchronize()
    host = ingest.read_depth_frames(paths, 2)
    assert np.array_equal(host, np.stack(frames))
    assert np.array_equal(dev.cpu().numpy().view(np.uint16), host)
