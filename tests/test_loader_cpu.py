"""Dataset loader: PNG decoding, sensor folder, keyframe file, pose conversion.  Host-only code of the
product library — no device calls, so these run without a GPU.  PNG parity is pinned two ways: against Pillow's decoder on files Pillow
wrote, and against an independent encoder in this file (all five scanline filters, Adam7, 1/2/4/16-bit samples) that Pillow cannot write."""
import io
import struct
import zlib

import numpy as np
import pytest
from PIL import Image
from scipy.spatial.transform import Rotation

from intrinsic3d_amd import binding as B


# ------------------------------------------------------------------------------------------------------------ independent PNG writer
def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _pack_rows(smp, depth):
    """smp [h][w*cin] integer samples -> list of packed scanlines (bytes)"""
    rows = []
    for r in smp:
        if depth == 16:
            rows.append(np.asarray(r, ">u2").tobytes())
        elif depth == 8:
            rows.append(np.asarray(r, np.uint8).tobytes())
        else:
            bits = np.zeros(((len(r) * depth + 7) // 8) * 8, np.uint8)
            for i, v in enumerate(r):
                for b in range(depth):
                    bits[i * depth + b] = (int(v) >> (depth - 1 - b)) & 1
            rows.append(np.packbits(bits).tobytes())
    return rows


def _filter_rows(rows, bpp, rng):
    out = bytearray(); prev = bytes(len(rows[0])) if rows else b""
    for cur in rows:
        ft = int(rng.integers(0, 5)); line = bytearray(len(cur))
        for i in range(len(cur)):
            a = cur[i - bpp] if i >= bpp else 0; b = prev[i]; c = prev[i - bpp] if i >= bpp else 0
            if ft == 0: pred = 0
            elif ft == 1: pred = a
            elif ft == 2: pred = b
            elif ft == 3: pred = (a + b) >> 1
            else:
                p = a + b - c; pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            line[i] = (cur[i] - pred) & 0xFF
        out.append(ft); out += line; prev = cur
    return bytes(out)


def encode_png(smp, depth, ctype, interlace=False, plte=None, trns=None, seed=0):
    """smp: [h][w][cin] integer samples at `depth` bits"""
    rng = np.random.default_rng(seed)
    h, w, cin = smp.shape
    bpp = max(1, cin * depth // 8)
    raw = b""
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)] if interlace else [(0, 0, 1, 1)]
    for xs, ys, dx, dy in passes:
        sub = smp[ys::dy, xs::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        raw += _filter_rows(_pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, rng)
    z = zlib.compress(raw, 6)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        out += _chunk(b"PLTE", np.asarray(plte, np.uint8).tobytes())
    if trns is not None:
        out += _chunk(b"tRNS", bytes(trns))
    out += _chunk(b"tEXt", b"Comment\x00made by the test")
    half = len(z) // 2                                             # two IDAT chunks: the stream must be concatenated
    out += _chunk(b"IDAT", z[:half]) + _chunk(b"IDAT", z[half:]) + _chunk(b"IEND", b"")
    return out


def _pil_png(img, **kw):
    buf = io.BytesIO(); img.save(buf, format="PNG", **kw); return buf.getvalue()


# ------------------------------------------------------------------------------------------------------------ PNG
def test_png_matches_pillow():
    rng = np.random.default_rng(1)
    g8 = rng.integers(0, 256, (37, 53), np.uint8)
    np.testing.assert_array_equal(B.png_decode(_pil_png(Image.fromarray(g8))), g8)
    g16 = rng.integers(0, 65536, (48, 64)).astype(np.uint16)
    dec = B.png_decode(_pil_png(Image.fromarray(g16)))
    assert dec.dtype == np.uint16
    np.testing.assert_array_equal(dec, g16)
    rgb = rng.integers(0, 256, (41, 29, 3), np.uint8)
    np.testing.assert_array_equal(B.png_decode(_pil_png(Image.fromarray(rgb))), rgb[:, :, ::-1])           # imdecode: BGR
    rgba = rng.integers(0, 256, (17, 23, 4), np.uint8)
    np.testing.assert_array_equal(B.png_decode(_pil_png(Image.fromarray(rgba))), rgba[:, :, [2, 1, 0, 3]])
    la = rng.integers(0, 256, (9, 11, 2), np.uint8)
    dec = B.png_decode(_pil_png(Image.fromarray(la)))
    np.testing.assert_array_equal(dec, np.stack([la[:, :, 0]] * 3 + [la[:, :, 1]], -1))                           # grey+alpha -> B=G=R, A
    # palette image: expanded through the palette, BGR order
    pal = Image.fromarray(rgb).quantize(32)
    np.testing.assert_array_equal(B.png_decode(_pil_png(pal)), np.asarray(pal.convert("RGB"))[:, :, ::-1])
    # 1-bit: 0 / 255
    bw = Image.fromarray((g8 > 127).astype(np.uint8) * 255, "L").convert("1")
    np.testing.assert_array_equal(B.png_decode(_pil_png(bw)), np.asarray(bw.convert("L")))
    # a large smooth image exercises the filter heuristics of a real encoder
    yy, xx = np.mgrid[0:480, 0:640]
    smooth = np.stack([(xx // 3) % 256, (yy // 2) % 256, ((xx + yy) // 5) % 256], -1).astype(np.uint8)
    np.testing.assert_array_equal(B.png_decode(_pil_png(Image.fromarray(smooth), optimize=True)), smooth[:, :, ::-1])


@pytest.mark.parametrize("interlace", [False, True])
def test_png_own_encoder_all_formats(interlace):
    rng = np.random.default_rng(2)
    for (h, w) in [(1, 1), (3, 5), (19, 23), (8, 8)]:
        for depth in (1, 2, 4, 8, 16):                                     # grey
            s = rng.integers(0, 1 << depth, (h, w, 1))
            dec = B.png_decode(encode_png(s, depth, 0, interlace, seed=h + depth))
            want = s[:, :, 0] * (255 // ((1 << depth) - 1)) if depth < 8 else s[:, :, 0]
            np.testing.assert_array_equal(dec, want)
            # Pillow agrees on what these bytes mean (where it supports the format)
            if depth in (8, 16):
                np.testing.assert_array_equal(np.asarray(Image.open(io.BytesIO(encode_png(s, depth, 0, interlace, seed=h + depth)))), s[:, :, 0])
        for depth in (8, 16):
            s = rng.integers(0, 1 << depth, (h, w, 3))                     # RGB
            np.testing.assert_array_equal(B.png_decode(encode_png(s, depth, 2, interlace, seed=7)), s[:, :, ::-1])
            s = rng.integers(0, 1 << depth, (h, w, 4))                     # RGBA
            np.testing.assert_array_equal(B.png_decode(encode_png(s, depth, 6, interlace, seed=8)), s[:, :, [2, 1, 0, 3]])
            s = rng.integers(0, 1 << depth, (h, w, 2))                     # grey + alpha
            np.testing.assert_array_equal(B.png_decode(encode_png(s, depth, 4, interlace, seed=9)), s[:, :, [0, 0, 0, 1]])
        for depth in (1, 2, 4, 8):                                         # palette (+ tRNS -> 4 channels)
            ncol = 1 << depth
            plte = rng.integers(0, 256, (ncol, 3))
            s = rng.integers(0, ncol, (h, w, 1))
            np.testing.assert_array_equal(B.png_decode(encode_png(s, depth, 3, interlace, plte=plte)), plte[s[:, :, 0]][:, :, ::-1])
            trns = rng.integers(0, 256, ncol // 2 + 1).astype(np.uint8)
            alpha = np.concatenate([trns, np.full(ncol - trns.size, 255, np.uint8)]) if trns.size < ncol else trns[:ncol]
            dec = B.png_decode(encode_png(s, depth, 3, interlace, plte=plte, trns=trns[:ncol]))
            np.testing.assert_array_equal(dec, np.concatenate([plte[s[:, :, 0]][:, :, ::-1], alpha[s[:, :, 0]][:, :, None]], -1))
    # RGB with a colour key: alpha channel 0 exactly on the key
    s = rng.integers(0, 4, (6, 7, 3)) * 60
    key = s[2, 3]
    dec = B.png_decode(encode_png(s, 8, 2, interlace, trns=struct.pack(">HHH", *[int(v) for v in key])))
    assert dec.shape == (6, 7, 4)
    np.testing.assert_array_equal(dec[:, :, 3] == 0, (s == key).all(-1))


def test_png_rejects_damage():
    rng = np.random.default_rng(3)
    good = encode_png(rng.integers(0, 256, (12, 12, 3)), 8, 2)
    assert B.png_decode(good).shape == (12, 12, 3)
    k = good.index(b"IDAT") + 10                                          # a flipped byte inside the compressed stream: CRC mismatch on a critical chunk
    t = good.index(b"tEXt") + 6
    np.testing.assert_array_equal(B.png_decode(good[:t] + b"?" + good[t + 1:]), B.png_decode(good))     # damaged ancillary chunk: skipped
    for bad in (good[:40], b"JFIF" + good[4:], good[:k] + bytes([good[k] ^ 0xFF]) + good[k + 1:], good.replace(b"IEND", b"IENX")):
        with pytest.raises(B.I3DError):
            B.png_decode(bad)


# ------------------------------------------------------------------------------------------------------------ poses / keyframes
def test_pose_mat_to_vec6_matches_rotation_log():
    rng = np.random.default_rng(4)
    for i in range(200):
        rv = rng.normal(size=3); rv *= rng.uniform(0, np.pi * 0.999) / np.linalg.norm(rv)
        if i == 0: rv[:] = 0
        if i == 1: rv = np.array([1e-9, 0, 0])
        if i == 2: rv = np.array([0, np.pi * 0.9999, 0])                    # near pi: the w < 0 / trace <= 0 branches
        Rwc = Rotation.from_rotvec(rv).as_matrix(); twc = rng.normal(size=3)
        Tcw = np.eye(4); Tcw[:3, :3] = Rwc.T; Tcw[:3, 3] = -Rwc.T @ twc          # camera-to-world of that world-to-camera pose
        T32 = Tcw.astype(np.float32)
        got = B.pose_mat_to_vec6(T32)
        inv = np.linalg.inv(T32.astype(np.float64))
        want_rv = Rotation.from_matrix(inv[:3, :3]).as_rotvec()
        assert np.allclose(got[:3], want_rv, atol=2e-6), (i, got[:3], want_rv)   # float32 input: the rotation is only orthonormal to ~1e-7
        assert np.allclose(got[3:], inv[:3, 3], rtol=0, atol=1e-12)


def test_keyframes_file_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    scores = rng.uniform(0, 1, 47); scores[20:30] = 0.0                     # an all-zero window selects its first frame
    kf = B.keyframes_select(10, scores)
    assert kf.sum() == 5 and kf[20]
    for j in range(5):
        win = slice(10 * j, min(10 * j + 10, 47))
        assert kf[win].sum() == 1 and (scores[win][kf[win]][0] == scores[win].max())
    p = str(tmp_path / "keyframes.txt")
    B.keyframes_save(p, 10, scores, kf)
    lines = open(p).read().splitlines()
    assert lines[0] == "10" and lines[1] == f"{scores[0]:.6f} {int(kf[0])}" and len(lines) == 48
    win, s2, k2 = B.keyframes_load(p)
    assert win == 10 and np.array_equal(k2, kf) and np.allclose(s2, scores, atol=5e-7)
    with pytest.raises(B.I3DError):
        B.keyframes_load(str(tmp_path / "missing.txt"))


# ------------------------------------------------------------------------------------------------------------ sensor folder
def _make_dataset(folder, n, rng, cw=64, ch=48, dw=32, dh=24):
    folder.mkdir()
    Kc = np.eye(4); Kc[0, 0] = 70.0; Kc[1, 1] = 71.0; Kc[0, 2] = 31.5; Kc[1, 2] = 23.5
    Kd = np.eye(4); Kd[0, 0] = 35.0; Kd[1, 1] = 35.5; Kd[0, 2] = 15.5; Kd[1, 2] = 11.5
    np.savetxt(folder / "colorIntrinsics.txt", Kc); np.savetxt(folder / "depthIntrinsics.txt", Kd)
    data = []
    for i in range(n):
        col = rng.integers(0, 256, (ch, cw, 3), np.uint8)
        dep = rng.integers(0, 3000, (dh, dw)).astype(np.uint16)
        T = np.eye(4); T[:3, :3] = Rotation.from_rotvec(rng.normal(size=3) * 0.3).as_matrix(); T[:3, 3] = rng.normal(size=3)
        Image.fromarray(col).save(folder / f"frame-{i:06d}.color.png")
        Image.fromarray(dep).save(folder / f"frame-{i:06d}.depth.png")
        np.savetxt(folder / f"frame-{i:06d}.pose.txt", T)
        data.append((col, dep, T))
    return Kc, Kd, data


def test_sensor_folder(tmp_path):
    rng = np.random.default_rng(6)
    Kc, Kd, data = _make_dataset(tmp_path / "rgbd", 5, rng)
    s = B.Sensor(tmp_path / "rgbd", 0, 0.1, 2.0)
    assert s.num_frames == 5 and s.num_loaded == 5 and s.color_size == (64, 48) and s.depth_size == (32, 24)
    np.testing.assert_allclose(s.color_intrinsics, [70.0, 71.0, 31.5, 23.5]); np.testing.assert_allclose(s.depth_intrinsics, [35.0, 35.5, 15.5, 11.5])
    for i, (col, dep, T) in enumerate(data):
        np.testing.assert_array_equal(s.color(i), col[:, :, ::-1])
        d = dep.astype(np.float32) * np.float32(0.001)
        d[~(d > np.float32(0.1))] = 0; d[d > np.float32(2.0)] = 0
        np.testing.assert_array_equal(s.depth(i), d)
        np.testing.assert_array_equal(s.pose(i), T.astype(np.float32))
    np.testing.assert_array_equal(s.pose(17), np.eye(4, dtype=np.float32))          # SensorI3d::pose: identity for unknown ids
    # no thresholds: raw metres
    s0 = B.Sensor(tmp_path / "rgbd")
    np.testing.assert_array_equal(s0.depth(2), data[2][1].astype(np.float32) * np.float32(0.001))
    # max_frames: all files are listed, only the first two are stored
    s2 = B.Sensor(tmp_path / "rgbd", 2)
    assert s2.num_frames == 5 and s2.num_loaded == 2
    with pytest.raises(B.I3DError):
        s2.color(3)
    # save_poses is Sensor::savePoses; the write-back of a refined world->camera vector round-trips through it
    p6 = B.pose_mat_to_vec6(s.pose(1))
    s.set_pose_vec6(1, p6)
    assert np.allclose(s.pose(1), data[1][2], atol=1e-6)
    s.save_poses(tmp_path / "poses.txt")
    B.write_poses(str(tmp_path / "poses_ref.txt"), np.arange(5.0), np.stack([B.pose_mat_to_vec6(s.pose(i)) for i in range(5)]))
    a = np.loadtxt(tmp_path / "poses.txt"); b = np.loadtxt(tmp_path / "poses_ref.txt")
    assert a.shape == (5, 8) and np.allclose(a, b, atol=2e-6)
    q = Rotation.from_matrix(data[3][2][:3, :3]).as_quat()
    assert np.allclose(a[3, 1:4], data[3][2][:3, 3], atol=1e-6) and min(np.abs(a[3, 4:] - q).max(), np.abs(a[3, 4:] + q).max()) < 2e-6
    # a gap in the numbering ends the listing (listFiles stops at the first missing depth map)
    (tmp_path / "rgbd" / "frame-000003.depth.png").unlink()
    assert B.Sensor(tmp_path / "rgbd").num_frames == 3
    with pytest.raises(B.I3DError):
        B.Sensor("")


def test_yaml_get_and_cli_usage(tmp_path):
    """Settings::get<std::string> over the reference's flat yml files, and the argument handling of the two CLIs (no device needed)"""
    import os, subprocess
    yml = tmp_path / "sensor.yml"
    yml.write_text('%YAML:1.0\n\n# rgbd sensor config\n# ------------------\n\n# dataset file/folder \ndataset: "./rgbd/"\nmax_frames: "0"\n# minimum depth\nmin_depth: "0.1"\nempty: ""\n')
    assert B.yaml_get(yml, "dataset") == "./rgbd/" and B.yaml_get(yml, "min_depth") == "0.1" and B.yaml_get(yml, "empty") == ""
    assert B.yaml_get(yml, "missing", default="7") == "7"
    with pytest.raises(B.I3DError):
        B.yaml_get(yml, "missing")
    with pytest.raises(B.I3DError):
        B.yaml_get(tmp_path / "nope.yml", "dataset")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for app, flag in (("app_intrinsic3d", "-i"), ("app_fusion", "-f")):
        exe = os.path.join(root, "apps", app)
        if not os.path.exists(exe):
            pytest.skip("apps not built")
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "usage" in r.stderr
        r = subprocess.run([exe, "-s", str(yml)], capture_output=True, text=True)                  # second config missing
        assert r.returncode == 2
        r = subprocess.run([exe, f"--sensor={yml}", flag, str(yml)], capture_output=True, text=True)   # no frames in ./rgbd/: Sensor::create fails
        assert r.returncode == 1 and "RGB-D sensor could not be initialized" in r.stderr


def test_blur_score_and_app_keyframes(tmp_path):
    """KeyframeSelection::estimateBlur (Crete et al. 2007) against a numpy/scipy formulation, and apps/app_keyframes end to end (host only)"""
    import os, subprocess, sys
    from scipy.ndimage import correlate1d, gaussian_filter
    rng = np.random.default_rng(8)
    yy, xx = np.mgrid[0:120, 0:160]
    sharp = ((np.sin(xx * 0.9) * np.cos(yy * 0.7) > 0) * 200 + rng.integers(0, 40, xx.shape)).astype(np.uint8)
    bgr = np.stack([sharp, np.roll(sharp, 3, 1), np.roll(sharp, 5, 0)], -1)

    def crete(img_bgr):
        b, g, r = (img_bgr[..., c].astype(np.int64) for c in range(3))
        grey = ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.float32) * np.float32(1.0 / 255.0)
        k = np.full(9, np.float32(1.0 / 9.0), np.float32)
        bv = correlate1d(grey, k, axis=0, mode="mirror"); bh = correlate1d(grey, k, axis=1, mode="mirror")
        dfv = np.abs(np.diff(grey, axis=0)); dbv = np.abs(np.diff(bv, axis=0)); dfh = np.abs(np.diff(grey, axis=1)); dbh = np.abs(np.diff(bh, axis=1))
        sfv, svv = dfv.sum(dtype=np.float64), np.maximum(0, dfv - dbv).sum(dtype=np.float64)
        sfh, svh = dfh.sum(dtype=np.float64), np.maximum(0, dfh - dbh).sum(dtype=np.float64)
        return 1.0 - max((sfv - svv) / sfv, (sfh - svh) / sfh)

    s_sharp = B.blur_score(bgr)
    assert abs(s_sharp - crete(bgr)) < 1e-5
    soft = np.stack([gaussian_filter(bgr[..., c].astype(np.float32), 2.0) for c in range(3)], -1).astype(np.uint8)
    s_soft = B.blur_score(soft)
    assert abs(s_soft - crete(soft)) < 1e-5 and 0.0 < s_soft < s_sharp <= 1.0
    assert abs(B.blur_score(sharp) - B.blur_score(np.stack([sharp] * 3, -1))) < 1e-3          # grey input skips cvtColor; same image up to its rounding
    # the CLI: scores of all frames, the sharpest of every window of 3 is the keyframe
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "apps", "app_keyframes")
    if not os.path.exists(exe):
        pytest.skip("apps not built")
    _make_dataset(tmp_path / "rgbd", 7, rng)
    for i in (1, 5):                                                                           # blur two frames: they must not be selected
        p = tmp_path / "rgbd" / f"frame-{i:06d}.color.png"
        im = np.asarray(Image.open(p)).astype(np.float32)
        Image.fromarray(np.stack([gaussian_filter(im[..., c], 1.5) for c in range(3)], -1).astype(np.uint8)).save(p)
    (tmp_path / "sensor.yml").write_text('%YAML:1.0\ndataset: "./rgbd/"\nmax_frames: "0"\nmin_depth: "0.1"\nmax_depth: "2.0"\n')
    (tmp_path / "keyframes.yml").write_text('%YAML:1.0\nwindow_size: "3"\nfilename: "./fusion/keyframes.txt"\nshow_keyframes: "0"\n')
    r = subprocess.run([exe, "-s", str(tmp_path / "sensor.yml"), "-k", str(tmp_path / "keyframes.yml")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    win, scores, kf = B.keyframes_load(str(tmp_path / "fusion" / "keyframes.txt"))
    s = B.Sensor(tmp_path / "rgbd")
    want = np.array([B.blur_score(s.color(i)) for i in range(7)])
    assert win == 3 and np.allclose(scores, want, atol=1e-6) and np.array_equal(kf, B.keyframes_select(3, want))
    assert kf.sum() == 3 and not kf[1] and not kf[5] and kf[6]
