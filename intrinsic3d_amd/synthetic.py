"""Seeded synthetic RGB-D + voxel-SDF scenes for the parity tests and bench.py.

The reference ships no data (Lion / Tomb-statuary are not on disk), so every
configuration is exercised on generated inputs of the same shape the reference consumes:
a sparse `.tsdf`-style voxel list (sparse_voxel_grid.cpp:484-569: int32 key, f32 sdf, f32 weight,
u8 rgb) and per-keyframe luminance / depth / colour pyramids (rgbd/pyramid.cpp:59-166) with
world->camera angle-axis poses (intrinsic3d.cpp:184-188).

Nothing here is on the hot path: it is numpy plumbing that feeds the C-ABI.
"""
from __future__ import annotations

import numpy as np

SH_TRUE = np.array([0.8, 0.1, 0.3, -0.1, 0.05, 0.02, 0.04, -0.03, 0.02], dtype=np.float64)


def sh_basis(n: np.ndarray) -> np.ndarray:
    """9 SH basis terms in the reference's order (shading.h:53-67). n: (...,3)."""
    nx, ny, nz = n[..., 0], n[..., 1], n[..., 2]
    return np.stack([np.ones_like(nx), ny, nz, nx, nx * ny, ny * nz,
                     -nx * nx - ny * ny + 2.0 * nz * nz, nx * nz, nx * nx - ny * ny], axis=-1)


class Scene:
    """Bumpy sphere: sdf(p) = |p-c| - R + amp*sin(f x)sin(f y)sin(f z); albedo field; SH lighting."""

    def __init__(self, center, radius, bump_amp, bump_freq, sh=SH_TRUE, albedo_freq=25.0, albedo_amp=0.2):
        self.albedo_freq = float(albedo_freq)
        self.albedo_amp = float(albedo_amp)
        self.c = np.asarray(center, dtype=np.float64)
        self.R = float(radius)
        self.amp = float(bump_amp)
        self.freq = float(bump_freq)
        self.sh = np.asarray(sh, dtype=np.float64)

    def sdf(self, p):
        d = p - self.c
        r = np.sqrt((d * d).sum(-1))
        f = self.freq
        return r - self.R + self.amp * np.sin(f * p[..., 0]) * np.sin(f * p[..., 1]) * np.sin(f * p[..., 2])

    def normal(self, p):
        d = p - self.c
        r = np.sqrt((d * d).sum(-1, keepdims=True))
        g = d / np.maximum(r, 1e-12)
        f, a = self.freq, self.amp
        sx, sy, sz = np.sin(f * p[..., 0]), np.sin(f * p[..., 1]), np.sin(f * p[..., 2])
        cx, cy, cz = np.cos(f * p[..., 0]), np.cos(f * p[..., 1]), np.cos(f * p[..., 2])
        g = g + a * f * np.stack([cx * sy * sz, sx * cy * sz, sx * sy * cz], axis=-1)
        return g / np.sqrt((g * g).sum(-1, keepdims=True))

    def albedo(self, p):
        f = self.albedo_freq
        if f == 25.0:
            return 0.6 + self.albedo_amp * np.sin(25.0 * p[..., 0]) * np.cos(25.0 * p[..., 1])
        # a textured variant (three incommensurate directions, so that no rotation about the centre maps the pattern onto itself)
        return 0.6 + self.albedo_amp * (np.sin(f * p[..., 0] + 0.3) * np.cos(0.83 * f * p[..., 1]) + 0.5 * np.sin(1.31 * f * p[..., 2] + 0.47 * f * p[..., 0])) / 1.5

    def shade(self, p):
        return self.albedo(p) * (sh_basis(self.normal(p)) @ self.sh)


def look_at_pose(eye, target, up=(0.0, 1.0, 0.0)):
    """world->camera pose as (angle-axis[3], t[3]); camera looks along +z (pinhole model of camera.cpp:124)."""
    eye = np.asarray(eye, dtype=np.float64)
    z = np.asarray(target, dtype=np.float64) - eye
    z /= np.linalg.norm(z)
    x = np.cross(np.asarray(up, dtype=np.float64), z)
    if np.linalg.norm(x) < 1e-6:
        x = np.cross(np.array([1.0, 0.0, 0.0]), z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z], axis=0)
    t = -R @ eye
    return np.concatenate([rotmat_to_aa(R), t])


def rotmat_to_aa(R):
    tr = np.clip((np.trace(R) - 1.0) * 0.5, -1.0, 1.0)
    ang = np.arccos(tr)
    if ang < 1e-12:
        return np.zeros(3)
    ax = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    n = np.linalg.norm(ax)
    if n < 1e-9:  # angle ~ pi
        w, v = np.linalg.eigh((R + R.T) * 0.5)
        ax = v[:, -1]
        return ax * ang
    return ax / n * ang


def aa_to_rotmat(aa):
    th = np.linalg.norm(aa)
    if th < 1e-15:
        return np.eye(3)
    k = aa / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def pyr_down(img: np.ndarray) -> np.ndarray:
    """cv::pyrDown on a float image: 5x5 Gaussian [1 4 6 4 1]/16, BORDER_REFLECT_101, size (w/2, h/2)
    [OpenCV behaviour, not in reference; called at rgbd/pyramid.cpp:111]."""
    k = np.array([1.0, 4.0, 6.0, 4.0, 1.0], dtype=np.float32) / 16.0
    h, w = img.shape
    oh, ow = h // 2, w // 2
    p = np.pad(img, ((2, 2), (2, 2)), mode="reflect")
    tmp = np.zeros((h + 4, ow), dtype=np.float32)
    for i in range(5):
        tmp += k[i] * p[:, i:i + 2 * ow:2]
    out = np.zeros((oh, ow), dtype=np.float32)
    for i in range(5):
        out += k[i] * tmp[i:i + 2 * oh:2, :]
    return out


def depth_down(d: np.ndarray) -> np.ndarray:
    """Pyramid::downsampleDepth (rgbd/pyramid.cpp:115-143): mean of the valid taps of each 2x2 block."""
    h, w = d.shape[0] // 2, d.shape[1] // 2
    b = np.stack([d[0:2 * h:2, 0:2 * w:2], d[0:2 * h:2, 1:2 * w:2], d[1:2 * h:2, 0:2 * w:2], d[1:2 * h:2, 1:2 * w:2]])
    valid = b > 0
    cnt = valid.sum(0)
    s = np.where(valid, b, np.float32(0)).astype(np.float32)
    tot = ((s[0] + s[1]) + s[2]) + s[3]
    return np.where(cnt > 0, tot / np.maximum(cnt, 1).astype(np.float32), np.float32(0)).astype(np.float32)


def render_frame(scene: Scene, pose6, intr, w, h, noise_sigma=0.0, rng=None):
    """Analytic depth (ray / base sphere) + luminance (albedo * SH shading of the bumpy normal)."""
    fx, fy, cx, cy = intr
    R = aa_to_rotmat(pose6[:3])
    t = pose6[3:]
    eye = -R.T @ t
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    dc = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1)        # z_cam = 1
    dw = dc @ R                                                                   # R^T d
    oc = eye - scene.c
    a = (dw * dw).sum(-1)
    b = 2.0 * (dw @ oc)
    c = oc @ oc - scene.R ** 2
    disc = b * b - 4 * a * c
    hit = disc > 0
    s = np.where(hit, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), 0.0)          # z_cam depth since dc.z = 1
    hit &= s > 0
    p = eye + dw * s[..., None]
    lum = np.where(hit, scene.shade(p), 0.0)
    if noise_sigma > 0:
        lum = lum + rng.normal(0.0, noise_sigma, lum.shape)
    lum = np.clip(lum, 0.0, 1.0).astype(np.float32)
    depth = np.where(hit, s, 0.0).astype(np.float32)
    g = np.clip(lum * 255.0, 0, 255).astype(np.uint8)
    bgr = np.stack([g, g, g], axis=-1)
    return lum, depth, bgr


def shell_voxels(scene: Scene, voxel_size, band_vox, lo, hi, truncation=None):
    """Enumerate integer voxels in [lo,hi)^3 with |sdf| <= band_vox*voxel_size, slab by slab.
    A cheap squared-distance prefilter keeps the transcendental sdf evaluation to the candidate annulus."""
    if truncation is None:
        truncation = 5.0 * voxel_size
    band = band_vox * voxel_size
    keys, sdfs = [], []
    xs = np.arange(lo[0], hi[0], dtype=np.int32)
    ys = np.arange(lo[1], hi[1], dtype=np.int32)
    dx2 = (xs.astype(np.float64) * voxel_size - scene.c[0]) ** 2
    dy2 = (ys.astype(np.float64) * voxel_size - scene.c[1]) ** 2
    dense = band > 1e6
    m = band + abs(scene.amp) + voxel_size
    r_in2 = max(scene.R - m, 0.0) ** 2
    r_out2 = (scene.R + m) ** 2
    for z in range(lo[2], hi[2]):
        zc = z * voxel_size
        dz2 = (zc - scene.c[2]) ** 2
        if not dense and dz2 > r_out2:
            continue
        if dense:
            ix, iy = np.meshgrid(np.arange(xs.size), np.arange(ys.size), indexing="ij")
            ix = ix.ravel(); iy = iy.ravel()
        else:
            d2 = dx2[:, None] + dy2[None, :] + dz2
            ix, iy = np.nonzero((d2 >= r_in2) & (d2 <= r_out2))
        if ix.size == 0:
            continue
        P = np.stack([xs[ix].astype(np.float64) * voxel_size, ys[iy].astype(np.float64) * voxel_size, np.full(ix.size, zc)], axis=-1)
        s = scene.sdf(P)
        sel = np.abs(s) <= band
        if not sel.any():
            continue
        k = np.stack([xs[ix][sel], ys[iy][sel], np.full(int(sel.sum()), z, dtype=np.int32)], axis=-1)
        keys.append(k.astype(np.int32))
        sdfs.append(np.clip(s[sel], -truncation, truncation).astype(np.float32))
    return np.concatenate(keys), np.concatenate(sdfs)


def make_scene(radius_vox=24, voxel_size=0.004, K=4, width=160, height=120, levels=1, band_vox=3.2,
               dense_res=None, seed=0, cam_dist=None, bump_amp_vox=0.5, bump_freq=40.0, pose_noise=(0.0, 0.0),
               lum_noise=0.0, fx=None, shuffle=True, tint=True, albedo_freq=25.0, albedo_amp=0.2):
    """Returns a dict with the voxel list (file order), frames, poses, intrinsics and ground truth.

    dense_res: if given, a dense res^3 grid is emitted (config C1) instead of a thin shell.
    """
    rng = np.random.default_rng(seed)
    R = radius_vox * voxel_size
    if dense_res is not None:
        center = np.full(3, dense_res * voxel_size * 0.5)
        lo, hi = (0, 0, 0), (dense_res,) * 3
        band = 1e9
    else:
        margin = int(np.ceil(radius_vox + band_vox + 4))
        center = np.full(3, (margin + 2) * voxel_size)
        lo, hi = (0, 0, 0), (2 * margin + 4,) * 3
        band = band_vox
    scene = Scene(center, R, bump_amp_vox * voxel_size, bump_freq, albedo_freq=albedo_freq, albedo_amp=albedo_amp)
    keys, sdf = shell_voxels(scene, voxel_size, band, lo, hi)
    n = keys.shape[0]
    if shuffle:
        perm = rng.permutation(n)
        keys, sdf = keys[perm], sdf[perm]
    weight = np.ones(n, dtype=np.float32)
    P = keys.astype(np.float64) * voxel_size
    nrm = scene.normal(P)
    Piso = P - nrm * sdf[:, None].astype(np.float64)
    shade = np.clip(scene.shade(Piso), 0.0, 1.0)
    if tint:
        tintv = 1.0 + 0.15 * np.stack([np.sin(31.0 * P[:, 0]), np.sin(29.0 * P[:, 1] + 1.0), np.sin(37.0 * P[:, 2] + 2.0)], axis=-1)
    else:
        tintv = np.ones((n, 3))
    color = np.clip(shade[:, None] * tintv * 255.0, 0, 255).astype(np.uint8)

    if fx is None:
        fx = 525.0 * width / 640.0
    intr = np.array([fx, fx, (width - 1) * 0.5, (height - 1) * 0.5], dtype=np.float64)
    if cam_dist is None:
        # object fills ~70% of the image height
        cam_dist = R * fx / (0.35 * height) if R * fx / (0.35 * height) > 2.5 * R else 2.5 * R
    poses = np.zeros((K, 6))
    gold = (1 + 5 ** 0.5) / 2
    for f in range(K):
        zf = 1 - 2 * (f + 0.5) / K if K > 1 else 0.3
        rr = np.sqrt(max(0.0, 1 - zf * zf))
        ph = 2 * np.pi * f / gold
        d = cam_dist * (1.0 + 0.1 * np.sin(1.7 * f))
        eye = center + d * np.array([rr * np.cos(ph), zf, rr * np.sin(ph)])
        poses[f] = look_at_pose(eye, center)
    frames = []
    for f in range(K):
        lum, depth, bgr = render_frame(scene, poses[f], intr, width, height, lum_noise, rng)
        lums, depths, bgrs = [lum], [depth], [bgr]
        for _ in range(1, levels):
            lums.append(pyr_down(lums[-1]))
            depths.append(depth_down(depths[-1]))
            g = np.clip(lums[-1] * 255.0, 0, 255).astype(np.uint8)
            bgrs.append(np.stack([g, g, g], axis=-1))
        frames.append({"lum": lums, "depth": depths, "bgr": bgrs})
    if pose_noise[0] > 0 or pose_noise[1] > 0:
        poses[:, 3:] += rng.normal(0, pose_noise[0], (K, 3))
        poses[:, :3] += rng.normal(0, pose_noise[1], (K, 3))
    return {
        "voxel_size": np.float32(voxel_size), "keys": np.ascontiguousarray(keys), "sdf": sdf, "weight": weight,
        "color": np.ascontiguousarray(color), "frames": frames, "poses": poses, "intr": intr,
        "dist": np.zeros(5), "K": K, "width": width, "height": height, "levels": levels,
        "scene": scene, "center": center,
    }
