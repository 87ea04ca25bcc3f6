"""Seeded synthetic frames for tests and bench (SURVEY.md section 8(d) "Synthetic inputs").

A scene is a float texture (Gaussian blobs + checker patches) larger than the frame; a sequence
resamples it bilinearly under a smooth global motion (translation <= 3 px/frame, rotation
<= 0.2 deg/frame) and adds +-2 grey levels of noise, so that most features survive between
frames and the true flow is known.  numpy only; deterministic for a given seed on any host.
"""
import numpy as np

MARGIN = 48


def make_scene(seed, w, h, blobs_per_px=1.0 / 400.0, checkers=12):
    rng = np.random.Generator(np.random.PCG64(0xC0FFEE + int(seed)))
    W, H = w + 2 * MARGIN, h + 2 * MARGIN
    img = np.full((H, W), 128.0, np.float32)
    k = max(1, int(W * H * blobs_per_px))
    cx = rng.uniform(0, W, k)
    cy = rng.uniform(0, H, k)
    sig = rng.uniform(1.5, 6.0, k)
    amp = rng.uniform(20.0, 100.0, k) * rng.choice([-1.0, 1.0], k)
    for i in range(k):
        r = int(np.ceil(3.5 * sig[i]))
        x0, x1 = max(0, int(cx[i]) - r), min(W, int(cx[i]) + r + 1)
        y0, y1 = max(0, int(cy[i]) - r), min(H, int(cy[i]) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        xs = np.arange(x0, x1, dtype=np.float32) - np.float32(cx[i])
        ys = np.arange(y0, y1, dtype=np.float32) - np.float32(cy[i])
        g = np.exp(-(ys[:, None] ** 2 + xs[None, :] ** 2) / np.float32(2.0 * sig[i] ** 2))
        img[y0:y1, x0:x1] += np.float32(amp[i]) * g
    for _ in range(checkers):
        s = int(rng.integers(6, 14))
        nx, ny = int(rng.integers(2, 5)), int(rng.integers(2, 5))
        x0 = int(rng.integers(0, max(1, W - s * nx)))
        y0 = int(rng.integers(0, max(1, H - s * ny)))
        a = float(rng.uniform(30.0, 70.0))
        for j in range(ny):
            for i in range(nx):
                sgn = 1.0 if (i + j) % 2 == 0 else -1.0
                img[y0 + j * s:y0 + (j + 1) * s, x0 + i * s:x0 + (i + 1) * s] += sgn * a
    return img


def render(scene, w, h, tx=0.0, ty=0.0, theta_deg=0.0, noise_seed=None):
    """8-bit frame = scene sampled at R(theta)*(p - c) + c + (tx, ty) (+ MARGIN), bilinear."""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    cx, cy = (w - 1) * 0.5, (h - 1) * 0.5
    th = np.deg2rad(theta_deg)
    c, s = np.float32(np.cos(th)), np.float32(np.sin(th))
    u = c * (xs - cx) - s * (ys - cy) + np.float32(cx + tx + MARGIN)
    v = s * (xs - cx) + c * (ys - cy) + np.float32(cy + ty + MARGIN)
    u = np.clip(u, 0, scene.shape[1] - 1.001)
    v = np.clip(v, 0, scene.shape[0] - 1.001)
    iu, iv = u.astype(np.int32), v.astype(np.int32)
    fu, fv = u - iu, v - iv
    p = (scene[iv, iu] * (1 - fu) * (1 - fv) + scene[iv, iu + 1] * fu * (1 - fv) +
         scene[iv + 1, iu] * (1 - fu) * fv + scene[iv + 1, iu + 1] * fu * fv)
    if noise_seed is not None:
        rng = np.random.Generator(np.random.PCG64(int(noise_seed)))
        p = p + rng.integers(-2, 3, p.shape).astype(np.float32)
    return np.clip(np.rint(p), 0, 255).astype(np.uint8)


def motion(seed, n_frames, max_t=3.0, max_rot=0.2):
    """Cumulative smooth motion (tx, ty, theta_deg) per frame."""
    rng = np.random.Generator(np.random.PCG64(0xBEEF + int(seed)))
    vx, vy = rng.uniform(-max_t, max_t, 2) * 0.7
    vr = rng.uniform(-max_rot, max_rot) * 0.7
    out = []
    tx = ty = th = 0.0
    for _ in range(n_frames):
        out.append((tx, ty, th))
        vx = float(np.clip(vx + rng.uniform(-0.3, 0.3), -max_t, max_t))
        vy = float(np.clip(vy + rng.uniform(-0.3, 0.3), -max_t, max_t))
        vr = float(np.clip(vr + rng.uniform(-0.02, 0.02), -max_rot, max_rot))
        tx, ty, th = tx + vx, ty + vy, th + vr
    return out


def sequence(seed, w, h, n_frames, noise=True):
    scene = make_scene(seed, w, h)
    mot = motion(seed, n_frames)
    # keep the accumulated translation inside the scene margin
    frames = []
    for i, (tx, ty, th) in enumerate(mot):
        tx = float(np.clip(tx, -MARGIN + 8, MARGIN - 8))
        ty = float(np.clip(ty, -MARGIN + 8, MARGIN - 8))
        frames.append(render(scene, w, h, tx, ty, th, noise_seed=(seed * 1000 + i) if noise else None))
    return np.stack(frames), mot
