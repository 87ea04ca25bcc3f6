"""Frontend step around the path (KimeraFrontend.cpp:247-304: bearings, aged count, keyframe decision,
age increment).  CPU: oracle known answers; GPU: bit-exact against the oracle through the C ABI."""
import numpy as np
import pytest

EUROC_DS = dict(fx=349.7560023050409, fy=348.72454229977037, cx=365.89440762590149, cy=249.32995565708704,
                xi=-0.24425478822495668, alpha=0.57657248532475368)


def test_oracle_double_sphere_known_answers(oracle):
    cam = oracle.CameraDS(**EUROC_DS)
    # principal point -> optical axis; unit length everywhere; inverse of the model's projection
    b, ok = oracle.ds_unproject(cam, [[cam.cx, cam.cy]])
    assert ok[0] and np.allclose(b[0], [0, 0, 1], atol=1e-15)
    rng = np.random.default_rng(0)
    xy = np.stack([rng.uniform(0, 752, 500), rng.uniform(0, 480, 500)], 1)
    b, ok = oracle.ds_unproject(cam, xy)
    assert ok.all() and np.allclose(np.linalg.norm(b, axis=1), 1.0, atol=1e-12)
    # project back (Usenko et al. eq. 40-45) in numpy
    x, y, z = b.T
    d1 = np.sqrt(x * x + y * y + z * z)
    d2 = np.sqrt(x * x + y * y + (cam.xi * d1 + z) ** 2)
    den = cam.alpha * d2 + (1 - cam.alpha) * (cam.xi * d1 + z)
    assert np.allclose(np.stack([cam.fx * x / den + cam.cx, cam.fy * y / den + cam.cy], 1), xy, atol=1e-9)
    # alpha > 0.5: pixels beyond r2 = 1 / (2 alpha - 1) are outside the model
    far = [[cam.cx + cam.fx * 3.0, cam.cy]]
    assert not oracle.ds_unproject(cam, far)[1][0]
    pin = oracle.CameraDS(fx=400, fy=400, cx=320, cy=240, xi=0.0, alpha=0.0)  # pinhole limit
    b, ok = oracle.ds_unproject(pin, [[420.0, 140.0]])
    assert ok[0] and np.allclose(b[0] / b[0][2], [0.25, -0.25, 1.0], atol=1e-14)


def test_oracle_keyframe_rules(oracle):
    cam = oracle.CameraDS(**EUROC_DS)
    xy = np.full((10, 2), 100.0)
    args = (2, 5, 0.5, 5.0)  # counting age, low number, min interval, interval
    age = np.array([1, 2, 3, 1, 1, 2, 1, 1, 1, 1], np.uint32)
    kf, _, aged = oracle.frontend_post_track(cam, xy, age.copy(), *args, 1.0)
    assert aged == 3 and kf                                   # 3 < 5 aged features, interval satisfied
    kf, _, _ = oracle.frontend_post_track(cam, xy, age.copy(), *args, 0.4)
    assert not kf                                             # too soon after the last keyframe
    old = np.full(10, 4, np.uint32)
    kf, _, aged = oracle.frontend_post_track(cam, xy, old, *args, 1.0)
    assert aged == 10 and not kf and (old == 4).all()         # enough aged features, ages untouched
    kf, _, _ = oracle.frontend_post_track(cam, xy, old, *args, 5.5)
    assert kf and (old == 5).all()                            # interval exceeded: keyframe, ages incremented
    # counting_feature_age is int32 compared with uint32 ages: a negative value never counts
    kf, _, aged = oracle.frontend_post_track(cam, xy, age.copy(), -1, 5, 0.5, 5.0, 1.0)
    assert aged == 0


@pytest.mark.gpu
def test_frontend_post_track_bit_exact(vsf, oracle):
    import torch
    B = 6
    ctx = vsf.Context(vsf.default_params(width=752, height=480, streams=B, max_features=300))
    cap = ctx.capacity
    rng = np.random.default_rng(5)
    ids = np.zeros((B, cap), np.uint64)
    xy = np.zeros((B, cap, 2))
    age = np.zeros((B, cap), np.uint32)
    n = rng.integers(0, cap, B).astype(np.int32)
    n[0], n[1] = 0, cap
    for s in range(B):
        ids[s, :n[s]] = np.arange(n[s])
        xy[s, :n[s]] = np.stack([rng.uniform(-400, 1200, n[s]), rng.uniform(-300, 800, n[s])], 1)
        age[s, :n[s]] = rng.integers(1, 4, n[s])
    age[2, :n[2]] = 3
    ctx.batch_upload(ids, xy, age, n)
    cam_args = EUROC_DS
    cam, ocam = vsf.VsfCameraDS(**cam_args), oracle.CameraDS(**cam_args)
    kp = vsf.keyframe_params(low_keyframe_feature_number=150)
    dt = np.array([1.0, 0.2, 1.0, 6.0, 0.6, 4.9])
    bearings = torch.zeros((B, cap, 3), dtype=torch.float64, device="cuda")
    kf, aged = ctx.frontend_post_track(cam, kp, dt, bearings.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _, _, age2, n2, _ = ctx.batch_download()
    got = bearings.cpu().numpy()
    assert np.array_equal(n2, n)
    for s in range(B):
        a = age[s, :n[s]].copy()
        rkf, rb, raged = oracle.frontend_post_track(ocam, xy[s, :n[s]], a, 2, 150, 0.5, 5.0, dt[s])
        assert bool(kf[s]) == rkf and aged[s] == raged
        assert np.array_equal(got[s, :n[s]].view(np.uint64), rb.view(np.uint64))
        assert np.array_equal(age2[s, :n[s]], a)
    invalid = ~oracle.ds_unproject(ocam, xy[1, :n[1]])[1]
    assert invalid.any() and (got[1, :n[1]][invalid] == [0, 0, 1]).all()
    b, ok = ctx.unproject(cam, xy[1, :n[1]])
    rb, rok = oracle.ds_unproject(ocam, xy[1, :n[1]])
    assert np.array_equal(ok, rok) and np.array_equal(b[ok].view(np.uint64), rb[rok].view(np.uint64))
