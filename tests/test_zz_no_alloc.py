"""vsf_lk_points_noalloc (the entry point that allocates nothing; include/vsf.h) against the existing entry point
vsf_debug_lk, through the C ABI, bit for bit: two calls in a row, the second on a SMALLER frame with fewer points
(every call on a GPU shares that GPU's scratch array, which keeps the last call's contents), then a third at the
first size again; plus the size limit and the argument checks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_both(vsf, oracle, synth, torch, seed, w, h, win, ml, n, pad=0):
    frames, _ = synth.sequence(seed, w, h, 2)
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)], 1).astype(np.float32)
    p = vsf.default_params(width=w, height=h, klt_win=win, klt_max_level=ml)
    ctx = vsf.Context(p)
    ref, rst = ctx.debug_lk(frames[0], frames[1], pts)          # the existing entry point
    ctx.close()
    # device-resident inputs and outputs, rows `w + pad` apart
    buf = np.zeros((2, h, w + pad), np.uint8)
    buf[:, :, :w] = frames
    d_fr = torch.from_numpy(buf).cuda()
    d_pts = torch.from_numpy(pts).cuda()
    d_out = torch.full((n, 2), -7.0, dtype=torch.float32, device="cuda")
    d_st = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    vsf.lk_points_noalloc(p, d_fr[0].data_ptr(), w + pad, d_fr[1].data_ptr(), w + pad, d_pts.data_ptr(),
                          d_out.data_ptr(), d_st.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got, gst = d_out.cpu().numpy(), d_st.cpu().numpy()
    assert np.array_equal(gst, rst)
    ok = rst == 1
    assert ok.sum() > n // 4
    assert np.array_equal(got[ok].view(np.uint32), ref[ok].view(np.uint32))
    # ... and against the CPU oracle directly (exact-sum mode: bit for bit), not only through vsf_debug_lk
    oref, ost, _ = oracle.calc_optical_flow_pyr_lk(frames[0], frames[1], pts, win, ml, 30, 0.01, acc_mode=oracle.ACC_EXACT)
    assert np.array_equal(gst, ost)
    assert np.array_equal(got[ok].view(np.uint32), oref[ok].view(np.uint32))


def test_noalloc_matches_debug_lk_two_calls_in_a_row(vsf, oracle, synth):
    import torch
    _run_both(vsf, oracle, synth, torch, 3, 752, 480, 21, 3, 500)
    _run_both(vsf, oracle, synth, torch, 4, 320, 240, 15, 2, 37, pad=8)     # smaller frame, fewer points, strided rows
    _run_both(vsf, oracle, synth, torch, 5, 752, 480, 21, 3, 300)
    _run_both(vsf, oracle, synth, torch, 6, 401, 303, 31, 3, 64)            # unaligned width: byte-wise pyramid kernel


def test_noalloc_allocates_nothing_and_replays_from_a_graph(vsf, oracle, synth):
    """The property the entry point exists for: device memory in use does not move across calls (hipMemGetInfo
    before / after), and the call can be captured into a HIP graph and replayed with the same bits."""
    import torch
    w, h, n = 640, 360, 257
    frames, _ = synth.sequence(8, w, h, 2)
    rng = np.random.default_rng(8)
    pts = np.stack([rng.uniform(10, w - 10, n), rng.uniform(10, h - 10, n)], 1).astype(np.float32)
    p = vsf.default_params(width=w, height=h)
    d_fr = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
    d_pts = torch.from_numpy(pts).cuda()
    d_out = torch.zeros((n, 2), dtype=torch.float32, device="cuda")
    d_st = torch.zeros((n,), dtype=torch.uint8, device="cuda")

    def call(stream):
        vsf.lk_points_noalloc(p, d_fr[0].data_ptr(), w, d_fr[1].data_ptr(), w, d_pts.data_ptr(), d_out.data_ptr(),
                              d_st.data_ptr(), n, stream)

    call(torch.cuda.current_stream().cuda_stream)   # first call on the device: takes the scratch array's address
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        call(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, "device memory in use changed across calls: %d -> %d" % (free0, free1)
    ref_out, ref_st = d_out.cpu().numpy().copy(), d_st.cpu().numpy().copy()
    oref, ost, _ = oracle.calc_optical_flow_pyr_lk(frames[0], frames[1], pts, 21, 3, 30, 0.01, acc_mode=oracle.ACC_EXACT)
    assert np.array_equal(ref_st, ost) and np.array_equal(ref_out[ost == 1].view(np.uint32), oref[ost == 1].view(np.uint32))
    # captured and replayed: allocation during capture would fail the capture
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            call(side.cuda_stream)
    d_out.zero_()
    d_st.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(d_st.cpu().numpy(), ref_st)
    assert np.array_equal(d_out.cpu().numpy()[ref_st == 1].view(np.uint32), ref_out[ref_st == 1].view(np.uint32))


def test_noalloc_limits_and_arguments(vsf):
    import ctypes as C
    import torch
    L = vsf.lib()
    d = torch.zeros(64, dtype=torch.uint8, device="cuda")
    big = vsf.default_params(width=3840, height=2160, klt_max_level=4)   # two 4K pyramids exceed the 4 MiB array
    args = (C.c_void_p(d.data_ptr()), 3840, C.c_void_p(d.data_ptr()), 3840, C.c_void_p(d.data_ptr()),
            C.c_void_p(d.data_ptr()), C.c_void_p(d.data_ptr()), 1, C.c_void_p(0))
    assert L.vsf_lk_points_noalloc(C.byref(big), *args) == vsf.VSF_E_CAPACITY
    ok = vsf.default_params(width=64, height=48, klt_win=9)
    assert L.vsf_lk_points_noalloc(C.byref(ok), C.c_void_p(0), 64, C.c_void_p(d.data_ptr()), 64,
                                   C.c_void_p(d.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(d.data_ptr()), 1,
                                   C.c_void_p(0)) == vsf.VSF_E_INVALID
    assert L.vsf_lk_points_noalloc(C.byref(ok), C.c_void_p(d.data_ptr()), 32, C.c_void_p(d.data_ptr()), 64,
                                   C.c_void_p(d.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(d.data_ptr()), 1,
                                   C.c_void_p(0)) == vsf.VSF_E_INVALID   # stride < width
    assert L.vsf_lk_points_noalloc(C.byref(ok), C.c_void_p(d.data_ptr()), 64, C.c_void_p(d.data_ptr()), 64,
                                   C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), 0, C.c_void_p(0)) == vsf.VSF_OK  # n = 0
