"""Keyframe point set-up: numpy oracle known answers on CPU, HIP kernels vs the oracle on GPU."""
import numpy as np
import pytest


def make_image(seed, H, W, dtype=np.uint8):
    """A smooth random texture with a few flat regions (zero gradient: exercises the MAX break rule and exact ties)."""
    rng = np.random.default_rng(seed)
    img = rng.standard_normal((H, W))
    for _ in range(3):                                           # cheap blur so gradients are not pure noise
        img = (img + np.roll(img, 1, 0) + np.roll(img, 1, 1) + np.roll(img, -1, 0) + np.roll(img, -1, 1)) / 5.0
    img = (img - img.min()) / (img.max() - img.min())
    img[H // 3:H // 3 + 45, W // 4:W // 4 + 50] = 0.5            # flat block covering whole cells
    img[:20, :20] = 0.25                                         # a flat first cell
    if dtype == np.uint8:
        return np.round(img * 255).astype(np.uint8)
    return img.astype(dtype)


def make_depth_map(seed, H, W, m):
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.uniform(0, W - 1, m), rng.uniform(0, H - 1, m)], axis=1)
    # leave a region without depth support so that cleanPoints has something to drop
    xy = xy[~((xy[:, 0] > 0.6 * W) & (xy[:, 1] > 0.5 * H))]
    return xy, rng.uniform(0.2, 1.0, len(xy))


# ---- oracle known answers (CPU) ---------------------------------------------------------------------------------
def test_oracle_log_and_sobel_known_answers():
    import np_keyframe_oracle as ko
    img = np.arange(12, dtype=np.float64).reshape(3, 4)
    L = ko.normalise_log(img)
    assert L[0, 0] == pytest.approx(np.log(0.2), rel=1e-7) and L[2, 3] == pytest.approx(np.log(1.2), rel=1e-7)
    ramp = np.add.outer(3.0 * np.arange(6), 2.0 * np.arange(7))               # 3 per row, 2 per column
    gx, gy = ko.sobel3(ramp)
    assert np.allclose(gx[1:-1, 1:-1], 8 * 2.0) and np.allclose(gy[1:-1, 1:-1], 8 * 3.0)     # [1 2 1] . [-1 0 1] = 8 x slope
    assert np.allclose(gx[:, 0], 0.0) and np.allclose(gy[0, :], 0.0)         # reflect-101: the border derivative vanishes
    assert ko.magnitude(np.array([3.0]), np.array([4.0]))[0] == 5.0
    # aperture 7 (KeyFrame.cpp:239-240): sum(smooth) = 64, sum(j * derivative) = 32 -> 2048 x slope on a ramp; symmetric, so a
    # constant image has no gradient; and the kernels are the ones cv::getSobelKernels builds: (1 1)^6 and (1 1)^5 * (-1 1)
    gx7, gy7 = ko.sobel7(np.add.outer(3.0 * np.arange(12), 2.0 * np.arange(13)))
    assert np.allclose(gx7[3:-3, 3:-3], 2048 * 2.0) and np.allclose(gy7[3:-3, 3:-3], 2048 * 3.0)
    assert np.all(ko.sobel7(np.full((9, 9), 7.0))[0] == 0.0)
    smooth, deriv = np.poly1d([1, 1]) ** 6, np.poly1d([1, 1]) ** 5 * np.poly1d([1, -1])
    assert list(smooth.coeffs) == [1, 6, 15, 20, 15, 6, 1] and list(deriv.coeffs) == [1, 4, 5, 0, -5, -4, -1]
    imp = np.zeros((15, 15)); imp[7, 7] = 1.0                      # impulse response = the (flipped) separable kernel
    gxi, gyi = ko.sobel7(imp)
    assert np.array_equal(gxi[4:11, 4:11], np.outer([1, 6, 15, 20, 15, 6, 1], [1, 4, 5, 0, -5, -4, -1]))
    assert np.array_equal(gyi[4:11, 4:11], np.outer([1, 4, 5, 0, -5, -4, -1], [1, 6, 15, 20, 15, 6, 1]))


def test_oracle_candidate_points_rules():
    import np_keyframe_oracle as ko
    mag = np.zeros((40, 65))                                     # 2 x 3 whole cells, 5 columns of remainder ignored
    mag[3, 7] = 5.0; mag[3, 9] = 5.0; mag[10, 2] = 7.0           # cell (0,0): a tie at 5 -> first in row-major first
    mag[25, 45] = 1.0                                            # cell (1,2)
    mag[5, 62] = 9.0                                             # outside every whole cell
    pts = ko.candidate_points(mag, 20, ko.MAX, num_points=6 * 3)
    assert pts.tolist() == [[2, 10], [7, 3], [9, 3], [5 + 40, 25]]          # (x, y); zeros are never picked (max == min)
    pts1 = ko.candidate_points(mag, 20, ko.MAX, num_points=6)
    assert pts1.tolist() == [[2, 10], [45, 25]]
    flat = np.full((20, 20), 3.0)
    assert len(ko.candidate_points(flat, 20, ko.MAX, num_points=5)) == 0      # constant cell: max == min at once
    rng = np.random.default_rng(0)
    m2 = rng.random((20, 40))
    med = ko.candidate_points(m2, 20, ko.MEDIAN)
    assert len(med) == 2 * 199                                   # 400 distinct values: 199 above the element at index 200
    assert (np.diff(med[:199, 1]) >= 0).all()                    # row-major inside the cell


def test_oracle_depth_association_and_clean():
    import np_keyframe_oracle as ko
    coord = np.array([[10.0, 10.0], [50.0, 10.0], [30.0, 40.0]])
    dxy = np.array([[11.0, 10.0], [50.0, 14.0], [100.0, 100.0]])
    idp, w = ko.set_depth_map(coord, dxy, np.array([0.5, 0.25, 0.125]), 1.0, 3.0)
    assert idp.tolist() == [0.5, 0.25, 0.25]
    d = np.array([1.0, 4.0, np.hypot(20.0, 26.0)])
    assert np.allclose(w, 1.0 - (d - d.min()) / (d.max() - d.min())) and w[0] == 1.0 and w[2] == 0.0
    idp0, w0 = ko.set_depth_map(coord, None, None, 1.0, 3.0)
    assert np.all(idp0 == 1.0) and np.all(w0 == 1.0)             # 1 / ((3 - 1)/2)
    img = make_image(5, 60, 80)
    xy, di = make_depth_map(6, 60, 80, 150)
    kf = ko.keyframe(img, (70.0, 70.0, 39.5, 29.5), ko.MAX, num_points=12 * 10, depth_xy=xy, depth_idp=di)
    assert 0 < len(kf["coord"]) < kf["num_candidates"] and kf["weights"].min() >= 0.7
    assert np.allclose(kf["norm_coord"][:, 0] * 70.0 + 39.5, kf["coord"][:, 0])


# ---- HIP vs oracle (GPU) ----------------------------------------------------------------------------------------
def _compare(out, ref):
    assert out["coord"].shape == ref["coord"].shape
    assert np.array_equal(out["coord"], ref["coord"])                          # integer pixels: exact
    assert np.array_equal(out["norm_coord"], ref["norm_coord"])
    assert np.abs(out["grad"] - ref["grad"]).max() <= 1e-12 * max(1.0, np.abs(ref["grad"]).max())    # device vs libm log: ulps
    assert np.array_equal(out["idp"], ref["idp"])
    assert np.abs(out["weights"] - ref["weights"]).max() <= 1e-14


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(480, 640), (181, 243)], ids=["vga", "odd"])
@pytest.mark.parametrize("method,npts", [(0, 2000), (0, 30000), (1, 0)], ids=["max2000", "max30000", "median"])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32], ids=["u8", "f32"])
def test_build_keyframe_vs_oracle(gpu, capi, shape, method, npts, dtype):
    import np_keyframe_oracle as ko
    H, W = shape
    img = make_image(17, H, W, dtype)
    K = (0.78 * W, 0.78 * W, (W - 1) / 2, (H - 1) / 2)
    xy, di = make_depth_map(18, H, W, 3000)
    ref = ko.keyframe(img, K, method, npts, depth_xy=xy, depth_idp=di)
    h = capi.Handle(capi.default_config(), 2, H * W, H, W)
    out = h.build_keyframe(1, img, K, method=method, num_points=npts, depth_xy=xy, depth_idp=di)
    _compare(out, ref)
    # without a depth map: constant initial depth, unit weights, nothing cleaned
    ref0 = ko.keyframe(img, K, method, npts, min_depth=0.5, max_depth=4.5)
    out0 = h.build_keyframe(0, img, K, method=method, num_points=npts, min_depth=0.5, max_depth=4.5)
    _compare(out0, ref0)
    assert len(out0["coord"]) == ref0["num_candidates"] and np.all(out0["idp"] == 0.5)
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("method,npts", [(0, 2000), (1, 0)], ids=["max2000", "median"])
def test_build_keyframe_sobel7_vs_oracle(gpu, capi, method, npts):
    """eds_kf_select.sobel_ksize = 7: the aperture of the reference's KeyFrame constructor (KeyFrame.cpp:239-240)."""
    import np_keyframe_oracle as ko
    H, W = 181, 243
    img = make_image(23, H, W, np.float32)
    K = (0.78 * W, 0.78 * W, (W - 1) / 2, (H - 1) / 2)
    xy, di = make_depth_map(24, H, W, 2000)
    ref = ko.keyframe(img, K, method, npts, depth_xy=xy, depth_idp=di, sobel_ksize=7)
    h = capi.Handle(capi.default_config(), 1, H * W, H, W)
    out = h.build_keyframe(0, img, K, method=method, num_points=npts, depth_xy=xy, depth_idp=di, sobel_ksize=7)
    _compare(out, ref)
    ref3 = ko.keyframe(img, K, method, npts, depth_xy=xy, depth_idp=di)
    assert np.median(np.abs(ref["grad"])) > 50 * np.median(np.abs(ref3["grad"]))          # a large constant factor, as SURVEY's appendix notes
    h.close()


@pytest.mark.gpu
def test_built_keyframe_equals_uploaded_keyframe(gpu, capi, synth, po):
    """The slot filled on the device behaves exactly like one filled by set_keyframe with the same arrays."""
    al = synth.make_alignment(91, H=240, W=320, N=500)
    img = make_image(23, al.H, al.W)
    K = (al.fx, al.fy, al.cx, al.cy)
    xy, di = make_depth_map(24, al.H, al.W, 2000)
    h = capi.Handle(capi.default_config(exec=capi.EXEC_HOST), 2, 4096, al.H, al.W)
    out = h.build_keyframe(0, img, K, method=capi.KF_MAX, num_points=2000, depth_xy=xy, depth_idp=di)
    N = len(out["idp"])
    assert 500 < N <= 2000
    h.set_keyframe(1, out["norm_coord"], out["grad"], out["idp"], out["weights"], *K)
    for s in (0, 1):
        h.set_event_frame(s, al.frame)
    rng = np.random.default_rng(1)
    p, q = 0.002 * rng.standard_normal(3), synth.quat_from_axis_angle(rng.standard_normal(3), 0.003)
    a, b = h.eval(0, p, q, al.v_true, ncols=12), h.eval(1, p, q, al.v_true, ncols=12)
    assert np.array_equal(a["r"], b["r"]) and np.array_equal(a["J"], b["J"]) and np.array_equal(a["JtJ"], b["JtJ"])
    # and like the oracle fed with the same arrays
    al2 = type(al)(**{**al.__dict__, "norm_coord": out["norm_coord"], "grad": out["grad"], "idp": out["idp"],
                      "weights": out["weights"], "coord": out["coord"]})
    e = po.Oracle(al2).eval12(p, q, al.v_true)
    assert np.abs(a["r"] - e["r_raw"]).max() <= 1e-5 * np.abs(e["r_raw"]).max()
    # a solve on the device-built keyframe runs
    h.set_config(capi.default_config(exec=capi.EXEC_DEVICE, solver=capi.SOLVER_LM6, max_num_iterations=5))
    _, _, _, info = h.optimize(0, p=al.p0, q=al.q0, v=al.v_true)
    assert info["success"] and info["num_points"] == N
    h.close()


@pytest.mark.gpu
def test_build_keyframe_errors(gpu, capi):
    H, W = 60, 80
    img = make_image(3, H, W)
    K = (70.0, 70.0, 39.5, 29.5)
    h = capi.Handle(capi.default_config(), 1, 50, H, W)
    with pytest.raises(capi.EdsError):
        h.build_keyframe(0, img, K, method=capi.KF_MEDIAN)               # ~half the pixels: far more than max_points = 50
    with pytest.raises(capi.EdsError):
        h.build_keyframe(0, img, K, method=capi.KF_MAX, num_points=5)    # 5 / 12 cells = 0 per cell: no candidate
    with pytest.raises(capi.EdsError):
        h.build_keyframe(0, img, K, cell=64)
    with pytest.raises(capi.EdsError):
        h.build_keyframe(0, img[:, :40], K)
    out = h.build_keyframe(0, img, K, method=capi.KF_MAX, num_points=36)     # 3 per cell
    assert 0 < len(out["idp"]) <= 36
    h.close()


@pytest.mark.gpu
def test_keyframe_mirror_create(gpu, capi):
    """KeyFrame.create of the Python mirror follows the reference's method choice (KeyFrame.cpp:406-411)."""
    import importlib
    import np_keyframe_oracle as ko
    trk = importlib.import_module("slam-eds_amd.tracker")
    H, W = 120, 160
    img = make_image(31, H, W)
    K = np.array([[125.0, 0, 79.5], [0, 125.0, 59.5], [0, 0, 1]])
    xy, di = make_depth_map(32, H, W, 800)
    kf = trk.KeyFrame.create(img, K, xy, di, trk.SELECT_MAX, percent_points=5.0)
    ref = ko.keyframe(img, (125.0, 125.0, 79.5, 59.5), ko.MAX, int(H * W * 0.05), depth_xy=xy, depth_idp=di)
    assert np.array_equal(kf.coord, ref["coord"]) and np.array_equal(kf.inv_depth, ref["idp"]) and kf.rows == H and kf.cols == W
    kf0 = trk.KeyFrame.create(img, K, xy, di, trk.SELECT_MAX, percent_points=0.0)      # no target: MEDIAN
    ref0 = ko.keyframe(img, (125.0, 125.0, 79.5, 59.5), ko.MEDIAN, 0, depth_xy=xy, depth_idp=di)
    assert np.array_equal(kf0.coord, ref0["coord"])


# ---- image preparation: out_scale != 1 and colour input (KeyFrame.cpp:352-362) ---------------------------------------------
def test_oracle_image_preparation_known_answers():
    import np_keyframe_oracle as ko
    rng = np.random.default_rng(5)
    u8 = rng.integers(0, 256, (12, 16), dtype=np.uint8)
    half = ko.resize_cv_default(u8, 6, 8)                        # exact factor 2: rounded block mean
    blk = u8.reshape(6, 2, 8, 2).astype(np.int32).sum(axis=(1, 3))
    assert half.dtype == np.uint8 and np.array_equal(half, (blk + 2) >> 2)
    assert np.array_equal(ko.resize_cv_default(u8, 12, 16), u8)
    const = np.full((9, 13), 77, np.uint8)
    assert np.all(ko.resize_cv_default(const, 6, 7) == 77)       # fixed-point weights sum to 2048: constants survive
    f32 = rng.standard_normal((9, 12)).astype(np.float32)
    r = ko.resize_cv_default(f32, 6, 8)
    assert r.dtype == np.float32 and r.shape == (6, 8) and r[0, 0] != f32[0, 0]
    rgb = np.zeros((2, 2, 3), np.uint8); rgb[..., 0] = 255
    assert np.all(ko.rgb_to_gray(rgb) == (255 * 4899 + (1 << 13)) >> 14)      # 76
    white = np.full((2, 2, 3), 255, np.uint8)
    assert np.all(ko.rgb_to_gray(white) == 255)                  # 4899 + 9617 + 1868 = 2^14
    g = ko.rgb_to_gray(np.ones((2, 2, 3), np.float32))
    assert g.dtype == np.float32 and np.allclose(g, 1.0, atol=1e-6)
    assert ko.prepare_image(rng.integers(0, 256, (20, 30, 3), dtype=np.uint8), 10, 15).shape == (10, 15)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.float64], ids=["u8", "f32", "f64"])
@pytest.mark.parametrize("src,channels", [((240, 320), 1), ((240, 320), 3), ((180, 240), 3), ((120, 160), 3), ((100, 130), 1)],
                         ids=["half-grey", "half-rgb", "x1.5-rgb", "same-rgb", "up-grey"])
def test_build_keyframe_from_camera_image(gpu, capi, dtype, src, channels):
    """eds_trk_build_keyframe_image: resize (out_scale != 1) and RGB -> grey on the device, then the usual set-up; against the
    numpy restatement of OpenCV's per-type arithmetic."""
    import np_keyframe_oracle as ko
    if dtype == np.float64 and channels == 3:
        h = capi.Handle(capi.default_config(), 1, 1000, 120, 160)
        with pytest.raises(capi.EdsError):                       # cv::cvtColor has no CV_64F colour path
            h.build_keyframe(0, np.zeros(src + (3,), np.float64), (100.0, 100.0, 80.0, 60.0))
        h.close()
        return
    H, W = 120, 160
    sH, sW = src
    base = make_image(23, sH, sW, np.float64)
    if channels == 3:
        rng = np.random.default_rng(2)
        img = np.stack([base, np.roll(base, 3, 1) * 0.8 + 0.1, np.clip(base + 0.1 * rng.standard_normal(base.shape), 0, 1)], axis=2)
    else:
        img = base
    img = np.round(img * 255).astype(np.uint8) if dtype == np.uint8 else img.astype(dtype)
    K = (0.78 * W, 0.78 * W, (W - 1) / 2, (H - 1) / 2)
    xy, di = make_depth_map(24, H, W, 800)
    grey = ko.prepare_image(img, H, W)
    assert grey.shape == (H, W) and grey.dtype == img.dtype
    ref = ko.keyframe(grey, K, 0, 1500, depth_xy=xy, depth_idp=di)
    h = capi.Handle(capi.default_config(), 1, H * W, H, W)
    out = h.build_keyframe(0, img, K, method=0, num_points=1500, depth_xy=xy, depth_idp=di)
    _compare(out, ref)
    h.close()
