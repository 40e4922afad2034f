"""CPU oracle of the keyframe point set-up.  TEST INFRASTRUCTURE ONLY.

numpy restatement of what the reference does once per keyframe with OpenCV:
    KeyFrame::create            src/tracking/KeyFrame.cpp:333-463   (normalise, log, Sobel, magnitude, select,
                                                                     norm_coord, grad, depth association, clean)
    KeyFrame::candidatePoints   src/tracking/KeyFrame.cpp:740-823   (20x20 cells; MAX = repeated arg-max with zeroing,
                                                                     MEDIAN = everything above the cell's median)
    utils::medianMat            src/utils/Utils.cpp:491-499         (nth_element at size/2)
    KeyFrame::setDepthMap       src/tracking/KeyFrame.cpp:1137-1198 (nearest depth-map point, distance -> weight)
    KeyFrame::cleanPoints       src/tracking/KeyFrame.cpp:1566-1587 (drop weight < 0.7, order preserved)

PARITY UNPINNED: OpenCV is not available here and the reference has no tests; cv::Sobel (ksize 3, CV_64F,
BORDER_REFLECT_101 default), cv::log, cv::minMaxLoc (first extremum in row-major order) and cv::cartToPolar
(sqrt(x^2 + y^2)) are restated from their documented behaviour.
"""
import numpy as np

LOG_EPS = float(np.float32(0.2))       # `static constexpr float log_eps = 0.2` (KeyFrame.hpp:54) promoted to double
MAX, MEDIAN = 0, 1                     # eds::tracking::CANDIDATE_POINT_METHOD


def _resize_coords(n_dst, n_src):
    """OpenCV resize (INTER_LINEAR) source taps and fp32 fraction per destination index (imgproc/resize.cpp)."""
    scale = n_src / n_dst
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0; s[lo] = 0
    hi = s >= n_src - 1
    f[hi] = 0.0; s[hi] = n_src - 1
    return s, np.minimum(s + 1, n_src - 1), f


def resize_cv_default(img, H, W):
    """``cv::resize(img, img, cv::Size(W, H), cv::INTER_CUBIC)`` as KeyFrame::create calls it (KeyFrame.cpp:355): INTER_CUBIC sits in
    the `fx` parameter, the interpolation is the default INTER_LINEAR — the 2x2 block mean when both scales are exactly 2 — with
    OpenCV's arithmetic for the element type: uint8 in fixed point (11-bit weights; vertical pass
    ``((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 >> 2``), float32 in fp32, float64 in fp64.  Channels are independent.
    Restated from OpenCV's published source; OpenCV is absent here (unpinned)."""
    img = np.asarray(img)
    if img.ndim == 3:
        return np.stack([resize_cv_default(img[:, :, k], H, W) for k in range(img.shape[2])], axis=2)
    sH, sW = img.shape
    if (sH, sW) == (H, W):
        return img.copy()
    if sH == 2 * H and sW == 2 * W:
        a, b, c, d = img[0::2, 0::2], img[0::2, 1::2], img[1::2, 0::2], img[1::2, 1::2]
        if img.dtype == np.uint8:
            return ((a.astype(np.int32) + b + c + d + 2) >> 2).astype(np.uint8)
        z = img.dtype.type(0)
        return ((((z + a) + b) + c) + d) * img.dtype.type(0.25)
    x0, x1, fx = _resize_coords(W, sW)
    y0, y1, fy = _resize_coords(H, sH)
    one = np.float32(1.0)
    if img.dtype == np.uint8:
        rs = lambda v: np.clip(np.rint(v), -32768, 32767).astype(np.int64)     # saturate_cast<short>: round half to even
        a0, a1 = rs((one - fx) * np.float32(2048.0)), rs(fx * np.float32(2048.0))
        b0, b1 = rs((one - fy) * np.float32(2048.0)), rs(fy * np.float32(2048.0))
        s = img.astype(np.int64)
        S0 = s[y0][:, x0] * a0[None, :] + s[y0][:, x1] * a1[None, :]
        S1 = s[y1][:, x0] * a0[None, :] + s[y1][:, x1] * a1[None, :]
        return ((((b0[:, None] * (S0 >> 4)) >> 16) + ((b1[:, None] * (S1 >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
    T = img.dtype.type
    a0, a1, b0, b1 = (one - fx).astype(T), fx.astype(T), (one - fy).astype(T), fy.astype(T)
    S0 = img[y0][:, x0] * a0[None, :] + img[y0][:, x1] * a1[None, :]
    S1 = img[y1][:, x0] * a0[None, :] + img[y1][:, x1] * a1[None, :]
    return S0 * b0[:, None] + S1 * b1[:, None]


def rgb_to_gray(img):
    """cv::cvtColor(img, img, cv::COLOR_RGB2GRAY) (KeyFrame.cpp:361): uint8 (R 4899 + G 9617 + B 1868 + 2^13) >> 14; float32
    R 0.299f + G 0.587f + B 0.114f evaluated left to right in fp32.  (No CV_64F colour images in OpenCV.)"""
    img = np.asarray(img)
    if img.ndim == 2:
        return img
    R, G, B = img[:, :, 0], img[:, :, 1], img[:, :, 2]
    if img.dtype == np.uint8:
        return ((R.astype(np.int32) * 4899 + G.astype(np.int32) * 9617 + B.astype(np.int32) * 1868 + (1 << 13)) >> 14).astype(np.uint8)
    assert img.dtype == np.float32
    return (R * np.float32(0.299) + G * np.float32(0.587)) + B * np.float32(0.114)


def prepare_image(img, H, W):
    """The image KeyFrame::create works on: resized to H x W when out_scale != 1, grey (KeyFrame.cpp:352-362)."""
    return rgb_to_gray(resize_cv_default(img, H, W))


def normalise_log(img):
    """KeyFrame.cpp:363-374: convertTo(CV_64F), (img - min)/(max - min), log(img + log_eps)."""
    img = np.asarray(img, dtype=np.float64)
    mn, mx = img.min(), img.max()
    return np.log((img - mn) / (mx - mn) + LOG_EPS)


def sobel3(L):
    """cv::Sobel(L, CV_64F, 1, 0, 3) and (0, 1, 3): kernels [-1 0 1] x [1 2 1]^T, border reflect-101, no scale.
    The association of the three column differences is fixed (top + 2*middle + bottom) so that the device code can
    reproduce the sums bit for bit."""
    p = np.pad(L, 1, mode="reflect")
    t, m, b = p[:-2], p[1:-1], p[2:]
    gx = ((t[:, 2:] - t[:, :-2]) + 2.0 * (m[:, 2:] - m[:, :-2])) + (b[:, 2:] - b[:, :-2])
    l, c, r = p[:, :-2], p[:, 1:-1], p[:, 2:]
    gy = ((l[2:] - l[:-2]) + 2.0 * (c[2:] - c[:-2])) + (r[2:] - r[:-2])
    return gx, gy


def sobel7(L):
    """cv::Sobel(L, CV_64F, 1, 0, 7) and (0, 1, 7) — the KeyFrame constructor's aperture (reference KeyFrame.cpp:239-240): separable
    kernels smooth [1 6 15 20 15 6 1] and derivative [-1 -4 -5 0 5 4 1] (cv::getSobelKernels), border reflect-101, no scale; row pass
    then column pass, centre term first and the symmetric pairs outwards (the order the device code reproduces bit for bit)."""
    p = np.pad(L, 3, mode="reflect")
    H, W = L.shape
    def col(j):                                 # column c + j - 3 of every padded row
        return p[:, j:j + W]
    rs = ((20.0 * col(3) + 15.0 * (col(4) + col(2))) + 6.0 * (col(5) + col(1))) + (col(6) + col(0))
    rd = (5.0 * (col(4) - col(2)) + 4.0 * (col(5) - col(1))) + (col(6) - col(0))
    def row(a, k):
        return a[k:k + H]
    gx = ((20.0 * row(rd, 3) + 15.0 * (row(rd, 4) + row(rd, 2))) + 6.0 * (row(rd, 5) + row(rd, 1))) + (row(rd, 6) + row(rd, 0))
    gy = (5.0 * (row(rs, 4) - row(rs, 2)) + 4.0 * (row(rs, 5) - row(rs, 1))) + (row(rs, 6) - row(rs, 0))
    return gx, gy


def magnitude(gx, gy):
    return np.sqrt(gx * gx + gy * gy)


def candidate_points(mag, cell=20, method=MEDIAN, num_points=0):
    """KeyFrame.cpp:740-823.  Returns integer pixel coordinates (x, y), N x 2, in the reference's push order."""
    H, W = mag.shape
    cells = [(x, y) for y in range(0, H - cell + 1, cell) for x in range(0, W - cell + 1, cell)]
    out = []
    if method == MAX:
        k = int(num_points) // len(cells)
        for (x0, y0) in cells:
            patch = mag[y0:y0 + cell, x0:x0 + cell].copy()
            for _ in range(k):
                mx, mn = patch.max(), patch.min()
                if mx == mn:
                    break
                loc = int(np.argmax(patch))                  # minMaxLoc: first maximum in row-major order
                out.append((x0 + loc % cell, y0 + loc // cell))
                patch.flat[loc] = 0.0
    else:
        for (x0, y0) in cells:
            patch = mag[y0:y0 + cell, x0:x0 + cell]
            med = np.sort(patch.ravel())[patch.size // 2]    # nth_element(size / 2)
            ys, xs = np.nonzero(patch > med)                 # row-major scan
            out.extend(zip((xs + x0).tolist(), (ys + y0).tolist()))
    return np.asarray(out, dtype=np.int64).reshape(-1, 2)


def set_depth_map(coord, depth_xy, depth_idp, min_depth, max_depth):
    """KeyFrame.cpp:1137-1198: idp of the nearest depth-map point and weight 1 - (d - min)/(max - min).
    Exact ties in distance resolve to the lowest depth-map index (the reference's KD-tree order is unspecified)."""
    n = len(coord)
    if depth_xy is None or len(depth_xy) == 0:
        return np.full(n, 1.0 / ((max_depth - min_depth) / 2.0)), np.ones(n)
    depth_xy = np.asarray(depth_xy, dtype=np.float64)
    c = np.asarray(coord, dtype=np.float64)
    idx = np.empty(n, dtype=np.int64)
    for s in range(0, n, 2048):                              # brute force, chunked
        dx = c[s:s + 2048, None, 0] - depth_xy[None, :, 0]
        dy = c[s:s + 2048, None, 1] - depth_xy[None, :, 1]
        idx[s:s + 2048] = np.argmin(dx * dx + dy * dy, axis=1)
    dx = depth_xy[idx, 0] - c[:, 0]
    dy = depth_xy[idx, 1] - c[:, 1]
    dist = np.sqrt(dx * dx + dy * dy)
    mn, mx = dist.min(), dist.max()
    w = 1.0 - ((dist - mn) / (mx - mn)) if mn != mx else np.ones(n)
    return np.asarray(depth_idp, dtype=np.float64)[idx], w


def keyframe(img, K, method=MEDIAN, num_points=0, cell=20, depth_xy=None, depth_idp=None, min_depth=1.0, max_depth=3.0,
             weight_threshold=0.7, sobel_ksize=3):
    """The arrays KeyFrame::create leaves behind for the tracker (index-aligned, after cleanPoints)."""
    fx, fy, cx, cy = K
    L = normalise_log(img)
    gx, gy = sobel7(L) if sobel_ksize == 7 else sobel3(L)
    mag = magnitude(gx, gy)
    pts = candidate_points(mag, cell, method, num_points)
    coord = pts.astype(np.float64)
    norm = np.stack([(coord[:, 0] - cx) / fx, (coord[:, 1] - cy) / fy], axis=1) if len(pts) else np.zeros((0, 2))
    grad = np.stack([gx[pts[:, 1], pts[:, 0]], gy[pts[:, 1], pts[:, 0]]], axis=1) if len(pts) else np.zeros((0, 2))
    idp, w = set_depth_map(coord, depth_xy, depth_idp, min_depth, max_depth) if len(pts) else (np.zeros(0), np.zeros(0))
    keep = ~(w < weight_threshold)
    return {"coord": coord[keep], "norm_coord": norm[keep], "grad": grad[keep], "idp": idp[keep], "weights": w[keep],
            "num_candidates": len(pts), "log_img": L, "gx": gx, "gy": gy, "mag": mag}
