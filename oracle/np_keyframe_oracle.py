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
             weight_threshold=0.7):
    """The arrays KeyFrame::create leaves behind for the tracker (index-aligned, after cleanPoints)."""
    fx, fy, cx, cy = K
    L = normalise_log(img)
    gx, gy = sobel3(L)
    mag = magnitude(gx, gy)
    pts = candidate_points(mag, cell, method, num_points)
    coord = pts.astype(np.float64)
    norm = np.stack([(coord[:, 0] - cx) / fx, (coord[:, 1] - cy) / fy], axis=1) if len(pts) else np.zeros((0, 2))
    grad = np.stack([gx[pts[:, 1], pts[:, 0]], gy[pts[:, 1], pts[:, 0]]], axis=1) if len(pts) else np.zeros((0, 2))
    idp, w = set_depth_map(coord, depth_xy, depth_idp, min_depth, max_depth) if len(pts) else (np.zeros(0), np.zeros(0))
    keep = ~(w < weight_threshold)
    return {"coord": coord[keep], "norm_coord": norm[keep], "grad": grad[keep], "idp": idp[keep], "weights": w[keep],
            "num_candidates": len(pts), "log_img": L, "gx": gx, "gy": gy, "mag": mag}
