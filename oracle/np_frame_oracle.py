"""CPU oracle of the event-frame construction — TEST INFRASTRUCTURE ONLY.

fp64 numpy restatement of reference ``EventFrame::create`` for ``out_scale == 1``
(src/tracking/EventFrame.cpp:302-389) and of its helper
``eds::utils::drawValuesPoints(points, values, H, W, "bilinear", 0.5, true)`` (src/utils/Utils.cpp:50-122), plus the
published behaviour of the OpenCV calls it makes (``cv::GaussianBlur`` with ``ksize = (3, 3)``, ``sigma = 0.5`` and the
default reflect-101 border; ``cv::dilate`` / ``cv::erode`` with a rectangular element and the default border value, which
ignores pixels outside the image; ``cv::norm`` = Frobenius).  PARITY UNPINNED: the reference has no tests or fixtures and
OpenCV is absent from this image.
"""
from __future__ import annotations

import numpy as np


def exp_weight(idx, window_size=1.0):
    """eds::utils::expWeight (src/utils/Utils.hpp:542-546)."""
    value = (idx - window_size / 2.0) / (window_size / 6.0)
    return np.exp(-0.5 * value * value)


def draw_values_points(ux, uy, values, H, W, s=0.5, use_exp_weights=True):
    """drawValuesPoints(..., "bilinear", s, use_exp_weights) — Utils.cpp:50-122."""
    n = len(values)
    img = np.zeros((H, W))
    if n:
        idx = np.arange(n, dtype=np.float64)
        weight = exp_weight(idx / n, 1.0) if use_exp_weights else np.ones(n)     # :66  idx / window_size
        x0 = np.floor(ux).astype(np.int64)
        y0 = np.floor(uy).astype(np.int64)
        x1, y1 = x0 + 1, y0 + 1
        inside = lambda x, y: (x < W) & (y < H) & (x >= 0) & (y >= 0)
        wa = np.where(inside(x0, y0), (x1 - ux) * (y1 - uy), 0.0)                 # :92-95
        wb = np.where(inside(x0, y1), (x1 - ux) * (uy - y0), 0.0)
        wc = np.where(inside(x1, y0), (ux - x0) * (y1 - uy), 0.0)
        wd = np.where(inside(x1, y1), (ux - x0) * (uy - y0), 0.0)
        cx0, cx1 = np.clip(x0, 0, W - 1), np.clip(x1, 0, W - 1)                   # :97-100
        cy0, cy1 = np.clip(y0, 0, H - 1), np.clip(y1, 0, H - 1)
        val = weight * values
        np.add.at(img, (cy0, cx0), val * wa)                                      # :103-106
        np.add.at(img, (cy1, cx0), val * wb)
        np.add.at(img, (cy0, cx1), val * wc)
        np.add.at(img, (cy1, cx1), val * wd)
    if s > 0:
        img = gaussian_blur_3x3(img, s)                                           # :113-119: k_w = int(1.25*240/100) = 3 = k_h
    return img


def gaussian_blur_3x3(img, sigma):
    """cv::GaussianBlur(img, img, Size(3, 3), sigma, sigma): separable, kernel exp(-x^2/(2 sigma^2)) normalised
    (cv::getGaussianKernel with sigma > 0), BORDER_REFLECT_101."""
    t = np.exp(-0.5 / (sigma * sigma))
    k = np.array([t, 1.0, t]) / (1.0 + 2.0 * t)
    pad = np.pad(img, ((0, 0), (1, 1)), mode="reflect")
    rows = k[0] * pad[:, :-2] + k[1] * pad[:, 1:-1] + k[2] * pad[:, 2:]
    pad = np.pad(rows, ((1, 1), (0, 0)), mode="reflect")
    return k[0] * pad[:-2, :] + k[1] * pad[1:-1, :] + k[2] * pad[2:, :]


def morph_level(img, i):
    """EventFrame.cpp:350-357: dilate + erode with a (2i+1)^2 rectangle anchored at its centre."""
    from scipy.ndimage import maximum_filter, minimum_filter
    k = 2 * i + 1
    return maximum_filter(img, size=k, mode="constant", cval=-np.inf) + minimum_filter(img, size=k, mode="constant", cval=np.inf)


def event_frame(x, y, polarity, H, W, mapx=None, mapy=None, level=0, sigma=0.5, use_exp_weights=True):
    """Returns (frame / ||frame||_F as float64 H x W, ||frame||_F) — EventFrame::event_frame[level], norm[level]."""
    x = np.asarray(x, dtype=np.int64)
    y = np.asarray(y, dtype=np.int64)
    if mapx is not None:
        ux = np.asarray(mapx, dtype=np.float32)[y, x].astype(np.float64)          # EventFrame.cpp:316-317
        uy = np.asarray(mapy, dtype=np.float32)[y, x].astype(np.float64)
    else:
        ux, uy = x.astype(np.float64), y.astype(np.float64)
    pol = np.where(np.asarray(polarity) != 0, 1.0, -1.0)                          # :318
    img = draw_values_points(ux, uy, pol, H, W, sigma, use_exp_weights)           # :339
    if level > 0:
        img = morph_level(img, level)
    norm = np.linalg.norm(img)                                                    # :359-364 cv::norm
    return img / norm, norm


def resize_cv_default(img, H, W):
    """What ``cv::resize(img, img, cv::Size(W, H), cv::INTER_CUBIC)`` does AS THE REFERENCE WRITES IT (EventFrame.cpp:345,
    KeyFrame.cpp:355): the fourth positional parameter of cv::resize is ``fx``, so the interpolation stays at its default
    INTER_LINEAR — and OpenCV's resize() replaces INTER_LINEAR by the 2x2 block average of its INTER_AREA fast path when both
    scales are exactly 2.  Published OpenCV behaviour (modules/imgproc/src/resize.cpp), restated for CV_64F images: source
    coordinate ``fx = float((dx + 0.5) * scale - 0.5)``, ``sx = floor(fx)``, fraction and both weights in fp32, taps clamped to
    the image; horizontal pass, then vertical pass, in fp64.  OpenCV is absent from this image: unpinned."""
    img = np.asarray(img, dtype=np.float64)
    sH, sW = img.shape
    if sH == 2 * H and sW == 2 * W:
        a, b = img[0::2, 0::2], img[0::2, 1::2]
        c, d = img[1::2, 0::2], img[1::2, 1::2]
        return ((((0.0 + a) + b) + c) + d) * 0.25

    def coords(n_dst, n_src):
        scale = n_src / n_dst
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        f[lo] = 0.0; s[lo] = 0
        hi = s >= n_src - 1
        f[hi] = 0.0; s[hi] = n_src - 1
        return s, np.minimum(s + 1, n_src - 1), (np.float32(1.0) - f).astype(np.float64), f.astype(np.float64)

    x0, x1, a0, a1 = coords(W, sW)
    y0, y1, b0, b1 = coords(H, sH)
    h0 = img[y0][:, x0] * a0[None, :] + img[y0][:, x1] * a1[None, :]
    h1 = img[y1][:, x0] * a0[None, :] + img[y1][:, x1] * a1[None, :]
    return h0 * b0[:, None] + h1 * b1[:, None]


def event_frames(x, y, polarity, sensor_H, sensor_W, H, W, num_levels, mapx=None, mapy=None, sigma=0.5, use_exp_weights=True):
    """EventFrame::create as a whole (EventFrame.cpp:302-389): the `num_levels` normalised frames of one event slice and their
    norms; events and LUT at the sensor's size, frames at H x W (resized when out_scale != 1, :342-346)."""
    x = np.asarray(x, dtype=np.int64)
    y = np.asarray(y, dtype=np.int64)
    if mapx is not None:
        ux = np.asarray(mapx, dtype=np.float32)[y, x].astype(np.float64)
        uy = np.asarray(mapy, dtype=np.float32)[y, x].astype(np.float64)
    else:
        ux, uy = x.astype(np.float64), y.astype(np.float64)
    pol = np.where(np.asarray(polarity) != 0, 1.0, -1.0)
    img = draw_values_points(ux, uy, pol, sensor_H, sensor_W, sigma, use_exp_weights)     # :339
    if (sensor_H, sensor_W) != (H, W):
        img = resize_cv_default(img, H, W)                                                # :342-346
    frames, norms = [], []
    for i in range(num_levels):                                                           # :348-357
        lv = img if i == 0 else morph_level(img, i)
        n = np.linalg.norm(lv)
        frames.append(lv / n); norms.append(n)
    return frames, norms
