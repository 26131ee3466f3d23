"""Oracle: ``cv2.findContours(mask, cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)`` + ``cv2.contourArea`` + ``cv2.boundingRect``
as the reference uses them (add_shadow.py:40-47, shadow_for_attack.py:30-35), restated on the CPU.

Test infrastructure only.  PARITY UNPINNED: OpenCV (opencv-python, version not pinned by the reference: it ships no requirements
file) is absent from this image and the reference holds no contour fixtures, so nothing here can be checked against cv2 itself.
What is restated is the published algorithm OpenCV implements -- S. Suzuki, K. Abe, "Topological structural analysis of digitized
binary images by border following", CVGIP 30 (1985), Algorithm 1 (``modules/imgproc/src/contours.cpp``) -- as plain Python
loops, for small masks:

* every nonzero pixel is foreground; the image is padded by one background pixel; raster scan; an outer border starts at a
  1-pixel whose left neighbour is 0, a hole border at a pixel >= 1 whose right neighbour is 0; the parent follows from the type
  of the last border met on the scan line (LNBD);
* RETR_EXTERNAL keeps the outer borders whose parent is the frame;
* ``contourArea`` is Green's formula over the border's pixel centres (CHAIN_APPROX_SIMPLE only drops collinear points, the
  polygon and its area are the same); ``boundingRect`` the inclusive box of those points;
* list order: OpenCV links every new contour in FRONT of the ones found before, so ``contours[0]`` is the LAST one the raster
  scan met (observed behaviour of cv2.findContours: bottom of the image first).

The product computes the same quantities without border following (csrc/contours.hip: background flood from the frame, component
labelling, a 2x2-cell area formula); tests compare the two, and both against ``scipy.ndimage.label``.
"""
import numpy as np

# clockwise neighbour ring starting east (image coordinates: i down, j right)
_RING = [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1)]


def _ring_index(di, dj):
    return _RING.index((di, dj))


def suzuki_borders(mask):
    """All borders of ``mask`` (2-D array, nonzero = foreground): list of dicts {points [(x, y)...], hole, parent, seq} in the
    order the raster scan meets them; border numbers start at 2 (1 is the frame), ``parent`` is a border number."""
    m = np.asarray(mask)
    H, W = m.shape
    f = np.zeros((H + 2, W + 2), dtype=np.int64)
    f[1:-1, 1:-1] = (m != 0).astype(np.int64)
    nbd = 1
    borders = {1: dict(hole=True, parent=0, points=[])}
    order = []
    for i in range(1, H + 1):
        lnbd = 1
        for j in range(1, W + 1):
            if f[i, j] == 0:
                continue
            start = None
            if f[i, j] == 1 and f[i, j - 1] == 0:
                nbd += 1
                start, hole = (i, j - 1), False
            elif f[i, j] >= 1 and f[i, j + 1] == 0:
                nbd += 1
                start, hole = (i, j + 1), True
                if f[i, j] > 1:
                    lnbd = int(f[i, j])
            if start is not None:
                bp = borders[lnbd]
                parent = (bp["parent"] if bp["hole"] == hole else lnbd)      # Table 1 of the paper
                pts = []
                # (3.1) clockwise from `start` around (i, j)
                k0 = _ring_index(start[0] - i, start[1] - j)
                found = None
                for s in range(8):
                    di, dj = _RING[(k0 + s) % 8]
                    if f[i + di, j + dj] != 0:
                        found = (i + di, j + dj)
                        break
                if found is None:
                    f[i, j] = -nbd
                    pts.append((j - 1, i - 1))
                else:
                    i1, j1 = found
                    i2, j2 = i1, j1
                    i3, j3 = i, j
                    while True:
                        # (3.3) counter-clockwise from the element after (i2, j2) around (i3, j3)
                        k = _ring_index(i2 - i3, j2 - j3)
                        east_zero_seen = False
                        for s in range(1, 9):
                            di, dj = _RING[(k - s) % 8]
                            if f[i3 + di, j3 + dj] != 0:
                                i4, j4 = i3 + di, j3 + dj
                                break
                            if (di, dj) == (0, 1):
                                east_zero_seen = True
                        # (3.4)
                        if east_zero_seen:
                            f[i3, j3] = -nbd
                        elif f[i3, j3] == 1:
                            f[i3, j3] = nbd
                        pts.append((j3 - 1, i3 - 1))
                        # (3.5)
                        if (i4, j4) == (i, j) and (i3, j3) == (i1, j1):
                            break
                        i2, j2, i3, j3 = i3, j3, i4, j4
                borders[nbd] = dict(hole=hole, parent=parent, points=pts, seq=nbd)
                order.append(nbd)
            if f[i, j] != 1:
                lnbd = abs(int(f[i, j]))
    return [borders[n] for n in order]


def contour_area2(points):
    """Twice ``cv2.contourArea``: |sum of x_{k-1} * y_k - x_k * y_{k-1}| over the closed polygon (exact integer)."""
    a = 0
    n = len(points)
    for k in range(n):
        x0, y0 = points[k - 1]
        x1, y1 = points[k]
        a += x0 * y1 - x1 * y0
    return abs(a)


def external_contours(mask):
    """[(x, y, w, h, 2 * area, first_pixel)] of the external contours in OpenCV's list order (last found first)."""
    m = np.asarray(mask)
    W = m.shape[1]
    out = []
    for b in suzuki_borders(m):
        if b["hole"] or b["parent"] != 1:
            continue
        xs = [p[0] for p in b["points"]]
        ys = [p[1] for p in b["points"]]
        x0, y0 = b["points"][0]                        # where the raster scan met the border
        out.append((min(xs), min(ys), max(xs) - min(xs) + 1, max(ys) - min(ys) + 1, contour_area2(b["points"]), y0 * W + x0))
    return out[::-1]


def cv_gray(mask):
    """``np.array(mask)`` then ``cv2.cvtColor(..., COLOR_RGB2GRAY)`` for 3-channel masks: OpenCV's 8-bit fixed point."""
    a = np.asarray(mask)
    if a.dtype == np.bool_:
        a = a.astype(np.uint8) * 255
    if a.ndim == 3:
        r, g, b = (a[..., k].astype(np.int64) for k in range(3))
        a = (r * 4899 + g * 9617 + b * 1868 + 8192) >> 14
    return a.astype(np.uint8)
