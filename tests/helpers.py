"""Seeded synthetic inputs shared by the tests, smoke() and bench.py (no reference data exists:
the reference ships no images, no model and no golden outputs -- SURVEY.md section 4)."""
import numpy as np

STRIDES = (32, 16, 8)
A = 2


def head_shapes(n, net_h, net_w):
    out = []
    for s in STRIDES:
        h, w = net_h // s, net_w // s
        out += [(n, 2 * A, h, w), (n, 4 * A, h, w), (n, 10 * A, h, w)]
    return out


def make_heads(seed, n, net_h=640, net_w=640, cand_rate=0.006, n_faces=0, quantize=0, delta_std=0.3):
    """Random head tensors in the reference's contract.  cand_rate = fraction of anchors whose fg
    score clears 0.7 at random; n_faces > 0 additionally plants clusters of high-score anchors that
    regress to common boxes (so NMS has real work); quantize > 0 rounds scores to multiples of
    1/quantize (exact ties exercise the stable-sort order)."""
    rng = np.random.default_rng(seed)
    heads = []
    for li, s in enumerate(STRIDES):
        h, w = net_h // s, net_w // s
        fg = rng.uniform(0.0, 0.7, size=(n, A, h, w)).astype(np.float32) * 0.999
        hot = rng.uniform(size=(n, A, h, w)) < cand_rate
        fg[hot] = rng.uniform(0.7, 1.0, size=int(hot.sum())).astype(np.float32)
        bbox = rng.normal(0, delta_std, size=(n, 4 * A, h, w)).astype(np.float32)
        lmk = rng.normal(0, 0.4, size=(n, 10 * A, h, w)).astype(np.float32)
        for b in range(n):
            for _ in range(n_faces):
                cy, cx = rng.integers(0, h), rng.integers(0, w)
                r = int(rng.integers(1, 3))
                y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, h), max(cx - r, 0), min(cx + r + 1, w)
                a = int(rng.integers(0, A))
                fg[b, a, y0:y1, x0:x1] = rng.uniform(0.72, 0.999, size=(y1 - y0, x1 - x0)).astype(np.float32)
                # deltas pointing (roughly) at the cluster centre: dx,dy proportional to the offset
                yy, xx = np.meshgrid(np.arange(y0, y1), np.arange(x0, x1), indexing="ij")
                aw = 16.0 * (32, 16, 8, 4, 2, 1)[2 * li + a]
                bbox[b, 4 * a + 0, y0:y1, x0:x1] = (cx - xx) * s / aw + rng.normal(0, 0.02, size=xx.shape)
                bbox[b, 4 * a + 1, y0:y1, x0:x1] = (cy - yy) * s / aw + rng.normal(0, 0.02, size=xx.shape)
                bbox[b, 4 * a + 2, y0:y1, x0:x1] = rng.normal(0, 0.05, size=xx.shape)
                bbox[b, 4 * a + 3, y0:y1, x0:x1] = rng.normal(0, 0.05, size=xx.shape)
        if quantize:
            fg = (np.round(fg * quantize) / quantize).astype(np.float32)
        cls = np.concatenate([1.0 - fg, fg], axis=1).astype(np.float32)
        heads += [cls, bbox.astype(np.float32), lmk]
    return heads


def make_image(seed, h, w, n_blobs=12):
    """Noise + bright ellipses, HxWx3 u8 (stands in for a decoded BGR frame)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n_blobs):
        cy, cx = rng.integers(0, h), rng.integers(0, w)
        ry, rx = rng.integers(max(h // 40, 2), max(h // 8, 4)), rng.integers(max(w // 40, 2), max(w // 8, 4))
        m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        img[m] = np.clip(img[m].astype(np.int32) // 4 + rng.integers(120, 200, size=3), 0, 255).astype(np.uint8)
    return img


def bf16_round(x):
    """float32 -> nearest-even bf16 -> float32 (numpy)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32).reshape(np.shape(x))


def bf16_bits_to_f32(u16):
    return (np.ascontiguousarray(u16, np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def make_face_kps(seed, h, w, scale_range=(0.6, 3.5), max_rot_deg=35.0):
    """Five key points = the 112x112 ArcFace template under a random similarity (scale, rotation, shift) somewhere in
    an h x w frame, plus a little per-point jitter; returns (kps [5,2] f32, box [5] f32)."""
    rng = np.random.default_rng(seed)
    tmpl = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366], [41.5493, 92.3655], [70.7299, 92.2041]])
    s = rng.uniform(*scale_range)
    th = np.deg2rad(rng.uniform(-max_rot_deg, max_rot_deg))
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    c = np.array([rng.uniform(0.1 * w, 0.9 * w), rng.uniform(0.1 * h, 0.9 * h)])
    kps = (tmpl - 56.0) @ R.T * s + c + rng.normal(0, 0.6, size=(5, 2))
    lo, hi = kps.min(0) - 25 * s, kps.max(0) + 25 * s
    box = np.array([lo[0], lo[1], hi[0], hi[1], 0.9], np.float32)
    return kps.astype(np.float32), box
