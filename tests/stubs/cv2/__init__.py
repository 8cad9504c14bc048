"""Stand-in for the two OpenCV calls the reference's WarpFrame makes (see ../README.md).  Not OpenCV's code: `resize` with
INTER_AREA is the area-weighted mean by definition in exact integer arithmetic, round half up."""
import numpy as np

INTER_AREA = 3
COLOR_RGB2GRAY = 7


class ocl:
    @staticmethod
    def setUseOpenCL(flag):
        pass


def _overlap(src, out):
    """m[o, s] = overlap of output cell o with source pixel s, in units of 1/out source pixels"""
    m = np.zeros((out, src), np.int64)
    for o in range(out):
        lo, hi = o * src, (o + 1) * src
        for s in range(lo // out, src):
            if s * out >= hi:
                break
            m[o, s] = min(hi, (s + 1) * out) - max(lo, s * out)
    return m


def resize(src, dsize, interpolation=INTER_AREA):
    if interpolation != INTER_AREA:
        raise NotImplementedError("stand-in implements INTER_AREA only")
    ow, oh = dsize
    img = np.asarray(src)
    if img.ndim == 3 and img.shape[2] == 1:
        img = img[:, :, 0]                       # OpenCV hands back (h, w) for a one-channel image
    if img.ndim != 2 or img.dtype != np.uint8:
        raise NotImplementedError("stand-in handles one-channel uint8 images")
    h, w = img.shape
    acc = _overlap(h, oh) @ img.astype(np.int64) @ _overlap(w, ow).T
    return ((acc + (h * w) // 2) // (h * w)).astype(np.uint8)


def cvtColor(src, code):
    raise NotImplementedError("Toybox frames arrive gray (atari_wrappers.py: WarpFrame.observation)")
