"""NHWC view <-> NCHW helpers for the plugin boundary: tools_origin.py hands the kernels
``img.permute(0, 2, 3, 1)`` views (:32,58,210) and permutes the result back, so going through
these helpers is copy-free on the hot path."""


def to_nchw(img):
    if img.dim() != 4:
        raise ValueError('expected a 4-D NHWC image batch, got %s' % (tuple(img.shape),))
    return img.permute(0, 3, 1, 2)


def to_nhwc(img):
    return img.permute(0, 2, 3, 1)
