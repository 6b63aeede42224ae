"""``pytv.utils`` of the reference (pytv/utils.py:46-56) provides ``cameraman()``, the 256 x 256 grayscale test image its
README loops start from (README.md:113,141).  The image file is the reference's data and is not redistributed here:
``cameraman()`` loads it from ``PYTV_CAMERAMAN`` (path of a ``.npy``), from ``pytv/media/cameraman.npy`` if the user has put
it there, or takes scikit-image's ``camera()`` (512 x 512, averaged 2 x 2 down to 256 x 256) when that package is
installed; otherwise it says exactly that."""
import os
import warnings

import numpy as np

__all__ = ["cameraman"]


def cameraman():
    """The 256 x 256 grayscale cameraman image as a NumPy array (reference: pytv/utils.py:46-56)."""
    here = os.path.dirname(os.path.abspath(__file__))
    for path in (os.environ.get("PYTV_CAMERAMAN"), os.path.join(here, "media", "cameraman.npy")):
        if path and os.path.exists(path):
            return np.load(path)
    try:
        from skimage import data
    except ImportError:
        raise FileNotFoundError(
            "pytv.utils.cameraman(): the reference's media/cameraman.npy is not shipped with this package. Set PYTV_CAMERAMAN "
            "to a 256 x 256 .npy image, copy the reference's file to %s, or install scikit-image." % os.path.join(here, "media", "cameraman.npy"))
    warnings.warn("pytv.utils.cameraman(): the reference's media/cameraman.npy is not available; returning scikit-image's "
                  "camera() averaged 2 x 2 down to 256 x 256 -- a DIFFERENT photograph: TV and loss values will not match the "
                  "numbers of the reference's README.  Set PYTV_CAMERAMAN to the reference's file to compare.", stacklevel=2)
    img = np.asarray(data.camera(), dtype=np.float64)
    return img.reshape(256, 2, 256, 2).mean(axis=(1, 3)).astype(np.int64)
