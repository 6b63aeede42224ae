"""MI355X-native drop-in for the GPU half of PyTV-4D (eboigne/PyTV-4D v1.1.2).

    import pytv
    tv, G = pytv.tv_GPU.tv_hybrid(img)                  # pytv/tv_GPU.py:47
    Dx    = pytv.tv_operators_GPU.D_hybrid(img)         # pytv/tv_operators_GPU.py:134

Everything computes in hand-written HIP kernels (``libpytv4d_hip.so``, C-ABI in
``include/pytv4d.h``).  Importing this package without the built library raises ImportError: there
is deliberately no CPU or PyTorch fallback.  The reference's CPU twins (``tv_CPU``,
``tv_operators_CPU``) are NOT part of this package -- a NumPy restatement of them lives under
``oracle/`` as test infrastructure only.
"""
__version__ = "1.1.2+mi355x.1"

from . import _native

_native.lib()          # fail loudly, now, if the HIP library is missing

from . import tv_operators_GPU   # noqa: E402
from . import tv_GPU             # noqa: E402
from . import solvers            # noqa: E402
from . import slab               # noqa: E402
from . import restoration        # noqa: E402
from . import utils              # noqa: E402
from .restoration import denoise_tv_chambolle   # noqa: E402

__all__ = ["tv_GPU", "tv_operators_GPU", "solvers", "slab", "restoration", "utils", "denoise_tv_chambolle"]


def __getattr__(name):
    if name in ("tv_CPU", "tv_operators_CPU", "tests"):
        raise AttributeError(
            "pytv.%s is not part of the MI355X build: it replaces the GPU half of PyTV-4D (pytv.tv_GPU, pytv.tv_operators_GPU). "
            "Use the reference package for its NumPy twins and self-test harness; a NumPy restatement lives under oracle/ of "
            "this repository as test infrastructure." % name)
    raise AttributeError("module 'pytv' has no attribute %r" % name)
