"""`pyfft` import name for the MI355X build ("package importable as pyfft, pyfft.VERSION kept").

    import pyfft; pyfft.VERSION                 # the reference's version tuple (pyfft/__init__.py:1 in the reference)
    from pyfft.hip import Plan                  # was: from pyfft.cuda import Plan (doc/source/index.rst:37-60)

The backend module is `pyfft.hip` (= pyfft_amd.hip).  There is deliberately no `pyfft.cuda` / `pyfft.cl` alias: this build has one
backend and no CUDA / OpenCL compatibility layer.
"""
import sys

import pyfft_amd
from pyfft_amd import hip

# the API level this build is a drop-in for (the reference's own VERSION); the build's own version is pyfft_amd.VERSION
VERSION = (0, 3, 9)
BUILD_VERSION = pyfft_amd.VERSION

sys.modules[__name__ + ".hip"] = hip      # `from pyfft.hip import Plan` / `import pyfft.hip`
