"""Import alias: the product package lives in the directory the build contract names,
``unsupervised-keypoint-learning-for-guiding-class-conditional-video-prediction_amd/``, which is not a valid Python
identifier.  ``import kpx_amd`` executes that directory's ``__init__.py`` with this module's ``__path__`` pointing there,
so ``kpx_amd.ops`` etc. resolve to the files in the long-named directory (no code lives here)."""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         'unsupervised-keypoint-learning-for-guiding-class-conditional-video-prediction_amd')
__path__ = [_PKG_DIR]
with open(_os.path.join(_PKG_DIR, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, '__init__.py'), 'exec'))
