"""MI355X-native online TSDF fusion + semantics engine (per-frame hot path only).

Drop-in counterparts of the reference's ``modules/pipeline.py`` Pipeline and
``modules/database.py`` Database, backed by hand-written gfx950 HIP kernels behind the
C ABI declared in ``include/ojf.h``.
"""
__version__ = '0.1.0'
