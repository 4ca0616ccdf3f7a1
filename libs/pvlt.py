"""`libs.pvlt` -- same import path as the reference (libs/pvlt.py), MI355X-native implementation (mvlt_amd.pvlt)."""
from mvlt_amd.pvlt import (PyramidVisionLanguageTransformer, pvlt_large, pvlt_medium, pvlt_small,  # noqa: F401
                           pvlt_tiny)

__all__ = ['pvlt_tiny', 'pvlt_small', 'pvlt_medium', 'pvlt_large']
