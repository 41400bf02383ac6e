"""genie-smem_amd: MI355X-native batched SMEM finder -- a drop-in for the hot path of
jgkellymit/GENIE-SMEM (seed lookup, suffix-array interval search, SMEM extension loop).

Import name: `genie_smem_amd` (the repo-root alias module maps it onto this directory).
Compute goes through the C-ABI library libgenie_smem.so (include/genie_smem.h); there is no
CPU fallback.
"""
from . import _native
from .exact_match import ExactMatch
from .index import GenieIndex
from .lut import LUT
from .rmi import RMI
from .rmi_lut import RMI_LUT
from .smem import SMEM, create_query_from_ref, create_random_query
from . import packing, parallel

__all__ = ["ExactMatch", "LUT", "RMI", "RMI_LUT", "SMEM", "GenieIndex", "parallel", "packing", "_native",
           "create_query_from_ref", "create_random_query"]
