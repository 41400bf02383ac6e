"""ctypes binding of the C-ABI library (include/genie_smem.h -> libgenie_smem.so).

There is no CPU fallback: if the library is missing or a call fails this module raises.
torch is imported first on purpose -- PyTorch-ROCm ships its own libamdhip64.so.7; loading it
first makes our library resolve the same HIP runtime, so torch tensors' device pointers and
streams are valid inside our launches.
"""
import ctypes as C
import os
import subprocess

import torch  # noqa: F401  (must precede the CDLL below, see docstring)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libgenie_smem.so")
CSRC = os.path.join(_PKG, "csrc")

ABI_VERSION = 2
HEADER_BYTES = 512
MAX_K = 16
MAX_READ_LEN = 8192
MODE_BWA, MODE_LUT, MODE_RMI = 0, 1, 2
MODES = {"bwa": MODE_BWA, "lut": MODE_LUT, "rmi": MODE_RMI}
IMAGE_NO_SEED_TABLE = 1
TABLE_WIDE, TABLE_COMPACT = 1 << 8, 2 << 8        # genie_index_create_ex: ORed into table_bits
OPT_SEARCH_ALL = 2
OPT_GROUP_POSITIONS = 4
OPT_SEARCH_ONLY = 5
OPT_SEARCH_BLOCKS_PER_CU = 6
OPT_SEARCH_STAGES_OFF = 7
OPT_SCHEDULING = 8
READ_OK, READ_BAD_BASE, READ_TOO_SHORT, READ_ABSENT_BASE, READ_OVERFLOW = 0, 1, 2, 3, 4

# every symbol include/genie_smem.h declares (tests check the library exports all of them)
SYMBOLS = [
    "genie_abi_version", "genie_index_create", "genie_index_create_from_sa", "genie_index_create_ex", "genie_index_set_rmi",
    "genie_index_info", "genie_index_suffix_array", "genie_index_lut_arrays", "genie_index_blob_bytes",
    "genie_index_serialize", "genie_index_image_bytes", "genie_index_serialize_image", "genie_index_open", "genie_index_validate", "genie_index_to_device", "genie_index_destroy",
    "genie_sa_interval", "genie_seed_lookup", "genie_find_smems", "genie_find_smems_csr", "genie_find_smems_packed", "genie_find_smems_packed6", "genie_find_smems_workspace_bytes", "genie_find_smems_workspace_rows",
    "genie_compact_tmp_bytes",
    "genie_compact_smems", "genie_locate_tmp_bytes", "genie_locate", "genie_index_train_rmi", "genie_index_rmi_models", "genie_launch_info", "genie_search_kernel_name", "genie_index_set_option", "genie_index_set_stage_events", "genie_strerror", "genie_last_hip_error",
]


class GenieInfo(C.Structure):
    _fields_ = [("n", C.c_int64), ("K", C.c_int32), ("dir_bits", C.c_int32), ("lut_keys", C.c_int64),
                ("lut_slots", C.c_int64), ("rmi_levels", C.c_int32), ("has_host", C.c_int32),
                ("has_device", C.c_int32), ("device", C.c_int32), ("blob_bytes", C.c_int64)]


class GenieError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = lib().genie_strerror(status).decode()
        if status == -5:
            msg += " [" + lib().genie_last_hip_error().decode() + "]"
        super().__init__(f"{where}: {msg} (status {status})")


def build(force=False):
    """Compile libgenie_smem.so for gfx950 with hipcc (csrc/Makefile), in-tree."""
    srcs = [os.path.join(CSRC, f) for f in ("kernels.hip", "short_read_kernel.inc", "match_table_kernel.inc", "capi.cpp", "index_host.cpp", "genie_internal.h", "Makefile")]
    srcs.append(os.path.join(os.path.dirname(_PKG), "include", "genie_smem.h"))
    newest = max(os.path.getmtime(s) for s in srcs)
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        subprocess.check_call(["make", "-s", "-C", CSRC] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP library has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C genie-smem_amd/csrc`). "
            "There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    u8p, i32p, dp = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_double)
    vpp = C.POINTER(C.c_void_p)
    sig = {
        "genie_abi_version": (C.c_int, []),
        "genie_index_create": (C.c_int, [u8p, i64, i32, i32, vpp]),
        "genie_index_create_from_sa": (C.c_int, [u8p, i64, i32p, i32, i32, vpp]),
        "genie_index_create_ex": (C.c_int, [u8p, i64, i32p, i32, i32, i32, vpp]),
        "genie_index_set_rmi": (C.c_int, [vp, i32, i32p, i32p, dp, dp]),
        "genie_index_info": (C.c_int, [vp, C.POINTER(GenieInfo)]),
        "genie_index_suffix_array": (i32p, [vp]),
        "genie_index_lut_arrays": (C.c_int, [vp, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(i32p), C.POINTER(i32p)]),
        "genie_index_blob_bytes": (i64, [vp]),
        "genie_index_serialize": (C.c_int, [vp, vp, i64]),
        "genie_index_image_bytes": (i64, [vp, i32]),
        "genie_index_serialize_image": (C.c_int, [vp, i32, vp, i64]),
        "genie_index_open": (C.c_int, [vp, vp, i64, i32, vpp]),
        "genie_index_validate": (C.c_int, [vp, C.POINTER(C.c_uint32), vp]),
        "genie_index_to_device": (C.c_int, [vp, i32]),
        "genie_index_destroy": (None, [vp]),
        "genie_sa_interval": (C.c_int, [vp, vp, vp, i64, i32, i32, vp, vp]),
        "genie_seed_lookup": (C.c_int, [vp, i32, vp, i64, vp, vp, vp]),
        "genie_find_smems": (C.c_int, [vp, i32, vp, vp, i64, i32, i32, i32, vp, vp, i32, vp, vp, i64, vp]),
        "genie_find_smems_csr": (C.c_int, [vp, i32, vp, vp, i64, i32, i32, i32, vp, vp, i64, vp, vp, i64, vp]),
        "genie_find_smems_packed": (C.c_int, [vp, i32, vp, vp, i64, i32, i32, i32, vp, vp, vp, i64, vp, vp, i64, vp, i64, vp]),
        "genie_find_smems_packed6": (C.c_int, [vp, i32, vp, vp, i64, i32, i32, i32, vp, vp, vp, i64, vp, vp, i64, vp, i64, vp]),
        "genie_find_smems_workspace_bytes": (i64, [i64, i32]),
        "genie_find_smems_workspace_rows": (C.c_int, [i32, i32p]),
        "genie_compact_tmp_bytes": (i64, [i64]),
        "genie_compact_smems": (C.c_int, [vp, vp, i64, i32, vp, vp, i64, vp, vp]),
        "genie_index_train_rmi": (C.c_int, [vp, i32, vp, vp, vp]),
        "genie_index_rmi_models": (C.c_int, [vp, vp, vp, vp]),
        "genie_locate_tmp_bytes": (i64, [i64]),
        "genie_locate": (C.c_int, [vp, vp, i32, i64, vp, vp, i64, vp, i64, vp]),
        "genie_launch_info": (C.c_int, [vp, i32, i32, i32p, i32p, i32p]),
        "genie_search_kernel_name": (C.c_int, [vp, i32, i32, C.c_char_p, i32]),
        "genie_index_set_option": (C.c_int, [vp, i32, i32]),
        "genie_index_set_stage_events": (C.c_int, [vp, vp, vp]),
        "genie_strerror": (C.c_char_p, [C.c_int]),
        "genie_last_hip_error": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.genie_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libgenie_smem.so: ABI version {L.genie_abi_version()}, this binding is for {ABI_VERSION} "
                           "(rebuild: make -C genie-smem_amd/csrc)")
    _lib = L
    return L


def check(status, where):
    if status != 0:
        raise GenieError(status, where)
