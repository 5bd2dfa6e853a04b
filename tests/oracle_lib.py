"""ctypes access to the CPU checker in oracle/ (tests and bench baseline only)."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libnw_oracle.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


class OracleResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("score", "columns", "matches", "mismatches", "gaps", "similarity_bp")]


class RunStats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("pairs", C.c_uint64), ("cells", C.c_uint64), ("rows", C.c_uint64)]


_lib = None


def oracle():
    global _lib
    if _lib is None:
        lib = C.CDLL(ORACLE_SO)
        lib.sa_oracle_align.restype = C.c_int
        lib.sa_oracle_align.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int32), C.c_int,
                                        C.c_char_p, C.c_char_p, C.POINTER(OracleResult)]
        lib.sa_oracle_similarity_bp.restype = C.c_int
        lib.sa_oracle_similarity_bp.argtypes = [C.c_int, C.c_int]
        lib.sa_oracle_run_file.restype = C.c_int
        lib.sa_oracle_run_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                           C.c_uint64, C.POINTER(RunStats)]
        lib.sa_oracle_default_matrix.restype = None
        lib.sa_oracle_default_matrix.argtypes = [C.POINTER(C.c_int32)]
        _lib = lib
    return _lib


def align(s1, s2, gap=-4, matrix=None):
    """(aligned1, aligned2, OracleResult) for one pair."""
    if isinstance(s1, str):
        s1 = s1.encode()
    if isinstance(s2, str):
        s2 = s2.encode()
    a1 = C.create_string_buffer(len(s1) + len(s2) + 1)
    a2 = C.create_string_buffer(len(s1) + len(s2) + 1)
    r = OracleResult()
    mat = None
    if matrix is not None:
        mat = (C.c_int32 * 400)(*[int(x) for x in matrix])
    rc = oracle().sa_oracle_align(s1, len(s1), s2, len(s2), mat, gap, a1, a2, C.byref(r))
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}")
    return a1.value, a2.value, r


def run_file(in_path, out_path, schedule, threads=1, gap=-4, max_pairs=0, matrix=None):
    st = RunStats()
    mat = None
    if matrix is not None:
        mat = (C.c_int32 * 400)(*[int(x) for x in matrix])
    rc = oracle().sa_oracle_run_file(in_path.encode(), out_path.encode() if out_path else None, schedule, threads,
                                     gap, mat, max_pairs, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"oracle run error {rc}")
    return st
