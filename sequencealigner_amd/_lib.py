"""ctypes binding of libsa_hip.so (C ABI: include/sa_hip.h).

Mirrors the seam of the reference program: sequences are parsed once
(src/main.c:25-37), a batch of pairs is dispatched (src/main.c:54-62) and the
results come back in pair order (src/main.c:65-73).
"""
import ctypes as C
import os

import numpy as np

ADJACENT = 0
TRIANGULAR = 1
DEVICE_NONE = -1  # host-only context: schedule arithmetic and work lists, no compute
ALPHABET = "ARNDCQEGHILKMFPSTWYV"
NOALLOC_MAX_PAIRS = 1 << 22   # SA_NOALLOC_MAX_PAIRS (include/sa_hip.h)

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # SA_HIP_LIBRARY names another build of the same library (A/B timing of two builds in one session)
    return os.environ.get("SA_HIP_LIBRARY") or os.path.join(_HERE, "libsa_hip.so")


class SaError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libsa_hip error {code}: {text}")
        self.code = code


class AlignmentStats(C.Structure):
    _fields_ = [
        ("score", C.c_int32),
        ("columns", C.c_int32),
        ("matches", C.c_int32),
        ("mismatches", C.c_int32),
        ("gaps", C.c_int32),
        ("similarity_bp", C.c_int32),
    ]


class Settings(C.Structure):
    """Mirror of sa_settings (include/sa_host.h): the reference's user.h keys in lower case."""
    _fields_ = [
        ("input_file", C.c_char * 4096),
        ("output_file", C.c_char * 4096),
        ("max_csv_line", C.c_uint32),
        ("max_seq_len", C.c_uint32),
        ("gap_penalty", C.c_int32),
        ("read_csv_seq_pos", C.c_uint32),
        ("read_csv_cols", C.c_uint32),
        ("write_csv_header", C.c_char * 1024),
        ("write_csv_seq1_pos", C.c_uint32),
        ("write_csv_score_pos", C.c_uint32),
        ("write_csv_align_pos", C.c_uint32),
        ("write_csv_matches_pos", C.c_uint32),
        ("write_csv_mismatches_pos", C.c_uint32),
        ("write_csv_gaps_pos", C.c_uint32),
        ("write_csv_similarity_pos", C.c_uint32),
        ("write_csv_align_fmt", C.c_char * 128),
        ("mode_write", C.c_int32),
        ("similarity_analysis", C.c_int32),
        ("batch_size", C.c_uint32),
        ("schedule", C.c_int32),
        ("num_devices", C.c_int32),
        ("devices", C.c_int32 * 8),
        ("format_threads", C.c_int32),
        ("pairs_per_submit", C.c_uint64),
        ("pairing", C.c_int32),
        ("device_format", C.c_int32),
        ("scoring_matrix_file", C.c_char * 4096),
        ("cooperative", C.c_int32),
        ("compact_records", C.c_int32),
        ("force_devices", C.c_int32),
    ]


TEXT_NONE, TEXT_SEQ1, TEXT_SEQ2, TEXT_SCORE, TEXT_ALIGN, TEXT_MATCHES, TEXT_MISMATCHES, TEXT_GAPS, TEXT_SIMILARITY, \
    TEXT_OTHER1, TEXT_OTHER2 = range(11)


class TextLayout(C.Structure):
    """Mirror of sa_text_layout (include/sa_hip.h)."""
    _fields_ = [("ncols", C.c_uint32), ("kind", C.c_int8 * 32), ("align_fmt", C.c_char * 128)]


class RunReport(C.Structure):
    _fields_ = [
        ("rows", C.c_uint64),
        ("pairs", C.c_uint64),
        ("cells", C.c_uint64),
        ("bytes_written", C.c_uint64),
        ("chunks_done", C.c_uint64),
        ("seconds", C.c_double),
        ("seconds_ingest", C.c_double),
        ("seconds_upload", C.c_double),
        ("seconds_gpu_wait", C.c_double),
        ("seconds_format", C.c_double),
        ("seconds_write", C.c_double),
        ("message", C.c_char * 256),
        ("device_submit", C.c_double * 8),
        ("device_wait", C.c_double * 8),
        ("device_format", C.c_double * 8),
        ("device_write", C.c_double * 8),
        ("record_bytes", C.c_uint64),
        ("devices_used", C.c_int32),
    ]


_lib = None

# every entry point include/sa_hip.h declares: name -> (restype, argtypes)
_u8p = C.POINTER(C.c_uint8)
SIGNATURES = {
    "sa_init": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32, C.c_uint32, C.c_int]),
    "sa_destroy": (None, [C.c_void_p]),
    "sa_strerror": (C.c_char_p, [C.c_int]),
    "sa_encode": (C.c_int, [C.c_char_p, C.c_uint32, _u8p]),
    "sa_upload": (C.c_int, [C.c_void_p, _u8p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint64]),
    "sa_prepare_schedule": (C.c_int, [C.c_void_p, C.c_int]),
    "sa_clone_sequences": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sa_schedule_rows": (C.c_uint64, [C.c_void_p, C.c_int]),
    "sa_pair_index": (C.c_uint64, [C.c_void_p, C.c_int, C.c_uint64]),
    "sa_pair_count": (C.c_uint64, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64]),
    "sa_cell_count": (C.c_uint64, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64]),
    "sa_record_stride": (C.c_uint32, [C.c_void_p]),
    "sa_record_row_bytes": (C.c_uint32, [C.c_void_p]),
    "sa_shard_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_uint64)]),
    "sa_submit": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64]),
    "sa_reserve_submission": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32]),
    "sa_next": (C.c_int, [C.c_void_p, C.POINTER(_u8p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "sa_release": (C.c_int, [C.c_void_p]),
    "sa_prepare_resident": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64]),
    "sa_run_resident": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "sa_fetch_resident": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, _u8p]),
    "sa_digest_resident": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "sa_set_tuning": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "sa_set_bucket_width": (C.c_int, [C.c_void_p, C.c_uint32]),
    "sa_last_kernel": (C.c_char_p, [C.c_void_p]),
    "sa_last_grid": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "sa_set_pairing": (C.c_int, [C.c_void_p, C.c_int]),
    "sa_set_peptide_tiles": (C.c_int, [C.c_void_p, C.c_int]),
    "sa_preload_kernels": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "sa_set_cooperative": (C.c_int, [C.c_void_p, C.c_int]),
    "sa_set_cooperative_waves": (C.c_int, [C.c_void_p, C.c_int]),
    "sa_set_long_checkpoint_rows": (C.c_int, [C.c_void_p, C.c_uint32]),
    "sa_set_compact_records": (C.c_int, [C.c_void_p, C.c_int]),
    "sa_compact_record_size": (C.c_uint32, [C.c_void_p, C.c_uint64]),
    "sa_compact_pair_offset": (C.c_uint64, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64]),
    "sa_compact_bytes": (C.c_uint64, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64]),
    "sa_set_text_layout": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sa_upload_passthrough": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_uint64]),
    "sa_reserve_text": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "sa_submit_text": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64]),
    "sa_next_text": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "sa_expand_record": (
        C.c_int,
        [_u8p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p, C.c_char_p,
         C.POINTER(AlignmentStats)],
    ),
    "sa_debug_phase_cycles": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]),
    "sa_debug_work_items": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32), C.c_uint64,
                                      C.POINTER(C.c_uint64)]),
    "sa_debug_bucket_order": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint64)]),
    "sa_align_adjacent_noalloc": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_int32), C.c_int32,
                                            C.c_void_p, C.c_uint32]),
    "sa_similarity_bp": (C.c_int, [C.c_int32, C.c_int32]),
    "sa_default_matrix": (None, [C.POINTER(C.c_int32)]),
    # include/sa_host.h
    "sa_settings_default": (None, [C.POINTER(Settings)]),
    "sa_settings_load": (C.c_int, [C.POINTER(Settings), C.c_char_p]),
    "sa_settings_validate": (C.c_int, [C.POINTER(Settings), C.c_char_p, C.c_uint32]),
    "sa_run_csv": (C.c_int, [C.POINTER(Settings), C.POINTER(RunReport)]),
    "sa_load_matrix_file": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.c_char_p, C.c_uint32]),
}


def load_library():
    """Load libsa_hip.so and bind every declared symbol (raises if one is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise SaError(-1, f"{path} is not built (run `make lib`); there is no CPU fallback")
    lib = C.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise SaError(rc, load_library().sa_strerror(rc).decode())


def similarity_bp(matches, columns):
    return load_library().sa_similarity_bp(matches, columns)


def default_settings():
    st = Settings()
    load_library().sa_settings_default(C.byref(st))
    return st


def load_settings(path):
    """Settings from a file of '#define NAME value' lines (a user.h written by the reference's GUI)."""
    st = default_settings()
    _check(load_library().sa_settings_load(C.byref(st), os.fsencode(path)))
    return st


def validate_settings(st):
    msg = C.create_string_buffer(256)
    rc = load_library().sa_settings_validate(C.byref(st), msg, 256)
    return rc, msg.value.decode()


def run_csv(input_file, output_file, schedule=ADJACENT, settings=None, **overrides):
    """CSV in -> CSV out on the GPU(s); returns the RunReport.  Mirrors the reference's main()."""
    st = settings if settings is not None else default_settings()
    st.input_file = os.fsencode(input_file)
    st.output_file = os.fsencode(output_file) if output_file else b""
    st.schedule = schedule
    if not output_file:
        st.mode_write = 0
    for k, v in overrides.items():
        setattr(st, k, v)
    rep = RunReport()
    rc = load_library().sa_run_csv(C.byref(st), C.byref(rep))
    if rc != 0:
        raise SaError(rc, rep.message.decode() or load_library().sa_strerror(rc).decode())
    return rep


class Aligner:
    """One GPU context: upload sequences once, align row ranges of a schedule."""

    def __init__(self, gap=-4, matrix=None, max_seq_len=64, device=0):
        lib = load_library()
        self._lib = lib
        self._ctx = C.c_void_p()
        mat = None
        if matrix is not None:
            m = np.ascontiguousarray(matrix, dtype=np.int32).reshape(400)
            mat = m.ctypes.data_as(C.POINTER(C.c_int32))
            self._matrix = m
        _check(lib.sa_init(C.byref(self._ctx), mat, gap, max_seq_len, device))
        self.seqs = []

    def close(self):
        if self._ctx:
            self._lib.sa_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- sequences
    def upload(self, seqs):
        seqs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        lens = np.array([len(s) for s in seqs], dtype=np.uint32)
        offs = np.zeros(len(seqs), dtype=np.uint64)
        if len(seqs) > 1:
            offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        blob = b"".join(seqs)
        codes = np.zeros(max(1, len(blob)), dtype=np.uint8)
        _check(self._lib.sa_encode(blob, len(blob), codes.ctypes.data_as(_u8p)))
        _check(
            self._lib.sa_upload(
                self._ctx,
                codes.ctypes.data_as(_u8p),
                offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                lens.ctypes.data_as(C.POINTER(C.c_uint32)),
                len(seqs),
            )
        )
        self.seqs = seqs

    def prepare_schedule(self, schedule):
        _check(self._lib.sa_prepare_schedule(self._ctx, schedule))

    def clone_from(self, donor):
        """Take the donor context's sequences and tables (device-to-device copies, no host work)."""
        _check(self._lib.sa_clone_sequences(self._ctx, donor._ctx))
        self.seqs = donor.seqs

    # -- schedule arithmetic
    def rows(self, schedule):
        return self._lib.sa_schedule_rows(self._ctx, schedule)

    def pair_index(self, schedule, row):
        return self._lib.sa_pair_index(self._ctx, schedule, row)

    def pair_count(self, schedule, r0, r1):
        return self._lib.sa_pair_count(self._ctx, schedule, r0, r1)

    def cell_count(self, schedule, r0, r1):
        return self._lib.sa_cell_count(self._ctx, schedule, r0, r1)

    @property
    def stride(self):
        return self._lib.sa_record_stride(self._ctx)

    @property
    def row_bytes(self):
        return self._lib.sa_record_row_bytes(self._ctx)

    def shard_rows(self, schedule, parts):
        b = (C.c_uint64 * (parts + 1))()
        _check(self._lib.sa_shard_rows(self._ctx, schedule, parts, b))
        return list(b)

    def set_tuning(self, tiles_per_item=0, wg_per_cu=0):
        _check(self._lib.sa_set_tuning(self._ctx, tiles_per_item, wg_per_cu))

    def set_cooperative(self, mode):
        """Wave-per-pair kernel for long sequences: 0 = by pair count (default), 1 = always, -1 = never."""
        _check(self._lib.sa_set_cooperative(self._ctx, mode))

    def set_cooperative_waves(self, waves):
        """Waves per pair in the wave-per-pair kernel: 0 = from the pair count (default), else 1, 2, 4, 8 or 16."""
        _check(self._lib.sa_set_cooperative_waves(self._ctx, waves))

    def set_long_checkpoint_rows(self, rows):
        """Packed long-sequence kernel: 64 or 32 rows between checkpoints, 0 = whole direction matrix, None = the
        library's choice (the default: 64 from a longest sequence of 512 residues, else 0)."""
        if rows is None:
            rows = 0xFFFFFFFF
        _check(self._lib.sa_set_long_checkpoint_rows(self._ctx, rows))

    def set_bucket_width(self, width):
        _check(self._lib.sa_set_bucket_width(self._ctx, width))

    def set_pairing(self, enable):
        """Two first sequences per wave in the TRIANGULAR peptide kernel (default on); off = one, 32-bit cells."""
        _check(self._lib.sa_set_pairing(self._ctx, 1 if enable else 0))

    # -- dispatch
    def set_peptide_tiles(self, enable):
        """Sets with long sequences and peptides: peptide-only tiles on the peptide kernel (default) or not."""
        _check(self._lib.sa_set_peptide_tiles(self._ctx, 1 if enable else 0))

    def submit(self, schedule, r0, r1):
        _check(self._lib.sa_submit(self._ctx, schedule, r0, r1))

    def next(self):
        """Records of the oldest submission as a (npairs, stride) uint8 view, plus its first pair index."""
        rec = _u8p()
        first = C.c_uint64()
        n = C.c_uint64()
        _check(self._lib.sa_next(self._ctx, C.byref(rec), C.byref(first), C.byref(n)))
        if n.value == 0:
            return np.zeros((0, self.stride), dtype=np.uint8), first.value
        arr = np.ctypeslib.as_array(rec, shape=(n.value * self.stride,)).reshape(n.value, self.stride)
        return arr, first.value

    def release(self):
        _check(self._lib.sa_release(self._ctx))

    # -- records without their padding
    def set_compact_records(self, enable):
        """Submissions hand out packed records: each cut to its own second sequence, 4-byte aligned, in pair order."""
        _check(self._lib.sa_set_compact_records(self._ctx, 1 if enable else 0))

    def compact_record_size(self, second_sequence):
        return self._lib.sa_compact_record_size(self._ctx, second_sequence)

    def compact_pair_offset(self, schedule, i, j):
        return self._lib.sa_compact_pair_offset(self._ctx, schedule, i, j)

    def compact_bytes(self, schedule, r0, r1):
        return self._lib.sa_compact_bytes(self._ctx, schedule, r0, r1)

    def next_compact(self, schedule, r0, r1):
        """The oldest submission's packed records as a flat uint8 view (compact mode), plus its first pair index."""
        rec = _u8p()
        first = C.c_uint64()
        n = C.c_uint64()
        _check(self._lib.sa_next(self._ctx, C.byref(rec), C.byref(first), C.byref(n)))
        nbytes = self.compact_bytes(schedule, r0, r1)
        if n.value == 0 or nbytes == 0:
            return np.zeros(0, dtype=np.uint8), first.value
        return np.ctypeslib.as_array(rec, shape=(nbytes,)), first.value

    def align_compact(self, schedule, r0=0, r1=None):
        """Align rows [r0, r1) with compact records on and return a copy of the packed stream."""
        if r1 is None:
            r1 = self.rows(schedule)
        self.set_compact_records(True)
        try:
            self.submit(schedule, r0, r1)
            rec, _ = self.next_compact(schedule, r0, r1)
            out = rec.copy()
            self.release()
        finally:
            self.set_compact_records(False)
        return out

    def align(self, schedule, r0=0, r1=None):
        """Align rows [r0, r1) and return a copy of their records."""
        if r1 is None:
            r1 = self.rows(schedule)
        self.submit(schedule, r0, r1)
        rec, _ = self.next()
        out = rec.copy()
        self.release()
        return out

    # -- rows of CSV text written on the GPU
    def set_text_layout(self, kinds, align_fmt="\"('%s', '%s')\""):
        """kinds: one TEXT_* per output column; align_fmt: the alignment column's format with two %s."""
        lay = TextLayout()
        lay.ncols = len(kinds)
        for k, v in enumerate(kinds[:32]):     # a longer list is refused by the library through ncols
            lay.kind[k] = v
        lay.align_fmt = align_fmt.encode()
        _check(self._lib.sa_set_text_layout(self._ctx, C.byref(lay)))

    def upload_passthrough(self, texts):
        """The pass-through text of every uploaded row (what OTHER1/OTHER2 columns print)."""
        blob = b"".join(t.encode() if isinstance(t, str) else t for t in texts)
        offs = (C.c_uint64 * (len(texts) + 1))()
        at = 0
        for k, t in enumerate(texts):
            offs[k] = at
            at += len(t)
        offs[len(texts)] = at
        _check(self._lib.sa_upload_passthrough(self._ctx, blob, offs, len(texts)))

    def submit_text(self, schedule, r0, r1):
        _check(self._lib.sa_submit_text(self._ctx, schedule, r0, r1))

    def next_text(self):
        """(address, byte count, first pair, pairs) of the oldest text submission; the text lives in the
        context and stays valid until the next-but-one next_text, also after release()."""
        ptr, nbytes, first, n = C.c_void_p(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(self._lib.sa_next_text(self._ctx, C.byref(ptr), C.byref(nbytes), C.byref(first), C.byref(n)))
        return ptr.value, nbytes.value, first.value, n.value

    def align_text(self, schedule, r0=0, r1=None):
        """Rows [r0, r1) of the schedule as finished CSV text (bytes)."""
        if r1 is None:
            r1 = self.rows(schedule)
        self.submit_text(schedule, r0, r1)
        ptr, nbytes, _, _ = self.next_text()
        out = C.string_at(ptr, nbytes) if nbytes else b""
        self.release()
        return out

    # -- measurement path (records stay on the device)
    def prepare_resident(self, schedule, r0, r1):
        _check(self._lib.sa_prepare_resident(self._ctx, schedule, r0, r1))

    def run_resident(self):
        ms = C.c_float()
        _check(self._lib.sa_run_resident(self._ctx, C.byref(ms)))
        return ms.value

    def fetch_resident(self, first, n):
        out = np.zeros((n, self.stride), dtype=np.uint8)
        _check(self._lib.sa_fetch_resident(self._ctx, first, n, out.ctypes.data_as(_u8p)))
        return out

    def digest_resident(self, first, n):
        """128-bit digest (two uint64) of records [first, first+n) of the resident range, computed on the device."""
        d = (C.c_uint64 * 2)()
        _check(self._lib.sa_digest_resident(self._ctx, first, n, d))
        return (d[0], d[1])

    def last_kernel(self):
        """Name of the fill + traceback kernel the most recent launch ran."""
        return self._lib.sa_last_kernel(self._ctx).decode()

    def last_grid(self):
        """(workgroups of the most recent launch, workgroups it would have had with memory to spare)."""
        a, b = C.c_uint32(), C.c_uint32()
        _check(self._lib.sa_last_grid(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def phase_cycles(self, enable=True):
        """Shader-cycle totals per kernel phase of the previous timed launch; (re)arms the timers."""
        tot = (C.c_uint64 * 8)()
        _check(self._lib.sa_debug_phase_cycles(self._ctx, 1 if enable else 0, tot))
        return list(tot)

    def work_items(self, schedule, r0, r1):
        """(n, 8) uint32 array {p, v0, count, b0, c0, 0, 0, 0}: the work list for rows [r0, r1)."""
        n = C.c_uint64()
        _check(self._lib.sa_debug_work_items(self._ctx, schedule, r0, r1, None, 0, C.byref(n)))
        arr = np.zeros((max(1, n.value), 8), dtype=np.uint32)
        _check(self._lib.sa_debug_work_items(self._ctx, schedule, r0, r1, arr.ctypes.data_as(C.POINTER(C.c_uint32)),
                                             n.value, C.byref(n)))
        return arr[: n.value]

    def bucket_order(self):
        """File index of every sequence in bucket order (length bucket, then file index)."""
        n = C.c_uint64()
        _check(self._lib.sa_debug_bucket_order(self._ctx, None, 0, C.byref(n)))
        arr = np.zeros(max(1, n.value), dtype=np.uint32)
        _check(self._lib.sa_debug_bucket_order(self._ctx, arr.ctypes.data_as(C.POINTER(C.c_uint32)), n.value,
                                               C.byref(n)))
        return arr[: n.value]

    # -- records
    def expand(self, record, i, j):
        """Aligned strings and statistics of the pair (file rows i, j) from its record."""
        s1, s2 = self.seqs[i], self.seqs[j]
        n = len(s1) + len(s2) + 1
        a1 = C.create_string_buffer(n)
        a2 = C.create_string_buffer(n)
        st = AlignmentStats()
        rec = np.ascontiguousarray(record, dtype=np.uint8)
        _check(
            self._lib.sa_expand_record(
                rec.ctypes.data_as(_u8p), rec.size, self.row_bytes, s1, len(s1), s2, len(s2), a1, a2, C.byref(st)
            )
        )
        return a1.value, a2.value, st
