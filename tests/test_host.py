"""Host side of the library without a GPU: the shared library loads and exports
the whole C ABI, records expand correctly, schedules and work lists cover every
pair exactly once, settings load and validate.  No compute call is made."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest

import oracle_lib
import sequencealigner_amd as sa
from sequencealigner_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALPHABET = "ARNDCQEGHILKMFPSTWYV"


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sa_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = sa.load_library()
    names = _declared("sa_hip.h") + _declared("sa_host.h")
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no Python binding"


def test_compute_fails_loudly_without_device():
    al = sa.Aligner(device=sa.DEVICE_NONE)
    al.upload(["ARND", "ARNDC", "WWW"])
    with pytest.raises(sa.SaError):
        al.submit(sa.TRIANGULAR, 0, 2)
    with pytest.raises(sa.SaError):
        al.prepare_resident(sa.TRIANGULAR, 0, 2)


def test_text_layout_validation_needs_no_device():
    al = sa.Aligner(device=sa.DEVICE_NONE)
    al.upload(["ARND", "ARNDC", "WWW"])
    al.set_text_layout([_lib.TEXT_SEQ1, _lib.TEXT_SEQ2, _lib.TEXT_SCORE, _lib.TEXT_ALIGN])
    with pytest.raises(sa.SaError):
        al.set_text_layout([])                                      # no columns
    with pytest.raises(sa.SaError):
        al.set_text_layout([_lib.TEXT_SCORE] * 33)                  # more than 32 columns
    with pytest.raises(sa.SaError):
        al.set_text_layout([_lib.TEXT_OTHER2 + 1])                  # unknown column kind
    with pytest.raises(sa.SaError):
        al.set_text_layout([_lib.TEXT_ALIGN], align_fmt="(%s)")     # the format needs two %s
    al.set_text_layout([_lib.TEXT_ALIGN], align_fmt="%s|%s")        # empty head and tail are fine
    with pytest.raises(sa.SaError):
        al.upload_passthrough(["a", "b", "c"])                      # device-side state: no device, no upload
    with pytest.raises(sa.SaError):
        al.submit_text(sa.TRIANGULAR, 0, 2)


def test_encode_rejects_non_alphabet():
    lib = sa.load_library()
    buf = (C.c_uint8 * 8)()
    assert lib.sa_encode(b"ARND", 4, buf) == 0 and list(buf)[:4] == [0, 1, 2, 3]
    for bad in (b"ARNB", b"arnd", b"AR-D", b"ARNX"):
        assert lib.sa_encode(bad, 4, buf) == -2


def test_upload_rejects_empty_and_too_long():
    al = sa.Aligner(device=sa.DEVICE_NONE, max_seq_len=63)
    with pytest.raises(sa.SaError):
        al.upload(["ARND", ""])
    with pytest.raises(sa.SaError):
        al.upload(["A" * 64, "ARND"])
    al.upload(["A" * 63, "ARND"])
    assert al.row_bytes == 1
    with pytest.raises(sa.SaError):
        sa.Aligner(device=sa.DEVICE_NONE, max_seq_len=4097)


@pytest.mark.parametrize(
    "m,p,bp",
    [(57, 100, 5699), (69, 100, 6899), (43, 125, 3439), (51, 125, 4079), (57, 200, 2849), (1, 6, 1666), (6, 6, 10000)],
)
def test_similarity_known_answers(m, p, bp):
    assert sa.similarity_bp(m, p) == bp


def test_similarity_equals_oracle_everywhere():
    lib = sa.load_library()
    orc = oracle_lib.oracle()
    for p in range(1, 130):
        for m in range(0, p + 1):
            assert lib.sa_similarity_bp(m, p) == orc.sa_oracle_similarity_bp(m, p)


def _record_from_alignment(a1, a2, score, stride, row_bytes=1):
    """Encode an alignment the way the kernels do: one byte (or uint16) per row of sequence 2."""
    rec = np.zeros(stride, dtype=np.uint8)
    matches = sum(1 for x, y in zip(a1, a2) if x == y)
    gaps = sum(1 for x, y in zip(a1, a2) if x != y and x == "-")
    rec[0:4] = np.frombuffer(np.int32(score).tobytes(), dtype=np.uint8)
    if row_bytes == 1:
        rec[4:6] = np.frombuffer(np.uint16(matches).tobytes(), dtype=np.uint8)
        rec[6:8] = np.frombuffer(np.uint16(gaps).tobytes(), dtype=np.uint8)
        header = 8
    else:
        rec[4:8] = np.frombuffer(np.uint32(matches).tobytes(), dtype=np.uint8)
        rec[8:12] = np.frombuffer(np.uint32(gaps).tobytes(), dtype=np.uint8)
        header = 16
    # walk the columns from the end: LEFT moves (gap in sequence 2) accumulate until a column consumes a row
    row = sum(1 for y in a2 if y != "-")
    run = 0
    for x, y in zip(reversed(a1), reversed(a2)):
        if y == "-":
            run += 1
        else:
            v = run * 2 + (1 if x == "-" else 0)
            if row_bytes == 1:
                rec[header + row - 1] = v
            else:
                rec[header + 2 * (row - 1) : header + 2 * row] = np.frombuffer(np.uint16(v).tobytes(), dtype=np.uint8)
            run = 0
            row -= 1
    return rec


def test_expand_record_round_trip_against_oracle():
    rng = random.Random(3)
    al = sa.Aligner(device=sa.DEVICE_NONE)
    seqs = ["".join(rng.choice(ALPHABET) for _ in range(rng.randint(1, 64))) for _ in range(300)]
    al.upload(seqs)
    stride = al.stride
    for k in range(len(seqs) - 1):
        o1, o2, r = oracle_lib.align(seqs[k], seqs[k + 1])
        rec = _record_from_alignment(o1.decode(), o2.decode(), r.score, stride)
        a1, a2, st = al.expand(rec, k, k + 1)
        assert (a1, a2) == (o1, o2)
        assert (st.score, st.columns, st.matches, st.mismatches, st.gaps, st.similarity_bp) == (
            r.score, r.columns, r.matches, r.mismatches, r.gaps, r.similarity_bp)


def test_expand_long_record_round_trip_against_oracle():
    rng = random.Random(4)
    al = sa.Aligner(device=sa.DEVICE_NONE, max_seq_len=700)
    seqs = ["".join(rng.choice(ALPHABET) for _ in range(rng.choice([1, 64, 65, 130, 333, 700]))) for _ in range(40)]
    seqs.append("A" * 500)       # long runs of gaps against the next one
    seqs.append("W" * 3)
    al.upload(seqs)
    assert al.row_bytes == 2 and al.stride >= 16 + 2 * 700
    for k in range(len(seqs) - 1):
        o1, o2, r = oracle_lib.align(seqs[k], seqs[k + 1])
        rec = _record_from_alignment(o1.decode(), o2.decode(), r.score, al.stride, row_bytes=2)
        a1, a2, st = al.expand(rec, k, k + 1)
        assert (a1, a2) == (o1, o2)
        assert (st.score, st.columns, st.matches, st.mismatches, st.gaps, st.similarity_bp) == (
            r.score, r.columns, r.matches, r.mismatches, r.gaps, r.similarity_bp)


def test_expand_record_rejects_garbage():
    al = sa.Aligner(device=sa.DEVICE_NONE)
    al.upload(["ARNDC", "ARN"])
    rec = np.zeros(al.stride, dtype=np.uint8)
    rec[8:11] = [200, 0, 0]          # 100 LEFT moves in a 5-residue sequence
    with pytest.raises(sa.SaError):
        al.expand(rec, 0, 1)


def _pairs_of_items(items, order, schedule):
    out = []
    for p, v0, count, _b0, _c0, p2, _y, _z in items.tolist():
        if schedule == sa.ADJACENT:
            out.extend((p + l, p + l + 1) for l in range(count))
        else:
            entries = [fi for fi in order if fi > p][v0 : v0 + 64 * count]
            out.extend((p, fi) for fi in entries)
            if p2 != 0xFFFFFFFF:       # a second first sequence rides along on the same entries
                assert p2 > p
                out.extend((p2, fi) for fi in entries if fi > p2)
    return out


@pytest.mark.parametrize("pairing", [True, False])
@pytest.mark.parametrize("n,lo,hi", [(2, 5, 5), (65, 1, 64), (300, 6, 50), (1042, 6, 46), (3000, 10, 12)])
def test_work_items_cover_every_pair_once(n, lo, hi, pairing):
    seqs, _ = synth.generate(n, lo, hi, seed=n)
    al = sa.Aligner(device=sa.DEVICE_NONE)
    al.upload(seqs)
    al.set_pairing(pairing)
    order = al.bucket_order().tolist()
    assert sorted(order) == list(range(n))
    for schedule in (sa.ADJACENT, sa.TRIANGULAR):
        rows = al.rows(schedule)
        cuts = sorted({0, rows // 3, rows // 2, rows})
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            got = sorted(_pairs_of_items(al.work_items(schedule, r0, r1), order, schedule))
            if schedule == sa.ADJACENT:
                want = [(k, k + 1) for k in range(r0, r1)]
            else:
                want = [(i, j) for i in range(r0, r1) for j in range(i + 1, n)]
            assert got == want
            if schedule == sa.TRIANGULAR and n > 2:
                has_second = any(it[5] != 0xFFFFFFFF for it in al.work_items(schedule, r0, r1).tolist())
                assert has_second == pairing
            assert al.pair_count(schedule, r0, r1) == len(want)
            cells = sum(len(seqs[i]) * len(seqs[j]) for i, j in want)
            assert al.cell_count(schedule, r0, r1) == cells


def _mixed_set(n, n_long, lo, hi, seed):
    """n peptides of 6-50 residues with n_long sequences of lo-hi residues spread among them"""
    seqs, _ = synth.generate(n, 6, 50, seed=seed)
    more, _ = synth.generate(n_long, lo, hi, seed=seed + 1)
    step = n // n_long
    for k, s in enumerate(more):
        seqs.insert(k * step + k + 3, s)
    return seqs


@pytest.mark.parametrize("n,n_long,lo,hi", [(400, 6, 70, 300), (300, 40, 60, 90), (200, 3, 500, 900)])
def test_work_items_of_a_set_of_peptides_and_long_sequences(n, n_long, lo, hi):
    """A set with sequences above 64 residues is a long-sequence context; the tiles whose sequences are all peptides form
    the end of the work list (several tiles per item, for the peptide kernel).  Every pair still appears exactly once,
    with the split on and off, and the split takes most of the peptide-against-peptide pairs."""
    seqs = _mixed_set(n, n_long, lo, hi, seed=n)
    total = len(seqs)
    al = sa.Aligner(device=sa.DEVICE_NONE, max_seq_len=hi)
    al.upload(seqs)
    assert al.row_bytes == 2
    order = al.bucket_order().tolist()
    rows = al.rows(sa.TRIANGULAR)
    short = [len(x) <= 64 for x in seqs]
    for tiles_on in (True, False):
        al.set_peptide_tiles(tiles_on)
        for r0, r1 in ((0, rows), (rows // 3, rows // 2)):
            items = al.work_items(sa.TRIANGULAR, r0, r1)
            got = sorted(_pairs_of_items(items, order, sa.TRIANGULAR))
            assert got == [(i, j) for i in range(r0, r1) for j in range(i + 1, total)]
            # the items at the end whose first sequences and entries are all peptides
            tail_pairs = 0
            for it in items.tolist()[::-1]:
                ps = _pairs_of_items(np.array([it], dtype=items.dtype), order, sa.TRIANGULAR)
                if not all(short[i] and short[j] for i, j in ps):
                    break
                tail_pairs += len(ps)
            peptide_pairs = sum(1 for i, j in got if short[i] and short[j])
            if tiles_on:
                assert tail_pairs >= 0.8 * peptide_pairs - 64 * (r1 - r0)
            else:
                assert all(it[2] == 1 for it in items.tolist())      # one tile per item, as the long kernels take them
    al.close()


def test_pair_index_is_row_major_triangular():
    seqs, _ = synth.generate(50, 6, 20, seed=5)
    al = sa.Aligner(device=sa.DEVICE_NONE)
    al.upload(seqs)
    idx = 0
    for i in range(49):
        assert al.pair_index(sa.TRIANGULAR, i) == idx
        idx += 49 - i
    assert al.pair_index(sa.ADJACENT, 17) == 17


@pytest.mark.parametrize("parts", [1, 2, 4, 8])
def test_shard_rows_partition_with_balanced_cells(parts):
    seqs, _ = synth.generate(4000, 6, 50, seed=1)
    al = sa.Aligner(device=sa.DEVICE_NONE)
    al.upload(seqs)
    for schedule in (sa.ADJACENT, sa.TRIANGULAR):
        b = al.shard_rows(schedule, parts)
        assert b[0] == 0 and b[-1] == al.rows(schedule) and all(x <= y for x, y in zip(b, b[1:]))
        cells = [al.cell_count(schedule, lo, hi) for lo, hi in zip(b, b[1:])]
        assert sum(cells) == al.cell_count(schedule, 0, al.rows(schedule))
        assert max(cells) <= 1.02 * sum(cells) / parts + 64 * 64 * 4000


def test_synthetic_generator_is_deterministic():
    a, la = synth.generate(100, 6, 50, seed=1)
    b, lb = synth.generate(100, 6, 50, seed=1)
    assert a == b and la == lb
    assert all(6 <= len(s) <= 50 and set(s) <= set(ALPHABET.encode()) for s in a)
    assert a[0] != synth.generate(100, 6, 50, seed=2)[0][0]


def test_settings_defaults_match_reference_user_h():
    st = sa.default_settings()
    assert (st.max_csv_line, st.max_seq_len, st.gap_penalty, st.batch_size) == (256, 64, -4, 32768)
    assert (st.read_csv_seq_pos, st.read_csv_cols, st.mode_write, st.similarity_analysis) == (0, 2, 1, 1)
    assert st.write_csv_header == b"sequence1,sequence2,label1,label2,score,alignment,matches,mismatches,gaps,similarity\n"
    assert st.write_csv_align_fmt == b"\"('%s', '%s')\""
    assert sa.validate_settings(st) == (0, "")


def test_settings_load_from_user_h_style_file(tmp_path):
    f = tmp_path / "user.h"
    f.write_text(
        '#ifndef USER_H\n#define USER_H\n// comment\n#define MAX_CSV_LINE 512\n#define MAX_SEQ_LEN 48\n'
        '#define GAP_PENALTY -7\n#define READ_CSV_HEADER "sequence,label\\n"\n#define READ_CSV_SEQ_POS 0\n'
        '#define READ_CSV_COLS 2\n#define WRITE_CSV_HEADER "a,b,c,d,e,f\\n"\n#define WRITE_CSV_SEQ1_POS 2\n'
        '#define WRITE_CSV_SCORE_POS 0\n#define WRITE_CSV_ALIGN_POS 1\n#define WRITE_CSV_ALIGN_FMT "%s|%s"\n'
        '#define INPUT_FILE "/data/in.csv"\n#define OUTPUT_FILE "/data/out.csv"\n#define MODE_MULTITHREAD 1\n'
        '#define SIMILARITY_ANALYSIS 0\n#define MODE_WRITE 1\n#define BATCH_SIZE 4096\n#endif\n')
    st = sa.load_settings(str(f))
    assert (st.max_csv_line, st.max_seq_len, st.gap_penalty, st.batch_size) == (512, 48, -7, 4096)
    assert st.input_file == b"/data/in.csv" and st.output_file == b"/data/out.csv"
    assert st.write_csv_header == b"a,b,c,d,e,f\n" and st.write_csv_align_fmt == b"%s|%s"
    assert (st.write_csv_seq1_pos, st.write_csv_score_pos, st.write_csv_align_pos, st.similarity_analysis) == (2, 0, 1, 0)
    assert sa.validate_settings(st)[0] == 0


def test_settings_validation_rules():
    st = sa.default_settings()
    st.gap_penalty = 0
    assert sa.validate_settings(st)[0] != 0
    st = sa.default_settings()
    st.write_csv_score_pos = 0          # collides with sequence 1
    assert sa.validate_settings(st)[0] != 0
    st = sa.default_settings()
    st.write_csv_align_fmt = b"only one %s"
    assert sa.validate_settings(st)[0] != 0
    st = sa.default_settings()
    st.write_csv_header = b"too,few,columns\n"
    assert sa.validate_settings(st)[0] != 0


# ---- data set tool (C): same files as the Python generator, mega layout of scripts/create_mega_dataset.py:121-167

DATASET_TOOL = os.path.join(os.path.dirname(_lib.lib_path()), "sa_dataset")


def test_c_dataset_tool_equals_python_generator(tmp_path):
    import subprocess
    c_out, p_out = tmp_path / "c.csv", tmp_path / "p.csv"
    subprocess.run([DATASET_TOOL, "synth", "700", "1", "64", "42", str(c_out)], check=True)
    seqs, labels = synth.generate(700, 1, 64, 42)
    synth.write_csv(str(p_out), seqs, labels)
    assert c_out.read_bytes() == p_out.read_bytes()


def test_c_dataset_tool_mega_layout(tmp_path):
    import subprocess
    src = tmp_path / "in.csv"
    src.write_bytes(b"h1,h2\nAAA,0\nCCC,1\nDDD,0\n")
    out = tmp_path / "mega.csv"
    subprocess.run([DATASET_TOOL, "mega", str(src), str(out), "2"], check=True)
    body = b"AAA,0\nCCC,1\nDDD,0\n"
    rev = b"DDD,0\nCCC,1\nAAA,0\n"
    assert out.read_bytes() == b"h1,h2\n" + (body + rev) * 2
    p_out = tmp_path / "pmega.csv"
    synth.make_mega(str(src), str(p_out), 2)
    assert p_out.read_bytes() == out.read_bytes()


def test_c_dataset_tool_usage_error():
    import subprocess
    assert subprocess.run([DATASET_TOOL, "nothing"], capture_output=True).returncode == 2


def test_tiles_per_item_follow_the_size_of_the_row_range():
    """Unless set, a work item takes 1 .. 16 tiles: 16 when the range holds at least 32 such items per workgroup,
    fewer on small ranges (a host-only context counts 256 CUs)."""
    seqs, _ = synth.generate(3000, 20, 21, 5)
    al = sa.Aligner(device=sa.DEVICE_NONE)
    al.upload(seqs)
    small = al.work_items(sa.TRIANGULAR, 0, 2999)
    assert max(int(it[2]) for it in small.tolist()) == 1          # 4.5 M pairs: 35 000 tiles for 4 000 workgroups
    al.set_tuning(tiles_per_item=4)
    assert max(int(it[2]) for it in al.work_items(sa.TRIANGULAR, 0, 2999).tolist()) == 4


def _write_matrix(path, matrix, header=True):
    letters = "ARNDCQEGHILKMFPSTWYV"
    with open(path, "w") as f:
        f.write("# substitution matrix, rows = residue of sequence 1\n")
        if header:
            f.write("   " + "  ".join(letters) + "\n")
        for a in range(20):
            row = " ".join("%3d" % matrix[a * 20 + b] for b in range(20))
            f.write((letters[a] + " " if header else "") + row + "\n")


def test_scoring_matrix_file_loader(tmp_path):
    """SA_SCORING_MATRIX: 400 integers in alphabet order; comment lines, a header row of letters and a leading
    letter per row are skipped; anything else is refused with a message."""
    import ctypes as C
    import random
    lib = sa.load_library()
    rng = random.Random(4)
    matrix = [rng.randint(-9, 14) for _ in range(400)]
    S = (C.c_int32 * 400)()
    msg = C.create_string_buffer(200)
    for header in (True, False):
        path = tmp_path / ("m%d.txt" % header)
        _write_matrix(str(path), matrix, header)
        assert lib.sa_load_matrix_file(str(path).encode(), S, msg, 200) == 0, msg.value
        assert list(S) == matrix
    commas = tmp_path / "commas.csv"
    commas.write_text("\n".join(",".join(str(matrix[a * 20 + b]) for b in range(20)) for a in range(20)))
    assert lib.sa_load_matrix_file(str(commas).encode(), S, msg, 200) == 0 and list(S) == matrix
    short = tmp_path / "short.txt"
    short.write_text(" ".join(str(x) for x in matrix[:399]))
    assert lib.sa_load_matrix_file(str(short).encode(), S, msg, 200) == -1 and b"399" in msg.value
    junk = tmp_path / "junk.txt"
    junk.write_text(" ".join(str(x) for x in matrix[:100]) + " 3.5 " + " ".join(str(x) for x in matrix[100:]))
    assert lib.sa_load_matrix_file(str(junk).encode(), S, msg, 200) == -1 and b"3.5" in msg.value
    assert lib.sa_load_matrix_file(str(tmp_path / "missing.txt").encode(), S, msg, 200) == -1
    # the setting: read from a user.h-style file, checked by the validator
    good = tmp_path / "m1.txt"
    cfg = tmp_path / "user.h"
    cfg.write_text('#define GAP_PENALTY -3\n#define SA_SCORING_MATRIX "%s"\n' % good)
    st = sa.load_settings(str(cfg))
    assert st.scoring_matrix_file.decode() == str(good) and st.gap_penalty == -3
    rc, text = sa.validate_settings(st)
    assert rc == 0, text
    st.scoring_matrix_file = str(short).encode()
    rc, text = sa.validate_settings(st)
    assert rc != 0 and "399" in text


def test_bench_refuses_more_gpus_than_there_are():
    """`bench.py --gpus N` outside a torchrun environment starts its ranks itself, or says why it cannot: it must
    not measure one GPU and report it as N."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], env=env, capture_output=True,
                         text=True, timeout=200)
    assert out.returncode != 0 and "--gpus 64" in out.stderr and out.stdout.strip() == ""


def test_compact_record_arithmetic_without_a_device():
    """Where a pair's packed record starts (sa_compact_pair_offset) and how long it is (sa_compact_record_size), against
    the plain sums, for both schedules, peptides (8-byte header, a byte per row) and long sequences (16 + 2 per row)."""
    import random
    rng = random.Random(4)
    for lo, hi, header, per_row in ((1, 64, 8, 1), (10, 300, 16, 2)):
        seqs = ["".join(rng.choice("ACDEFGHIK") for _ in range(rng.randint(lo, hi))) for _ in range(41)]
        al = sa.Aligner(device=sa.DEVICE_NONE, max_seq_len=hi)
        al.upload(seqs)
        n = len(seqs)
        size = [(header + per_row * len(s) + 3) & ~3 for s in seqs]
        assert [al.compact_record_size(j) for j in range(n)] == size
        at = 0
        for i in range(n - 1):
            for j in range(i + 1, n):
                assert al.compact_pair_offset(sa.TRIANGULAR, i, j) == at
                at += size[j]
        assert al.compact_bytes(sa.TRIANGULAR, 0, n - 1) == at
        assert al.compact_bytes(sa.TRIANGULAR, 3, 7) == al.compact_pair_offset(sa.TRIANGULAR, 7, 8) - al.compact_pair_offset(sa.TRIANGULAR, 3, 4)
        at = 0
        for k in range(n - 1):
            assert al.compact_pair_offset(sa.ADJACENT, k, k + 1) == at
            at += size[k + 1]
        assert al.compact_bytes(sa.ADJACENT, 0, n - 1) == at
        assert al.compact_bytes(sa.ADJACENT, 5, 9) == sum(size[6:10])
        al.close()


def test_lazy_tables_built_once_under_concurrent_queries():
    """The per-sequence tables behind sa_cell_count / sa_shard_rows / sa_compact_bytes are built on first use, not at
    upload: eight threads asking at once straight after an upload (ctypes drops the GIL, so they really do run together)
    must all see finished tables and the answers a single thread gets."""
    import random
    import threading
    rng = random.Random(11)
    seqs = ["".join(rng.choice("ACDEFGHIKLMNPQRSTVWY") for _ in range(rng.randint(1, 64))) for _ in range(3000)]
    n = len(seqs)
    ref = sa.Aligner(device=sa.DEVICE_NONE)
    ref.upload(seqs)
    want = (ref.cell_count(sa.TRIANGULAR, 0, n - 1), ref.compact_bytes(sa.TRIANGULAR, 0, n - 1),
            ref.compact_bytes(sa.ADJACENT, 0, n - 1), list(ref.shard_rows(sa.TRIANGULAR, 8)))
    ref.close()
    for _ in range(5):
        al = sa.Aligner(device=sa.DEVICE_NONE)
        al.upload(seqs)
        got, start = [None] * 8, threading.Barrier(8)

        def ask(k):
            start.wait()
            if k % 2:
                got[k] = (al.cell_count(sa.TRIANGULAR, 0, n - 1), al.compact_bytes(sa.TRIANGULAR, 0, n - 1),
                          al.compact_bytes(sa.ADJACENT, 0, n - 1), list(al.shard_rows(sa.TRIANGULAR, 8)))
            else:
                s = list(al.shard_rows(sa.TRIANGULAR, 8))
                got[k] = (al.cell_count(sa.TRIANGULAR, 0, n - 1), al.compact_bytes(sa.TRIANGULAR, 0, n - 1),
                          al.compact_bytes(sa.ADJACENT, 0, n - 1), s)

        threads = [threading.Thread(target=ask, args=(k,)) for k in range(8)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert all(g == want for g in got)
        al.close()
