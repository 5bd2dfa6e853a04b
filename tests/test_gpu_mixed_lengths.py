"""Sets that hold peptides and sequences of more than 64 residues: the context is a long-sequence one (records in the
long form), and the tiles whose sequences are all peptides run on the paired peptide kernel in its long-record form.
Bar: the records are byte for byte those of the long-sequence kernels alone, and equal to the CPU checker's."""
import random

import numpy as np
import pytest

import oracle_lib
import sequencealigner_amd as sa
from sequencealigner_amd import synth

pytestmark = pytest.mark.gpu
ALPHABET = "ARNDCQEGHILKMFPSTWYV"


def _mixed(n, lo, hi, n_long, long_lo, long_hi, seed, alphabet=ALPHABET):
    rng = random.Random(seed)
    seqs = ["".join(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))) for _ in range(n)]
    for _ in range(n_long):
        seqs.insert(rng.randrange(len(seqs) + 1), "".join(rng.choice(alphabet) for _ in range(rng.randint(long_lo, long_hi))))
    return seqs


def _check(al, seqs, records, pairs, gap=-4, matrix=None):
    for k, (i, j) in pairs:
        o1, o2, r = oracle_lib.align(seqs[i], seqs[j], gap=gap, matrix=matrix)
        a1, a2, st = al.expand(records[k], i, j)
        assert (a1, a2) == (o1, o2), (i, j)
        assert (st.score, st.columns, st.matches, st.mismatches, st.gaps, st.similarity_bp) == (
            r.score, r.columns, r.matches, r.mismatches, r.gaps, r.similarity_bp), (i, j)


@pytest.mark.parametrize("n,lo,hi,n_long,long_lo,long_hi,seed,long_kernel", [
    (1500, 6, 50, 12, 70, 200, 1, "sa_long_pair_kernel<4>"),                 # 52-column peptide kernel
    (900, 1, 64, 5, 300, 900, 2, "sa_long_ckpt_kernel<64, 64>"),             # 64-column one; checkpointing kernel for the rest
    (700, 30, 64, 200, 65, 90, 3, "sa_long_pair_kernel<4>"),                 # many sequences just above the border
    (400, 60, 64, 3, 65, 66, 4, "sa_long_pair_kernel<4>"),                   # the shortest long-form stride there is
])
def test_peptide_tiles_write_the_long_kernels_records(n, lo, hi, n_long, long_lo, long_hi, seed, long_kernel):
    seqs = _mixed(n, lo, hi, n_long, long_lo, long_hi, seed)
    total = len(seqs)
    al = sa.Aligner(max_seq_len=long_hi)
    al.upload(seqs)
    al.set_cooperative(-1)
    assert al.row_bytes == 2
    rows = al.rows(sa.TRIANGULAR)
    both = al.align(sa.TRIANGULAR)
    ng = 13 if max(len(s) for s in seqs if len(s) <= 64) <= 52 else 16
    assert al.last_kernel() == "%s + sa_peptide_pair_longrec_kernel<%d, 4>" % (long_kernel, ng)
    al.set_peptide_tiles(False)
    alone = al.align(sa.TRIANGULAR)
    assert al.last_kernel() == long_kernel
    assert np.array_equal(both, alone)
    # row ranges (their own work lists, some without a partner for the first sequence) and packed records
    al.set_peptide_tiles(True)
    cuts = [0, 1, rows // 3, rows // 3 + 7, rows]
    parts = [al.align(sa.TRIANGULAR, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(parts), both)
    packed_on = al.align_compact(sa.TRIANGULAR)
    al.set_peptide_tiles(False)
    packed_off = al.align_compact(sa.TRIANGULAR)
    assert np.array_equal(packed_on, packed_off)
    # against the checker: pairs of every kind
    rng = random.Random(seed)
    long_ix = [k for k, s in enumerate(seqs) if len(s) > 64]
    sample = []
    for _ in range(250):
        i = rng.randrange(total - 1)
        sample.append((i, rng.randrange(i + 1, total)))
    for i in long_ix[:4]:
        sample += [(min(i, j), max(i, j)) for j in (rng.randrange(total) for _ in range(6)) if j != i]
    index = lambda i, j: al.pair_index(sa.TRIANGULAR, i) + (j - i - 1)
    _check(al, seqs, both, [(index(i, j), (i, j)) for i, j in sample])
    al.close()


@pytest.mark.parametrize("gap,kernels", [
    (-14, "sa_long_pair_wide_kernel<4> + sa_peptide_pair_wide_longrec_kernel<16, 4>"),
    # S' up to 61: beyond the long kernels' packed cells (32-bit cells, one first sequence per wave), within the peptide kernel's
    (-23, "sa_long_kernel<0, 2> + sa_peptide_pair_wide_longrec_kernel<16, 4>"),
])
def test_peptide_tiles_with_a_table_beyond_the_half_precision_range(gap, kernels):
    """gap -14: S' up to 43, 64 residues x 43 = 2752: the peptide tiles take the wide form, and so do the long kernels."""
    seqs = _mixed(600, 6, 64, 6, 80, 260, 7)
    al = sa.Aligner(max_seq_len=260, gap=gap)
    al.upload(seqs)
    al.set_cooperative(-1)
    both = al.align(sa.TRIANGULAR)
    assert al.last_kernel() == kernels
    al.set_pairing(False)
    plain = al.align(sa.TRIANGULAR)
    assert al.last_kernel() == "sa_long_kernel<0, 2>"
    assert np.array_equal(both, plain)
    rng = random.Random(gap)
    total = len(seqs)
    sample = [(i, rng.randrange(i + 1, total)) for i in (rng.randrange(total - 1) for _ in range(60))]
    _check(al, seqs, both, [(al.pair_index(sa.TRIANGULAR, i) + j - i - 1, (i, j)) for i, j in sample], gap=gap)
    al.close()


def test_peptide_tiles_in_the_csv_driver_and_in_text_submissions(tmp_path):
    """End to end on a file of mixed lengths: rows formatted on the device and on the host, both equal to the checker's file."""
    seqs = _mixed(500, 6, 50, 8, 70, 400, 11)
    labels = [k % 2 for k in range(len(seqs))]
    src = tmp_path / "in.csv"
    synth.write_csv(str(src), [x.encode() for x in seqs], labels)
    want = tmp_path / "oracle.csv"
    oracle_lib.run_file(str(src), str(want), sa.TRIANGULAR, threads=8)
    for device_format in (1, 0):
        got = tmp_path / ("gpu%d.csv" % device_format)
        rep = sa.run_csv(str(src), str(got), sa.TRIANGULAR, device_format=device_format, max_seq_len=401, cooperative=-1)
        assert got.read_bytes() == want.read_bytes()


def test_peptide_tiles_survive_a_clone_and_a_second_upload():
    seqs = _mixed(500, 6, 50, 6, 70, 300, 13)
    al = sa.Aligner(max_seq_len=300)
    al.upload(seqs)
    al.set_cooperative(-1)
    first = al.align(sa.TRIANGULAR)
    other = sa.Aligner(max_seq_len=300)
    other.clone_from(al)
    other.set_cooperative(-1)
    assert np.array_equal(other.align(sa.TRIANGULAR), first)
    assert "longrec" in other.last_kernel()
    # a set without peptides next: no second launch; then the first set again
    longs = _mixed(0, 0, 0, 80, 70, 300, 14)
    al.upload(longs)
    al.align(sa.TRIANGULAR)
    assert al.last_kernel() == "sa_long_pair_kernel<4>"
    al.upload(seqs)
    assert np.array_equal(al.align(sa.TRIANGULAR), first)
    al.close()
    other.close()


def test_a_refused_allocation_leaves_the_context_usable():
    """The records of every pair at the stride of a 4096-residue sequence do not fit the card: sa_prepare_resident says
    out of memory, and the calls after it (which size their submissions themselves) work as if nothing had happened."""
    seqs = _mixed(12000, 66, 70, 1, 4096, 4096, 17)
    al = sa.Aligner(max_seq_len=4096)
    al.upload(seqs)
    rows = al.rows(sa.TRIANGULAR)
    assert al.pair_count(sa.TRIANGULAR, 0, rows) * al.stride > 400e9
    with pytest.raises(sa.SaError) as err:
        al.prepare_resident(sa.TRIANGULAR, 0, rows)
    assert "memory" in str(err.value)
    recs = al.align(sa.TRIANGULAR, 0, 2)
    _check(al, seqs, recs, [(k, (0, k + 1)) for k in range(0, 12000, 1500)])
    small = _mixed(300, 6, 50, 0, 0, 0, 18)
    al.upload(small)
    recs = al.align(sa.TRIANGULAR)
    _check(al, small, recs, [(k, (0, k + 1)) for k in range(0, 299, 30)])
    al.close()


def test_large_mixed_set_every_record_by_digest():
    """8 000 peptides with 20 sequences of 80-120 residues among them, resident: all 3.2e7 records, digested on the device,
    are the same with the peptide tiles on the peptide kernel and with every tile on the long-sequence kernel; sampled
    pairs against the checker."""
    seqs, _ = synth.generate(8000, 6, 50, 3)
    more, _ = synth.generate(20, 80, 120, 5)
    for k, s in enumerate(more):
        seqs.insert(k * 400 + k, s)
    total = len(seqs)
    al = sa.Aligner(max_seq_len=120)
    al.upload(seqs)
    rows = al.rows(sa.TRIANGULAR)
    pairs = al.pair_count(sa.TRIANGULAR, 0, rows)
    digests = []
    for tiles in (True, False):
        al.set_peptide_tiles(tiles)
        al.prepare_resident(sa.TRIANGULAR, 0, rows)
        al.run_resident()
        assert ("longrec" in al.last_kernel()) == tiles
        digests.append([al.digest_resident(a, min(4_000_000, pairs - a)) for a in range(0, pairs, 4_000_000)])
    assert digests[0] == digests[1]
    al.set_peptide_tiles(True)
    al.prepare_resident(sa.TRIANGULAR, 0, rows)
    al.run_resident()
    rng = random.Random(5)
    for _ in range(40):
        i = rng.randrange(total - 1)
        first = al.pair_index(sa.TRIANGULAR, i)
        n = min(30, total - 1 - i)
        recs = al.fetch_resident(first, n)
        _check(al, [x.decode() if isinstance(x, bytes) else x for x in seqs], recs, [(k, (i, i + 1 + k)) for k in range(0, n, 7)])
    al.close()
