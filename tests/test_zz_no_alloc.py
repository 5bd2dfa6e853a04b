"""sa_align_adjacent_noalloc (include/sa_hip.h): the ADJACENT peptide path with caller-owned device
buffers and a fixed scratch array inside the library, against sa_submit / sa_next on the same
sequences.  Bar: byte-identical records (integer and byte work), on two calls in a row, the second
with a smaller input, so that whatever the first call left in the scratch array is seen to do no harm.
"""
import ctypes as C
import random

import numpy as np
import pytest
import torch

import sequencealigner_amd as sa
from sequencealigner_amd import _lib, synth

pytestmark = pytest.mark.gpu
ALPHABET = "ARNDCQEGHILKMFPSTWYV"


def _device_inputs(seqs):
    """64-byte slots of residue codes (padding 20) and uint16 lengths, as torch tensors on cuda:0."""
    lut = np.full(256, 255, dtype=np.uint8)
    lut[np.frombuffer(ALPHABET.encode(), dtype=np.uint8)] = np.arange(20, dtype=np.uint8)
    slots = np.full((len(seqs), 64), 20, dtype=np.uint8)
    lens = np.zeros(len(seqs), dtype=np.uint16)
    for k, s in enumerate(seqs):
        b = s if isinstance(s, bytes) else s.encode()
        slots[k, : len(b)] = lut[np.frombuffer(b, dtype=np.uint8)]
        lens[k] = len(b)
    return torch.from_numpy(slots).cuda(), torch.from_numpy(lens.view(np.int16)).cuda()


def _noalloc(seqs, stride, matrix=None, gap=-4, device=0, check_nothing_allocated=False):
    lib = sa.load_library()
    with torch.cuda.device(device):
        d_slots, d_lens = _device_inputs(seqs)
        out = torch.full(((len(seqs) - 1) * stride,), 0xEE, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
    mat = None
    if matrix is not None:
        mat = (C.c_int32 * 400)(*matrix)
    current = torch.cuda.current_device()
    free_before = torch.cuda.mem_get_info(device)[0]
    rc = lib.sa_align_adjacent_noalloc(device, d_slots.data_ptr(), d_lens.data_ptr(), len(seqs), mat, gap, out.data_ptr(), stride)
    assert torch.cuda.current_device() == current          # the caller's device is put back
    hip_device = C.c_int(-1)
    assert C.CDLL("libamdhip64.so").hipGetDevice(C.byref(hip_device)) == 0 and hip_device.value == current
    if check_nothing_allocated:
        # what the driver reports as free on the card is what it was before the call: nothing allocated, nothing freed
        assert torch.cuda.mem_get_info(device)[0] == free_before
    return rc, out.cpu().numpy().reshape(len(seqs) - 1, stride)


def _reference_records(seqs, matrix=None, gap=-4):
    al = sa.Aligner(gap=gap, matrix=matrix, max_seq_len=64)
    al.upload(seqs)
    rec = al.align(sa.ADJACENT)
    stride = al.stride
    al.close()
    return rec, stride


def test_noalloc_entry_point_equals_submit_on_two_calls_in_a_row():
    # first call: many pairs, every length up to 64, several work items per workgroup
    big, _ = synth.generate(150000, 1, 64, seed=11)
    want_big, stride_big = _reference_records(big)
    rc, got_big = _noalloc(big, stride_big)
    assert rc == 0, sa.load_library().sa_strerror(rc)
    assert np.array_equal(got_big, want_big)
    # second call, smaller and different: shorter sequences, another table and gap, fewer items than
    # the first call left in the scratch array
    rng = random.Random(5)
    matrix = [rng.randint(-6, 12) for _ in range(400)]
    small = ["".join(rng.choice(ALPHABET) for _ in range(rng.randint(1, 40))) for _ in range(777)]
    want_small, stride_small = _reference_records(small, matrix=matrix, gap=-3)
    assert stride_small < stride_big
    rc, got_small = _noalloc(small, stride_small, matrix=matrix, gap=-3, check_nothing_allocated=True)
    assert rc == 0, sa.load_library().sa_strerror(rc)
    assert np.array_equal(got_small, want_small)
    # and the first input once more: same bytes as before
    rc, again = _noalloc(big, stride_big, check_nothing_allocated=True)
    assert rc == 0 and np.array_equal(again, want_big)
    if torch.cuda.device_count() > 1:
        # another GPU has its own copy of the scratch array at its own address; device 0 stays the current one
        rc, other = _noalloc(small, stride_small, matrix=matrix, gap=-3, device=1)
        assert rc == 0 and np.array_equal(other, want_small)


def test_noalloc_entry_point_rejects_what_it_cannot_serve():
    lib = sa.load_library()
    seqs = ["ACDEFGHIK", "ACDF", "W" * 64]
    rc, _ = _noalloc(seqs, 64)             # 8 + 64 bytes do not fit a 64-byte record
    assert rc == -3
    rc, rec = _noalloc(seqs, 80)
    assert rc == 0
    d_slots, d_lens = _device_inputs(seqs)
    out = torch.zeros(80 * 2, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    too_many = _lib.NOALLOC_MAX_PAIRS + 2
    assert lib.sa_align_adjacent_noalloc(0, d_slots.data_ptr(), d_lens.data_ptr(), too_many, None, -4, out.data_ptr(), 80) == -8
    assert b"scratch" in lib.sa_strerror(-8)
    assert lib.sa_align_adjacent_noalloc(0, d_slots.data_ptr(), d_lens.data_ptr(), 3, None, -4, out.data_ptr(), 72) == -1
    assert lib.sa_align_adjacent_noalloc(0, None, d_lens.data_ptr(), 3, None, -4, out.data_ptr(), 80) == -1
    # a residue code outside the alphabet inside a sequence
    bad = d_slots.clone()
    bad[1, 2] = 23
    torch.cuda.synchronize()
    assert lib.sa_align_adjacent_noalloc(0, bad.data_ptr(), d_lens.data_ptr(), 3, None, -4, out.data_ptr(), 80) == -2
    # a zero length
    bad_len = d_lens.clone()
    bad_len[0] = 0
    torch.cuda.synchronize()
    assert lib.sa_align_adjacent_noalloc(0, d_slots.data_ptr(), bad_len.data_ptr(), 3, None, -4, out.data_ptr(), 80) == -3
    # after the refusals the entry point still works
    rc, rec2 = _noalloc(seqs, 80)
    assert rc == 0 and np.array_equal(rec, rec2)
