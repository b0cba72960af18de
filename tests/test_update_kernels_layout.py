"""CPU: index algebra of two update kernels restated in Python.

rc_dense_update_rows_dev (csrc/dense_opt.hip): a workgroup owns a 4,096-element chunk of a [rows, row_w] tensor and finds an
element's row flag as row0 + (rem0 + i) / row_w with 32-bit arithmetic inside the chunk -- must equal (base + i) / row_w for every
chunk and width, and in the float4 path (row_w % 4 == 0) a float4 must never straddle two rows.

plan_rows_kernel's indexed body (csrc/plan_update.hip): lane i of a wave resolves row base + i; the data phase hands rows
r0 + h * GPW + grp to lane-group grp -- every one of the 64 rows exactly once, for every row width.
Reference semantics: torch.optim.Adam / SGD over nn.Embedding tables (helpers/BaseRunner.py:110-114,206)."""
import numpy as np
import pytest

CHUNK = 256 * 16


@pytest.mark.parametrize("row_w", [1, 3, 4, 20, 64, 128, 4096, 5000])
def test_row_of_an_element_inside_a_chunk(row_w):
    n_rows = 3 * CHUNK // row_w + 7
    n = n_rows * row_w
    for chunk in range((n + CHUNK - 1) // CHUNK):
        base = chunk * CHUNK
        length = min(CHUNK, n - base)
        row0 = base // row_w
        rem0 = base - row0 * row_w
        i = np.arange(length, dtype=np.uint32)
        got = row0 + (np.uint32(rem0) + i) // np.uint32(row_w)
        assert np.array_equal(got, (base + i.astype(np.int64)) // row_w)
        if row_w % 4 == 0:
            i4 = np.arange(0, length - length % 4, 4)
            assert np.array_equal((base + i4) // row_w, (base + i4 + 3) // row_w)     # a float4 lies inside one row


@pytest.mark.parametrize("d", [16, 32, 64, 128, 256])
def test_indexed_rows_body_hands_out_every_row_once(d):
    lpr, h_rows = d // 4, 2
    gpw = 64 // lpr
    seen = np.zeros(64, dtype=int)
    for r0 in range(0, 64, gpw * h_rows):
        for h in range(h_rows):
            for grp in range(gpw):
                sl = r0 + h * gpw + grp
                assert 0 <= sl < 64
                seen[sl] += 1
    assert (seen == 1).all()
