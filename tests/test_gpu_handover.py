"""Particle hand-over kernels of the library (csrc/handover.hip): selection by the ownership rule
of the reference's CPU path (fbpic/boundaries/particle_buffer_handling.py:58-172: left if
z < zbox_min, right if z > zbox_max), packing into the fixed-size messages, compaction and
append - against NumPy set arithmetic on the same inputs (bit-exact: the kernels only move
values).  The Python protocol around them (one exchange, counts on the device, remainder
message on overflow) is exercised rank by rank against the reference running decomposed in
tests/test_gpu_multirank_golden.py and tests/test_gpu_lwfa.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HEADER = 8


def _setup(n, seed):
    import torch
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(8, n))
    A[2] = rng.uniform(0., 1., n)              # z
    A[7] = np.arange(n) + 0.5                  # a unique tag per particle
    return A, [torch.tensor(A[k], device='cuda') for k in range(8)]


def _select(arrs, n, zlo, zhi, cap_l, cap_r, idx_cap, prefix=None, cuts=(-1, -1, -1, -1),
            has_l=True, has_r=True):
    import torch
    from fbpic_amd import _capi
    lib, p, pa = _capi.lib(), _capi.ptr, _capi.ptr_array
    sl = torch.full((HEADER + 8 * cap_l,), np.nan, dtype=torch.float64, device='cuda') if has_l else None
    sr = torch.full((HEADER + 8 * cap_r,), np.nan, dtype=torch.float64, device='cuda') if has_r else None
    idx = torch.full((2, idx_cap), -1, dtype=torch.int32, device='cuda')
    counts = torch.full((8,), 77, dtype=torch.int64, device='cuda')
    _capi.check(lib.fb_handover_select_pack(n, p(arrs[2]), p(prefix), cuts[0], cuts[1], cuts[2], cuts[3],
                                            zlo, zhi, 8, pa(arrs), cap_l, cap_r, idx_cap, p(sl), p(sr),
                                            p(idx[0]), p(idx[1]), p(counts), _capi.stream()),
                'fb_handover_select_pack')
    torch.cuda.synchronize()
    return sl, sr, idx.cpu().numpy(), counts.cpu().numpy()


@pytest.mark.parametrize('n', [0, 1, 63, 5000, 200001])
def test_select_pack_full_scan(n):
    A, arrs = _setup(n, 1)
    zlo, zhi = 0.1, 0.85
    exp_l, exp_r = np.nonzero(A[2] < zlo)[0], np.nonzero(A[2] > zhi)[0]
    cap = max(len(exp_l), len(exp_r), 1) + 5
    sl, sr, idx, counts = _select(arrs, n, zlo, zhi, cap, cap, cap)
    assert counts[0] == len(exp_l) and counts[1] == len(exp_r)
    for buf, exp, row in ((sl, exp_l, 0), (sr, exp_r, 1)):
        b = buf.cpu().numpy()
        assert b[0] == len(exp)                                  # header = count
        got_idx = idx[row, :len(exp)]
        assert np.array_equal(np.sort(got_idx), exp)             # the rule, bit-exact
        rows = b[HEADER:].reshape(8, cap)[:, :len(exp)]
        assert np.array_equal(rows, A[:, got_idx])               # packed values = the selected particles


def test_select_pack_header_is_the_final_count_every_time():
    """The message header is written by the last workgroup to finish - after EVERY wave of every
    workgroup has added its particles.  (Rounds 3-4: thread 0 of a workgroup reported it finished
    without waiting for its other waves; once in ~20 runs of the 8-slab C4 test a header was a few
    waves' worth of particles short of the count the host read later, the receiver posted a
    remainder message 64 particles shorter than the sender's, and the transport aborted that rank.)
    Many launches with most particles leaving, headers against the counts and the expected sets."""
    import torch
    n = 1 << 20
    A, arrs = _setup(n, 5)
    zlo, zhi = 0.45, 0.55
    nl, nr = int((A[2] < zlo).sum()), int((A[2] > zhi).sum())
    cap = 1024                      # (tiny messages: the launch is the selection and the counting)
    for rep in range(60):
        sl, sr, idx, counts = _select(arrs, n, zlo, zhi, cap, cap, n)
        assert counts[0] == nl and counts[1] == nr
        assert float(sl[0]) == nl and float(sr[0]) == nr, rep


def test_select_pack_overflow_and_open_end():
    """More leavers than the message holds: the header still carries the full count, the index
    list is complete and the first `cap` are packed (the caller sends the rest in a second
    message).  A missing neighbour (open end): the leavers are listed but nothing is packed."""
    n = 40000
    A, arrs = _setup(n, 2)
    zlo, zhi = 0.2, 0.7
    exp_l, exp_r = np.nonzero(A[2] < zlo)[0], np.nonzero(A[2] > zhi)[0]
    cap = 1000
    sl, sr, idx, counts = _select(arrs, n, zlo, zhi, cap, cap, n, has_r=False)
    assert sr is None
    assert counts[0] == len(exp_l) > cap and counts[1] == len(exp_r) > cap
    assert np.array_equal(np.sort(idx[0, :len(exp_l)]), exp_l)
    assert np.array_equal(np.sort(idx[1, :len(exp_r)]), exp_r)
    b = sl.cpu().numpy()
    assert b[0] == len(exp_l)
    assert np.array_equal(b[HEADER:].reshape(8, cap), A[:, idx[0, :cap]])


def test_select_with_prefix_sum_equals_full_scan():
    """Cell-sorted arrays: only the cell rows next to the box edges are compared; everything
    before / after them leaves without a test.  Same sets as the full scan."""
    import torch
    n, Nz, Nrp = 60000, 64, 5
    A, _ = _setup(n, 3)
    z = A[2]
    iz = np.clip(np.ceil(z * Nz - 0.5).astype(int), 0, Nz - 1)      # iz_upper of cuda_sorting.py:68-88
    ir = np.random.default_rng(4).integers(0, Nrp, n)
    cell = ir + iz * Nrp
    order = np.argsort(cell, kind='stable')
    A = A[:, order]
    cell = cell[order]
    prefix = np.cumsum(np.bincount(cell, minlength=Nz * Nrp)).astype(np.int32)
    arrs = [torch.tensor(A[k], device='cuda') for k in range(8)]
    d_prefix = torch.tensor(prefix, device='cuda')
    ng = 8
    zlo, zhi = (ng - 0.3) / Nz, (Nz - ng + 0.2) / Nz           # inside the cell rows ng / Nz - ng
    rows = [ng - 1, ng + 2, Nz - ng - 1, Nz - ng + 2]
    cuts = tuple(r * Nrp - 1 for r in rows)
    exp_l, exp_r = np.nonzero(A[2] < zlo)[0], np.nonzero(A[2] > zhi)[0]
    cap = max(len(exp_l), len(exp_r)) + 3
    for prefix_arg, c in ((None, (-1, -1, -1, -1)), (d_prefix, cuts)):
        sl, sr, idx, counts = _select(arrs, n, zlo, zhi, cap, cap, cap, prefix=prefix_arg, cuts=c)
        assert counts[0] == len(exp_l) and counts[1] == len(exp_r)
        assert np.array_equal(np.sort(idx[0, :len(exp_l)]), exp_l)
        assert np.array_equal(np.sort(idx[1, :len(exp_r)]), exp_r)
        assert np.array_equal(sl.cpu().numpy()[HEADER:].reshape(8, cap)[:, :len(exp_l)], A[:, idx[0, :len(exp_l)]])


@pytest.mark.parametrize('frac', [0.0005, 0.02, 0.45])
def test_compact_and_append(frac):
    """After compaction the first n - n_leave slots hold exactly the survivors (any order); the
    arrivals follow, with the periodic shift applied to z only."""
    import torch
    from fbpic_amd import _capi
    lib, p, pa = _capi.lib(), _capi.ptr, _capi.ptr_array
    n = 30000
    A, _ = _setup(n, 5)
    cap_total = 2 * n
    arrs = [torch.zeros(cap_total, dtype=torch.float64, device='cuda') for _ in range(8)]
    for k in range(8):
        arrs[k][:n] = torch.tensor(A[k], device='cuda')
    zlo, zhi = frac, 1. - frac
    exp_l, exp_r = np.nonzero(A[2] < zlo)[0], np.nonzero(A[2] > zhi)[0]
    views = [a[:n] for a in arrs]
    sl, sr, idx, counts = _select(views, n, zlo, zhi, n, n, n)
    n_l, n_r = int(counts[0]), int(counts[1])
    d_idx = torch.tensor(idx, device='cuda')
    nb = int(lib.fb_handover_workspace_bytes(n_l + n_r))
    ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
    _capi.check(lib.fb_handover_compact(n, n_l, p(d_idx[0]), n_r, p(d_idx[1]), 8, pa(arrs), p(ws), nb,
                                        _capi.stream()), 'fb_handover_compact')
    m = n - n_l - n_r
    got = np.array([a[:m].cpu().numpy() for a in arrs])
    keep = np.ones(n, dtype=bool)
    keep[exp_l] = False
    keep[exp_r] = False
    ref = A[:, keep]
    assert got.shape == ref.shape
    assert np.array_equal(got[:, np.argsort(got[7])], ref[:, np.argsort(ref[7])])
    # append what left to the right, re-entering from the left of a periodic box of length 1
    cap = n
    _capi.check(lib.fb_handover_append_shift(n_r, m, 8, pa(arrs), p(sr[HEADER:]), cap, 2, -1., _capi.stream()),
                'fb_handover_append_shift')
    torch.cuda.synchronize()
    tail = np.array([a[m:m + n_r].cpu().numpy() for a in arrs])
    sent = A[:, idx[1, :n_r]].copy()
    sent[2] += -1.
    assert np.array_equal(tail, sent)
