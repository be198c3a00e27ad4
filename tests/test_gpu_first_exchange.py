"""The first message between two GPUs (runs whenever the box has >= 2; ordered - by file name - in
front of tests/test_gpu_multirank_golden.py::test_decomposed_on_real_gpus).

Two processes, one GPU each:  fb_comm_unique_id on rank 0 -> (file rendezvous) -> fb_comm_init on both
-> ONE fb_exchange of a checksummed 1 MiB buffer each way on the compute stream, stream-ordered behind
the kernel that fills it; then the same 1 MiB through the `torch` transport
(torch.distributed `nccl` = RCCL, batch_isend_irecv - what FBPIC_AMD_TRANSPORT=torch selects).
Replaces the MPI Isend / Irecv / Wait pair of
/root/reference/fbpic/boundaries/boundary_communicator.py:674-707.
A failure surfaces here, within a bounded time and with RCCL's own error string (fb_last_error),
instead of inside a 4-rank trajectory.

On the single-GPU boxes the test is skipped; what it would run is exercised rank-locally by
test_gpu_kernels.py::test_exchange_rccl_loopback.  The payload / checksum helpers and the
file rendezvous are covered on CPU by test_first_exchange_helpers (below, no gpu mark).
"""
import os
import socket
import tempfile
import time
import zlib
import numpy as np
import pytest

NBYTES = 1 << 20


def payload(rank, nbytes=NBYTES):
    """Deterministic, rank-specific bytes (so that a message delivered to the wrong side or from the
    wrong peer cannot pass) and their CRC32."""
    rng = np.random.default_rng(1234 + rank)
    a = rng.integers(0, 256, nbytes, dtype=np.uint8)
    return a, zlib.crc32(a.tobytes())


def publish_id(path, idbytes):
    """Rank 0 -> the others: the 128-byte RCCL unique id through a file, renamed into place."""
    tmp = path + '.tmp'
    with open(tmp, 'wb') as f:
        f.write(idbytes)
    os.replace(tmp, path)


def wait_id(path, timeout=30.):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise TimeoutError('no unique id from rank 0 after %.0f s' % timeout)
        time.sleep(0.01)
    with open(path, 'rb') as f:
        b = f.read()
    assert len(b) == 128
    return b


def test_first_exchange_helpers(tmp_path):
    a0, c0 = payload(0)
    a1, c1 = payload(1)
    assert a0.nbytes == NBYTES and c0 != c1 and zlib.crc32(payload(0)[0].tobytes()) == c0
    p = str(tmp_path / 'id')
    publish_id(p, bytes(range(128)))
    assert wait_id(p, 1.) == bytes(range(128))
    with pytest.raises(TimeoutError):
        wait_id(str(tmp_path / 'none'), 0.05)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, idfile, q):
    try:
        import ctypes
        os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
        import torch
        from fbpic_amd import _capi
        torch.cuda.set_device(rank)
        lib = _capi.lib()
        _capi.check(lib.fb_set_device(rank), 'fb_set_device')
        peer = 1 - rank
        mine, _ = payload(rank)
        _, crc_peer = payload(peer)
        # ---- the library's own transport
        if rank == 0:
            idbuf = ctypes.create_string_buffer(128)
            _capi.check(lib.fb_comm_unique_id(idbuf), 'fb_comm_unique_id')
            publish_id(idfile, idbuf.raw)
        idbuf = ctypes.create_string_buffer(wait_id(idfile), 128)
        comm = ctypes.c_void_p()
        _capi.check(lib.fb_comm_init(idbuf, rank, world, ctypes.byref(comm)), 'fb_comm_init')
        host = torch.from_numpy(mine)
        send_l = torch.empty(NBYTES, dtype=torch.uint8, device='cuda')
        send_r = torch.empty(NBYTES, dtype=torch.uint8, device='cuda')
        recv_l = torch.zeros(NBYTES, dtype=torch.uint8, device='cuda')
        recv_r = torch.zeros(NBYTES, dtype=torch.uint8, device='cuda')
        # filled on the compute stream, not synchronised: the messages are ordered behind these copies
        send_l.copy_(host, non_blocking=True)
        send_r.copy_(host.flip(0), non_blocking=True)
        t0 = time.time()
        _capi.check(lib.fb_exchange(comm, peer, peer, send_l.data_ptr(), NBYTES, send_r.data_ptr(), NBYTES,
                                    recv_l.data_ptr(), NBYTES, recv_r.data_ptr(), NBYTES, _capi.stream()),
                    'fb_exchange')
        torch.cuda.synchronize()
        dt_lib = time.time() - t0
        # 2-rank ring: what the peer sent to ITS left arrives from my right, and vice versa
        got_r = zlib.crc32(recv_r.cpu().numpy().tobytes())
        got_l = zlib.crc32(recv_l.cpu().numpy().flip(0).tobytes())
        assert got_r == crc_peer, 'fb_exchange: message from the right is not the peer\'s send-to-left'
        assert got_l == crc_peer, 'fb_exchange: message from the left is not the peer\'s send-to-right'
        _capi.check(lib.fb_comm_destroy(comm), 'fb_comm_destroy')
        # ---- the torch transport (RCCL under torch.distributed)
        import torch.distributed as dist
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world,
                                device_id=torch.device('cuda', rank))
        recv_l.zero_()
        recv_r.zero_()
        t0 = time.time()
        ops = [dist.P2POp(dist.isend, send_l, peer), dist.P2POp(dist.isend, send_r, peer),
               dist.P2POp(dist.irecv, recv_r, peer), dist.P2POp(dist.irecv, recv_l, peer)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        torch.cuda.synchronize()
        dt_torch = time.time() - t0
        assert zlib.crc32(recv_r.cpu().numpy().tobytes()) == crc_peer, 'torch transport: from the right'
        assert zlib.crc32(recv_l.cpu().numpy().flip(0).tobytes()) == crc_peer, 'torch transport: from the left'
        dist.destroy_process_group()
        q.put((rank, 'ok', 'fb_exchange %.1f ms (incl. connection set-up), torch %.1f ms' % (1e3 * dt_lib, 1e3 * dt_torch)))
    except BaseException as exc:                 # noqa: BLE001 - reported through the queue
        import traceback
        try:
            from fbpic_amd import _capi
            last = _capi.lib().fb_last_error().decode()
        except Exception:
            last = '?'
        q.put((rank, 'fail', '%s\nfb_last_error: %s\n%s' % (exc, last, traceback.format_exc())))


@pytest.mark.gpu
def test_first_message_between_two_gpus():
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs, %d visible' % torch.cuda.device_count())
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        idfile = os.path.join(d, 'rccl_id')
        procs = [ctx.Process(target=_rank, args=(r, 2, port, idfile, q)) for r in range(2)]
        for p in procs:
            p.start()
        res, t0 = {}, time.time()
        while len(res) < 2 and time.time() - t0 < 120.:
            try:
                r, st, msg = q.get(timeout=1.)
                res[r] = (st, msg)
                if st != 'ok':
                    break
            except Exception:
                if not any(p.is_alive() for p in procs) and q.empty():
                    break
        for p in procs:
            p.join(timeout=5.)
            if p.is_alive():
                p.kill()
    for r in range(2):
        assert r in res, 'rank %d did not report within 120 s (hung in RCCL?): %r' % (r, res)
        assert res[r][0] == 'ok', 'rank %d: %s' % (r, res[r][1])
        print('rank %d: %s' % (r, res[r][1]))
