"""Threaded stand-in for mpi4py.MPI (TEST INFRASTRUCTURE, see __init__.py).

Only the calls the reference makes on its PIC-cycle path exist: COMM_WORLD.rank / .size,
Isend / Irecv (+ Request.Wait), bcast, allgather, gather, barrier.  A rank is a thread:
`set_rank(r)` binds the calling thread to rank r, messages are numpy copies in queues keyed
by (source, destination, tag)."""
import queue
import threading
import numpy as np

_tl = threading.local()
_state = {'size': 1, 'barrier': threading.Barrier(1)}
_mail = {}
_lock = threading.Lock()
TIMEOUT = 1800.


def set_world(size):
    _state['size'] = size
    _state['barrier'] = threading.Barrier(size)
    with _lock:
        _mail.clear()


def set_rank(rank):
    _tl.rank = rank


def _box(key):
    with _lock:
        if key not in _mail:
            _mail[key] = queue.Queue()
        return _mail[key]


class Request(object):
    def __init__(self, fn=None):
        self._fn = fn

    def Wait(self):
        if self._fn is not None:
            self._fn()
            self._fn = None


class _Comm(object):
    @property
    def rank(self):
        return getattr(_tl, 'rank', 0)

    @property
    def size(self):
        return _state['size']

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def Isend(self, buf, dest, tag=0):
        _box((self.rank, dest, tag)).put(np.array(buf, copy=True))
        return Request()

    def Irecv(self, buf, source, tag=0):
        me = self.rank

        def complete():
            data = _box((source, me, tag)).get(timeout=TIMEOUT)
            np.copyto(buf, data.reshape(np.shape(buf)))
        return Request(complete)

    def barrier(self):
        _state['barrier'].wait(timeout=TIMEOUT)

    Barrier = barrier

    def bcast(self, obj, root=0):
        if self.size == 1:
            return obj
        if self.rank == root:
            for r in range(self.size):
                if r != root:
                    _box(('bcast', root, r)).put(obj)
            return obj
        return _box(('bcast', root, self.rank)).get(timeout=TIMEOUT)

    def allgather(self, x):
        for r in range(self.size):
            _box(('allgather', self.rank, r)).put(x)
        return [_box(('allgather', r, self.rank)).get(timeout=TIMEOUT) for r in range(self.size)]

    def gather(self, x, root=0):
        _box(('gather', self.rank, root)).put(x)
        if self.rank != root:
            return None
        return [_box(('gather', r, root)).get(timeout=TIMEOUT) for r in range(self.size)]


    def Gatherv(self, sendbuf, recvbuf, root=0):
        """sendbuf = [array, count]; recvbuf = [array, counts, displacements, type] on root."""
        data = np.ascontiguousarray(sendbuf[0]).ravel()[:int(sendbuf[1])].copy()
        _box(('gatherv', self.rank, root)).put(data)
        if self.rank == root:
            out, counts, displs = recvbuf[0], recvbuf[1], recvbuf[2]
            flat = out.reshape(-1)
            for r in range(self.size):
                d = _box(('gatherv', r, root)).get(timeout=TIMEOUT)
                flat[int(displs[r]):int(displs[r]) + int(counts[r])] = d

    def Scatterv(self, sendbuf, recvbuf, root=0):
        """sendbuf = [array, counts, displacements, type] on root; recvbuf = [array, count]."""
        if self.rank == root:
            flat = np.ascontiguousarray(sendbuf[0]).reshape(-1)
            counts, displs = sendbuf[1], sendbuf[2]
            for r in range(self.size):
                _box(('scatterv', root, r)).put(flat[int(displs[r]):int(displs[r]) + int(counts[r])].copy())
        d = _box(('scatterv', root, self.rank)).get(timeout=TIMEOUT)
        recvbuf[0].reshape(-1)[:int(recvbuf[1])] = d


COMM_WORLD = _Comm()
REAL4 = 'REAL4'
REAL8 = 'REAL8'
COMPLEX8 = 'COMPLEX8'
COMPLEX16 = 'COMPLEX16'
UINT64_T = 'UINT64_T'
