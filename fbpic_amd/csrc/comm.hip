// Guard-cell / particle transport between the z-slabs of neighbouring ranks: RCCL point-to-point
// inside the library (replaces the MPI Isend / Irecv / Wait of
// fbpic/boundaries/boundary_communicator.py:674-707).  One grouped launch per exchange on the
// CALLER'S stream: the messages are ordered after the kernels that packed them and before the
// kernels that unpack them, with no host synchronisation - on MI355X the payload moves GPU to
// GPU over xGMI.
//
// RCCL is bound at run time (dlopen / dlsym of the seven entry points used): a process that
// already holds an RCCL - PyTorch ships its own copy, torch.distributed's "nccl" backend loads
// it - must not get a second one, and single-GPU users need no RCCL at all.
#include "fb_common.h"
#include <dlfcn.h>
#include <string.h>

namespace fb {

typedef struct { char internal[128]; } RcclUniqueId;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES)
typedef void *RcclComm;                                    // ncclComm_t
constexpr int RCCL_INT8 = 0;                               // ncclInt8 / ncclChar

struct Rccl {
    int (*GetUniqueId)(RcclUniqueId *);
    int (*CommInitRank)(RcclComm *, int, RcclUniqueId, int);
    int (*CommDestroy)(RcclComm);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    int (*Send)(const void *, size_t, int, int, RcclComm, hipStream_t);
    int (*Recv)(void *, size_t, int, int, RcclComm, hipStream_t);
    const char *(*GetErrorString)(int);
    bool ok;
};

static Rccl *rccl()
{
    static Rccl r = {};
    static bool tried = false;
    if (tried) return r.ok ? &r : nullptr;
    tried = true;
    void *h = nullptr;
    // 1) the copy this process already loaded (torch's), 2) the ROCm one
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !h; pass++)
        for (const char *n : names) {
            h = dlopen(n, (pass == 0 ? RTLD_NOLOAD : 0) | RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    if (!h) { set_error("rccl", "librccl.so not found (dlopen)"); return nullptr; }
#define SYM(field, name) \
    *(void **)(&r.field) = dlsym(h, name); \
    if (!r.field) { set_error("rccl", "symbol " name " missing in librccl.so"); return nullptr; }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    r.ok = true;
    return &r;
}

static int rcheck(Rccl *R, int e, const char *where)
{
    if (e == 0) return 0;
    set_error(where, R->GetErrorString(e));
    return e;
}

}  // namespace fb

using namespace fb;

extern "C" int fb_comm_unique_id(void *id128)
{
    Rccl *R = rccl();
    if (!R) return -1;
    RcclUniqueId id;
    int e = rcheck(R, R->GetUniqueId(&id), "fb_comm_unique_id");
    if (e) return e;
    memcpy(id128, id.internal, 128);
    return 0;
}

extern "C" int fb_comm_init(const void *id128, int rank, int size, void **comm)
{
    Rccl *R = rccl();
    if (!R) return -1;
    if (!comm || rank < 0 || rank >= size) { set_error("fb_comm_init", "bad rank / size"); return -1; }
    RcclUniqueId id;
    memcpy(id.internal, id128, 128);
    RcclComm c = nullptr;
    int e = rcheck(R, R->CommInitRank(&c, size, id, rank), "fb_comm_init");
    if (e) return e;
    *comm = c;
    return 0;
}

extern "C" int fb_comm_destroy(void *comm)
{
    Rccl *R = rccl();
    if (!R) return -1;
    if (!comm) return 0;
    return rcheck(R, R->CommDestroy((RcclComm)comm), "fb_comm_destroy");
}

extern "C" int fb_exchange(void *comm, int left_rank, int right_rank,
                           const void *send_left, size_t send_left_bytes,
                           const void *send_right, size_t send_right_bytes,
                           void *recv_left, size_t recv_left_bytes,
                           void *recv_right, size_t recv_right_bytes, void *stream)
{
    Rccl *R = rccl();
    if (!R) return -1;
    if (!comm) { set_error("fb_exchange", "no communicator (fb_comm_init)"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    RcclComm c = (RcclComm)comm;
    const bool l = left_rank >= 0, r = right_rank >= 0;
    int e = rcheck(R, R->GroupStart(), "fb_exchange");
    if (e) return e;
    // Messages between one pair of ranks match in posting order.  When both neighbours are the
    // same rank (2-rank periodic ring, or the 1-rank loopback of the tests) the peer's first
    // send is its send-to-left, which is my message FROM THE RIGHT: that receive comes first.
    const bool same_peer = l && r && left_rank == right_rank;
    if (l && send_left_bytes) e = e ? e : R->Send(send_left, send_left_bytes, RCCL_INT8, left_rank, c, s);
    if (r && send_right_bytes) e = e ? e : R->Send(send_right, send_right_bytes, RCCL_INT8, right_rank, c, s);
    if (same_peer) {
        if (recv_right_bytes) e = e ? e : R->Recv(recv_right, recv_right_bytes, RCCL_INT8, right_rank, c, s);
        if (recv_left_bytes) e = e ? e : R->Recv(recv_left, recv_left_bytes, RCCL_INT8, left_rank, c, s);
    } else {
        if (l && recv_left_bytes) e = e ? e : R->Recv(recv_left, recv_left_bytes, RCCL_INT8, left_rank, c, s);
        if (r && recv_right_bytes) e = e ? e : R->Recv(recv_right, recv_right_bytes, RCCL_INT8, right_rank, c, s);
    }
    const int e2 = R->GroupEnd();
    if (e) return rcheck(R, e, "fb_exchange");
    return rcheck(R, e2, "fb_exchange");
}
