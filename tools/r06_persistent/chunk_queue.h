// Persistent waves + a work queue for the kernels that walk the cell-sorted particle stream in chunks
// of 64 particles (round 6).
//
// Until round 5 such a launch was cut statically: wave w owned `chunks_per_wave` consecutive chunks
// (4 at C2: 16 384 one-wave workgroups for the 3 072 wave slots of the GPU, 5.33 per slot).  Two costs
// came with that, and they trade against each other, which is why every cpw between 2 and 6 measured
// the same:
//   * the launch ends with a generation of waves that fills a third of the slots (5.33 -> 6 wave
//     lifetimes per slot: ~11 % of the kernel's duration at 1/3 occupancy or less);
//   * every wave pays the prologue (engine set-up, lane roles, a cold first chunk whose loads nobody
//     has prefetched) and the final flush once per 4 chunks: 520 of ~5 000 VALU instructions.
// Here the launch holds as many waves as the GPU keeps resident (a few more do no harm: a late wave finds
// the queue empty) and a wave takes RANGES of chunks from a queue until it is empty: the set-up is paid
// once per wave lifetime (~21 chunks), the software pipeline (positions two chunks ahead, momenta and
// node values one) runs across range boundaries because the next range is requested - one atomic whose
// result is consumed a range later - while the current one is processed, and the waves stop within
// one or two chunks of each other because the ranges shrink towards the end (guided: half the remaining
// work of the XCD / the waves of the XCD, between 1 and `rmax` chunks).
//
// The XCD-contiguous walk of round 1 is kept: ONE QUEUE PER XCD over a contiguous eighth of the stream
// (workgroup b runs on XCD b % 8), so neighbouring cells meet in one L2 and the waves in flight work on
// a compact front of each eighth (also what keeps their HBM streams in few DRAM pages).  A wave whose
// own queue is empty takes from the next XCD's (balance over locality, at the very end only).
// The mapping b % 8 -> XCD is used for speed only: any mapping gives every chunk to exactly one wave.
//
// Atomics.  A device-scope atomic with a return value is executed at the memory side (the L2s of the XCDs are
// not coherent with each other) and costs ~10 ns of a serial resource per operation ON ONE LINE: with all
// eight counters in one 64-B line and a request per range of <= 4 chunks the first build of this file ran
// the C2 launch in 0.70 ms (R = 1: 1.21 ms) where the static cut took 0.27 (profiles/r06_queue_scan.txt).
// Hence (i) every counter has a 256-B block of its own, (ii) most of the stream is dealt out WITHOUT the queue:
// wave j of the W waves of an XCD owns ranges j, j + W, j + 2 W ... of the first `kstatic` rounds (the same
// moving front as the queue would give, no atomics), and only the last part of every XCD's eighth goes
// through the queue, where the balance is decided.
//
// Queue memory: (8 + 1) x 256 B per launch slot in a small device buffer owned by the library (q[0..7] next
// chunk of every XCD, q[8] waves that have finished).  The last wave to finish resets its slot, so no
// memset precedes a launch; launches that may overlap (different streams) get different slots
// (round-robin over 64).
#pragma once
#include "fb_common.h"

namespace fb {

// (every counter in a 256-B block of its own: see "atomics" in the header comment)
constexpr int FB_QUEUE_SLOTS = 64, FB_QUEUE_STRIDE = 64, FB_QUEUE_UINTS = (FB_NXCD + 1) * FB_QUEUE_STRIDE;

struct ChunkQueueArgs {
    unsigned *q;          // this launch's slot: q[x * FB_QUEUE_STRIDE] next dynamic chunk of XCD x (relative to
                          // the start of its dynamic region), q[FB_NXCD * FB_QUEUE_STRIDE] waves finished
    int nchunks;
    int per;              // chunks per XCD (its contiguous eighth of the stream)
    int rmax;             // chunks per range (static part), largest range (dynamic part)
    int W;                // waves per XCD (the launch has FB_NXCD * W one-wave workgroups)
    int nstat;            // chunks at the start of every XCD's eighth that are dealt out statically: wave j
                          // takes ranges j, j + W, j + 2 W ... of them
    int shift;            // guided range size of the dynamic part: remaining >> shift (~ remaining / 2 W)
};

// host: the library's queue buffer (allocated and zeroed on first use) and the slot of the next launch
unsigned *chunk_queue_slot();
// host: waves to launch for a kernel that keeps `per_cu` waves resident on each CU (cached device query)
int chunk_queue_waves(int per_cu, long nchunks);
// host: fill the plan (static_percent of every XCD's chunks are dealt out statically, in whole rounds)
void chunk_queue_plan(ChunkQueueArgs &Q, long nchunks, int nwaves, int rmax, int static_percent);

#ifdef __HIPCC__
// Wave-uniform walk over the chunks this wave processes.  next() returns the next chunk index or -1.
// The persistent state is a few 32-bit scalars and the arithmetic of a range change is adds, mins and a
// shift (the kernels that use it have neither SGPRs nor VALU issue slots to spare: everything else is
// precomputed by chunk_queue_plan); the launch constants are passed to every call by a functor `QA` - a
// kernel short of SGPRs re-reads them from its kernel-argument segment there.
struct ChunkWalk {
    // The state lives in the lanes of ONE vector register (read with v_readlane where it is needed,
    // written with v_writelane), not in scalar registers: the kernels that walk chunks are at the limit of
    // the 102 SGPRs of a wave, and seven more live scalars made the compiler move wave-uniform values of
    // the chunk loop into vector registers (k_cycle_linear<2>: 162 -> 197 VGPRs = one wave per SIMD less).
    enum { CUR = 0, CUR_END, SNEXT, XCD, TRIES, SEEN, PEND_N };
    int st;
    unsigned pend;            // lane 0: result of the outstanding request

    __device__ __forceinline__ int get(int k) const { return __builtin_amdgcn_readlane(st, k); }
    // (no v_writelane builtin in this compiler; `k` is a compile-time lane number)
    __device__ __forceinline__ void set(int k, int v) { asm("v_writelane_b32 %0, %1, %2" : "+v"(st) : "s"(v), "n"(k)); }

    // dynamic region of XCD x_: [dbeg, cend)
    static __device__ __forceinline__ void region(const ChunkQueueArgs &A, int x_, int &dbeg, int &cend)
    {
        const int cbeg = x_ * A.per;
        cend = min(cbeg + A.per, A.nchunks);
        dbeg = min(cbeg + A.nstat, cend);
    }
    __device__ __forceinline__ void request(const ChunkQueueArgs &A, int x, int seen)
    {
        int dbeg, cend;
        region(A, x, dbeg, cend);
        if (dbeg >= cend) { set(PEND_N, 0); return; }
        int n = ((cend - dbeg) - seen) >> (A.shift & 31);
        n = n < 1 ? 1 : (n > A.rmax ? A.rmax : n);
        set(PEND_N, n);
        pend = 0u;
        if ((threadIdx.x & 63) == 0) pend = atomicAdd(A.q + x * FB_QUEUE_STRIDE, (unsigned)n);
    }
    template <class QA> __device__ __forceinline__ void init(QA qa)
    {
        const ChunkQueueArgs A = qa();
        st = 0;
        pend = 0u;
        const int x = (int)(blockIdx.x % FB_NXCD);
        set(XCD, x);
        set(SNEXT, x * A.per + (int)(blockIdx.x / FB_NXCD) * A.rmax);
    }
    template <class QA> __device__ __forceinline__ int next(QA qa)
    {
        {
            const int cur = get(CUR);
            if (cur < get(CUR_END)) { set(CUR, cur + 1); return cur; }
        }
        const ChunkQueueArgs A = qa();
        int x = get(XCD);
        const int snext = get(SNEXT);
        if (snext >= 0) {
            int dbeg, cend;
            region(A, x, dbeg, cend);
            const int nxt = snext + A.W * A.rmax;
            const bool have = snext < dbeg;
            if (!have || nxt >= dbeg) {           // the last static range of this wave: the first dynamic one is
                set(SNEXT, -1);                   // asked for now and consumed a range later
                if (A.nstat < A.per) request(A, x, 0);
            } else {
                set(SNEXT, nxt);
            }
            if (have) {
                set(CUR, snext + 1);
                set(CUR_END, min(snext + A.rmax, dbeg));
                return snext;
            }
        }
        if (A.nstat >= A.per) return -1;          // everything was dealt out statically: no queue, no atomics
        int tries = get(TRIES);
        while (true) {
            const int n = get(PEND_N);
            if (n > 0) {
                const int r = (int)__builtin_amdgcn_readfirstlane(pend);
                int dbeg, cend;
                region(A, x, dbeg, cend);
                set(PEND_N, 0);
                const int start = dbeg + r;
                if (r >= 0 && start < cend) {
                    set(SEEN, r + n);
                    set(CUR, start + 1);
                    set(CUR_END, min(start + n, cend));
                    request(A, x, r + n);         // the range after this one: consumed a range later
                    return start;
                }
            }
            // this XCD's queue is empty: the next one's
            if (++tries >= FB_NXCD) { set(TRIES, tries); return -1; }
            x = (x + 1) % FB_NXCD;
            set(TRIES, tries);
            set(XCD, x);
            request(A, x, 0);
        }
    }
    // the wave is done: count it; the last one resets the slot for the launch that uses it next
    template <class QA> __device__ __forceinline__ void finish(QA qa)
    {
        const ChunkQueueArgs A = qa();
        if (A.nstat >= A.per) return;             // (the queue was not touched)
        if ((threadIdx.x & 63) == 0) {
            __threadfence();
            const unsigned done = atomicAdd(A.q + FB_NXCD * FB_QUEUE_STRIDE, 1u);
            if (done == (unsigned)(FB_NXCD * A.W - 1)) {
                for (int i = 0; i <= FB_NXCD; i++) atomicExch(A.q + i * FB_QUEUE_STRIDE, 0u);
            }
        }
    }
};
#endif

}  // namespace fb
