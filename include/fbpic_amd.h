/*
 * fbpic_amd.h -- C ABI of libfbpic_amd.so: the MI355X (gfx950) backend for FBPIC's
 * per-step PIC cycle.
 *
 * The reference (pure Python) has no FFI on this path; its backend boundary is the
 * `use_cuda` branch of every Particles / Fields / SpectralGrid / DHT / FFT method,
 * where a Numba-CUDA kernel is launched as `kernel[grid, block](*device_arrays, *scalars)`
 * (SURVEY.md 8b).  Each entry point below replaces exactly one of those launch sites
 * (cited per function, paths relative to fbpic/), keeps the reference's argument order
 * and meaning, and adds only what a C ABI needs: explicit sizes, an explicit row stride
 * for (Nz, Nr) grids, the physical constants the reference reads from scipy.constants,
 * and a trailing hipStream_t.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter is documented "host";
 *   - grids are complex128, r contiguous, `row_stride` = complex elements between
 *     consecutive z rows (== Nr for the reference's C-contiguous (Nz, Nr) arrays);
 *   - particle arrays are float64[n] (structure of arrays);
 *   - all functions are asynchronous on `stream`, return 0 on success and a non-zero
 *     hipError_t / rocfft_status otherwise (fb_last_error() gives the text); none throws.
 *   - ownership: the caller owns every buffer; the library owns only FFT plans.
 */
#ifndef FBPIC_AMD_H
#define FBPIC_AMD_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FB_SHAPE_LINEAR 1
#define FB_SHAPE_CUBIC 3
#define FB_MAX_MODES 8
/* doubles in front of the attribute rows of a particle hand-over message ([0] = particle count) */
#define FB_HANDOVER_HEADER 8

/* ---- runtime ---------------------------------------------------------------- */
/* revision of this header; fb_abi_version() returns the one the library was built from (the
 * Python binding refuses a library of another revision) */
#define FB_ABI_VERSION 7
int fb_abi_version(void);
const char *fb_last_error(void);
/* target architecture and the effective compiler options of the build (incl. whether hipcc accepted the
 * tuned scheduling options of the one-pass particle kernel); bench.py logs it next to its numbers */
const char *fb_build_info(void);
/* same text as fb_last_error (the name SURVEY.md 8b lists) */
const char *fb_last_error_string(void);
/* utils/cuda.py:261-299 (GPU selection) -> explicit device binding per process */
int fb_set_device(int device);
int fb_sync(void *stream);
/* Device arrays for a host application that has no array library of its own (the reference wraps
 * numba.cuda device arrays: utils/cuda.py:101-137 `cuda.to_device` / `copy_to_host`; fbpic_amd itself hands
 * in PyTorch-ROCm tensors and never calls these).  The caller still owns every buffer: the library frees
 * nothing on its own.  fb_h2d / fb_d2h are asynchronous on `stream` (host memory that is not page-locked
 * makes the runtime stage the copy; fb_sync(stream) before the host buffer is reused or read). */
int fb_malloc(size_t nbytes, void **device_ptr);
int fb_free(void *device_ptr);
int fb_h2d(void *device_dst, const void *host_src, size_t nbytes, void *stream);
int fb_d2h(void *host_dst, const void *device_src, size_t nbytes, void *stream);

/* ---- transport between the z-slabs of neighbouring ranks (one rank per GPU) ----------------
 * Replaces BoundaryCommunicator.exchange_domains, fbpic/boundaries/boundary_communicator.py:
 * 674-707 (mpi_comm.Isend / Irecv / Wait of the four guard-cell or particle buffers), with RCCL
 * point-to-point over xGMI INSIDE the library: a host application that binds this C ABI (CuPy +
 * mpi4py in the reference) gets the multi-GPU path without torch.distributed.
 *   fb_comm_unique_id : 128-byte id created on one rank; the host distributes it to the others
 *                       by whatever it already has (MPI bcast, a file, torch.distributed)
 *   fb_comm_init      : collective over `size` ranks (ncclCommInitRank); call after
 *                       fb_set_device
 *   fb_exchange       : one grouped launch (ncclGroupStart, <= 2 ncclSend + 2 ncclRecv,
 *                       ncclGroupEnd) on `stream`: ordered after the kernels that filled the send
 *                       buffers (e.g. fb_guard_buffers mode 0) and before those that read the
 *                       receive buffers; no host synchronisation.  left_rank / right_rank < 0:
 *                       no neighbour on that side (open boundary).  Sizes in bytes; both sides
 *                       must agree on them (send_left of a rank pairs with recv_right of its left
 *                       neighbour).  A 2-rank periodic ring (left_rank == right_rank) is handled.
 * RCCL is loaded at run time (the copy already in the process, else librccl.so.1 of ROCm). */
int fb_comm_unique_id(void *id128);
int fb_comm_init(const void *id128, int rank, int size, void **comm);
int fb_comm_destroy(void *comm);
int fb_exchange(void *comm, int left_rank, int right_rank,
                const void *send_left, size_t send_left_bytes,
                const void *send_right, size_t send_right_bytes,
                void *recv_left, size_t recv_left_bytes,
                void *recv_right, size_t recv_right_bytes, void *stream);

/* ---- particle push ---------------------------------------------------------- */
/* particles/particles.py:660-663 -> push_x_gpu (push/cuda_methods.py:16-52);
 * arithmetic order of the CPU twin push_x_numba (push/numba_methods.py:16-32). */
int fb_push_x(long n, double *x, double *y, double *z,
              const double *ux, const double *uy, const double *uz,
              const double *inv_gamma, double c, double dt,
              double x_push, double y_push, double z_push, void *stream);

/* particles/particles.py:609-613 -> push_p_gpu (push/cuda_methods.py:54-99),
 * Vay pusher push/inline_functions.py:11-48. */
int fb_push_p(long n, double *ux, double *uy, double *uz, double *inv_gamma,
              const double *Ex, const double *Ey, const double *Ez,
              const double *Bx, const double *By, const double *Bz,
              double q, double m, double c, double dt, void *stream);

/* boundaries/particle_buffer_handling.py:528-531 -> shift_particles_periodic_cuda */
int fb_shift_periodic(long n, double *z, double zmin, double zmax, void *stream);

/* ---- gather ------------------------------------------------------------------ */
/* particles/particles.py:703-800 -> gather_field_gpu_{linear,cubic}[_one_mode]
 * (gathering/cuda_methods.py:26,209; cuda_methods_one_mode.py:22,46,216).
 * One launch for any Nm.  grids (HOST array of 6*Nm device pointers): for mode m,
 * grids[6m..6m+5] = Er, Et, Ez, Br, Bt, Bz.  (Cubic shape, Nm = 2..4: the stencil sums run as
 * fp64 MFMA products per group of 16 particles; same results to ~2e-16 of max|F|.) */
int fb_gather(int shape, int Nm, long n,
              const double *x, const double *y, const double *z, double rmax_gather,
              double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
              const void *const *grids, long row_stride,
              double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
              void *stream);

/* Fusion of three consecutive launch sites of the PIC cycle (main.py:469-490):
 * gather (above) -> push_p (particles.py:609-613) -> push_x(dt_x) (particles.py:660-663)
 * in one pass over the particles: E,B stay in registers, the momenta are read and written
 * once.  Same arithmetic as the three separate entry points (bit-identical momenta and
 * positions for identical fields).  Ex..Bz may be NULL (fields not stored); dt_x = 0 skips
 * the position push.  wrap_zmax > wrap_zmin: z is first wrapped into [wrap_zmin, wrap_zmax)
 * exactly as fb_shift_periodic does (the single-domain periodic exchange_particles that
 * precedes the gather, main.py:442 / particle_buffer_handling.py:528-531). */
int fb_gather_push(int shape, int Nm, long n, double *x, double *y, double *z,
                   double *ux, double *uy, double *uz, double *inv_gamma,
                   double rmax_gather, double invdz, double zmin, int Nz,
                   double invdr, double rmin, int Nr,
                   const void *const *grids, long row_stride,
                   double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
                   double q, double m, double c, double dt, double dt_x,
                   double wrap_zmin, double wrap_zmax, void *stream);

/* fb_gather_push that also prepares the counting sort which Simulation.step runs after the NEXT
 * push_x (main.py:519-528): for every particle, the cell of the position pushed once more by
 * (dt_push, x_push, y_push, z_push) - from the momenta this call has just updated - and its
 * rank in that cell go to `sort_workspace` (layout of fb_bin_sort_workspace_bytes), exactly as
 * fb_deposit_J_rank_next does.  Follow with fb_push_x_bin_sort_particles /
 * fb_push_x_sort_deposit_rho / fb_push_x_sort_deposit_J_rho(..., preranked = 1) for that push.
 * The particle arrays must not be modified in between.  Needs dt_x != 0. */
int fb_gather_push_rank_next(int shape, int Nm, long n, double *x, double *y, double *z,
                             double *ux, double *uy, double *uz, double *inv_gamma,
                             double rmax_gather, double invdz, double zmin, int Nz, double invdr,
                             double rmin, int Nr, const void *const *grids, long row_stride,
                             double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
                             double q, double m, double c, double dt, double dt_x,
                             double wrap_zmin, double wrap_zmax,
                             double dt_push, double x_push, double y_push, double z_push, int ncell,
                             void *sort_workspace, size_t workspace_bytes, int counts_are_zero,
                             void *stream);

/* fb_gather_push_rank_next restricted to a contiguous range of the (cell-sorted) particle arrays,
 * [*range_lo, *range_hi) (range_mode 1), or to everything outside it (range_mode 2); mode 0 = all.
 * The bounds are DEVICE pointers (entries of the per-cell prefix sum; NULL = 0), so no offset is
 * read back by the host.  Lets Simulation.step (main.py:469-490 after :719-769) gather + push
 * the particles whose stencil lies in rows that the pending guard-cell exchange of E, B does not
 * touch while that exchange is in flight on another stream, then the rest.  The two calls of a
 * step share one sort workspace: pass counts_are_zero = 1 to the second.  (Not with the cubic
 * matrix-core gather: returns an error.) */
int fb_gather_push_rank_next_range(int shape, int Nm, long n, double *x, double *y, double *z,
                             double *ux, double *uy, double *uz, double *inv_gamma,
                             double rmax_gather, double invdz, double zmin, int Nz, double invdr,
                             double rmin, int Nr, const void *const *grids, long row_stride,
                             double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
                             double q, double m, double c, double dt, double dt_x,
                             double wrap_zmin, double wrap_zmax,
                             double dt_push, double x_push, double y_push, double z_push, int ncell,
                             void *sort_workspace, size_t workspace_bytes, int counts_are_zero,
                             const int *range_lo, const int *range_hi, int range_mode, void *stream);

/* ---- cell sort ---------------------------------------------------------------- */
/* particles/particles.py:1075-1081 -> get_cell_idx_per_particle
 * (utilities/cuda_sorting.py:21-88): cell_idx = ir_upper + iz_upper*(Nr+1);
 * sorted_idx[i] = i. */
int fb_cell_index(long n, const double *x, const double *y, const double *z,
                  double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
                  int *cell_idx, int *sorted_idx, void *stream);

/* particles/particles.py:1083-1092 -> sort_particles_per_cell (Thrust argsort,
 * cuda_sorting.py:90-122) + prefill_prefix_sum + incl_prefix_sum (:124-190).
 * Stable radix sort (rocPRIM) of the (cell_idx, sorted_idx) pairs, then the inclusive
 * per-cell particle count prefix_sum[ncell].  The sort ping-pongs between the primary
 * and the *_alt buffers (all int32[n]); on return *result_in_alt (HOST int) tells which
 * pair holds the sorted data, so the caller swaps its handles instead of paying a copy.
 * workspace: fb_sort_workspace_bytes() of scratch. */
size_t fb_sort_workspace_bytes(long n, int ncell);
int fb_sort_by_cell(long n, int ncell, int *cell_idx, int *sorted_idx,
                    int *cell_idx_alt, int *sorted_idx_alt, int *result_in_alt,
                    int *prefix_sum, void *workspace, size_t workspace_bytes, void *stream);

/* The whole of Particles.sort_particles (particles/particles.py:1049-1094: cell index,
 * argsort, prefix sum, rearrange) as a counting sort specialised for the almost-sorted
 * stream of the PIC cycle: cell index + per-cell rank (one atomic per run of equal cells),
 * scan of the per-cell counts, one scatter of all nattr attributes (dst[k][new] =
 * src[k][old]).  Outputs as the reference's sort: cell_idx_sorted[n], sorted_idx[n]
 * (old index of each sorted particle), prefix_sum[ncell] (inclusive).  Unlike Thrust's
 * stable argsort the order INSIDE a cell is unspecified (gather and deposition do not
 * depend on it).  src/dst: HOST arrays of nattr (<= 16) device pointers.  cell_idx_sorted
 * and sorted_idx may be NULL when the caller does not need them (8 B per particle less). */
size_t fb_bin_sort_workspace_bytes(long n, int ncell);
int fb_bin_sort_particles(long n, int ncell, const double *x, const double *y, const double *z,
                          double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
                          int nattr, const double *const *src, double *const *dst,
                          int *cell_idx_sorted, int *sorted_idx, int *prefix_sum,
                          void *workspace, size_t workspace_bytes, void *stream);

/* push_x (particles/particles.py:639-671, push/numba_methods.py:16-32) folded into the sort
 * that follows it in Simulation.step (main.py:519-528: push_x, then deposit('rho_next')
 * re-sorts): identical result to fb_push_x(n, x, y, z, ..., c, dt, x_push, y_push, z_push)
 * followed by fb_bin_sort_particles, but the pushed positions are only written once, at
 * their sorted slot.  src[0..7] must be x, y, z, ux, uy, uz, w, inv_gamma (the order of
 * the reference's particle buffers, particle_buffer_handling.py:150-160).
 * preranked != 0: the cell / rank / per-cell counts in `workspace` were already produced by
 * fb_deposit_J_rank_next for exactly this push (same dt and push factors, particle arrays
 * untouched since); the sort then only scans the counts and scatters.
 * On return (n > 0) the per-cell counters at the head of `workspace` are zero again. */
int fb_push_x_bin_sort_particles(long n, int ncell, const double *x, const double *y,
                                 const double *z, const double *ux, const double *uy,
                                 const double *uz, const double *inv_gamma, double c, double dt,
                                 double x_push, double y_push, double z_push,
                                 double invdz, double zmin, int Nz, double invdr, double rmin,
                                 int Nr, int nattr, const double *const *src, double *const *dst,
                                 int *cell_idx_sorted, int *sorted_idx, int *prefix_sum,
                                 void *workspace, size_t workspace_bytes, int preranked,
                                 void *stream);

/* particles/particles.py:519-538 -> write_sorting_buffer (cuda_sorting.py:192-213),
 * all attributes in one launch: dst[k][i] = src[k][sorted_idx[i]].
 * src, dst: HOST arrays of nattr device pointers (nattr <= 16). */
int fb_permute(long n, const int *sorted_idx, int nattr,
               const double *const *src, double *const *dst, void *stream);

/* ---- particle hand-over between the z-slabs -------------------------------------------- */
/* boundaries/particle_buffer_handling.py:17-172 (remove_outside_particles: the send buffers) and
 * :289-417 (add_buffers_to_particles), as three data movements with every attribute in one
 * launch each; indices are int64 device arrays, `arrays` a HOST array of nattr device pointers,
 * `buf` a device buffer of nattr rows of buf_row_stride doubles (the message layout of the
 * reference: one row per attribute).
 *   fb_handover_pack   : buf[k][i] = arrays[k][idx[i]], i < n      (the particles that leave)
 *   fb_handover_move   : arrays[k][dst_idx[i]] = arrays[k][src_idx[i]]  (src, dst disjoint: the
 *                        holes left in the head are filled with survivors of the tail)
 *   fb_handover_append : arrays[k][first + i] = buf[k][i], i < n   (the arrivals) */
int fb_handover_pack(long n, const long *idx, int nattr, const double *const *arrays,
                     double *buf, long buf_row_stride, void *stream);
int fb_handover_move(long n, const long *src_idx, const long *dst_idx, int nattr,
                     double *const *arrays, void *stream);
int fb_handover_append(long n, long first, int nattr, double *const *arrays,
                       const double *buf, long buf_row_stride, void *stream);

/* The same hand-over with the SELECTION in the library and every count on the device
 * (boundaries/particle_buffer_handling.py:178-236 is the reference's GPU cut; the ownership rule
 * here is the one of its CPU path, :58-172: left if z < zbox_min, right if z > zbox_max).
 *
 * fb_handover_select_pack: one launch selects the leaving particles of both sides, writes their
 *   indices (int32, at most idx_cap per side) and packs them into the two fixed-size messages
 *   `send_left` / `send_right` = FB_HANDOVER_HEADER doubles ([0] = number selected, which may
 *   exceed the capacity: then only the first `cap` are packed and the caller sends the rest in a
 *   second message) followed by nattr rows of `cap` doubles.  A NULL message = open end (the
 *   leavers are dropped).  With `prefix_sum` (cell-sorted arrays, inclusive per-cell prefix sum)
 *   only the particles before the offset prefix_sum[cut1] and after prefix_sum[cut2] are tested
 *   (everything before prefix_sum[cut0] / after prefix_sum[cut3] leaves without a test; a cut
 *   of -1 stands for offset 0); with prefix_sum = NULL every particle is tested.
 *   counts (device, 8 longs, zeroed here): [0] n_left, [1] n_right.
 * fb_handover_recv_counts: counts[2], counts[3] = header counts of the two received messages.
 * fb_handover_compact: removes the listed particles from the length-n arrays in O(n_left + n_right):
 *   the holes below n - n_left - n_right are filled with the survivors above it (any order).
 * fb_handover_append_shift: fb_handover_append with `shift` added to attribute `shift_attr`
 *   (the periodic wrap of z across the ends of the global box, :401-417); shift_attr < 0: none. */
int fb_handover_select_pack(long n, const double *z, const int *prefix_sum,
                            long cut0, long cut1, long cut2, long cut3,
                            double zbox_min, double zbox_max, int nattr,
                            const double *const *arrays, long cap_left, long cap_right, long idx_cap,
                            double *send_left, double *send_right, int *idx_left, int *idx_right,
                            long *counts, void *stream);
int fb_handover_recv_counts(const double *recv_left, const double *recv_right, long *counts,
                            void *stream);
size_t fb_handover_workspace_bytes(long max_leavers);
int fb_handover_compact(long n, long n_left, const int *idx_left, long n_right, const int *idx_right,
                        int nattr, double *const *arrays, void *workspace, size_t workspace_bytes,
                        void *stream);
int fb_handover_append_shift(long n, long first, int nattr, double *const *arrays, const double *buf,
                             long buf_row_stride, int shift_attr, double shift, void *stream);

/* ---- deposition ---------------------------------------------------------------- */
/* particles/particles.py:893-936 -> deposit_rho_gpu_{linear,cubic}[_one_mode]
 * (deposition/cuda_methods.py:27,465; cuda_methods_one_mode.py).  Particles should be
 * cell-sorted for speed (runs of equal cells are accumulated in registers); the result
 * does not depend on the order.  prefix_sum is accepted for signature compatibility with
 * the reference launch and is not read (may be NULL).
 * rho: HOST array of Nm device pointers; element (iz, ir) of each at
 * base + iz*row_stride + ir*col_stride (complex elements).  col_stride = 1 (or 0) for the
 * (Nz, Nr) grids of the reference; a record length > 1 selects a node-major target in which
 * all components and modes of a node share a cache line: global atomics cost one L2 operation
 * per line an instruction touches, so that layout makes the flush of a cell 6-12x cheaper
 * (used inside Simulation.step, where the z-FFT then reads the records directly).
 * ruyten_m0 / ruyten_mh: Ruyten coefficients (Nr+1) for mode 0 / modes >= 1.
 * nflush (device uint64[1024], may be NULL): the counters are incremented so that their
 * SUM grows by the number of runs of equal cells met in the particle stream -- n/sum is the
 * mean run length, the host's measure of how stale the cell sort has become. */
int fb_deposit_rho(int shape, int Nm, long n,
                   const double *x, const double *y, const double *z, const double *w, double q,
                   double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
                   void *const *rho, long row_stride, long col_stride,
                   const int *prefix_sum, const double *ruyten_m0, const double *ruyten_mh,
                   unsigned long long *nflush, void *stream);

/* particles/particles.py:937-985 -> deposit_J_gpu_{linear,cubic}[_one_mode]
 * (deposition/cuda_methods.py:201,750).  J: HOST array of 3*Nm device pointers,
 * J[3m..3m+2] = Jr, Jt, Jz of mode m. */
int fb_deposit_J(int shape, int Nm, long n,
                 const double *x, const double *y, const double *z, const double *w, double q,
                 const double *ux, const double *uy, const double *uz, const double *inv_gamma,
                 double c,
                 double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
                 void *const *J, long row_stride, long col_stride,
                 const int *prefix_sum, const double *ruyten_m0, const double *ruyten_mh,
                 unsigned long long *nflush, void *stream);

/* push_x + counting sort + deposit('rho') in ONE pass over the particles: the tail of
 * Simulation.step's particle work (main.py:519-528: push_x(dt/2), then deposit('rho_next'),
 * whose Particles.deposit re-sorts, particles.py:1049-1094, and launches deposit_rho_gpu_*,
 * deposition/cuda_methods.py:84,517).  Identical result to
 *   fb_push_x_bin_sort_particles(n, ncell, x, ..., preranked, stream)   followed by
 *   fb_deposit_rho(shape, Nm, n, dst[0], dst[1], dst[2], dst[6], q, ..., stream)
 * (same arithmetic per particle; the deposition sums differ by summation order only).
 * The particles are walked in DESTINATION order: sorted_idx (required, n ints; on return the
 * old index of each sorted particle, as from the sort) is built from the rank pass, every lane
 * reads its 8 attributes through it, pushes the position in registers, writes all attributes
 * contiguously at the sorted slot and deposits its charge - the stand-alone scatter (72 B read
 * + 64 B written per particle) and the re-read of x, y, z, w by the deposition (32 B) become
 * one pass of 68 B read + 64 B written, whose arithmetic overlaps its memory stalls.
 * src[0..7] = x, y, z, ux, uy, uz, w, inv_gamma; arguments otherwise as the two calls above. */
int fb_push_x_sort_deposit_rho(long n, int ncell, const double *x, const double *y,
                               const double *z, const double *ux, const double *uy,
                               const double *uz, const double *inv_gamma, double c, double dt,
                               double x_push, double y_push, double z_push,
                               double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
                               int nattr, const double *const *src, double *const *dst,
                               int *cell_idx_sorted, int *sorted_idx, int *prefix_sum,
                               void *workspace, size_t workspace_bytes, int preranked,
                               int shape, int Nm, double q, void *const *rho, long row_stride,
                               long col_stride, const double *ruyten_m0, const double *ruyten_mh,
                               void *stream);

/* The whole tail of Simulation.step's particle work in ONE pass: deposit('J') from the
 * positions at t = n+1/2 (main.py:515-517), push_x(dt/2) (:519-522), the re-sort and
 * deposit('rho_next') from the positions at t = n+1 (:528).  Identical result to
 *   fb_deposit_J(shape, Nm, n, x, ..., zmin = zmin_J, J, ...)   then
 *   fb_push_x_sort_deposit_rho(...)
 * (per-particle arithmetic unchanged; the deposited sums differ by summation order only).  The
 * particles are walked in destination order as in fb_push_x_sort_deposit_rho; every lane holds
 * x, y, z, u, w, inv_gamma in registers, deposits its current from the position it read, pushes
 * it, stores all attributes at the sorted slot and deposits its charge from the pushed position:
 * the stand-alone J pass (64 B per particle read again) disappears and the arithmetic of both
 * depositions overlaps the memory stalls of the permutation.  zmin_J: grid position for the J
 * deposit (= zmin unless the grid moved in between).  Nm <= 4.
 * engine: 0 = the library's choice - for the linear shape, both depositions on one grid geometry and
 * targets with one base and the same strides, J and rho are staged together and reduced in ONE
 * traversal of the runs (csrc/cycle_dep.h), else one after the other; 1 = always one after the
 * other (faster where many particles change cell within the push: a laser wake); same results. */
int fb_push_x_sort_deposit_J_rho(long n, int ncell, const double *x, const double *y,
                                 const double *z, const double *ux, const double *uy,
                                 const double *uz, const double *inv_gamma, double c, double dt,
                                 double x_push, double y_push, double z_push,
                                 double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
                                 int nattr, const double *const *src, double *const *dst,
                                 int *cell_idx_sorted, int *sorted_idx, int *prefix_sum,
                                 void *workspace, size_t workspace_bytes, int preranked,
                                 int shape, int Nm, double q, double zmin_J, void *const *J,
                                 long J_row_stride, long J_col_stride, void *const *rho,
                                 long row_stride, long col_stride, const double *ruyten_m0,
                                 const double *ruyten_mh, int engine, void *stream);

/* The particle work of a whole PIC step in ONE pass (csrc/cycle.hip): identical result to
 *   fb_gather_push(shape, Nm, n, x .. inv_gamma, ..., dt, dt_x, wrap_zmin, wrap_zmax)   main.py:469-490
 *   fb_deposit_J(shape, Nm, n, x, y, z, w, q, ux, uy, uz, inv_gamma, c, ..., J, ...)    main.py:515-517
 *   fb_push_x(n, x, y, z, ux, uy, uz, inv_gamma, c, dt_x, 1, 1, 1)                      main.py:519-522
 *   fb_deposit_rho(shape, Nm, n, x, y, z, w, q, ..., rho, ...)                          main.py:528
 * (momenta and positions bit-identical, deposited sums equal up to summation order): every particle
 * attribute is read once and written once, 64 B + 56 B per particle.  The arrays are NOT re-sorted
 * by this call.  `home_cell[i]` = cell (ir_upper + iz_upper (Nr + 1), the `cell_idx_sorted` output
 * of fb_bin_sort_particles) of particle i at the last sort: runs of equal home cells replace the
 * runs of equal cells of the sorted depositions, and particles that have since left their home
 * cell are gathered / deposited on their own.  Any `home_cell` content gives the same result; how
 * many particles take the slow path is what changes, and `stats` (optional, 1024 counters that
 * the call ADDS to) receives their number for the caller's re-sort policy: the sum of [0, 512) is the
 * number of such particles (J deposition), the sum of [512, 1024) the number of 64-particle chunks
 * in which more than 16 particles took it.
 * `home_cell_shift` is subtracted from every home cell before use: n_move (Nr + 1) when the grid
 * has advanced by n_move cells since the sort (moving window, boundaries/moving_window.py:60-239 -
 * the cell of a particle that stays where it is moves n_move rows down), 0 otherwise.
 * A chunk of 64 particles of which more than 12 have left their home cells (a laser wake) is regrouped inside
 * the wave - runs of its particles' CURRENT cells instead of one scatter per particle; same result.
 * Linear or cubic shape (round 6: cubic = the same pass with the 4 x 4 stencil, slower than the two passes
 * at 2048 x 512, Nm = 4 - the caller decides), Nm <= 4 (fb_gather_push_deposit_supported). */
int fb_gather_push_deposit_supported(int shape, int Nm);
int fb_gather_push_deposit_J_rho(int shape, int Nm, long n,
                                 double *x, double *y, double *z, double *ux, double *uy,
                                 double *uz, double *inv_gamma, const double *w,
                                 const int *home_cell, double rmax_gather,
                                 double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
                                 const void *const *grids, long row_stride,
                                 double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
                                 double q, double m, double c, double dt, double dt_x,
                                 double wrap_zmin, double wrap_zmax,
                                 void *const *J, long J_row_stride, long J_col_stride,
                                 void *const *rho, long rho_row_stride, long rho_col_stride,
                                 const double *ruyten_m0, const double *ruyten_mh,
                                 unsigned long long *stats, int home_cell_shift, void *stream);

/* fb_gather_push_rank_next for arrays that were cell-sorted some steps ago (csrc/cycle.hip): the same
 * results - momenta, x(n+1/2), and cell + rank of x(n+1) in `sort_workspace`, ranks of a cell in a
 * different order - with the segments of the gather taken from `home_cell` (runs of equal home cells,
 * particles that have left theirs staged on their own: see fb_gather_push_deposit_J_rho) instead of
 * from runs of equal cells, which a stale order fragments (C2, 3 steps after a sort: 184 -> ~125 us).
 * dt_x, the following push: dt_push = dt_x, x_push = y_push = z_push = 1 (the half steps of
 * Simulation.step).  Linear shape, Nm <= 4. */
int fb_gather_push_rank_next_home(int shape, int Nm, long n,
                                  double *x, double *y, double *z, double *ux, double *uy, double *uz,
                                  double *inv_gamma, const int *home_cell, double rmax_gather,
                                  double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
                                  const void *const *grids, long row_stride,
                                  double *Ex, double *Ey, double *Ez, double *Bx, double *By, double *Bz,
                                  double q, double m, double c, double dt, double dt_x,
                                  double wrap_zmin, double wrap_zmax, int ncell, void *sort_workspace,
                                  size_t workspace_bytes, int counts_are_zero, int home_cell_shift,
                                  void *stream);

/* fb_deposit_J that also prepares the counting sort which Simulation.step runs after the
 * next push_x (main.py:515-528: deposit J, push_x(dt/2), re-sort for deposit rho_next): for
 * every particle, the cell of the position pushed by (dt_push, x_push, y_push, z_push) and
 * its rank in that cell go to `sort_workspace` (layout of fb_bin_sort_workspace_bytes).
 * Follow with fb_push_x_bin_sort_particles(..., preranked = 1).  J itself is deposited
 * exactly as by fb_deposit_J.
 * counts_are_zero != 0: the caller guarantees that the per-cell counters at the head of
 * `sort_workspace` are all zero - true after any completed fb_*bin_sort_particles call on this
 * workspace (the scatter pass leaves them zeroed) - and the memset launch is skipped. */
int fb_deposit_J_rank_next(int shape, int Nm, long n, const double *x, const double *y,
                           const double *z, const double *w, double q, const double *ux,
                           const double *uy, const double *uz, const double *inv_gamma, double c,
                           double invdz, double zmin, int Nz, double invdr, double rmin, int Nr,
                           void *const *J, long row_stride, long col_stride, const double *ruyten_m0,
                           const double *ruyten_mh, unsigned long long *nflush,
                           double dt_push, double x_push, double y_push, double z_push,
                           int ncell, void *sort_workspace, size_t workspace_bytes,
                           int counts_are_zero, void *stream);

/* ---- interpolation-grid kernels ------------------------------------------------ */
/* fields/interpolation_grid.py:236-250 -> cuda_erase_scalar/vector (fields/cuda_methods.py:18,40).
 * ptrs: HOST array of nfields device pointers (nfields <= 48). */
int fb_erase(int nfields, void *const *ptrs, long row_stride, int Nz, int Nr, void *stream);
/* fields/interpolation_grid.py:286-296 -> cuda_divide_{scalar,vector}_by_volume (:68,92) */
int fb_divide_by_volume(int nfields, void *const *ptrs, long row_stride,
                        const double *invvol, int Nz, int Nr, void *stream);

/* ---- spectral-grid kernels ------------------------------------------------------ */
/* fields/spectral_grid.py:437-454 -> cuda_filter_{scalar,vector} (fields/cuda_methods.py:467,492) */
int fb_filter(int nfields, void *const *ptrs, long row_stride,
              const double *filter_z, const double *filter_r, int Nz, int Nr, void *stream);
/* fields/spectral_grid.py:225-230 -> cuda_correct_currents_curlfree_standard (:121).
 * kz, kr, inv_k2: real (Nz, Nr) contiguous, as the reference's d_kz/d_kr/d_inv_k2. */
int fb_correct_currents_curlfree_standard(const void *rho_prev, const void *rho_next,
        void *Jp, void *Jm, void *Jz, long row_stride,
        const double *kz, const double *kr, const double *inv_k2, double inv_dt,
        int Nz, int Nr, void *stream);
/* fields/spectral_grid.py:350-355 -> cuda_push_eb_standard (:235); same argument order
 * as numba_push_eb_standard (fields/numba_methods.py:118-185) + c, epsilon_0, mu_0. */
int fb_push_eb_standard(void *Ep, void *Em, void *Ez, void *Bp, void *Bm, void *Bz,
        const void *Jp, const void *Jm, const void *Jz,
        const void *rho_prev, const void *rho_next, long row_stride,
        const double *rho_prev_coef, const double *rho_next_coef, const double *j_coef,
        const double *C, const double *S_w, const double *kr, const double *kz,
        double dt, int use_true_rho, double c, double epsilon_0, double mu_0,
        int Nz, int Nr, void *stream);
/* Galilean / comoving-current PSATD (fields/spectral_grid.py:240-247, 357-368 ->
 * cuda_correct_currents_curlfree_comoving, cuda_push_eb_comoving; same argument order as
 * fields/numba_methods.py:217-241, 278-355 + c, epsilon_0, mu_0).  rho_prev_coef,
 * rho_next_coef, j_coef, T_eb, T_cc, T_rho, j_corr_coef: complex128 (Nz,Nr) contiguous
 * tables; C, S_w, kr, kz, inv_k2: float64. */
int fb_correct_currents_curlfree_comoving(const void *rho_prev, const void *rho_next,
        void *Jp, void *Jm, void *Jz, long row_stride, const double *kz, const double *kr,
        const double *inv_k2, const void *j_corr_coef, const void *T_eb, const void *T_cc,
        int Nz, int Nr, void *stream);
/* Cross-deposition current correction (fields/spectral_grid.py:231-238, 250-258 ->
 * cuda_correct_currents_crossdeposition_{standard,comoving}; same argument order as
 * fields/numba_methods.py:87-116, 243-275).  rho_next_z / rho_next_xy: spectral charge
 * densities deposited at (z[n+1], x[n]) and (z[n], x[n+1]) by Simulation.cross_deposit
 * (main.py:672-716); same row stride as the other spectral fields. */
int fb_correct_currents_crossdeposition_standard(const void *rho_prev, const void *rho_next,
        const void *rho_next_z, const void *rho_next_xy, void *Jp, void *Jm, void *Jz,
        long row_stride, const double *kz, const double *kr, double inv_dt,
        int Nz, int Nr, void *stream);
int fb_correct_currents_crossdeposition_comoving(const void *rho_prev, const void *rho_next,
        const void *rho_next_z, const void *rho_next_xy, void *Jp, void *Jm, void *Jz,
        long row_stride, const double *kz, const double *kr, const void *j_corr_coef,
        const void *T_eb, const void *T_cc, int Nz, int Nr, void *stream);
int fb_push_eb_comoving(void *Ep, void *Em, void *Ez, void *Bp, void *Bm, void *Bz,
        const void *Jp, const void *Jm, const void *Jz,
        const void *rho_prev, const void *rho_next, long row_stride,
        const void *rho_prev_coef, const void *rho_next_coef, const void *j_coef,
        const double *C, const double *S_w, const void *T_eb, const void *T_cc,
        const void *T_rho, const double *kr, const double *kz, double dt, double V,
        int use_true_rho, double c, double epsilon_0, double mu_0, int Nz, int Nr, void *stream);
/* The r-spectral part of a step of the standard PSATD scheme on a single z-periodic domain in ONE
 * launch (csrc/spectral_cycle.hip): identical result to
 *   fb_hankel_rt_to_pm_scaled(4 Nm jobs: J (r,t,z) and rho_next of every mode, divide-by-volume,
 *                             spectral filter)                                   fields.py:313-368
 *   fb_psatd_step_standard(Nm, fields, ..., correct_currents, use_true_rho)      main.py:530-557
 *   fb_hankel(6 Nm jobs: E, B of every mode, inverse matrices)                   fields.py:370-429
 * A workgroup owns 8 kz rows of one mode: J and rho_next never leave its accumulators between the
 * forward products and the cell-local update, the new E, B go to the inverse products through LDS.
 *   src[4m ..]   : Jr, Jt, Jz, rho of mode m after the forward z-FFT (un-normalised), row stride
 *                  src_row_stride; invvol[m]: 1 / cell volume (Nr)
 *   fwd_mats / inv_mats[3m ..] : Hankel matrices (Nr x Nr, row-major [k][n]) of the p, m and z / scalar
 *                  components of mode m (orders m+1, m-1, m) and their inverses
 *   filter_z / filter_r[m] : spectral filter (both NULL: none)
 *   fields[11m ..], tables[8m ..] : as fb_psatd_step_standard
 *   out[6m ..]   : E (p, m, z), B (p, m, z) of mode m in (kz, r) space: the input of fb_zfft_pm_to_rt
 * Must not alias: src and out (different workgroups read / write the same rows of different fields).
 * Nr <= 128 (fb_spect_cycle_supported); correct_currents 0 / 1, or
 * correct_currents = 2: the launch ends behind the curl-free correction - identical to
 *   fb_hankel_rt_to_pm_scaled(...) + fb_psatd_step_standard(..., correct_currents = 2, ...):
 * the corrected J and rho_next are stored, E, B, rho_prev are not touched, inv_mats / out may be
 * NULL.  For z-decomposed runs, whose J guard cells are added between the correction and the push
 * (main.py:530-542). */
int fb_spect_cycle_supported(int Nm, int Nr);
int fb_spect_cycle_standard(int Nm, const void *const *src, long src_row_stride,
                            const double *const *invvol, const double *const *fwd_mats,
                            const double *const *inv_mats, const double *const *filter_z,
                            const double *const *filter_r, void *const *fields,
                            long spect_row_stride, const double *const *tables, double dt,
                            int correct_currents, int use_true_rho, double c, double epsilon_0,
                            double mu_0, void *const *out, long out_row_stride, int Nz, int Nr,
                            void *stream);

/* Single-launch fusion of the three cell-local spectral updates for ALL modes:
 * fields/spectral_grid.py:225-230 (curl-free correction, optional) -> :350-355 (PSATD push)
 * -> :416-417 (rho shift).  fields: HOST array of 11*Nm device pointers, per mode in
 * SpectralGrid order Ep,Em,Ez,Bp,Bm,Bz,Jp,Jm,Jz,rho_prev,rho_next; tables: HOST array of
 * 8*Nm device pointers, per mode rho_prev_coef,rho_next_coef,j_coef,C,S_w,kr,kz,inv_k2
 * (each real (Nz,Nr) contiguous).  Same results as the three separate entry points.
 * correct_currents: 0 = push + rho shift only, 1 = correction + push + rho shift,
 * 2 = correction only (decomposed domains: the J guard exchange of main.py:537 sits between
 * the correction and the push, so the step makes one call with 2 and one with 0). */
int fb_psatd_step_standard(int Nm, void *const *fields, long row_stride,
        const double *const *tables, double dt, int correct_currents, int use_true_rho,
        double c, double epsilon_0, double mu_0, int Nz, int Nr, void *stream);
/* fb_psatd_step_standard followed by fb_shift_spect(E, B, rho_prev, J; shift, n_move) in the
 * same launch: the translation of the moving window (boundaries/moving_window.py:176-239, which
 * Simulation.step applies right after the field push, main.py:546-549) multiplies the values
 * the push writes anyway.  shift: complex128[Nz] = exp(i kz_true dz) (SpectralGrid.field_shift);
 * n_move = 0 or shift = NULL: identical to fb_psatd_step_standard.  Not with correct_currents = 2. */
int fb_psatd_step_standard_shift(int Nm, void *const *fields, long row_stride,
                                 const double *const *tables, double dt, int correct_currents,
                                 int use_true_rho, double c, double epsilon_0, double mu_0, int Nz,
                                 int Nr, const void *shift, int n_move, void *stream);
/* fields/spectral_grid.py:416-417 -> cuda_push_rho (:443) */
int fb_push_rho(void *rho_prev, void *rho_next, long row_stride, int Nz, int Nr, void *stream);
/* fields/spectral_transform/spectral_transformer.py:140-142, 208-210 ->
 * cuda_pm_to_rt / cuda_rt_to_pm (spectral_transform/cuda_methods.py:119,140).
 * In-place allowed (p aliases r, m aliases t), as in the reference's shared buffers. */
int fb_rt_to_pm(int npairs, const void *const *r, const void *const *t,
                void *const *p, void *const *m, long row_stride, int Nz, int Nr, void *stream);
int fb_pm_to_rt(int npairs, const void *const *p, const void *const *m,
                void *const *r, void *const *t, long row_stride, int Nz, int Nr, void *stream);
/* cupy.multiply(..., 1/Nz) of fourier.py:157 for the FFT-only (partial) transforms */
int fb_scale(int nfields, void *const *ptrs, long row_stride, double factor,
             int Nz, int Nr, void *stream);

/* boundaries/moving_window.py:243-278 -> shift_spect_array_gpu: translate spectral fields
 * by n_move cells, F[iz, :] *= shift[iz]^n_move with shift = exp(i kz_true dz) (complex128[Nz]). */
int fb_shift_spect(int nfields, void *const *ptrs, long row_stride, const void *shift,
                   int n_move, int Nz, int Nr, void *stream);

/* ---- guard cells of the z-domain decomposition ------------------------------------- */
/* boundaries/field_buffer_handling.py:114-470 -> boundaries/cuda_methods.py:12-195
 * (copy_{vec,scal}_to_gpu_buffer), :197-372 (replace_*_from_gpu_buffer), :374-484
 * (add_*_from_gpu_buffer), all modes and components and both z ends in ONE launch.
 * slab: first element of the field group in row 0 of a z-major slab (row_stride complex
 * elements between z rows; `ncontig` = n_fields * Nr adjacent complex values per row).
 * Rows [z_left, z_left + nrows) <-> buf_left, rows [z_right, z_right + nrows) <-> buf_right
 * (contiguous complex128[nrows * ncontig]; NULL = no neighbour on that side).
 * mode 0: slab -> buffers; 1: buffers replace the slab rows; 2: buffers are added. */
int fb_guard_buffers(int mode, void *slab, long row_stride, long ncontig, int z_left,
                     int z_right, int nrows, void *buf_left, void *buf_right, void *stream);
/* boundaries/boundary_communicator.py:828-907 -> cuda_damp_EB_left / cuda_damp_EB_right
 * (boundaries/cuda_methods.py:486-640): rows [0, nd_left) *= damp_left[iz], rows
 * [Nz - nd_right, Nz) *= damp_right[iz - (Nz - nd_right)] (NULL = open end elsewhere). */
int fb_damp_rows(void *slab, long row_stride, long ncontig, const double *damp_left, int nd_left,
                 const double *damp_right, int nd_right, int Nz, void *stream);

/* ---- FFT along z (rocFFT) --------------------------------------------------------- */
/* fields/spectral_transform/fourier.py:78 (cufft Plan1d) and :116-160.
 * One plan transforms `ncols` columns at once: element (iz, col) lives at
 * base + iz*stride + col (stride in complex elements), i.e. a (Nz, ncols) strided view
 * covering one or several side-by-side (Nz, Nr) grids.  No transpose copies.
 * direction: -1 forward (unnormalised), +1 backward (includes the 1/Nz of
 * fourier.py:157 through the rocFFT plan's scale factor).
 * Plan creation fails (rocfft_status 1) for some lengths, e.g. Nz = 4416 = 2^6*3*23 (in any
 * layout): the host then uses fb_fft_generic below. */
int fb_fft_plan_create(int Nz, long ncols, long in_stride, long out_stride, int inplace,
                       void **plan_fwd_bwd);
int fb_fft_exec(void *plan, int direction, const void *in, void *out, void *stream);
int fb_fft_plan_destroy(void *plan);

/* Hand-written z-FFT for Nz = 2^k in [64, 4096] and 9 * 2^k in [576, 2304] (csrc/zfft.hip): same transform and
 * conventions as fb_fft_exec (fourier.py:104-168; forward unnormalised, backward x 1/Nz) on
 * the strided (Nz, ncols) view, without a plan object; in == out allowed when the strides
 * are equal.  fb_zfft_supported(Nz) tells the host which path to take; other lengths go
 * through the rocFFT plans above. */
int fb_zfft_supported(int Nz);
int fb_zfft(int Nz, long ncols, const void *in, long in_stride, void *out, long out_stride,
            int direction, void *stream);

/* Backward fb_zfft with the (p, m) -> (r, t) combination of spectral_transformer.py:140-142
 * (numba_pm_to_rt) folded into its first pass: the ncols columns are groups of (p, m, z)
 * fields of Nr columns each; an r column reads p + m, a t column i (p - m), a z column
 * itself.  Same arithmetic as fb_pm_to_rt followed by fb_zfft(..., +1), one sweep less.
 * Out of place only. */
int fb_zfft_pm_to_rt(int Nz, long ncols, const void *in, long in_stride, void *out,
                     long out_stride, int Nr, void *stream);

/* Forward fb_zfft whose input is the node-major record array the in-step deposition fills
 * (fb_deposit_* with col_stride = record): in[iz*in_stride + ir*record + f], f < nfields, is
 * gathered as column (f, ir); the output is the usual z-major slab view of nfields*Nr
 * columns.  Out of place only; lengths as fb_zfft. */
int fb_zfft_from_records(int Nz, int nfields, int Nr, const void *in, long in_stride, int record,
                         void *out, long out_stride, void *stream);
/* fb_zfft_from_records that also zeroes the record array it has read (nfields == record: every
 * element is read by exactly one lane): the array is then ready for the next step's deposition
 * and the erase launch before it (fields/fields.py erase -> interpolation_grid.py:236-263)
 * disappears. */
int fb_zfft_from_records_consume(int Nz, int nfields, int Nr, void *in, long in_stride, int record,
                                 void *out, long out_stride, void *stream);

/* The same for the lengths the single-launch LDS transform does not hold but fb_fft_generic
 * takes in two sweeps (Nz = 192 x R with a single-pass R, e.g. 4416 = 192 x 23: the local
 * grid of the 4096-cell laser-wakefield window): the head of the transform gathers the records
 * and zeroes them.  `scratch`: a slab of Nz rows x >= nfields*Nr columns distinct from in / out.
 * fb_fft_generic_from_records_supported(Nz) tells whether a length takes this path. */
int fb_fft_generic_from_records_supported(int Nz);
int fb_fft_generic_from_records_consume(int Nz, int nfields, int Nr, void *in, long in_stride,
                                        int record, void *out, long out_stride, void *scratch,
                                        long scratch_stride, void *stream);

/* Self-contained fallback for every other length whose prime factors are <= 31 (rocFFT of
 * ROCm 7.2 refuses some, e.g. Nz = 4416): one Stockham pass per launch through global
 * memory, ping-pong between `out` and a caller-provided scratch slab of the same shape
 * (scratch_stride in complex elements); in == out allowed.  Lengths 192 x R with a single-pass
 * R (4416 = 192 x 23) take two sweeps: the 192-point factor in one LDS launch, then radix R. */
int fb_fft_generic_supported(int Nz);
int fb_fft_generic(int Nz, long ncols, const void *in, long in_stride, void *out,
                   long out_stride, void *scratch, long scratch_stride, int direction,
                   void *stream);

/* ---- Hankel transform along r (fp64 MFMA GEMM) ------------------------------------ */
/* fields/spectral_transform/hankel.py:196-205, 227-236 (copy_2dC_to_2dR + cublas dgemm
 * + copy_2dR_to_2dC).  For each job j < njobs:
 *     out_j[iz, :] = alpha * ( in_j[iz, :] . mat_j )        (complex row times real matrix)
 * in/out/mat: HOST arrays of njobs device pointers; mat_j is (Nr, Nr) float64 row-major
 * (the reference's M / invM).  The complex<->real split of the reference is fused into
 * the operand loads/stores. */
int fb_hankel(int njobs, const void *const *in, long in_row_stride,
              void *const *out, long out_row_stride, const double *const *mat,
              double alpha, int Nz, int Nr, void *stream);
/* Same transform with three optional per-job real scalings fused in (each table is a HOST
 * array of njobs device pointers, or NULL; individual entries may be NULL):
 *   in_col_scale[j][k]  : input column k is multiplied first -- the divide-by-volume of
 *                         fields/interpolation_grid.py:286-296, which commutes with the z-FFT;
 *   out_row_scale[j][iz], out_col_scale[j][n] : the spectral filter fz[iz]*fr[n] of
 *                         fields/spectral_grid.py:437-454 applied to the result. */
int fb_hankel_scaled(int njobs, const void *const *in, long in_row_stride,
                     void *const *out, long out_row_stride, const double *const *mat,
                     const double *const *in_col_scale, const double *const *out_row_scale,
                     const double *const *out_col_scale, double alpha, int Nz, int Nr,
                     void *stream);

/* fb_hankel_scaled with the (r, t) -> (p | m) combination of spectral_transformer.py:208-210
 * (numba_rt_to_pm) fused into the operand load: the input of job j is
 * 0.5 * (in[j] + pair_sign[j] * i * in2[j]) when in2[j] is not NULL (p: in = r, in2 = t,
 * sign -1; m: in = r, in2 = t, sign +1), in[j] itself otherwise (z components, scalars).
 * Same arithmetic as fb_rt_to_pm followed by fb_hankel_scaled, one sweep less. */
int fb_hankel_rt_to_pm_scaled(int njobs, const void *const *in, const void *const *in2,
                              const double *pair_sign, long in_row_stride,
                              void *const *out, long out_row_stride,
                              const double *const *mat, const double *const *in_col_scale,
                              const double *const *out_row_scale,
                              const double *const *out_col_scale, double alpha, int Nz, int Nr,
                              void *stream);
/* Backward counterpart: the Hankel transform of a vector field with (p, m) -> (r, t)
 * (spectral_transformer.py:89-155 numba_pm_to_rt) folded into the GEMM.  A job with
 * in2[j] != NULL transforms a (p, m) pair, p' = in . mat and m' = in2 . mat2, and writes
 *   out = p' + m'  (r)   and   out2 = i (p' - m')  (t);
 * jobs with in2[j] == NULL are plain transforms (z components, out2 unused).  Same MFMA
 * work as the plain transforms.  Used where the input is already in z-real space (after
 * the guard-cell exchange, main.py:741-766), so that no further FFT is needed to reach the
 * interpolation grid. */
int fb_hankel_pm_to_rt(int njobs, const void *const *in, const void *const *in2,
                       long in_row_stride, void *const *out, void *const *out2,
                       long out_row_stride, const double *const *mat, const double *const *mat2,
                       double alpha, int Nz, int Nr, void *stream);

#ifdef __cplusplus
}
#endif
#endif
