/* htf_standin.h -- stand-in for the parts of HOOMD-blue that sit either side of the
 * force path (integrator, cell-list neighbor search) so that "MD steps/s" can be
 * measured without HOOMD.  NOT part of the drop-in boundary (that is htf_amd.h): a real
 * deployment keeps HOOMD's own integrator and NeighborList and never links this.
 * Arrays use HOOMD layouts (Scalar4 pos/vel/force; n_neigh/head_list/nlist, FULL mode).
 */
#ifndef HTF_STANDIN_H_
#define HTF_STANDIN_H_
#include "htf_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Leapfrog form of velocity Verlet (unit mass): v += dt*f; x += dt*v; wrap into the
 * periodic box; pos.w (type bits) untouched.  IntegratorTwoStep + TwoStepNVE analogue. */
HTF_API int htfs_nve_step(void *d_pos, void *d_vel, const void *d_force, int dtype, unsigned N,
                          double dt, const htf_box *box, htf_stream stream);

/* *d_out (float, caller zeroes) = max_i |minimage(pos_i - ref_i)|^2: the neighbor list
 * must be rebuilt once this exceeds (r_buff/2)^2 (NeighborList::distanceCheck). */
HTF_API int htfs_max_displacement2(const void *d_pos, const void *d_ref, int dtype, unsigned N,
                                   const htf_box *box, float *d_out, htf_stream stream);

/* The same check for a replayed (hipGraph) cycle in ONE launch: d_work (2 words, zero before the first call and left zero by every
 * call) accumulates, the last workgroup publishes d_out[0] = the largest squared displacement and d_out[1] += 1 (the cycle number:
 * an UNSIGNED 32-bit word stored in the float slot's bits, compared modulo 2^32 by the host -- a float value stopped at 2^24).  h_out (nullable; PINNED host memory, device-accessible) receives the same two words, the cycle number last behind
 * a system-scope fence; `mirror` (nullable) lists up to HTFS_MIRROR_MAX word ranges the same workgroup copies from device to
 * pinned host memory BEFORE that -- status words earlier kernels left behind (htfs_brick counts, the list's largest row) -- so a
 * host that reads cycle c in h_out reads everything up to the end of cycle c - 1 beside it, without one copy node in the chain. */
#define HTFS_MIRROR_MAX 4
typedef struct htfs_mirror {
    const void *src[HTFS_MIRROR_MAX];
    void *dst[HTFS_MIRROR_MAX];
    unsigned words[HTFS_MIRROR_MAX];
    unsigned n;
} htfs_mirror;
HTF_API int htfs_check_displacement2(const void *d_pos, const void *d_ref, int dtype, unsigned N, const htf_box *box,
                                     unsigned *d_work, float *d_out, float *h_out, const htfs_mirror *mirror, htf_stream stream);

/* Cell-list neighbor search in HOOMD layout (NeighborListGPUBinned analogue).
 * d_pos_sorted [Ntot] (htfs_gather4_tagged: pos[order] with the particle's index in w, so a cell's
 * members are contiguous and one load brings a candidate's position and identity) and d_cell_start
 * [ncell+1] are produced by the caller (binning + sort + gather are plumbing); this kernel walks
 * the neighbor cells of each local particle -- stencil3[d] = 0 (one cell along d), 1 (cells at
 * least r_list wide, 3 per direction) or 2 (at least r_list / 2 wide, 5 per direction) -- and writes
 *   nlist[i*pitch + c] = k  for every k != i with |minimage(r_k - r_i)| <= r_list,
 *   n_neigh[i] = count, head_list[i] = i*pitch.
 * *d_max_neigh is set to the largest count (> pitch means the list overflowed and must be
 * rebuilt with a larger pitch; a call the gate below holds back leaves the previous value).  type_split >= 0: pairs whose types lie on different
 * sides of it are left out (hoomd.md.nlist.rcut set_pair(..., -1) between all-atom and mapped
 * bead types, tensorflowcompute.py:284-305); -1: no type filter.
 * d_ranges: the caller's candidate-range table, 16-byte aligned, 4 * ncell * (2 stencil3[1] + 1) * (2 stencil3[2] + 1) words
 * (HTFS_RANGE_WORDS).  Every call rewrites it before the search reads it; it belongs to the list (ABI 3: until then a
 * thread-local buffer of the library, re-allocated under any hipGraph that had captured its address). */
#define HTFS_RANGE_WORDS(ncell, stencil_y, stencil_z) (4u * (size_t)(ncell) * (2u * (stencil_y) + 1u) * (2u * (stencil_z) + 1u))
HTF_API int htfs_build_nlist(const void *d_pos, const void *d_pos_sorted, int dtype, unsigned N, unsigned Ntot,
                             const htf_box *box, double r_list, const int *ncell3, const int *stencil3,
                             const unsigned *d_cell_start, unsigned pitch, int type_split,
                             unsigned *d_n_neigh, unsigned *d_head_list, unsigned *d_nlist, unsigned *d_max_neigh,
                             void *d_ranges, htf_stream stream);

/* dest[i] = src[order[i]] for Scalar4 arrays: the cell-sorted position copy */
/* Cell binning: d_order <- particle indices sorted by cell (ascending index inside a cell: deterministic),
 * d_cell_start[c] <- first slot of cell c (ncell + 1 entries).  d_scratch: 2 * ncell words, 16-byte aligned (checked).  Its
 * first half (the per-cell counts) is left zeroed by every call that runs; a call under htfs_set_gate clears it again only
 * when the calling thread's last call on this scratch used another ncell (or there was none). */
HTF_API int htfs_cell_sort(const unsigned *d_cell_of, unsigned Ntot, unsigned ncell, unsigned *d_scratch,
                           unsigned *d_cell_start, unsigned *d_order, htf_stream stream);
HTF_API int htfs_gather4(void *d_dest, const void *d_src, const int *d_order, int dtype, unsigned n, htf_stream stream);
/* the same with w replaced by  order[i] | (type >= type_split) << 31  (type_split < 0: the index alone; n <= 2^31):
 * the candidate array htfs_build_nlist reads, called with the same type_split */
HTF_API int htfs_gather4_tagged(void *d_dest, const void *d_src, const int *d_order, int dtype, unsigned n, int type_split,
                                htf_stream stream);

/* the same for arrays with INERT rows (x = NaN: fixed-capacity arrays of a decomposed system, hoomd_tf_amd/brick.py -- such rows
 * are in no cell, htfs_cell_index gives them 0xFFFFFFFF and htfs_cell_sort leaves them out): only the first *d_n_live entries of
 * d_order exist, d_n_live = d_cell_start + ncell (the binned total), read on the device */
HTF_API int htfs_gather4_tagged_live(void *d_dest, const void *d_src, const int *d_order, int dtype, unsigned n_max,
                                     const unsigned *d_n_live, int type_split, htf_stream stream);

/* cell index of every particle (x fastest): d_cell_of[i] */
HTF_API int htfs_cell_index(const void *d_pos, int dtype, unsigned Ntot, const htf_box *box,
                            const int *ncell3, unsigned *d_cell_of, htf_stream stream);

/* Conditional rebuild WITHOUT a host decision: after htfs_set_gate(d_disp2, threshold2) every binning /
 * search kernel of this header launched by the calling thread (htfs_cell_index, htfs_cell_sort, htfs_gather4[_tagged],
 * htfs_build_nlist, htfs_commit_rebuild) returns at entry unless *d_disp2 > threshold2 when it RUNS, with
 * d_disp2 the word htfs_max_displacement2 has just filled on the same stream; htfs_set_gate(NULL, 0) ends
 * it.  The caller enqueues the whole rebuild behind every distance check and never reads the result back. */
HTF_API int htfs_set_gate(const float *d_disp2, double threshold2);

/* Tail of a rebuild (gated like the rest): ref[i] = pos[i] for i < N, and *d_counter (nullable) += 1. */
HTF_API int htfs_commit_rebuild(void *d_ref, const void *d_pos, int dtype, unsigned N, unsigned *d_counter,
                                htf_stream stream);

/* htfs_cell_index + htfs_cell_sort + htfs_gather4_tagged + htfs_build_nlist + htfs_commit_rebuild of a single-domain system
 * (N particles, no ghosts) on the same arguments, in six launches instead of nine; gated like them.  d_scratch: as
 * htfs_cell_sort's; d_ranges: as htfs_build_nlist's; d_ref / d_counter nullable. */
HTF_API int htfs_rebuild_nlist(const void *d_pos, int dtype, unsigned N, const htf_box *box, double r_list, const int *ncell3,
                               const int *stencil3, unsigned *d_cell_of, unsigned *d_scratch, unsigned *d_cell_start,
                               unsigned *d_order, void *d_pos_sorted, unsigned pitch, int type_split, unsigned *d_n_neigh,
                               unsigned *d_head_list, unsigned *d_nlist, unsigned *d_max_neigh, void *d_ref,
                               unsigned *d_counter, void *d_ranges, htf_stream stream);

/* the same with ghosts (and inert rows): Ntot >= N positions are binned, the N local rows searched and committed.
 * scratch_clean != 0: the caller guarantees that the first ncell words of d_scratch are zero -- they are after any COMPLETED
 * htfs_cell_sort / htfs_rebuild_nlist* call on this scratch with the same ncell, as long as nothing else has written the buffer --
 * and the call skips its memset (two dependent nodes of a captured rebuild).
 * image_L (nullable, 3 doubles on the host; 0 = leave the axis alone): along an axis the grid is NOT periodic on, the period the
 * caller's coordinates have -- the logical box length of a decomposed system -- so that a coordinate is binned, and copied into
 * d_pos_sorted, as its image nearest the grid's centre (a row that left the brick through a face on the box boundary was wrapped
 * to the far side by the integrator; the search measures plain differences along such an axis). */
HTF_API int htfs_rebuild_nlist_ghosts(const void *d_pos, int dtype, unsigned N, unsigned Ntot, const htf_box *box, double r_list,
                                      const int *ncell3, const int *stencil3, unsigned *d_cell_of, unsigned *d_scratch,
                                      unsigned *d_cell_start, unsigned *d_order, void *d_pos_sorted, unsigned pitch, int type_split,
                                      unsigned *d_n_neigh, unsigned *d_head_list, unsigned *d_nlist, unsigned *d_max_neigh, void *d_ref,
                                      unsigned *d_counter, void *d_ranges, int scratch_clean, const double *image_L, htf_stream stream);

/* One check step of a device-decided list in one call: *d_disp2 <- 0, htfs_max_displacement2 into it, htfs_set_gate(d_disp2,
 * threshold2), htfs_rebuild_nlist (d_stat2[0] = largest row, d_stat2[1] = rebuild counter), htfs_set_gate(NULL, 0), and -- if
 * h_stat2 (pinned host memory) is given -- an asynchronous copy of the two status words for a later check to read. */
HTF_API int htfs_check_rebuild_nlist(const void *d_pos, int dtype, unsigned N, const htf_box *box, double r_list, const int *ncell3,
                                     const int *stencil3, unsigned *d_cell_of, unsigned *d_scratch, unsigned *d_cell_start,
                                     unsigned *d_order, void *d_pos_sorted, unsigned pitch, int type_split, unsigned *d_n_neigh,
                                     unsigned *d_head_list, unsigned *d_nlist, unsigned *d_stat2, void *d_ref, float *d_disp2,
                                     double threshold2, unsigned *h_stat2, void *d_ranges, htf_stream stream);

/* Slab decomposition (the stand-in for HOOMD's Communicator; hoomd_tf_amd/domain.py): the migration + ghost plan of a
 * rebuild.  d_key[i] = destination * 4 + ghost class of local particle i (destination: 0 stay, 1 left neighbor,
 * 2 right neighbor, 3 beyond; class in the slab it ends up in: 0 interior, 1 near the left face only, 2 near both,
 * 3 near the right face only).  d_bounds: world + 1 slab boundaries along x in the positions' dtype.  htfs_key_sort16 over
 * these keys then yields the stable (destination, class) order and the 16 counts. */
HTF_API int htfs_slab_classify(const void *d_pos, int dtype, unsigned N, const void *d_bounds, int world, int rank,
                               double r_ghost, unsigned *d_key, htf_stream stream);

/* Stable counting sort of N elements by a key < 16: d_order <- indices grouped by key, ascending index inside a key;
 * d_start[k] <- first slot of key k (17 entries).  d_scratch: 16 * ceil(N / 4096) words.  (htfs_cell_sort is the tool for
 * many cells of a few members; this one for a few keys of many members.) */
HTF_API int htfs_key_sort16(const unsigned *d_key, unsigned N, unsigned *d_scratch, unsigned *d_start, unsigned *d_order,
                            htf_stream stream);

/* dst[dst_start[s] + j] = src[src_start[s] + j] for j < count[s], s < n_segments <= HTFS_MAX_SEGMENTS, rows of
 * row_bytes (a multiple of 4): merges class-sorted segments into one class-sorted array in one launch. */
#define HTFS_MAX_SEGMENTS 16
HTF_API int htfs_segment_copy(void *d_dst, const void *d_src, unsigned row_bytes, unsigned n_segments,
                              const unsigned *src_start, const unsigned *dst_start, const unsigned *count, htf_stream stream);

/* ---- Brick decomposition with fixed-capacity arrays (round 5; hoomd_tf_amd/brick.py BrickDomain) --------------------------------
 * The stand-in for HOOMD's Communicator on a px x py (x 1) grid of ranks, built so that NOTHING a rebuild produces has to be read by
 * the host: local rows live in [0, cap_int) (particles with no ghost neighbor, then INERT rows) and [cap_int, cap_int + cap_bnd)
 * (particles within r_ghost of a face, ordered by class, then inert rows); ghosts of message m live in a fixed region of
 * ghost_cap[m] rows behind them.  An inert row has x = NaN (never a neighbor, in no cell, empty neighbor row, zero velocity), so
 * every launch is sized by a capacity and particle counts, message counts and class boundaries stay on the device (d_counts).
 * Messages are numbered by neighbor offset o in {-1, 0, +1}^ndim: index = sum_d (o_d + 1) 3^d with the centre skipped.
 * Class of a particle along a decomposed axis: 0 away from both faces, 1 within r_ghost of the low face only, 2 of both, 3 of the
 * high face only; class key = k_0 + 4 k_1; interior = key 0.  Message to offset o carries the rows whose class is near the low
 * (o_d = -1) / high (o_d = +1) face along every axis with o_d != 0, in row order.
 * Replica mode: one rank that is its own neighbor in every direction (a p-fold periodic replication of its brick): messages are
 * shifted by -o_d * width_d (shift[m]), so ghosts and returning migrants appear where a real neighbor's would. */
#define HTFS_BRICK_MAX_MSG 8
#define HTFS_BRICK_MAX_P 64
typedef struct htfs_brick {
    int ndim;                                   /* decomposed axes: 1 or 2 */
    int axis[2];                                /* box axis of each (0 x, 1 y, 2 z) */
    int p[2];                                   /* bricks along it */
    int me[2];                                  /* this rank's brick coordinate */
    int n_msg;                                  /* 3^ndim - 1 */
    int replica;                                /* this rank is its own neighbor in every direction (messages are shifted) */
    double r_ghost;
    unsigned cap_int, cap_bnd;
    unsigned ghost_cap[HTFS_BRICK_MAX_MSG];     /* rows of halo message m (sent AND received: capacities are symmetric) */
    unsigned ghost_off[HTFS_BRICK_MAX_MSG];     /* first row of message m in the packed send buffer; a message RECEIVED from the neighbor
                                                 * at offset o lands at ghost row ghost_off[index(o)] (source offsets ascending) */
    unsigned mig_cap[HTFS_BRICK_MAX_MSG];       /* rows of migration message m, its header row included */
    unsigned mig_off[HTFS_BRICK_MAX_MSG];       /* first row of message m in the migration send buffer / of the message from offset o
                                                 * in the receive buffer */
    double shift[HTFS_BRICK_MAX_MSG][3];        /* added to the positions HALO message m carries: replica mode -o_d * brick width; with a
                                                 * brick-local cell grid also the box vector of a message that crosses the periodic
                                                 * boundary (ghosts then sit next to the brick, as HOOMD wraps its ghosts) */
    double mig_shift[HTFS_BRICK_MAX_MSG][3];    /* the same for MIGRATION message m (replica mode only: a real migrant has already been
                                                 * wrapped into its new owner's brick by the integrator) */
    int halo_wrap, mig_wrap;                    /* wrap a shifted position back into the global box (halo: only when the list is binned
                                                 * on the global box's grid; migration: replica mode) */
    double box_lo[3], box_L[3];                 /* the global (in replica mode: logical) periodic box: a shifted position is wrapped
                                                 * back into it, as the integrator wraps (ghosts and migrants keep coordinates the
                                                 * cell list can bin; the pair-vector build takes the minimum image anyway) */
} htfs_brick;

/* device words of d_counts */
enum {
    HTFS_BC_N_INT = 0,      /* particles in the interior segment */
    HTFS_BC_N_BND = 1,      /* particles in the boundary segment */
    HTFS_BC_N_CAND = 2,     /* stayed + arrived at the last rebuild */
    HTFS_BC_N_ARRIVED = 3,  /* migrants received, cumulative */
    HTFS_BC_FLAGS = 4,      /* HTFS_BF_* bits, sticky until the caller clears them */
    HTFS_BC_REBUILDS = 5,
    HTFS_BC_MSG = 8,        /* [8, 16): rows of halo message m */
    HTFS_BC_CLASS = 16,     /* [16, 16 + 18): first candidate of class c in class order (c = 0..16), then the total */
    HTFS_BC_SLOT = 64,      /* [64, 64 + 16 * 8): first slot of class c in halo message m (word 64 + 8 c + m), 0xFFFFFFFF: not carried */
    HTFS_BC_WORDS = 192
};
enum {
    HTFS_BF_LOST = 1,           /* a particle crossed more than one brick between rebuilds */
    HTFS_BF_MIG_OVERFLOW = 2,   /* more migrants than a migration message holds */
    HTFS_BF_INT_OVERFLOW = 4,
    HTFS_BF_BND_OVERFLOW = 8,
    HTFS_BF_GHOST_OVERFLOW = 16,
    HTFS_BF_HALO_TIMEOUT = 32   /* transport "peer": a neighbor's message did not arrive within the spin limit */
};

/* scratch of a rebuild, all caller-owned: cand = cap_int + cap_bnd + sum_m mig_cap[m] candidate slots */
typedef struct htfs_brick_work {
    unsigned *key;          /* [cand] */
    unsigned *order;        /* [cand] */
    unsigned *sort_scratch; /* [32 * ceil(cand / 1024)] */
    unsigned *start1;       /* [17]  destination-key starts of the first sort */
    unsigned *start2;       /* [33]  class-key starts of the second */
    void *tmp_pos;          /* [cand] Scalar4 */
    void *tmp_vel;          /* [cand] Scalar4 */
} htfs_brick_work;

/* The halo WITHOUT a communication library (transport "peer", round 5): the packing kernels store every message row straight
 * into the RECEIVER's staging buffer -- its own memory for a replica rank, an IPC-mapped buffer of another process / device over xGMI
 * otherwise -- and the last workgroup to finish publishes, per message, the row count and the exchange's sequence number
 * (system-scope release); the receiver's htfs_brick_unpack_halo waits for the numbers of its incoming messages (acquire, bounded
 * spin) and copies the rows into the ghost region.  Everything is ordinary kernel work on ONE stream -- capturable into a hipGraph,
 * and the stores travel while the interior rows are evaluated.  Staging is double-buffered by the sequence number's parity: a
 * neighbor that is a step ahead writes the other half.  Every rank performs the same sequence of exchanges (lockstep counters). */
typedef struct htfs_peer {
    void *inbox[HTFS_BRICK_MAX_MSG];       /* message m's destination: the inbox of the neighbor at offset m, [2][ghost rows] Scalar4 */
    unsigned *signal[HTFS_BRICK_MAX_MSG];  /* ... and that neighbor's signal words, [2 * n_msg]: {sequence, rows} per source offset */
    void *my_inbox;                        /* this rank's own inbox and signal words (what htfs_brick_unpack_halo reads) */
    unsigned *my_signal;
    unsigned *state;                       /* this rank's [4]: exchanges done, workgroups finished, timeouts seen, spare */
    unsigned spin_limit;                   /* polls of a signal word before giving up (HTFS_BF_HALO_TIMEOUT) */
} htfs_peer;

/* First half of a rebuild: destination of every local particle (d_bounds: per decomposed axis HTFS_BRICK_MAX_P + 1 boundaries in
 * the positions' dtype, axis-major), stable sort by destination, migrants packed into d_mig_send (rows of 8 scalars: position,
 * velocity; row 0 of each message = its count), shifted by shift[m]. */
HTF_API int htfs_brick_migrate_pack(const htfs_brick *g, const void *d_pos, const void *d_vel, int dtype, const void *d_bounds,
                                    const htfs_brick_work *w, void *d_mig_send, unsigned *d_counts, htf_stream stream);
/* Second half, after d_mig_recv holds the neighbors' messages: candidates = [stayed | from offset index n_msg-1 | ... | from 0],
 * classed in this brick, stable sort by class, written into the two segments with inert rows behind them (position NaN, velocity 0,
 * mass kept; d_n_neigh[row] = 0 if given); counts, class boundaries, halo message sizes and overflow flags into d_counts. */
HTF_API int htfs_brick_migrate_merge(const htfs_brick *g, void *d_pos, void *d_vel, int dtype, const void *d_bounds,
                                     const htfs_brick_work *w, const void *d_mig_recv, unsigned *d_n_neigh, unsigned *d_counts,
                                     htf_stream stream);
/* Per step: the halo messages packed from the boundary segment (inert rows behind each message's count) into d_send
 * [sum_m ghost_cap[m]] Scalar4; d_ghost_direct (nullable; replica mode without a transport): also written straight into the ghost
 * region, message m at the rows its receiver -- this rank -- expects it (source offset -o). */
HTF_API int htfs_brick_pack_halo(const htfs_brick *g, const void *d_pos, int dtype, const unsigned *d_counts, void *d_send,
                                 void *d_ghost_direct, htf_stream stream);
/* the same two packers with the rows stored into the neighbors' inboxes and signalled (transport "peer"), and the receiving side */
HTF_API int htfs_brick_pack_halo_peer(const htfs_brick *g, const void *d_pos, int dtype, const unsigned *d_counts, const htfs_peer *peer,
                                      htf_stream stream);
HTF_API int htfs_brick_unpack_halo(const htfs_brick *g, void *d_pos, int dtype, const htfs_peer *peer, unsigned *d_counts,
                                   htf_stream stream);
/* htfs_nve_step over the local rows AND htfs_brick_pack_halo in one launch: every boundary row writes its new position into the
 * messages that carry it (their inert tails stay as the last rebuild's pack left them).  Same bits as the two calls. */
HTF_API int htfs_brick_nve_halo(const htfs_brick *g, void *d_pos, void *d_vel, const void *d_force, int dtype, double dt,
                                const htf_box *box, const unsigned *d_counts, void *d_send, void *d_ghost_direct, htf_stream stream);
HTF_API int htfs_brick_nve_halo_peer(const htfs_brick *g, void *d_pos, void *d_vel, const void *d_force, int dtype, double dt,
                                     const htf_box *box, const unsigned *d_counts, const htfs_peer *peer, htf_stream stream);

/* ---- the integrator (and a brick's halo pack) as the EPILOGUE of the one-kernel step (round 6).  The lanes that hold a finished
 * row's force go on with htfs_nve_step's update of that row -- v += f dt in place, x(t + dt) into d_pos_next, the OTHER position
 * buffer (every wave still reads x(t) of everybody; the caller swaps the two arrays behind the launch) -- and, for a boundary row of
 * a brick, with its new position in every halo message that carries it (htfs_brick_nve_halo's rows, found through d_row_slots).
 * A plain step is then ONE launch where it was force kernel + integrator (+ pack): same bits (tests/test_gpu_standin.py).
 * A context holds up to HTFS_EPILOGUE_SLOTS descriptors in device memory (the two directions of a ping-pong between two position
 * arrays): htfs_set_step_epilogue registers one (a blocking upload: set-up time, not the step loop), htfs_use_step_epilogue names
 * the one every later htf_compute_forces[_rows] of the context carries (-1: none) -- a host-side word, free per step.  The launch
 * honours it on the one-kernel route without a virial (htf_config.fused != 0, a built-in closed form, batch_size 0, period 1, no
 * check_nlist, fp32 positions and forces: the fp64 wire measured slower with it); *applies says whether this context will -- when not, the caller integrates as
 * before. */
#define HTFS_EPILOGUE_SLOTS 2
typedef struct htfs_step_epilogue {
    void *d_vel;                 /* Scalar4[N] */
    void *d_pos_next;            /* Scalar4[N + ghosts] */
    int dtype;                   /* HTF_F32 / HTF_F64: the Scalar of both (and of the context) */
    double dt;
    htf_box box;                 /* the integrator's box (htfs_nve_step's) */
    const htfs_brick *brick;     /* NULL: no halo pack */
    const unsigned *d_row_slots; /* [cap_bnd][HTFS_BRICK_MAX_MSG]: htfs_brick_row_slots */
    void *d_halo_send;           /* [sum ghost_cap] Scalar4 (a transport sends it), or NULL */
    void *d_ghost_direct;        /* the ghost region of d_pos_next (this rank is its own neighbor), or NULL */
} htfs_step_epilogue;
HTF_API int htfs_set_step_epilogue(htf_ctx *ctx, int slot, const htfs_step_epilogue *ep, int *applies);
HTF_API int htfs_use_step_epilogue(htf_ctx *ctx, int slot);
/* d_row_slots[j][m] <- the slot of boundary row j in halo message m, 0xFFFFFFFF where the row is not in it (or beyond the message's
 * count): the class boundaries and slot table of the last re-plan, once per re-plan instead of once per row and step. */
HTF_API int htfs_brick_row_slots(const htfs_brick *g, const unsigned *d_counts, unsigned *d_row_slots, htf_stream stream);

/* ---- messages between ranks without a communication library (csrc/mailbox.hip): the migration messages of a re-plan and the
 * all-reduced distance check of transport "peer", so that every graph of a decomposed run is library-free.  Memory a neighbor's
 * kernels store into is allocated here -- fine-grained (hipExtMallocWithFlags(hipDeviceMallocFinegrained): coherent between
 * agents while kernels run, what a mapping across xGMI wants) or ordinary -- zeroed, and travels as a hipIpc handle. */
#define HTFS_IPC_HANDLE_BYTES 64
#define HTFS_MBOX_MAX_MSG 8
#define HTFS_MBOX_MAX_RANKS 64
HTF_API int htfs_shared_alloc(size_t bytes, int finegrained, void **out);
HTF_API int htfs_shared_free(void *p);
HTF_API int htfs_ipc_export(const void *p, void *handle64);          /* HTFS_IPC_HANDLE_BYTES bytes */
HTF_API int htfs_ipc_import(const void *handle64, void **out);        /* maps another process's allocation on the current device */
HTF_API int htfs_ipc_close(void *p);
/* One rank's view of a message channel.  Units are 16 bytes.  Every rank owns a mailbox of [2][half_units] units (two halves, by
 * the exchange number's parity) and one signal word per incoming message; message m of an exchange is stored at box_off[m] of the
 * receiver's half and its number published to remote_signal[m].  All ranks perform the same sequence of exchanges. */
typedef struct htfs_mailbox {
    void *remote[HTFS_MBOX_MAX_MSG];             /* message m's destination: the receiver's mailbox base */
    unsigned *remote_signal[HTFS_MBOX_MAX_MSG];  /* ... and the receiver's signal word for it */
    void *mine;                                  /* this rank's mailbox, */
    unsigned *my_signal;                         /* its signal words [n_msg], */
    unsigned *state;                             /* and [4]: exchanges done, workgroups finished, timeouts seen, spare */
    unsigned spin_limit;
    unsigned half_units;
} htfs_mailbox;
/* message m = units [send_off[m], + units[m]) of d_send -> box_off[m] of its receiver's mailbox.  row_units > 0: a message is rows
 * of that many units whose row 0 starts with its row count -- only row 0 and that many rows travel. */
HTF_API int htfs_mailbox_push(const htfs_mailbox *mb, int n_msg, const void *d_send, const unsigned *send_off, const unsigned *units,
                              const unsigned *box_off, int row_units, htf_stream stream);
/* the receiving half: message j of THIS exchange (its units at box_off[j] of my mailbox) -> units [recv_off[j], ...) of d_recv, behind
 * a bounded wait for its number; a message that does not arrive is delivered EMPTY (row count 0) and *d_flags |= flag_bit. */
HTF_API int htfs_mailbox_pull(const htfs_mailbox *mb, int n_msg, void *d_recv, const unsigned *recv_off, const unsigned *units,
                              const unsigned *box_off, int row_units, unsigned *d_flags, unsigned flag_bit, htf_stream stream);
/* d_value[0] <- max over the ranks (non-negative floats), one launch: every rank stores {exchange number, value} as one 8-byte
 * word into slot `rank` of every rank's table ([world][2] words) and polls its own.  A rank that does not answer within the spin
 * limit: the result is 3e38 (the caller rebuilds rather than trusts its list) and *d_flags |= flag_bit. */
typedef struct htfs_reduce_box {
    void *remote[HTFS_MBOX_MAX_RANKS];   /* rank r's table, r = 0 .. world - 1 (this rank's own included) */
    void *mine;
    unsigned *state;                     /* [4] */
    unsigned spin_limit;
    int world, rank;
} htfs_reduce_box;
HTF_API int htfs_mailbox_allreduce_max_f32(const htfs_reduce_box *rb, float *d_value, unsigned *d_flags, unsigned flag_bit, htf_stream stream);

#ifdef __cplusplus
}
#endif
#endif
