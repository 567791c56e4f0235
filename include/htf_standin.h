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

#ifdef __cplusplus
}
#endif
#endif
