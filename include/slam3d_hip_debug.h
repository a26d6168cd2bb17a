/* slam3d_hip_debug.h - the S3D_DBG_* bits of s3d_exec_options.debug_flags (include/slam3d_hip.h).
 *
 * NOT part of the drop-in API.  Every bit switches ONE fast path of the library off (or forces one of two equivalent
 * forms); none may change a bit of any result - that is what tests/ and tools_dev/ use them for (A/B timing, bitwise
 * neutrality tests).  A product caller leaves debug_flags = 0.  The values may change between revisions of the
 * library without an S3D_ABI_VERSION bump.
 */
#ifndef SLAM3D_HIP_DEBUG_H
#define SLAM3D_HIP_DEBUG_H

#define S3D_DBG_NN_NO_REVALIDATE   0x00000040u /* every pass searches every query (no triangle-inequality shortcut)      */
#define S3D_DBG_NN_NO_FAR_SEED     0x00000080u /* far previous neighbours are never trusted seeds                        */
#define S3D_DBG_KNN_FORCE_RINGS    0x00000200u /* k-NN pre-pass of a large batch: the ring search whatever the length of the far list (the
                                               * device hands a list shorter than 1 % of the points on to the exact search)              */
#define S3D_DBG_NO_K4_OVERLAP      0x00000400u /* a small batch's k-NN pre-pass on the context's own stream (not beside the first
                                               * correspondence pass on a second one)                                                  */
#define S3D_DBG_NN_NO_COOP         0x00000800u /* no wave-cooperative wide search                                        */
#define S3D_DBG_NN_NO_COMPACT      0x00010000u /* pass 4 (small batches: passes 3-5) without the block compaction         */
#define S3D_DBG_NN_NO_FIRST_KERNEL 0x00040000u /* pass 1 through the general kernel (implies the next one)               */
#define S3D_DBG_NN_NO_SCAN27       0x00080000u /* passes 2-3 through the general kernel                                  */
#define S3D_DBG_NN_NO_SETTLED      0x00100000u /* settled passes query by query (no record-level re-validation)          */
#define S3D_DBG_KNN_EXACT64        0x00200000u /* k-NN pre-pass: the exact 64-bit search for every point                 */
#define S3D_DBG_SORT_CLASSIC       0x00400000u /* radix sort: three kernels per pass, whatever the batch size            */
#define S3D_DBG_SORT_ONESWEEP      0x00800000u /* radix sort: one sweep per pass (decoupled look-back), whatever ...      */
#define S3D_DBG_SCAN27_NO_COMPACT  0x01000000u /* pass 3 without the block compaction                                    */
#define S3D_DBG_PRINT_KNN          0x02000000u /* stderr: how many points took the eigen fallback / the exact-search redo */
#define S3D_DBG_NN_FORCE_SETTLED   0x08000000u /* record-level re-validation for a small batch too (the host takes it from
                                                  65 536 records = 42 pairs of 100 k points: below that it is slower)   */
#define S3D_DBG_SORT_FULL_KEYS     0x04000000u /* radix sort: 8-bit digits everywhere (grid: 3 passes instead of 2 x 9 bits) */

#define S3D_DBG_NO_FUSED_PREPASS   0x10000000u /* registration pre-pass as two sorts (voxel keys, then cell ids) instead of
                                                  the one sort on (cell, voxel) keys                                     */
#define S3D_DBG_KNN_NO_FAR_COOP    0x20000000u /* k-NN pre-pass: the far declines of the fast path through the per-lane exact
                                                  search, whatever the batch size (default: small batches take the
                                                  wave-cooperative kernel)                                               */
#define S3D_DBG_KNN_FORCE_FAR_COOP 0x40000000u /* ... through the wave-cooperative kernel, whatever the batch size          */
#define S3D_DBG_KNN_NO_RINGS       0x80000000u /* k-NN pre-pass of a large batch: the sparse-region declines of the fast path
                                                  through the per-lane exact search instead of the ring-by-ring med3 search */

/* ---- test hooks: exported by the library, used by tests/ only ---------------------------------------------------- */
#ifdef __cplusplus
extern "C" {
#endif
/* throws inside an entry point, under the same guard every entry point has: kind 0 a HIP error, 1 std::bad_alloc,
 * 2 std::length_error, 3 std::runtime_error, 4 something that is not a std::exception, 5 std::bad_alloc on a worker
 * thread of s3d_cloud_upload_many's kind.  Must come back as a status (BACKEND_ERROR; INVALID_ARGUMENT for 2) with
 * s3d_last_error set, never as an exception or an abort (the reference's callers catch std::exception and log:
 * ScanSensor.cpp:74-77, :124-127). */
int s3d_debug_raise(s3d_context* ctx, int kind);
/* batches of this context that the fused pre-pass could not serve and that ran again on the two-sort path */
long long s3d_debug_fused_reruns(s3d_context* ctx);
/* the registration's pre-pass (fused != 0: one sort; 0: two) of two device clouds at voxel size `leaf`, then one
 * nearest-neighbour pass of the filtered target's points against the filtered source: the cell-sorted points of both
 * (xyzw, w = the tie-breaking id as raw bits), and per target point the position of its neighbour in the source array
 * (-1: none within max_distance) and the float squared distance.  capacity: points per output array. */
int s3d_debug_filtered_nn(s3d_context* ctx, s3d_cloud* source, s3d_cloud* target, double leaf, int fused,
                          double max_distance, int capacity, float* source_sorted_xyzw, int* n_source,
                          float* target_sorted_xyzw, int* n_target, int* corr_pos, float* corr_d2, int* fused_ok);
/* ---- measurement hook (bench.py roofline): time `reps` launches of the NN-search kernel
 *          as a FIRST correspondence pass (transformation_ = I, no radius hints, no re-validation
 *          of earlier correspondences — the most expensive pass of a registration) with HIP events
 *          on the context's stream.  n_queries / n_targets: points after the voxel filter. */
int  s3d_profile_nn_kernel(s3d_context* ctx, int n_pairs, s3d_cloud* const* sources, s3d_cloud* const* targets,
                           const double* guesses, const s3d_reg_params* params, int reps, double* avg_ms,
                           long long* n_queries, long long* n_targets);
#ifdef __cplusplus
}
#endif

#endif /* SLAM3D_HIP_DEBUG_H */
