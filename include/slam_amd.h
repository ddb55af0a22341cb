/*
 * slam_amd — C ABI of the MI355X-native two-view initialisation path
 * (8-point fundamental-matrix RANSAC + linear DLT triangulation with cheirality selection).
 *
 * The reference (patosai/slam) has no FFI layer: its boundary is the Python module surface of
 * src/epipolar.py and src/triangulation.py.  Each entry point below names the reference function it
 * replaces; slam_amd/epipolar.py and slam_amd/triangulation.py bind them with ctypes.
 *
 * Conventions
 *  - All matrices are row-major float64.  Point arrays are (N,2) row-major float64, exactly numpy's layout.
 *  - "host" entry points take host pointers, copy in/out and return after the work has finished.
 *  - "_dev" entry points take DEVICE pointers, enqueue work on the context's stream and return without
 *    synchronising; the caller owns all buffers.  They allocate only when a size exceeds what the context
 *    has seen before.
 *  - Return value: 0 on success, a negative SLAM_E* code otherwise; slam_last_error() gives the text.
 *  - A context is bound to one GPU and one stream; use one context per thread.
 *  - There is no CPU implementation behind this ABI: slam_create fails when no gfx950 device is usable.
 */
#ifndef SLAM_AMD_H
#define SLAM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slam_ctx slam_ctx;

enum {
    SLAM_OK = 0,
    SLAM_EINVAL = -1,      /* bad argument (NULL pointer, N < 8, H < 1, ...): the reference asserts here */
    SLAM_EHIP = -2,        /* a HIP runtime call failed */
    SLAM_ENOMEM = -3,      /* device or host allocation failed */
    SLAM_ESAMPLER = -4,    /* the sampler ran out of pre-generated random words (see slam_sample_indices) */
    SLAM_ENODEVICE = -5    /* no usable GPU */
};

/* the reference's constants: src/epipolar.py:54 (iterations), :63 (threshold), :70 (seed base) */
#define SLAM_DEFAULT_ITERATIONS 1000
#define SLAM_DEFAULT_THRESHOLD 0.005
#define SLAM_SEED_BASE 0x1337BEEFull

/* sampler kinds for the RANSAC entry points */
#define SLAM_SAMPLER_CPYTHON 0 /* random.seed(seed + it); random.sample(range(N), 8) : src/epipolar.py:70-71 */
#define SLAM_SAMPLER_COUNTER 1 /* Philox4x32-10 keyed by (seed, it): any GPU can draw any hypothesis */

/* record written by the RANSAC entry points: 12 float64 values */
#define SLAM_REC_COUNT 0  /* inlier count of the winner (integer value) */
#define SLAM_REC_ERRSUM 1 /* sum of its inlier residuals */
#define SLAM_REC_INDEX 2  /* its iteration index, it0 + h (integer value) */
#define SLAM_REC_F 3      /* its 3x3 F, row-major, entries 3..11 */
#define SLAM_REC_LEN 12

/* ---- lifecycle -------------------------------------------------------------------------------- */
/* The context's streams are created at the device's greatest stream priority, so that they get hardware queues of
 * their own (the HIP runtime pools queues per priority level and shares a pool among all streams of the level);
 * the environment variable SLAM_AMD_STREAM_PRIORITY=normal|low selects another level. */
int slam_create(int device, slam_ctx** out);
void slam_destroy(slam_ctx* ctx);
const char* slam_last_error(const slam_ctx* ctx); /* ctx may be NULL: text of the last slam_create failure */
/* Identifies the sources this library was built from: the first 16 hex digits of the SHA-256 of the library's source
 * files (slam_amd/csrc/Makefile: HASH).  Measurement files name it, so that counters taken with one build are not
 * quoted for another.  No counterpart in the reference. */
const char* slam_build_hash(void);
/* use the caller's HIP stream (e.g. torch's current stream) instead of the context's own; NULL restores it */
int slam_set_stream(slam_ctx* ctx, void* hip_stream);
/* Number of internal streams (1..4, default 1) that successive slam_ransac_f_dev calls rotate over.  With more than
 * one, consecutive calls may run concurrently: they must be independent of each other, use distinct output buffers,
 * and their outputs are ready only after slam_synchronize().  In that mode a call with more than 2048 matches, at most
 * 4096 hypotheses and no per-hypothesis outputs is ONE kernel launch: its selection (record and mask) is carried out
 * by the first workgroups of the kernel of the next such call on the same internal stream, or launched by
 * slam_synchronize (or any other entry point that waits for the context) when no next call comes; its fused kernel
 * gives a workgroup sixteen hypotheses instead of eight.  Every result is bit for bit that of a call on its own.
 * Every buffer a call names, inputs included, must stay allocated and unchanged until that synchronisation (the carried
 * selection reads the matches again for the mask).
 * Ignored while a caller-supplied stream is set.  No counterpart in the reference (it is single-threaded). */
int slam_set_concurrency(slam_ctx* ctx, int lanes);
/* Residual the RANSAC entry points score with (slam_ransac_f, slam_ransac_f_dev, slam_ransac_f_adaptive; the pair
 * pipelines always use the reference's).  SLAM_SCORE_ALGEBRAIC (default) is the reference's squared algebraic
 * residual ((x1^T F) x0)^2 (src/epipolar.py:73).  SLAM_SCORE_SAMPSON divides it by
 * (F^T x1)_0^2 + (F^T x1)_1^2 + (F x0)_0^2 + (F x0)_1^2, the first-order geometric distance; `threshold` then applies
 * to that quantity.  No counterpart in the reference. */
#define SLAM_SCORE_ALGEBRAIC 0
#define SLAM_SCORE_SAMPSON 1
int slam_set_scoring(slam_ctx* ctx, int mode);
/* Waits for everything the context has queued, on every internal stream.  Returns SLAM_ESAMPLER when a sampler of an
 * asynchronous (_dev / job) call queued since the last check ran out of pre-generated words: those calls cannot
 * report it when they return.  The flag is one sticky bit per context, reported ONCE by whichever checking call comes
 * first (this one, or any synchronous host entry point, which checks before it returns) and cleared by that report:
 * it says "some call queued on this context since the previous check overflowed", not which one.  A caller that
 * mixes asynchronous and synchronous calls and needs to attribute the error calls slam_synchronize before the
 * synchronous call.  (The condition needs 96 consecutive rejected words: probability below 1e-19 per iteration.) */
int slam_synchronize(slam_ctx* ctx);
/* Page-locked host memory for callers that want the host entry points to transfer at the full PCIe rate (the library
 * copies arrays in pageable memory through a staging area; arrays in page-locked memory go directly when they are
 * larger than that area).  No counterpart in the reference. */
int slam_host_alloc(slam_ctx* ctx, uint64_t bytes, void** out);
int slam_host_free(slam_ctx* ctx, void* p); /* ctx may be NULL: the memory may outlive its context */

/* ---- sampler: src/epipolar.py:62,69-71 ---------------------------------------------------------- */
/* number of hypotheses the reference runs: min(iterations, C(N, 8))   (src/epipolar.py:69) */
int64_t slam_num_hypotheses(int64_t n_points, int64_t iterations);
/* idx[h*k + j] = j-th index of the sample drawn in iteration it0 + h.  kind = SLAM_SAMPLER_*.
 * k is 8 for the fundamental-matrix loop and 6 for the tracking loop (src/triangulation.py:155).
 * The draw runs on the GPU from a table of MT19937 output words kept per context. */
int slam_sample_indices(slam_ctx* ctx, int kind, uint64_t seed, int64_t it0, int H, int N, int k, int32_t* idx);

/* ---- eight-point solve: src/epipolar.py:18-51  calculate_fundamental_matrix ---------------------- */
/* p0 = img0_points, p1 = img1_points, (M,2) each, M >= 8.  F: 9 values, rank 2, in the coordinates of the
 * input points, sign arbitrary (as the reference's). */
int slam_eightpoint(slam_ctx* ctx, const double* p0, const double* p1, int M, double* F);

/* ---- RANSAC: src/epipolar.py:54-84  calculate_fundamental_matrix_with_ransac --------------------- */
/* Scores H hypotheses (iterations it0 .. it0+H-1) on N matches and selects by the reference's rule
 * (more inliers, then smaller inlier-error sum, then earlier iteration: src/epipolar.py:78-81).
 * idx: H*8 pre-drawn indices, or NULL to draw them with `sampler`/`seed`.
 * Outputs (any may be NULL except record): record[SLAM_REC_LEN]; mask[N] (1 = inlier of the winner);
 * counts[H], errsums[H], F_all[H*9] per hypothesis. */
int slam_ransac_f(slam_ctx* ctx, const double* p0, const double* p1, int N, const int32_t* idx, int sampler,
                  uint64_t seed, int64_t it0, int H, double threshold, double* record, uint8_t* mask,
                  int32_t* counts, double* errsums, double* F_all);
/* same with device pointers, asynchronous on the context's stream.
 * A call with more than 4096 hypotheses and at least 2^27 (hypothesis, match) pairs that scores the reference's
 * residual and passes d_errsums == NULL counts inliers in packed float32, evaluates every slice of matches that has a
 * lane near the threshold again in float64, and computes float64 error sums only for the hypotheses that tie on the
 * largest count: record, mask and counts are bit for bit those of the float64 kernel, which a call that passes
 * d_errsums takes (environment variable SLAM_AMD_F32_COUNTS=0: always). */
int slam_ransac_f_dev(slam_ctx* ctx, const double* d_p0, const double* d_p1, int N, const int32_t* d_idx,
                      int sampler, uint64_t seed, int64_t it0, int H, double threshold, double* d_record,
                      uint8_t* d_mask, int32_t* d_counts, double* d_errsums, double* d_F_all);
/* slam_ransac_f_dev for callers that cannot allocate.  No counterpart in the reference: it exists for code that runs
 * under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  The call allocates nothing on the
 * device or through the HIP runtime, frees nothing and synchronises nothing: the per-hypothesis scratch (F, inlier
 * counts, inlier-error sums) lives in a device array that is part of the library's code object (1.3 MiB per GPU, held
 * until the process ends).  Differences from slam_ransac_f_dev:
 *  - H <= SLAM_NOALLOC_MAX_H, otherwise SLAM_EINVAL (split a longer range into several calls and merge the records
 *    with slam_merge_records);
 *  - the per-hypothesis outputs are not available (they are the scratch);
 *  - with d_idx == NULL and SLAM_SAMPLER_CPYTHON the context's MT19937 word table must already cover
 *    [it0, it0 + H) for this seed (any earlier call with that seed and range made it); otherwise SLAM_EINVAL.
 *    Pre-drawn indices and the counter-based sampler need no table;
 *  - under HIP stream capture (the context's stream set with slam_set_stream is capturing) the CPython-compatible
 *    sampler without pre-drawn indices is refused with SLAM_EINVAL: its word table belongs to the context and any
 *    later call with another seed or range rewrites or moves it, so a replayed graph would draw other samples or read
 *    freed memory.  Capture with d_idx (draw the indices once with slam_sample_indices and keep them in a buffer of
 *    your own) or with SLAM_SAMPLER_COUNTER;
 *  - always on the context's ordered stream (slam_set_concurrency does not apply);
 *  - more than 4096 hypotheses on more than 4096 matches are scored by the fused kernel in ranges of 4096 hypotheses
 *    (the per-chunk partial tables of the usual large-H path are a buffer that grows): same bits, but every workgroup
 *    reads all matches, so this corner is slower than slam_ransac_f_dev.
 * Every call on one GPU shares that GPU's array, from whatever context: calls that overlap on one GPU (two contexts,
 * or two streams set with slam_set_stream) race.  Results are bit-identical to slam_ransac_f_dev's. */
#define SLAM_NOALLOC_MAX_H 16384
int slam_ransac_f_dev_noalloc(slam_ctx* ctx, const double* d_p0, const double* d_p1, int N, const int32_t* d_idx,
                              int sampler, uint64_t seed, int64_t it0, int H, double threshold, double* d_record,
                              uint8_t* d_mask);
/* A prepared slam_ransac_f_dev call.  The arguments are checked and stored once; slam_job_run enqueues the same work
 * again, exactly as slam_ransac_f_dev would (including the rotation over the context's internal streams).  For
 * callers whose per-call argument marshalling is a cost of its own (ctypes spends 2 us on the 15 arguments, as much as
 * the two kernel launches): frame loops that score the same buffers again and again.  The buffers must stay valid
 * until the job is destroyed.  No counterpart in the reference. */
typedef struct slam_job slam_job;
int slam_ransac_job_create(slam_ctx* ctx, const double* d_p0, const double* d_p1, int N, const int32_t* d_idx,
                           int sampler, uint64_t seed, int64_t it0, int H, double threshold, double* d_record,
                           uint8_t* d_mask, int32_t* d_counts, double* d_errsums, double* d_F_all, slam_job** out);
int slam_job_run(slam_job* job);
void slam_job_destroy(slam_job* job);
/* Adaptive termination (the standard RANSAC stopping rule; no counterpart in the reference, which always runs every
 * iteration).  Iterations 0, 1, 2, ... are scored in chunks of `chunk`; after each chunk the number of iterations
 * needed to have drawn, with probability `confidence`, at least one sample of 8 inliers is re-estimated from the best
 * inlier ratio w so far, ceil(log(1 - confidence) / log(1 - w^8)), and the loop stops once that many (or
 * max_iterations) have been scored.  The outcome is exactly the fixed-length outcome of slam_ransac_f for
 * *iterations_run iterations: chunks are contiguous ranges merged by the reference's rule.  Host pointers. */
int slam_ransac_f_adaptive(slam_ctx* ctx, const double* p0, const double* p1, int N, int sampler, uint64_t seed,
                           int64_t max_iterations, int chunk, double confidence, double threshold, double* record,
                           uint8_t* mask, int64_t* iterations_run);
/* Merge per-rank winner records (host memory, n_records * SLAM_REC_LEN) by the same rule; used after the
 * all-gather of a hypothesis-sharded run.  Writes the winning record to out and returns its position. */
int slam_merge_records(const double* records, int n_records, double* out);

/* ---- exchange step of sharded runs (no counterpart in the reference, which is a single process) ---------------- */
/* The reference's selection rule (src/epipolar.py:78-81) is what these apply ACROSS ranks; the collectives are RCCL
 * calls on the context's stream, so a rank's kernels, the exchange and whatever follows stay queued on the device
 * without a host round trip.  RCCL is loaded at run time (librccl.so.1; SLAM_AMD_RCCL names another file): a process
 * that never calls these never loads it.
 * Communicator: either the context's own - rank 0 calls slam_comm_unique_id, hands the 128 bytes to the other ranks
 * by any means (a file, MPI, torch.distributed), every rank calls slam_comm_init_rank - or any RCCL communicator of
 * the caller's whose device is the context's, passed as rccl_comm (NULL selects the context's own). */
#define SLAM_COMM_ID_BYTES 128
int slam_comm_unique_id(slam_ctx* ctx, void* id128);
int slam_comm_init_rank(slam_ctx* ctx, const void* id128, int n_ranks, int rank); /* collective; destroyed with the context */
/* Number of ranks RCCL itself reports for a communicator (ncclCommCount): the context's own when rccl_comm is NULL.
 * For launchers and benchmarks that want to state how many ranks the exchange calls below really span. */
int slam_comm_count(slam_ctx* ctx, void* rccl_comm, int* n_ranks);
/* Hypothesis-sharded RANSAC: every rank passes the record its own range produced (device memory, SLAM_REC_LEN
 * doubles, e.g. d_record of slam_ransac_f_dev); after the call every rank's d_record holds the global winner:
 * all-gather of the records, then the rule above (more inliers, smaller inlier-error sum, earlier iteration) applied
 * by a kernel.  Asynchronous on the context's stream, after everything the context has queued so far (with
 * slam_set_concurrency(n > 1) that stream first waits, on the device, for the internal streams). */
int slam_exchange_winner_dev(slam_ctx* ctx, void* rccl_comm, double* d_record);
/* Point-sharded pose recovery: sums the four in-front counts of slam_pose_candidates_dev over the ranks in place
 * (int64[4], device memory), between that call and slam_pose_select_dev.  Asynchronous on the context's stream. */
int slam_allreduce_counts4_dev(slam_ctx* ctx, void* rccl_comm, int64_t* d_counts4);
/* slam_merge_records on records in device memory (n_records x SLAM_REC_LEN), asynchronous on the context's stream:
 * the merge step alone, for callers that gather the records by other means. */
int slam_merge_records_dev(slam_ctx* ctx, const double* d_records, int n_records, double* d_out);

/* ---- caller-side glue of custom_slam.py --------------------------------------------------------- */
/* custom_slam.py:85-93 normalize_points: out (N,2), T 3x3 (with the reference's s where 1/s is expected) */
int slam_normalize_points(slam_ctx* ctx, const double* p, int N, double* out, double* T);
/* src/epipolar.py:87-92 fundamental_to_essential_matrix: E = K^T F K / ||.||_F */
int slam_fundamental_to_essential(slam_ctx* ctx, const double* F, const double* K, double* E);

/* ---- triangulation: src/triangulation.py:8-75  triangulate_points_from_pose --------------------- */
/* P0, P1: first three rows (3x4, 12 values) of the two camera matrices; X: (N,3). */
int slam_triangulate(slam_ctx* ctx, const double* P0, const double* P1, const double* p0, const double* p1,
                     int64_t N, double* X);
int slam_triangulate_dev(slam_ctx* ctx, const double* P0_host, const double* P1_host, const double* d_p0,
                         const double* d_p1, int64_t N, double* d_X);

/* ---- pose recovery: src/epipolar.py:95-136  calculate_pose_from_essential_matrix ---------------- */
/* Triangulates all N matches for the four (R,t) decompositions of E and keeps the one with most points in
 * front of both cameras (strict >, candidates in the order (R0,t0),(R0,t1),(R1,t0),(R1,t1)).
 * pose: 4x4; X: (N,3) points of the winner; counts4: in-front counts per candidate; winner: its position.
 * The ORDER of the candidates follows this library's own SVD sign choices, not LAPACK's. */
int slam_pose_from_E(slam_ctx* ctx, const double* E, const double* K, const double* p0, const double* p1,
                     int64_t N, double* pose, double* X, int32_t* counts4, int32_t* winner);
/* Two-phase device form for point-sharded runs: phase 1 triangulates this rank's points for all four
 * candidates and leaves the four in-front counts in d_counts4 (int64, to be summed across ranks by the
 * caller); phase 2 picks the winner from the (summed) counts and writes pose (16) and this rank's X. */
int slam_pose_candidates_dev(slam_ctx* ctx, const double* E_host, const double* K_host, const double* d_p0,
                             const double* d_p1, int64_t N, int64_t* d_counts4);
int slam_pose_select_dev(slam_ctx* ctx, const int64_t* d_counts4, int64_t N, double* d_pose, double* d_X,
                         int32_t* d_winner);

/* ---- whole pair: custom_slam.py:96-115 (normalise, RANSAC, denormalise, F->E, pose, z>0 mask) ----- */
/* pts0, pts1: pixel matches (N,2).  Outputs: F (denormalised), E, pose 4x4, X (N,3), good[N] (X_z > 0),
 * record (as slam_ransac_f), counts4, winner.  One upload and one download per call. */
int slam_init_pair(slam_ctx* ctx, const double* pts0, const double* pts1, int N, const double* K, int iterations,
                   double* F, double* E, double* pose, double* X, uint8_t* good, double* record,
                   int32_t* counts4, int32_t* winner);

/* A batch of independent pairs in one call: one upload, one download, one synchronisation, and ONE launch per
 * pipeline stage for the whole batch on the context's stream (every kernel picks its pair with blockIdx.y; the fused
 * RANSAC kernel is launched once per scoring shape present, at most three times).  The large per-match outputs X and
 * good are copied straight into the caller's arrays (page-locked arrays from slam_host_alloc transfer fastest); the
 * results are complete when the call returns.  pts0/pts1: the pairs' matches concatenated, (offsets[P], 2);
 * pair p owns rows offsets[p] .. offsets[p+1]-1 (at least 8).  Outputs as slam_init_pair, per pair: F, E (P x 9),
 * pose (P x 16), record (P x 12), counts4 (P x 4), winner (P); per match: X (offsets[P] x 3), good (offsets[P]).
 * Any output may be NULL.  Results are bit-identical to P calls of slam_init_pair. */
int slam_init_pairs(slam_ctx* ctx, const double* pts0, const double* pts1, const int64_t* offsets, int P,
                    const double* K, int iterations, double* F, double* E, double* pose, double* X, uint8_t* good,
                    double* record, int32_t* counts4, int32_t* winner);

/* ---- tracking step: src/triangulation.py:78-182 ------------------------------------------------- */
/* number of hypotheses the tracking loop runs: min(iterations, C(M, 6))   (src/triangulation.py:145-146) */
int64_t slam_num_tracking_hypotheses(int64_t n_known, int64_t iterations);
/* src/triangulation.py:78-132 triangulate_pose_from_points: camera pose from M >= 6 matches between image points
 * (M,2) and known 3D points (M,3) by DLT resection.  pose: 4x4.  extra (may be NULL): 24 values, the quantities the
 * reference prints: calculated intrinsic matrix (9), rotation (9), translation (3), camera centre (3).
 * The intrinsic matrix argument of the reference is not used by it and has no counterpart here. */
int slam_pose_from_points(slam_ctx* ctx, const double* image_points, const double* known_3d_points, int M,
                          double* pose, double* extra);
/* src/triangulation.py:135-182 triangulate_pose_from_points_with_ransac.
 * previous_pose: 4x4 (host).  prev_known/next_known (M,2), known_3d_points (M,3): the matches with a known 3D
 * position; prev_all/next_all (N,2): every match, triangulated at the end with (previous_pose, winning pose) as the
 * reference does (4x4 poses, no K).  Draws min(iterations, C(M,6)) samples of 6 with the reference's seeds
 * (src/triangulation.py:154-155), solves, scores each by points in front of both cameras and then by the summed 3D
 * distance to the known points, and keeps the first best.
 * Outputs: pose (16); X (N,3) (may be NULL when N == 0); best[3] = winner position, its count, its error (may be
 * NULL); counts[H], errs[H], poses[H*16] per hypothesis (each may be NULL). */
int slam_track_pose(slam_ctx* ctx, const double* previous_pose, const double* prev_known, const double* next_known,
                    const double* known_3d_points, int M, const double* prev_all, const double* next_all, int64_t N,
                    const double* K, int iterations, double* pose, double* X, double* best, int32_t* counts,
                    double* errs, double* poses);
/* same with device pointers for the point arrays and outputs (previous_pose and K stay host pointers, they travel
 * as kernel arguments); asynchronous on the context's stream.  d_best: 3 doubles. */
int slam_track_pose_dev(slam_ctx* ctx, const double* previous_pose, const double* d_prev_known,
                        const double* d_next_known, const double* d_known_3d_points, int M, const double* d_prev_all,
                        const double* d_next_all, int64_t N, const double* K, int iterations, double* d_pose,
                        double* d_X, double* d_best);

/* ---- descriptor matching: custom_slam.py:14-26, 56-82 --------------------------------------------- */
/* Descriptors are packed 256-bit strings, 32 bytes each (what ORB emits; the reference unpacks them to one byte
 * per bit before matching, custom_slam.py:51, which changes nothing in the distances).  At most 2^22 train
 * descriptors.
 * custom_slam.py:14-26 match_knn_brute_force with num_matches = k (1 to 8): index, distance are (n0, k) int32, nearest
 * first, equal distances in ascending train index (the reference's stable sort); -1 where n1 < k. */
int slam_match_knn(slam_ctx* ctx, const uint8_t* des0, int n0, const uint8_t* des1, int n1, int k, int32_t* index,
                   int32_t* distance);
/* custom_slam.py:60-73: 2-NN of every descriptor of image 0 among those of image 1, the ratio test
 * (best distance < 0.5 * second best) and the matched point lists, in image-0 keypoint order.
 * pts0 (n0,2), pts1 (n1,2): keypoint coordinates.  matched0/matched1: room for (n0,2) each; *n_matched rows are
 * written.  index/distance: (n0,2) int32 2-NN results, may be NULL. */
int slam_match_points(slam_ctx* ctx, const uint8_t* des0, const double* pts0, int n0, const uint8_t* des1,
                      const double* pts1, int n1, double* matched0, double* matched1, int32_t* n_matched,
                      int32_t* index, int32_t* distance);
/* same with device pointers (descriptor pointers 16-byte aligned), asynchronous on the context's stream;
 * d_n_matched: one int32 on the device.  The matched lists can go straight into slam_ransac_f_dev once the count
 * has been read back. */
int slam_match_points_dev(slam_ctx* ctx, const uint8_t* d_des0, const double* d_pts0, int n0, const uint8_t* d_des1,
                          const double* d_pts1, int n1, double* d_matched0, double* d_matched1, int32_t* d_n_matched,
                          int32_t* d_index, int32_t* d_distance);

/* ---- a frame pair from features: custom_slam.py:105-115 after the ORB calls ------------------------- */
/* Keypoints and packed descriptors of two images in; 2-NN matching, ratio test, then the whole-pair pipeline of
 * slam_init_pair on the kept matches, which never leave the device in between (only their count comes back, to size
 * the launches).  matched0/matched1: room for (n0,2) each, *n_matched rows written; the per-match outputs X
 * (n0 x 3 room) and good (n0 room) have *n_matched rows.  Returns SLAM_EINVAL (after writing the matches) when fewer
 * than 8 matches survive, where the reference's RANSAC asserts. */
int slam_frame_pair(slam_ctx* ctx, const uint8_t* des0, const double* pts0, int n0, const uint8_t* des1,
                    const double* pts1, int n1, const double* K, int iterations, double* matched0, double* matched1,
                    int32_t* n_matched, double* F, double* E, double* pose, double* X, uint8_t* good, double* record,
                    int32_t* counts4, int32_t* winner);

/* ---- kernel timing for benchmarks ---------------------------------------------------------------- */
#define SLAM_K_SOLVE 0
#define SLAM_K_SCORE 1
#define SLAM_K_SELECT 2
#define SLAM_K_TRIANGULATE 3
#define SLAM_K_RESECT 4
#define SLAM_K_TRACK_SCORE 5
#define SLAM_K_MATCH 6
#define SLAM_K_COUNT 7
/* When enabled, every launch of the kernels above is bracketed by HIP events on the context's stream. */
int slam_profile_enable(slam_ctx* ctx, int on);
/* Synchronises, then returns total milliseconds and number of launches recorded for `kernel` since the last reset. */
int slam_profile_read(slam_ctx* ctx, int kernel, double* total_ms, int64_t* launches);
int slam_profile_reset(slam_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* SLAM_AMD_H */
