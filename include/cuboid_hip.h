/*
 * cuboid_hip.h - C-ABI of libcuboid_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the per-frame point-cloud path of dash-robotics/perception:
 *   crop -> voxel-downsample -> RANSAC ground plane -> extract -> Euclidean clusters ->
 *   point-to-point ICP of each cluster against a template cuboid.
 *
 * The reference has no FFI of its own for this path: every numeric step is a PCL object
 * used as "configure -> setInput -> one blocking compute call -> read outputs" inside the
 * ROS callbacks.  Each entry point below replaces one such PCL call site; the citation
 * next to it is the reference line it stands in for (paths relative to the reference
 * repository root; gps.cpp = cuboid_detection/src/ground_plane_segmentation.cpp,
 * icp.cpp = cuboid_detection/src/iterative_closest_point.cpp,
 * opd.cpp = object_detection/src/object_pose_detection.cpp).
 *
 * Conventions
 *  - plain pointers and sizes only; caller owns every buffer; capacities are passed in and
 *    counts are returned.  Point inputs are (base pointer, byte stride, count): x,y,z are
 *    float32 at byte offsets 0/4/8 of each record, so pcl::PointXYZ (16 B),
 *    pcl::PointXYZRGB (32 B) and raw sensor_msgs/PointCloud2 blobs pass without repacking.
 *  - every function returns CD_OK (0) or a negative cd_status; nothing throws or aborts
 *    across the boundary; cd_last_error() returns a message for the last failure.
 *  - a context is NOT thread-safe (mirrors the reference's single ros::spin() thread,
 *    gps.cpp:153); distinct contexts are independent (one per GPU / per process).
 *  - there is no CPU fallback: if no HIP device is usable cd_create() fails with
 *    CD_ERR_DEVICE and nothing else can be called.
 *  - results are a pure function of the inputs (the only randomness is PCL's fixed-seed
 *    RANSAC sampler, re-seeded per frame exactly as PCL does).
 */
#ifndef CUBOID_HIP_H
#define CUBOID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CD_ABI_VERSION 4
#define CD_MAX_TEMPLATES 8          /* template slots per context (BASELINE config 5 uses 5) */
#define CD_MAX_CLUSTERS_PER_FRAME 8 /* cluster slots in the fixed-size per-frame record; a frame with more clusters still
                                     * gets an ICP for every one of them (opd.cpp:376): see cd_get_cluster_results       */
#define CD_FRAME_MORE_CLUSTERS 1    /* cd_frame_result.flags: n_clusters > CD_MAX_CLUSTERS_PER_FRAME                    */

typedef struct cd_context cd_context;

typedef enum cd_status {
    CD_OK = 0,
    CD_ERR_INVALID_ARG = -1,   /* null pointer, bad stride, count out of range             */
    CD_ERR_CAPACITY = -2,      /* input larger than the context / output buffer too small   */
    CD_ERR_DEVICE = -3,        /* HIP runtime error or no usable device                     */
    CD_ERR_NO_MODEL = -4,      /* RANSAC found no plane (PCL: empty inliers+coefficients)   */
    CD_ERR_FEW_CORRESPONDENCES = -5, /* ICP source has < 3 points (PCL: "Not enough correspondences") */
    CD_ERR_LEAF_TOO_SMALL = -6,/* voxel grid would overflow int32 indices (PCL warns)       */
    CD_ERR_NO_TEMPLATE = -7    /* template slot empty                                       */
} cd_status;

/* Parameters of the whole chain.  Defaults (cd_default_params) are the cuboid_detection
 * launch values with object_detection's clustering constants. */
typedef struct cd_params {
    /* S0 PassThrough "z" then "x" (gps.cpp:53-65, opd.cpp:273-289); limits inclusive, double */
    double crop_z_min, crop_z_max;      /* 0.0, 0.9  */
    double crop_x_min, crop_x_max;      /* -0.2, 0.2 */
    /* S1 VoxelGrid leaf (gps.cpp:72; launch: 0.005 cuboid / 0.001 object) */
    float leaf_size;
    int32_t rgb_offset;                 /* byte offset of packed rgb in a record, -1 = none */
    /* S2 SACSegmentation PLANE/RANSAC (gps.cpp:85-89) */
    double plane_distance_threshold;    /* launch: 0.015 */
    int32_t plane_max_iterations;       /* 1000 */
    int32_t plane_optimize;             /* setOptimizeCoefficients(true) */
    double plane_probability;           /* PCL default 0.99 */
    /* S3 ExtractIndices (gps.cpp:100): 1 keeps the non-plane points */
    int32_t extract_negative;
    /* S3b second PassThrough "z" on the extracted cloud (opd.cpp:331-336); 0 disables */
    int32_t crop2_enable;
    double crop2_z_min, crop2_z_max;    /* 0.0, 0.75 */
    /* S5 EuclideanClusterExtraction (opd.cpp:356-358); cluster_enable=0 feeds the whole
     * extracted cloud to ICP as one source (the cuboid_detection flavour, icp.cpp:156,171) */
    int32_t cluster_enable;
    int32_t cluster_min_size, cluster_max_size; /* 200, 25000 */
    double cluster_tolerance;           /* 0.02 */
    /* S6 IterativeClosestPoint (icp.cpp:173-176, opd.cpp:223-226) */
    int32_t icp_max_iterations;         /* 5000 */
    int32_t template_slot;              /* >= 0: that slot; -1: every loaded template, best fitness wins (BASELINE config 5) */
    double icp_transformation_epsilon;  /* 1e-9 */
    double icp_euclidean_fitness_epsilon; /* = icp_fitness_score param, 0.0004 (relative MSE) */
    double icp_accept_fitness;          /* acceptance test of icp.cpp:182, 0.0004 */
    /* Optional image-space gate of cuboid_detection/src/bbox_filter.cpp (within_bbox, :30-51): a point of
     * the extracted cloud is kept iff its projection u = (P0 x + P1 y + P2 z + P3)/w, v = ..., lies
     * strictly inside the rectangle (x1 < u < x2, y1 < v < y2).  P = CameraInfo.P (:60-62), row-major
     * 3x4 doubles; rectangle = the Rectangle message (:69-75).  0 disables (the launch default). */
    double bbox_P[12];
    int32_t bbox_enable;
    int32_t bbox_rect[4];               /* x1, y1, x2, y2 */
    /* Axis-constrained plane models of cuboid_detection/src/surface_normal_estimation.cpp:118-123
     * (seg.setModelType / setAxis / setEpsAngle): CD_PLANE = SACMODEL_PLANE (every other call site),
     * CD_PLANE_PERPENDICULAR = SACMODEL_PERPENDICULAR_PLANE (normal within eps of the axis),
     * CD_PLANE_PARALLEL = SACMODEL_PARALLEL_PLANE (normal within eps of perpendicular to the axis).
     * A hypothesis that violates the constraint scores 0 inliers but still counts as an iteration
     * (PCL: countWithinDistance starts with isModelValid); the refined model is tested again. */
    int32_t plane_model;
    float plane_axis[3];
    double plane_eps_angle;             /* radians; sne.cpp:123 uses 0.1 */
    /* Initial guess of the registration: pcl::Registration::align(output, guess).  The reference's live path calls
     * align(output) - the identity guess - and that is the default here (CD_GUESS_NONE).  Its authors meant to start from
     * the pose surface_normal_estimation publishes (icp.cpp:130-134 stores it, :165-167 moves the template by it; both
     * commented out / inert, and iterative_closest_point.launch:17-18 leaves the sne node out), so the guess is opt-in:
     * CD_GUESS_PARAMS uses icp_guess for every ICP of the call, CD_GUESS_PER_FRAME the matrix cd_set_frame_guesses stored
     * for the cluster's frame.  PCL semantics: the source is first moved by the guess (input_transformed = guess * source,
     * float32 4x4 * point), final_transformation_ starts as the guess, the iterations and the convergence tests run on the
     * moved cloud, getFinalTransformation() includes the guess and getFitnessScore() is that of final * source. */
    int32_t icp_use_guess;
    float icp_guess[16];                /* row-major, scene -> template */
} cd_params;

enum { CD_GUESS_NONE = 0, CD_GUESS_PARAMS = 1, CD_GUESS_PER_FRAME = 2 };

enum { CD_PLANE = 0, CD_PLANE_PERPENDICULAR = 1, CD_PLANE_PARALLEL = 2 };

/* Output of cd_surface_frame: what surface_normal_estimation.cpp's callback (:167-234) derives from
 * three constrained plane fits.  Planes are in the callback's final order (most points first). */
typedef struct cd_surface_frame_result {
    float Rt[16];          /* row-major 4x4: columns normal[2], normal[1], normal[0], centroid (sne.cpp:218-222) */
    float coeff[3][4];     /* plane coefficients, sorted order                                  */
    float midpoint[3][4];  /* pcl::compute3DCentroid of each plane's points                      */
    int32_t n_points[3];   /* points of each plane                                               */
    int32_t iterations[3]; /* RANSAC iterations of the three fits, in FIT order                  */
    int32_t reserved[2];
} cd_surface_frame_result;

/* One ICP result: what icp.cpp:178-182 / opd.cpp:228-235 read back from PCL. */
typedef struct cd_cluster_result {
    int32_t size;          /* N_s, points in the cluster                              */
    int32_t iterations;    /* nr_iterations_                                          */
    int32_t converged;     /* icp.hasConverged()                                      */
    int32_t accepted;      /* converged && fitness < icp_accept_fitness (icp.cpp:182) */
    int32_t template_slot; /* slot this result was registered against (best fitness when all slots run) */
    int32_t reserved;
    float T[16];           /* getFinalTransformation(), row-major, scene -> template  */
    double fitness;        /* getFitnessScore()                                       */
    double pose[16];       /* T.cast<double>().inverse() (icp.cpp:179), row-major     */
} cd_cluster_result;

/* Fixed-size per-frame record (this is what is gathered across GPUs). */
typedef struct cd_frame_result {
    int32_t status;        /* cd_status of this frame                                 */
    int32_t n_cropped;     /* N_c after S0                                            */
    int32_t n_voxels;      /* N_v after S1                                            */
    int32_t n_plane;       /* refined plane inliers (S2)                              */
    int32_t n_objects;     /* N_o points after S3(+S3b)                               */
    int32_t n_clusters;    /* K clusters found (may exceed the slots below)           */
    int32_t ransac_iterations; /* PCL iterations_ consumed by the adaptive loop       */
    int32_t flags;         /* CD_FRAME_MORE_CLUSTERS                                  */
    float plane[4];        /* refined coefficients a,b,c,d                            */
    float pad[4];
    cd_cluster_result clusters[CD_MAX_CLUSTERS_PER_FRAME]; /* size-descending          */
} cd_frame_result;

void cd_default_params(cd_params* p);
int cd_abi_version(void);
/* sizeof() of the ABI structs, for FFI layers to verify their mirror of this header:
 * which = 0 cd_params, 1 cd_cluster_result, 2 cd_frame_result, 3 cd_timing. */
int cd_struct_size(int which);

/* Object lifetimes (replaces construction/destruction of the PCL objects and the node's
 * globals, icp.cpp:26-46).  max_points = largest N per frame, max_frames = largest batch. */
int cd_create(int device_id, int max_points, int max_frames, cd_context** out);
void cd_destroy(cd_context* ctx);
const char* cd_last_error(const cd_context* ctx);

/* pcl::io::loadPCDFile + icp.setInputTarget (icp.cpp:159,172; opd.cpp:398,222): the
 * template is uploaded once and stays device-resident. */
int cd_set_template(cd_context* ctx, int slot, const void* xyz, size_t stride_bytes, int m);

/* What cd_set_template found: the number of lattice faces of the slot's template, 0 when it is an arbitrary cloud (negative
 * cd_status on a bad slot).  cuboid_detection/templates/make_cuboid.py:38-55 writes every cuboid template as faces that are
 * Cartesian products of shared axis tables; cd_set_template verifies that structure bit by bit and, when it holds, the ICP
 * finds nearest neighbours in closed form instead of searching (same neighbour, same lowest-index tie rule). */
int cd_template_lattice_faces(const cd_context* ctx, int slot);

/* The correspondence search of icp.align on its own (pcl::registration::CorrespondenceEstimation -> KdTreeFLANN
 * nearestKSearch(k = 1), behind icp.cpp:178 / opd.cpp:228): for each of n query points the ORIGINAL index of the nearest
 * point of the slot's template (ties: lowest index) and the squared distance, float32, (dx*dx + dy*dy) + dz*dz.  Runs the
 * closed-form lattice search; CD_ERR_INVALID_ARG for a template that is not a lattice (its searches only exist inside the ICP
 * kernels). */
int cd_template_nearest(cd_context* ctx, int slot, const void* queries, size_t stride_bytes, int n, int32_t* out_index,
                        float* out_d2);

/* Host-only (no context, no GPU): the lattice test of cd_set_template.  out (may be NULL) receives up to 8 faces x
 * {constant axis, fast axis, first index, points along the fast axis, points along the slow axis}.  Returns the number of
 * faces, 0 = not a lattice. */
int cd_lattice_detect(const void* xyz, size_t stride_bytes, int m, int32_t* out);

/* S0+S1: two PassThrough filters + VoxelGrid::filter (gps.cpp:53-73).  out_xyz receives
 * N_v * 3 floats in ascending voxel-index order, out_rgb (may be NULL) N_v packed rgb. */
int cd_crop_voxel(cd_context* ctx, const void* points, size_t stride_bytes, int n,
                  const cd_params* prm, float* out_xyz, uint32_t* out_rgb, int capacity,
                  int* out_n_cropped, int* out_n_voxels);

/* One PassThrough filter as a call of its own: pcl::PassThrough<PCLPointCloud2>::filter (gps.cpp:53-58 "z", :61-65 "x";
 * opd.cpp:273-289, :331-336).  field = 0 / 1 / 2 for setFilterFieldName("x" / "y" / "z"), -1 for none (only non-finite points
 * go).  A record is kept iff x, y, z and the field are finite and !(v > limit_max || v < limit_min), v compared as double
 * (negative != 0, setFilterLimitsNegative: kept iff !(v < limit_max && v > limit_min)); kept records are copied whole
 * (stride_bytes each, a multiple of 4, >= 12) in their input order.  The fused calls (cd_crop_voxel, cd_ground_plane,
 * cd_process_*) apply the two crops of the launch files inside their first kernel; this entry is for a PassThrough that
 * stands alone. */
int cd_passthrough(cd_context* ctx, const void* points, size_t stride_bytes, int n, int field, double limit_min,
                   double limit_max, int negative, void* out_points, int capacity, int* out_n);

/* S2: seg.segment(*inliers, *coefficients) (gps.cpp:93).  inliers ascending. */
int cd_segment_plane(cd_context* ctx, const void* xyz, size_t stride_bytes, int n,
                     const cd_params* prm, float coeff[4], int32_t* inliers, int capacity,
                     int* out_n_inliers, int* out_iterations);

/* surface_normal_estimation.cpp:167-234, the whole callback: plane 0 = SACMODEL_PERPENDICULAR_PLANE to
 * `table_normal` (the plane parallel to the table top), planes 1,2 = SACMODEL_PARALLEL_PLANE, each fitted
 * (prm: distance threshold, iterations, optimize; eps 0.1 rad) on what the previous fit left over
 * (getNormal, :105-165; `invert` = the node's parameter of that name), then sorted by size, made
 * right-handed and assembled into the pose that the node broadcasts as estimated_cuboid_frame.
 * Returns CD_ERR_NO_MODEL when one of the three fits finds no plane (the reference reads an empty
 * coefficient vector there). */
int cd_surface_frame(cd_context* ctx, const void* xyz, size_t stride_bytes, int n, const float table_normal[3],
                     int invert, const cd_params* prm, cd_surface_frame_result* out);

/* S3 as a call of its own: pcl::ExtractIndices<PCLPointCloud2> (gps.cpp:96-101, opd.cpp:320-326; the fused calls extract
 * internally).  negative == 0: the records at `indices`, in the order of the list; negative != 0: the records whose index
 * is NOT in the list, in their original order (what both launch files use: setNegative(invert = true)).  Records are copied
 * whole - every field of the PointCloud2 blob - `stride_bytes` (a multiple of 4) each.  out_points holds `capacity` records. */
int cd_extract(cd_context* ctx, const void* points, size_t stride_bytes, int n, const int32_t* indices, int n_indices,
               int negative, void* out_points, int capacity, int* out_n);

/* bbox_filter: indices (ascending) of the points whose projection lies strictly inside the image
 * rectangle - cuboid_detection/src/bbox_filter.cpp:30-51 (within_bbox) and :84-103 (pcl_cb builds the
 * inlier list that its ExtractIndices keeps).  P = CameraInfo.P, row-major 3x4; rect = x1,y1,x2,y2.
 * The same test runs fused in cd_process_batch when cd_params.bbox_enable is set. */
int cd_bbox_filter(cd_context* ctx, const void* xyz, size_t stride, int n, const double P[12], const int32_t rect[4],
                   int32_t* out_indices, int capacity, int* out_n);

/* S5: ec.extract(cluster_indices) (opd.cpp:362).  labels[i] = cluster rank (0 = largest,
 * ties -> smaller first member index) or -1; sizes[k] for k < min(K, sizes_capacity). */
int cd_cluster(cd_context* ctx, const void* xyz, size_t stride_bytes, int n,
               const cd_params* prm, int32_t* labels, int32_t* sizes, int sizes_capacity,
               int* out_k);

/* S6: icp.align + getFinalTransformation + hasConverged + getFitnessScore
 * (icp.cpp:170-182).  aligned (may be NULL) receives n*3 floats. */
int cd_icp(cd_context* ctx, int slot, const void* src_xyz, size_t stride_bytes, int n,
           const cd_params* prm, cd_cluster_result* out, float* aligned);

/* Whole chain, one call per batch: what opd.cpp:270-413 does per frame (this is what the
 * frames/s metric times).  `frames` holds n_frames * points_per_frame records.
 * plane_inliers / labels (may be NULL) receive per frame `points_per_frame` int32 slots:
 * the refined plane inlier indices (into the voxel cloud, ascending, -1 padded) and the
 * cluster label of every point of the extracted cloud (-1 padded). */
int cd_process_batch(cd_context* ctx, const void* frames, size_t stride_bytes,
                     int points_per_frame, int n_frames, const cd_params* prm,
                     cd_frame_result* results, int32_t* plane_inliers, int32_t* labels);

/* One frame: the body of the reference's callback / service handler (gps.cpp:43-112 followed by icp.cpp:150-182, or
 * opd.cpp:270-441) as one call - cd_process_batch with n_frames = 1 and `n` points. */
int cd_process_frame(cd_context* ctx, const void* points, size_t stride_bytes, int n, const cd_params* prm,
                     cd_frame_result* result, int32_t* plane_inliers, int32_t* labels);

/* opd.cpp:376-413 runs ICP on EVERY cluster of the frame and picks among all of them (:416-423).  The fixed-size record
 * carries the CD_MAX_CLUSTERS_PER_FRAME largest; when a frame has more (flags & CD_FRAME_MORE_CLUSTERS) the others - all of
 * them were registered as well - are read here.  Copies the results of clusters [first, first + capacity) of `frame`
 * (rank order: size descending) of the LAST cd_process_batch* call of this context into `out`; *out_total (may be NULL)
 * receives the number of clusters of that frame.  Returns the number of results copied, or a negative cd_status. */
int cd_get_cluster_results(const cd_context* ctx, int frame, int first, int capacity, cd_cluster_result* out,
                           int* out_total);

/* What the reference's nodes publish besides poses are CLOUDS of the frame they just processed; after a fused call those
 * are still resident on the device and are read back here (last cd_process_batch* / cd_process_frame call of this context;
 * any other compute call of the context invalidates them: CD_ERR_INVALID_ARG).
 *
 * cd_get_frame_cloud: which = CD_CLOUD_VOXELS, the VoxelGrid output (gps.cpp:73, opd.cpp:298); CD_CLOUD_OBJECTS, what is left
 * after the plane has been taken out (ExtractIndices, gps.cpp:96-101) and the second z crop - the cloud object_pose_detection
 * publishes on its <output> topic (opd.cpp:331-343).  Records are written in a PointCloud2 layout of the caller's choice:
 * `stride_bytes` (a multiple of 4, >= 12) per point, x,y,z float32 at byte offsets 0/4/8, the averaged packed colour at
 * rgb_offset (-1: none; a multiple of 4, >= 12), every other byte zero - with the INPUT's point_step and rgb offset this is
 * what fromPCL(ExtractIndices<PCLPointCloud2>(VoxelGrid<PCLPointCloud2>(input))) hands to the publisher. */
enum { CD_CLOUD_VOXELS = 0, CD_CLOUD_OBJECTS = 1 };
int cd_get_frame_cloud(cd_context* ctx, int frame, int which, void* out_records, size_t stride_bytes, int rgb_offset,
                       int capacity, int* out_n);

/* The points of cluster k (rank order, as in cd_get_cluster_results) of `frame`: aligned == 0 the cluster as extracted
 * (ExtractIndices of opd.cpp:378-387), aligned != 0 the cloud icp.align(output) returned for it (opd.cpp:228, the cloud
 * behind /icp/registered_pcl, opd.cpp:259-262; icp.cpp:178,193 -> /icp/aligned_points).  Records of `stride_bytes`
 * (multiple of 4, >= 12): x,y,z at 0/4/8; with stride_bytes >= 16 the fourth word is 1.0f, which is what pcl::PointXYZ holds
 * there and pcl::toROSMsg(PointCloud<PointXYZ>) puts on the wire (point_step 16); further bytes zero.
 * (When several templates were matched in separate passes and the best one was not the last, the aligned cloud is
 * final_transformation * cluster evaluated once, not the iterated cloud - equal up to float32 rounding.) */
int cd_get_cluster_points(cd_context* ctx, int frame, int k, int aligned, void* out_points, size_t stride_bytes,
                          int capacity, int* out_n);

/* The body of ground_plane_segmentation's callback (gps.cpp:43-112) as ONE call: two PassThrough filters, VoxelGrid,
 * SACSegmentation, ExtractIndices - one upload of the PointCloud2 blob, one download of the cloud to publish.
 * out_records receives the kept voxels in the INPUT's record layout (stride_bytes per record, centroid at 0/4/8, averaged
 * colour at prm->rgb_offset, other bytes zero), *out_n their number, coeff the refined plane, *out_n_inliers the size of
 * the refined inlier set.  prm->crop2_enable / bbox_enable apply as in the fused chain (the gps node sets both 0).
 * When RANSAC finds no plane the status is CD_ERR_NO_MODEL, coeff is untouched and the records are what PCL's
 * ExtractIndices yields for an empty index list (negative: every voxel; otherwise none), as gps.cpp:93-107 publishes. */
int cd_ground_plane(cd_context* ctx, const void* points, size_t stride_bytes, int n, const cd_params* prm, float coeff[4],
                    void* out_records, int capacity, int* out_n, int* out_n_inliers);

/* Per-frame initial guesses for cd_params.icp_use_guess == CD_GUESS_PER_FRAME: n_frames row-major 4x4 float32 matrices
 * (scene -> template), frame f of the following cd_process_batch* calls uses guesses[16 f .. 16 f + 15] for each of its
 * clusters (the granularity of the reference: one sne pose per frame, icp.cpp:130-134).  n_frames = 0 clears them. */
int cd_set_frame_guesses(cd_context* ctx, const float* guesses, int n_frames);

/* Same, input already resident in device memory (HBM) of the context's GPU.  A context works on its own
 * non-blocking HIP stream: there is no implicit ordering against the NULL stream or any other stream, so the
 * caller must have completed its writes to d_frames (e.g. hipStreamSynchronize on the producing stream)
 * before the call.  The call itself is synchronous: results are final when it returns. */
int cd_process_batch_device(cd_context* ctx, const void* d_frames, size_t stride_bytes,
                            int points_per_frame, int n_frames, const cd_params* prm,
                            cd_frame_result* results, int32_t* plane_inliers,
                            int32_t* labels);

/* S7 helpers: tf::Matrix3x3::getRotation + position (icp.cpp:55-82) and the 8 bbox
 * corners in the order of icp.cpp:99-106 transformed by pose.cast<float>() (icp.cpp:110). */
void cd_pose_to_position_quaternion(const double pose[16], double position[3],
                                    double quat_xyzw[4]);
void cd_bbox_corners(const double pose[16], double length, double width, double height,
                     float corners_xyz[24]);

/* Timing of the last cd_process_batch* call, milliseconds per stage measured with HIP
 * events on the context's stream: [0] crop+voxel, [1] plane, [2] extract+cluster,
 * [3] icp, [4] total device time.  Also the ICP kernel's launch count and summed time. */
typedef struct cd_timing {
    float stage_ms[5];
    float icp_kernel_ms;
    int32_t icp_kernel_launches;
    int32_t icp_pair_tests_lo, icp_pair_tests_hi; /* 64-bit count of point-pair distance tests */
    int32_t icp_persist_gave_up;                  /* single-launch ICPs of this call that gave up at a grid barrier and were redone by the
                                                   * multi-launch loop (same results, tens of ms slower): 0 in a healthy run           */
    int64_t algorithmic_bytes;                    /* B_alg of SURVEY 8(d) for this batch */
    int64_t icp_algorithmic_bytes;                /* the S6 term of B_alg: sum 12*M + 12*N_s*(I_c+1) */
    int32_t scan_retries;                         /* 1 when this call was redone because a chained scan reported a stall (cannot happen on
                                                   * its own since the scans take their tile ids from atomic tickets; the redo runs with
                                                   * the device to itself): 0 in a healthy run                                          */
    int32_t icp_regime;                           /* launch shape of the whole-cluster ICP kernel of this call: (clusters in flight per
                                                   * workgroup << 16) | workgroups; 0 = no such launch (sliced driver).  The shape is
                                                   * chosen from the calls in flight on the device (scheduling only: results do not
                                                   * depend on it), so a measurement can say which shape it measured                    */
    int32_t icp_handovers;                        /* running clusters that changed workgroup inside that launch (a call that has the GPU to
                                                   * itself lets workgroups without work take over clusters from those that still have
                                                   * several: scheduling only, results do not depend on it)                             */
    int32_t icp_search;                           /* which nearest-neighbour search the ICPs of this call ran: 0 the pruned searches over an
                                                   * arbitrary template, 1 the closed form for a template that is a union of axis-aligned
                                                   * lattices (every make_cuboid.py template; cd_template_lattice_faces), 2 both (mixed batch).
                                                   * Results do not depend on it                                                            */
    int32_t icp_handover_lost;                    /* a cluster in hand-over between two workgroups was claimed and never arrived, or a waiting
                                                   * workgroup ran out of polls: the call has FAILED with CD_ERR_DEVICE (its records are not
                                                   * complete).  Cannot happen in a healthy launch; 0 otherwise                              */
    float icp_wave_ms;                            /* lattice ICP launches: sum over the launch's workgroups of (lifetime x waves), in
                                                   * wave-milliseconds - what the batch's ICP held of the chip's wave slots (256 CUs x 16
                                                   * waves at this kernel's register count); with batches in flight this, not the launch's
                                                   * duration, is what a batch's ICP costs.  0 for the other ICP drivers                     */
} cd_timing;
int cd_get_timing(const cd_context* ctx, cd_timing* out);

#ifdef __cplusplus
}
#endif
#endif /* CUBOID_HIP_H */
