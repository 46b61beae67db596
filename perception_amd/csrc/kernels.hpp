// kernels.hpp - host-callable launchers of every HIP kernel (one .hip file per stage).
#pragma once
#include "common.hpp"

namespace cd {

// k_voxel.hip
void launch_crop_count(hipStream_t s, const void* in, size_t stride, int N, int F, int rgb_off, CropLimits lim, int T,
                       FrameState* fs, int* tile_cnt);
struct ScanJob { int* counts; int* totals; int* mirror; };   // exclusive scan of counts[row][0..T) in place, row totals to totals / mirror[row * pitch]
void launch_scan_tiles(hipStream_t s, int* counts, int rows, int T, int* totals, int total_pitch);
void launch_scan_tiles2(hipStream_t s, const ScanJob& a, const ScanJob& b, int rows, int T, int total_pitch);   // two scans, one launch
void launch_voxel_setup(hipStream_t s, FrameState* fs, int F, float leaf, const uint32_t* ghist, FrameState* mirror);   // (mirror: device-visible host copy of fs, or nullptr)
void launch_crop_fused(hipStream_t s, const void* in, size_t stride, int N, int pitch, int F, int rgb_off, CropLimits lim,
                       int T, float leaf, KeyPack kp, FrameState* fs, int* state, float4* cpt, uint32_t* keys, int* ticket);
void launch_crop_runs(hipStream_t s, const void* in, size_t stride, int N, int pitch, int F, int rgb_off, CropLimits lim,
                      int T, float leaf, KeyPack kp, FrameState* fs, unsigned long long* state, float4* cpt, uint32_t* rkeys,
                      uint32_t* rvals, uint32_t* ghist, int* ticket, int direct);
void launch_crop_compact(hipStream_t s, const void* in, size_t stride, int N, int pitch, int F, int rgb_off, CropLimits lim,
                         int T, float leaf, const FrameState* fs, const int* tile_off, float4* cpt, uint32_t* keys);
void launch_voxel_centroid(hipStream_t s, const uint32_t* keys, const uint32_t* vals, const float4* cpt, int N, int F,
                           int T, int Tact, int rgb_on, FrameState* fs, int* state, float4* vox, int* ticket);
void launch_voxel_centroid_runs(hipStream_t s, const uint32_t* keys, const uint32_t* vals, const float4* cpt, int N, int F,
                                int T, int Tact, int rgb_on, FrameState* fs, int* state, float4* vox, int* ticket, int lanes, int pts_pitch);

void launch_mark_indices(hipStream_t s, const int* idx, int m, int n, int* flag);
struct ZeroRegions { int n; uint32_t* ptr[16]; size_t words[16]; FrameState* fs; int nfs; };   // fs != nullptr: nfs FrameStates set to their initial value (zero, mn = +max) as one more region
void launch_zero_regions(hipStream_t s, const ZeroRegions& r);
struct CopySeg { uint32_t* dst; const uint32_t* src; size_t dpitch_w, spitch_w; int width_w, rows; };   // pitches and width in 4-byte words
struct CopyList { int n; CopySeg seg[8]; };
void launch_copy_list(hipStream_t s, const CopyList& L);
void launch_copy_rows(hipStream_t s, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, int rows);
void launch_passthrough_mark(hipStream_t s, const void* in, size_t stride, int n, int field_off, double lo, double hi, int negative, int* flag);
void launch_select_unmarked(hipStream_t s, const int* flag, int n, int* state, FrameState* fs, int* out, int* ticket);
void launch_gather_records(hipStream_t s, const void* in, int words, const int* idx, int m, void* out);
void launch_pack_records(hipStream_t s, const float4* pts, int m, int words, int rgb_word, uint32_t pad3, void* out);

// k_sort.hip : segmented (per frame) stable LSD radix sort of (key, value) pairs, all passes
constexpr int SORT_MAX_PASSES_HOST = 4;
int launch_radix_sort(hipStream_t s, uint32_t* const key[2], uint32_t* const val[2], int N, int F, int Tact, int npass,
                      FrameState* fs, uint32_t* ghist, int* state, KeyPack kp, int* ticket);
// the same sort over RUNS of equal voxel index (k_voxel_runs: 2-3 x fewer elements on organised clouds); tile_state: [F][T] ints
// the scatters alone, over run records whose digit histograms are already in ghist (k_crop_runs): `digits` lists the 8-bit digits
// of the key that vary (ascending); the records are in key[0] / val[0]
int launch_radix_scatter_runs(hipStream_t s, uint32_t* const key[2], uint32_t* const val[2], int N, int F, int Tact, const int* digits, int ndigits,
                              FrameState* fs, const uint32_t* ghist, int* state, int* ticket);
int launch_radix_sort_runs(hipStream_t s, uint32_t* const key[2], uint32_t* const val[2], int N, int F, int T, int Tact, int npass,
                           FrameState* fs, uint32_t* ghist, int* state, int* tile_state, KeyPack kp, int* ticket);

// k_plane.hip
void launch_ransac_sample(hipStream_t s, const float4* vox, int N, int F, FrameState* fs, const int* rnd_table,
                          int h_target, const int* active, float4* models, int* valid);
void launch_ransac_count(hipStream_t s, const float4* vox, int N, int F, int Tact, const FrameState* fs,
                         const float4* models, const int* valid, const int* active, int h0, int h1, float thr, int* counts);
void launch_plane_cov(hipStream_t s, const float4* vox, int N, int F, int Tact, const FrameState* fs, const float4* model,
                      const int* have, float thr, unsigned long long* sums);
void launch_plane_flag_count(hipStream_t s, const float4* vox, int N, int F, int T, int Tact, const FrameState* fs,
                             const float4* model, const int* have, float thr, int negative, int crop2, float z2lo,
                             float z2hi, const BBoxGate& gate, int* cnt_plane, int* cnt_obj);
void launch_extract_scatter(hipStream_t s, const float4* vox, int N, int F, int T, int Tact, const FrameState* fs,
                            const float4* model, const int* have, float thr, int negative, int crop2, float z2lo,
                            float z2hi, const BBoxGate& gate, const int* off_plane, const int* off_obj, int* plane_idx, float4* obj);

// k_cluster.hip
void launch_cluster_lds(hipStream_t s, const float4* obj, int N, int F, FrameState* fs, float inv_cell, float r2,
                        int* parent, int* csize, int* rank_of_root);
void launch_cluster_cells(hipStream_t s, const float4* obj, int N, int F, FrameState* fs, float inv_cell, float r2,
                          int* parent, int* csize, int* rank_of_root, float4* sorted);
void launch_cluster_build(hipStream_t s, const float4* obj, int N, int F, int Tact, const FrameState* fs, float inv_cell,
                          int* head, int* next, int* parent, int* csize, int* rank_of_root);
void launch_cluster_hook(hipStream_t s, const float4* obj, int N, int F, int Tact, const FrameState* fs, float inv_cell,
                         float r2, const int* head, const int* next, int* parent);
void launch_cluster_flatten(hipStream_t s, int N, int F, int Tact, const FrameState* fs, int* parent, int* csize);
void launch_cluster_rank(hipStream_t s, int N, int F, FrameState* fs, int enable, int min_sz, int max_sz,
                         const int* parent, const int* csize, int* cand, int* rank_of_root, int* sizes_sorted, FrameState* mirror);
void launch_label_count(hipStream_t s, int N, int F, int T, int Tact, const FrameState* fs, int enable, const int* parent,
                        const int* rank_of_root, int* label, int* tile_cnt, int kbase);
void launch_label_scatter(hipStream_t s, const float4* obj, int N, int F, int T, int Tact, const FrameState* fs,
                          const int* label, const int* tile_off, float4* src0, float4* src, int kbase, const int* koff_tab);

// k_icp.hip
void launch_icp_iter(hipStream_t s, int it, int n_work, int ncl, const IcpWork* work, const IcpCluster* cl, IcpState* st,
                     unsigned long long* acc, const float4* tpl, const float4* tlo, const float4* thi, const IcpGrid* grids,
                     float4* src, int* nn, float* d2buf, int qslice, int* queue, int n_cu, IcpParams prm);
void launch_icp_persist(hipStream_t s, int n_work, int n_wg, int max_it, const IcpWork* work, const IcpCluster* cl, IcpState* st,
                         unsigned long long* acc, unsigned long long* accf, const float4* tpl, const float4* tlo, const float4* thi,
                         const IcpGrid* grids, float4* src, const float4* src0, int* nn, float* d2buf, int qslice, unsigned* bar,
                         int* abort_flag, int n_open, int* closed, IcpParams prm);
void launch_icp_fitness(hipStream_t s, int n_work, const IcpWork* work, const IcpCluster* cl, const IcpState* st,
                        int parity, unsigned long long* accf, const float4* tpl, const float4* tlo, const float4* thi,
                        const IcpGrid* grids, const float4* src0, int* nn, float* d2buf, int qslice);

void launch_icp_apply_guess(hipStream_t s, int ncl, int max_n, const IcpCluster* cl, const float* guesses, int per_frame,
                            const float4* src0, float4* src);
void launch_icp_pipe(hipStream_t s, int ncl, const int* order, const IcpCluster* cl, IcpState* st, unsigned long long* accf,
                     const float4* tpl, const float4* tlo, const float4* thi, const unsigned short* kdmap, const IcpGrid* grids,
                     const unsigned short* tcell, float4* src, const float4* src0, int* nn,
                     int* queue, int n_wg, const int* wgtab, IcpParams prm);
void launch_icp_pipe_big(hipStream_t s, int ncl, const int* order, const IcpCluster* cl, IcpState* st, unsigned long long* accf,
                         const float4* tpl, const float4* tplk, const float4* tlok, const float4* thik, const unsigned short* kdmap,
                         const IcpGrid* grids, const IcpSuper* supers, const unsigned short* tcell, float4* src, const float4* src0,
                         int* nn, int* queue, int n_wg, const int* wgtab, IcpParams prm);
void launch_icp_cluster(hipStream_t s, int ncl, const int* order, const IcpCluster* cl, IcpState* st, unsigned long long* accf,
                        const float4* tpl, const float4* tlo, const float4* thi, const IcpGrid* grids,
                        const unsigned short* tcell, float4* src, const float4* src0, int* nn,
                        int* queue, int n_cu, IcpParams prm);

// k_icp_lat.hip : templates that are unions of axis-aligned lattices (IcpLattice) - closed-form nearest neighbour
void launch_icp_lat(hipStream_t s, int nitems, int cpw, int wpc, int n_wg, const int* order, const IcpCluster* cl, IcpState* st,
                    unsigned long long* accf, const IcpLattice* lats, float4* src, const float4* src0, int* queue, unsigned long long* busy,
                    unsigned long long* busy_out, IcpParams prm);
void launch_lat_nn(hipStream_t s, const IcpLattice* lat, const float4* q, int n, int* out_idx, float* out_d2);

}  // namespace cd
