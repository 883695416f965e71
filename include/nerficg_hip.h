/*
 * nerficg_hip.h -- C ABI of libnerficg_hip.so, the MI355X (gfx950) replacement for the native ops behind
 * nerficg's src/Methods plugins.  Plain pointers + sizes, no torch types.  All pointers are DEVICE pointers unless a
 * parameter is documented as host.  `stream` is a hipStream_t passed as void* (NULL = the null stream).
 * Every function returns NRC_OK (0) or a negative NRC_ERR_* code; nothing is allocated and nothing synchronises
 * inside the library (workspaces are caller-provided), so every entry point is hipGraph-capturable.
 *
 * Each group cites the reference interface it replaces (paths relative to the nerficg repository root).
 */
#ifndef NERFICG_HIP_H
#define NERFICG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nrc_stream_t;

enum {
    NRC_OK = 0,
    NRC_ERR_INVALID = -1, /* bad argument (null pointer, negative size, unsupported configuration) */
    NRC_ERR_LAUNCH = -2,  /* hipGetLastError() != hipSuccess after a launch */
    NRC_ERR_UNSUPPORTED = -3
};

/* Version of THIS header's contract (signatures, buffer sizes, meaning of the counters).  Bumped on every incompatible change; a caller
 * compares nrc_abi_version() of the library it loaded with the NRC_ABI_VERSION it was built against and refuses a mismatch (a stale .so under
 * newer bindings -- or the reverse -- would pass a stream where a pointer is expected, or under-allocate a workspace).
 * History: 1 = rounds 1-2; 2 = round 3 (nrc_gs_backward gained grad_records, nrc_ngp_render_count writes 2 * n_tiles ints into tile_rows,
 * counter[1] = total samples, save buffers padded to nrc_nwie_save_rows); 3 = round 4 (see the notes at the changed entry points);
 * 4 = round 4, later: nrc_gs_preprocess and nrc_ngp_render_count gained count_mailbox / mailbox_ticket, nrc_host_mailbox_alloc / _free are new;
 * nrc_ngp_query_samples gained arena_tile_off / arena_rows, nrc_ngp_composite_image arena_rows, nrc_ngp_render_write accepts ts = NULL;
 * nrc_photometric_loss_* are new; 5 = round 5: group 13 (the fused InstantNGP training iteration) is new; nrc_ngp_train_query_forward gained
 * n_samples_dev (NULL = every row, as before). */
#define NRC_ABI_VERSION 6
/* library identification; also used by the loader's symbol check */
int nrc_abi_version(void);
const char* nrc_build_info(void);
/* text of the HIP error behind the calling thread's most recent NRC_ERR_LAUNCH */
const char* nrc_last_error(void);

/* =====================================================================================================
 * Group 1 -- VolumeRenderingV2  (replaces the pybind module of
 *            src/Methods/InstantNGP/VolumeRenderingV2/csrc/binding.cpp:234-250)
 * dtypes are fixed like in the reference: f32 data, i64 rays_a / alive_indices / hits_voxel_idx,
 * i32 counter / N_eff_samples / Morton codes, u8 bitfield.  All arrays dense row-major.
 * ===================================================================================================== */

/* binding.cpp:4-16 -> intersection.cu:59-100.  hits_t (n_rays,max_hits,2) and hits_voxel_idx (n_rays,max_hits) are
 * fully written (-1 fill, then hits, then sorted ascending by t1 like the reference's torch::sort). hit_cnt (n_rays). */
int nrc_ray_aabb_intersect(const float* rays_o, const float* rays_d, const float* centers, const float* half_sizes,
                           int64_t n_rays, int64_t n_voxels, int32_t max_hits, int32_t* hit_cnt, float* hits_t,
                           int64_t* hits_voxel_idx, nrc_stream_t stream);
/* binding.cpp:19-31 -> intersection.cu:156-196 */
int nrc_ray_sphere_intersect(const float* rays_o, const float* rays_d, const float* centers, const float* radii,
                             int64_t n_rays, int64_t n_spheres, int32_t max_hits, int32_t* hit_cnt, float* hits_t,
                             int64_t* hits_sphere_idx, nrc_stream_t stream);
/* binding.cpp:46-50 -> raymarching.cu:72-88.  coords (n,3) i32 -> indices (n) i32 */
int nrc_morton3D(const int32_t* coords, int64_t n, int32_t* indices, nrc_stream_t stream);
/* binding.cpp:53-57 -> raymarching.cu:103-119 */
int nrc_morton3D_invert(const int32_t* indices, int64_t n, int32_t* coords, nrc_stream_t stream);
/* binding.cpp:34-43 -> raymarching.cu:143-161.  grid_dtype: 0 = f32, 1 = f16.  n_bytes = len(bitfield); grid has 8*n_bytes cells. */
int nrc_packbits(const void* density_grid, int32_t grid_dtype, int64_t n_bytes, float density_threshold,
                 uint8_t* density_bitfield, nrc_stream_t stream);

/* binding.cpp:60-81 -> raymarching.cu:283-332, split in two calls so that the caller can allocate exactly
 * counter[0] sample rows instead of the reference's n_rays*max_samples zero-filled rows.
 *   _count : marches every ray once, writes rays_a (n_rays,3) = (ray_idx, start_idx, n_samples) in RAY ORDER with
 *            start_idx = exclusive prefix sum (deterministic; the reference's order is atomic-arrival order) and
 *            counter[0] = total samples, counter[1] = n_rays.  workspace: nrc_raymarching_train_ws_bytes(n_rays, max_samples) bytes;
 *            for batches of <= 32768 rays the pass also parks every sample's t there (max_samples f32 per ray).
 *   _write : writes xyzs/dirs (total,3), deltas/ts (total): with the workspace of the count pass it expands the parked positions
 *            (no second march); with workspace == NULL it marches again. */
int64_t nrc_raymarching_train_ws_bytes(int64_t n_rays, int32_t max_samples);
int nrc_raymarching_train_count(const float* rays_o, const float* rays_d, const float* hits_t,
                                const uint8_t* density_bitfield, int32_t cascades, float scale, float exp_step_factor,
                                const float* noise, int32_t grid_size, int32_t max_samples, int64_t n_rays,
                                int64_t* rays_a, int32_t* counter, void* workspace, nrc_stream_t stream);
/* nrc_raymarching_train_count for batches of at most 32 768 rays as TWO launches (wave-per-ray march + one workgroup that scans the counts: no block
 * sums to clear, no separate assignment pass), whose scan also stores (total samples, n_rays, mailbox_ticket) in HOST memory from
 * nrc_host_mailbox_alloc (group 4) -- the caller polls the ticket and sizes xyzs / dirs / deltas / ts without a device-to-host copy or a stream wait
 * (the one host read of the reference's training march, custom_functions.py:112-119).  Same rays_a / counter / parked positions as the plain call. */
int nrc_raymarching_train_count_posted(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* density_bitfield, int32_t cascades,
                                       float scale, float exp_step_factor, const float* noise, int32_t grid_size, int32_t max_samples, int64_t n_rays,
                                       int64_t* rays_a, int32_t* counter, void* workspace, int64_t* count_mailbox, int64_t mailbox_ticket,
                                       nrc_stream_t stream);
int nrc_raymarching_train_write(const float* rays_o, const float* rays_d, const float* hits_t,
                                const uint8_t* density_bitfield, int32_t cascades, float scale, float exp_step_factor,
                                const float* noise, int32_t grid_size, int32_t max_samples, int64_t n_rays,
                                const int64_t* rays_a, float* xyzs, float* dirs, float* deltas, float* ts,
                                const void* workspace, nrc_stream_t stream);
/* The ray preparation of InstantNGPRenderer.render_rays (Renderer.py:55-77: origin - centre, ray_aabb_intersect against the scene box,
 * clamp of the interval to [near, far]) as ONE launch: origin_centred (n,3), spans (n,2) = (t_in, t_out), a miss is (near, -1).
 * center3 / half3: HOST float[3].  Same values as nrc_ray_aabb_intersect + the two torch clamps. */
int nrc_ngp_clip_rays(int64_t n_rays, const float* origin, const float* dirs, const float* center3, const float* half3,
                      float near_plane, float far_plane, float* origin_centred, float* spans, nrc_stream_t stream);
/* What render_rays_training does with the composited sums (Renderer.py:80-84), one launch each way: rgb_out = rgb + (1 - opacity) * bg,
 * depth_out = depth / (opacity + 1e-6); bg_dev: DEVICE float[3].  _bw turns the gradients of (rgb_out, opacity as alpha, depth_out) --
 * each may be NULL -- into dL_dopacity / dL_ddepth for nrc_composite_train_bw (dL_drgb is g_rgb itself).  nrc_composite_train_bw accepts
 * NULL for dL_dopacity, dL_ddepth and dL_dws (no gradient reached that output). */
int nrc_ngp_train_pixels_fw(int64_t n_rays, const float* opacity, const float* depth, const float* rgb, const float* bg_dev,
                            float* rgb_out, float* depth_out, nrc_stream_t stream);
int nrc_ngp_train_pixels_bw(int64_t n_rays, const float* g_rgb, const float* g_alpha, const float* g_depth, const float* opacity,
                            const float* depth, const float* bg_dev, float* dL_dopacity, float* dL_ddepth, nrc_stream_t stream);
/* Fixed-capacity variant for graph capture (no host read of counter[0] between _count and _write): call between the two passes with
 * sample buffers of `sample_capacity` rows.  Rays whose segment would cross the capacity keep the samples that fit (n_samples in rays_a is
 * cut, start_idx <= sample_capacity), rows [min(counter[0], capacity), capacity) are filled with inert samples (position = box centre,
 * direction +z, delta = t = 0) that no ray references.  counter[0] keeps the uncut total: counter[0] > sample_capacity <=> samples were
 * dropped.  The reference has no counterpart: it allocates n_rays*max_samples rows and slices by counter[0] on the host
 * (custom_functions.py:112-119). */
int nrc_raymarching_train_cap(int64_t n_rays, int64_t sample_capacity, const int32_t* counter, int64_t* rays_a, float* xyzs,
                              float* dirs, float* deltas, float* ts, nrc_stream_t stream);
/* Same, and *overflow (device int64, may be NULL) = max(counter[0] - sample_capacity, 0): the number of dropped samples without two more
 * element-wise launches in a recorded iteration. */
int nrc_raymarching_train_cap_overflow(int64_t n_rays, int64_t sample_capacity, const int32_t* counter, int64_t* rays_a, float* xyzs,
                                       float* dirs, float* deltas, float* ts, int64_t* overflow, nrc_stream_t stream);
/* count + cap + write for a batch of at most 32 768 rays with a fixed sample capacity, as THREE launches (march with parked positions, one
 * workgroup that scans the per-ray counts and writes the cut rays_a / counter / *overflow, expansion of the parked positions + inert tail):
 * what nrc_raymarching_train_count, _cap_overflow and _write do in seven.  workspace: nrc_raymarching_train_ws_bytes(n_rays, max_samples). */
int nrc_raymarching_train_capped(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* density_bitfield, int32_t cascades,
                                 float scale, float exp_step_factor, const float* noise, int32_t grid_size, int32_t max_samples, int64_t n_rays,
                                 int64_t sample_capacity, int64_t* rays_a, int32_t* counter, float* xyzs, float* dirs, float* deltas, float* ts,
                                 int64_t* overflow, void* workspace, nrc_stream_t stream);
/* The batch of a training iteration out of the resident ray pool (RayPoolSampler.get, src/Optim/Samplers/DatasetSamplers.py:53-66:
 * ray_pool[indices], one fancy-index gather per field) as ONE launch: rows ids[i] of up to four pools of the same length -- three of
 * row width 3 (origin, view direction, rgb; any may be NULL) and one of width 1 (alpha; may be NULL) -- into dense outputs.
 * ids: int64, -n_pool <= ids[i] < n_pool; a negative id counts from the end like torch's indexing; an id out of range (torch: device
 * assert) writes a NaN row, so a broken sampler shows up in the loss / the GradScaler's check instead of training on black rays. */
int nrc_gather_ray_batch(const int64_t* ids, int64_t n, int64_t n_pool, const float* pool_a3, const float* pool_b3, const float* pool_c3,
                         const float* pool_d1, float* out_a3, float* out_b3, float* out_c3, float* out_d1, nrc_stream_t stream);
/* The colour term of the InstantNGP loss with the GradScaler's multiplication folded in (Trainer.py:87-89: mse_loss(rgb, target),
 * scaler.scale(loss)): out2[0] = mean((pred - target)^2) over n values, out2[1] = out2[0] * *scale (scale: DEVICE float, NULL = 1).
 * One workgroup, fixed summation order (the result does not depend on the run).  _backward: grad_pred = 2 / n * (pred - target) *
 * (g_loss + g_scaled * scale); g_loss / g_scaled: DEVICE floats, either may be NULL (no gradient reached that output).  Replaces
 * seven element-wise / reduction launches of the op-by-op expression in a recorded iteration.  n <= 2^24. */
int nrc_mse_scaled_forward(int64_t n, const float* pred, const float* target, const float* scale, float* out2, nrc_stream_t stream);
int nrc_mse_scaled_backward(int64_t n, const float* pred, const float* target, const float* scale, const float* g_loss,
                            const float* g_scaled, float* grad_pred, nrc_stream_t stream);
/* binding.cpp:84-106 -> raymarching.cu:407-454.  hits_t (n_total_rays,2) is advanced in place.  Outputs
 * (n_alive,N_samples[,3]) are fully written (zero beyond N_eff_samples), N_eff_samples (n_alive) i32. */
int nrc_raymarching_test(const float* rays_o, const float* rays_d, float* hits_t, const int64_t* alive_indices,
                         int64_t n_alive, const uint8_t* density_bitfield, int32_t cascades, float scale,
                         float exp_step_factor, int32_t grid_size, int32_t max_samples, int32_t N_samples, float* xyzs,
                         float* dirs, float* deltas, float* ts, int32_t* N_eff_samples, nrc_stream_t stream);

/* binding.cpp:109-126 -> volumerendering.cu:48-84.  rays_a (n_rays,3) i64; sample arrays have n_samples rows.
 * Outputs are fully written: total_samples (n_rays) i64, opacity/depth (n_rays), rgb (n_rays,3), ws (n_samples). */
int nrc_composite_train_fw(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                           const int64_t* rays_a, int64_t n_rays, int64_t n_samples, float T_threshold,
                           int64_t* total_samples, float* opacity, float* depth, float* rgb, float* ws,
                           nrc_stream_t stream);
/* binding.cpp:129-163 -> volumerendering.cu:154-202 */
int nrc_composite_train_bw(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb, const float* dL_dws,
                           const float* sigmas, const float* rgbs, const float* ws, const float* deltas, const float* ts,
                           const int64_t* rays_a, const float* opacity, const float* depth, const float* rgb,
                           int64_t n_rays, int64_t n_samples, float T_threshold, float* dL_dsigmas, float* dL_drgbs,
                           nrc_stream_t stream);
/* binding.cpp:166-194 -> volumerendering.cu:252-285.  In place on opacity/depth/rgb (n_total_rays[,3]) and on
 * alive_indices (entries of dead rays become -1). */
int nrc_composite_test_fw(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                          int64_t* alive_indices, int64_t n_alive, int32_t N_samples, float T_threshold,
                          const int32_t* N_eff_samples, float* opacity, float* depth, float* rgb, nrc_stream_t stream);
/* Backward of raymarching_train w.r.t. the rays (the reference does this in Python with torch_scatter.segment_csr,
 * custom_functions.py:122-137): row n of rays_a = (ray, start, count) gets dL_drays_o[n] = sum dL_dxyzs[start:start+count] and
 * dL_drays_d[n] = sum (ts * dL_dxyzs + dL_ddirs)[start:start+count].  dL_ddirs may be NULL (no gradient reached the directions). */
int nrc_raymarching_train_bw(const float* dL_dxyzs, const float* dL_ddirs, const float* ts, const int64_t* rays_a, int64_t n_rays,
                             int64_t n_samples, float* dL_drays_o, float* dL_drays_d, nrc_stream_t stream);
/* binding.cpp:197-209 -> losses.cu:64-109 */
int nrc_distortion_loss_fw(const float* ws, const float* deltas, const float* ts, const int64_t* rays_a, int64_t n_rays,
                           int64_t n_samples, float* loss, float* ws_inclusive_scan, float* wts_inclusive_scan,
                           nrc_stream_t stream);
/* binding.cpp:212-231 -> losses.cu:145-174 */
int nrc_distortion_loss_bw(const float* dL_dloss, const float* ws_inclusive_scan, const float* wts_inclusive_scan,
                           const float* ws, const float* deltas, const float* ts, const int64_t* rays_a, int64_t n_rays,
                           int64_t n_samples, float* dL_dws, nrc_stream_t stream);

/* =====================================================================================================
 * Group 2 -- MortonEncoding._C  (replaces src/CudaUtils/MortonEncoding/MortonEncoding/morton_encoding.cu:48-79)
 * positions (n,3) f32 -> codes (n) i64.  The bounding cube (reference: host aminmax, :54-57) is reduced on device
 * into `workspace` (nrc_morton_encode_ws_bytes(n) bytes) by the same call.
 * ===================================================================================================== */
int64_t nrc_morton_encode_ws_bytes(int64_t n);
int nrc_morton_encode(const float* positions, int64_t n, int64_t* codes, void* workspace, nrc_stream_t stream);

/* =====================================================================================================
 * Group 3 -- tinycudann subset  (replaces the tiny-cuda-nn modules built at src/Methods/InstantNGP/Model.py:58-114 and
 *            queried at src/Methods/InstantNGP/Renderer.py:48-60; external dependency src/Thirdparty/TinyCudaNN.py:10)
 * Networks: input encoding (32 features) -> 64-wide ReLU MLP (n_hidden 1 or 2, no bias) -> 16 padded outputs.
 *   encoding 0: multiresolution hash grid, fp16 table of nrc_grid_layout()[n_levels] entries x F features; F = (encoding >> 8) & 0xff (0 = 2).
 *               16 levels x 2 features (the shipped yaml) runs on the tuned kernels; any other F in {2, 4} with n_levels * F <= 32 -- what
 *               HASHGRID_N_LEVELS / HASHGRID_N_FEATURES_PER_LEVEL of src/Methods/InstantNGP/Model.py:18-29 can ask for within the 32 first-layer
 *               inputs -- on a general kernel (input k = level * F + c, zero behind n_levels * F; backward: nrc_grid_backward_general)
 *   encoding 1: [SphericalHarmonics degree 4 of input dims 0..2 | Identity of input dims 3..18]
 * weights_f16: [W0 (64,32) | hidden (64,64) x (n_hidden-1) | Wout (16,64)] row-major fp16 (rows >= n_out_rows read as 0).
 * ===================================================================================================== */

/* HOST helper: entry offsets of the levels, offsets_host[n_levels+1] (last = total entries); NULL only validates */
int nrc_grid_layout(int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale,
                    uint32_t* offsets_host);
/* backward of the general grid: d_in (M,32) f32 as nrc_nwie_backward writes it with d_in_pair_major = 0; grad_table (entries, F) f32, ACCUMULATED */
int nrc_grid_backward_general(const float* x01, int64_t M, const float* d_in, int32_t n_levels, int32_t n_features, int32_t log2_hashmap_size,
                              int32_t base_resolution, float per_level_scale, float* grad_table, nrc_stream_t stream);
/* fp32 master parameters -> fp16 compute copy */
int nrc_f32_to_f16(const float* src, void* dst_f16, int64_t n, nrc_stream_t stream);
/* NetworkWithInputEncoding.forward.  input: encoding 0 -> (M,3) f32 in [0,1], input_ld ignored; encoding 1 -> (M,input_ld>=19)
 * fp16 rows [d01(3) | features(16)].  out_act: 0 none, 1 sigmoid.  out (M,out_ld) fp16, columns [0,n_store) written
 * (n_store in {4,8,12,16}).  save_in (R,32) fp16 and save_acts (n_hidden,R,64) fp16, R = nrc_nwie_save_rows(M) (M rounded up to whole
 * 32-sample tiles), receive the encoded inputs and the post-ReLU activations for the backward pass (both NULL for inference) in a
 * layout private to the two calls (MFMA-fragment-major, so that both sides move whole cache lines).  Of a network with two hidden layers only the FIRST layer's
 * activations are written (the buffer keeps its documented size): the backward pass recomputes the second from the first with the forward's own eight matrix
 * instructions per 32 samples -- 128 B per sample less in either direction (round 6). */
int64_t nrc_nwie_save_rows(int64_t M);
/* workspace (optional, grid encoding only): nrc_nwie_forward_ws_bytes(M) bytes -> the encoding runs as its own kernel (faster);
 * NULL -> one kernel gathers and runs the MLP. */
int64_t nrc_nwie_forward_ws_bytes(int64_t M);
int nrc_nwie_forward(int32_t encoding, const void* input, int32_t input_ld, int64_t M, const void* weights_f16,
                     const void* table_f16, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution,
                     float per_level_scale, int32_t n_hidden, int32_t out_act, int32_t n_out_rows, void* out_f16,
                     int32_t out_ld, int32_t n_store, void* save_in, void* save_acts, void* workspace, nrc_stream_t stream);
/* NetworkWithInputEncoding.backward through the MLP.  d_out / out: (M,out_ld) fp16 (upstream gradient of, and the forward
 * value of, the stored output columns).  Upstream gradients are multiplied by loss_scale before they are rounded to fp16
 * MFMA operands and every result is divided by it again (tiny-cuda-nn's internal loss scale).  grad_weights: f32, layout
 * of weights_f16, ACCUMULATED atomically (caller zeroes).  d_in f32 = gradient w.r.t. the 32 encoded inputs, laid out (M,32)
 * (d_in_pair_major = 0) or as 16 feature pairs [16][M][2] (d_in_pair_major = 1: the layout nrc_grid_backward reads coalesced). */
int nrc_nwie_backward(int64_t M, const void* weights_f16, int32_t n_hidden, int32_t out_act, int32_t n_out_rows,
                      const void* d_out_f16, const void* out_f16, int32_t out_ld, const void* save_in,
                      const void* save_acts, float loss_scale, float* grad_weights, float* d_in, int32_t d_in_pair_major,
                      nrc_stream_t stream);
/* hash-grid backward: grad_table (entries,2) f32 += trilinear scatter of d_features f32, (M, 2*n_levels) or pair-major
 * [n_levels][M][2] (caller zeroes grad_table).  workspace: optional (NULL allowed), nrc_grid_backward_ws_bytes(...) bytes, 16-byte
 * aligned -- with it (and pair-major gradients, M >= 16384) the hashed levels are split into per-slice record buckets once and
 * accumulated in LDS by one workgroup per slice instead of by atomics */
int64_t nrc_grid_backward_ws_bytes(int64_t M, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution,
                                   float per_level_scale);
int nrc_grid_backward(const float* x01, int64_t M, const float* d_features, int32_t d_features_pair_major, int32_t n_levels,
                      int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale, float* grad_table,
                      void* workspace, nrc_stream_t stream);
/* nrc_grid_backward over the rows that hold samples: n_samples_dev (DEVICE i32[1]) <= M, the capacity the launch and the workspace are sized for
 * (the pair-major layout of d_features keeps following M).  NULL = all M rows. */
int nrc_grid_backward_live(const float* x01, int64_t M, const float* d_features, int32_t d_features_pair_major, int32_t n_levels,
                           int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale, float* grad_table, void* workspace,
                           const int32_t* n_samples_dev, nrc_stream_t stream);
/* query_model (src/Methods/InstantNGP/Renderer.py:48-53) for TRAINING as one forward and one backward call instead of ~55 small
 * launches: world positions xyzs (M,3) / directions dirs (M,3) f32 -> sigmas (M), rgbs (M,3) f32 (what VolumeRenderer consumes), and
 * dL/dsigmas, dL/drgbs -> gradients of both parameter vectors (ACCUMULATED; caller zeroes; layout of the tinycudann modules).
 * Kept between the calls: x01 (M,3) f32, h (M,16) f16, rgb (M,4) f16, save_in_* (R,32) f16, save_acts_d (1,R,64) / save_acts_c (2,R,64) f16,
 * R = nrc_nwie_save_rows(M).
 * forward workspace: nrc_ngp_train_query_ws_bytes(M); backward scratch: nrc_ngp_train_query_scratch_bytes(M). */
int64_t nrc_ngp_train_query_ws_bytes(int64_t M);
int64_t nrc_ngp_train_query_scratch_bytes(int64_t M);
int nrc_ngp_train_query_forward(const float* xyzs, const float* dirs, int64_t M, const float* xyz_min3, const float* xyz_size3,
                                const void* density_weights_f16, const void* color_weights_f16, const void* table_f16,
                                int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale,
                                float* x01, void* h_f16, void* rgb_f16, float* sigmas, float* rgbs, void* save_in_d,
                                void* save_acts_d, void* save_in_c, void* save_acts_c, void* workspace,
                                const int32_t* n_samples_dev, nrc_stream_t stream);
int nrc_ngp_train_query_backward(const float* dL_dsigmas, const float* dL_drgbs, int64_t M, const float* x01,
                                 const void* density_weights_f16, const void* color_weights_f16, int32_t n_levels,
                                 int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale, const void* h_f16,
                                 const void* rgb_f16, const void* save_in_d, const void* save_acts_d, const void* save_in_c,
                                 const void* save_acts_c, float loss_scale, float* grad_density_params, float* grad_color_params,
                                 int64_t n_density_mlp_params, void* scratch, nrc_stream_t stream);
/* Same with UNINITIALISED gradient buffers of n_density_params / n_color_params values: the call SETS them (no accumulation).  What it
 * saves: the caller's 49 MB fill of the table gradient, and the read half of the read-modify-write of the hashed levels' slices -- their
 * owners write every entry, only the small levels and the MLP parts are zeroed first (one launch). */
int nrc_ngp_train_query_backward_set(const float* dL_dsigmas, const float* dL_drgbs, int64_t M, const float* x01,
                                     const void* density_weights_f16, const void* color_weights_f16, int32_t n_levels,
                                     int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale, const void* h_f16,
                                     const void* rgb_f16, const void* save_in_d, const void* save_acts_d, const void* save_in_c,
                                     const void* save_acts_c, float loss_scale, float* grad_density_params, float* grad_color_params,
                                     int64_t n_density_mlp_params, int64_t n_density_params, int64_t n_color_params, void* scratch,
                                     nrc_stream_t stream);
/* InstantNGPRayRenderingComponent.query_model (Renderer.py:48-53) as an encode + MLP kernel pair over Infinity-Cache sized
 * chunks (workspace: nrc_ngp_query_ws_bytes(M) bytes): xyz01 (M,3) f32 in [0,1], dirs (M,3) f32 unit
 * vectors -> sigmas (M) f32 = exp(fp16 feature 0), rgbs (M,3) f32 = fp16 sigmoid outputs. */
int64_t nrc_ngp_query_ws_bytes(int64_t M);
int nrc_ngp_query_fused(const float* xyz01, const float* dirs, int64_t M, const void* density_weights_f16,
                        const void* color_weights_f16, const void* table_f16, int32_t n_levels, int32_t log2_hashmap_size,
                        int32_t base_resolution, float per_level_scale, float* sigmas, float* rgbs, void* workspace,
                        nrc_stream_t stream);

/* =====================================================================================================
 * Group 4 -- diff_gaussian_rasterization  (replaces the external CUDA rasterizer pinned at
 *            src/Thirdparty/DiffGaussianRasterization.py:9 behind src/Methods/GaussianSplatting/Renderer.py:60-81,94-153,163-183)
 * P Gaussians, SH degree D (0..3), M = SH coefficients per Gaussian in `shs` (P,M,3); exactly one of shs | colors_precomp (P,3)
 * and exactly one of (scales (P,3), rotations (P,4)) | cov3D_precomp (P,6).  viewmatrix / projmatrix (16 floats each, the
 * (4,4) tensors the reference passes: w2c.T and w2c.T @ P.T), campos (3), bg (3) are HOST pointers.  16x16-pixel tiles.
 * Per-Gaussian state (geometry buffer): radii (P) i32, depths (P), points_xy (P,2), conic_opacity (P,4), rgb (P,3),
 * clamped (P) u8 bit mask (bit c = channel c clamped), cov3D (P,6), tiles_touched (P) u32.
 * Model-side parameters (extension beyond the reference's call, for callers that own the Gaussians: src/Methods/GaussianSplatting/Model.py:45-87):
 * shs_rest != NULL: `shs` is the DC part (P,1,3) and shs_rest the (P,M-1,3) remainder as the model stores them -- the (P,M,3) concatenation
 * of get_features is never built; raw_parameters != 0: opacities are logits, scales log-scales, rotations unnormalised, and sigmoid / exp /
 * normalisation run inside preprocess (the backward returns gradients w.r.t. the raw values; dL_dsh_rest receives the remainder's gradient).
 * splat_records (P,16) f32: one 64-byte line per Gaussian with everything the blend kernels read of it (screen position, conic, opacity,
 * colour) and the per-Gaussian part of their block-culling test; written by nrc_gs_preprocess, read by nrc_gs_bin_render / nrc_gs_backward.
 * (tile_fill doubles as `tile_order`: the tiles sorted by list length, longest first = launch order of the blend kernels.  ABI 3: with a binning
 * workspace it is written by nrc_gs_preprocess already -- the last workgroup of the tile scan orders the tiles --, otherwise by nrc_gs_bin_render;
 * either way it is valid after nrc_gs_bin_render and must reach nrc_gs_backward unchanged.)
 * Binning state: tile_counts, tile_fill (n_tiles) u32, ranges (n_tiles,2) u32, keys (num_rendered) u64, point_list
 * (num_rendered) i32.  Image state: n_contrib (H*W) u32, final_T (H*W).
 *   nrc_gs_preprocess : stages 1-2; num_rendered (DEVICE i64[2]): [0] = number of (tile, Gaussian) instances -- the caller reads it
 *                       to size keys / point_list (the reference's rasterizer pays the same device->host read); [1] = number of
 *                       (tile row, Gaussian) span records the binning needed: if it exceeds the span capacity the workspace was
 *                       sized for, [0] is not valid and the call is repeated with a workspace for at least [1] spans.
 *                       instance_capacity > 0 (fixed-capacity mode, for graph capture): point_list has that many entries and nothing
 *                       needs to be read back -- tile ranges are cut at the capacity, the instances that would land behind it are
 *                       dropped (the farthest of the last tiles), and [0] > instance_capacity / [1] > span capacity report it.
 *                       count_mailbox (ABI 4, optional, binning-workspace path only): HOST memory from nrc_host_mailbox_alloc.  The kernel
 *                       that finishes the counts also stores {[0], [1], mailbox_ticket} into mailbox[0..2], the ticket last: a host that wants the
 *                       counts as soon as they exist polls mailbox[2] for the ticket it passed (a value it has not used on this mailbox
 *                       before) instead of placing an event / copy / stream wait behind the call -- which on this runtime costs ~6 us of idle
 *                       GPU between this call's kernels and the next, and arrives ~15 us later.  One frame in flight per mailbox.
 *   camera_dev        : optional DEVICE float[38] = viewmatrix (16), projmatrix (16), campos (3), bg (3).  When given, the kernels read the
 *                       pose from it and the host arrays (viewmatrix, projmatrix, campos, bg) may be NULL: a camera that lives in device
 *                       tensors (GaussianRasterizationSettings) never crosses to the host, and a recorded graph follows its updates.
 *   nrc_gs_bin_render : stages 3-5 -> out_color (3,H,W) = C + T * bg, n_contrib, final_T.
 *   nrc_gs_backward   : dL_dpix (3,H,W) -> every gradient (all fully written; dL_dmean2D (P,3) is the screen-space gradient
 *                       consumed by densification, src/Methods/GaussianSplatting/Model.py:258).  grad_records: WORKSPACE (P,16) f32, 64-byte
 *                       aligned: the blend backward accumulates its nine per-Gaussian sums in one 64-byte record per
 *                       Gaussian (colour 3, opacity 1, mean2D 2, conic 3) so that a tile's flush for a Gaussian is one contiguous group of one
 *                       atomic instruction; dL_dmean2D / dL_dopacity (and dL_dconic (P,4) / dL_dcolor (P,3), which may be NULL) are written
 *                       from the records by the per-Gaussian backward, WHICH LEAVES EVERY RECORD IT READ CLEARED (ABI 3).  records_clear = 0:
 *                       the call clears the workspace first (any contents); records_clear != 0: the caller guarantees that all P x 16 floats
 *                       are zero -- e.g. the same buffer after a previous nrc_gs_backward of the same P -- and the clearing launch is skipped.
 * ===================================================================================================== */
/* bytes of the binning workspace `bin_hist` (depth pre-sort buffers, row-span records, cursors) for `span_capacity` span records
 * (0 = default 4 P + 65536); returns 0 when the image has more than 256 tile rows or columns: pass NULL, the per-tile key sort is used.
 * The same span_capacity goes to nrc_gs_preprocess and nrc_gs_bin_render. */
int64_t nrc_gs_bin_hist_bytes(int32_t P, int32_t W, int32_t H, int64_t span_capacity);
int nrc_gs_preprocess(int32_t P, int32_t D, int32_t M, int32_t W, int32_t H, const float* means3D, const float* shs,
                      const float* shs_rest, int32_t raw_parameters, const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                      const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                      const float* campos, const float* camera_dev, float tan_fovx, float tan_fovy, int32_t* radii, float* depths,
                      float* points_xy, float* conic_opacity, float* rgb, uint8_t* clamped, float* cov3D, uint32_t* tiles_touched,
                      uint32_t* tile_counts, uint32_t* ranges, uint32_t* tile_fill, uint32_t* bin_hist, int64_t span_capacity,
                      int64_t instance_capacity, float* splat_records, int64_t* num_rendered, int64_t* count_mailbox, int64_t mailbox_ticket,
                      nrc_stream_t stream);
/* 64 bytes of pinned, device-mapped, coherent host memory (zeroed) that kernels of this library may write and the host may poll: one address on
 * both sides; NRC_ERR_UNSUPPORTED when the runtime maps it elsewhere (use the device counters then).  Free with nrc_host_mailbox_free. */
int nrc_host_mailbox_alloc(int64_t** mailbox);
int nrc_host_mailbox_free(int64_t* mailbox);
/* camera_dev of a frame from a DEVICE pose: c2w_dev = the first 12 floats of a row-major camera-to-world (3,4) / (4,4), proj_t_dev = P^T (16) of
 * Cameras/Perspective.py's projection, bg3_dev (optional): out[0..16) = viewmatrix = w2c^T, [16..32) = viewmatrix @ P^T, [32..35) = camera position,
 * [35..38) = background (GaussianSplatting/Renderer.py:60-74 in one launch; the pose never visits the host). */
int nrc_gs_camera_block(const float* c2w_dev, const float* proj_t_dev, const float* bg3_dev, float* camera_dev_out, nrc_stream_t stream);
int nrc_gs_bin_render(int32_t P, int32_t W, int32_t H, const float* bg, const float* camera_dev, const int32_t* radii,
                      const float* depths, const float* points_xy, const float* conic_opacity, const float* rgb,
                      const uint32_t* ranges, uint32_t* tile_fill, const uint32_t* bin_hist, int64_t span_capacity,
                      int64_t instance_capacity, uint64_t* keys, int32_t* point_list, const float* splat_records, float* out_color,
                      uint32_t* n_contrib, float* final_T, nrc_stream_t stream);
int nrc_gs_backward(int32_t P, int32_t D, int32_t M, int32_t W, int32_t H, const float* bg, const float* means3D, const float* shs,
                    const float* shs_rest, int32_t raw_parameters, const float* opacities, const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                    const float* camera_dev, float tan_fovx, float tan_fovy, const int32_t* radii, const float* points_xy, const float* conic_opacity,
                    const float* rgb, const uint8_t* clamped, const float* cov3D, const int32_t* point_list,
                    const uint32_t* ranges, const float* splat_records, const uint32_t* tile_order, const uint32_t* n_contrib,
                    const float* final_T, const float* dL_dpix,
                    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                    float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest, float* dL_dscale, float* dL_drot, float* grad_records,
                    int32_t records_clear, nrc_stream_t stream);
/* nrc_gs_backward + the optimizer step of the `rest` SH tensor in one pass (round 6): 45 of a Gaussian's 59 parameters are SH coefficients above the dc term, and
 * their gradient rows sit in LDS at the end of the preprocessing backward -- Adam (apex FusedAdam, src/Methods/GaussianSplatting/Model.py:121-138, group 'f_rest':
 * no weight decay, eps as given) is applied THERE: dL/d shs_rest is never written and read back (180 B per Gaussian each way), the parameters are not read a second time.
 * shs_rest_param: the (P, M-1, 3) tensor the forward read, updated in place; rest_exp_avg / rest_exp_avg_sq its moments; bias corrections as host values.  Device camera
 * block only.  Same per-element arithmetic as nrc_adam_step.  The caller's optimizer must skip that tensor in this step (its .grad stays empty). */
int nrc_gs_backward_rest_step(int32_t P, int32_t D, int32_t M, int32_t W, int32_t H, const float* bg_host, const float* means3D, const float* shs,
                    float* shs_rest_param, int32_t raw_parameters, const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                    const float* camera_dev, float tan_fovx, float tan_fovy, const int32_t* radii, const float* points_xy, const float* conic_opacity,
                    const float* rgb, const uint8_t* clamped, const float* cov3D, const int32_t* point_list, const uint32_t* ranges,
                    const float* splat_records, const uint32_t* tile_order, const uint32_t* n_contrib, const float* final_T, const float* dL_dpix,
                    float* dL_dmean2D, float* dL_dopacity, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                    float* dL_drot, float* grad_records, int32_t records_clear, float* rest_exp_avg, float* rest_exp_avg_sq, float lr, float beta1, float beta2,
                    float eps, float bias_correction1, float bias_correction2, nrc_stream_t stream);

/* =====================================================================================================
 * Group 5 -- ray generation (replaces PerspectiveCamera.compute_local_ray_directions src/Cameras/Perspective.py:64-94
 * + View.get_rays / cam_to_world src/Datasets/utils.py:1033-1074 for undistorted perspective cameras).
 * intrinsics (HOST, 4 doubles): focal_x, focal_y, center_x, center_y.  c2w (HOST, 16 doubles, row-major 4x4).
 * Outputs (H*W,3) f32 each, y-major then x; any output pointer may be NULL.
 * ===================================================================================================== */
int nrc_generate_rays(int32_t width, int32_t height, const double* intrinsics, const double* c2w, float* origin,
                      float* direction, float* view_direction, nrc_stream_t stream);

/* =====================================================================================================
 * Group 6 -- fused InstantNGP image pipeline (MI355X-native restructuring of InstantNGPRenderer.render_image ->
 *            render_rays_inference, src/Methods/InstantNGP/Renderer.py:30-46,86-138,172-180): no ray tensors, no
 *            per-iteration host syncs, 4-byte sample records, 8-byte sample values, TILE-INTERLEAVED sample layout:
 *            a tile = 8x8 pixels (lane = (y&7)*8 + (x&7)); sample k of lane l of local tile T is slot
 *            (tile_off[T] + k) * 64 + l.  A shard is a range of tiles [tile_begin, tile_begin + n_tiles) of the
 *            ceil(W/8) x ceil(H/8) tile grid (row-major); per-ray arrays have n_tiles*64 entries.
 *   1. nrc_ngp_render_count : pixel -> ray (Group 5 semantics) -> centre shift, box slab test, near/far clamp -> DDA sample
 *                             count; writes ray_od (n_tiles,6,64) = per-tile SoA of (o - centre, d), ray_t (n,2), ray_cnt (n), tile_rows (2*n_tiles: rows per tile,
 *                             then samples per tile), tile_off (n_tiles+1), counter = (total rows, total samples).  intr/c2w/center3/half3: HOST pointers.
 *   2. nrc_ngp_render_write : ts (rows*64) f32 (-1 = hole), row_tile (rows) i32.
 *   3. nrc_ngp_query_samples: slots -> packed (h0, r, g, b) fp16 (sigma = exp(h0)); xyz_min3/xyz_size3 HOST pointers;
 *                             n_ray_tiles = tiles in ray_od (their SH coefficients are evaluated once per ray);
 *                             workspace: nrc_ngp_query_samples_ws_bytes(rows, n_ray_tiles) bytes.
 *      ABI 3, fixed row capacity (a frame without the host read of `counter`, e.g. inside a stream capture): the caller sizes ts / row_tile /
 *      packed / workspace for `row_capacity` rows, passes it to nrc_ngp_render_write (rows that would land behind it are not written; 0 = no
 *      bound) and as n_rows to nrc_ngp_query_samples together with n_rows_dev = counter (DEVICE, the count pass's total): the kernels then
 *      process only the rows that exist.  counter[0] > row_capacity afterwards means the frame overflowed and must be rendered again.
 *   4. nrc_ngp_composite_image: serial per-ray compositing + background / clamps (bg3 HOST) into full-image buffers
 *                             rgb (H*W,3), alpha (H*W), depth (H*W) -- only the shard's pixels are written.
 * ===================================================================================================== */
/* pixel footprint of a tile (tile width x height = 64; the tile grid is ceil(W / tw) x ceil(H / th), row-major) */
int nrc_ngp_tile_width(void);
int nrc_ngp_tile_height(void);
int nrc_ngp_render_count(int32_t width, int32_t height, const double* intrinsics, const double* c2w, const float* center3,
                         const float* half3, float near_plane, float far_plane, int64_t tile_begin, int64_t n_tiles,
                         const uint8_t* density_bitfield, int32_t cascades, float scale, float exp_step_factor,
                         int32_t grid_size, int32_t max_samples, float* ray_od, float* ray_t, int32_t* ray_cnt,
                         int32_t* tile_rows, int32_t* tile_off, int32_t* counter, float* ts_provisional, int64_t* count_mailbox,
                         int64_t mailbox_ticket, nrc_stream_t stream);
/* count_mailbox (ABI 4, optional, NULL allowed): HOST memory from nrc_host_mailbox_alloc (group 4).  The scan that closes the count pass stores
 * {counter[0], counter[1], mailbox_ticket} into mailbox[0..2], the ticket last; the host polls mailbox[2] for its ticket instead of copying
 * `counter` back (no device-to-host copy and no stream wait between the count pass and nrc_ngp_render_write: 42 -> ~10 us of idle GPU per frame). */
/* ts_provisional (optional, NULL allowed; nrc_ngp_render_provisional_bytes(n_tiles, max_samples) bytes = max_samples rows of 256 B per
 * tile): the count pass parks every sample's t there and steps 2 copy them into their final rows instead of marching the rays a
 * second time (pass the same buffer to both calls).  HBM is plentiful on this part: 2.6 GB for an 800x800 image. */
int64_t nrc_ngp_render_provisional_bytes(int64_t n_tiles, int32_t max_samples);
int nrc_ngp_render_write(int64_t n_tiles, const uint8_t* density_bitfield, int32_t cascades, float scale,
                         float exp_step_factor, int32_t grid_size, int32_t max_samples, const float* ray_od,
                         const float* ray_t, const int32_t* ray_cnt, const int32_t* tile_off, float* ts, int32_t* row_tile,
                         const float* ts_provisional, int64_t row_capacity, nrc_stream_t stream);
int64_t nrc_ngp_query_samples_ws_bytes(int64_t n_rows, int64_t n_ray_tiles);
int nrc_ngp_query_samples(const float* ts, int32_t* row_tile, const float* ray_od, int64_t n_rows, int64_t n_ray_tiles,
                          const float* xyz_min3, const float* xyz_size3, const void* density_weights_f16, const void* color_weights_f16,
                          const void* table_f16, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution,
                          float per_level_scale, void* packed_f16, void* workspace, const int32_t* n_rows_dev, const int32_t* arena_tile_off,
                          int32_t arena_rows, nrc_stream_t stream);
/* ABI 4, the arena queried in place (the default of the single-pass frame): with ts_provisional the count pass has every sample's t already in
 * 256-byte rows per tile -- sample k of local tile lt in arena row lt * max_samples + k, holes (-1) up to the tile's longest ray included -- and the
 * compact `ts` rows differ from it only by WHERE a row lives.  nrc_ngp_render_write is then not needed at all (the 2 x 320 MB copy of an 800x800
 * frame is skipped: 150 us): nrc_ngp_query_samples takes ts = ts_provisional, arena_tile_off = tile_off, arena_rows = max_samples, FILLS row_tile
 * (n_rows entries, from tile_off, in the launch that evaluates the rays' SH) and reads slot i of row r = i >> 6 at arena row
 * row_tile[r] * arena_rows + (r - tile_off[row_tile[r]]); nrc_ngp_composite_image takes the same pointer and arena_rows (0 = compact rows).
 * (nrc_ngp_render_write(ts = NULL) writes row_tile only, for callers of the single stages.)  `packed` stays indexed by compact rows.  Same values,
 * same pictures. */
/* Layer-major variant of steps 2-4: rows ordered by sample index k first (all tiles' k = 0, then k = 1, ...), so that consecutive
 * chunks of rows are depth slabs of the image; after every slab the finished tiles (all rays saturated below T_threshold or out of
 * samples) write their pixels and their remaining rows are skipped -- the early termination of the reference's alive-ray loop
 * (src/Methods/InstantNGP/Renderer.py:118-132) at slab granularity, without host round trips.  Same image as steps 2-4.
 *   nrc_ngp_render_write_layers: like step 2, plus layer_off (max_samples + 1) i32 and row_of (rows) i32 = the row of (tile, k).
 *   nrc_ngp_render_layers      : steps 3 + 4 interleaved per slab; workspace nrc_ngp_render_layers_ws_bytes(rows, n_tiles);
 *                                skipped_rows (optional, 1 i32): number of rows the early termination saved.
 *   ABI 4: row_tile is READ AND WRITTEN -- when a tile finishes, the slab compositor overwrites the entries of its remaining rows with -1 - tile,
 *   which is where the query kernels of the following slabs look first (no feature is computed, read or written for such a row).
 *   ABI 4, the arena queried in place (as for the single pass): nrc_ngp_render_write_layers(ts = NULL, ..., row_k) copies nothing and also writes
 *   row_k (rows) i32 = which sample of its tile a row is; nrc_ngp_render_layers takes ts = ts_provisional, arena_row_k = row_k, arena_rows =
 *   max_samples and reads slot i of row r at arena row row_tile[r] * arena_rows + row_k[r] (305 -> ~10 us for the write step of an 800x800 frame). */
int nrc_ngp_render_write_layers(int64_t n_tiles, const uint8_t* density_bitfield, int32_t cascades, float scale, float exp_step_factor,
                                int32_t grid_size, int32_t max_samples, const float* ray_od, const float* ray_t, const int32_t* ray_cnt,
                                const int32_t* tile_rows, const int32_t* tile_off, float* ts, int32_t* row_tile, int32_t* layer_off,
                                int32_t* row_of, const float* ts_provisional, int32_t* row_k, nrc_stream_t stream);
int64_t nrc_ngp_render_layers_ws_bytes(int64_t n_rows, int64_t n_ray_tiles);
int nrc_ngp_render_layers(const float* ts, int32_t* row_tile, const float* ray_od, int64_t n_rows, int64_t n_ray_tiles,
                          const float* xyz_min3, const float* xyz_size3, const void* density_weights_f16,
                          const void* color_weights_f16, const void* table_f16, int32_t n_levels, int32_t log2_hashmap_size,
                          int32_t base_resolution, float per_level_scale, const int32_t* ray_cnt, const int32_t* tile_rows,
                          const int32_t* tile_off, const int32_t* row_of, int32_t width, int32_t height, int64_t tile_begin,
                          int32_t cascades, float exp_step_factor, int32_t grid_size, int32_t max_samples, float T_threshold,
                          const float* bg3_host, void* packed_f16, float* rgb, float* alpha, float* depth, int32_t* skipped_rows,
                          void* workspace, const int32_t* arena_row_k, int32_t arena_rows, nrc_stream_t stream);
/* stage 3a on its own (the dominant kernel of the pipeline; used by bench.py's roofline leg): hash-grid features of the n_rows (<= 131072) rows
 * first_row .. first_row + n_rows - 1 of the frame -- ts / row_tile are the FRAME's arrays, not offset ones (ABI 4) --, fragment-major: the 16-byte
 * vector [((j>>5)*4 + ((g + (j>>5))&3))*32 + (j&31)] = levels 4g..4g+3 (fp16x2) of slot j of the chunk.  arena_tile_off / arena_rows: as for
 * nrc_ngp_query_samples (NULL / 0: compact rows), so that the kernel is timed in the form the frame runs it */
/* The brick of the tiled layout that ONE WAVE of the encoder gathers for: 2^log2_x x 2^log2_y pixels of a ray tile x 2^(6 - log2_x - log2_y) consecutive
 * steps (default 8 x 2 x 4).  Hash-table entries are contiguous along world x only, so which brick shares the most cache lines depends on how the
 * image axes and the viewing direction lie to that axis: the host may pick a shape per pose (InstantNGPRenderer does, from the camera's axes).
 * Both < 0: back to the default.  A setting of the CALLING HOST THREAD, read when that thread launches the encoder (set it on the thread that
 * enqueues the frame); the features are identical for every shape. */
int nrc_ngp_set_encoder_shape(int32_t log2_x, int32_t log2_y);
int nrc_ngp_encode_samples(const float* ts, const int32_t* row_tile, const float* ray_od, int64_t first_row, int64_t n_rows, const float* xyz_min3,
                           const float* xyz_size3, const void* table_f16, int32_t n_levels, int32_t log2_hashmap_size,
                           int32_t base_resolution, float per_level_scale, void* features_f16, const int32_t* arena_tile_off,
                           int32_t arena_rows, nrc_stream_t stream);
/* stage 3b on its own (the MFMA kernel; bench.py's second roofline object): features as written by nrc_ngp_encode_samples for the same
 * n_rows (<= 131072) rows -> packed (h0, r, g, b) fp16 of the FRAME (indexed by the frame's slots); ray_sh_workspace: n_ray_tiles * 2048 bytes,
 * filled by this call with the rays' SH coefficients; n_ray_tiles = 0: it holds them already (they belong to the image, a caller looping over
 * chunks fills it once) */
int nrc_ngp_mlp_samples(const float* ts, const int32_t* row_tile, const float* ray_od, int64_t first_row, int64_t n_rows, int64_t n_ray_tiles,
                        const void* features_f16, const void* density_weights_f16, const void* color_weights_f16,
                        void* packed_f16, void* ray_sh_workspace, const int32_t* arena_tile_off, int32_t arena_rows, nrc_stream_t stream);
int nrc_ngp_composite_image(const void* packed_f16, const float* ts, const int32_t* ray_cnt, const int32_t* tile_off,
                            int32_t width, int32_t height, int64_t tile_begin, int64_t n_tiles, int32_t cascades,
                            float exp_step_factor, int32_t grid_size, int32_t max_samples, float T_threshold,
                            const float* bg3, float* rgb, float* alpha, float* depth, int64_t row_capacity, int32_t arena_rows,
                            nrc_stream_t stream);

/* =====================================================================================================
 * Group 7 -- SSIM map and its gradient (3DGS loss): replaces fused_ssim as imported at
 *            src/Thirdparty/FusedSSIM.py:15 and called at src/Optim/Losses/DSSIM.py:11-18.  Images are (planes, H, W) f32 with
 *            planes = batch * channels; 11x11 Gaussian window (sigma 1.5), zero "same" padding.  The three derivative maps
 *            (d ssim / d mu1, d sigma1^2, d sigma12) are written when all three pointers are non-null (training).
 * ===================================================================================================== */
int nrc_ssim_forward(const float* img1, const float* img2, int64_t planes, int32_t H, int32_t W, float C1, float C2, float* ssim_map,
                     float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, nrc_stream_t stream);
int nrc_ssim_backward(const float* img1, const float* img2, int64_t planes, int32_t H, int32_t W, const float* dL_dmap,
                      const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1,
                      nrc_stream_t stream);
/* The whole photometric loss of the 3DGS trainer (src/Methods/GaussianSplatting/Loss.py:11-23: lambda_l1 * L1 + lambda_dssim * (1 - SSIM), weights
 * Trainer.py:34-35) in three launches instead of the ~22 that the L1 term, the two means and the weighting take as tensor operations around
 * nrc_ssim_forward / _backward (each of them >= 5 us on a scalar): the SSIM stencil also sums |image - target| and the SSIM values per workgroup,
 * one workgroup adds the partial sums up in a fixed order (double accumulators), and the backward stencil takes dL/dmap = -lambda_dssim * g / n as a
 * constant and adds lambda_l1 * g / n * sign(image - target); g = upstream_dev[0], the gradient of the loss VALUE, read on the device (NULL: 1).
 * loss3 (DEVICE float[3]) = {loss, mean |image - target|, mean SSIM}; workspace: nrc_photometric_loss_ws_floats(planes, H, W) floats; the three
 * derivative maps (planes x H x W each) as for nrc_ssim_forward (all NULL: value only).  planes = B * C, "same" padding. */
int64_t nrc_photometric_loss_ws_floats(int64_t planes, int32_t H, int32_t W);
int nrc_photometric_loss_forward(const float* image, const float* target, int64_t planes, int32_t H, int32_t W, float C1, float C2,
                                 float lambda_l1, float lambda_dssim, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                                 float* workspace, float* loss3, nrc_stream_t stream);
int nrc_photometric_loss_backward(const float* image, const float* target, int64_t planes, int32_t H, int32_t W, float lambda_l1,
                                  float lambda_dssim, const float* upstream_dev, const float* dm_dmu1, const float* dm_dsigma1_sq,
                                  const float* dm_dsigma12, float* dL_dimage, nrc_stream_t stream);

/* =====================================================================================================
 * Group 8 -- fused Adam step: replaces apex.optimizers.FusedAdam (src/Thirdparty/Apex.py:17) as constructed at
 *            src/Methods/InstantNGP/Trainer.py:33-38 and src/Methods/GaussianSplatting/Model.py:131-136.  One flat f32 tensor
 *            per call (16-byte aligned pointers take the vector path).  bias_correction_k = 1 - beta_k^step (host).  adam_w_mode = 0: L2
 *            weight decay added to the gradient (apex ADAM_MODE_0), 1: decoupled.  grad_scale / found_inf: optional DEVICE
 *            scalars of torch.amp.GradScaler (gradient divided by *grad_scale; the whole step is skipped when *found_inf != 0).
 *            bias_corrections_dev (optional DEVICE float[2]) overrides the two host values: nrc_adam_prepare writes it once per
 *            group and step as 1 - beta^(host_step - *skipped_steps) and advances *skipped_steps when *found_inf != 0, so the
 *            effective step count stands still on an overflow-skipped step (with apex the scaler does not call step() then)
 *            without a host read of found_inf.  With device_step (optional DEVICE int32, apex's capturable mode) host_step / skipped_steps are
 *            not used: the kernel advances *device_step itself unless *found_inf != 0, so a step recorded in a HIP graph counts on replay.
 *            lr_dev (optional DEVICE float) overrides lr for the same reason.  l2_slice_coeff / l2_slice_count: the gradient of the first
 *            l2_slice_count elements gets + l2_slice_coeff * parameter -- the reference's weight-decay LOSS term on the MLP weights in front of
 *            the hash table (InstantNGP/Model.py:38-44, Loss.py:15: 0.5e-6 * mean w^2 -> coeff 1e-6 / n) applied where the parameters are
 *            read anyway, instead of a dense 12 M-element gradient through autograd; 0 / 0 = off.  param_f16_out (optional): the updated parameters are also written as fp16 --
 *            the compute copy the tinycudann replacement reads (Group 3), which therefore can never go stale after a step.
 * ===================================================================================================== */
/* GradScaler's inf / NaN check of one f32 gradient tensor (torch.amp.GradScaler._check_inf_per_device, used at InstantNGP/Trainer.py:89-93 with an
 * optimizer that applies the scale itself): *found_inf (DEVICE f32, not cleared here) is set to 1 when any of the n values is not finite. */
int nrc_nonfinite_check(const float* grad, int64_t n, float* found_inf, nrc_stream_t stream);
/* The same for up to four tensors in ONE launch (an entry with n = 0 is skipped): the two parameter vectors of the InstantNGP model. */
int nrc_nonfinite_check4(const float* g0, int64_t n0, const float* g1, int64_t n1, const float* g2, int64_t n2, const float* g3, int64_t n3,
                         float* found_inf, nrc_stream_t stream);
int nrc_adam_prepare(int32_t host_step, float beta1, float beta2, const float* found_inf, int32_t* skipped_steps,
                     int32_t* device_step, float* bias_corrections, nrc_stream_t stream);
int nrc_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int32_t adam_w_mode, float bias_correction1,
                  float bias_correction2, const float* bias_corrections_dev, const float* lr_dev, const float* grad_scale,
                  const float* found_inf, void* param_f16_out, float l2_slice_coeff, int64_t l2_slice_count, nrc_stream_t stream);
/* nrc_adam_step on up to 12 tensors in ONE launch, each with its own learning rate and bias corrections (host values: the plain, non-capturable step of
 * apex.optimizers.FusedAdam over several single-tensor groups -- src/Methods/GaussianSplatting/Model.py:121-138 builds six; apex's multi_tensor_apply).
 * params / grads / exp_avg / exp_avg_sq: HOST arrays of n_tensors device pointers; sizes / lrs / bias_correction1 / bias_correction2: HOST arrays.
 * Same arithmetic per element as nrc_adam_step (csrc/adam_math.h). */
int nrc_adam_step_multi(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                        const int64_t* sizes, const float* lrs, const float* bias_correction1, const float* bias_correction2, float beta1, float beta2,
                        float eps, float weight_decay, int32_t adam_w_mode, nrc_stream_t stream);

/* =====================================================================================================
 * Group 9 -- mean squared distance to the 3 nearest neighbours (3DGS scale initialisation): replaces simple_knn._C.distCUDA2
 *            (src/Thirdparty/SimpleKNN.py:17-18, used at src/Optim/knn_utils.py:34-38).  The points must arrive in Morton order
 *            (Group 2 codes + a sort: nerficg_amd/simple_knn does both); out_sorted[i] belongs to sorted point i.  n >= 4.
 *            workspace: nrc_knn3_ws_bytes(n) bytes.  Exact (not approximate) neighbours.
 * ===================================================================================================== */
int64_t nrc_knn3_ws_bytes(int64_t n);
int nrc_knn3_mean_sq_dist(const float* points_morton_sorted, int64_t n, float* out_sorted, void* workspace, nrc_stream_t stream);

/* =====================================================================================================
 * Group 10 -- 3DGS densification bookkeeping on the device: replaces the boolean-mask / torch.cat surgery of
 *            src/Methods/GaussianSplatting/Model.py:157-246 and src/Optim/adam_utils.py:21-98.
 *   nrc_gs_densify_stats : add_densification_stats (Model.py:243-246): where radii > 0, grad_accum += |viewspace_grad[:, :2]|,
 *                          n_observations += 1.  viewspace_grad is (P, ld) f32 (ld = 3 or 4 as the rasterizer hands it out).
 *   nrc_gs_densify_plan  : densify_and_prune (Model.py:226-241) as a row list.  grads = accum / max(n_obs, 1); clone where
 *                          grads >= grad_threshold and max exp(log_scale) <= dense_extent (= percent_dense * cameras_extent), split
 *                          where it is larger; prune where sigmoid(opacity_logit) < min_opacity or (max_scale > 0 and) max scale >
 *                          max_scale, applied to originals, clones and the two /1.6 children like the reference's final prune.
 *                          Row o of the resulting arrays comes from Gaussian src[o]; kind[o] = 0 kept original, 1 clone, 2 split
 *                          child with aux[o] = row of the (2 * counts[4], 3) standard-normal tensor (else -1).  Order as the
 *                          reference leaves it: originals, clones, children copy 0, children copy 1.  src/kind/aux need 2 * P
 *                          entries.  counts[5] = {rows out, kept originals, kept clones, kept children per copy, split selected}.
 *                          grad_accum == n_observations == NULL: prune only.  grad_threshold must be > 0 (a threshold <= 0 would
 *                          also split the fresh clones; NRC_ERR_UNSUPPORTED).  workspace: nrc_gs_densify_plan_ws_bytes(P).
 *   nrc_gs_densify_split_children : positions / log-scales of the kind-2 rows (Model.py:196-202): R(q) (z * exp(log_scale)) + p and
 *                          log(exp(log_scale) / 1.6); z = noise[aux[o]] are the draws of torch.normal(0, std) before scaling.
 *   nrc_gather_rows      : out[t][o, :] = in[t][src[o], :] for up to 24 f32 tensors of row_floats[t] floats per row in one launch
 *                          (prune / extend / sort of src/Optim/adam_utils.py:21-98 for every group and both Adam moments at once);
 *                          tensors with zero_new[t] != 0 get zeros in rows with kind[o] != 0 (the moments of new Gaussians).  in, out,
 *                          row_floats, zero_new are HOST arrays of n_tensors entries; kind may be NULL (pure permutation / prune).
 *   nrc_scatter_rows     : the inverse for a list of distinct rows: out[t][dst[r], :] = in[t][r, :], r < n_in (the reduced union rows of a view-parallel
 *                          step go back into the gradient tensors: nerficg_amd.parallel.UnionRowExchange; no reference counterpart).
 *   nrc_compact_mask     : indices[0..*count) = ascending positions where mask != 0 (u8); workspace nrc_compact_mask_ws_bytes(n).
 * ===================================================================================================== */
int nrc_gs_densify_stats(const float* viewspace_grad, int32_t ld, const int32_t* radii, int64_t P, float* grad_accum,
                         int32_t* n_observations, nrc_stream_t stream);
int64_t nrc_gs_densify_plan_ws_bytes(int64_t P);
int nrc_gs_densify_plan(const float* grad_accum, const int32_t* n_observations, const float* log_scales, const float* opacity_logits,
                        int64_t P, float grad_threshold, float dense_extent, float min_opacity, float max_scale, int32_t* src,
                        int32_t* kind, int32_t* aux, int32_t* counts, void* workspace, nrc_stream_t stream);
int nrc_gs_densify_split_children(const int32_t* src, const int32_t* kind, const int32_t* aux, int64_t n_out, const float* positions,
                                  const float* log_scales, const float* rotations, const float* noise, float* positions_out,
                                  float* log_scales_out, nrc_stream_t stream);
int nrc_gather_rows(const float* const* in, float* const* out, const int32_t* row_floats, const int32_t* zero_new, int32_t n_tensors,
                    const int32_t* src, const int32_t* kind, int64_t n_out, nrc_stream_t stream);
int nrc_scatter_rows(const float* const* in, float* const* out, const int32_t* row_floats, int32_t n_tensors, const int32_t* dst, int64_t n_in,
                     nrc_stream_t stream);
int64_t nrc_compact_mask_ws_bytes(int64_t n);
int nrc_compact_mask(const uint8_t* mask, int64_t n, int32_t* indices, int32_t* count, void* workspace, nrc_stream_t stream);

/* =====================================================================================================
 * Group 11 -- InstantNGP occupancy-grid maintenance in one call: replaces the scratch-grid index_put, the
 *            where/maximum EMA, the masked mean pulled to the host and the packbits call of
 *            src/Methods/InstantNGP/Renderer.py:258-272.  grid (cascades, cells_per_cascade) f32 is updated in place:
 *            cells >= 0 become max(grid * decay, density sampled for the cell this round (0 if none; the maximum if several)), cells < 0
 *            (carved) stay.  threshold_out[0] = min(mean of the cells > 0, density_threshold) (NaN when there is none, like the
 *            reference's empty mean), threshold_out[1] = that mean; bitfield bit = cell > threshold_out[0] (raymarching.cu:138-161).
 *            cell_indices (cascades, samples_per_cascade) i64 Morton indices, densities the same shape, f32 (dtype 0) or f16 (1).
 *            Everything stays on the device (threshold_out is a DEVICE pointer to 2 floats).  cells_per_cascade % 8 == 0; grid and
 *            workspace 16-byte aligned; workspace: nrc_occupancy_update_ws_bytes(cascades * cells_per_cascade).
 * ===================================================================================================== */
int64_t nrc_occupancy_update_ws_bytes(int64_t n_cells_total);
int nrc_occupancy_update(float* grid, const int64_t* cell_indices, const void* densities, int32_t densities_dtype, int32_t cascades,
                         int64_t cells_per_cascade, int64_t samples_per_cascade, float decay, float density_threshold,
                         uint8_t* bitfield, float* threshold_out, void* workspace, nrc_stream_t stream);
/* Which cells are re-queried and where (src/Methods/InstantNGP/Renderer.py:183-206 cell choice, :251-258 positions): mode 0 = every cell
 * of every cascade (warm-up; per cascade grid_size^3 entries in Morton order), mode 1 = per cascade n_per_half cells drawn uniformly from
 * the grid followed by n_per_half cells drawn uniformly from occupied_indices[c * occupied_stride + [0, occupied_counts[c])) (Morton
 * indices of the cells with grid > threshold, e.g. from nrc_compact_mask; a cascade without any gets ignored entries, index -1).
 * cell_indices (cascades, per) i64 Morton indices, points (cascades * per, 3) f32 = box-centred query positions
 * ((coord / (G-1)) * 2 - 1) * (s - s/G) + U(-1,1) * s/G with s = min(2^(c-1), scale).  seed: DEVICE pointer to one i64; the draws are a pure
 * function of it (counter-based generator), so no host synchronisation and no dependence on the launch geometry. */
int nrc_occupancy_draw_cells(int32_t cascades, int32_t grid_size, float scale, int32_t mode, int64_t n_per_half, const int64_t* seed,
                             const int32_t* occupied_indices, const int32_t* occupied_counts, int64_t occupied_stride,
                             int64_t* cell_indices, float* points, nrc_stream_t stream);
/* Frustum / alpha-mask carving (Renderer.py:208-245).  remaining (cascades, G^3) u8, cell (x,y,z) at x + G*(y + G*z), initialised by the
 * caller to `subtractive`; one call per view: remaining = remaining OR seen (AND when subtractive), seen = the cell centre projects inside
 * the image (world -> camera (p - position) @ R, Datasets/utils.py:1027-1031; pinhole Perspective.py:39-52) with near < depth < far and,
 * when alpha_mask (height, width) f32 on the device is given, some pixel of the 3x3 neighbourhood of the hit pixel has alpha > 0.
 * c2w_rotation9 / position3 / center3 are HOST arrays.  nrc_occupancy_carve_finish dilates the survivors by one cell (3x3x3) and
 * writes grid[c][morton(x,y,z)] = 0 for them and -1 (never sampled again) for the rest. */
int nrc_occupancy_carve_view(uint8_t* remaining, int32_t cascades, int32_t grid_size, float scale, const float* center3,
                             const float* c2w_rotation9, const float* position3, float focal_x, float focal_y, float center_x,
                             float center_y, int32_t width, int32_t height, float near_plane, float far_plane,
                             const float* alpha_mask, int32_t subtractive, nrc_stream_t stream);
int nrc_occupancy_carve_finish(const uint8_t* remaining, int32_t cascades, int32_t grid_size, float* grid, nrc_stream_t stream);

/* =====================================================================================================
 * Group 12 -- stage timer (measurement; no reference counterpart).  bench.py's per-kernel roofline entries must be
 * timed with HIP events on the launch stream INSIDE the run that reports them; the kernels of one entry point are
 * launched back to back inside the library, where a caller cannot place events.  While armed, the multi-kernel entry
 * points (nrc_gs_preprocess / nrc_gs_bin_render / nrc_gs_backward, nrc_grid_backward, nrc_ngp_train_query_forward /
 * _backward) record an event behind each of their kernels; nothing is recorded inside a stream capture.
 *   _begin(capacity): clears the list, arms the timer (at most `capacity` marks are kept).
 *   _end: disarms, waits for the last mark, writes up to max_stages (name, milliseconds) pairs in launch order -- names as
 *         32-byte zero-terminated slots -- and the number written to *count (host pointers).
 * One harness, one stream, one call sequence at a time (not thread-safe).
 * ===================================================================================================== */
int nrc_stage_timer_begin(int32_t capacity);
int nrc_stage_timer_end(int32_t max_stages, char* names, float* ms, int32_t* count);

/* =====================================================================================================
 * Group 13 -- the InstantNGP training iteration on device-resident state (round 5).  Replaces, for ONE iteration of
 *            src/Methods/InstantNGP/Trainer.py:79-94: RayPoolSampler.get (Optim/Samplers/DatasetSamplers.py:53-66: ray_pool[ids]), `torch.rand(3)`,
 *            InstantNGPRayRenderingComponent.forward + render_rays_training (Renderer.py:55-84), InstantNGPLoss's colour term (Loss.py:15-22: mean
 *            squared error; the weight-decay term is FusedAdam's L2 slice, group 8), GradScaler.scale / .step / .update (Trainer.py:88-91) and
 *            FusedAdam.step -- as five enqueue-only calls, 13 kernel launches (the recorded iteration of round 4: 33).  Nothing is read back;
 *            every count the host may want (marched samples, dropped samples, loss) stays in device memory for the caller to look at when convenient.
 *
 *   nrc_ngp_train_march : the batch and its samples.  Ray n of the batch is pool row ids[n] (ids != NULL) or order[cursor[0] + n] (the
 *            resident sampling order; the call advances cursor[0] by the number of live rays).  n_rays_dev (DEVICE i32[1], NULL = ray_capacity):
 *            live rays; rows behind them become rays that miss the box (the batch size can change between replays of a recorded iteration).
 *            Outputs: rays_o (box-centred) / rays_d (ray_capacity,3), hits_t (ray_capacity,2) = [t_in, t_out] clipped to [near, far]
 *            (nrc_ngp_clip_rays values), target_rgb (ray_capacity,3) = pool_rgb, or lerp(bg, pool_rgb, pool_alpha).clamp(0,1) when pool_alpha is
 *            given (Datasets/utils.py:185-189), bg (3) = this iteration's random background; then the march of nrc_raymarching_train_capped on
 *            those rays with jitter noise[n]: rays_a, counter (uncut samples, live rays), xyzs / dirs / deltas / ts (sample_capacity rows, the
 *            unused tail inert), *overflow (may be NULL).  Random numbers: Philox4x32-10, key = rng_state[0] (seed), counter = (global ray
 *            index ray_offset + n | iteration rng_state[1]); the call advances rng_state[1].  bg_in (3) / noise_in (ray_capacity), when given,
 *            are used instead of the draws.  center3 / half3: HOST.  workspace: nrc_ngp_train_march_ws_bytes, ZEROED once by the caller
 *            (it starts with arrival counters that every call leaves at zero).  ray_capacity <= 32 768 (one wave per ray).
 *   nrc_ngp_train_query_forward (group 3) on (xyzs, dirs), n_samples_dev = counter: a launch sized for sample_capacity rows works on the rows that
 *            hold samples (min(counter[0], sample_capacity)); the same pointer goes to nrc_ngp_train_query_backward_cleared.
 *   nrc_ngp_train_loss  : compositing, `rgb + (1 - alpha) bg`, mean squared error against target_rgb over the live rays (counter[1]), times
 *            *loss_scale_dev (NULL = 1), and the whole way back: dL_dsigmas (M), dL_drgbs (M,3) of loss2[1] -- every row written.  loss2[0] =
 *            the loss, loss2[1] = scaled; ray_rgb / ray_alpha / ray_depth (optional) = the 'rgb' / 'alpha' / 'depth' of Renderer.py:83.  The same
 *            launch clears zero_a[0, n_zero_a) and zero_b[0, n_zero_b) (f32): what nrc_ngp_train_query_backward_cleared wants cleared.
 *            workspace: nrc_ngp_train_loss_ws_bytes(ray_capacity), zeroed once.
 *   nrc_ngp_train_query_backward_cleared : nrc_ngp_train_query_backward_set without its clearing launch; the caller has cleared
 *            grad_density_params[0, nrc_ngp_train_query_clear_floats(...)) and all of grad_color_params.
 *   nrc_amp_adam_step (group 8 family): GradScaler.step + FusedAdam(capturable).step + GradScaler.update for the (up to two) tensors of one
 *            parameter group in two launches.  Launch 1 checks the gradients for inf / NaN; its last workgroup holds or advances *device_step,
 *            writes bias_corrections (2), state4[1] = found_inf of this step, state4[2] = 1 / scale the gradients carry, and applies torch's
 *            scale update rule (back off on overflow; grow after growth_interval clean steps) to *scale / *growth_tracker (scale NULL: no
 *            scaler).  state4[3] (sticky): set when the Adam launch met an inf / NaN gradient element and left that element alone (round 6).  Launch 2 is the Adam update of nrc_adam_step for both tensors (skipped when found_inf).  state4 (f32[4]) and ticket
 *            (u32[272]: a two-level arrival counter) zeroed once by the caller.  lr_dev (NULL: the host value lr).  skipped_steps (optional, i32[1]): counts
 *            the steps an overflow skipped.
 * ===================================================================================================== */
/* InstantNGPModel.weight_decay_mlp (src/Methods/InstantNGP/Model.py:38-44: mean squared MLP weight over both networks) as one launch each way.
 *   nrc_sum_squares_two: out[0] = (sum a[0,n_a)^2 + sum b[0,n_b)^2) * inv_n, one workgroup, fixed order.
 *   nrc_clear_seed_two : what stands in front of nrc_ngp_train_query_backward_cleared -- grad_a[0,clear_a) / grad_b[0,clear_b) are cleared, and their
 *                        leading seed_a / seed_b elements start at coeff * upstream_dev[0] * w instead of zero: the gradient of the weight-decay term
 *                        (coeff = 2 / n, upstream = dL/d(term), DEVICE scalar) joins the networks' gradients without a dense 12 M-element tensor. */
/* InstantNGPLoss.forward (src/Methods/InstantNGP/Loss.py:18-26) as one launch: out3 = (mse + weight_decay_weight * wd, mse, wd) with
 * mse = mean((pred - target)^2) over n elements and wd = (sum a^2 + sum b^2) * inv_n_weights.  Backward: nrc_mse_scaled_backward for pred, and the
 * weight-decay gradient as seeds of nrc_clear_seed_two. */
int nrc_ngp_loss_forward(int64_t n, const float* pred, const float* target, const float* weights_a, int64_t n_a, const float* weights_b, int64_t n_b,
                         float inv_n_weights, float weight_decay_weight, float* out3, nrc_stream_t stream);
int nrc_sum_squares_two(const float* a, int64_t n_a, const float* b, int64_t n_b, float inv_n, float* out, nrc_stream_t stream);
int nrc_clear_seed_two(float* grad_a, int64_t clear_a, const float* w_a, int64_t seed_a, float* grad_b, int64_t clear_b, const float* w_b, int64_t seed_b,
                       const float* upstream_dev, float coeff, nrc_stream_t stream);
int64_t nrc_ngp_train_march_ws_bytes(int64_t ray_capacity, int32_t max_samples);
int nrc_ngp_train_march(const int64_t* ids, const int64_t* order, int64_t* cursor, const int32_t* n_rays_dev, int64_t ray_capacity, int64_t n_pool,
                        int64_t ray_offset, const float* pool_origin, const float* pool_dir, const float* pool_rgb, const float* pool_alpha,
                        const float* center3, const float* half3, float near_plane, float far_plane, const uint8_t* density_bitfield, int32_t cascades,
                        float scale, float exp_step_factor, int32_t grid_size, int32_t max_samples, uint64_t* rng_state, const float* bg_in,
                        const float* noise_in, int64_t sample_capacity, float* rays_o, float* rays_d, float* hits_t, float* target_rgb, float* bg,
                        int64_t* rays_a, int32_t* counter, float* xyzs, float* dirs, float* deltas, float* ts, int64_t* overflow, void* workspace,
                        nrc_stream_t stream);
int64_t nrc_ngp_train_loss_ws_bytes(int64_t ray_capacity);
int nrc_ngp_train_loss(const float* sigmas, const float* rgbs, const float* deltas, const float* ts, const int64_t* rays_a, const int32_t* counter,
                       int64_t ray_capacity, int64_t sample_capacity, float T_threshold, const float* bg_dev, const float* target_rgb,
                       const float* loss_scale_dev, float* ray_rgb, float* ray_alpha, float* ray_depth, float* loss2, float* dL_dsigmas, float* dL_drgbs,
                       float* zero_a, int64_t n_zero_a, float* zero_b, int64_t n_zero_b, void* workspace, nrc_stream_t stream);
int64_t nrc_ngp_train_query_clear_floats(int64_t M, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale,
                                         int64_t n_density_mlp_params, int64_t n_density_params);
int nrc_ngp_train_query_backward_cleared(const float* dL_dsigmas, const float* dL_drgbs, int64_t M, const float* x01,
                                         const void* density_weights_f16, const void* color_weights_f16, int32_t n_levels,
                                         int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale, const void* h_f16,
                                         const void* rgb_f16, const void* save_in_d, const void* save_acts_d, const void* save_in_c,
                                         const void* save_acts_c, float loss_scale, float* grad_density_params, float* grad_color_params,
                                         int64_t n_density_mlp_params, int64_t n_density_params, int64_t n_color_params, void* scratch,
                                         const int32_t* n_samples_dev, nrc_stream_t fork_stream, nrc_stream_t stream);
/* nrc_ngp_train_query_backward_cleared + nrc_amp_adam_step as ONE call WITHOUT the pass over the gradients: found_inf is known before the grid
 * backward starts -- the two network backward launches flag every inf / NaN they hand on (input gradients, weight-gradient sums: state4[0]), and the
 * hash-grid gradient is a finite-weighted sum of those -- so one thread settles step counter / bias corrections / scale (state4 as in
 * nrc_amp_adam_step) behind them, and the 49 MB read of nrc_amp_adam_step's check goes away (7 launches: 2 network backward, 1 settle, 3 grid
 * backward, 1 Adam over both vectors).  fork_stream (optional, also on nrc_ngp_train_query_backward_cleared): a second stream of the caller's on
 * which the dense levels' atomics run next to the hashed levels' split / accumulate; joined before the call returns.  Same arithmetic per parameter
 * as nrc_adam_step (csrc/adam_math.h). */
int nrc_ngp_train_backward_step(const float* dL_dsigmas, const float* dL_drgbs, int64_t M, const float* x01, const void* density_weights_f16,
                                const void* color_weights_f16, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution,
                                float per_level_scale, const void* h_f16, const void* rgb_f16, const void* save_in_d, const void* save_acts_d,
                                const void* save_in_c, const void* save_acts_c, float loss_scale, float* grad_density_params,
                                float* grad_color_params, int64_t n_density_mlp_params, int64_t n_density_params, int64_t n_color_params,
                                void* scratch, const int32_t* n_samples_dev, float* param_d, float* exp_avg_d, float* exp_avg_sq_d,
                                void* param_f16_d, float l2_coeff_d, int64_t l2_count_d, float* param_c, float* exp_avg_c, float* exp_avg_sq_c,
                                void* param_f16_c, float l2_coeff_c, int64_t l2_count_c, float lr, const float* lr_dev, float beta1, float beta2,
                                float eps, float weight_decay, int32_t adam_w_mode, int32_t* device_step, float* bias_corrections, float* scale,
                                int32_t* growth_tracker, float growth_factor, float backoff_factor, int32_t growth_interval, float* state4,
                                nrc_stream_t fork_stream, nrc_stream_t stream);
int nrc_amp_adam_step(float* param_a, const float* grad_a, float* exp_avg_a, float* exp_avg_sq_a, void* param_f16_a, int64_t n_a, float l2_coeff_a,
                      int64_t l2_count_a, float* param_b, const float* grad_b, float* exp_avg_b, float* exp_avg_sq_b, void* param_f16_b, int64_t n_b,
                      float l2_coeff_b, int64_t l2_count_b, float lr, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                      int32_t adam_w_mode, int32_t* device_step, float* bias_corrections, float* scale, int32_t* growth_tracker, float growth_factor,
                      float backoff_factor, int32_t growth_interval, float* state4, void* ticket, int32_t* skipped_steps, nrc_stream_t stream);

/* =====================================================================================================
 * Group 14 -- the training iteration of a DATA-PARALLEL rank, in pieces (round 6).  The reference has no counterpart (its DataParallel wrapper is a
 *            no-op for ray batches, src/Methods/Base/Renderer.py:24-33); SURVEY 8(e) names the shape: reduce-scatter of the gradients, Adam on the
 *            rank's 1/N of the parameters, all-gather of what the kernels read.  The collectives are the host's (torch.distributed over RCCL); these
 *            calls are what stands between them, enqueue-only like group 13:
 *   nrc_ngp_train_networks_backward : the two network launches of nrc_ngp_train_query_backward_cleared (same arguments); *nonfinite_flag (DEVICE f32,
 *            cleared by the caller) is set to 1 when a launch hands on an inf / NaN.  After it the MLP-weight gradients are final -- the caller's small
 *            all-reduce (both MLPs' gradients + the flag) can start while the grid backward runs.
 *   nrc_ngp_train_grid_backward     : the hash-grid backward of the same call (reads the input gradient the first call left in `scratch`).
 *   nrc_amp_settle                  : one thread: found_inf = (*flag_sum != 0) (a sum of the ranks' flags), step counter, bias corrections, scale update
 *            rule, state4[2] = 1 / (scale * grad_divisor) -- grad_divisor = world size when the gradients are a SUM over the ranks.  state4 as in nrc_amp_adam_step.
 *   nrc_wire_pack_f16 / nrc_wire_unpack_f16 : the optional 16-bit wire of the table's reduce-scatter -- f32 -> fp16 round-to-nearest, saturating at +-65504
 *            (*saturated, optional DEVICE u64, counts the clamped values; NaN stays NaN), and the reduced shard back to f32.  Halves the reduce-scatter's
 *            bytes; the sum over the ranks is then formed in fp16 (what tiny-cuda-nn's own fp16 gradient storage does under the same loss scale).  Off by default.
 *   nrc_amp_adam_slices             : the Adam launch of nrc_amp_adam_step on two arbitrary slices (pointers already offset; l2_count relative to the
 *            slice) -- a rank's shard of the table, or the replicated MLP weights.  Skipped when state4[1] says so.  Either slice may be empty.
 * ===================================================================================================== */
int nrc_ngp_train_networks_backward(const float* dL_dsigmas, const float* dL_drgbs, int64_t M, const float* x01, const void* density_weights_f16,
                                    const void* color_weights_f16, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution,
                                    float per_level_scale, const void* h_f16, const void* rgb_f16, const void* save_in_d, const void* save_acts_d,
                                    const void* save_in_c, const void* save_acts_c, float loss_scale, float* grad_density_params,
                                    float* grad_color_params, int64_t n_density_mlp_params, int64_t n_density_params, int64_t n_color_params,
                                    void* scratch, const int32_t* n_samples_dev, float* nonfinite_flag, nrc_stream_t stream);
int nrc_ngp_train_grid_backward(int64_t M, const float* x01, const void* density_weights_f16, const void* color_weights_f16, int32_t n_levels,
                                int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale, float* grad_density_params,
                                float* grad_color_params, int64_t n_density_mlp_params, int64_t n_density_params, int64_t n_color_params, void* scratch,
                                const int32_t* n_samples_dev, nrc_stream_t fork_stream, nrc_stream_t stream);
int nrc_wire_pack_f16(const float* src, void* dst_f16, int64_t n, uint64_t* saturated, nrc_stream_t stream);
int nrc_wire_unpack_f16(const void* src_f16, float* dst, int64_t n, nrc_stream_t stream);
int nrc_amp_settle(const float* flag_sum, float grad_divisor, float beta1, float beta2, int32_t* device_step, float* bias_corrections, float* scale,
                   int32_t* growth_tracker, float growth_factor, float backoff_factor, int32_t growth_interval, float* state4, int32_t* skipped_steps,
                   nrc_stream_t stream);
int nrc_amp_adam_slices(float* param_a, const float* grad_a, float* exp_avg_a, float* exp_avg_sq_a, void* param_f16_a, int64_t n_a, float l2_coeff_a,
                        int64_t l2_count_a, float* param_b, const float* grad_b, float* exp_avg_b, float* exp_avg_sq_b, void* param_f16_b, int64_t n_b,
                        float l2_coeff_b, int64_t l2_count_b, float lr, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                        int32_t adam_w_mode, const float* bias_corrections, const float* state4, nrc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFICG_HIP_H */
