/*
 * panfusion_hip.h -- C ABI of the MI355X-native (gfx950) PanFusion denoising hot path.
 *
 * Drop-in boundary.  The reference (chengzhag/PanFusion) is pure Python and has
 * no FFI of its own; the arithmetic on its denoising path lives in third-party CUDA wheels reached
 * through the call sites cited on every entry point below (paths relative to the reference tree).
 * This header is what a ctypes / cffi binding on the reference side binds (INTEGRATION.md shows
 * the stub); `panfusion_amd/_lib.py` is that binding for the in-tree Python host.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no C++/torch types.  All data pointers are DEVICE
 *     pointers owned by the caller unless the name says `host_`.  Scratch is passed in
 *     (`workspace`, size from the matching *_workspace_size); the library owns NO device memory
 *     (per-camera constants of the geometry calls travel by value in the kernel arguments).
 *   - All work is enqueued on `stream` (a hipStream_t passed as void*); no implicit device sync,
 *     no allocation, host arrays (`host_*`) are consumed before the call returns.
 *   - Activations are token-major / NHWC: [image][y][x][channel]; MFMA operands are 16-bit (PF_BF16 or
 *     PF_F16); in the mixed scheme the residual-stream tensors are PF_F32 (the entry points that take a
 *     stream tensor say so).  Statistics, biases, tables and accumulators are fp32.
 *   - Every function returns pf_status; on failure pf_last_error_string() describes the problem
 *     (thread-local).  Arguments are validated before any launch.  Nothing throws or aborts.
 */
#ifndef PANFUSION_HIP_H
#define PANFUSION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { PF_OK = 0, PF_ERR_ARG = 1, PF_ERR_LAUNCH = 2, PF_ERR_UNSUPPORTED = 3 } pf_status;
typedef enum { PF_BF16 = 0, PF_F16 = 1, PF_F32 = 2, PF_U8 = 3 } pf_dtype;

int pf_version(void);
const char* pf_last_error_string(void);

/* ------------------------------------------------------------------------------------------
 * Spherical geometry (replaces external/Perspective_and_Equirectangular/{e2p,p2e}.py and
 * models/pano/utils.py; angles in DEGREES as host doubles, one entry per camera).
 * ------------------------------------------------------------------------------------------ */

/* e2p.py:39-51 map_pers_pix_to_equi for `ncam` cameras: sampling position (pixel units of an
 * (eh,ew) panorama) of every pixel of an (h,w) view.  map_x/map_y: [ncam][h][w] fp32 (the
 * float64 result cast as e2p.py:74-75 does).  lonlat (optional, may be NULL): [ncam][h][w][2]
 * fp32 = e2p.py:9-36 map_pers_coords_to_equi (lon, lat) -- models/pano/utils.py:97-104. */
pf_status pf_e2p_grid(const double* host_fov, const double* host_theta, const double* host_phi,
                      int ncam, int eh, int ew, int h, int w,
                      float* map_x, float* map_y, float* lonlat, void* stream);

/* p2e.py:9-49 map_equi_pix_to_pers: for every pixel of an (h,w) panorama its position in the
 * (ph,pw) view (0 where invisible) and the visibility mask.  [ncam][h][w]. */
pf_status pf_p2e_grid(const double* host_fov, const double* host_theta, const double* host_phi,
                      int ncam, int ph, int pw, int h, int w,
                      float* map_u, float* map_v, uint8_t* mask, void* stream);

/* Dataset-side view cropping, external/py360convert/e2p.py:6-43 (called from utils/pano.py:160-161
 * Equirectangular.to_perspective, dataset/PanoDataset.py:133-140): `ncam` perspective crops (oh, ow) of ONE
 * equirectangular image img [H][W][C] (PF_U8 or PF_F32, channels last like the numpy array) -> out
 * [ncam][oh][ow][C], same dtype.  Per camera (host arrays, DEGREES): horizontal / vertical field of view,
 * yaw u, pitch v, in-plane rotation (may be NULL = 0).  order 0 = nearest, 1 = bilinear; sampling is
 * scipy.ndimage.map_coordinates(mode='wrap') on the pole-padded image (utils.py:120-133), restated bit for bit
 * (double accumulation; PF_U8 rounds half up and clamps like scipy's integer outputs). */
pf_status pf_py360_e2p(const void* img, int dtype, int H, int W, int C, const double* host_hfov,
                       const double* host_vfov, const double* host_u, const double* host_v,
                       const double* host_in_rot, int ncam, int oh, int ow, int order, void* out,
                       void* stream);

/* Integer gather indices of kornia.remap(mode='nearest', align_corners=True) -> F.grid_sample
 * (e2p.py:76): idx[i] = iy*src_w+ix or -1 when the sample falls outside.  n entries. */
pf_status pf_nearest_indices(const float* map_x, const float* map_y, long n, int src_h, int src_w,
                             int32_t* idx, void* stream);

/* kornia.remap == grid_sample(zeros padding, align_corners=True) on NCHW images (e2p.py:76,
 * p2e.py:70-71).  src [B][C][hs][ws], maps [map_batch][ho][wo] (map_batch == B or 1), dst
 * [B][C][ho][wo]; mode 0 = nearest, 1 = bilinear; mask (optional) [map_batch][ho][wo]
 * multiplies the result (p2e.py:71).  dtype: PF_F32 / PF_F16 / PF_BF16 (same in and out). */
pf_status pf_remap(const void* src, int dtype, int B, int C, int hs, int ws,
                   const float* map_x, const float* map_y, const uint8_t* mask, int map_batch,
                   int ho, int wo, int mode, void* dst, void* stream);

/* models/pano/utils.py:92-95: (lon,lat) of every pixel of an (H,W) panorama, [H][W][2] fp32. */
pf_status pf_equi_coords(int H, int W, float* lonlat, void* stream);

/* models/modules/transformer.py:185-201 SphericalPE.forward: coords [n][2] fp32, freq_bands
 * [nfreq] fp32 -> out [n][4*nfreq] fp32 = [sin(lon f) | sin(lat f) | cos(lon f) | cos(lat f)].
 * Full-range accurate sinf/cosf (frequencies reach 2^79). */
pf_status pf_spherical_pe(const float* coords, long n, const float* freq_bands, int nfreq,
                          float* out, void* stream);

/* models/pano/utils.py:10-84 get_masks, restated as shift-invariant tables: bias_x = mask + 1
 * in [0,2] (softmax is shift invariant per query row, and 98-99% of the entries are exactly 0).
 *   bias_e [E][m*P]   panorama pixel queries, keys ordered (view, y, x)   (modules.py:47)
 *   bias_p [m*P][E]   view pixel queries                                   (modules.py:53)
 *   flags_e [ceil(E/32)][ceil(m*P/32)], flags_p [ceil(m*P/32)][ceil(E/32)]: 1 where the 32x32
 *   tile of the table holds any non-zero.  E = eh*ew, P = ph*pw.  m = ncam. */
size_t pf_epa_tables_workspace_size(int ncam, int ph, int pw, int eh, int ew);
pf_status pf_epa_tables_build(const double* host_fov, const double* host_theta,
                              const double* host_phi, int ncam, int ph, int pw, int eh, int ew,
                              float* bias_e, float* bias_p, uint8_t* flags_e, uint8_t* flags_p,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Normalisation / element-wise (replace torch GroupNorm / LayerNorm / SiLU / GELU / F.pad /
 * torch.cat / torch.roll kernels reached from diffusers ResnetBlock2D, Transformer2DModel and
 * models/modules/transformer.py:151-162).
 * ------------------------------------------------------------------------------------------ */

/* GroupNorm statistics folded with the affine: for x [n_img][hw][C] (C = c0+c1 when two
 * sources are concatenated along channels, MVGenModel.py:231-233) writes scale/shift
 * [n_img][C] fp32 such that GN(x) = x*scale + shift.  workspace: pf_groupnorm_workspace_size.
 * dtype (both sources): PF_BF16 / PF_F16, or PF_F32 for the fp32 residual stream of the mixed scheme. */
size_t pf_groupnorm_workspace_size(int n_img, int hw, int C);
pf_status pf_groupnorm_stats(const void* x0, int c0, const void* x1, int c1, int dtype,
                             int n_img, int hw, int groups, float eps,
                             const float* gamma, const float* beta,
                             float* scale, float* shift, void* workspace, size_t workspace_bytes,
                             void* stream);

/* GroupNorm statistics of the tensor x [n_img][h][w][C] as if its width had been padded circularly by wrap_pad columns first
 * (the first and last wrap_pad columns count twice): what GroupNorm sees inside pad_pano(x, 2) -> ResnetBlock2D -> unpad_pano
 * of the panorama branch (MVGenModel.py:110-115), without the padded copy.  Workspace as pf_groupnorm_stats (hw = h * w). */
pf_status pf_groupnorm_stats_wrap(const void* x0, int c0, const void* x1, int c1, int dtype,
                                  int n_img, int h, int w, int wrap_pad, int groups, float eps,
                                  const float* gamma, const float* beta,
                                  float* scale, float* shift, void* workspace, size_t workspace_bytes,
                                  void* stream);

/* The same scale / shift from the per-column-pair moments a pf_conv_gemm epilogue left in pf_conv_desc.gn_partial: source s
 * is fp32 [n_img * hw / rows_s][2][c_s / 2] (rows_s = pf_conv_gemm_gn_rows of the launch that produced it); part1 = NULL: one
 * source.  (c0 + c1) / groups, c0 and c1 must be even.  No pass over the activation itself. */
pf_status pf_groupnorm_from_partials(const float* part0, int c0, int rows0, const float* part1, int c1, int rows1,
                                     int n_img, int hw, int groups, float eps, const float* gamma, const float* beta,
                                     float* scale, float* shift, void* stream);

/* y = act(x*scale + shift), act 0 = identity, 1 = SiLU (scale = shift = NULL: y = act(x)).  Same concat
 * convention.  dtype = type of the sources (16-bit or PF_F32).  Output:
 *   out_dtype 16-bit, out_split 0:  y [n_img][hw][C]
 *   out_dtype 16-bit, out_split 1:  y [n_img][hw][2C]: per block of 32 channels [hi(32) | lo(32)], hi = round16(v),
 *                                   lo = round16(v - hi) (C %% 32 == 0): the A operand of a split-precision GEMM
 *                                   (pf_conv_desc.split3: A_hi W_hi + A_lo W_hi + A_hi W_lo reproduce the fp32
 *                                   product to ~2^-22)
 *   out_dtype PF_F32:               y [n_img][hw][C] fp32. */
pf_status pf_scale_shift_act(const void* x0, int c0, const void* x1, int c1, int dtype,
                             int n_img, int hw, const float* scale, const float* shift, int act,
                             int out_dtype, int out_split, void* y, void* stream);
/* fp32 sources, 16-bit y as above, PLUS the un-normalised input as the split pair raw_pair [n_img][hw][2C] (same layout)
 * in the same pass (the A operand of a ResnetBlock2D's split-precision conv_shortcut next to norm1 + SiLU of the same
 * tensor: one read of the fp32 stream instead of two). */
pf_status pf_scale_shift_act_pair(const void* x0, int c0, const void* x1, int c1, int n_img, int hw,
                                  const float* scale, const float* shift, int act, int out_dtype, void* y,
                                  void* raw_pair, void* stream);

/* y = LayerNorm(x + pe) * gamma + beta over the last dim; x,y [rows][C]; pe (optional) fp32
 * [pe_rows][C], row r uses pe row (r % pe_rows) (transformer.py:155-158; eps 1e-5).
 * dtype: type of x (16-bit == out_dtype, or PF_F32); out_dtype: 16-bit type of y. */
pf_status pf_layernorm(const void* x, const float* pe, long pe_rows, int dtype, long rows, int C,
                       const float* gamma, const float* beta, float eps, int out_dtype, void* y, void* stream);

/* Sampling of the VAE posterior (diffusers DiagonalGaussianDistribution.sample, PanoGenerator.py:214-225):
 * moments fp32 NHWC [n][hw][2L] = (mean | logvar), eps fp32 NCHW [n][L][hw] ->
 * z NCHW [n][L][hw] = (mean + exp(0.5 * clamp(logvar, -30, 20)) * eps) * scale. */
pf_status pf_vae_sample(const float* moments, const float* eps, int n, int L, long hw, float scale, float* z, void* stream);

/* out = a * x + b * y on n fp32 elements (scheduler.add_noise of the training step, PanFusion.py:83-84:
 * sqrt(abar_t) * latents + sqrt(1 - abar_t) * noise). */
pf_status pf_axpby(const float* x, const float* y, float a, float b, long n, float* out, void* stream);

/* GEGLU: in [rows][2*inner] = [a | gate] -> out [rows][inner] = a * gelu(gate) (erf GELU). */
pf_status pf_geglu(const void* in, int dtype, long rows, int inner, void* out, void* stream);

/* Sinusoidal timestep features [cos | sin] (diffusers Timesteps, flip_sin_to_cos, shift 0)
 * followed by nothing: t [n] int64 -> out [n][dim] in `dtype`. */
pf_status pf_timestep_features(const int64_t* t, int n, int dim, int out_dtype, void* out, void* stream);
/* The same with the n timesteps t_stride words apart: `timestep[:, 0]` of the (b, m) tensor the reference hands the panorama
 * branch (MVGenModel.py:49-50, :85) read in place -- no strided-copy kernel in the step. */
pf_status pf_timestep_features_strided(const int64_t* t, long t_stride, int n, int dim, int out_dtype, void* out, void* stream);

/* y = silu(x) element-wise, n elements. */
pf_status pf_silu(const void* x, int dtype, long n, void* y, void* stream);

/* Circular width pad (utils/pano.py:74-99) and crop (:102-105) on NHWC: x [n][h][w][C], 16-bit or PF_F32. */
pf_status pf_pad_width(const void* x, int dtype, int n, int h, int w, int C, int pad, void* y, void* stream);
pf_status pf_crop_width(const void* x, int dtype, int n, int h, int w, int C, int crop, void* y, void* stream);

/* The reference's pad_pano on NCHW tensors (utils/pano.py:74-99): x [rows][w] -> y [rows][w+2 pad],
 * circular; elem_bytes 2 or 4. */
pf_status pf_pad_width_rows(const void* x, int elem_bytes, long rows, int w, int pad, void* y, void* stream);
/* torch.roll(x, shift, dims=-1) of rotate_latent (PanoGenerator.py:264-269): y[r][(i+shift) mod w] = x[r][i]. */
pf_status pf_roll_width_rows(const void* x, int elem_bytes, long rows, int w, int shift, void* y, void* stream);

/* Layout/precision converters at the boundary: NCHW (src_dtype) <-> NHWC (dst_dtype). */
pf_status pf_nchw_to_nhwc(const void* x, int src_dtype, int n, int C, int h, int w, int dst_dtype, void* y, void* stream);
pf_status pf_nhwc_to_nchw(const void* x, int src_dtype, int n, int C, int h, int w, int dst_dtype, void* y, void* stream);

/* y = a + b element-wise (ControlNet residual adds, MVGenModel.py:154-170,200-203); a and y of type
 * dtype_a, b of type dtype_b (any of PF_BF16 / PF_F16 / PF_F32). */
pf_status pf_add(const void* a, int dtype_a, const void* b, int dtype_b, long n, void* y, void* stream);

/* Token + position embedding of the CLIP text encoder (transformers CLIPTextEmbeddings, reached from
 * PanoGenerator.py:197-211 encode_text): ids int64 [B][L] -> out [B][Lp][C] (out_dtype 16-bit or PF_F32),
 * out[b][t] = tok[ids[b][t]] + pos[t] for t < L, zero rows for L <= t < Lp (the sequence is padded to a
 * multiple of 4 keys for the masked attention).  tok [vocab][C], pos [>= L][C] fp32. */
pf_status pf_embed_tokens(const int64_t* ids, int B, int L, int Lp, int C, long vocab, const float* tok,
                          const float* pos, int out_dtype, void* out, void* stream);

/* Row softmax of fp32 scores: probs[r][j] = exp(scale (s[r][j] - max_j)) / sum_j, 16-bit out.  The VAE decoder's
 * mid-block attention has ONE head of width 512 (diffusers Attention in AutoencoderKL, reached from
 * PanoGenerator.py:213-220 decode_latent): its scores and the P.V product run on pf_conv_gemm (batched), this
 * kernel sits between them.  scores [rows][scores_ld] fp32, probs [rows][probs_ld] (out_dtype 16-bit). */
pf_status pf_softmax_rows(const float* scores, long rows, int n, long scores_ld, float scale, int out_dtype,
                          void* probs, long probs_ld, void* stream);

/* models/modules/utils.py:9-15 tensor_to_image: x fp32 NCHW in [-1, 1] -> y uint8 NHWC,
 * round((x / 2 + 0.5).clamp(0, 1) * 255) (half to even, like torch.round). */
pf_status pf_tensor_to_image(const float* x, int n, int C, int h, int w, uint8_t* y, void* stream);

/* Fused classifier-free-guidance merge + DDIM update (+ optional width roll of the result):
 * eps = eps_uncond + g*(eps_cond - eps_uncond)                (PanoGenerator.py:253-262)
 * x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t); x' = sqrt(a_prev) x0 + sqrt(1-a_prev) eps  (DDIM eta=0)
 * out[..., (w + roll) mod W] = x'                              (PanoGenerator.py:264-269)
 * x, out: fp32 [n_img][C][H][W]; eps_uncond/eps_cond fp32 same shape. */
pf_status pf_cfg_ddim_step(const float* x, const float* eps_uncond, const float* eps_cond,
                           float guidance, float sqrt_a_t, float sqrt_1m_a_t, float sqrt_a_prev,
                           float sqrt_1m_a_prev, long rows, int W, int roll, float* out, void* stream);

/* The same update as ONE launch per latent with everything the NEXT denoiser call needs (round 5: no torch.cat / copy_ / fill_
 * kernels between two calls of the loop, PanFusion.py:149-162): out may alias x for ANY roll (a block owns whole rows);
 * out2 (or NULL) receives a second copy -- the other half of the CFG pair torch.cat([x] * 2) of PanoGenerator.py:240-251;
 * tstep (or NULL): n_tstep int64 words set to t_next, the timestep tensor of the next call (PanFusion.py:147).
 * W <= 16384; out2 must not alias x / out. */
pf_status pf_cfg_ddim_step_pair(const float* x, const float* eps_uncond, const float* eps_cond,
                                float guidance, float sqrt_a_t, float sqrt_1m_a_t, float sqrt_a_prev,
                                float sqrt_1m_a_prev, long rows, int W, int roll, float* out, float* out2,
                                int64_t* tstep, int n_tstep, int64_t t_next, void* stream);

/* ------------------------------------------------------------------------------------------
 * MFMA GEMM / implicit-GEMM convolution (replaces cuDNN/cuBLAS behind diffusers Conv2d/Linear:
 * MVGenModel.py:86-144,174-198,224-294 and transformer.py:57-74,8-38).
 *   out[m][n] = sum_k A[m][k] * W[n][k]  (+ bias[n]) (+ rowvec[img(m)][n]) (+ residual[m][n])
 * A is gathered on the fly: m = (img, yo, xo), k = (ky, kx, c); zero padding; optional nearest
 * x2 upsampling of the input (Upsample2D) and optional channel concat of two sources.
 * A plain linear layer is ksize = 1, n_img = 1, h_in = 1, w_in = rows.
 * Requirements: (c0 + c1) % 64 == 0, c0 % 64 == 0; pointers 16-byte aligned; ld's % 8 == 0.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const void* a0;      /* source 0, [n_img][h_in][w_in][a0_ld]                            */
    const void* a1;      /* source 1 (channel concat after source 0) or NULL                */
    int c0, c1;          /* channels taken from each source                                 */
    int a0_ld, a1_ld;    /* per-pixel stride in elements (>= c0 / c1)                       */
    int n_img, h_in, w_in;
    int h_out, w_out;    /* output spatial size: (h + 2 pad - k) / stride + 1, or that of the input with ONE more zero
                          * row and column at the bottom / right (F.pad(x, (0,1,0,1)) + conv, the VAE encoder's
                          * Downsample2D(padding=0)): pixels past the input read as zero                       */
    int ksize;           /* 1 or 3                                                          */
    int stride;          /* 1 or 2                                                          */
    int pad;             /* 0 or 1 (zeros)                                                  */
    int upsample;        /* 1: input is nearest-upsampled x2 before the convolution         */
    const void* w;       /* [n_out][ksize*ksize*(c0+c1)], k ordered (ky, kx, c)             */
    int n_out;
    const float* bias;   /* [n_out] or NULL                                                 */
    const float* rowvec; /* [n_img][rowvec_ld] added per image (time embedding) or NULL     */
    int rowvec_ld;
    const void* residual;/* [M][res_ld] (res_dtype) or NULL                                 */
    int res_ld;
    int res_dtype;       /* == dtype, or PF_F32: the fp32 residual stream of the mixed scheme */
    void* out;           /* [M][out_ld]                                                     */
    int out_ld;
    int out_dtype;       /* PF_BF16 / PF_F16 (== dtype) or PF_F32                           */
    int dtype;           /* PF_BF16 or PF_F16: A, W, residual                               */
    int batch;           /* >= 1: independent problems, strides below (elements)            */
    long a_bstride, w_bstride, out_bstride, res_bstride;
    int epilogue;        /* PF_EPILOGUE_NONE; PF_EPILOGUE_GEGLU: W rows interleaved (value_j, gate_j),
                          * out [M][n_out/2] = value * gelu(gate) (transformer.py:8-21, erf GELU);
                          * PF_EPILOGUE_SPLIT: out is the 16-bit pair [M][2 n_out] of the fp32 result, per block of 32
                          * columns [hi(32) | lo(32)] (hi = round16(v), lo = round16(v - hi)), out_ld >= 2 n_out,
                          * n_out %% 32 == 0: the A operand of a following split3 GEMM without a separate split pass */
    void* workspace;     /* split-K scratch (may be NULL: no split) of pf_conv_gemm_workspace_size   */
    size_t workspace_bytes;
    float* gn_partial;   /* optional by-product (NULL: none): GroupNorm moments of the OUTPUT for the next layer's
                          * norm (diffusers ResnetBlock2D.norm1 / norm2, Transformer2DModel.norm; call sites
                          * MVGenModel.py:102-144,174-198,224-277), so that no statistics pass re-reads the tensor:
                          * fp32 [M / R][2][n_out / 2] = (sum, sum of squares) of each column PAIR (2 k, 2 k + 1) over each run of R output rows,
                          * R = pf_conv_gemm_gn_rows(desc) (> 0: possible for this problem; images are whole runs).
                          * Consumed by pf_groupnorm_from_partials.  Fixed summation order, no atomics.      */
    int wrap_pad;        /* 0..2: the input is read as if its WIDTH had been padded circularly by wrap_pad columns on both
                          * sides first (pad_pano, utils/pano.py:74-99), in pre-upsampling columns; the zero padding `pad`
                          * applies outside that virtual tensor.  No padded copy exists.                              */
    int crop;            /* 0..2: output columns cropped by `crop` on both sides (unpad_pano, utils/pano.py:102-105):
                          * w_out = ((w_in + 2 wrap_pad) << upsample + 2 pad - ksize) / stride + 1 - 2 crop.
                          * Together: pad_pano(x, p) -> conv -> unpad_pano(., c) of the panorama branch in one launch
                          * (MVGenModel.py:110-115 resnets p = c = 2 [conv1: wrap 2 / crop 0, conv2: wrap 0 / crop 2 on the
                          * padded intermediate], :138-144 down-sampling 2 / 1, :272-277 up-sampling 1 / 2).           */
    int* tickets;        /* optional (NULL: the split-K slabs are combined by a second kernel): n_tickets int32 arrival counters,
                          * ALL ZERO when the call is enqueued and zero again when it has run, not shared with a launch that may
                          * run concurrently -- the K-slice workgroup that arrives last combines the slabs inside the launch
                          * (same sums in split order as the second kernel: bit-identical).  n_tickets >= tiles x batch of the
                          * split launch (<= 1024 for every plan this library makes).                                  */
    int n_tickets;
    int split3;          /* split-precision walk: a0 is a PAIR tensor -- per block of 32 channels [hi(32) | lo(32)], c0 = 2 x channels (what
                          * pf_scale_shift_act(out_split) / PF_EPILOGUE_SPLIT write) -- and w holds per tap and block of 32 channels
                          * [W_hi(32) | W_lo(32)]; every 64-element K block is multiplied as W_hi A_hi + W_hi A_lo + W_lo A_hi (2^-22
                          * relative: an fp32-grade product from 16-bit MFMA operands).  a1 must be NULL.                       */
    int subpixel;        /* 1: a nearest-x2-upsampling 3x3 convolution (ksize 3, upsample 1, stride 1, pad 1; diffusers Upsample2D:
                          * F.interpolate(scale_factor=2, mode="nearest") + Conv2d, reference call sites MVGenModel.py:272-277) computed as FOUR 2x2
                          * convolutions on the LOW-resolution grid, one per output phase (a, b) in {0, 1}^2: output pixel (2 y + a, 2 x + b) only
                          * ever reads input rows y - 1 + a, y + a and columns x - 1 + b, x + b, so the nine taps collapse to four with summed
                          * weights -- 4 instead of 9 MACs per (output value, input channel), the same sums in exact arithmetic.  `w` then holds
                          * [4 phases][n_out][2][2][c0 + c1] (phase-major, 16-bit; panfusion_amd.engine._subpixel_weight builds it: row taps of
                          * phase a = 0: {W[0]}, {W[1] + W[2]}; a = 1: {W[0] + W[1]}, {W[2]}; columns alike).  Bias only (no row vector,
                          * residual, GEGLU / pair epilogue; split3 allowed: c0 = 2 x channels per tap), batch 1, h_out / w_out even; wrap_pad 0..1 with crop 2 x wrap_pad is the
                          * panorama's pad 1 / upsample / conv / crop 2.  gn_partial is supported (runs of R LOW-resolution rows).            */
} pf_conv_desc;

enum { PF_EPILOGUE_NONE = 0, PF_EPILOGUE_GEGLU = 1, PF_EPILOGUE_SPLIT = 2 };

/* Bytes of fp32 scratch that let pf_conv_gemm split the K range of a problem whose output grid alone
 * cannot fill the 256 CUs (the 8x8 / 16x16 levels and the whole panorama branch); 0 = not wanted.
 * The slabs are summed in split order by a second kernel: results do not depend on scheduling. */
size_t pf_conv_gemm_workspace_size(const pf_conv_desc* desc);
/* Rows per moment run R of pf_conv_desc.gn_partial for this problem, or 0 when the kernel that serves it cannot emit
 * the moments (split-K plans, batches, GEGLU / pair epilogues, images that are not whole runs of R rows). */
int pf_conv_gemm_gn_rows(const pf_conv_desc* desc);
/* Diagnostics / tests / the benchmark's shape classes: which tile kernel pf_conv_gemm's plan gives this problem --
 * 0 the 4-wave 16x16x32 kernel (2 blocks per CU), 1 the persistent 8-wave 16x16x32 kernel (256 x 160 tiles),
 * 2 the persistent 32x32x16 kernel of round 6 (256 x 320 tiles, pf_gemm32.hip); -1 for a bad descriptor.  Like the two
 * queries above it reflects the plan only (workspace assumed available), it validates nothing.  The cuDNN / cuBLAS
 * heuristics behind diffusers Conv2d / Linear are what it stands in for (MVGenModel.py:102-144,174-198,224-277). */
int pf_conv_gemm_kernel_id(const pf_conv_desc* desc);
/* Diagnostics only: while `device_buffer` (capacity_blocks x 32 uint64) is set, every pf_conv_gemm launch
 * with at most capacity_blocks workgroups records 4 shader-clock stamps per workgroup (entry, first
 * operand tile landed, K loop done, exit; the 8-wave kernel adds per-wave K-loop time split into
 * wait+barrier / DMA issue / ds_read+MFMA at [4 + 3*wave + {0,1,2}]).  NULL switches it off (the default). */
pf_status pf_debug_gemm_profile(void* device_buffer, long capacity_blocks);
pf_status pf_conv_gemm(const pf_conv_desc* desc, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight-stationary linear for the token layers of the two finest levels: the nn.Linear layers of their transformer blocks
 * and EPA blocks (models/modules/transformer.py:57-74 to_q / to_k / to_v / to_out, :8-38 GEGLU FeedForward; diffusers
 * BasicTransformerBlock behind models/pano/MVGenModel.py:104-106).  Two shapes:
 *   K == 320, N a multiple of 320 (every mode): a workgroup keeps 320 output channels of the weights in registers and streams
 *             64-token tiles of `a` through LDS;
 *   K == 640, N a multiple of 256 (PF_LWS_16, PF_LWS_GEGLU: q | k and FF1 of the 32^2 level): 256 channels per workgroup,
 *             32-token tiles; N a multiple of 128 only (PF_LWS_16: to_q, N = 640): 128 channels per workgroup;
 *   K == 1280, N a multiple of 128 (PF_LWS_16, PF_LWS_GEGLU: the 16^2 level): 128 channels per workgroup, 16-token tiles.
 *   PF_LWS_16:    out 16-bit [M][out_ld]      = a w^T + bias
 *   PF_LWS_F32:   out fp32   [M][out_ld]      = a w^T + bias + residual (fp32 [M][res_ld] or NULL)
 *   PF_LWS_GEGLU: out 16-bit [M][out_ld], N/2 columns: w / bias rows interleaved (value_j, gate_j), out = value * gelu(gate)
 *   PF_LWS_QKV:   N == 960 = (q | k | v): out 16-bit [M][out_ld] receives the 640 columns (q | k); V leaves TRANSPOSED as
 *                 out_vt[b][c][key] (b = m / rows_per_batch, key = m % rows_per_batch, row stride vt_ld, batch stride vt_bs):
 *                 the layout pf_attention reads.  rows_per_batch a multiple of 64.
 *   PF_LWS_VT:    the whole output TRANSPOSED to out_vt[b][n][key] as in PF_LWS_QKV (the V projection of the 32^2 / 16^2 self-attentions and of
 *                 the EPA blocks: replaces the operand-swapped pf_conv_gemm launch); out unused.  K == 320: N a multiple of 320; K == 640 / 1280:
 *                 N a multiple of 128.  rows_per_batch a multiple of 64 / 32 / 16.
 *   PF_LWS_F32_LN: N == 320: PF_LWS_F32, and the LayerNorm of the result rides along (the norm2 / norm3 that follow the two
 *                 attention output projections of a BasicTransformerBlock): ln_out 16-bit [M][ln_ld] =
 *                 LayerNorm(out; ln_eps) * ln_gamma + ln_beta -- no separate pass over the stream tensor.
 * a 16-bit [M][a_ld]; w 16-bit [N][K]; bias fp32 [N] or NULL.  pf_linear_ws_supported: 1 if (M, N, K, mode) is served. */
typedef struct {
    const void* a; int a_ld;
    const void* w;
    const float* bias;
    const float* residual; int res_ld;
    void* out; int out_ld;
    void* out_vt; int vt_ld; int rows_per_batch; long vt_bs;
    const float* ln_gamma; const float* ln_beta; float ln_eps; void* ln_out; int ln_ld;
    int M, N, K;
    int dtype; int mode;
} pf_linear_ws_desc;
enum { PF_LWS_16 = 0, PF_LWS_F32 = 1, PF_LWS_GEGLU = 2, PF_LWS_QKV = 3, PF_LWS_F32_LN = 4, PF_LWS_VT = 5 };
int pf_linear_ws_supported(long M, int N, int K, int mode);
pf_status pf_linear_ws(const pf_linear_ws_desc* desc, void* stream);

/* 3x3 convolutions with 4 input or 4 output channels at the UNet boundary:
 * conv_in  (MVGenModel.py:86,89): x fp32 NCHW [n][cin][h][w] -> y NHWC [n][h][w][cout] (out_dtype 16-bit
 *          or PF_F32), weights fp32 [3][3][cin][cout];
 * conv_out (MVGenModel.py:283,292): x NHWC [n][h][w][cin] (dtype 16-bit or PF_F32) -> y fp32 NCHW
 *          [n][cout][h][w], weights fp32 [cout][3][3][cin], cout <= 8.
 * bias fp32 [cout]; zero padding 1 in y.  wrap = 1: the width axis is circular, which is exactly
 * pad_pano(x,1) -> conv(zero pad) -> unpad_pano(.,1) of the pano branch (MVGenModel.py:87-91,
 * 290-294); wrap = 0: zero padding in x. */
pf_status pf_conv_in(const float* x, int n, int cin, int h, int w, const float* wgt, const float* bias,
                     int cout, int wrap, int out_dtype, void* y, void* stream);
pf_status pf_conv_out(const void* x, int dtype, int n, int cin, int h, int w, const float* wgt,
                      const float* bias, int cout, int wrap, float* y, void* stream);
/* The UNets' head in one launch (MVGenModel.py:279-294: conv_norm_out -> SiLU -> conv_out): x fp32 NHWC
 * [n][h][w][cin] BEFORE the GroupNorm, scale / shift fp32 [n][cin] from pf_groupnorm_* (y = act(x * scale + shift),
 * act 1 = SiLU), wgt_t fp32 [3][3][cin][4] (the conv_out weight transposed, output channels padded to 4 with
 * zeros), cout <= 4, cin % 32 == 0; y fp32 NCHW [n][cout][h][w].  Zero padding / wrap apply to the ACTIVATED
 * tensor, as in pf_conv_out. */
pf_status pf_conv_out_gn(const float* x, int n, int cin, int h, int w, const float* scale, const float* shift, int act,
                         const float* wgt_t, const float* bias, int cout, int wrap, float* y, void* stream);

/* ------------------------------------------------------------------------------------------
 * Flash attention on MFMA (replaces xformers memory_efficient_attention, transformer.py:71,
 * and the diffusers AttnProcessor baddbmm+softmax+bmm inside unet.*.attentions[j]).
 *   out[b][i][h*D + :] = softmax_j(scale * q_i.k_j + bias[i][j]) v_j      D in {32, 64}
 * q [B][nq][q_ld], k [B][nk][k_ld] (head h at column h*D), vt = V transposed:
 * [B][H*D][vt_ld] (keys contiguous), out [B][nq][o_ld].  bias (optional) fp32 [nq][bias_ld]
 * shared by all batches and heads, consulted only for 32x32 tiles whose flag byte is non-zero.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const void* q; const void* k; const void* vt; void* out;
    int dtype;
    int B, H, D;
    int nq, nk;
    int q_ld, k_ld, vt_ld, o_ld;
    long q_bs, k_bs, vt_bs, o_bs;
    float scale;
    const float* bias; long bias_ld;
    const uint8_t* flags; int flags_ld;
    float* lse;      /* optional fp32 [B][H][nq]: log2 of the softmax denominator in the exp2 domain,
                      * lse = log2(sum_j exp2(log2e * (scale * q.k_j + bias_j))) -- what pf_attention_bwd reads */
    void* workspace; size_t workspace_bytes; /* optional scratch of pf_attention_workspace_size(desc) bytes (round 6): lets a launch with FEW query
                      * blocks and MANY keys (the panorama-query direction of an EPA block: 2048 queries x 20 480 keys, 320-640 workgroups for 1024
                      * slots) split its key range over several workgroups -- normalised 16-bit partial outputs + their log-sum-exps, combined
                      * by a second launch in a fixed order (no atomics).  NULL: never split. */
} pf_attn_desc;

pf_status pf_attention(const pf_attn_desc* desc, void* stream);
/* Bytes of pf_attn_desc.workspace that let pf_attention split the key range of this problem (0: it would not split). */
size_t pf_attention_workspace_size(const pf_attn_desc* desc);

/* ------------------------------------------------------------------------------------------
 * Training: backward of the EPA block (reference models/pano/modules.py:15-59 under autograd;
 * models/modules/transformer.py:77-161 -- the reference recomputes the block in backward,
 * CheckpointFunction, and so does the caller of these entry points).
 * ------------------------------------------------------------------------------------------ */

/* delta[b][h][i] = sum_d dout[b][i][h*D+d] * out[b][i][h*D+d]   (fp32 [B][H][nq]; rows of width ld, batch stride bs). */
pf_status pf_attention_delta(const void* out, const void* dout, int dtype, int B, int H, int D, long nq,
                             int ld, long bs, float* delta, void* stream);

/* Backward of pf_attention given lse (forward) and delta:
 *   P = exp2(log2e * (scale * q.k + bias) - lse),  dV = P^T dO,  dS = P o (dO V^T - delta),
 *   dQ = scale * dS K,  dK = scale * dS^T Q.
 * All of q, k, v, dout are ROW-major ([B][n][ld], head h at column h*D); kt, qt, dot are the transposes of k, q, dout
 * ([B][H*D][*_ld], tokens contiguous) -- the A operands of the products whose reduction runs over tokens.
 * dq [B][nq][dq_ld], dk / dv [B][nk][dk_ld / dv_ld] in `dtype`.  Token counts that are not multiples of 32 take a
 * guarded instantiation (the 4x4 level of a 256^2 view has 16 tokens); with a bias nk % 4 == 0.  Two launches: queries-stationary (dq) and keys-stationary (dk, dv)
 * (+ a reduce launch when the query range was split, see `workspace`); no atomics, results do not depend on scheduling. */
typedef struct {
    const void* q; const void* k; const void* v; const void* dout;
    const void* qt; const void* kt; const void* dot;
    void* dq; void* dk; void* dv;
    int dtype;
    int B, H, D;
    int nq, nk;
    int q_ld, k_ld, v_ld, do_ld;          /* row-major operands */
    int qt_ld, kt_ld, dot_ld;             /* transposed operands */
    int dq_ld, dk_ld, dv_ld;
    long q_bs, k_bs, v_bs, do_bs, qt_bs, kt_bs, dot_bs, dq_bs, dk_bs, dv_bs;
    float scale;
    const float* bias; long bias_ld;      /* as pf_attention: [nq][bias_ld] */
    const uint8_t* flags; int flags_ld;
    const float* lse; const float* delta; /* fp32 [B][H][nq] */
    void* workspace; size_t workspace_bytes; /* optional fp32 scratch of pf_attention_bwd_workspace_size(desc) bytes: lets the
                                              * keys-stationary launch split its query range when it has few blocks (the 128
                                              * text keys of a cross-attention, one panorama sample); partial sums are added in a
                                              * fixed order.  NULL: never split. */
} pf_attn_bwd_desc;

size_t pf_attention_bwd_workspace_size(const pf_attn_bwd_desc* desc);
pf_status pf_attention_bwd(const pf_attn_bwd_desc* desc, void* stream);

/* LayerNorm backward (rows of width C <= 2048, C % 8 == 0).  x (+ pe, as pf_layernorm) is the forward input, dy fp32
 * [rows][C] the gradient of the normalised output, dres (optional fp32 [rows][C]) a gradient that by-passes the norm:
 *   dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma.
 * partials: fp32 [2][n_part][C] -- per-block sums of dy * xhat (gamma) and dy (beta) over the block's rows, reduced in
 * a fixed order by pf_colsum afterwards; n_part = pf_layernorm_bwd_parts(rows). */
int pf_layernorm_bwd_parts(long rows);
pf_status pf_layernorm_bwd(const void* x, const float* pe, long pe_rows, int dtype, long rows, int C,
                           const float* gamma, float eps, const float* dy, const float* dres, float* dx,
                           float* partials, void* stream);

/* GroupNorm (+ SiLU) backward for the forward y = act(x * scale + shift) of pf_groupnorm_stats + pf_scale_shift_act:
 * x = channel concat of x0 [n][hw][c0] and x1 [n][hw][c1] (16-bit or PF_F32), scale / shift fp32 [n][c0+c1] as the forward
 * produced them, dy fp32 [n][hw][c0+c1], dres optional fp32 gradient of the same shape added to the result;
 * dx0 [n][hw][c0], dx1 [n][hw][c1] fp32.  gamma / beta take no gradient here (frozen UNet).  act: 0 none, 1 SiLU. */
size_t pf_groupnorm_bwd_workspace_size(int n_img, int hw, int groups);
pf_status pf_groupnorm_bwd(const void* x0, int c0, const void* x1, int c1, int dtype, int n_img, int hw, int groups,
                           float eps, const float* gamma, const float* scale, const float* shift, int act,
                           const float* dy, const float* dres, float* dx0, float* dx1, void* workspace,
                           size_t workspace_bytes, void* stream);

/* Parameter gradients of a TRAINABLE GroupNorm (+ SiLU): the ControlNet's own parameters train when layout conditions
 * are on (reference models/pano/PanoGenerator.py:153-157 `get_cn`: list(cn.parameters()) at lr x 0.1; autograd through
 * diffusers' GroupNorm at MVGenModel.py:68-83).  Forward y = act(x * scale + shift) as for pf_groupnorm_bwd;
 * unit_scale / unit_shift fp32 [n][c0+c1]: the same statistics with gamma = 1, beta = 0 (xhat = x * unit_scale +
 * unit_shift).  dgamma_dbeta fp32 [2 * (c0+c1)] = (sum dz * xhat | sum dz), dz = dy * act'(x * scale + shift), summed
 * over images and pixels in a fixed order. */
size_t pf_groupnorm_param_grads_workspace_size(int n_img, int hw, int C);
pf_status pf_groupnorm_param_grads(const void* x0, int c0, const void* x1, int c1, int dtype, int n_img, int hw,
                                   const float* scale, const float* shift, const float* unit_scale, const float* unit_shift,
                                   int act, const float* dy, float* dgamma_dbeta, void* workspace, size_t workspace_bytes,
                                   void* stream);

/* dz = dy * silu'(z), z 16-bit or PF_F32 [n] (a pre-activation the forward kept), dy / dz fp32: the SiLUs of the ControlNet's
 * conditioning embedding and of the timestep embedding under autograd (diffusers ControlNetConditioningEmbedding /
 * TimestepEmbedding, reached from MVGenModel.py:68-83). */
pf_status pf_silu_bwd(const void* z, int dtype, const float* dy, long n, float* dz, void* stream);

/* im2col of a 3x3 / pad 1 convolution, stride 1 or 2: x [n][h][w][C] 16-bit -> y [n*ho*wo][9][C] (tap = 3 ky + kx, zero
 * outside the image; ho = (h-1)/stride+1).  The weight gradient of a trainable convolution (torch autograd through
 * nn.Conv2d in the ControlNet) is then one token-reducing pf_conv_gemm: dW [cout][9 C] = dY^T y. */
pf_status pf_im2col3(const void* x, int dtype, int n, int h, int w, int C, int stride, void* y, void* stream);

/* LoRA fold, once per projection and optimizer step of a training run (the rank-4 LoRA of models/pano/PanoGenerator.py:129-151
 * on q / k / v / out of every attention; diffusers LoRALinearLayer: y = W x + scale * up(down(x))):
 *   out [N][out_ld] (16-bit `dtype`) = W + scale * up @ down,   W fp32 [N][K], up fp32 [N][r], down fp32 [r][K], r <= 64
 * (r = 0: a plain fp32 -> 16-bit conversion).  Optional by-products for the backward, NULL to skip: out_t [K][out_t_ld] the
 * transpose of the folded weight; d_out [r][d_ld] = down and u_out [r][u_ld] = up^T in 16 bit (rows / a block of the
 * stacked matrices the LoRA gradient GEMMs read).  `out` may be a row slice of a larger packed weight. */
pf_status pf_lora_fold(const float* w, const float* up, const float* down, int N, int K, int r, float scale, int dtype,
                       void* out, long out_ld, void* out_t, long out_t_ld, void* d_out, long d_ld, void* u_out, long u_ld,
                       void* stream);

/* Data movement of the backward pass (NHWC):
 *   pf_zero_insert2   y [n][2h][2w][C] = x at the even positions, zero elsewhere (16-bit): the data gradient of a
 *                     stride-2 convolution is the stride-1 convolution of this with the flipped kernel;
 *   pf_sum2x2         y [n][h][w][C] = sums of the 2x2 blocks of x [n][2h][2w][C] (fp32): nearest x2 up-sampling backward;
 *   pf_pad_width_bwd  dy [n][h][w+2pad][C] -> dx [n][h][w][C]: the margins fold back onto the columns they copy (fp32);
 *   pf_crop_width_bwd dy [n][h][w-2crop][C] -> dx [n][h][w][C] with zero margins (fp32). */
pf_status pf_zero_insert2(const void* x, int dtype, int n, int h, int w, int C, void* y, void* stream);
pf_status pf_sum2x2(const float* x, int n, int h, int w, int C, float* y, void* stream);
pf_status pf_pad_width_bwd(const float* dy, int n, int h, int w, int C, int pad, float* dx, void* stream);
pf_status pf_crop_width_bwd(const float* dy, int n, int h, int w, int C, int crop, float* dx, void* stream);

/* x [B][T][C] -> y [B][C][T] (16-bit): the token-contiguous operand layout of the products that reduce over tokens
 * (attention backward, weight gradients).  C % 4 == 0. */
pf_status pf_transpose_tokens(const void* x, int dtype, int B, long T, int C, void* y, void* stream);

/* GEGLU backward: u [rows][2*inner] = [a | gate] (forward input of pf_geglu), dg [rows][inner] ->
 * du [rows][2*inner] = [dg * gelu(gate) | dg * a * gelu'(gate)]. */
pf_status pf_geglu_bwd(const void* u, const void* dg, int dtype, long rows, int inner, void* du, void* stream);

/* Column sums: x [rows][N] (16-bit or PF_F32, row stride ld) -> out fp32 [N], fixed summation order.
 * workspace: fp32 scratch of pf_colsum_workspace_size(rows, N) bytes. */
size_t pf_colsum_workspace_size(long rows, int N);
pf_status pf_colsum(const void* x, int dtype, long rows, int N, long ld, float* out, void* workspace,
                    size_t workspace_bytes, void* stream);

/* Weighted column sums, the token-reducing half of the LoRA gradients (rank-4 LoRA of PanoGenerator.py:129-151 under autograd:
 * d_up = dY^T (X down^T), d_down = (dY up)^T X):  out[r][c] = scale * sum_t w[r][t] * x[t][c]  for x [T][C] 16-bit row-major
 * (row stride ld) and w fp32 [R][w_ld >= T], R a multiple of 4 (the stacked ranks of a
 * projection group: 12 for q / k / v at rank 4, 24 at rank 8).  x is read once per 16 rows of w, as the backward holds it (no transposed copy); partial
 * sums per row slab are added in a fixed order.  scale = host_scale * (*dev_scale if given: the gradient-normalisation factor
 * on the device).  blocks (host, n_blocks x 4 ints (row0, rows, col0, cols), n_blocks <= 4): when given, only these blocks of
 * [R][C] are written, one after the other, block b TRANSPOSED as [cols_b][rows_b] (the [N_i][rank] layout of a LoRA up
 * matrix); else out is [R][C].  workspace: pf_weighted_colsum_workspace_size(T, C, R) bytes. */
size_t pf_weighted_colsum_workspace_size(long T, int C, int R);
pf_status pf_weighted_colsum(const void* x, int dtype, long T, int C, long ld, const float* w, int R, long w_ld,
                             const float* dev_scale, float host_scale, const int* blocks, int n_blocks, float* out,
                             void* workspace, size_t workspace_bytes, void* stream);

/* Gradient normalisation for 16-bit backward operands.  state: 4 floats on the device.
 *   pf_amax_f32: state[0] = max(state[0], max |x|) (reset = 1 clears it first);
 *   pf_pow2_scale: state[1] = 2^-e with amax * 2^-e in [1, 2) (1 when amax is 0 or not finite), state[2] = 2^e;
 *   pf_scale_f32: y = x * state[index]; y fp32 (may alias x) or 16-bit. */
pf_status pf_amax_f32(const float* x, long n, float* state, int reset, void* stream);
pf_status pf_pow2_scale(float* state, void* stream);
pf_status pf_scale_f32(const float* x, long n, const float* state, int index, int out_dtype, void* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PANFUSION_HIP_H */
