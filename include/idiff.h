/*
 * idiff.h -- C ABI of the MI355X-native InstanceDiff hot path (gfx950 HIP kernels).
 *
 * The reference (zyc-123/InstanceDiff) is pure Python: its "FFI" for this path is the YAML-keyed
 * plugin registry (models/__init__.py:4-12 -> create_net / create_sde) and every device kernel is an
 * implicit ATen/cuDNN dispatch.  Each entry point below names the reference call site(s) whose
 * implicit kernels it replaces.  The Python host side (instancediff_amd/) binds these with ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers to fp32 unless stated; NCHW, contiguous inside a sample;
 *     "bstride" = elements between consecutive samples (lets a tensor be a channel slice of a bigger one)
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work (no sync, no alloc, no free)
 *   - caller owns every buffer, including workspaces; no pointer is retained after return
 *   - return 0 on success, <0 on error (IDIFF_E_*); idiff_last_error() gives a message
 */
#ifndef IDIFF_H
#define IDIFF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IDIFF_OK 0
#define IDIFF_E_BADARG (-1)
#define IDIFF_E_UNSUPPORTED (-2)
#define IDIFF_E_HIP (-3)

typedef void* idiff_stream_t;

const char* idiff_last_error(void);
int idiff_version(void);
/* kernel launches this library has enqueued since it was loaded (bench.py reports launches per denoising step; under HIP-graph
 * capture the counter advances at capture time, not per replay) */
int64_t idiff_launch_count(void);
/* number of compute units etc. of the current device (plumbing for grid sizing / tests) */
int idiff_device_info(int* num_cu, int* wave_size, char* arch_name, int arch_name_len);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the f32 matrix cores (v_mfma_f32_32x32x2_f32), NCHW fp32.
 * Replaces: every nn.Conv2d inside the (missing) UNet body -- ResBlock convs, init 7x7, final 3x3,
 * res 1x1, Down/Up-sample (call sites models/drift_noise_model.py:250-268).
 * ---------------------------------------------------------------------------------------------- */
#define IDIFF_CONV_NORMAL 0     /* out HxW = in HxW, pad = ks/2                                         */
#define IDIFF_CONV_UPSAMPLE2 1  /* nearest x2 upsample fused into the gather: out = 2Hin x 2Win         */
#define IDIFF_CONV_UNSHUFFLE2 2 /* pixel_unshuffle(2) fused: virtual Cin = 4*C0, out = Hin/2 x Win/2, ks=1 */

typedef struct {
    const float* src0;      /* [B, C0, Hin, Win]                                                      */
    const float* src1;      /* optional second source, channels appended after src0 (virtual concat)  */
    int64_t src0_bstride, src1_bstride;
    int32_t C0, C1;
    int32_t B, Hin, Win;
    int32_t mode;           /* IDIFF_CONV_*                                                            */
    int32_t ks;             /* 1, 3 or 7                                                               */
    int32_t Cout;
    const float* wpk;       /* packed weights [ks*ks][Cin][Cout] (idiff_pack_conv_weight)             */
    const float* bias;      /* [Cout] or NULL                                                          */
    /* prologue on src0 (requires C1 == 0): v = silu(pro_a[b,c]*v + pro_b[b,c]); zero padding applies
       AFTER the activation (it pads the activated tensor). NULL = raw input.                          */
    const float* pro_a;
    const float* pro_b;
    /* epilogue */
    float* out;             /* [B, Cout, Hout, Wout]                                                   */
    int64_t out_bstride;
    const float* res;       /* optional residual, same shape as out (res_bstride)                      */
    int64_t res_bstride;
    const float* vec;       /* optional per-(b,co) additive vector [B, Cout]                           */
    /* optional epilogue term: out += silu(aux_a[b,co]*aux[b,co,y,x] + aux_b[b,co])  (GN+SiLU of another tensor) */
    const float* aux;
    int64_t aux_bstride;
    const float* aux_a;
    const float* aux_b;
    float* stats;           /* optional GroupNorm partials [B][ntiles][Cout][2] (sum, sumsq) of the value
                               acc+bias, ntiles = idiff_conv2d_num_tiles(Hout,Wout)                     */
    const float* wwino;     /* optional Winograd-domain copy of the same 3x3 weights (idiff_pack_conv_weight_wino):
                               when set and the shape tiles exactly (Cin % 8 == 0 per source, Cout % 16 == 0,
                               Hout % 8 == 0, Wout % 32 == 0, NORMAL / UPSAMPLE2) the F(2x2,3x3) kernel runs
                               (2.25x fewer matrix-core flops, same epilogue); otherwise wpk is used       */
    const float* wwino4;    /* optional F(4x4,3x3)-domain copy (idiff_pack_conv_weight_wino4): preferred over wwino when
                               Hout % 4 == 0, Wout % 4 == 0, Wout >= 24, Cin % 8 == 0, Cout % 16 == 0 and a sample has at
                               least 16 items of 16x32 pixels x 64 channels (4x fewer matrix-core flops than direct) */
    /* Optional GroupNorm finalize of `stats` behind this conv (gn_out_a != NULL; requires stats): the per-(sample, channel) affine
       idiff_gn_finalize would compute, by an idiff_gn_finalize launch enqueued behind the conv by this call (one C call per conv +
       finalize).  Arguments as idiff_gn_finalize.  (r03 / r04 also had the finalize as the TAIL of the F(4x4,3x3) launches -- last
       arriving workgroups reducing the partials, bit-identical -- behind a `gn_ticket` field: measured slower in both rounds,
       +0.78 ms per step on the r04 kernel, and removed) */
    const float* gn_gamma;
    const float* gn_beta;
    const float* gn_film;
    int64_t gn_film_ld;
    float gn_eps;
    int32_t gn_groups;
    float* gn_out_a;
    float* gn_out_b;
    float* gn_mean_rstd;    /* optional [B, groups, 2] */
    int32_t algo_request;   /* 0 = the library picks (by the layer's per-sample shape only, never by the batch); 1 + IDIFF_CONV_ALGO_x =
                               run exactly that kernel or fail with IDIFF_E_ARG if the shape does not tile for it (a per-call
                               request: profiling, parity tests of a kernel at small sizes; a request for the F(4x4,3x3) kernel
                               waives its 16-items-per-sample threshold, nothing else); -(1 + IDIFF_CONV_ALGO_x) = prefer that
                               kernel where the shape tiles for it, the library's own choice elsewhere */
    const void* wx3;        /* ks == 1, normal mode: the weights split three ways into bf16 planes (idiff_pack_conv1x1_x3) or NULL.
                               With it, 1x1 layers whose pixels tile by 256 (Cout % 64 == 0, C0 % 8 == 0, Cin % 8 == 0, Cin >= 32, no prologue, no
                               statistics) run on the bf16 matrix cores with six products per fp32 product (IDIFF_CONV_ALGO_X3:
                               fp32-class result, csrc/conv1x1_x3.hip); NULL keeps them on the f32 matrix cores */
} idiff_conv_desc;

int idiff_conv2d_num_tiles(int Hout, int Wout);
int idiff_conv2d_fwd(const idiff_conv_desc* d, idiff_stream_t stream);
/* which kernel the calling thread's last idiff_conv2d_fwd launched (profiling / tests) */
#define IDIFF_CONV_ALGO_DIRECT 0   /* implicit GEMM, conv_igemm.hip  */
#define IDIFF_CONV_ALGO_WINOGRAD 1 /* F(2x2,3x3),   conv_wino.hip   */
#define IDIFF_CONV_ALGO_STREAM1X1 2 /* weight gradient only: streaming 1x1 product */
#define IDIFF_CONV_ALGO_WINOGRAD4 3 /* F(4x4,3x3),  conv_wino4.hip  */
#define IDIFF_CONV_ALGO_WINOGRAD4H 4 /* F(4x4,3x3), half-patch items, two workgroups per CU: conv_wino4h.hip */
#define IDIFF_CONV_ALGO_X3 5 /* 1x1, fp32 operands as three bf16 planes, six bf16 MFMAs per product: conv1x1_x3.hip */
int idiff_conv2d_last_algo(void);
/* Which kernel idiff_conv2d_fwd(d, .) WOULD launch for this descriptor: every check and selection rule of the call, no launch
 * (IDIFF_CONV_ALGO_* >= 0, or IDIFF_E_*).  The weight-image pointers of `d` only have to point at memory of the right size: a
 * caller whose weights change every step (training) asks first and then fills only the image the chosen kernel reads -- one pack
 * launch per use instead of three.  Does not change idiff_conv2d_last_algo(). */
int idiff_conv2d_plan(const idiff_conv_desc* d);
/* w [Cout][Cin] (the ks == 1 weight, torch layout) -> the three-plane bf16 image idiff_conv_desc.wx3 points to:
 * [chunk of 32 ci][block of 64 co][plane][octet of 8 ci][co][8 bf16], zero beyond Cin / Cout; x = plane0 + plane1 + plane2 exactly.
 * `image` holds idiff_conv1x1_x3_image_bytes(Cout, Cin) bytes, 16-byte aligned. */
long long idiff_conv1x1_x3_image_bytes(int Cout, int Cin);
int idiff_pack_conv1x1_x3(const float* w, void* image, int Cout, int Cin, idiff_stream_t stream);
/* w [Cout][Cin][ks][ks] (torch layout) -> wpk [ks*ks][Cin][Cout] */
int idiff_pack_conv_weight(const float* w, float* wpk, int Cout, int Cin, int ks, idiff_stream_t stream);
/* same, but spatially flipped and in/out swapped: wpk_T [ks*ks][Cout][Cin] for the data-gradient conv */
int idiff_pack_conv_weight_T(const float* w, float* wpk, int Cout, int Cin, int ks, idiff_stream_t stream);
/* 3x3 weights -> Winograd F(2x2,3x3) domain U = G g G^T, laid out as the kernel stages it:
 * [Cin/8][ceil(Cout/64)][16 xi][4 co-blocks][4 k][16 co][2]  (16*Cin*ceil(Cout/64)*64 floats; Cin % 8 == 0,
 * Cout % 16 == 0: a partial last block is zero-filled), Winograd rows
 * in the kernel's order (u0, u1, -u3, u2) -- an opaque image, only idiff_conv2d_fwd reads it.
 * transpose != 0: the flipped, in/out-swapped weights of the data-gradient conv (then the conv has Cin' = Cout, Cout' = Cin). */
int idiff_pack_conv_weight_wino(const float* w, float* wwino, int Cout, int Cin, int transpose, idiff_stream_t stream);
/* 3x3 weights -> Winograd F(4x4,3x3) domain (6x6 positions), laid out as conv_wino4.hip stages it:
 * [Cin/4][ceil(Cout/64)][9 position quads][4 co-blocks][4 k][16 co][4]  (36*Cin*ceil(Cout/64)*64 floats; Cin % 8 == 0,
 * Cout % 16 == 0: a partial last block is zero-filled) -- an opaque image, only idiff_conv2d_fwd reads it. */
int idiff_pack_conv_weight_wino4(const float* w, float* wwino4, int Cout, int Cin, int transpose, idiff_stream_t stream);
/* Layers with fewer than 16 items PER SAMPLE (16x32 pixels x 64 output channels each) stay on the F(2x2,3x3) kernel; the batch
 * size never enters the choice (a sample's bits do not depend on its batch).  There is no process-wide switch: a caller that
 * wants another kernel for ONE call says so in idiff_conv_desc.algo_request. */

/* ------------------------------------------------------------------------------------------------
 * GroupNorm (+FiLM) folded to a per-(sample,channel) affine, applied by the consumer kernel.
 * Replaces: nn.GroupNorm + x*(scale+1)+shift inside every ResBlock.
 *   mean/var per (b,group) from the conv partials (fp64 finalize), then
 *   a[b,c] = rstd*gamma[c]*(1+scale[b,c]);  b[b,c] = (beta[c]-mean*rstd*gamma[c])*(1+scale[b,c]) + shift[b,c]
 *   film: [B, 2C] (scale = film[b, c], shift = film[b, C+c], row stride film_ld) or NULL.
 *   mean_rstd (optional out): [B, groups, 2] for the backward pass.
 * ---------------------------------------------------------------------------------------------- */
int idiff_gn_finalize(const float* stats, int ntiles, int B, int C, int groups, int HW, const float* gamma,
                      const float* beta, const float* film, int64_t film_ld, float eps, float* out_a, float* out_b,
                      float* mean_rstd, idiff_stream_t stream);

/* out[b,c,p] = silu(a[b,c]*h[b,c,p] + b[b,c]) + res[b,c,p] + vec[b,c]   (res, vec, a/b optional)
 * Replaces: GroupNorm->SiLU->(+residual) tail of each ResBlock and the M=1 cross-attention add. */
int idiff_affine_silu_add(const float* h, int64_t h_bstride, const float* a, const float* b, const float* res,
                          int64_t res_bstride, const float* vec, float* out, int64_t out_bstride, int B, int C, int HW,
                          idiff_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Token-side ops (small row counts): Linear, LayerNorm, sinusoidal time embedding.
 * Replaces: timestep-embedding MLP, per-ResBlock time projections, image-context K/V projections and
 * the ScoreMapModule text branch (_modified_BiomedCLIP.py:531-541,1205-1223).
 * ---------------------------------------------------------------------------------------------- */
#define IDIFF_ACT_NONE 0
#define IDIFF_ACT_SILU 1
#define IDIFF_ACT_GELU 2
/* out[r, n] = res[r,n] + gscale[n] * ( sum_k act_in(x[r,k]) * w[n,k] + bias[n] ), then act_out.
 * x: [R, K] row stride ldx; w: [N, K] row stride ldw (torch nn.Linear layout); out row stride ldo;
 * res (row stride ldr), bias, gscale optional (NULL). */
int idiff_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* res,
                     int64_t ldr, const float* gscale, float* out, int64_t ldo, int R, int K, int N, int act_in,
                     int act_out, idiff_stream_t stream);
/* same contract with a PRE-TRANSPOSED weight wT [K, N] (row stride ldw): lane = output feature, unit-stride weight
 * reads, no cross-lane reduction -- the low-latency form used for the ScoreMapModule token chain. */
int idiff_linear_t_fwd(const float* x, int64_t ldx, const float* wT, int64_t ldw, const float* bias, const float* res,
                       int64_t ldr, const float* gscale, float* out, int64_t ldo, int R, int K, int N, int act_in,
                       int act_out, idiff_stream_t stream);
/* Up to IDIFF_LINEAR_MAX_GROUPS independent idiff_linear_t_fwd / idiff_linear_t_ln_fwd problems (own operands and shapes; ln_g == NULL:
 * no LayerNorm) in ONE launch: the same token-side linear of the ScoreMapModules of all four UNet levels, or of the heads of one
 * folded projection.  Latency-bound launches of ~10 us each: one grouped launch costs what one of them did.  The descriptors are
 * copied into the kernel arguments (nothing is retained).  Matrix-core form only: N % 4 == 0, ldw % 4 == 0, wT 16-byte aligned. */
#define IDIFF_LINEAR_MAX_GROUPS 16
typedef struct {
    const float* x;       /* [R, K], row stride ldx */
    int64_t ldx;
    const float* wT;      /* [K, N] pre-transposed weight, row stride ldw */
    int64_t ldw;
    const float* bias;    /* [N] or NULL */
    const float* res;     /* [R, N] residual (row stride ldr) or NULL */
    int64_t ldr;
    const float* gscale;  /* [N] per-column gain or NULL */
    float* out;           /* [R, N], row stride ldo */
    int64_t ldo;
    const float* ln_g;    /* LayerNorm over K in front of the product (with ln_b; act_in must be IDIFF_ACT_NONE) or NULL */
    const float* ln_b;
    float ln_eps;
    int32_t R, K, N, act_in, act_out;
} idiff_linear_group;
int idiff_linear_t_grouped_fwd(const idiff_linear_group* groups, int ngroups, idiff_stream_t stream);
/* idiff_linear_t_fwd on LayerNorm(x) (per row over K, parameters ln_g / ln_b [K]): the LayerNorm runs on the rows the
 * kernel has staged anyway -- the LayerNorm -> Linear pairs of the ScoreMapModule decoder in one launch each. */
int idiff_linear_t_ln_fwd(const float* x, int64_t ldx, const float* ln_g, const float* ln_b, float ln_eps, const float* wT,
                          int64_t ldw, const float* bias, const float* res, int64_t ldr, const float* gscale, float* out,
                          int64_t ldo, int R, int K, int N, int act_out, idiff_stream_t stream);
/* `heads` independent products of that form in ONE launch (no res / gscale / activation): head h uses x + h*x_hs,
 * wT + h*w_hs, bias + h*b_hs, out + h*o_hs -- the per-head k/v folds of the ScoreMapModule cross-attention, whose
 * operands are column / row blocks of shared matrices. */
int idiff_linear_t_heads_fwd(const float* x, int64_t ldx, int64_t x_hs, const float* wT, int64_t ldw, int64_t w_hs,
                             const float* bias, int64_t b_hs, float* out, int64_t ldo, int64_t o_hs, int R, int K, int N,
                             int heads, idiff_stream_t stream);
/* ScoreMapModule memory projection fused in one pass (ContextDecoder.memory_proj, _modified_BiomedCLIP.py:1205-1209):
 * out[b,:,p] = LayerNorm_256( wpk^T . LayerNorm_C(feat[b,:,p]) + bias );  feat [B,C,N] (feat_bstride), wpk [C][256]
 * (idiff_pack_conv_weight of the [256,C,1,1] view), out [B,256,N]. */
int idiff_smm_memproj_fwd(const float* feat, int64_t feat_bstride, const float* ln1_g, const float* ln1_b,
                          const float* wpk, const float* bias, const float* ln2_g, const float* ln2_b, float* out, int B,
                          int C, int N, float eps, idiff_stream_t stream);
/* Compact form of the same memory for narrow feature maps (C + 1 <= Cm < 256).  With xhat = LayerNorm_C(feat[b,:,p]),
 * z = W.xhat + bias, rstd = 1/sqrt(var(z)+eps2):   LayerNorm_256(z) = g2 * ((Wc.xhat + bc) * rstd) + b2
 * (Wc, bc = W and bias centred over the 256 outputs), i.e. the 256-wide memory is an affine image of the (C+1)-vector
 * [xhat*rstd ; rstd].  out [B,Cm,N] = rows [xhat*rstd (C) ; rstd (1) ; zeros]; the host folds g2.[Wc|bc] into the query
 * and value projections of the cross-attention (b2 cancels in the softmax and re-enters as a bias), so
 * idiff_smm_xattn_fwd streams (C+1)/256 of the bytes.  The variance itself is evaluated as the quadratic form
 * var = xhat^T G xhat + 2 h.xhat + e with  gram = Wc^T Wc / 256  [C][C],  hvec = Wc^T bc / 256  [C],  evar = |bc|^2 / 256
 * (host-prepared in fp64): a C -> C product instead of C -> 256.  Same arithmetic as idiff_smm_memproj_fwd up to fp32
 * rounding.  C = 64 or 128 (a thread holds C / 4 channels of its pixel in registers). */
int idiff_smm_memproj_compact_fwd(const float* feat, int64_t feat_bstride, const float* ln1_g, const float* ln1_b,
                                  const float* gram, const float* hvec, float evar, float* out, int B, int C, int N, int Cm,
                                  float eps1, float eps2, idiff_stream_t stream);
/* Training step (r05): the same projection with the quadratic form's constant read from device memory (it is a function of the
 * weights, evaluated on the device every step), C = 64, and its backward.  Given dm [B, Cm, N] (rows 0..C carry gradient):
 *   dfeat [B, C, N] (dfeat_bstride) and dparams [C*C + 3C + 1] = d gram (row-major) | d ln1_g | d ln1_b | d hvec | d evar.
 * ws: idiff_smm_memproj_compact_bwd_ws_floats(B, C, N) floats (per-workgroup partial rows, reduced in a fixed order). */
int idiff_smm_memproj_compact_train_fwd(const float* feat, int64_t feat_bstride, const float* ln1_g, const float* ln1_b, const float* gram,
                                        const float* hvec, const float* evar_dev, float* out, int B, int C, int N, int Cm, float eps1,
                                        float eps2, idiff_stream_t stream);
int64_t idiff_smm_memproj_compact_bwd_ws_floats(int B, int C, int N);
int idiff_smm_memproj_compact_bwd(const float* feat, int64_t feat_bstride, const float* ln1_g, const float* ln1_b, const float* gram,
                                  const float* hvec, const float* evar_dev, const float* dm, float* dfeat, int64_t dfeat_bstride,
                                  float* dparams, float* ws, int B, int C, int N, int Cm, float eps1, float eps2, idiff_stream_t stream);
/* Several idiff_smm_memproj_compact_fwd problems (the ScoreMapModules of a net's levels) in ONE launch; same arithmetic per problem,
 * same bits.  The launch reserves the LDS of its widest problem: group levels of equal C. */
#define IDIFF_MEMPROJ_MAX_GROUPS 4
typedef struct idiff_memproj_group {
    const float* feat;
    int64_t feat_bstride;
    const float *ln1_g, *ln1_b, *gram, *hvec;
    float evar;
    float* out;
    int32_t C, N, Cm;
} idiff_memproj_group;
int idiff_smm_memproj_compact_grouped_fwd(const idiff_memproj_group* groups, int ngroups, int B, float eps1, float eps2,
                                          idiff_stream_t stream);
int idiff_layernorm_rows_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* out,
                             int64_t ldo, int R, int C, float eps, float* mean_rstd, idiff_stream_t stream);
/* sinusoidal embedding, [sin | cos] halves; freqs [dim/2] = host-built table exp(-ln(1e4) * i/(half-1))
 * (NULL = computed on device) */
int idiff_time_embed_fwd(const float* t, const float* freqs, int B, int dim, float* out, idiff_stream_t stream);
/* The UNet's time-embedding MLP (frozen spec, DESIGN.md section 2: temb = Linear(GELU(Linear(sinusoidal_dim(t))))) in ONE launch:
 * out [B, nout] = w2 . GELU(w0 . [sin(t f) ; cos(t f)] + b0) + b2 with w0 [hid, dim], w2 [nout, hid] (torch layout), freqs [dim/2] or
 * NULL; dim = 64, hid = 256, nout <= 256 (the UNet's nf = 64).  The sums run in a different order than the
 * idiff_time_embed_fwd -> idiff_linear_fwd(act_out = GELU) -> idiff_linear_fwd chain it replaces (fp32 rounding differences only). */
int idiff_time_mlp_fwd(const float* t, const float* freqs, const float* w0, const float* b0, const float* w2, const float* b2, float* out,
                       int B, int dim, int hid, int nout, idiff_stream_t stream);

/* LayerNorm over the channel dim of an NCHW map (per pixel). Replaces the pre-norm of attention blocks. */
int idiff_chan_layernorm_fwd(const float* x, int64_t x_bstride, const float* gamma, const float* beta, float* out,
                             int64_t out_bstride, int B, int C, int HW, float eps, float* mean_rstd,
                             idiff_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Attention.  All follow _modified_BiomedCLIP.py:464-478:  softmax(q k^T * scale) v  per head.
 * ---------------------------------------------------------------------------------------------- */
/* self-attention over a feature map: qkv [B, 3C, N] channel-major (q rows 0..C-1, k C..2C-1, v 2C..3C-1),
 * out [B, C, N]; head h owns channels h*dh..(h+1)*dh-1; dh must be 64 or 32; flash-style on f32 MFMA.
 * lse (optional) [B, heads, N] = log-sum-exp per query for the backward pass. */
int idiff_attn_self_fwd(const float* qkv, float* out, float* lse, int B, int C, int N, int heads, float scale,
                        idiff_stream_t stream);
/* REDUCED-PRECISION VARIANT of idiff_attn_self_fwd (off by default; BASELINE config c5 "fp16 MFMA attention"): Q K^T and P V on
 * the bf16 matrix cores, fp32 softmax and accumulation; head dim 64.  Reported by bench.py as a separate, labelled line. */
int idiff_attn_self_bf16_fwd(const float* qkv, float* out, int B, int C, int N, int heads, float scale, idiff_stream_t stream);
/* the same with fp16 operands clamped to +-255 before the cast -- the reference's own half-precision form (flash-attn branch,
 * models/_modified_BiomedCLIP.py:509-513; BASELINE config c5 "fp16 MFMA attention"); labelled variant, off by default */
int idiff_attn_self_f16_fwd(const float* qkv, float* out, int B, int C, int N, int heads, float scale, idiff_stream_t stream);
/* pixels attend to M context tokens: q [B,C,N] channel-major, k,v [B,M,C] token-major, out [B,C,N]; M <= 32 */
int idiff_attn_ctx_fwd(const float* q, const float* k, const float* v, float* out, int B, int C, int N, int M,
                       int heads, float scale, idiff_stream_t stream);
/* tiny token-major attention (ScoreMapModule decoder self-attention): q [B,Nq,C] (row stride ldq), k,v [B,M,C]
 * (row stride ldkv) -- strides let q/k/v be slices of one packed qkv projection; out [B,Nq,C] dense; Nq,M <= 64 */
int idiff_attn_tokens_fwd(const float* q, const float* k, const float* v, float* out, int B, int Nq, int M, int C,
                          int heads, float scale, int64_t ldq, int64_t ldkv, idiff_stream_t stream);
/* Backward of idiff_attn_tokens_fwd for few tokens (Nq, M <= 8; head dim <= 64) -- the class-token self-attention of the ScoreMapModule
 * decoder in the training step: P is recomputed, dq/dk/dv [rows, C] with row strides lddq / lddkv (packed q|k|v buffers allowed). */
int idiff_attn_tokens_bwd(const float* q, const float* k, const float* v, const float* d_o, float* dq, float* dk, float* dv, int B, int Nq,
                          int M, int C, int heads, float scale, int64_t ldq, int64_t ldkv, int64_t ldo, int64_t lddq, int64_t lddkv,
                          idiff_stream_t stream);
/* idiff_attn_tokens_fwd for ngroups (<= IDIFF_LINEAR_MAX_GROUPS) operand sets of one shape in ONE launch (host arrays of device pointers) */
int idiff_attn_tokens_grouped_fwd(const float* const* q, const float* const* k, const float* const* v, float* const* out, int ngroups,
                                  int B, int Nq, int M, int C, int heads, float scale, int64_t ldq, int64_t ldkv, idiff_stream_t stream);
/* The reference's half-precision attention form for the ScoreMapModule decoder (Attention_flash, models/_modified_BiomedCLIP.py:481-517,
 * selected by TransformerDecoderLayer_scaled(if_flash=True), :552-590): q / k / v clamped to +-255 and rounded to fp16, fp32 scores and
 * softmax statistics, probabilities rounded to fp16 for the second product, fp32 accumulation, result rounded to fp16 (returned as
 * fp32).  Labelled REDUCED-PRECISION VARIANT (model option score_map_if_flash, inference only); csrc/attention_flash.hip.
 * idiff_attn_tokens_f16_fwd: the operands of idiff_attn_tokens_fwd.
 * idiff_smm_xattn_kv_f16_fwd: q [B, Nq, heads*64] (after q_proj), k, v [B, heads*64, N] channel-major (after k_proj / v_proj: unfolded --
 * the folded form has no k / v tensor to clamp), out [B, Nq, heads*64]; heads == 4, Nq <= 8, N % 4 == 0;
 * ws: idiff_smm_xattn_kv_f16_ws_floats(B, N) floats. */
int idiff_attn_tokens_f16_fwd(const float* q, const float* k, const float* v, float* out, int B, int Nq, int M, int C, int heads, float scale,
                              int64_t ldq, int64_t ldkv, idiff_stream_t stream);
int64_t idiff_smm_xattn_kv_f16_ws_floats(int B, int N);
int idiff_smm_xattn_kv_f16_fwd(const float* q, const float* k, const float* v, float* out, float* ws, int B, int Nq, int heads, int C, int N,
                               float scale, idiff_stream_t stream);
/* ScoreMapModule cross-attention, K/V projections folded onto the query side:
 *   S[b,h,q,n] = scale * sum_c qf[b,q,h,c] * mem[b,c,n];  P = softmax_n(S);  o[b,q,h,c] = sum_n P * mem[b,c,n]
 * qf, o: [B, Nq, heads, Cm];  mem: [B, Cm, N] channel-major;  Nq*heads <= 32, Cm in {72, 136, 256}.
 * ws: workspace of idiff_smm_xattn_ws_floats(B,Nq,heads,Cm,N) floats. */
int64_t idiff_smm_xattn_ws_floats(int B, int Nq, int heads, int Cm, int N);
int idiff_smm_xattn_fwd(const float* qf, const float* mem, float* o, float* ws, int B, int Nq, int heads, int Cm,
                        int N, float scale, idiff_stream_t stream);
/* Up to IDIFF_XATTN_MAX_GROUPS independent idiff_smm_xattn_fwd problems of one (B, Nq, heads, scale) -- the cross-attentions of a
 * net's four ScoreMapModules in one decoder layer -- in ONE attention launch and ONE merge launch.  Each problem keeps the kernel,
 * the key split and the merge order of its single launch: the results are the same bits.  ws per problem as for the single call. */
#define IDIFF_XATTN_MAX_GROUPS 8
typedef struct idiff_xattn_group {
    const float* qf;   /* [B, Nq, heads, Cm] */
    const float* mem;  /* [B, Cm, N] */
    float* o;          /* [B, Nq, heads, Cm] */
    float* ws;         /* idiff_smm_xattn_ws_floats(B, Nq, heads, Cm, N) floats */
    int32_t Cm, N;
} idiff_xattn_group;
int idiff_smm_xattn_grouped_fwd(const idiff_xattn_group* groups, int ngroups, int B, int Nq, int heads, float scale,
                                idiff_stream_t stream);
/* Training path of the same attention over the full 256-row memory (Cm = 256, rows = Nq*heads <= 32, row-major qf / o [B, rows, 256]):
 * the forward also returns lse [B, rows] (log-sum-exp of the scaled scores); the backward makes ONE pass over the keys,
 *   dP = do mem, D = rowsum(do*o), G = scale * P * (dP - D), dqf = G mem^T, dmem = do^T P + qf^T G     (P = exp(scale*qf mem - lse)),
 * replacing the batched-GEMM / softmax chains autograd derived (each of which streamed the [B,256,N] memory or [B,rows,N] scores).
 * ws: idiff_smm_xattn_ws_floats(B, rows, 1, 256, N) floats for either call; dmem [B,256,N] is overwritten, or added to when
 * accumulate != 0 (the decoder layers of a ScoreMapModule attend to one memory: their gradients are summed in place, in call order). */
int idiff_smm_xattn_lse_fwd(const float* qf, const float* mem, float* o, float* lse, float* ws, int B, int rows, int N, float scale,
                            idiff_stream_t stream);
int idiff_smm_xattn_bwd(const float* qf, const float* mem, const float* o, const float* lse, const float* d_o, float* dqf, float* dmem,
                        int accumulate, float* ws, int B, int rows, int N, float scale, idiff_stream_t stream);
/* The same pair over a memory of Cm rows (row-major qf / o / dqf [B, rows, Cm], mem / dmem [B, Cm, N]): Cm = 256, or 72 for the compact
 * (C + 1)-row memory of the 64-channel levels (r05: the training step attends to the compact memory too; its padding rows carry no
 * gradient).  ws: idiff_smm_xattn_ws_floats(B, rows, 1, Cm, N) floats. */
int idiff_smm_xattn_cm_lse_fwd(const float* qf, const float* mem, float* o, float* lse, float* ws, int B, int rows, int Cm, int N,
                               float scale, idiff_stream_t stream);
int idiff_smm_xattn_cm_bwd(const float* qf, const float* mem, const float* o, const float* lse, const float* d_o, float* dqf, float* dmem,
                           int accumulate, float* ws, int B, int rows, int Cm, int N, float scale, idiff_stream_t stream);
/* score map: out[b,k,p] = <feat[b,:,p]/max(|feat[b,:,p]|,eps), tv[b,k,:]/max(|tv[b,k,:]|,eps)>; K <= 8
 * sel (optional) [B, HW] = out[b, idx[b], :]  (idx int32 [B]) */
int idiff_scoremap_fwd(const float* feat, int64_t feat_bstride, const float* tv, float* out, const int32_t* idx,
                       float* sel, int B, int C, int HW, int K, idiff_stream_t stream);
/* The score maps of several levels in ONE launch (same K, idx); 16-byte form only (HW % 4 == 0, aligned operands). */
#define IDIFF_SCOREMAP_MAX_GROUPS 4
typedef struct idiff_scoremap_group {
    const float* feat;
    int64_t feat_bstride;
    const float* tv;  /* [B, K, C] */
    float* out;       /* [B, K, HW] */
    float* sel;       /* [B, HW] or NULL (with idx) */
    int32_t C, HW;
} idiff_scoremap_group;
int idiff_scoremap_grouped_fwd(const idiff_scoremap_group* groups, int ngroups, const int32_t* idx, int B, int K, idiff_stream_t stream);
/* Output layer fused with the class gather (replaces the UNet's final 3x3 conv to out_nc channels followed by the
 * per-sample channel pick): out[b,0,y,x] = bias[idx[b]] + sum_{ci,ky,kx} w[idx[b],ci,ky,kx] * x[b,ci,y+ky-1,x+kx-1]
 * w [K,C,3,3] (torch layout), bias [K] or NULL, idx int32 [B] in [0,K), out [B,1,H,W]. */
int idiff_conv3x3_select_fwd(const float* x, int64_t x_bstride, const float* w, const float* bias, const int32_t* idx, float* out,
                             int B, int C, int K, int H, int W, idiff_stream_t stream);
/* Its backward (training step): dx [B,C,H,W] (dx_bstride) = the transposed conv of dpred [B,1,H,W] with each sample's class kernel,
 * dw [K,C,3,3] / db [K] (db may be NULL) = sums over the samples of each class (classes without a sample get zeros); either of dx / dw
 * may be NULL.  W % 4 == 0.  ws: idiff_conv3x3_select_bwd_ws_floats(B, C, H, W) floats (per-tile partials, reduced in a fixed order). */
int64_t idiff_conv3x3_select_bwd_ws_floats(int B, int C, int H, int W);
int idiff_conv3x3_select_bwd(const float* x, int64_t x_bstride, const float* w, const int32_t* idx, const float* dpred, float* dx,
                             int64_t dx_bstride, float* dw, float* db, float* ws, int B, int C, int K, int H, int W, idiff_stream_t stream);
/* out[b,0,p] = x[b, idx[b], p] */
int idiff_gather_channel(const float* x, const int32_t* idx, float* out, int B, int C, int HW, idiff_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * SDE updates.  Replaces the 4-6 ATen elementwise launches + randn_like per step of
 * utils/sde_utils.py:45-46,178-188,196-199 (IRSDE) and the driftSDE reverse update.
 * Noise: z != NULL -> injected draws (parity mode); z == NULL -> on-device Philox4x32-10 + Box-Muller
 * keyed by (seed, offset) (throughput mode).
 * ---------------------------------------------------------------------------------------------- */
#define IDIFF_SDE_STEP 0  /* reverse_sde_step       (sde_utils.py:45-46)  */
#define IDIFF_SDE_MEAN 1  /* reverse_sde_step_mean  (:41-42)              */
#define IDIFF_SDE_ODE 2   /* reverse_ode_step       (:48-49,181-182)      */
/* x_out = x - (theta*(mu-x) - coef*sigma^2*score)*dt [- sigma*(z*sqrt_dt)],  score = -noise_pred/sigma_bar;
 * same fp32 operation order as the reference (bit-exact given identical z). */
int idiff_irsde_reverse_step(const float* x, const float* mu, const float* noise_pred, const float* z, float* x_out,
                             int64_t n, float theta, float sigma, float sigma_bar, float dt, float sqrt_dt, int mode,
                             uint64_t seed, uint64_t offset, idiff_stream_t stream);
/* The pieces the reference composes its steps and closed forms from, one launch each, every product / sum / quotient rounded
 * once in the reference's order (bit-exact against the CPU reference given identical draws).  Operands a, b, z are [B][per]
 * (NULL where an op does not read them; z == NULL on an op that draws noise -> Philox keyed by (seed, offset));
 * mu NULL -> the scalar mu_scalar (the reference's default `self.mu = 0.`, sde_utils.py:152).
 * Coefficients k[0..5]: by value in `k` when coef_dev == NULL (python-int t), else per-sample device rows coef_dev[B][6]
 * (t given as a [B,1,1,1] tensor, sde_utils.py:330-336).  The host fills them from the schedule tables. */
#define IDIFF_IRSDE_SCORE_FROM_NOISE 0 /* get_score_from_noise :187-188   (-a) / k0                       k0 = sigma_bar_t        */
#define IDIFF_IRSDE_MU_BAR 1           /* mu_bar :169-170                  mu + (a - mu)*k0                k0 = exp(-cumsum_t dt)  */
#define IDIFF_IRSDE_DRIFT 2            /* drift :175-176                   (k0*(mu - a))*k1                k0 = theta_t, k1 = dt   */
#define IDIFF_IRSDE_REV_DRIFT 3        /* sde_/ode_reverse_drift :178-182  (k0*(mu - a) - k1*b)*k2         k1 = sigma_t^2 [*0.5], k2 = dt */
#define IDIFF_IRSDE_DISPERSION 4       /* dispersion :184-185              k0*(z*k1)                       k0 = sigma_t, k1 = sqrt(dt) */
#define IDIFF_IRSDE_STEP_MEAN 5        /* reverse_sde_step_mean / reverse_ode_step :41-42,48-49   a - REV_DRIFT(a, b)              */
#define IDIFF_IRSDE_STEP_SDE 6         /* reverse_sde_step :45-46          (a - REV_DRIFT(a, b)) - k3*(z*k4)   k3 = sigma_t, k4 = sqrt(dt) */
#define IDIFF_IRSDE_FORWARD_STEP 7     /* forward_step :38-39              (a + (k0*(mu - a))*k1) + k3*(z*k4)                      */
#define IDIFF_IRSDE_OPT_STEP 8         /* reverse_optimum_step :206-214    ((k0*(a - mu)) + (k1*(b - mu))) + mu   k0 = term1, k1 = term2 */
#define IDIFF_IRSDE_REAL_NOISE 9       /* get_real_noise :222-223          (a - (mu + (b - mu)*k0)) / k1   k1 = sigma_bar_t        */
#define IDIFF_IRSDE_REAL_SCORE 10      /* get_real_score :225-226          (-(a - (mu + (b - mu)*k0))) / k1   k1 = sigma_bar_t^2   */
#define IDIFF_IRSDE_INIT_FROM_NOISE 11 /* get_init_state_from_noise :228-230   (((a - mu) - k0*b)*k1) + mu   k0 = sigma_bar_t, k1 = exp(+cumsum_t dt) */
#define IDIFF_IRSDE_RANDOM_STATES 12   /* generate_random_states :333-336  (z*k1) + (mu + (a - mu)*k0)     k0 = exp(-cumsum_t dt), k1 = sigma_bar_t */
#define IDIFF_IRSDE_NUM_OPS 13
int idiff_irsde_map(int op, const float* a, const float* b, const float* z, const float* mu, float mu_scalar, float* out, int B,
                    int64_t per_sample, const float* coef_dev, const float* k, uint64_t seed, uint64_t offset,
                    idiff_stream_t stream);
/* x_out = ((x - a*r_hat) - b*e_hat) + c*z ;  xa_out (optional) = x_out - cond  (next step's network input) */
int idiff_drift_reverse_step(const float* x, const float* r_hat, const float* e_hat, const float* z, const float* cond,
                             float* x_out, float* xa_out, int64_t n, float a, float b, float c, uint64_t seed,
                             uint64_t offset, idiff_stream_t stream);
/* Graph-replayable drift step: all per-step scalars live in device memory, so ONE captured HIP graph of a denoising step
 * (two UNet forwards + this update + idiff_step_state_advance) replays for every t with no host input in between.
 *   state int32[3] = {t, Philox call count, step index of this run};  coef float[3][Tp1] = tables of (a_t, b_t, c_t);
 *   a, b, c = coef[.][t];  Philox offset = offset_base + state[1]*nper (offset_base: the counters this stream's earlier draws, of
 *   whatever size, have consumed -- successive draws never share counters);  z_base (optional, parity runs) [steps][n] indexed by state[2].
 * In place:  x <- ((x - a*r_hat) - b*e_hat) + c*z ;  xa <- x - cond.   Same fp32 operation order as idiff_drift_reverse_step. */
int idiff_drift_reverse_step_dev(float* x, const float* r_hat, const float* e_hat, const float* z_base, const float* cond,
                                 float* xa, int64_t n, const float* coef, int Tp1, const int32_t* state, uint64_t seed,
                                 uint64_t nper, uint64_t offset_base, idiff_stream_t stream);
/* t <- t-1 (back to T once t <= t_stop), both counters += 1, tdev[0..B) = (float)t   (the UNets' timestep input) */
int idiff_step_state_advance(int32_t* state, float* tdev, int B, int T, int t_stop, idiff_stream_t stream);
/* Inverted dropout (training mode of the ScoreMapModule decoder blocks: nn.Dropout(0.1) in TransformerDecoderLayer / Attention.proj_drop,
 * models/_modified_BiomedCLIP.py:448-478,520-549): out[i] = x[i] / (1-p) where u_i >= p, else 0, with u_i = (w >> 8) * 2^-24 from word
 * i % 4 of Philox counter offset + i / 4 under `seed`.  The mask is a function of (seed, offset, i) alone: the backward pass is the SAME
 * call on the gradient (nothing is stored), x == out allowed. */
int idiff_dropout(const float* x, float* out, int64_t n, float p, uint64_t seed, uint64_t offset, idiff_stream_t stream);
/* standard normals (Philox4x32-10, Box-Muller), element i uses counter (offset + i/4) lane i%4 */
int idiff_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, idiff_stream_t stream);
/* raw Philox4x32-10 words for tests: out[4*i..4*i+3] = philox(counter = offset+i, key = seed) */
int idiff_philox_raw(uint32_t* out, int64_t ncounters, uint64_t seed, uint64_t offset, idiff_stream_t stream);
/* out = alpha*x + beta*y */
int idiff_axpby(const float* x, const float* y, float* out, int64_t n, float alpha, float beta, idiff_stream_t stream);
/* Gradient fan-in in one pass: out[b, :] = srcs[0][b, :] + srcs[1][b, :] (+ ..), 2..4 sources with their own batch strides (a source may
 * be a channel slice of a bigger tensor), summed in argument order.  Replaces torch.autograd's pairwise accumulation (n - 1 passes
 * plus a copy per non-contiguous operand) where a feature map has several consumers.  per_sample % 4 == 0, 16-byte aligned rows. */
int idiff_sum_n(const float* const* srcs, const int64_t* src_bstrides, int nsrc, float* out, int64_t out_bstride, int B,
                int64_t per_sample, idiff_stream_t stream);
/* Many tensors -> one flat buffer in one launch (replaces the per-parameter gradient accumulate / copy launches of
 * torch.autograd's AccumulateGrad + optimizer packing; models/drift_noise_model.py:292-296 semantics unchanged).  segs_dev: device
 * array of nseg records {const float* src (NULL = zero fill); int64 dst_offset; int64 n; int64 first_block}, first_block = running
 * sum of ceil(n / 4096), ascending; nblocks = its total. */
int idiff_gather_segments(const void* segs_dev, int nseg, int64_t nblocks, float* dst, idiff_stream_t stream);
/* forward marginals with per-sample coefficients (training-state samplers):
 *   out[b,:] = c0[b]*x0[b,:] + c1[b]*cond[b,:] + c2[b]*eps[b,:]      (driftSDE.forward_diffusion, IRSDE.generate_random_states) */
int idiff_mix3_per_sample(const float* x0, const float* cond, const float* eps, const float* c0, const float* c1,
                          const float* c2, float* out, int B, int64_t per_sample, idiff_stream_t stream);

/* ================================================================================================
 * TRAINING PATH (models/drift_noise_model.py:242-312: loss.backward() + Adam).  The reference gets its
 * backward from ATen autograd; here every gradient kernel is explicit.
 * ============================================================================================== */

/* Weight gradient of idiff_conv2d_fwd for the SAME descriptor (sources, mode, ks, prologue are re-gathered the
 * way the forward gathered them; wpk/bias/out/epilogue fields are ignored):
 *   dw[co][ci][ky][kx] (torch layout) (+)= sum_{b,y,x} dy[b,co,y,x] * X[b,ci,y+ky-p,x+kx-p]
 * ws: workspace of idiff_conv2d_wgrad_ws_floats(d) floats (per-split partials, reduced in a fixed order). */
int64_t idiff_conv2d_wgrad_ws_floats(const idiff_conv_desc* d);
int idiff_conv2d_wgrad(const idiff_conv_desc* d, const float* dy, int64_t dy_bstride, float* dw, int accumulate, float* ws,
                       idiff_stream_t stream);
/* IDIFF_CONV_ALGO_* of the calling thread's last idiff_conv2d_wgrad: the Winograd form (conv_wino_wgrad.hip) takes 3x3
 * layers with Cout % 64 == 0, Cin % 16 == 0 (C0 % 64 == 0 with a second source), Hout % 2 == 0, Wout % 16 == 0 */
int idiff_conv2d_wgrad_last_algo(void);
/* data-gradient helpers: the data gradient itself is idiff_conv2d_fwd on dy with idiff_pack_conv_weight_T weights */
int idiff_sumpool2x2(const float* x, float* out, int64_t planes, int h, int w, idiff_stream_t stream);     /* upsample^T   */
int idiff_pixel_shuffle2(const float* x, float* out, int B, int C, int h, int w, idiff_stream_t stream);    /* unshuffle^T  */
/* out_bc[b*C+c] = sum_p x[b,c,p];  out_c[c] (+)= sum_b in_bc[b*C+c]   (bias / per-(b,c) vector gradients) */
int idiff_plane_sum(const float* x, int64_t x_bstride, float* out_bc, int B, int C, int HW, idiff_stream_t stream);
int idiff_batch_sum(const float* in_bc, float* out_c, int B, int C, int accumulate, idiff_stream_t stream);

/* Backward of y = silu(a[b,c]*h + b[b,c]) with (a,b) = idiff_gn_finalize(stats(h), gamma, beta, film): gradient
 * w.r.t. h THROUGH the GroupNorm statistics, dgamma/dbeta [C], dfilm [B,2C] (scale | shift).  a, b, mean_rstd as
 * produced by idiff_gn_finalize.  ws: idiff_gn_silu_bwd_ws_floats(B,C,groups) floats.  Optional, from the same reads (no pass of
 * their own): dh_sum [C] = sum over samples and pixels of dh (the bias gradient of the conv that produced h), dy_sum [B,C] = sum
 * over pixels of dy (the gradient of a per-(sample, channel) vector added behind the activation); NULL = not wanted. */
int64_t idiff_gn_silu_bwd_ws_floats(int B, int C, int groups);
int idiff_gn_silu_bwd(const float* dy, int64_t dy_bstride, const float* h, int64_t h_bstride, const float* a, const float* b,
                      const float* mean_rstd, const float* gamma, const float* beta, const float* film, int64_t film_ld,
                      float* dh, int64_t dh_bstride, float* dgamma, float* dbeta, float* dfilm, int64_t dfilm_ld, float* ws,
                      int B, int C, int groups, int HW, int accumulate, float* dh_sum, float* dy_sum, idiff_stream_t stream);

/* elementwise activations (token side) and their gradients: dx = dy * act'(x) */
int idiff_act_fwd(const float* x, float* y, int64_t n, int act, idiff_stream_t stream);
int idiff_act_bwd(const float* dy, const float* x, float* dx, int64_t n, int act, idiff_stream_t stream);
/* out[n] (+)= sum_r x[r,n]  (Linear bias gradient) */
int idiff_colsum(const float* x, int64_t ldx, float* out, int R, int N, int accumulate, idiff_stream_t stream);
/* out[r,n] = x[r,n]*g[n]  and  out[n] = sum_r x[r,n]*y[r,n]  (dense [R,N]; ScoreMapModule gamma and its gradient) */
int idiff_scale_cols(const float* x, const float* g, float* out, int R, int N, idiff_stream_t stream);
int idiff_colsum_prod(const float* x, const float* y, float* out, int R, int N, idiff_stream_t stream);
/* Grouped forms for STACKED token matrices (r05: the same layer of a net's four ScoreMapModule decoders as one [groups * Rg, .] matrix):
 * R rows in `groups` equal groups, each with its own parameter / output row -- gamma, beta, dgamma, dbeta [groups][C]; out [groups][N]. */
int idiff_layernorm_rows_g_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* out, int64_t ldo, int R, int C,
                               float eps, float* mean_rstd, int groups, idiff_stream_t stream);
int idiff_layernorm_rows_g_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean_rstd,
                               float* dx, int64_t lddx, float* dgamma, float* dbeta, int R, int C, int groups, idiff_stream_t stream);
int idiff_colsum_g(const float* x, int64_t ldx, float* out, int R, int N, int groups, idiff_stream_t stream);
int idiff_layernorm_rows_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                             const float* mean_rstd, float* dx, int64_t lddx, float* dgamma, float* dbeta, int R, int C,
                             int accumulate, idiff_stream_t stream);
/* ws: idiff_chan_layernorm_bwd_ws_floats(B, C, HW) floats (>= 2*B*C): per-(sample, workgroup, channel) partials of dgamma / dbeta,
 * reduced in a fixed order; dx and the partials come out of ONE pass over dy and x (C <= 256) */
int64_t idiff_chan_layernorm_bwd_ws_floats(int B, int C, int HW);
int idiff_chan_layernorm_bwd(const float* dy, int64_t dy_bstride, const float* x, int64_t x_bstride, const float* gamma,
                             const float* mean_rstd, float* dx, int64_t dx_bstride, float* dgamma, float* dbeta, float* ws,
                             int B, int C, int HW, int accumulate, idiff_stream_t stream);
/* y = x / max(|x|_2 over channels, 1e-12) per pixel (F.normalize(dim=1)); nrm [B,HW] */
int idiff_chan_normalize_fwd(const float* x, int64_t x_bstride, float* y, float* nrm, int B, int C, int HW,
                             idiff_stream_t stream);
int idiff_chan_normalize_bwd(const float* dy, const float* y, const float* nrm, float* dx, int64_t dx_bstride, int B, int C,
                             int HW, idiff_stream_t stream);
/* out[b, idx[b], p] = x[b,0,p], zeros elsewhere (gradient of idiff_gather_channel) */
int idiff_scatter_channel(const float* x, const int32_t* idx, float* out, int B, int C, int HW, idiff_stream_t stream);

/* generic batched GEMM on the f32 matrix cores:  C[b] = alpha * op(A[b]) (MxK) . op(B[b]) (KxN) + beta * C[b]
 * (row-major; transX != 0 -> the matrix is stored transposed); sA/sB/sC = batch strides in elements */
int64_t idiff_bgemm_ws_floats(int M, int N, int K, int batch); /* 0 unless the shape is K-split (long K, tiny output) */
int idiff_bgemm(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                int transA, int transB, int64_t sA, int64_t sB, int64_t sC, int batch, float alpha, float beta, float* ws,
                idiff_stream_t stream);
/* the same with a per-column bias per batch entry (colbias [batch][N] with batch stride sbias, or NULL), beta = 0, no K split */
int idiff_bgemm_bias(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int transA,
                     int transB, int64_t sA, int64_t sB, int64_t sC, int batch, float alpha, const float* colbias, int64_t sbias,
                     idiff_stream_t stream);
/* y [R,N] = x [R,K] . w [N,K]^T (+ bias [N] or NULL) through the same kernel: the training path's token-side linear layers
 * (reference: nn.Linear inside TransformerDecoderLayer / ContextDecoder, models/modules/_modified_BiomedCLIP.py); any R */
int idiff_linear_mfma_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* out, int64_t ldo, int R,
                          int K, int N, idiff_stream_t stream);
/* out = softmax(scale * x) over each row;  ds = scale * p * (dp - <p, dp>) */
int idiff_softmax_rows_fwd(const float* x, int64_t ldx, float* out, int64_t ldo, int R, int N, float scale,
                           idiff_stream_t stream);
int idiff_softmax_rows_bwd(const float* p, int64_t ldp, const float* dp, int64_t lddp, float* ds, int64_t ldds, int R, int N,
                           float scale, idiff_stream_t stream);

/* losses (drift_noise_model.py:234-240,270,279): bilinear resize of the label (align_corners=False, no antialias;
 * torchvision Resize is unpinned in the reference) and mean squared error with its gradient
 * grad = grad_scale * 2 (a-b) / n  (grad may be NULL); ws: 256 floats; loss: 1 float on the device */
int idiff_resize_bilinear(const float* x, float* out, int64_t planes, int H, int W, int oh, int ow, idiff_stream_t stream);
int idiff_mse_loss(const float* a, const float* b, float* loss, float* grad, float* ws, int64_t n, float grad_scale,
                   idiff_stream_t stream);
/* validation metrics of the drivers (trainUM.py:314-329, testUM.py:151-164) on x/2+0.5, data_range 1:
 * out_b3[b] = {RMSE, PSNR dB, SSIM (skimage: gaussian 11x11 sigma 1.5, K1 .01, K2 .03, interior crop)};
 * pred/target [B,H,W]; ws: B*128 floats */
int idiff_image_metrics(const float* pred, const float* target, float* out_b3, float* ws, int B, int H, int W,
                        idiff_stream_t stream);
/* torch.optim.Adam semantics (L2-in-gradient weight decay, config.yml:138-143), grad pre-scaled by grad_scale
 * (1/world after the flat RCCL all-reduce); step >= 1 */
int idiff_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, float grad_scale, int step, idiff_stream_t stream);

/* bf16 wire format of the data-parallel gradient exchange (replaces nothing in the reference, whose DDP buckets are fp32,
 * models/drift_noise_model.py:145-146; BASELINE config c3 asks for it): round-to-nearest-even pack of the flat fp32 gradient
 * buffer, and the widening unpack after the all-reduce.  The fp32 buffer stays the master copy Adam reads. */
int idiff_f32_to_bf16(const float* x, uint16_t* out, int64_t n, idiff_stream_t stream);
int idiff_bf16_to_f32(const uint16_t* x, float* out, int64_t n, idiff_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IDIFF_H */
