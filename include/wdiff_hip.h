/*
 * wdiff_hip.h - C ABI of libwdiff_hip.so: the MI355X (gfx950) kernels behind the WordDiffusion UNet
 * denoising hot path.
 *
 * The reference (aniketntnu/WordDiffusion) is pure Python on stock PyTorch and has no FFI of its own
 * (SURVEY.md fact 0.1, section 8b); the boundary a maintainer binds is the Python class surface
 * (worddiffusion_amd.UNetModel / UNetModelPhosc / Diffusion / EMA).  This C ABI is what that Python
 * surface calls (ctypes).  Every entry point:
 *   - takes raw DEVICE pointers, sizes and a hipStream_t (passed as void*); no torch types;
 *   - allocates nothing and takes no ownership; workspaces are passed in;
 *   - returns 0 on success, a negative WD_E* code on a bad argument or a failed launch;
 *   - is asynchronous on the given stream and safe to capture into a hipGraph.
 *
 * Data layout: feature maps are token-major ("NHWC"): row m = b*HW + y*W + x, C contiguous floats.
 * GEMM operands are "split-bf16 planes": two bf16 matrices hi = bf16(x), lo = bf16(x - hi), consumed
 * by v_mfma_f32_32x32x16_bf16 as hi*hi + hi*lo + lo*hi (npass = 3, ~2^-17 relative operand error,
 * fp32 accumulate) or hi*hi only (npass = 1, plain bf16).
 *
 * For each function the reference code it replaces is cited as file:line of /root/reference.
 */
#ifndef WDIFF_HIP_H
#define WDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WD_OK 0
#define WD_EINVAL (-1)  /* bad argument (null pointer, unsupported size) */
#define WD_ELAUNCH (-2) /* hip launch / runtime error */
#define WD_ESTATE (-3)  /* call order error (e.g. graph end without begin) */

#define WD_ACT_NONE 0
#define WD_ACT_SILU 1
#define WD_ACT_GEGLU 2 /* out[:, j] = x_j * gelu_erf(gate_j); weight/bias rows packed per BN-tile as [BN/2 x | BN/2 gates]; tile must be set */

typedef uint16_t wd_bf16;

#define WD_MAX_KSPLIT 64 /* largest wd_gemm_args.ksplit */

/* One A-operand source of the tap-gather GEMM.  K contribution = ntaps * c.
 * Row m of the output (sample b = m / hw_out, position p = m % hw_out) reads, for tap t,
 * source row  b*hw_src + gather[t*hw_out + p]  (all-zero row when that entry is < 0);
 * gather == NULL means the identity (row m, ntaps must be 1). */
typedef struct wd_src {
    const wd_bf16* hi;
    const wd_bf16* lo; /* may be NULL when npass == 1 */
    const int32_t* gather;
    int32_t ld;     /* row pitch of hi/lo in elements, multiple of 8 */
    int32_t c;      /* channels read per tap, multiple of 32 */
    int32_t ntaps;  /* 1 (1x1 / linear) or 9 (3x3) */
    int32_t hw_src; /* rows per sample in the source */
} wd_src;

/* out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] + rowvec[m / hw_out, n] + resid[rr(m), n] )
 * with A = [src0 taps | src1 taps] along k.
 * Replaces nn.Conv2d 3x3 / stride-2 / nearest-x2+3x3 / 1x1 and nn.Linear on the path:
 * unet.py:595,621,632 (ResBlock convs + skip), :540 (Downsample), :488-499 (Upsample), :364,375 (proj_in/out),
 * :175-183 (attention projections), :125,145 (GEGLU feed-forward), :1201-1205 (time MLP), :611 (emb_layers). */
typedef struct wd_gemm_args {
    wd_src src[2];
    int32_t nsrc;
    int32_t npass; /* 3 = split-bf16 (fp32-class), 1 = bf16 */
    const wd_bf16* w_hi; /* [n][ktot], k contiguous */
    const wd_bf16* w_lo;
    int32_t m, n, ktot;
    int32_t hw_out;       /* rows per sample of the output (1 for per-sample vectors) */
    const float* bias;    /* [n] or NULL */
    const float* rowvec;  /* [m / hw_out][rowvec_ld] or NULL: additive timestep/writer (FiLM) term, unet.py:660-669 */
    int32_t rowvec_ld;
    const float* resid;   /* [rows][resid_ld] or NULL: residual / skip input */
    int32_t resid_ld;
    const int64_t* resid_rows; /* NULL: row m; else row resid_rows[m] (label_emb gather, unet.py:1581) */
    int32_t act;
    float* out_f32;       /* [m][out_ld] or NULL */
    int32_t out_ld;
    wd_bf16* out_hi;      /* split-bf16 planes of the result for the next GEMM, or NULL */
    wd_bf16* out_lo;
    int32_t out_pl_ld;
    int32_t tile;         /* 0 = auto; else BM*1000+BN of a compiled tile (128064, 128128, 128160, 64064; 64320 / 64080 with w_layout 3) */
    int32_t w_layout;     /* 0: w = [n][ktot].  1: "slab order" [stage][n][32], stage = (32-channel chunk, tap) of src0
                           * (chunk-major, tap-minor) followed by the 32-channel chunks of src1: selects the kernel that
                           * keeps the source slab of a BM-row panel resident in LDS (3x3 taps re-read LDS, not L2)
                           * 3: fragment-major weights (wd_gemm_pack_w), loaded straight into registers; tile 64320 or 128160; or tile
                           *    64080 - all of K inside the workgroup, never a K cut: src[0] a 3x3 / pad 1 / stride 1 source over
                           *    64-position samples of width slab_rows (16 or 32; hw_out == hw_src == 64), the stride-2 table of a
                           *    Downsample onto such samples (slab_rows = 16 = the OUTPUT width, hw_src == 256) or an identity source, src[1]
                           *    (optional) an identity source, planes only, npass 3, n % 80 == 0, no activation / row gather / a32 / ln;
                           *    gn_* is then served by the launch itself (n % 160 == 0, 40 % gn_cpg == 0), ws is not needed.
                           *    Statistics (stat_part) are kept per row panel of the tile: nchunk = max(1, hw_out / 64) for the
                           *    64-row tile */
    int32_t slab_rows;    /* w_layout 1: max over 128-row panels of (max - min + 1) gathered source row; <= 192
                           * w_layout 2 or 3: w as for 0 (2) / fragment-major (3), src[0] is a 3x3 / pad 1 / stride 1 convolution (9 taps, its usual
                           * gather table) over images of width slab_rows, src[1] (optional) an identity source: selects the
                           * kernel that loads the A tile of a kernel row once for its three taps when WDIFF_CONV3=1; in every case
                           * it lets the kernel compute the source rows of a panel instead of reading the gather table */
    int32_t ksplit;       /* 1: off.  >1 (at most WD_MAX_KSPLIT): the K range is cut into ksplit slices run by separate workgroups
                           * (fills the chip when m*n is small); partial sums go to ws and are combined in fixed order.  0: automatic
                           * (splits only when ws is given and the tile grid would leave most CUs idle) */
    float* ws;            /* split-K workspace (ksplit * m * n floats are used) or NULL */
    int64_t ws_floats;    /* capacity of ws in floats */
    double* stat_part;    /* NULL, or [m / hw_out][nchunk][n / stat_cpg][2] (sum, sum of squares) of the finished output per
                           * (sample, 128-row chunk, channel group): GroupNorm statistics for the consumer (wd_gn_apply),
                           * nchunk = max(1, hw_out / 128).  Needs 128-row tiles, hw_out | 128 or 128 | hw_out, whole
                           * groups per column tile; not with GEGLU */
    int32_t stat_cpg;     /* channels per statistics group */
    int32_t dbg;          /* 0 in production.  0x400: take the two-workgroups-per-CU kernel (wd_gemm4_kernel) wherever it is
                             legal, whatever the grid size (parity tests); other bits are timing experiments */
    int32_t* tickets;     /* NULL: split-K partials are combined by a second launch.  Else ntickets ints, ALL ZERO before the
                           * call and left all zero by it, not shared with a launch that may run concurrently: one arrival
                           * counter per output tile - the workgroup that finishes a tile's last K slice sums the slices from
                           * ws in ascending slice order and runs the epilogue itself (same bits as the second launch, one
                           * launch less).  Used when the shapes allow it (16-byte aligned operands, n a multiple of the tile) */
    int32_t ntickets;
    /* GroupNorm of the RESULT in the combine launch (nn.GroupNorm of the consumer, unet.py:427-431 / :161-162, + SiLU): when
     * gn_gamma != NULL the planes out_hi / out_lo receive SiLU?((result - mean) * rstd * gn_gamma[col] + gn_beta[col]) with the
     * statistics of (sample, group of gn_cpg channels) instead of the result itself; out_f32 and stat_part are written as
     * usual.  Only where one combine tile holds whole (sample, group) blocks: hw_out == 64, m % 64 == 0, n % 40 == 0,
     * 40 % gn_cpg == 0, gn_cpg % stat_cpg == 0, stat_part and ws given, no activation, and the launch must be one that cuts K
     * (wd_gemm_auto_ksplit(...) > 1 with ksplit = 0) - anything else returns an error rather than skip the norm. */
    const float* gn_gamma;
    const float* gn_beta;
    float gn_eps;
    int32_t gn_silu;
    int32_t gn_cpg;
    /* GroupNorm of the INPUT while it is staged (w_layout 3, tile 64320, npass 3 only): a32 != NULL makes src[0] the fp32 map
     * a32 [rows][a32_ld] itself (src[0].hi / lo are ignored; c, ntaps, gather, hw_src as usual) and the rows enter the product as
     * SiLU?((x - mean) * rstd * a32_gamma[ch] + a32_beta[ch]) (nn.GroupNorm of the consumer, unet.py:427-431 / :161-162), with
     * the statistics of (sample, group of a32_cpg channels) summed from a32_part [batch][a32_nchunk][c / a32_pcpg][2]
     * (wd_gn_stats' / stat_part's layout; a32_pcpg divides a32_cpg).  Zero padding stays zero.  Needs hw_out % 64 == 0 and
     * hw_src == hw_out's sample grid (a row tile must lie inside one sample), c <= 1024, a32_ld % 4 == 0. */
    const float* a32;
    int32_t a32_ld;
    const double* a32_part;
    int32_t a32_nchunk;
    int32_t a32_pcpg;
    int32_t a32_cpg;
    const float* a32_gamma;
    const float* a32_beta;
    float a32_eps;
    int32_t a32_silu;
    /* LayerNorm of the RESULT's rows in the epilogue (w_layout 3, tile 64320, n == 320, no K cut): ln_gamma != NULL makes
     * out_hi / out_lo receive LayerNorm(result row) * ln_gamma + ln_beta (nn.LayerNorm of the consuming transformer block,
     * unetPhosc.py:241-246 norm1 / norm2 / norm3) instead of the result's own planes; out_f32 is still the result. */
    const float* ln_gamma;
    const float* ln_beta;
    float ln_eps;
    /* Weight groups (w_layout 3, tile 64320, no K cut): w_ngroups > 1 cuts the hw_out rows of every sample into w_ngroups equal
     * runs (each a multiple of 64 rows); run g multiplies with the fragment-major image at w_hi / w_lo + g * w_group_stride
     * elements.  The nearest x2 upsample + 3x3 convolution (Upsample.forward, unet.py:488-499) runs this way as its four output
     * phases (row parity, column parity): a phase reads a 2 x 2 neighbourhood of the SOURCE map with the kernel taps that fall
     * on the same source pixel summed - 4 taps instead of 9; the rows of a sample come out phase-major (wd_gn_apply2's perm_a
     * reads them back in raster order). */
    int32_t w_ngroups;
    int64_t w_group_stride;
} wd_gemm_args;

int wd_gemm(const wd_gemm_args* args, void* stream);

/* wd_gemm_args.w_layout == 3: the weights of a GEMM with n % 320 == 0, ktot % 64 == 0 in FRAGMENT-MAJOR order - the 16 bytes
 * lane l of a v_mfma_f32_16x16x32_bf16 B fragment holds, W[16 ct + (l & 15)][32 ks + 8 (l >> 4) .. + 8], at byte
 * ((ks * n/16 + ct) * 64 + l) * 16 of the plane - so a wave loads a fragment as one contiguous kilobyte straight into
 * registers and the weights never pass through LDS (64 x 320 tiles, csrc/wd_gemmw.hip; same nn.Conv2d / nn.Linear layers as
 * wd_gemm: unet.py:595,621,632,540,488,364,375,145).  wd_gemm_pack_w converts [n][ktot] planes (lo / out_lo may be NULL). */
int wd_gemm_pack_w(const wd_bf16* hi, const wd_bf16* lo, int n, int ktot, wd_bf16* out_hi, wd_bf16* out_lo, void* stream);

/* The GEGLU feed-forward of a transformer block with its residual in one launch per 64-token panel (csrc/wd_ff.hip):
 *     out = resid + GEGLU(x W1^T + b1) W2^T + b2        FeedForward / GEGLU unet.py:122-149, residual unet.py:343-344
 * x: split-bf16 planes of LayerNorm(norm3) of the tokens, [m][x_ld]; c = 320 channels, inner hidden units (a multiple of 128).
 * w1: wd_gemm_pack_w image of the [2 * inner][c] projection with its rows reordered in blocks of 16 x-rows followed by their 16
 * gate rows (unet.py:128 chunk(2): x = rows [0, inner), gate = rows [inner, 2 inner)); b1 in the same order.  w2: wd_gemm_pack_w
 * image of the [c][inner] output projection.  The hidden activations never leave the chip.  The result goes through the GEMM
 * epilogue: fp32 out_f32 and / or split-bf16 planes out_hi / out_lo, optional GroupNorm statistics (stat_part per 64-row panel:
 * nchunk = max(1, hw_out / 64), as wd_gemm with 64-row tiles).
 * w3 != NULL: the 1x1 proj_out of the SpatialTransformer and its residual ride in the same launch (unet.py:406-412),
 *     out = resid3 + (resid + GEGLU(...) W2^T + b2) W3^T + b3
 * with w3 the wd_gemm_pack_w image of the [c][c] projection; the feed-forward result itself is then not stored. */
typedef struct wd_ff_args {
    const wd_bf16* x_hi;
    const wd_bf16* x_lo;
    int32_t x_ld;
    int32_t m, c, inner;
    const wd_bf16* w1_hi;
    const wd_bf16* w1_lo;
    const float* b1;
    const wd_bf16* w2_hi;
    const wd_bf16* w2_lo;
    const float* b2;
    const float* resid; /* [m][resid_ld] or NULL */
    int32_t resid_ld;
    float* out_f32;
    int32_t out_ld;
    wd_bf16* out_hi;
    wd_bf16* out_lo;
    int32_t out_pl_ld;
    const wd_bf16* w3_hi;
    const wd_bf16* w3_lo;
    const float* b3;
    const float* resid3;
    int32_t resid3_ld;
    double* stat_part;
    int32_t stat_cpg;
    int32_t hw_out;
    int32_t npass;
} wd_ff_args;
int wd_ff_fused(const wd_ff_args* args, void* stream);
int wd_ff_supported(int c, int inner); /* 1 when wd_ff_fused covers the shape */
int wd_ff_args_bytes(void);

/* The number of K slices wd_gemm picks by itself (ksplit = 0, tile = 0) for an m x n x ktot product with a workspace of
 * ws_floats floats and no GEGLU: 1 = no cut.  (What a caller needs to know before it asks for the GroupNorm epilogue.) */
int wd_gemm_auto_ksplit(int m, int n, int ktot, int64_t ws_floats);
/* sizeof(wd_gemm_args) as this library was built: a binding that mirrors the struct (ctypes, cgo, ...) compares its own. */
int wd_gemm_args_bytes(void);
/* 1 when the library was built with -DWDIFF_EXPERIMENTAL: the opt-in GEMM variants that measured slower than the defaults (slab
 * order w_layout 1, the row-shared-taps convolution WDIFF_CONV3, the loader-wave ring WDIFF_GEMM_V8, the ping-pong K-half
 * WDIFF_GEMM_PP) are then compiled in; otherwise w_layout 1 is an error and the other switches are ignored. */
int wd_gemm_experimental(void);

/* GroupNorm statistics, unet.py:427-431 (eps 1e-5) and :161-162 (eps 1e-6).  x: [B*hw][ld] fp32 with c channels in
 * groups of cpg.  Writes per (sample, chunk, group) partial (sum, sumsq) in double: part[((b*nchunk + j)*(c/cpg) + g)*2],
 * nchunk = wd_gn_nchunk(hw).  (wd_gemm can produce the same array in its epilogue: wd_gemm_args.stat_part.) */
int wd_gn_nchunk(int hw);
int wd_gn_stats(const float* x, int ld, int batch, int hw, int c, int cpg, double* part, void* stream);
/* part[batch][nchunk][ngroups][2] -> out[batch][1][ngroups][2] (fixed-order sum over the chunks): lets wd_gn_apply run with
 * nchunk = 1 where a sample has many chunks (every workgroup of wd_gn_apply folds the chunks itself otherwise). */
int wd_gn_fold_chunks(const double* part, int batch, int nchunk, int ngroups, double* out, void* stream);

/* Normalise (+ optional SiLU) channels [0, c) of x with groups of cpg channels and emit split-bf16 planes
 * out[m][c_off + ch].  `part` holds nchunk partials per sample at a granularity of part_cpg channels (part_cpg | cpg: a
 * 320-channel tensor keeps 10-channel partials that also serve the 20-channel groups of a 640-channel concat norm).
 * gamma/beta are the norm's full affine vectors (indexed at c_off + ch).  raw_hi != NULL: also the un-normalised input
 * as planes for a 1x1 skip convolution (unet.py:632,671). */
int wd_gn_apply(const float* x, int ld, int batch, int hw, int c, int cpg, const double* part, int nchunk, int part_cpg,
                const float* gamma, const float* beta, float eps, int silu,
                wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, int c_off,
                wd_bf16* raw_hi, wd_bf16* raw_lo, void* stream);

/* The same over the channel concat of TWO tensors (the decoder ResBlocks: [h | skip], unet.py:1586-1588 + :427-431) in one
 * launch: source a fills columns [c_off_a, c_off_a + ca) of the planes, source b [c_off_b, c_off_b + cb); gamma / beta are
 * indexed by the concat channel. */
int wd_gn_apply2(const float* xa, int lda, int ca, const double* part_a, int nchunk_a, int part_cpg_a, int c_off_a,
                 const float* xb, int ldb, int cb, const double* part_b, int nchunk_b, int part_cpg_b, int c_off_b,
                 int batch, int hw, int cpg, const float* gamma, const float* beta, float eps, int silu,
                 wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, wd_bf16* raw_hi, wd_bf16* raw_lo, const int32_t* perm_a, void* stream);

/* out[b][o][y][x] (NCHW, o < oc <= 4) = Conv3x3(SiLU?(GroupNorm(x)))[o] + bias[o] in one launch, fp32 VALU: the UNet's last layer
 * (GroupNorm32, SiLU, conv 320 -> 4; unet.py:1453-1458) and any other few-output-channel 3x3 (pad 1, stride 1).
 * x: token-major fp32 [batch*h*w][ld]; part / nchunk / part_cpg: GroupNorm statistics as for wd_gn_apply; weight: the
 * parameter itself, fp32 [oc][c][3][3].  wd_gn_conv3x3_few_supported(c, w, oc): c % 64 == 0, w <= 64, oc <= 4, LDS fit. */
int wd_gn_conv3x3_few_supported(int c, int w, int oc);
int wd_gn_conv3x3_few(const float* x, int ld, int batch, int h, int w, int c, int cpg, const double* part, int nchunk, int part_cpg,
                      const float* gamma, const float* beta, float eps, int silu, const float* weight, const float* bias, int oc,
                      float* out, void* stream);

/* nn.LayerNorm(c, eps) over the last dim (unet.py:314-316), one row per token -> planes. */
int wd_layernorm(const float* x, int ld, int rows, int c, const float* gamma, const float* beta, float eps,
                 wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, void* stream);

/* fp32 -> split planes (optionally through SiLU). */
int wd_split(const float* x, int ld, int rows, int c, int silu, wd_bf16* out_hi, wd_bf16* out_lo, int out_ld,
             void* stream);

/* softmax(q k^T * scale) v per (sample, head): CrossAttention.forward unet.py:185-279 / unetPhosc.py:176-198
 * and Word_Attention unet.py:823-836 (heads = 1, scale = 1).  q: [B*nq][ldq], k/v: [B*nk][ldk/ldv], head h
 * occupies columns [h*d, (h+1)*d).  Output row = b*out_rows + out_row0 + i; fp32 and/or planes. */
int wd_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                 int batch, int heads, int nq, int nk, int d, float scale,
                 float* out_f32, wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, int out_rows, int out_row0,
                 void* stream);

/* Attention over a context that stays fixed for many calls (the PHOSC cross-attention, unetPhosc.py:241-246 over 10 + 769
 * tokens: the same K / V for all 999 steps of a sampling call).  wd_attention_pack_kv converts K / V [batch*nk][ld] (head h in
 * columns [h*d, (h+1)*d)) ONCE into the split-bf16 key-block images the MFMA kernel keeps in LDS
 * (wd_attention_packed_elems(...) bf16 elements, 16-byte aligned; 0 = shape not covered: nk <= 16 or d not in 16..96 step 16);
 * wd_attention_packed is wd_attention reading those images.  Same arithmetic as wd_attention on the same K / V. */
int64_t wd_attention_packed_elems(int batch, int heads, int nk, int d);
int wd_attention_pack_kv(const float* k, int ldk, const float* v, int ldv, int batch, int heads, int nk, int d,
                         wd_bf16* img, void* stream);
int wd_attention_packed(const float* q, int ldq, const wd_bf16* img, int batch, int heads, int nq, int nk, int d,
                        float scale, float* out_f32, wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, int out_rows,
                        int out_row0, void* stream);

/* timestep_embedding, unet.py:96-116: planes[b][0:half] = cos(t_b * freqs), [half:2*half] = sin(...).
 * freqs[half] is the fp32 table exp(-ln(1e4) k / half) computed by the caller with the reference's op order. */
int wd_timestep_embedding(const int64_t* t, int batch, const float* freqs, int half,
                          wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, void* stream);

/* Cross-attention over <= 10 context tokens, folded and fused (CrossAttention of unet.py:164-279 as used by
 * BasicTransformerBlock unet.py:337-345: x + to_out(softmax(to_q(LN(x)) K^T * scale) V)).
 * wd_xattn_fold (once per context): Mq[b][h*L+j][n] = scale * sum_c K[b*L+j][h*d+c] * Wq[h*d+c][n],
 *                                   Mo[b][h*L+j][n] = sum_c V[b*L+j][h*d+c] * Wo[n][h*d+c]   (fp32; wq [heads*d][c], wo [c][heads*d]).
 * wd_xattn_fused (per step): out = x + bias + softmax_heads(LN(x; gamma, beta, eps) . Mq[b]^T) . Mo[b]; when n_hi != NULL also
 * the following LayerNorm (gamma2, beta2, eps2) of `out` as split-bf16 planes.  Shapes: wd_xattn_supported(c, heads, L).
 * mq_pl / mot_pl (2 * 64 * c bf16 per sample each, zero-initialised by the caller, filled by wd_xattn_fold) are the same
 * matrices as split-bf16 MFMA operands, hi plane then lo plane, stored in the order the consuming MFMA reads them (opaque to
 * the caller: only wd_xattn_fold writes them); when given, the two products of wd_xattn_fused run on MFMA (fp32 VALU
 * otherwise). */
int wd_xattn_supported(int c, int heads, int L);
int wd_xattn_fold(const float* k, int ldk, const float* v, int ldv, int batch, int heads, int L, int d, float scale,
                  const float* wq, const float* wo, int c, float* mq, float* mo, wd_bf16* mq_pl, wd_bf16* mot_pl, void* stream);
int wd_xattn_fused(const float* x, int ld, int batch, int hw, int c, const float* gamma, const float* beta, float eps,
                   const float* mq, const float* mo, int heads, int L, const float* bias, float* out, int out_ld,
                   const float* gamma2, const float* beta2, float eps2, wd_bf16* n_hi, wd_bf16* n_lo, int n_ld,
                   const wd_bf16* mq_pl, const wd_bf16* mot_pl, void* stream);

/* Sampling-time tabulation of the FiLM path (time_embed + label_emb + SiLU + every emb_layers, unet.py:1550-1581,609-615):
 * wd_emb_combine writes SiLU(time[t] + label[y_b]) for T consecutive timesteps (time points at the first one) and every
 * sample b as operand planes (row t*B + b; y_b is clamped to [0, num_classes) - the host range-checks it, the reference
 * raises an index error), one wd_gemm then gives the FiLM vectors of those T steps, and wd_select_rows copies the rows of
 * the current step out of the resident chunk: row ((*t_dev) % chunk)*B + b (chunk = T of the table: one table for all steps). */
int wd_emb_combine(const float* time, const float* label, const int64_t* y, int num_classes, int T, int B, int ted,
                   wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, void* stream);
int wd_select_rows(const float* table, const int32_t* t_dev, int batch, int64_t row_floats, int chunk, float* out,
                   void* stream);

/* Two chained folded cross-attentions in one launch (attn1 then attn2 of a base-model BasicTransformerBlock, unet.py:337-345;
 * both read LayerNorm parameters of their own, here norm2 twice): out = B(A(x)) with A/B = x + bias + softmax(LN(x).Mq^T).Mo,
 * optionally followed by the next LayerNorm as operand planes.  MFMA form only (planes from wd_xattn_fold). */
int wd_xattn_pair(const float* x, int ld, int batch, int hw, int c, float eps, int heads, int L, const float* gamma_a,
                  const float* beta_a, const wd_bf16* mq_pl_a, const wd_bf16* mot_pl_a, const float* bias_a, const float* gamma_b,
                  const float* beta_b, const wd_bf16* mq_pl_b, const wd_bf16* mot_pl_b, const float* bias_b, float* out, int out_ld,
                  const float* gamma2, const float* beta2, float eps2, wd_bf16* n_hi, wd_bf16* n_lo, int n_ld, void* stream);

/* nn.Embedding lookup + positional encoding (CharacterEncoder, unet.py:860-872; PE skipped when pe == NULL,
 * unetPhosc.py:726-729): planes[r][:] = table[ids[r]][:] + pe[r % seq_len][:]. ids are int64 or int32. */
int wd_embed_tokens(const void* ids, int ids_are_i64, int rows, int seq_len, const float* table, int vocab, int c,
                    const float* pe, wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, void* stream);

/* x NCHW [B][cin][h][w] -> im2col planes [B*h*w][kpad], column tap*cin + ci (3x3, pad 1); unet.py:1251. */
int wd_im2col3x3(const float* x, int batch, int cin, int h, int w, wd_bf16* out_hi, wd_bf16* out_lo, int kpad,
                 void* stream);

/* layout changes at the API edge: NCHW [B][c][hw] <-> token-major [B*hw][ld]. */
int wd_nchw_to_tokens(const float* x, int batch, int c, int hw, float* out, int ld, void* stream);
int wd_tokens_to_nchw(const float* x, int ld, int batch, int c, int hw, float* out, void* stream);

/* One reverse-diffusion update, train.py:229-236, on n elements per sample (any layout):
 *   x = ca[t] * (x - cb[t] * eps) + cs[t] * z,   z = 0 when t <= 1,
 * evaluated with the reference's rounding order (no fma contraction), ca = 1/sqrt(alpha), cb = (1-alpha)/sqrt(1-alpha_hat),
 * cs = sqrt(beta) tabulated by the caller.  t = *t_dev (device int32, so that a captured graph can be replayed).
 * noise != NULL: z is read from it (parity tests); else z ~ N(0,1) from Philox4x32-10 keyed by seed with counter
 * (element/4, t, sample_offset + sample index): a sample's noise does not depend on how the batch is sharded. */
int wd_ddpm_step(float* x, const float* eps, int batch, int n_per_sample, const float* ca, const float* cb,
                 const float* cs, const int32_t* t_dev, const float* noise, uint64_t seed, uint64_t sample_offset,
                 void* stream);

/* *t_dev += delta; t64[b] = *t_dev for b < batch (the int64 timesteps vector the UNet takes, train.py:222). */
int wd_advance_timestep(int32_t* t_dev, int delta, int64_t* t64, int batch, void* stream);

/* N(0,1) fill with the same Philox stream family (x_T, train.py:217; noise_images eps, train.py:193). */
int wd_randn(float* out, int batch, int n_per_sample, uint64_t seed, uint64_t sample_offset, uint32_t stream_id,
             void* stream);

/* x_t = sqrt_ah[t_b] * x + sqrt_1m_ah[t_b] * eps  (Diffusion.noise_images, train.py:190-194); the two tables
 * sqrt(alpha_hat) and sqrt(1 - alpha_hat) are tabulated by the caller (fp32, reference op order). */
int wd_noise_images(const float* x, const float* eps, const int64_t* t, const float* sqrt_ah, const float* sqrt_1m_ah,
                    int batch, int n_per_sample, float* out, void* stream);

/* strided device-to-device copy (rows x width_bytes); used to materialise a channel concat (unet.py:1750) only when
 * GroupNorm groups straddle the concat boundary. */
int wd_copy2d(void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t width_bytes, int64_t rows,
              void* stream);

/* EMA of weights, train.py:151-159: ema = ema * beta + (1 - beta) * p over one flat fp32 buffer. */
int wd_ema_update(float* ema, const float* p, int64_t n, double beta, void* stream);

/* Multi-tensor fused AdamW (+ EMA of the weights) - the optimiser side of the training step, train.py:290-294,405,146-170.
 * table: device array of ntensor records {float* p; const float* g; float* m; float* v; float* ema (or NULL);
 * int64 n; int64 chunk0} (wd_adamw_table_entry_bytes() each), chunk0 = running sum of ceil(n / wd_adamw_chunk());
 * total_chunks = that sum over all tensors.  Same fp32 op order as torch.optim.AdamW's single-tensor update at `step`
 * (1-based).  ema_mode 0: none, 1: ema = p (the first 2000 steps of the reference), 2: ema = ema*beta + (1-beta)*p. */
int wd_adamw_table_entry_bytes(void);
int wd_adamw_chunk(void);
int wd_adamw_multi(const void* table, int ntensor, int64_t total_chunks, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int64_t step, int ema_mode, double ema_beta, void* stream);

/* Table-driven weight repack (one launch refreshes every packed operand after an optimiser step; the reference reads
 * its parameters in place, unet.py forward).  Entry layout (wd_repack_entry_bytes() = 80 bytes, little endian):
 *   u64 src0, u64 src1 (0 = none), u64 dst_hi, u64 dst_lo, i32 N, C, T, mode, i32 npad, ld, g, ntile_c, i64 chunk0,
 *   i32 row_off, col_off (mode 3 only)
 * mode 0: dst[perm_g(n)][t*C + c]   = split(src0[n][c][t])   (forward operand; perm_g = GEGLU x|gate interleave, g=0 none)
 * mode 1: dst[c][t*npad + n]        = split(src0[n][c][t]), 0 for N <= n < npad   (data-gradient operand)
 * mode 2: dstf[perm_g(i)]           = src0[i] (+ src1[i]), i < N   (fp32 vectors)
 * mode 3: mode 0's matrix written fragment-major (the image wd_gemm_pack_w makes of it); dst = the image's base, ld = the matrix's
 *         row count (% 16 == 0), the piece sits at (row_off, col_off); needs C % 32 == 0, T in {1, 9}, even col_off
 * dst pointers are pre-offset to the piece's (row, column) origin; ld = row pitch in elements.  A mode-0/1 piece owns
 * ceil(N or npad / tile) * ceil(C / tile) chunks (tile = wd_repack_tile()), a mode-2 piece ceil(N / wd_repack_vchunk()). */
int wd_repack_entry_bytes(void);
int wd_repack_tile(void);
int wd_repack_vchunk(void);
int wd_repack_multi(const void* table, int nentries, int64_t total_chunks, void* stream);

/* loss = mean((pred - target)^2) (nn.MSELoss, train.py:289; deterministic two-stage sum) and, when grad != NULL,
 * grad = d loss / d pred = 2 (pred - target) / n.  scratch: >= min(1024, ceil(n/256)) doubles. */
int wd_mse_loss(const float* pred, const float* target, int64_t n, float* grad, float* loss, double* scratch,
                int scratch_len, void* stream);

/* ---- backward building blocks (training step, train.py:290: loss.backward()).  Contractions reuse wd_gemm:
 *  d(input)  = wd_gemm over d(output) planes with the mirrored gather table and transposed weights;
 *  d(weight) = wd_gemm over the token dimension, operands = the transposed planes made by wd_transpose_planes. */

/* (weight gradient of nn.Conv2d / nn.Linear, unet.py:595,621,632,540,488,364,375,175-183,125,145,1201-1205,611 under autograd)
 * out[(t*c + ch)][mm] (tap_minor = 0) or out[(ch*ntaps + t)][mm] (tap_minor = 1: the OIHW order of a conv weight)
 *   = in[src(mm, t)][ch] for mm < m (0 beyond, up to mpad): split-bf16 planes [ntaps*c][mpad].
 * in: planes (in_is_f32 = 0; in_lo may be NULL) or one fp32 matrix (in_is_f32 = 1, split on the fly); src = row mm, or
 * through the 3x3 gather table as in wd_gemm (zero row for -1). */
int wd_transpose_planes(const void* in_hi, const void* in_lo, int in_is_f32, int ld, int c, const int32_t* gather, int ntaps,
                        int hw_out, int hw_src, int m, int mpad, int tap_minor, wd_bf16* out_hi, wd_bf16* out_lo,
                        void* stream);

/* Weight gradient straight from the row-major operand planes - no transposed copies (csrc/wd_dw.hip):
 *   grad[n][ci * ntaps + t] (+)= sum over tokens mm < m of  d[mm][n] * x[src(mm, t)][ci]
 * (the OIHW gradient of nn.Conv2d / nn.Linear under autograd, same layers as wd_transpose_planes above).  d: split-bf16 planes
 * of d(output) [m][d_ld] (wd_dout_prep's row-major planes); x: the layer's input planes [.][x_ld], pointers at the first of the
 * c channels; src = row mm (gather NULL, ntaps 1) or sample * hw_src + gather[t * hw_out + position] (zero row for -1) as in
 * wd_gemm.  The token range is cut into slices run by separate workgroups; their partial tiles go to ws
 * ([nslice][n][ntaps][c] floats) and are combined in fixed order.  Shapes: wd_dw_supported(). */
typedef struct wd_dw_item {   /* one problem of a grouped launch (wd_dw_group): the fields of wd_dw_args that differ between layers */
    const wd_bf16* d_hi;
    const wd_bf16* d_lo;
    const wd_bf16* x_hi;
    const wd_bf16* x_lo;
    float* grad;
    int32_t d_ld, x_ld, grad_ld, accumulate;
} wd_dw_item;
typedef struct wd_dw_args {
    const wd_bf16* d_hi;
    const wd_bf16* d_lo;      /* NULL with npass 1 */
    const wd_bf16* x_hi;
    const wd_bf16* x_lo;      /* NULL with npass 1 */
    const int32_t* gather;    /* [ntaps][hw_out] or NULL */
    float* grad;              /* [n][grad_ld], grad_ld >= c * ntaps */
    float* ws;
    int64_t ws_floats;
    int32_t d_ld, x_ld, grad_ld;
    int32_t ntaps, hw_out, hw_src;
    int32_t m, n, c;
    int32_t npass;            /* 3: hi.hi + hi.lo + lo.hi, 1: hi.hi */
    int32_t accumulate;       /* 1: grad += */
    int32_t nslice;           /* 0: automatic (wd_dw_slices), else <= m / 64 */
    int32_t dbg;
    int32_t reserved;
    void* stamps;             /* NULL (debug builds: 16 u64 of cycle sums, see csrc/wd_dw.hip) */
    const wd_dw_item* items;  /* set by wd_dw_group (device memory); callers leave it NULL */
    int32_t nitems;
    int32_t reserved2;
} wd_dw_args;
int wd_dw(const wd_dw_args* a, void* stream);
/* Several layers of the SAME shape (m, n, c, ntaps, hw_out, hw_src, gather, npass as in *a) in one launch: the 1x1 layers of a
 * transformer block have four output tiles each - alone they need 64 token slices of 256 tokens to fill the chip and spend their time
 * in prologue, epilogue and 26 MB of partial tiles; eight of them together run 2048-token slices.  items_host / items_dev: the same
 * nitems (<= 64) records in host memory (checked here) and in device memory (read by the kernel; must stay valid until the launch has
 * run - for a captured graph, as long as the graph).  ws: [nslice][nitems][n][ntaps][c] floats. */
int wd_dw_group(const wd_dw_args* a, const wd_dw_item* items_host, const wd_dw_item* items_dev, int nitems, void* stream);
int wd_dw_group_slices(int m, int n, int c, int ntaps, int nitems);
int wd_dw_item_bytes(void);
int wd_dw_supported(int m, int n, int c, int ntaps, int hw_out); /* m % 64, n % 160, c % 160, hw_out % 64 == 0, hw_out <= 1024 */
int wd_dw_slices(int m, int n, int c, int ntaps);               /* the automatic nslice */
int wd_dw_args_bytes(void);

/* (bias gradients of the layers above; gradient of the FiLM vector emb_out[..., None, None], unet.py:660-661)
 * out[s][col] (+)= scale * sum over rows [s*seg, (s+1)*seg) of x[row][col]; fixed summation order.
 * scratch: ceil(rows/seg) * ceil(seg/64) * c floats.  (bias gradients: seg = rows; FiLM gradient: seg = hw.) */
int wd_colsum(const float* x, int ld, int rows, int c, int seg, float* out, int out_ld, int accumulate, float scale,
              float* scratch, int64_t scratch_floats, void* stream);

/* GroupNorm (+SiLU) backward (unet.py:427-431,594,618 through autograd).  x: the forward input, dz: gradient w.r.t. the
 * normalised(+SiLU) output, columns [dz_off, dz_off + c) of a [B*hw][dz_ld] matrix (channel concat: one call per source);
 * part / nchunk_f / part_cpg: the forward statistics as for wd_gn_apply.
 * pass 1 -> sums[b][chunk][2][c] (chunk < wd_gn_bwd_nchunk(hw)): per-channel sums of dy and dy*xhat; their column sums over
 * (b, chunk) are [d beta | d gamma].  pass 2 -> dx (+= when accumulate). */
int wd_gn_bwd_nchunk(int hw);
int wd_gn_bwd_stats(const float* x, int ld, const float* dz, int dz_ld, int dz_off, int batch, int hw, int c, int cpg,
                    const double* part, int nchunk_f, int part_cpg, const float* gamma, const float* beta, int c_off, float eps,
                    int silu, float* sums, void* stream);
int wd_gn_bwd_apply(const float* x, int ld, const float* dz, int dz_ld, int dz_off, int batch, int hw, int c, int cpg,
                    const double* part, int nchunk_f, int part_cpg, const float* gamma, const float* beta, int c_off, float eps,
                    int silu, const float* sums, float* dx, int dx_ld, int accumulate, void* stream);

/* Both passes in one launch: a workgroup keeps its (sample, 40 channels = whole groups) tile of dy and xhat in LDS, so x and dz are
 * read once.  sums: [b][2][c] (the layout of wd_gn_bwd_stats with one chunk).  Shapes: wd_gn_bwd_fused_supported (40 | c, cpg | 40,
 * the hw x 40 tile pair within LDS: hw <= 448); everything else as wd_gn_bwd_stats / _apply. */
int wd_gn_bwd_fused_supported(int hw, int c, int cpg);
int wd_gn_bwd_fused(const float* x, int ld, const float* dz, int dz_ld, int dz_off, int batch, int hw, int c, int cpg,
                    const double* part, int nchunk_f, int part_cpg, const float* gamma, const float* beta, int c_off, float eps,
                    int silu, float* sums, float* dx, int dx_ld, int accumulate, void* stream);

/* LayerNorm backward (nn.LayerNorm of BasicTransformerBlock, unet.py:314-316): dx (+=), and colpart[blk][2][c] (blk < wd_layernorm_bwd_nblk(rows)) whose column sums are
 * [d gamma | d beta]. */
int wd_layernorm_bwd_nblk(int rows);
int wd_layernorm_bwd(const float* x, int ld, const float* dy, int dy_ld, int rows, int c, const float* gamma, float eps,
                     float* dx, int dx_ld, int accumulate, float* colpart, void* stream);

/* softmax-attention backward (CrossAttention.forward unet.py:185-279, Word_Attention :823-836) for nk <= 16 keys: dq[B*nq][lddq]; dkv_part[b][nwg][2][nk][heads*d] = per-workgroup partial
 * sums of (dK, dV) over their tokens (nwg returned; the caller column-sums them). */
/* number of per-workgroup dK/dV partial slabs wd_attention_bwd_small writes per batch element (0 = unsupported shape) */
int wd_attention_bwd_small_nwg(int heads, int nq, int nk, int d);
int wd_attention_bwd_small(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* dout,
                           int ldo, int batch, int heads, int nq, int nk, int d, float scale, float* dq, int lddq,
                           float* dkv_part, int* nwg_out, void* stream);

/* One pass over d(output) [m][n] fp32 (row pitch ld) producing what the layer's backward consumes; each output optional:
 *  pl_*   row-major split planes [m][npad] (columns n..npad-1 zero) - operand of the data-gradient GEMM;
 *  t_*    transposed split planes [n][mpad] (columns m..mpad-1 zero) - operand of the weight-gradient GEMM;
 *  colpart[ceil(mpad/64)][n] column sums of every 64-row block - wd_colsum_finish(colpart, nblk, n, nseg, ...) then sums
 *  nblk consecutive blocks per segment (bias gradient: one segment; FiLM gradient: one segment per sample). */
int wd_dout_prep_rows(void);
int wd_dout_prep(const float* d, int ld, int m, int n, int npad, int mpad, wd_bf16* pl_hi, wd_bf16* pl_lo, wd_bf16* t_hi,
                 wd_bf16* t_lo, float* colpart, void* stream);
/* The same outputs for d(output) of the GEGLU projection (unet.py:125-136 under autograd) WITHOUT materialising it: computed on the
 * fly from the saved pre-activation u = [x | gate] ([m][u_ld]) and the gradient dh [m][dh_ld] of x . gelu(gate); n = npad = 2 inner,
 * inner % 64 == 0.  Replaces wd_geglu_bwd + wd_dout_prep. */
int wd_dout_prep_geglu(const float* u, int u_ld, const float* dh, int dh_ld, int m, int inner, int mpad, wd_bf16* pl_hi, wd_bf16* pl_lo,
                       wd_bf16* t_hi, wd_bf16* t_lo, float* colpart, void* stream);
int wd_colsum_finish(const float* part, int nblk, int c, int nseg, float* out, int out_ld, int accumulate, float scale,
                     void* stream);
/* The same finish for `n` (partials -> gradient) pairs in one launch (nseg = 1 each).  `table` is a device array of
 * wd_colsum_entry_bytes()-sized records {const float* part; float* out; int32 nblk, c, ld, accumulate; float scale; int32 pad}:
 * out[0..c) (+)= scale * sum over k < nblk of part[k * ld + col]; max_c = the widest entry.  Entries of one launch must write
 * distinct `out` ranges.  Used by the training backward for every bias / norm-affine gradient at once (reference: those terms
 * of loss.backward(), train.py:291). */
int wd_colsum_entry_bytes(void);
int wd_colsum_finish_multi(const void* table, int n, int max_c, void* stream);

/* Attention backward (CrossAttention.forward unetPhosc.py:157-198, Word_Attention :696-708) for any number of keys <= 1024
 * (spatial self-attention, the 779-token PHOSC context): recomputes the
 * softmax rows, dq/dk/dv written in place of autograd's (row pitches ldd*; head h owns columns [h*d, (h+1)*d)).
 * scratch: wd_attention_bwd_scratch_floats() floats (P and dS, [batch][heads][nq][nk] each).  Deterministic. */
int64_t wd_attention_bwd_scratch_floats(int batch, int heads, int nq, int nk);
int wd_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* dout, int ldo,
                     int batch, int heads, int nq, int nk, int d, float scale, float* dq, int lddq, float* dk, int lddk,
                     float* dv, int lddv, float* scratch, int64_t scratch_floats, void* stream);

/* dst[i] += src[i] (gradient accumulation where a feature map has several consumers: residual adds, the skip stack
 * of unet.py:1750). 16-byte aligned pointers. */
int wd_add(float* dst, const float* src, int64_t n, void* stream);

/* packed weight gradient [n][tap*c + ch] (row pitch ld >= ntaps*c) -> the parameter's OIHW order [n][ch][tap]. */
int wd_permute_dw(const float* packed, int ld, int n, int c, int ntaps, float* out, void* stream);

/* GEGLU (unet.py:122-135) unfused (training keeps the pre-activation u = [a | g]): h = a * gelu_erf(g) -> planes; du from dh. */
int wd_geglu_fwd(const float* u, int ld, int64_t rows, int inner, wd_bf16* out_hi, wd_bf16* out_lo, int out_ld, void* stream);
int wd_geglu_bwd(const float* u, int ld, const float* dh, int dh_ld, int64_t rows, int inner, float* du, int du_ld,
                 void* stream);
/* nn.SiLU backward (time_embed / emb_layers, unet.py:1201-1205,609): dpre = dact * silu'(pre) */
int wd_silu_bwd(const float* pre, const float* dact, int64_t n, float* dpre, void* stream);
/* backward of F.interpolate(scale_factor=2, mode="nearest") (Upsample.forward, unet.py:497): in [B][2h][2w][c] -> out [B][h][w][c], 2x2 block sums */
int wd_pool2x2_sum(const float* in, int batch, int h, int w, int c, float* out, void* stream);
/* nn.Embedding backward (CharacterEncoder.embedding unet.py:845, label_emb :1245): dtable[v][:] (+)= sum of d[r][:] over rows with ids[r] == v (deterministic) */
int wd_embedding_bwd(const void* ids, int ids_are_i64, int rows, const float* d, int ld, int vocab, int c, float* dtable,
                     int accumulate, void* stream);

/* hipGraph capture of a launch sequence (one denoising step) on `stream`. */
int wd_graph_begin(void* stream);
int wd_graph_end(void* stream, void** graph_exec_out);
int wd_graph_launch(void* graph_exec, void* stream);
int wd_graph_destroy(void* graph_exec);

/* per-kernel-class timing with hipEvents recorded on the launch stream (bench.py roofline leg).
 * classes: 0 gemm (the LDS-staged wd_gemm2_kernel<128,160,...>), 1 gn_stats, 2 gn_apply, 3 layernorm, 4 attention, 5 other,
 * 6 gemm with other tile shapes, 7 split-K combine pass, 8 the two-workgroups-per-CU gemm kernel (wd_gemm4_kernel),
 * 9 the weights-to-registers gemm (wd_gemmw_kernel, 64 x 320 / 128 x 160 tiles), 10 the fused feed-forward (wd_ff_kernel),
 * 11 the weight-gradient kernel (wd_dw_kernel), 12 the whole-K kernel of the small maps (wd_gemmq_kernel).
 * wd_prof_collect: gemm_flops = the 2*M*N*K of class 0; wd_prof_collect_flops additionally returns the algorithmic FLOPs
 * (each multiply-add counted once) of every class that declares them (the contraction classes 0, 6, 8, 9, 10, 11, 12). */
#define WD_NCLASS 13
int wd_prof_enable(int on);
int wd_prof_collect(double* ms_per_class, int64_t* launches_per_class, double* gemm_flops); /* syncs */
int wd_prof_collect_flops(double* ms_per_class, int64_t* launches_per_class, double* flops_per_class); /* syncs */

const char* wd_version(void);
int wd_device_info(int* cu_count, int* lds_per_block, char* name, int name_len);

#ifdef __cplusplus
}
#endif
#endif
