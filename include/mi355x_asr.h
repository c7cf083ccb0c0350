/* mi355x_asr.h -- C ABI of libmi355x_asr.so: the MI355X (gfx950 / CDNA4) kernels behind the drop-in
 * Conformer-CTC training path (NeMo `EncDecCTCModel` + `ConformerEncoder`).
 *
 * The reference has NO C ABI on this path (its host code is Python calling ATen), so this boundary is defined
 * here, directly under the Python NeuralModule classes.  Conventions:
 *   - every entry point returns 0 on success, 1 = invalid argument (the Python binding raises ValueError, the
 *     reference's convention: conformer_encoder.py:569-578, features.py:288-305), 1000 + hipError_t = launch failure
 *     (RuntimeError);
 *   - all pointers are DEVICE pointers owned by the caller (torch allocates; kernels never allocate or free);
 *   - `stream` is a hipStream_t (0 = default stream); calls are asynchronous and re-entrant across streams;
 *   - dtype codes: 0 = float32, 1 = bfloat16 (raw 16-bit); lengths are int64 like the reference's LengthsType tensors;
 *   - activations are channels-last: [B, T, d] rows m = b*T + t; conv feature maps [B, T, F, C].
 *
 * Each declaration cites the reference interface (file:line under the NeMo tree) whose arithmetic it replaces.
 */
#ifndef MI355X_ASR_H
#define MI355X_ASR_H
#ifdef __cplusplus
extern "C" {
#endif

#define MI355X_DT_F32 0
#define MI355X_DT_BF16 1

/* ---- GEMM with fused epilogues ------------------------------------------------------------------------------
 * C[M,N] = epi( sum_k A(m,k) * B(n,k) ).  trans{A,B}=0: operand stored [rows][K] (K contiguous);
 * =1: stored [K][rows] (reduction-major: wgrad / P@V).  Batched: z in [0,batch): z0 = z % nb0, z1 = z / nb0,
 * operand offset = z0*s?0 + z1*s?1 (elements).  bf16 inputs run on MFMA, f32 inputs on an exact fp32 kernel.
 * Replaces: F.linear / Conv1d(k=1) / matmul / conv2d-as-GEMM in conformer_modules.py:382-387,321,343;
 * multi_head_attention.py:124-146,300-350; subsampling.py:231-253,431; conv_asr.py:445.                        */
enum {
  MI355X_EPI_STORE = 0,      /* C = alpha * dropout(acc + bias)                                                  */
  MI355X_EPI_SWISH_DROP = 1, /* aux_out = acc + bias ; C = dropout(swish(acc + bias))          (FFN linear1)     */
  MI355X_EPI_RESID = 2,      /* C(f32) = aux_in(f32) + alpha * dropout(acc + bias)             (residual branch) */
  MI355X_EPI_DSWISH = 3,     /* C = acc * dropmask * swish'(aux_in)                            (FFN dgrad)       */
  MI355X_EPI_RELU_MASK = 4,  /* C = relu(acc + bias) * [ (m % rows_per_b) / rows_inner < row_len[m / rows_per_b] ] */
  MI355X_EPI_MUL_POS = 5,    /* C = acc * (aux_in > 0)                                         (ReLU dgrad)      */
  /* the feed-forward pair with the Swish derivative taken in the FORWARD epilogue (which holds sigmoid(h) and the dropout mask
   * in registers anyway): aux_out = swish'(acc + bias) * dropmask instead of the pre-activation, and the backward epilogue is
   * one multiply (no exp / rcp / mask hash per element).  ConformerFeedForward, parts/submodules/conformer_modules.py:366-387 */
  MI355X_EPI_SWISH_DROP_G = 6, /* aux_out = swish'(acc + bias) * dropmask ; C = dropout(swish(acc + bias))        */
  MI355X_EPI_DSWISH_G = 7      /* C = acc * aux_in            (aux_in = the aux_out of MI355X_EPI_SWISH_DROP_G)   */
};
typedef struct mi355x_gemm_desc {
  const void* A; const void* B; void* C;
  int M, N, K;
  long long lda, ldb, ldc;          /* row pitch (elements) of the stored matrices                               */
  long long c_col_stride;           /* column stride of C (0/1 = dense); lets wgrad write the reference's layouts  */
  int transA, transB;
  int in_dtype;                     /* dtype of A and B                                                          */
  int c_dtype;                      /* dtype of C                                                                */
  int batch, nb0;
  long long sA0, sA1, sB0, sB1, sC0, sC1;
  const void* bias;                 /* f32 [N] or NULL                                                           */
  float alpha;
  int epilogue;
  int atomic;                       /* 1: atomicAdd into f32 C (required for splitk > 1)                         */
  int splitk;
  const void* aux_in; int aux_in_dtype;
  void* aux_out; int aux_out_dtype;
  long long ldaux;                  /* pitch of aux_in / aux_out (same batch offsets as C)                       */
  unsigned drop_key, drop_threshold; float drop_scale;   /* threshold 0 = dropout off                            */
  const void* row_len; int rows_per_b; int rows_inner;   /* EPI_RELU_MASK: int64 [B] valid lengths; optional with EPI_MUL_POS:
                                                          * the caller's promise that aux_in <= 0 on rows beyond them -- tiles that
                                                          * lie entirely there are zero-filled (through the row map) without a K loop */
  long long colsum_stride;          /* batch stride of colsum_out (elements)                                       */
  void* colsum_out;                 /* optional (bf16, transA=1, single-level batch): f32 [M] += sum_k A(k,m) -- the bias gradient
                                       of a Linear rides along with its weight-gradient GEMM                       */
  /* Implicit-GEMM convolution on channels-last feature maps (Conv2d of ConvSubsampling, subsampling.py:385-436, without
   * an im2col buffer).  gather (bf16, transA = 0, batch = 1, C % 64 == 0): row m = (b*nI + i)*nJ + j of A is gathered,
   * for K index tap*C + c, from A[b][i*si + di[tap]][j*sj + dj[tap]][c] of a [.., SI, SJ, C] grid (zero outside);
   * K must equal ntaps*C; lda is ignored.  rowmap: row m = (b*nI + i)*nJ + j of C and of aux_in/aux_out is stored at row
   * (b*OI + i*si + oi)*OJ + j*sj + oj (the dgrad of a strided conv writes one parity class of positions per launch). */
  const struct mi355x_conv_gather* gather;   /* NULL = dense A */
  const struct mi355x_row_map* rowmap;       /* NULL = dense C rows */
} mi355x_gemm_desc;
typedef struct mi355x_conv_gather {
  int nI, nJ, SI, SJ, C, si, sj, ntaps; int di[9], dj[9];
  int operand;   /* 0: A rows gathered as described above.  1: weight gradient -- B (transB = 1) is gathered instead: K runs
                    over the positions m = (b*nI + i)*nJ + j, N = C channels, batch index z (0..ntaps-1) selects the tap;
                    B points at the source grid, ldb / sB are ignored, transA must be 1 */
} mi355x_conv_gather;
typedef struct mi355x_row_map { int nI, nJ, OI, OJ, si, sj, oi, oj; } mi355x_row_map;
int mi355x_gemm(const mi355x_gemm_desc* desc, void* stream);
/* kernel-structure selection knobs for tests and A/B benchmarks (results do not depend on them beyond fp32 summation order).
 * key 4: the 256x256 tile structures (0 = never, 1 = where the tile count fills the chip (default), 2 = whenever N > 128);
 * key 5: the persistent 256x128 structure with overlapped epilogues (0 = never, 1 = where it measured faster (default),
 * 2 = wherever it can run); key 6: operands of the 256x256 structure prefetched through registers two K-tiles deep instead of
 * LDS-DMA (0 / 1 = default, for K-contiguous operands with K % 128 == 0); key 7: the same inside the 256x128 structure;
 * key 3: fp32 problems on the matrix cores, v_mfma_f32_32x32x2_f32 (1 = default, 0 = the vector-unit kernel);
 * key 8: the phase-staggered 256x256 structure on 16x16x32 MFMAs (0 never, 1 = default: K-contiguous layouts that fill the chip
 * with 256x256 tiles, dense or with gathered A rows; 2 = every problem it can run; 3 = as 1 plus the weight-gradient layouts,
 * 4 = as 1 plus the wide plain stores the persistent structure otherwise takes, 5 = as 1 plus its 128x256 tile where only that
 * fills the chip and K >= 768); key 9: start delay of every other first-round workgroup of that
 * structure in 10-ns ticks (experiment, default 0).
 * Returns the previous value (-1 = not yet read from the environment), or -1 for an unknown key. */
int mi355x_gemm_config(int key, int value);
/* Up to 12 independent weight-gradient problems in one launch: every desc must be bf16, transA = transB = 1, atomic
 * f32 C with dense columns, no bias / aux / epilogue, and share K (the token count); descs[0].splitk applies to all.
 * colsum_out (the bias gradient) is honoured per problem.  See csrc/gemm.hip: gemm_bf16_grouped_tn_kernel. */
int mi355x_gemm_grouped(const mi355x_gemm_desc* descs, int n, void* stream);

/* ---- fused macaron feed-forward block, d_model = 512 (ConformerFeedForward.forward, parts/submodules/conformer_modules.py:366-387:
 * Linear -> Swish -> Dropout -> Linear, with the residual `residual + dropout(ff(x)) * fc_factor` of :174-181 / :209-215).
 * One launch: h (bf16 [M, d_ff], pitch ldh) = y @ W1^T + b1 is the only intermediate that reaches memory (backward needs it);
 * out f32 [M, 512] = x_resid + alpha * drop_res( drop_in(swish(h)) @ W2^T + b2 ).  y = LayerNorm output, bf16 [M, 512].
 * Dropout indices are those of the unfused GEMM epilogues: element (m, n) of [M, d_ff] resp. [M, 512] -> m * N + n.
 * The weights are PACKED bf16 images written by mi355x_ffn_pack, MFMA-fragment-major in the order the kernel consumes them.  A
 * fragment = 32 rows x 16 k = 1 KiB stored [hh][lr][8]: element (row lr, k = 8 hh + e) at (hh*32 + lr)*8 + e.
 *   "k512"   order of a logical A [d_ff][512]: fragment of rows c*64 + mt*32 .. +31, k = k16*16 .. +15 at ((c*32 + k16)*2 + mt) KiB
 *   "kchunk" order of a logical B [512][d_ff]: fragment of rows q*128 + mt4*32 .. +31, k = t*16 .. +15 at ((t*4 + q)*4 + mt4) KiB
 * w1_packed = k512(W1), w2_packed = kchunk(W2).  128 <= d_ff <= 2048, d_ff % 64 == 0; all pointers 16-byte aligned; pitches
 * multiples of 8 (ldx: of 4). */
int mi355x_ffn_fwd(const void* y, long long ldy, const void* w1_packed, const void* b1, const void* w2_packed, const void* b2,
                   const void* x_resid, long long ldx, void* h, long long ldh, void* out, long long ldo, int M, int d_model,
                   int d_ff, float alpha, unsigned drop_in_key, unsigned drop_in_threshold, float drop_in_scale,
                   unsigned drop_res_key, unsigned drop_res_threshold, float drop_res_scale, void* stream);
/* Its input-gradient chain in one launch (autograd of the same lines): df bf16 [M, 512] = the gradient of the block's output
 * through the residual scale and dropout; g = df @ W2; dh = g * dropmask_in * swish'(h) -> bf16 [M, d_ff] (pitch ldh; operand of
 * the W1 weight gradient); act = drop_in(swish(h)) -> bf16 [M, d_ff] (pitch ldh; RECOMPUTED operand of the W2 weight gradient);
 * dy bf16 [M, 512] = dh @ W1 (gradient w.r.t. the LayerNorm output).  w2t_packed = k512(W2^T), w1t_packed = kchunk(W1^T). */
int mi355x_ffn_bwd_dgrad(const void* df, long long lddf, const void* w2t_packed, const void* w1t_packed, const void* h,
                         long long ldh, void* dh, void* act, void* dy, long long lddy, int M, int d_model, int d_ff,
                         unsigned drop_in_key, unsigned drop_in_threshold, float drop_in_scale, void* stream);

/* The four images of n weight matrices in one launch (no reference analogue; runs once per optimizer step next to
 * mi355x_pack_weights).  Entry i: src = fp32 master weight, either linear1.weight [d_ff][512] (is_w2 = 0: writes k512(W1) and
 * kchunk(W1^T)) or linear2.weight [512][d_ff] (is_w2 = 1: writes kchunk(W2) and k512(W2^T)); both destinations 512 * d_ff bf16. */
typedef struct mi355x_ffn_pack_entry { const void* src; void* k512; void* kchunk; int d_ff; int is_w2; } mi355x_ffn_pack_entry;
int mi355x_ffn_pack(const void* table_dev, int n_entries, int max_d_ff, void* stream);

/* ---- log-mel front-end: FilterbankFeatures.forward, parts/preprocessing/features.py:423-502 --------------------
 * audio f32 [B,S], audio_len i64 [B] -> out f32 [B,n_mels,T] = log(mel_power + log_guard), T = 1 + S/hop.
 * fb_* = sparse rows of the (persistent) `fb` buffer: for mel m, weights fb_w[fb_off[m] .. +fb_len[m]) apply to FFT
 * bins fb_start[m]...  n_fft must be 512 (Hann window `window[win]` is centred in the FFT frame like torch.stft).
 * Any row layout is accepted; when every fb_off[m] is a multiple of 4 and fb_off[m+1] >= fb_off[m] + 4 * ceil(fb_len[m] / 4)
 * (`sparsify_filterbank` lays the rows out so) the default kernel reads a row as 16-byte vectors without end-of-row checks. */
int mi355x_logmel_fwd(const void* audio, const void* audio_len, const void* window, int win, int hop, int n_fft,
                      const void* fb_start, const void* fb_len, const void* fb_off, const void* fb_w, int n_mels,
                      float preemph, float dither, unsigned seed, float log_guard, void* out, int B, int S, int T,
                      void* stream);
/* which front-end kernel runs (tests and A/B): 2 (default) = FFT in registers (16 lanes per frame, 16-point DFTs on a lane's own
 * points, one LDS transpose, dither evaluated once per sample; profiles/r6_logmel.md; an odd hop takes variant 1), 1 = the
 * wave-synchronised radix-4 kernel (two frames in flight per wave, filterbank in LDS; profiles/r4_logmel.md), 0 = the round-1 kernel.
 * variant < 0 only queries.  Returns the previous setting.  Environment: MI355X_LOGMEL.  (features.py:423-502) */
int mi355x_logmel_config(int variant);
/* normalize_batch(..., 'per_feature') + pad fill, features.py:59-93,490-493.  x f32 [B,n_mels,T] -> y (y_dtype) */
int mi355x_feat_normalize(const void* x, const void* seq_len, void* y, int y_dtype, int B, int n_mels, int T,
                          int normalize, float pad_value, void* stream);

/* ---- convolution sub-sampling pieces: ConvSubsampling.forward, parts/submodules/subsampling.py:385-436,725-759 - */
int mi355x_subsample_conv1_fwd(const void* mel /*f32 [B,F,T]*/, const void* w /*[C,1,3,3]*/, const void* bias,
                               void* out /*[B,T1,F1,C]*/, int out_dtype, const void* len0, const void* len1, int B, int F,
                               int T, int C, void* stream);
int mi355x_subsample_conv1_bwd(const void* dout, int dtype, const void* mel, const void* len0, void* dw, void* db, int B,
                               int F, int T, int C, void* scratch /* optional f32 [ceil(T1/32)*B*10*C]: two-stage reduction */,
                               long long scratch_elems, void* stream);
int mi355x_im2col_3x3s2(const void* in /*[B,T1,F1,C]*/, void* col /*[B*T2*F2, 9C]*/, int dtype, int B, int T1, int F1, int C,
                        void* stream);
int mi355x_col2im_3x3s2_relu(const void* dcol, const void* act, void* din, int dtype, int B, int T1, int F1, int C,
                             void* stream);
/* The same four with the padding as an argument: `pad` zero rows / columns in FRONT of the grid, one behind it -- 1 = Conv2d(
 * padding = 1) (what the entries above run), 2 = CausalConv2D of `causal_downsampling: true` (parts/submodules/causal_convs.py:24-72,
 * subsampling.py:147-149,222-224: F.pad (2, 1) on time AND frequency, then no padding).  Output extents: floor((n + pad - 2) / 2) + 1. */
int mi355x_subsample_conv1_fwd_pad(const void* mel, const void* w, const void* bias, void* out, int out_dtype, const void* len0,
                                   const void* len1, int B, int F, int T, int C, int pad, void* stream);
int mi355x_subsample_conv1_bwd_pad(const void* dout, int dtype, const void* mel, const void* len0, void* dw, void* db, int B,
                                   int F, int T, int C, int pad, void* scratch, long long scratch_elems, void* stream);
int mi355x_im2col_3x3s2_pad(const void* in, void* col, int dtype, int B, int T1, int F1, int C, int pad, void* stream);
int mi355x_col2im_3x3s2_relu_pad(const void* dcol, const void* act, void* din, int dtype, int B, int T1, int F1, int C, int pad,
                                 void* stream);

/* 'dw_striding' sub-sampling (FastConformer x8, Squeezeformer x4; subsampling.py:142-215): depthwise Conv2d(C, C, 3, stride 2,
 * padding 1, groups = C) on a channels-last map in [B,T1,F1,C] -> out [B,T2,F2,C] (+ bias); w f32 [C,1,3,3].  The pointwise
 * convolution + ReLU + time mask that follow are mi355x_gemm (EPI_RELU_MASK).  Backward: din = (in > 0) * dgrad (the ReLU of the
 * previous stage), dw / dbias accumulated (+=); scratch f32 [min(1024, ceil(B*T2*F2/64)) * 10 * C]. */
int mi355x_dwconv2d_s2_fwd(const void* in, const void* w, const void* bias, void* out, int dtype, int B, int T1, int F1, int C,
                           void* stream);
int mi355x_dwconv2d_s2_bwd(const void* dout, const void* in, const void* w, void* din, void* dw, void* dbias, int dtype, int B,
                           int T1, int F1, int C, void* scratch, long long scratch_elems, void* stream);
/* ... with `pad` as above (CausalConv2D stages of the cache-aware streaming FastConformer recipes) */
int mi355x_dwconv2d_s2_fwd_pad(const void* in, const void* w, const void* bias, void* out, int dtype, int B, int T1, int F1, int C,
                               int pad, void* stream);
int mi355x_dwconv2d_s2_bwd_pad(const void* dout, const void* in, const void* w, void* din, void* dw, void* dbias, int dtype, int B,
                               int T1, int F1, int C, int pad, void* scratch, long long scratch_elems, void* stream);

/* ---- transducer head (FastConformer-Transducer): RNNTDecoder.predict / RNNTJoint.joint_after_projection,
 * nemo/collections/asr/modules/rnnt.py:700-830, 1640-1720; LSTM = common/parts/rnn.py:151-230 (torch gate order i,f,g,o).
 * The gate / projection / output-layer contractions are mi355x_gemm; these are the pieces between them. */
/* out (dtype) [U+1,B,H]: row 0 = zero start-of-sequence frame, row u+1 = emb[targets[b,u]] (blank id = padding row = 0) */
int mi355x_embed_sos_fwd(const void* targets /*i64 [B,U]*/, const void* emb /*f32 [V+1,H]*/, void* out, int dtype, int B, int U,
                         int H, int blank, void* stream);
int mi355x_embed_sos_bwd(const void* targets, const void* dx /*[U+1,B,H]*/, int dtype, void* demb /*f32 +=*/, int B, int U, int H,
                         int blank, void* stream);
/* one LSTM step: z f32 [B,4H] holds x W_ih^T + b_ih + h_prev W_hh^T on entry and the ACTIVATED gates on exit; c_prev may be
 * NULL (zero state); h is written in f32 and in the GEMM operand dtype */
int mi355x_lstm_cell_fwd(void* z, const void* b_hh, const void* c_prev, void* c, void* h, void* h_lp, int lp_dtype, int B, int H,
                         void* stream);
/* dh f32 [B,H]; dc f32 [B,H] in = d/dc_t from step t+1, out = d/dc_{t-1}; dz (dtype) [B,4H] = pre-activation gradients */
int mi355x_lstm_cell_bwd(const void* dh, void* dc, const void* act, const void* c, const void* c_prev, void* dz, int dz_dtype,
                         int B, int H, void* stream);
/* h [B,T,U1,J] = dropout(relu(f[b,t,:] + g[b,u,:])); backward: dh <- dh * (h > 0) * drop_scale in place, df [B,T,J] = sum_u */
int mi355x_joint_combine_fwd(const void* f, const void* g, void* h, int dtype, unsigned drop_key, unsigned drop_threshold,
                             float drop_scale, int B, int T, int U1, int J, void* stream);
int mi355x_joint_combine_bwd(void* dh, const void* h, void* df, int dtype, float drop_scale, int B, int T, int U1, int J,
                             void* stream);
/* dst (dtype) [M, Np] (pitch ld_out) = alpha * src f32 [M, N] (pitch ld_in), columns N..Np-1 zero: GEMM operand rows */
int mi355x_cast_rows(const void* src, long long ld_in, void* dst, int dst_dtype, long long ld_out, long long M, int N, int Np,
                     float alpha, void* stream);

/* ---- Squeezeformer block glue (BASELINE.json configs[4]): squeezeformer_modules.py:30-57 (ScaleBiasLayer), :139-181 (post-LN
 * block order), conformer_modules.py:267-275 ('swish' point-wise activation), subsampling.py:589-646 (TimeReductionModule),
 * squeezeformer_encoder.py:352-361 (time recovery).  `ld` = row pitch of a produced GEMM operand (>= d, multiple of 4; columns
 * [d, ld) are zero-filled) so that d_model % 8 != 0 (Medium: 324) still gives 16-byte aligned bf16 rows. */
/* y (dtype) [M, ld] = x f32 [M, d] * scale + bias   (scale = bias = NULL: plain cast, adaptive_scale = False) */
int mi355x_scale_bias_fwd(const void* x, const void* scale, const void* bias, void* y, int y_dtype, long long M, int d, int ld,
                          void* stream);
/* dres f32 [M, d] += dy * scale;  dscale[d] += colsum(dy * x);  dbias[d] += colsum(dy)   (dscale = dbias = NULL: skipped);
 * scratch: f32 work space of at least min(M, 512) * 2 * d elements (per-workgroup partial column sums) */
int mi355x_scale_bias_bwd(const void* dy, int dy_dtype, int ld, const void* x, const void* scale, void* dres, void* dscale,
                          void* dbias, long long M, int d, void* scratch, long long scratch_elems, void* stream);
/* y (dtype) [M, ld] = alpha * dropmask(m*d + c) * x f32 [M, d]  -- the dropout mask of a GEMM epilogue with N = d */
int mi355x_cast_pitched(const void* x, void* y, int y_dtype, long long M, int d, int ld, float alpha, unsigned drop_key,
                        unsigned drop_threshold, float drop_scale, void* stream);
/* out [M, C] = swish(in) * (t < len[b]),  rows m = b*T + t;  backward: din = dout * swish'(in) * mask */
int mi355x_swish_mask_fwd(const void* in, void* out, int dtype, const void* len, int T, long long M, int C, void* stream);
int mi355x_swish_mask_bwd(const void* in, const void* dout, void* din, int dtype, const void* len, int T, long long M, int C,
                          void* stream);
/* masked depthwise Conv1d(k=5, stride 2, pad 3) over time, cropped to ceil(T/2) frames: x f32 [B,T,d] -> out (dtype)
 * [B*ceil(T/2), ld]; w f32 [d,1,5].  backward: dx f32 [B,T,d] +=, dw +=, dbias += */
int mi355x_time_reduce_dwconv_fwd(const void* x, const void* len, const void* w, const void* bias, void* out, int out_dtype, int B,
                                  int T, int d, int ld, void* stream);
int mi355x_time_reduce_dwconv_bwd(const void* dout, int dout_dtype, int ld, const void* x, const void* len, const void* w, void* dx,
                                  void* dw, void* dbias, int B, int T, int d, void* stream);
/* out f32 [B,T,d] = skip + ys[b, t/2, :]  (ys f32 [B, ceil(T/2), d]);  backward: dys (dtype) [B*ceil(T/2), ld] = dx[2t'] + dx[2t'+1] */
int mi355x_time_recover_fwd(const void* skip, const void* ys, void* out, int B, int T, int d, void* stream);
int mi355x_time_recover_bwd(const void* dx, void* dys, int dys_dtype, int B, int T, int d, int ld, void* stream);

/* ---- LayerNorm (torch.nn.LayerNorm x5 per layer, conformer_modules.py:174-215) -------------------------------- */
int mi355x_layernorm_fwd(const void* x, int x_dtype, const void* gamma, const void* beta, void* y, int y_dtype, void* mean,
                         void* rstd, int M, int d, float eps, void* stream);
/* y1 = LN(x; gamma1, beta1) in f32 and y2 = LN(y1; gamma2, beta2) in y2_dtype in one pass (d = 512, 32-byte aligned pointers):
 * a ConformerLayer's norm_out followed by the next layer's norm_feed_forward1 (conformer_modules.py:205-231, :174-178) */
int mi355x_layernorm2_fwd(const void* x, const void* gamma1, const void* beta1, void* y1, void* mean1, void* rstd1,
                          const void* gamma2, const void* beta2, void* y2, int y2_dtype, void* mean2, void* rstd2, int M, int d,
                          float eps, void* stream);
/* The backward of that pair in one pass (d = 512): g = dres_in + dLN1(dy1; x1) is the gradient w.r.t. y1 (kept in registers),
 * dres_out (f32) = dLN2(g; x2), cast_out (optional bf16) = cast_scale * dropmask * dres_out; the four parameter gradients are
 * accumulated (+=).  LN1 = the LATER LayerNorm in forward order (norm_feed_forward1 of layer i+1, input x1 = layer i's output),
 * LN2 = the earlier one (norm_out of layer i, input x2). */
int mi355x_layernorm2_bwd(const void* dy1, int dy1_dtype, const void* x1, const void* gamma1, const void* mean1, const void* rstd1,
                          void* dgamma1, void* dbeta1, const void* dres_in, const void* x2, const void* gamma2, const void* mean2,
                          const void* rstd2, void* dgamma2, void* dbeta2, void* dres_out, int M, int d, void* cast_out,
                          float cast_scale, unsigned drop_key, unsigned drop_threshold, float drop_scale, void* stream);
/* dres (f32 [M,d]) = (accumulate ? dres : 0) + dLN/dx ; dgamma/dbeta (f32 [d], may be NULL) are accumulated (+=)   */
int mi355x_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const void* gamma, const void* mean,
                         const void* rstd, void* dres, int accumulate, void* dgamma, void* dbeta, int M, int d, void* stream);
/* Same, and in the same pass cast_out (bf16 [M,d]) = cast_scale * dropmask * dres_new: the operand of the NEXT sub-block's
 * output-projection gradient GEMMs (the residual-branch gradient through that sub-block's dropout), which otherwise costs a
 * separate read of the fp32 gradient (mi355x_drop_scale_cast).  M*d % 8 == 0. */
int mi355x_layernorm_bwd_cast(const void* dy, int dy_dtype, const void* x, int x_dtype, const void* gamma, const void* mean,
                              const void* rstd, void* dres, int accumulate, void* dgamma, void* dbeta, int M, int d,
                              void* cast_out, float cast_scale, unsigned drop_key, unsigned drop_threshold, float drop_scale,
                              void* stream);
/* out[n] += alpha * sum_m x[m,n]  (bias / pos_bias gradients) */
int mi355x_colsum(const void* x, int x_dtype, long long ld, void* out, int M, int N, float alpha, void* stream);

/* ---- decoder log-softmax: conv_asr.py:468 ---------------------------------------------------------------------- */
int mi355x_log_softmax_fwd(const void* logits, long long ld_in, void* logp, long long ld_out, int M, int C, void* stream);
int mi355x_log_softmax_bwd(const void* dlogp, const void* logp, long long ld, void* dlogits, int out_dtype, long long ld_out,
                           int M, int C, float scale, void* stream);

/* ---- Conformer block glue (conformer_modules.py:324-331; multi_head_attention.py:259-270,305-307,343-346,137-140) */
/* row_offsets (optional, i64 [B+1], needs len): "PACKED ROWS" (length-aware kernels that skip padded frames) --
 * the [M, 2d] side (`in`, `din`) then holds only the valid frames of every utterance, utterance b at rows row_offsets[b] ..
 * row_offsets[b] + len[b] - 1, while the [M, d] side (`out`, `dout`) stays the padded [B*T, d] grid the depthwise convolution and
 * BatchNorm run on (their batch statistics include padded frames, conformer_modules.py:297,330-331).  M = B * T either way. */
int mi355x_glu_fwd(const void* in /*[M,2d]*/, void* out /*[M,d]*/, int dtype, const void* len, int T, long long M, int d,
                   const void* row_offsets, void* stream);
int mi355x_glu_bwd(const void* in, const void* dout, void* din, int dtype, const void* len, int T, long long M, int d,
                   const void* row_offsets, void* stream);
/* direction 0: packed[row_offsets[b] + t, 0:width] = padded[b*T + t, 0:width] for t < len[b];
 * direction 1: padded[b*T + t, 0:width] = t < len[b] ? packed[row_offsets[b] + t, 0:width] : 0 for every t < T.
 * M = B * T (rows of the padded grid), ld_* = row pitches in elements, width % (16 bytes / element size) == 0, dtype bf16 or f32. */
int mi355x_rows_pack(const void* src, void* dst, int dtype, long long ld_src, long long ld_dst, const void* len,
                     const void* row_offsets, int T, long long M, int width, int direction, void* stream);
int mi355x_drop_scale_cast(const void* in, int in_dtype, void* out, int out_dtype, long long n, float alpha,
                           unsigned drop_key, unsigned drop_threshold, float drop_scale, void* stream);
int mi355x_qbias(const void* qkv, long long ldq, const void* u, const void* v, void* qu, void* qv, int dtype, long long M,
                 int d, void* stream);
int mi355x_add2(const void* a, const void* b, int in_dtype, void* out, int out_dtype, long long ldo, long long M, int d,
                void* stream);
/* out[m, 0:d] = a + b (bf16, row pitch ldo) and, in the same pass, sum_ab[0:d] += column sums of a, sum_ab[d:2d] +=
 * column sums of b (f32).  Used for dq = dqu + dqv with the pos_bias_u / pos_bias_v gradients
 * (multi_head_attention.py:288-291): pos_bias_u.grad and pos_bias_v.grad must be adjacent ([2, d]).
 * scratch: f32 [ceil(M/32) * 2 * d] (two-stage reduction, no same-address atomics). */
int mi355x_add2_colsum(const void* a, const void* b, void* out, long long ldo, long long M, int d, void* sum_ab, void* scratch,
                       long long scratch_elems, void* stream);
/* ac f32 [H,B,T,Tp], bdf f32 [H,B,T,Pp] (bd before rel_shift) -> s (softmax, masked) and pd = dropout(s), pitch Tp   */
int mi355x_relpos_softmax_fwd(const void* ac, const void* bdf, void* s_out, void* pd_out, int out_dtype, const void* len,
                              int H, int B, int T, int Tp, int Pp, float scale, unsigned drop_key, unsigned drop_threshold,
                              float drop_scale, void* stream);
/* The same with a limited attention context (ConformerEncoder att_context_size = [left, right], att_context_style;
 * /root/reference/nemo/collections/asr/modules/conformer_encoder.py:794-823): ctx_style 0 = unlimited, 1 = 'regular', 2 =
 * 'chunked_limited'; a key outside the query's window is masked exactly like a padded one (multi_head_attention.py:137-146). */
int mi355x_relpos_softmax_fwd_ctx(const void* ac, const void* bdf, void* s_out, void* pd_out, int out_dtype, const void* len, int H,
                                  int B, int T, int Tp, int Pp, float scale, unsigned drop_key, unsigned drop_threshold,
                                  float drop_scale, int ctx_style, int ctx_left, int ctx_right, void* stream);
int mi355x_relpos_softmax_bwd(const void* dpd, int dpd_dtype, const void* s_in, void* dscore, void* dbdf, int s_dtype, int H,
                              int B, int T, int Tp, int Pp, float scale, unsigned drop_key, unsigned drop_threshold,
                              float drop_scale, void* stream);

/* ---- fused rel-pos attention (bf16; head width dk = 64 or 128 -- narrower / in-between heads are zero-padded inside the packed
 * weight images by the encoders: 44 -> 64, 81 -> 128): RelPositionMultiHeadAttention.forward, multi_head_attention.py:272-354.
 * qkv [B*T, ldq = 3d] (q|k|v), pos = linear_pos(pos_emb) [2T-1, ldp], bias_u/v f32 [H*d_k], len i64 [B]
 * -> ctx [B*T, ldo] bf16, lse f32 [B,H,T] (log-sum-exp of the masked scaled scores, kept for backward).
 * ctx_lo (optional, same layout as ctx): the bf16 ROUNDING RESIDUAL of ctx (O = ctx + ctx_lo to ~16 mantissa bits) for
 * mi355x_attn_delta -- the reference's softmax backward works on fp32 probabilities (multi_head_attention.py:137-138 under
 * autocast), i.e. without the error a delta taken from the rounded O alone puts on nearly-cancelling score gradients.
 * Dropout index of probability (b,h,i,j) = ((h*B+b)*T + i)*Tp + j.
 * row_offsets (optional, i64 [B+1]; every fused attention entry point takes it): PACKED ROWS -- the activation matrices (qkv, ctx,
 * ctx_lo; in backward qu, qv, dO, dqu, dqv, dq_out, dqkv) hold only the valid frames, utterance b at rows row_offsets[b] ..
 * row_offsets[b] + len[b] - 1, instead of b*T .. b*T + T - 1.  T stays the padded length: it fixes the positional geometry
 * (pos has 2T-1 rows) and the layout of the per-query statistics lse / delta [B,H,T] and of ds_out, which remain padded-indexed. */
int mi355x_relpos_flash_fwd(const void* qkv, long long ldq, const void* pos, long long ldp, const void* bias_u,
                            const void* bias_v, const void* len, void* ctx, void* ctx_lo, long long ldo, void* lse, int B, int H,
                            int T, int dk, int Tp, float scale, unsigned drop_key, unsigned drop_threshold, float drop_scale,
                            const void* row_offsets, void* stream);

/* backward of the fused attention.  delta[b,h,i] = sum_dv dO*(O + O_lo) (O_lo optional, see above).  dq kernel: qu = q+u, qv = q+v ([B*T,d] bf16, from
 * mi355x_qbias), recomputes P from lse, returns dQu and dQv ([B*T,d] bf16; dq = dQu+dQv, d pos_bias_{u,v} = column sums). */
int mi355x_attn_delta(const void* dO, const void* O, const void* O_lo, void* delta, int B, int H, int T, int d, const void* len,
                      const void* row_offsets, void* stream);
/* mi355x_attn_delta and mi355x_qbias (q = the first d columns of qkv rows of pitch ldq; multi_head_attention.py:288-291: q + pos_bias_u,
 * q + pos_bias_v) in ONE pass over the rows: everything the fused backward kernels need in front of them.  All pointers 16-byte
 * aligned, ldq % 8 == 0. */
int mi355x_attn_bwd_prep(const void* dO, const void* O, const void* O_lo, void* delta, const void* qkv, long long ldq,
                         const void* bias_u, const void* bias_v, void* qu, void* qv, int B, int H, int T, int d, const void* len,
                         const void* row_offsets, void* stream);
/* ds_out (optional): the score gradient in the layout of the reference's matrix_bd BEFORE rel_shift
 * (multi_head_attention.py:259-270), cut into 32 x 32 bf16 blocks for mi355x_relpos_flash_bwd_dpos:
 * X[h][b][it][s][q][cl] = dS[b,h, i = 32*it+q, j] at position c = T-1+j-i = T-32+32*(s-it)+cl, it < ceil(T/32), s <= ceil(T/32);
 * slots s <= ceil(len[b]/32) are written.  ds_elems = capacity of ds_out in elements, >= mi355x_relpos_ds_elems(B,H,T).
 * Outputs: dqu / dqv separately (both non-NULL), and / or dq_out = dQu + dQv as bf16 rows of pitch ld_dq (the q third of the
 * fused projection's [M, 3d] gradient) with bias_grads f32 [2 * H * 64] += column sums of dQu | dQv (d pos_bias_u | d pos_bias_v)
 * through cs_scratch, f32 [B * ceil(T / 128) * 2 * H * 64] (per-workgroup sums, added by a second-stage reduction launch). */
long long mi355x_relpos_ds_elems(int B, int H, int T);
int mi355x_relpos_flash_bwd_dq(const void* qu, const void* qv, const void* qkv, long long ldq, const void* pos, long long ldp,
                               const void* len, const void* dO, const void* lse, const void* delta, void* dqu, void* dqv,
                               void* ds_out, void* dq_out, long long ld_dq, void* bias_grads, void* cs_scratch,
                               long long cs_scratch_elems, int B, int H, int T, int dk, long long ds_elems, float scale,
                               unsigned drop_key, unsigned drop_threshold, float drop_scale, const void* row_offsets, void* stream);

/* dK and dV rows written into the k / v column blocks of dqkv [B*T, ldd = 3d] */
int mi355x_relpos_flash_bwd_dkv(const void* qu, const void* qv, const void* qkv, long long ldq, const void* pos, long long ldp,
                                const void* len, const void* dO, const void* lse, const void* delta, void* dqkv, long long ldd,
                                int B, int H, int T, int dk, int Tp, float scale, unsigned drop_key, unsigned drop_threshold,
                                float drop_scale, const void* row_offsets, void* stream);

/* dpos f32 [2T-1, ldd] += gradient w.r.t. pos = linear_pos(pos_emb) (multi_head_attention.py:296-300 backward), summed over
 * the batch: dpos[c, h, :] += sum_{b,i} dS[b,h,i,c-(T-1)+i] * qv[b,i,h,:], from the blocks written by
 * mi355x_relpos_flash_bwd_dq (ds_out).  partial (optional): f32 scratch of >= mi355x_relpos_dpos_partial_elems(B,H,T)
 * elements -> deterministic two-stage reduction instead of atomics.  dpos_cast (optional, needs `partial`): bf16 [2T-1, ldd]
 * copy of the updated dpos, written by the reduction stage (the operand of the linear_pos weight-gradient GEMM). */
long long mi355x_relpos_dpos_partial_elems(int B, int H, int T);
int mi355x_relpos_flash_bwd_dpos(const void* qv, const void* ds, const void* len, void* dpos, long long ldd, void* dpos_cast,
                                 void* partial, long long partial_elems, int B, int H, int T, int dk, long long ds_elems,
                                 const void* row_offsets, void* stream);

/* which depthwise-convolution kernels run (tests and A/B): 0 = the LDS-tile kernels (default: faster inside the training step),
 * 1 = the streaming kernel in the forward pass (bf16, k = 31, even d), 2 = in the backward pass too; level < 0 only queries.
 * Returns the previous level.
 * Environment: MI355X_DWCONV_STREAM.  (CausalConv1D depthwise, parts/submodules/conformer_modules.py:333-337) */
int mi355x_dwconv_config(int level);

/* ---- convolution module: depthwise conv + BatchNorm + Swish (conformer_modules.py:333-342, causal_convs.py:130-147) */
int mi355x_dwconv_fwd(const void* x, const void* w /*[d,1,k]*/, const void* bias, void* y, int dtype,
                      void* stats /*f64 [2,d] += (sum, sumsq) or NULL*/, int B, int T, int d, int ksize, void* stream);
/* GLU (+ pad mask) fused into the depthwise forward (conformer_modules.py:324-335: glu -> masked_fill -> depthwise_conv).  glu_in
 * [rows, 2d] = the pointwise conv's output (rows = B*T, or the packed valid frames with row_offsets i64 [B+1] as in mi355x_glu_fwd);
 * len i64 [B] (NULL: every frame valid); glu_out [B,T,d] receives the GLU output (zeros beyond len; backward's operand);
 * w, bias, y, stats as in mi355x_dwconv_fwd.  d must be a whole number of 16-byte chunks.  act: 0 = GLU as described, 1 = Swish of a
 * [rows, d] input instead (ConformerConvolution(pointwise_activation='swish'), squeezeformer_modules.py:60-203). */
int mi355x_dwconv_fwd_glu(const void* glu_in, const void* len, const void* row_offsets, void* glu_out, const void* w,
                          const void* bias, void* y, int dtype, void* stats, int B, int T, int d, int ksize, int act, void* stream);
int mi355x_dwconv_bwd(const void* dy, const void* x, const void* w, void* dx, void* dw, void* dbias, int dtype, int B, int T,
                      int d, int ksize, void* scratch /* optional f32 [4*B*(ksize+1)*d]: two-stage reduction, no atomics */,
                      long long scratch_elems, void* stream);
/* The depthwise convolution with asymmetric zero padding (ConformerConvolution's CausalConv1D, conv_context_size = [left, right] with
 * left + right + 1 = ksize; /root/reference/nemo/collections/asr/parts/submodules/causal_convs.py:89-150, conformer_modules.py:
 * 310-321): pad_left frames in front of the sequence, ksize - 1 - pad_left behind it; -1 = symmetric. */
int mi355x_dwconv_fwd_ctx(const void* x, const void* w, const void* bias, void* y, int dtype, void* stats, int B, int T, int d,
                          int ksize, int pad_left, void* stream);
int mi355x_dwconv_bwd_ctx(const void* dy, const void* x, const void* w, void* dx, void* dw, void* dbias, int dtype, int B, int T, int d,
                          int ksize, int pad_left, void* scratch, long long scratch_elems, void* stream);
int mi355x_bn_finalize(const void* stats, double count, void* mean, void* rstd, void* running_mean, void* running_var,
                       float momentum, float eps, int d, void* stream);
/* same, the element count read from device memory (f64 scalar): under SyncBatchNorm (conformer_ctc_bpe.yaml:209,
 * torch.nn.SyncBatchNorm gathers the per-rank counts) it is all-reduced together with the sums */
int mi355x_bn_finalize_dev_count(const void* stats, const void* count_dev, void* mean, void* rstd, void* running_mean,
                                 void* running_var, float momentum, float eps, int d, void* stream);
int mi355x_bn_eval_stats(const void* running_mean, const void* running_var, void* mean, void* rstd, float eps, int d,
                         void* stream);
int mi355x_bn_swish_fwd(const void* x, const void* mean, const void* rstd, const void* gamma, const void* beta, void* y,
                        int dtype, long long M, int d, void* stream);
/* training forward in one launch (nn.BatchNorm1d in training mode + Swish, conformer_modules.py:339-342): mean / rstd from the f64
 * sums (as mi355x_bn_finalize: `count` positions, or the count read from device memory when count_dev != NULL), written out for
 * backward, running statistics updated, y = swish(BN(x)).  Measured slower than mi355x_bn_finalize + mi355x_bn_swish_fwd at the
 * Large shape (every workgroup derives its channels' coefficients; tools/bn_bench.py) -- the encoders use the pair. */
int mi355x_bn_stats_swish_fwd(const void* x, const void* stats, double count, const void* count_dev, const void* gamma,
                              const void* beta, void* y, void* mean, void* rstd, void* running_mean, void* running_var,
                              float momentum, float eps, int dtype, long long M, int d, void* stream);
/* dgamma / dbeta (optional, both or neither): f32 [d] += the parameter gradients, i.e. the LOCAL sums (what
 * mi355x_bn_param_grad adds), accumulated by the reduction's second stage */
int mi355x_bn_swish_bwd_reduce(const void* dy, const void* x, const void* mean, const void* rstd, const void* gamma,
                               const void* beta, void* sums /*f64 [2,d] +=*/, void* dgamma, void* dbeta, int dtype, long long M,
                               int d, void* scratch /* optional f32 [ceil(M/32)*2*d]: two-stage reduction */,
                               long long scratch_elems, void* stream);
int mi355x_bn_swish_bwd_apply(const void* dy, const void* x, const void* mean, const void* rstd, const void* gamma,
                              const void* beta, const void* sums, double count, int training, void* dx, int dtype,
                              long long M, int d, void* stream);
int mi355x_bn_swish_bwd_apply_dev_count(const void* dy, const void* x, const void* mean, const void* rstd, const void* gamma,
                                        const void* beta, const void* sums, const void* count_dev, int training, void* dx,
                                        int dtype, long long M, int d, void* stream);
/* BatchNorm + Swish backward fused into the depthwise backward (the [B,T,d] gradient between mi355x_bn_swish_bwd_apply and
 * mi355x_dwconv_bwd is never materialised; same results, the intermediate rounded to the activation type as the two-launch form
 * rounds it).  dy = gradient w.r.t. the Swish output, cc = BatchNorm input (the depthwise conv's output), sums = f64 [2][d] left by
 * mi355x_bn_swish_bwd_reduce (all-reduced under SyncBatchNorm), count > 0 or count_dev (device f64).  x, w, dx, dw, dbias, scratch
 * as in mi355x_dwconv_bwd.  Optionally the GLU backward on the way out as well (glu_in != NULL: [rows, 2d] input of the GLU,
 * glu_din [rows, 2d] receives its gradient INSTEAD of dx; glu_len i64 [B] valid frames, zeros beyond them; glu_row_offsets i64 [B+1]:
 * glu_in / glu_din hold packed rows -- mi355x_glu_bwd's semantics; glu_act 1: Swish backward of a [rows, d] input instead,
 * mi355x_swish_mask_bwd's semantics).  Replaces conformer_modules.py:333-342 backward
 * (glu -> depthwise_conv -> batch_norm -> activation, walked backwards). */
int mi355x_dwconv_bwd_bnswish(const void* dy, const void* cc, const void* mean, const void* rstd, const void* gamma,
                              const void* beta, const void* sums, double count, const void* count_dev, int training, const void* x,
                              const void* w, void* dx, void* dw, void* dbias, const void* glu_in, void* glu_din, const void* glu_len,
                              const void* glu_row_offsets, int glu_act, int dtype, int B, int T, int d, int ksize, void* scratch,
                              long long scratch_elems, int defer_tap_reduce, void* stream);
/* defer_tap_reduce != 0 above: the kernel leaves the B * 4 partial slabs of the tap / bias gradients in `scratch` (which then has to
 * be the caller's own per layer) and THIS call adds them into dw [d, 1, k] / dbias [d] -- on whichever stream the caller likes:
 * only the optimizer reads them (depthwise_conv.weight.grad, conformer_modules.py:333). */
int mi355x_dwconv_tap_reduce(const void* scratch, long long scratch_elems, int B, int d, int ksize, void* dw, void* dbias,
                             void* stream);
int mi355x_bn_param_grad(const void* sums, void* dgamma, void* dbeta, int d, void* stream);

/* ---- CTC loss: CTCLoss.forward, losses/ctc.py:68-82 (torch ctc_loss, blank = V, zero_infinity) ------------------
 * logp f32 [B,Tmax,C]; targets i64 [B,Umax]; workspaces f32 [B,Tmax,2*Umax+1] each; nll f32 [B];
 * grad f32 [B,Tmax,C] = grad_scale * d(sum_b nll_b)/d logp  (may be NULL for loss only).                          */
int mi355x_ctc_loss(const void* logp, const void* targets, const void* in_len, const void* tgt_len, void* alpha_ws,
                    void* beta_ws, void* nll, void* grad, int B, int Tmax, int C, int Umax, int blank, float grad_scale,
                    int zero_infinity, void* stream);
/* lattice kernel of mi355x_ctc_loss: 1 (default; MI355X_CTC_WAVE) = alpha / beta rows resident in the registers of one wave each
 * (2U+1 <= 1024 states, no barrier per time-step), 0 = the LDS / barrier form for every size.  Returns the previous setting
 * (-1: not yet resolved from the environment). */
int mi355x_ctc_config(int wave);

/* x[r,:] *= vec[r] (f32): per-utterance upstream gradient applied to the CTC gradient (autograd of losses/ctc.py:52-66) */
int mi355x_row_scale(void* x, const void* vec, long long rows, long long cols, void* stream);

/* ---- optimizer / weight packing (modelPT.py:650-823 AdamW; no reference analogue for packing) ------------------ */
int mi355x_adamw_step(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, long long n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
/* AdamW with (a) an optional DEVICE scalar clip_coef[0] multiplied into the gradient -- global-norm clipping
 * (trainer.gradient_clip_val -> torch.nn.utils.clip_grad_norm_) without a host sync -- and (b) an optional EMA of the
 * weights updated in the same pass: ema = decay*ema + (1-decay)*w_new (nemo/collections/common/callbacks/ema.py:150-157). */
int mi355x_adamw_step_ex(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, long long n, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int step, float grad_scale, const void* clip_coef,
                         void* ema, float ema_decay, void* stream);
/* out_f64[0] += sum(grads^2) ; coef[0] = min(1, max_norm / (scale*sqrt(sum_i sumsq[i]) + 1e-6)), coef[1] = the norm */
int mi355x_grad_sumsq(const void* grads, long long n, void* out_f64, void* stream);
int mi355x_clip_coef(const void* sumsq_f64, int nbuf, float scale, float max_norm, void* coef_f32x2, void* stream);
typedef struct mi355x_pack_entry {
  const void* src; void* dst;       /* src f32; dst[r*pitch + c] = src[r1*sr1 + r2*sr2 + c1*sc1 + c2*sc2]              */
  int rows, cols, nr2, nc2;         /* r = r1*nr2 + r2 ; c = c1*nc2 + c2                                              */
  long long sr1, sr2, sc1, sc2, pitch;
  long long tile_begin;             /* exclusive prefix sum of ceil(rows/64)*ceil(cols/64); dst 16-B aligned           */
} mi355x_pack_entry;
int mi355x_pack_weights(const void* table_dev, int n_entries, long long total_tiles, int out_dtype, void* stream);
int mi355x_fill_f32(void* p, long long n, float value, void* stream);

/* library / build information */
const char* mi355x_asr_version(void);

/* Replayable launch sequences (the reference's analogue: nemo/utils/callbacks/cuda_graph.py:251, whole-step CUDA-graph capture).
 * A training step captured into a hipGraph freezes every kernel argument, including the dropout keys (drop_key above is a
 * by-value argument).  With a step word registered here, every kernel that takes a dropout key adds *dev_word (u32, device
 * memory) to the key at entry; the host advances the word once per step in front of the forward graph, so forward and
 * backward of one step regenerate the same masks and consecutive steps different ones.  The pointer is read when a launch is
 * ISSUED (it becomes a kernel argument): set it around the capture, reset it to NULL afterwards; launches issued with NULL
 * use their key as passed.  Process-wide. */
int mi355x_set_step_counter(const void* dev_word);
/* Measurement switch: with on = 1 every launch site of the library issues an EMPTY kernel on its stream instead of its own kernel
 * (results are garbage): the host pays the per-launch cost of a step while the GPU stays idle, i.e. the pure issue time of the
 * launch sequence (tools/host_phases.py).  Returns the previous value. */
int mi355x_set_null_launch(int on);

/* Launch tapes (csrc/tape.hip): a launch sequence captured as a hipGraph, re-issued as LIVE launches from one C loop.
 * What it replaces: the host-side walk over the encoder's modules -- nemo/collections/asr/modules/conformer_encoder.py:593-759
 * (forward_internal: pre_encode, pos_enc, the layer loop) and parts/submodules/conformer_modules.py:164-215 (one layer) -- once
 * a (shape, configuration) has been seen; the reference's own mechanism for the same cost is whole-step CUDA-graph capture
 * (nemo/utils/callbacks/cuda_graph.py:251).  A replayed hipGraph measured 3-4 % slower on the device than live launches on
 * this stack, the Python sequencer costs ~19 us of host time per launch: a tape issues the graph's nodes (kernels with
 * their frozen arguments, memsets, memcpys) in topological order with hipLaunchKernel on stream lanes -- lane 0 is the
 * stream passed to replay, lanes 1.. are the OTHER streams that took part in the capture (the weight-gradient side stream),
 * used again as they are -- with one event per cross-lane dependency.
 *   mi355x_tape_log_begin(origin)  before the capture starts: from here on every launch of the library notes the stream it is
 *                                  captured on (`origin` = the capturing stream -> lane 0; other streams -> lanes 1, 2, ...)
 *   mi355x_tape_from_graph         hip_graph = hipGraph_t (kept alive by the caller for as long as the tape is used: the kernel
 *                                  arguments stay inside its nodes); returns 0, 1 (bad argument), 2 (a node type a tape cannot
 *                                  re-issue, or arguments passed through `extra`: keep replaying the graph) or 1000 + hipError_t
 *   mi355x_tape_log_end()          after the last mi355x_tape_from_graph of the capture
 *   mi355x_tape_replay(t, s, join) join = 1: the ordering contract of hipGraphLaunch(exec, s) (s continues behind every lane);
 *                                  join = 0: the side lanes run on, as behind the live sequencer -- consumers of their results
 *                                  order themselves behind those streams (the optimizer slice / gradient bucket of a layer does)
 *   mi355x_tape_join(t, s)         s continues behind whatever t's side lanes hold now (what a replay with join = 0 left out)
 *   mi355x_tape_info               counts[6] = kernels, memsets, memcpys, empty nodes, lanes, cross-lane events */
/* A stream that belongs to its caller alone (non-blocking, HIP priority clamped to the device's range: 0 = default, negative =
 * more urgent).  torch.cuda.Stream() objects come out of a 32-entry round-robin pool and are SHARED once a process has created
 * more than 32: a capture stream, a weight-gradient stream and a copy stream must not be the same stream (csrc/tape.hip). */
int mi355x_stream_create(int priority, void** out_stream);
int mi355x_stream_destroy(void* stream);
typedef struct mi355x_tape mi355x_tape;
int mi355x_tape_log_begin(void* origin_stream);
int mi355x_tape_log_end(void);
int mi355x_tape_from_graph(void* hip_graph, int max_lanes, mi355x_tape** out);
int mi355x_tape_replay(mi355x_tape* tape, void* stream, int join);
int mi355x_tape_join(mi355x_tape* tape, void* stream);
int mi355x_tape_info(const mi355x_tape* tape, int* counts);
void mi355x_tape_destroy(mi355x_tape* tape);

/* Statistics mailbox (csrc/mailbox.hip): the SyncBatchNorm exchanges as ONE kernel launch each instead of a process-group
 * all_reduce.  What it replaces: the per-layer all_reduce of torch.nn.SyncBatchNorm (`trainer.sync_batchnorm: true`,
 * examples/asr/conf/conformer/conformer_ctc_bpe.yaml:209, over the BatchNorm1d of parts/submodules/conformer_modules.py:339 and
 * its backward) -- 36 latency-bound 8-KB collectives per Conformer-CTC-Large step that otherwise queue on the same RCCL stream
 * as the 64-MiB gradient buckets.  Every rank owns a mailbox in its HBM and maps its peers' through hipIpc handles; an exchange
 * stores the rank's n f64 values into every peer's mailbox (peer stores over xGMI), raises a sequence flag, waits for the `world`
 * flags in its own mailbox and sums the boxes in rank order (every rank gets bit-identical sums).  The sequence number lives
 * in device memory: a launch carries no per-call host state.
 *   mi355x_mailbox_create   collective in spirit (every rank creates one, same world / n_max): allocates + zeroes the mailbox and
 *                           writes its MI355X_MAILBOX_HANDLE_BYTES-byte IPC handle to handle_out -- the caller carries the handles
 *                           to the peers (torch.distributed.all_gather in nemo_amd/mailbox.py).  timeout_ms <= 0: 2000.
 *                           mem_kind: 0 = the first kind of memory that can be exported (uncached, fine-grained, plain), 1 | 2 | 3 = that one.
 *   mi355x_mailbox_open     map peer `peer`'s mailbox from its handle (peer == rank: no-op)
 *   mi355x_mailbox_exchange stats (f64 [n], n <= n_max, device memory) <- sum over ranks, in place, on `stream`
 *   mi355x_mailbox_status   out3 = {exchanges completed, 0 or 1 + the rank whose flag never arrived (latched: the exchange that timed
 *                           out and every later one return at once with `stats` set to NaN -- never the local sums -- and the
 *                           sequence number does not advance), kind of memory (1 uncached, 2 fine-grained, 3 plain)};
 *                           blocking copy -- diagnostics and tests
 *   mi355x_mailbox_poll     the same header WITHOUT blocking, for the training loop (once per optimizer step): enqueues a 16-byte
 *                           copy to pinned host memory on `stream` and reports what the previous poll's copy brought back;
 *                           out4 = {exchanges completed, latch, kind of memory, 1 if the values are new since the last call}
 * Returns 0, 1 (bad argument / a peer not opened), 2 (no memory of any kind) or 1000 + hipError_t. */
#define MI355X_MAILBOX_HANDLE_BYTES 64
typedef struct mi355x_mailbox mi355x_mailbox;
int mi355x_mailbox_create(int world, int rank, int n_max, int timeout_ms, int mem_kind, mi355x_mailbox** out, void* handle_out);
int mi355x_mailbox_open(mi355x_mailbox* mb, int peer, const void* handle);
int mi355x_mailbox_exchange(mi355x_mailbox* mb, void* stats_f64, int n, void* stream);
int mi355x_mailbox_status(mi355x_mailbox* mb, long long* out3);
int mi355x_mailbox_poll(mi355x_mailbox* mb, void* stream, long long* out4);
void mi355x_mailbox_destroy(mi355x_mailbox* mb);

/* Greedy CTC decoding on the device (GreedyCTCInfer._greedy_decode_logprobs, parts/submodules/ctc_greedy_decoding.py:333-361,
 * + the CTC collapse of AbstractCTCDecoding.decode_hypothesis, parts/submodules/ctc_decoding.py:545-575):
 * logp f32 [B,T,C], lens i64 [B] (NULL = T) -> tokens i32 [B,T] (folded, blank-free, -1 padded), out_len i32 [B],
 * score f32 [B] = sum of the arg-max log-probs of the non-blank frames. */
int mi355x_ctc_greedy_decode(const void* logp, const void* lens, void* tokens, void* out_len, void* score, int B, int T, int C,
                             int blank, void* stream);

/* SpectrogramAugmentation (nemo/collections/asr/modules/audio_preprocessing.py:443-553; SpecAugment._apply_masks
 * parts/submodules/spectr_augment.py:153-215, SpecCutout.forward :245-261): x[b, f0:f1, t0:t1] = value for n rectangles
 * rects[n][5] = (b, f0, f1, t0, t1) (int32, device memory; clipped to the tensor).  x: f32 [B, F, T], in place. */
int mi355x_fill_rects(void* x, const void* rects, int n, int B, int F, int T, float value, void* stream);
/* SpecAugment mask parameters (SpectrogramAugmentation's vectorised path, parts/submodules/spectr_augment.py:155-195) from the four
 * uniform draws the reference makes -- u_time_width, u_time_start f32 [B, time_masks]; u_freq_width, u_freq_start f32
 * [B, freq_masks] -- and the feature lengths len i64 [B]: rects i32 [B * (time_masks + freq_masks), 5] = (b, f0, f1, t0, t1), time
 * masks first, in the reference's f32 arithmetic (time_width: a fraction of the utterance when time_width_is_fraction, else frames). */
int mi355x_specaug_rects(const void* u_time_width, const void* u_time_start, const void* u_freq_width, const void* u_freq_start,
                         const void* len, void* rects, int B, int time_masks, int freq_masks, int F, int T, float time_width,
                         int time_width_is_fraction, int freq_width, void* stream);

/* ---- RNN-Transducer loss (FastConformer-Transducer, cfg 4) --------------------------------
 * Replaces the Numba-CUDA kernels behind RNNTLossNumba (nemo/collections/asr/parts/numba/rnnt_loss/rnnt_pytorch.py:39-98 ->
 * utils/cuda_utils/gpu_rnnt.py:125-231 -> gpu_rnnt_kernel.py:74-407 alphas / betas / grads, reduce.py denominator,
 * rnnt_helper.py:107-116 costs).  acts f32 [B,T,U1,V1] = joint LOGITS (the log-softmax is fused, as in the reference's
 * GPU path); labels i64 [B,U1-1] (padded); act_lens, label_lens i64 [B]; costs f32 [B] = -(1+fastemit_lambda) * log P(y|x);
 * grads f32 [B,T,U1,V1] (NULL = loss only) = grad_scale * d cost / d acts with the FastEmit term (gpu_rnnt_kernel.py:364-376)
 * and clamp (> 0: clip to [-clamp, clamp], :392-396); cells beyond an utterance's (T_b, U_b+1) are written as zeros.
 * workspace f32, at least mi355x_rnnt_workspace_elems(B,T,U1) elements.  U1 <= 1024.                                     */
int mi355x_rnnt_workspace_elems(int B, int T, int U1, long long* elems);
int mi355x_rnnt_loss(const void* acts, const void* labels, const void* act_lens, const void* label_lens, int B, int T, int U1,
                     int V1, int blank, float fastemit_lambda, float clamp, float grad_scale, void* costs, void* grads,
                     void* workspace, long long workspace_elems, void* stream);
/* Same, for the fused joint + loss path: logit rows with pitch ld_acts (>= V1; a multiple of 8 lets the joint GEMM store them
 * with vector accesses although V+1 = 1025 is odd), and the gradient written in grads_dtype with row pitch ld_grads --
 * MI355X_DT_BF16 with ld_grads % 8 == 0 is directly the K-contiguous operand of the joint's backward GEMMs (columns
 * [V1, ld_grads) zero-filled): no f32 gradient tensor, no cast pass. */
int mi355x_rnnt_loss_ex(const void* acts, long long ld_acts, const void* labels, const void* act_lens, const void* label_lens,
                        int B, int T, int U1, int V1, int blank, float fastemit_lambda, float clamp, float grad_scale, void* costs,
                        void* grads, int grads_dtype, long long ld_grads, void* workspace, long long workspace_elems,
                        void* stream);

/* ---- greedy transducer decoding on the device: GreedyBatchedRNNTInfer, nemo/collections/asr/parts/submodules/
 * rnnt_greedy_decoding.py:529-990 (batched greedy search; per utterance: state = 0, last = blank; for every frame t < enc_len[b], up
 * to max_symbols times: g = LSTM(emb[last], state), logits = out(relu(enc_proj[b,t] + pred(g))), k = argmax; blank -> next frame,
 * else emit (k, t), commit the state, last = k).  ONE launch for the batch, no host synchronisation per frame (the reference reads
 * `blank_mask.all()` back per inner iteration).  enc_proj [B, T, J] (f_dtype, row pitch ldf) = joint.enc(encoder output); emb f32
 * [V1, H] (row `blank` zero); one LSTM layer, torch gate order i, f, g, o: w_ih / w_hh [4H, H], w_pred [J, H], w_out [V1, J] in w_dtype
 * with row pitches ld_*; biases f32.  tokens / times i32 [B, max_out] (-1 padded; times optional), out_len i32 [B], score f32 [B]
 * (optional: sum of the emitted labels' log-probabilities), h_out / c_out f32 [B, H] (optional, both or neither: the final state).
 * max_symbols <= 0: unlimited.  H, J and the pitches multiples of 4. */
int mi355x_rnnt_greedy_decode(const void* enc_proj, int f_dtype, long long ldf, const void* enc_len, const void* emb,
                              const void* w_ih, long long ld_ih, const void* w_hh, long long ld_hh, const void* b_ih,
                              const void* b_hh, const void* w_pred, long long ld_pred, const void* b_pred, const void* w_out,
                              long long ld_out, const void* b_out, int w_dtype, int B, int T, int J, int H, int V1, int blank,
                              int max_symbols, void* tokens, void* times, void* out_len, void* score, int max_out, void* h_out,
                              void* c_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_ASR_H */
