/*
 * emogest.h -- C ABI of libemogest_hip.so: the MI355X (gfx950) implementation of the
 * EmotionGesture audio->gesture hot path.
 *
 * The reference (XingqunQi-lab/EmotionGestures) is pure Python/PyTorch and has no FFI; the
 * boundary it exposes for this path is the nn.Module surface used by
 * test_emotion_gesture_diversity_iterative.py:25-30,135-174,203-205.  Each entry point below
 * names the reference function it replaces (paths relative to the upstream repo).  The Python
 * host mirror (emotiongestures_amd.Full_model.*, emotiongestures_amd.CAVE.*) binds these with
 * ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Contract (all entry points):
 *   - plain C types only; every pointer named "d_*"/"arena"/"workspace" or documented as
 *     device memory is a HIP device pointer owned by the CALLER (PyTorch's caching allocator
 *     in the host mirror).  The library never allocates, frees or copies device memory behind
 *     the caller's back and never synchronises the device.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing touches
 *     the default stream.
 *   - return value: EG_OK (0) or a negative EgStatus; HIP launch errors are surfaced as
 *     EG_ERR_HIP.  No exceptions cross the ABI.  eg_last_error() gives a thread-local message.
 *   - stateless and re-entrant: EgGenerator / EgCvae handles are immutable host-side plans
 *     (offset tables); they hold no device memory and may be shared by threads.
 *   - activations are fp32 in HBM.  Convolution activations are NHWC inside the library; every
 *     tensor crossing this ABI uses the reference's own layout (stated per function).
 */
#ifndef EMOGEST_H
#define EMOGEST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum EgStatus {
    EG_OK = 0,
    EG_ERR_BAD_ARG = -1,        /* null pointer, negative size, inconsistent shape */
    EG_ERR_UNSUPPORTED = -2,    /* shape / channel count the kernels are not built for */
    EG_ERR_WORKSPACE = -3,      /* workspace too small (see *_workspace_bytes) */
    EG_ERR_HIP = -4,            /* hipGetLastError() != hipSuccess after a launch */
    EG_ERR_ALIGN = -5           /* pointer or leading dimension not 16-byte aligned */
} EgStatus;

/* Arithmetic mode of the contraction kernels (conv / GEMM).  Storage is fp32 in both. */
typedef enum EgPrecision {
    EG_PREC_F32 = 0,            /* v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate */
    EG_PREC_BF16X3 = 1,         /* split-bf16: hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 */
    EG_PREC_BF16 = 2            /* single bf16 product (fast mode; not parity-grade) */
} EgPrecision;

const char* eg_last_error(void);
const char* eg_version(void);
/* Process-wide default for the contraction kernels (EgPrecision).  Per-call structs below
 * carry their own field; this only seeds eg_*_default_config. */
int eg_set_default_precision(int precision);
int eg_get_default_precision(void);

/* Number of kernel launches the library has issued in this process (every launch function counts; captured launches count when they are
 * recorded, not when a hipGraph replays them).  bench.py reports launches per step from differences of this counter. */
int64_t eg_launch_count(void);
/* Diagnostic (process started with EG_LAUNCH_HIST=1, else empty): launches by launch-site label since the last reset, as "label count\n" lines
 * written NUL-terminated into buf (truncated to cap); returns the bytes the full text needs.  reset != 0 clears the counts. */
int64_t eg_launch_histogram(char* buf, int64_t cap, int32_t reset);
/* Optional per-launch timing of the contraction kernels (bench.py's roofline leg; a debugging facility, process
 * global, not thread safe).  While enabled, every eg_conv3x3 / eg_linear launch is bracketed by a hipEvent pair on its
 * own stream (no synchronisation).  eg_profile_read synchronises those events and returns, per record, a tag
 * (conv: cin*1000000 + cout*1000 + stride*100 + 1; linear: 2 = LDS-DMA kernel, 3 = pre-split-input kernel, 4 = causal-shift
 * kernel, 5 = f32 kernel), its work in FLOP and its duration in ms. */
int eg_profile_enable(int32_t max_records);
int eg_profile_disable(void);
int32_t eg_profile_read(int64_t* tags, double* flops, float* ms, int32_t capacity);
/* Workgroups of each recorded launch (0: not recorded -- a grid that covers the chip), in record order; call BEFORE eg_profile_read (which
 * resets the record list).  A product on 68 workgroups occupies a quarter of the 256 CUs for its duration: bench.py weighs a launch's
 * duration with min(1, workgroups / 256) when it reports the CU time of a kernel family beside its stand-alone duration. */
int32_t eg_profile_read_workgroups(int32_t* workgroups, int32_t capacity);

/* ------------------------------------------------------------------------------------------
 * Weight arena.  A model's parameters live in ONE caller-owned fp32 device buffer ("arena")
 * in kernel-ready layouts.  The library is authoritative for the layout: it publishes a
 * manifest (one entry per packed tensor) naming the reference state_dict key(s) each entry is
 * built from and the packing rule; the host packs on the CPU once per load_state_dict and
 * uploads.  Offsets and sizes are in floats.
 * ------------------------------------------------------------------------------------------ */
typedef enum EgPackKind {
    EG_PACK_RAW = 0,            /* tensor copied as is (row-major) */
    EG_PACK_LINEAR = 1,         /* nn.Linear weight [N,K] -> fp32 [Npad,Kpad] + tile-planar bf16 hi/lo images; Npad,Kpad % 64 == 0
                                   (dims = N,K,Npad,Kpad; numel = 2*Npad*Kpad floats) */
    EG_PACK_VEC_PAD = 2,        /* 1-D tensor zero padded to dims[1] (dims = n, npad) */
    EG_PACK_CONV3X3 = 3,        /* Conv2d OIHW [O,I,3,3] -> [tap][I/4][Opad][4] (dims = O,I,Opad) */
    EG_PACK_BN_SCALE = 4,       /* key = BN prefix: weight/sqrt(running_var+eps), padded to dims[1] with 0 */
    EG_PACK_BN_SHIFT = 5,       /* key = BN prefix: bias - running_mean*scale, padded to dims[1] with 0 */
    EG_PACK_CONV1X1 = 6,        /* Conv2d [O,I,1,1] -> [I][O] */
    EG_PACK_STEM = 7,           /* Conv2d [O,1,3,3] -> [9][O] */
    EG_PACK_WN_TAP = 8,         /* key = weight-norm conv prefix (weight_g, weight_v [O,I,k]); tap dims[2] of
                                   g*v/||v|| as [O,Ipad] (dims = O,I,tap,Ipad) */
    EG_PACK_CONV1D = 9,         /* Conv1d [O,I,k] -> [O][I][k] raw (alias of RAW, kept for readability) */
    EG_PACK_POS_TABLE = 10,     /* buffer [1,n_position,D] -> first dims[0] rows [frames,D] */
    EG_PACK_LINEAR_T = 11,      /* nn.Linear weight [N,K] -> transposed [K,N] */
    EG_PACK_LINEAR_FOLD = 12,   /* key = chain "last@...@first" of nn.Linear prefixes with only Dropout between them (eval mode: an
                                   affine chain): W = W_last ... W_first folded in float64, packed as EG_PACK_LINEAR (same dims);
                                   several chains joined by '|' are concatenated along N */
    EG_PACK_BIAS_FOLD = 13      /* the folded chain's bias, zero padded to dims[1] */
} EgPackKind;

typedef struct EgWeightEntry {
    char key[192];              /* reference state_dict key (or module prefix for BN / weight-norm kinds) */
    int32_t kind;               /* EgPackKind */
    int32_t dims[4];
    int64_t offset;             /* floats from arena base; 64-byte aligned */
    int64_t numel;              /* packed size in floats */
} EgWeightEntry;

/* ------------------------------------------------------------------------------------------
 * Generator = Transformer (Full_model/Models_spatial_memory.py:471-616, Full_model/Models_memory.py:426-565)
 * ------------------------------------------------------------------------------------------ */
typedef struct EgGeneratorConfig {
    int32_t frames;             /* Transformer(frames=...)       :475 */
    int32_t pose_dim;           /* pose_dim                       :475 */
    int32_t prior_frames;       /* prior_frames                   :475 */
    int32_t chunk;              /* args.chunk                     :263 */
    int32_t d_model;            /* must be 512-class: multiple of 64 */
    int32_t d_inner;
    int32_t n_layers;
    int32_t n_head;
    int32_t d_k;                /* == d_v */
    int32_t n_mels;             /* spectrogram rows (128) */
    int32_t spec_len;           /* spectrogram columns (124 for 4 s) */
    int32_t text_len;           /* 60 (Linear(60,60), :164-166) */
    int32_t n_words;            /* lang_model.n_words */
    int32_t embed_dim;          /* args.wordembed_dim (300) */
    int32_t tcn_hidden;         /* args.hidden_size (300) */
    int32_t tcn_layers;         /* args.n_layers (3) */
    int32_t variant;            /* 0 = Models_spatial_memory (SP_v2 no-op), 1 = Models_memory (SP_v1 + TM) */
    int32_t precision;          /* EgPrecision */
    int32_t n_position;         /* rows of the positional table held in the checkpoint (>= frames) */
    int32_t reserved[5];        /* [0] keep_taps  [1] branch streams  [2] fold_affine: fold the Dropout-only Linear chains
                                   (post_projector :528-536, emotion_proj / semantic_proj :488-496,509-517, post_header :360-364,
                                   audio fc1 -> fc2 :128-130) into one product each at pack time -- exact algebra in eval mode,
                                   different rounding, fewer FLOPs than the reference graph: OFF for parity runs
                                   [3] 1 = keep the SE tail of identity blocks as a separate pass (default 0: gate from conv1's output
                                   moments + relu(y*gate + x) in conv2's epilogue, eg_se_gate_pre / eg_conv3x3_se)
                                   [4] 1 = the caller keeps several batches in flight on this GPU (ClipPipeline lanes): the pre-split products
                                   take the 128 x 128 tile from 64 workgroups up (less CU time, more latency); 0 = tile for stand-alone latency */
} EgGeneratorConfig;

typedef struct EgGenerator EgGenerator;

int eg_generator_default_config(EgGeneratorConfig* cfg);          /* TED: 34/126/4, spec 128x124 */
int eg_generator_create(const EgGeneratorConfig* cfg, EgGenerator** out);
void eg_generator_destroy(EgGenerator* g);
int64_t eg_generator_arena_floats(const EgGenerator* g);
int32_t eg_generator_num_weights(const EgGenerator* g);
int eg_generator_weight_entry(const EgGenerator* g, int32_t index, EgWeightEntry* out);
int64_t eg_generator_workspace_bytes(const EgGenerator* g, int32_t batch);

/* Transformer.forward (Models_spatial_memory.py:566-616 / Models_memory.py:521-565), eval mode.
 *   spec      [B, n_mels, spec_len] fp32 (dB)           text  [B, text_len] int64
 *   prior     [B, prior_frames, pose_dim]               sampled [B, frames, d_model] or NULL
 * outputs (any may be NULL to skip the copy-out; the computation is still performed):
 *   pose [B, frames, pose_dim]   emotion_feature, semantic_feature [B, frames, d_model]
 *   emotion_prediction [B, 8]    text_embedding [B, text_len, 512]                      */
int eg_generator_forward(const EgGenerator* g, const float* arena, int32_t batch,
                         const float* spec, const int64_t* text, const float* prior, const float* sampled,
                         float* pose, float* emotion_feature, float* semantic_feature,
                         float* emotion_prediction, float* text_embedding,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* Diversity sampling (BASELINE config 5): the audio/semantic tower is run once per clip, then
 * fusion -> encoder -> decoder -> post_projector for `draws` sampled emotion maps per clip.
 *   sampled [B, draws, frames, d_model]    pose [B, draws, frames, pose_dim]               */
int eg_generator_forward_draws(const EgGenerator* g, const float* arena, int32_t batch, int32_t draws,
                               const float* spec, const float* prior, const float* sampled, float* pose,
                               void* workspace, int64_t workspace_bytes, void* stream);
int64_t eg_generator_draws_workspace_bytes(const EgGenerator* g, int32_t batch, int32_t draws);

/* Intermediate taps of the most recent eg_generator_forward on this workspace (for parity tests):
 * returns the device pointer inside `workspace` and the element count; names: "stem", "layer1", "layer2", "layer3"
 * (NHWC, only when the generator was created with keep_taps), "audio_map", "audio_feat", "prior_enc", "fusion", "enc_out",
 * "dec_out", and the buffers the attention blocks reuse -- contents of the LAST block that ran: "attn_q", "attn_qkv",
 * "attn_out", "mha_out", "proj". */
int eg_generator_tap(const EgGenerator* g, int32_t batch, void* workspace, const char* name,
                     float** d_ptr, int64_t* numel);

/* ------------------------------------------------------------------------------------------
 * Emotion CVAE = MLP_Reconstruct_v3 (CAVE/BEAT_CVAE.py:312-460)
 * ------------------------------------------------------------------------------------------ */
typedef struct EgCvaeConfig {
    int32_t frames;             /* decoder output channels; 60 hard-coded upstream (:365-368) */
    int32_t d_model;            /* 512 = 4 * latent map width 128 (:445) */
    int32_t latent;             /* 32 */
    int32_t n_classes;          /* 8 */
    int32_t reserved[4];
} EgCvaeConfig;
typedef struct EgCvae EgCvae;

int eg_cvae_default_config(EgCvaeConfig* cfg);
int eg_cvae_create(const EgCvaeConfig* cfg, EgCvae** out);
void eg_cvae_destroy(EgCvae* c);
int64_t eg_cvae_arena_floats(const EgCvae* c);
int32_t eg_cvae_num_weights(const EgCvae* c);
int eg_cvae_weight_entry(const EgCvae* c, int32_t index, EgWeightEntry* out);
int64_t eg_cvae_workspace_bytes(const EgCvae* c, int32_t n);

/* MLP_Reconstruct_v3.sample (:427-447).  y [n, 8] one-hot, z [n, 32] latent draw (the host draws it
 * with torch.randn on the CPU generator exactly as :441 does) -> out [n, frames, d_model]. */
int eg_cvae_sample(const EgCvae* c, const float* arena, int32_t n, const float* y, const float* z,
                   float* out, void* workspace, int64_t workspace_bytes, void* stream);
/* MLP_Reconstruct_v3.forward (:403-424), eval-mode BN.  x [n, frames, d_model], y [n,8], eps [n,32]
 * (reparameterize :389-399: z = eps*exp(0.5*logvar)+mu) -> recon [n, frames, d_model], mu, logvar [n,32]. */
int eg_cvae_forward(const EgCvae* c, const float* arena, int32_t n, const float* x, const float* y,
                    const float* eps, float* recon, float* mu, float* logvar,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Mel front-end = extract_melspectrogram (utils/train_utils_BEAT.py:186-190) + the loader's
 * column slice (data_loader/lmdb_loader_BEAT_full.py:229)
 * ------------------------------------------------------------------------------------------ */
/* audio [B, n_samples] fp32 16 kHz -> spec [B, 128, out_frames] fp32 holding fp16-rounded dB.
 * n_fft 1024, hop 512, centred (zero pad), periodic Hann, 128 Slaney mels, power_to_db(ref=max, top_db=80).
 * d_melfb_t [513,128] (transposed filterbank), d_window [1024], d_twiddle [512,2], d_band [128,2] are caller-provided
 * device copies of the tables eg_mel_tables fills on the host.
 * workspace >= eg_mel_workspace_bytes. */
int eg_mel_tables(float* h_melfb_t /*513*128, transposed*/, float* h_window /*1024*/, float* h_twiddle /*2*512*/,
                  int32_t* h_band /*128*2: non-zero bin range of each mel filter*/);
int64_t eg_mel_workspace_bytes(int32_t batch, int32_t n_samples);
int eg_melspectrogram(const float* audio, int32_t batch, int32_t n_samples, const float* d_melfb_t,
                      const float* d_window, const float* d_twiddle, const int32_t* d_band, float* spec, int32_t out_frames,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Block-level operators (the reference's L2 blocks), used by the module-level mirrors and by the
 * per-kernel parity tests.  Weights here are passed as individual device pointers in the PACKED
 * layouts named above.
 * ------------------------------------------------------------------------------------------ */

/* nn.Conv2d 3x3 pad 1 (Full_model/ResNetBlocks.py:12,14; ResNetSE34V2.py:21) on NHWC fp32 with the
 * fused epilogue  v = acc + bias[c]; if relu: v = max(v,0); v = v*scale[c] + shift[c].
 *   x [B,H,W,Cin]   w EG_PACK_CONV3X3   bias/scale/shift [Cout_pad] (NULL = 0/1/0)
 *   y NHWC [B,Ho,Wo,Cout] or, if nchw_out, [B,Cout,Ho*Wo]
 *   gap_partial (optional) [B, eg_conv3x3_gap_tiles(...), Cout]: per-tile channel sums of y (for SE). */
int eg_conv3x3(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
               float* y, float* gap_partial, int32_t batch, int32_t h, int32_t wdt, int32_t cin, int32_t cout,
               int32_t stride, int32_t relu, int32_t nchw_out, int32_t precision, void* stream);
int32_t eg_conv3x3_gap_tiles(int32_t h, int32_t wdt, int32_t cin, int32_t cout, int32_t stride);
/* eg_conv3x3 with the SEBasicBlock tail of an identity block fused into the epilogue (ResNetBlocks.py:28-36):
 *   y = relu(BN(conv(x)) * gate[b, co] + residual[pixel, co]),   gate [B, Cout] from eg_se_gate_pre, residual NHWC like y.
 * gate == residual == NULL: plain eg_conv3x3.  gate == NULL with a residual: y = BN(conv(x)) + residual, no ReLU -- the fused fan-in add of
 * the training path (an input gradient landing on a tensor that has a second consumer). */
int eg_conv3x3_se(const float* x, const float* w_packed, const float* bias, const float* scale, const float* shift, const float* gate,
                  const float* residual, float* y, float* gap_partial, int32_t batch, int32_t h, int32_t wdt, int32_t cin, int32_t cout,
                  int32_t stride, int32_t relu, int32_t nchw_out, int32_t precision, void* stream);
/* y = conv(x) + (bit ? residual : 0), stride 1, NHWC: the input gradient of an SEBasicBlock's first convolution with the identity shortcut's
 * gradient dout * [out > 0] (ResNetBlocks.py:33-36 under autograd) added in the epilogue from `dout` and the tail's ReLU bit mask (eg_se_tail_forward's
 * relu_bits: bit e & 31 of word e >> 5 for element e) -- the masked map is never stored (eg_se_tail_backward_apply with dres == NULL).
 * x = the upstream gradient, w_packed = the flipped filter image (eg_pack_conv3x3_device with flip_transpose). */
int eg_conv3x3_res_masked(const float* x, const float* w_packed, const float* residual, const uint32_t* res_bits, float* y, int32_t batch, int32_t h,
                          int32_t wdt, int32_t cin, int32_t cout, int32_t precision, void* stream);
/* Training forward of a tower convolution (nn.Conv2d -> optional ReLU, ResNetBlocks.py:24-27 under autograd) in the split-bf16 modes: y as
 * eg_conv3x3 plus BOTH per-(clip, tile) channel partials, sums of y and of y*y ([batch][eg_conv3x3_gap_tiles][cout] each), so that the train-mode
 * BatchNorm that follows takes mean and variance from them (eg_bn_train_forward_sq) without reading y again. */
int eg_conv3x3_sq(const float* x, const float* w_packed, const float* bias, float* y, float* gap_partial, float* gap_sq_partial, int32_t batch,
                  int32_t h, int32_t wdt, int32_t cin, int32_t cout, int32_t stride, int32_t relu, int32_t precision, void* stream);
/* The same convolution on x' = x * in_scale[ci] + in_shift[ci] (per INPUT channel; in-image pixels only, the zero padding stays zero): the train-mode
 * BatchNorm in front of the convolution (ResNetBlocks.py:26-27, bn1 -> conv2) folded into the operand staging -- the normalised map is never written.
 * in_scale / in_shift come from eg_bn_train_stats_sq.  Split-bf16 modes, cin % 32 == 0.  gap_partial / gap_sq_partial: both or neither. */
int eg_conv3x3_sq_in_affine(const float* x, const float* in_scale, const float* in_shift, const float* w_packed, const float* bias, float* y,
                            float* gap_partial, float* gap_sq_partial, int32_t batch, int32_t h, int32_t wdt, int32_t cin, int32_t cout, int32_t stride,
                            int32_t relu, int32_t precision, void* stream);
/* SELayer gate of a block computed BEFORE its conv2 runs: the spatial mean of BN2(conv2(t1)) is linear in window sums of t1
 * (total from conv1's gap partials, border lines / corners read from t1), so gate = sigmoid(W2 relu(W1 mean + b1) + b2) needs
 * only t1 and conv2's fp32 weight image (the head of its EG_PACK_CONV3X3 entry).  t1 NHWC [B,H,W,C], conv2: C -> C, stride 1. */
int eg_se_gate_pre(const float* t1, const float* gap_partial, int32_t tiles, const float* conv2_w_packed, const float* scale2,
                   const float* shift2, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                   int32_t batch, int32_t h, int32_t wdt, int32_t c, void* stream);
/* size in floats of one EG_PACK_CONV3X3 image (fp32 image + bf16 hi/lo images) */
int64_t eg_conv3x3_packed_floats(int32_t cin, int32_t cout_pad);

/* Stem: Conv2d(1->C,3x3,bias) -> ReLU -> BN (ResNetSE34V2.py:64-66).  x [B,H,W], y NHWC [B,H,W,C]. */
int eg_stem_conv(const float* x, const float* w9xc, const float* bias, const float* scale, const float* shift,
                 float* y, int32_t batch, int32_t h, int32_t wdt, int32_t c, void* stream);

/* SELayer gate (ResNetBlocks.py:92-96): s[b,c] = sigmoid(W2 relu(W1 mean_hw(y) + b1) + b2) from the
 * per-tile sums written by eg_conv3x3.  w1 [C/8,C], w2 [C,C/8] raw nn.Linear layouts. */
int eg_se_gate(const float* gap_partial, int32_t tiles, const float* w1, const float* b1, const float* w2,
               const float* b2, float* gate, int32_t batch, int32_t c, int32_t hw, void* stream);

/* SEBasicBlock tail (ResNetBlocks.py:28-36): out = relu(y*gate[b,c] + residual), residual = x_in or,
 * for the first block of a stage, BN(conv1x1_stride(x_in)) (ResNetSE34V2.py:43-47).
 *   y,out [B,Ho,Wo,C]; x_in [B,H,W,Cin]; ds_w EG_PACK_CONV1X1 or NULL. */
int eg_se_residual_relu(const float* y, const float* gate, const float* x_in, const float* ds_w,
                        const float* ds_scale, const float* ds_shift, float* out, int32_t batch, int32_t ho,
                        int32_t wo, int32_t c, int32_t h_in, int32_t w_in, int32_t cin, int32_t stride, void* stream);

/* nn.Linear family: Y[M,N] = epi(X[M,K] . W[N,K]^T).
 *   v = acc + bias[n] + res1[m,n];  if relu: v = max(v,0);  if res2: v = max(v + res2[m,n], 0)
 * lda/ldw/ldc/ldr in floats; K, lda, ldw multiples of 4 and X, W 16-byte aligned.
 * a_shift/a_seq implement the causal dilated tap of Full_model/tcn.py:18-24: source row of output row m is
 * m - a_shift, taken as zero when (m % a_seq) < a_shift (a_shift = 0 disables). */
int eg_linear(const float* x, int32_t lda, const float* w, int32_t ldw, const float* bias,
              const float* res1, const float* res2, int32_t ldr, float* y, int32_t ldc,
              int32_t m, int32_t n, int32_t k, int32_t relu, int32_t a_shift, int32_t a_seq,
              int32_t precision, void* stream);
/* Pre-split activations: eg_split_tiles converts fp32 X [M,K] into bf16 (hi, lo) tile-planar images
 * [ceil(M/64)][Kpad/8][64][8] (hi image then lo image; Kpad = K rounded up to 64; 4*ceil(M/64)*64*Kpad bytes);
 * eg_linear_presplit consumes them (same epilogue as eg_linear, bf16 modes only).  Used where one activation feeds several
 * products, so that it is split once instead of by every consuming workgroup. */
int eg_split_tiles(const float* x, int32_t lda, int32_t m, int32_t k, void* images, void* stream);
int eg_linear_presplit(const void* x_images, int32_t k_x, const float* w, int32_t ldw, const float* bias,
                       const float* res1, const float* res2, int32_t ldr, float* y, int32_t ldc,
                       int32_t m, int32_t n, int32_t k, int32_t relu, int32_t precision, void* stream);

/* Split-K variant for tall-K, short-M products (emotion_classifer_header.0: K = frames*d_model,
 * Models_spatial_memory.py:500).  partial >= splits*M*N floats. */
int eg_linear_splitk(const float* x, int32_t lda, const float* w, int32_t ldw, const float* bias, float* y,
                     int32_t ldc, int32_t m, int32_t n, int32_t k, int32_t relu, int32_t splits,
                     float* partial, int32_t precision, void* stream);

/* The extended product of the TRAINING path: eg_linear / eg_linear_splitk plus the two element masks a train()-mode transformer block needs
 * around a product, so that they cost no launch of their own (Full_model/SubLayers.py:54,79: `q = self.dropout(self.fc(q)); q += residual`,
 * `x = self.dropout(x); x += residual`; Models_spatial_memory.py:488-536: the Dropout(0.2) between the Linears of the projection MLPs):
 *   v = acc + bias[n]
 *   gate_src:  v = gate_src[m,n] > 0 ? v : 0      -- ReLU backward fused into the input-gradient product (gate_src = the forward's ReLU output)
 *   drop_p>0:  v = keep(seed, epoch, drop_offset + m*n_cols + n) ? v / (1 - drop_p) : 0   -- nn.Dropout on the product; the mask is the counter
 *              hash of eg_dropout_dev on the flat [M, N] index, so a backward pass re-draws it (here or with eg_dropout_dev) from the same scalars
 *   v += res1[m,n];  relu;  res2 as eg_linear.
 * splits >= 2: split-K (partial >= splits*M*N floats) with the same epilogue applied by the fixed-order fold.
 * precision f32: w is [N, ldw] fp32; bf16 modes: the EG_PACK_LINEAR image (eg_pack_linear_device), ldw % 64 == 0. */
typedef struct EgLinearArgs {
    const float* x; const float* w; const float* bias; const float* res1; const float* res2; float* y;
    const float* gate_src;          /* [M, ldg] or NULL */
    const int32_t* drop_epoch;      /* device-resident step counter or NULL (eg_dropout_dev) */
    float* partial;                 /* split-K scratch or NULL */
    const void* x_images;           /* NULL, or X as bf16 (hi, lo) tile-planar images of width k_x (eg_split_tiles layout; then x is ignored): the
                                       pre-split product of eg_linear_presplit with this epilogue */
    void* y_images;                 /* NULL, or a second output: Y as images of width y_k for the next product (written beside y) */
    uint64_t drop_offset;
    int32_t lda, ldw, ldr, ldc, ldg;
    int32_t m, n, k, relu, precision, splits;
    int32_t k_x, y_k;
    uint32_t drop_seed;
    float drop_p;
} EgLinearArgs;
int eg_linear_ex(const EgLinearArgs* args, void* stream);

/* nn.LayerNorm(D, eps) over the last axis (Full_model/SubLayers.py:55-57,80-82).  rows x D, D%4==0, D<=2048. */
int eg_layernorm(const float* x, const float* gamma, const float* beta, float* y, int32_t rows, int32_t d,
                 float eps, void* stream);
/* The same with a second output: y as bf16 (hi, lo) tile-planar images [ceil(rows/64)][d/8][64][8] (eg_split_tiles layout, d % 64 == 0) for the
 * pre-split product that consumes the row next (y_images may be NULL). */
int eg_layernorm_img(const float* x, const float* gamma, const float* beta, float* y, void* y_images, int32_t rows, int32_t d, float eps,
                     void* stream);

/* ScaledDotProductAttention (Full_model/Modules.py:13-23) for all heads, mask=None, eval mode:
 * out[b,i,h*dv:(h+1)*dv] = softmax_j((q[b,i,h]/sqrt(dk)) . k[b,j,h]) v[b,j,h].
 * q [B,Lq,H*dk] (row stride ldq), k/v [B,Lk,H*dk] (ldk/ldv), out [B,Lq,H*dk] (ldo).  dk == 64.
 * attn (optional) [B,H,Lq,Lk] receives the probabilities (the reference returns them).
 * Both products run on MFMA in the arithmetic mode `precision` (EG_PREC_*); pointers 16-byte aligned, Lk <= 256. */
int eg_attention(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv,
                 float* out, int32_t ldo, float* attn, int32_t batch, int32_t heads, int32_t lq, int32_t lk,
                 int32_t dk, int32_t precision, void* stream);
/* The same with the reference's mask argument (Modules.py:18-19: `attn = attn.masked_fill(mask == 0, -1e9)` before the softmax): mask bytes
 * [batch][1 or lq][lk], 0 = masked; mask_query_stride = 0 broadcasts one key row per clip over the queries, else it is the byte distance of
 * consecutive query rows (>= lk); mask_batch_stride the byte distance of consecutive clips.  The head axis is broadcast, as
 * MultiHeadAttention.forward does with `mask.unsqueeze(1)` (SubLayers.py:44-45).  The gesture path itself always passes mask = None
 * (Models_spatial_memory.py:574,611); this entry exists so that Encoder / Decoder / MultiHeadAttention keep their full signature. */
int eg_attention_masked(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, const uint8_t* mask,
                        int64_t mask_batch_stride, int32_t mask_query_stride, float* out, int32_t ldo, float* attn, int32_t batch,
                        int32_t heads, int32_t lq, int32_t lk, int32_t dk, int32_t precision, void* stream);

/* MultiHeadAttention.forward (Full_model/SubLayers.py:30-59): LN(fc(attn(q Wq, k Wk, v Wv)) + q).
 * xq [B*Lq, D], xkv [B*Lk, D]; wq/wk/wv packed nn.Linear [heads*64, D], wo packed [D, heads*64] (d_k = d_v = 64; D need not
 * equal heads*64: Motion_Discriminator runs D = 128 with 8 heads); out [B*Lq, D].  D % 4 == 0.
 * workspace >= eg_mha_workspace_bytes. */
int64_t eg_mha_workspace_bytes(int32_t batch, int32_t lq, int32_t lk, int32_t d_model, int32_t heads);
int eg_multi_head_attention(const float* xq, const float* xkv, const float* wq, const float* wk, const float* wv,
                            const float* wo, const float* ln_g, const float* ln_b, float* out, float* attn,
                            int32_t batch, int32_t lq, int32_t lk, int32_t d_model, int32_t heads,
                            int32_t precision, void* workspace, int64_t workspace_bytes, void* stream);

/* PositionwiseFeedForward.forward (Full_model/SubLayers.py:74-84): LN(w2 relu(w1 x + b1) + b2 + x). */
int64_t eg_ffn_workspace_bytes(int32_t rows, int32_t d_model, int32_t d_inner);
int eg_positionwise_ffn(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                        const float* ln_g, const float* ln_b, float* out, int32_t rows, int32_t d_model,
                        int32_t d_inner, int32_t precision, void* workspace, int64_t workspace_bytes, void* stream);

/* TemporalConvNet.forward (Full_model/tcn.py:63; TemporalBlock :43-47), channels-last:
 * x, y [B, L, C]; per level i (dilation 2^i) two weight-normed causal convs k=2 + ReLU, residual, ReLU.
 * w points at levels*2 convs, each {tap0 [C,Cpad], tap1 [C,Cpad], bias [C]} packed back to back
 * (EG_PACK_WN_TAP x2 + RAW); Cpad = C rounded up to 64.  workspace >= 4*B*L*Cpad floats. */
int eg_tcn_forward(const float* x, const float* w, float* y, int32_t batch, int32_t len, int32_t c,
                   int32_t levels, int32_t precision, void* workspace, int64_t workspace_bytes, void* stream);

/* out[r,:] = a[r,:] + table[r % period,:]  (PositionalEncoding.forward, Full_model/Models_spatial_memory.py:46-48);
 * period == 0: plain elementwise add of two [rows, d] tensors (fusion add, :601-605).  d % 4 == 0. */
int eg_add_rows(const float* a, const float* table, float* out, int64_t rows, int32_t d, int32_t period, void* stream);

/* VAE reparameterisation (CAVE/BEAT_CVAE.py:397-399): z = eps*exp(0.5*logvar) + mu, n elements. */
int eg_reparameterize(const float* mu, const float* logvar, const float* eps, float* z, int64_t n, void* stream);

/* nn.Conv1d over [n, cin, lin] -> [n, cout, lout], lout = (lin + 2*pad - k)/stride + 1, weight [cout, cin, k] (PyTorch layout):
 * y = bias + conv(x); act != 0: LeakyReLU(0.2), then (scale != NULL) y*scale[co] + shift[co]  -- the CVAE's conv -> LeakyReLU -> BN
 * order (CAVE/BEAT_CVAE.py:318-332); MotionAE's conv -> BN -> LeakyReLU (model/motion_ae.py:8-31) folds its BN into w / bias.
 * The input tile and the weights of one workgroup must fit 160 KB of LDS (else EG_ERR_UNSUPPORTED). */
int eg_conv1d(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* y, int32_t n,
              int32_t cin, int32_t cout, int32_t lin, int32_t k, int32_t stride, int32_t pad, int32_t act, void* stream);

/* SoftmaxContrastiveLoss (test_emotion_gesture_diversity_iterative.py:80-127), forward and evaluate in one call.
 * face, audio: [n, d] fp32.  Rows are L2-normalised (x / max(|x|, 1e-12)), cross[i][j] = max(1 / (|face_i - audio_j| + 1e-8), 1e-8),
 * loss = mean_i( logsumexp_j cross[i][j] - cross[i][i] )  (= F.cross_entropy(cross, arange(n))),  acc = mean_i( argmax_j cross[i][j] == i ).
 * cross ([n, n]) may be NULL; loss and acc are single floats on the device.  ws: eg_contrastive_workspace_bytes(n).
 * Deterministic: per-row results are reduced in a fixed order (no atomics).  1 <= n <= 4096, 1 <= d. */
int64_t eg_contrastive_workspace_bytes(int32_t n);
int eg_contrastive_loss(const float* face, const float* audio, int32_t n, int32_t d, float* cross, float* loss, float* acc,
                        void* workspace, int64_t workspace_bytes, void* stream);
/* Gradients of that loss (mode 'max') with respect to both feature sets, for an upstream gradient of 1: softmax of every row of cross_dist,
 * back through 1/(D + 1e-8), the pairwise distances and F.normalize.  workspace >= eg_contrastive_backward_workspace_bytes(n). */
int64_t eg_contrastive_backward_workspace_bytes(int32_t n);
int eg_contrastive_loss_backward(const float* face, const float* audio, int32_t n, int32_t d, float* gface, float* gaudio, void* workspace,
                                 int64_t workspace_bytes, void* stream);


/* ===================== training-path primitives (fp32, deterministic reductions) =====================
 * The reference trains through autograd over ATen (its one loop: train_audio_classifier_K_fold.py:155-175; generator
 * hints: test_emotion_gesture_diversity_iterative.py:64-127,355-366).  These are the backward-side kernels; the host
 * (emotiongestures_amd/train/) sequences them.  All buffers caller-owned, row-major, fp32. */

/* y[c*ldy + r] = x[r*ldx + c] */
int eg_transpose(const float* x, int32_t ldx, int32_t rows, int32_t cols, float* y, int32_t ldy, void* stream);
/* C[m,n] (+)= sum_k A[k,m] B[k,n] on v_mfma_f32_16x16x4_f32 (dW = dY^T X; conv wgrad over im2col rows); split-K through
 * `workspace` (eg_gemm_tn_workspace_floats; at least m*n floats when accumulate != 0). */
int64_t eg_gemm_tn_workspace_floats(int32_t m, int32_t n, int64_t k);
int eg_gemm_tn(const float* a, int32_t lda, const float* b, int32_t ldb, float* c, int32_t ldc, int32_t m, int32_t n, int64_t k,
               float* workspace, int64_t workspace_floats, int32_t accumulate, void* stream);
/* weight gradient of a 3x3 / pad 1 / stride s convolution as an implicit TN GEMM over the output pixels (no im2col buffer):
 * dw_mat [Cout][(kh*3+kw)*Cin + ci]; Cin % 4 == 0; workspace >= eg_gemm_tn_workspace_floats(cout, 9*cin, B*Ho*Wo) */
int eg_conv3x3_wgrad(const float* x, const float* dy, float* dw_mat, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                     int32_t stride, float* workspace, int64_t workspace_floats, void* stream);
/* 3x3 / pad 1 / stride s, NHWC: forward  col[(b,oy,ox)][tap*C + c] = x[b, oy*s+kh-1, ox*s+kw-1, c];
 * backward (x = dcol, col = dx [B,H,W,C]): the transpose in gather form (F.conv2d's input gradient after the GEMM). */
int eg_im2col3x3(const float* x, float* col, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t stride, int32_t backward, void* stream);
/* pixel subsample of a 1x1 stride-s conv (ResNetSE34V2.py:43-47) and its transpose */
int eg_subsample(const float* x, float* y, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t stride, int32_t backward, void* stream);
/* 1-D channels-last im2col: x [B,L,C] -> col [B*Lout, k*C] at l*stride + j*dilation - pad_left; backward = transpose */
int eg_im2col1d(const float* x, float* col, int32_t batch, int32_t len, int32_t c, int32_t k, int32_t stride, int32_t pad_left,
                int32_t dilation, int32_t lout, int32_t backward, void* stream);
/* The small 1-D convolutions of the training step (emotion CVAE Conv1d / ConvTranspose1d stacks, CAVE/BEAT_CVAE.py:318-332,355-369; the prior
 * encoder's pred_conv, Full_model/Models_spatial_memory.py:224-231) on channels-last activations, one fp32 launch per product, fixed-order sums:
 *   forward          y[b, lo, co]  = bias[co] + sum_{ci, j} x[b, lo*stride - pad + j*dilation, ci] * w[co][ci][j]     (w: nn.Conv1d's [cout][cin][k])
 *   backward_input   dx[b, li, ci] = bias[ci] + sum_{co, j : li + pad - j*dilation = lo*stride} dy[b, lo, co] * w[co][ci][j]
 *   backward_weight  dw[co][ci][j] = sum_{b, lo} dy[b, lo, co] * x[b, lo*stride - pad + j*dilation, ci];  db_dy[co] = sum dy;  db_x[ci] = sum x
 * x / dx: [batch, len, cin], y / dy: [batch, len_out, cout]; bias, db_dy, db_x may be NULL; 1 <= k <= 8.  nn.ConvTranspose1d (weight
 * [cin][cout][k]) is the adjoint: forward = backward_input with the layer's bias, input gradient = forward, weight gradient = backward_weight
 * with x and dy exchanged (db_x = the layer's bias gradient). */
int eg_conv1d_cl_forward(const float* x, const float* w, const float* bias, float* y, int32_t batch, int32_t len, int32_t cin, int32_t len_out,
                         int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t dilation, void* stream);
int eg_conv1d_cl_backward_input(const float* dy, const float* w, const float* bias, float* dx, int32_t batch, int32_t len, int32_t cin,
                                int32_t len_out, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t dilation, void* stream);
/* workspace: eg_conv1d_cl_backward_weight_workspace_floats(...) floats (0 for small shapes: one launch); per-workgroup partials, folded in a fixed
 * order by a second launch.  db_x != NULL always takes the one-launch kernel. */
int64_t eg_conv1d_cl_backward_weight_workspace_floats(int32_t batch, int32_t cin, int32_t len_out, int32_t cout, int32_t k, int32_t stride,
                                                      int32_t dilation);
int eg_conv1d_cl_backward_weight(const float* x, const float* dy, float* dw, float* db_dy, float* db_x, int32_t batch, int32_t len, int32_t cin,
                                 int32_t len_out, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t dilation, float* workspace,
                                 int64_t workspace_floats, void* stream);
/* [rows, k] -> [rows, k_padded] zero padded (rows of the GEMM operands are read in 16-byte pieces) */
int eg_pad_cols(const float* x, float* y, int64_t rows, int32_t k, int32_t k_padded, void* stream);
/* nn.BatchNorm{1,2}d in train() mode over channels-last rows [rows, C]: batch statistics (biased variance for the
 * normalisation, unbiased for running_var, momentum as torch), saved mean / rstd for the backward. */
int64_t eg_colreduce_workspace_floats(int32_t c);
int eg_bn_train_forward(const float* x, const float* gamma, const float* beta, float* y, float* save_mean, float* save_rstd,
                        float* running_mean, float* running_var, int64_t rows, int32_t c, float momentum, float eps, float* workspace,
                        void* stream);
/* relu_mask != 0: x is the output of a ReLU (conv1 -> ReLU -> bn1, ResNetBlocks.py:24-26) and dx is that ReLU's input gradient, dx * [x > 0]. */
int eg_bn_train_backward(const float* x, const float* dy, const float* gamma, const float* save_mean, const float* save_rstd, float* dx,
                         float* dgamma, float* dbeta, int64_t rows, int32_t c, int32_t relu_mask, float* workspace, void* stream);
/* BatchNorm (train mode) on a map that eg_conv3x3 just wrote together with gap_partial [batch][tiles][c] (per-tile channel sums): the mean
 * and clip_sum [batch][c] (nullable) come from the partials, one centred pass over x gives the variance.  y NULL: statistics only. */
int eg_bn_train_forward_gap(const float* x, const float* gap_partial, int32_t tiles, int32_t batch, const float* gamma, const float* beta,
                            float* y, float* save_mean, float* save_rstd, float* clip_sum, float* running_mean, float* running_var,
                            int64_t rows, int32_t c, float momentum, float eps, float* workspace, void* stream);
/* The same with the variance also from partials (eg_conv3x3_sq's sums of squares): two small launches, no pass over x; the per-tile partials are
 * fp32, everything across tiles / clips and E[x^2] - mean^2 is double (relative error of the variance ~1e-6 (1 + mean^2 / var): used in the
 * split-bf16 modes only).  x may be NULL when y is NULL. */
int eg_bn_train_forward_sq(const float* x, const float* gap_partial, const float* gap_sq_partial, int32_t tiles, int32_t batch, const float* gamma,
                           const float* beta, float* y, float* save_mean, float* save_rstd, float* clip_sum, float* running_mean, float* running_var,
                           int64_t rows, int32_t c, float momentum, float eps, float* workspace, void* stream);
/* Statistics only (no apply), plus the apply folded to one affine per channel: aff_scale = gamma * rstd, aff_shift = beta - mean * aff_scale -- for
 * eg_conv3x3_sq_in_affine / eg_conv3x3_wgrad_mfma_oihw_in_affine, which apply it while staging their operand. */
int eg_bn_train_stats_sq(const float* gap_partial, const float* gap_sq_partial, int32_t tiles, int32_t batch, const float* gamma, const float* beta,
                         float* save_mean, float* save_rstd, float* running_mean, float* running_var, float* aff_scale, float* aff_shift, int64_t rows,
                         int32_t c, float momentum, float eps, float* workspace, void* stream);
/* SEBasicBlock tail under autograd (ResNetBlocks.py:28-36,92-96), maps [batch, hw, c] channels-last, c % 8 == 0, c <= 256:
 *   forward:  pooled = mean_hw(bn2(c2)) from clip_sum; h = relu(W1 pooled + b1); gate = sigmoid(W2 h + b2)      (eg_se_gate_train_forward)
 *             out = relu(bn2(c2) * gate + res) in one pass, bn2's output never stored                            (eg_se_tail_forward)
 *   backward: one masked reduction pass (s1, s2raw per clip), the gate's backward per clip (dz2, dz1, dgap_hw, u1, u2), the sums over
 *             clips (bn2 / SE parameter gradients, m1, m2), and one apply pass writing dc2 and dres. */
int eg_se_gate_train_forward(const float* clip_sum, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w1,
                             const float* b1, const float* w2, const float* b2, float* pooled, float* h, float* gate, int32_t batch, int32_t hw,
                             int32_t c, void* stream);
/* relu_bits (optional, batch * hw * c / 32 words; needs batch * hw * c % 32 == 0): the tail's ReLU mask [out > 0] as one nibble per float4 of the
 * map, eight float4 indices per word.  The two backward passes take it INSTEAD of `out` (then `out` may be NULL): they read 1/32 of a map where
 * they read a whole one (3.4 GB of the 128-clip step's 78).  eg_se_tail_backward_apply: dres may be NULL when relu_bits is given -- the consumer of
 * the shortcut's gradient then masks `dout` itself (eg_conv3x3_res_masked). */
int eg_se_tail_forward(const float* c2, const float* res, const float* mean, const float* rstd, const float* gamma, const float* beta,
                       const float* gate, float* out, uint32_t* relu_bits, int32_t batch, int32_t hw, int32_t c, void* stream);
int eg_se_tail_backward_reduce(const float* dout, const float* out, const uint32_t* relu_bits, const float* c2, const float* mean, float* s1,
                               float* s2raw, int32_t batch, int32_t hw, int32_t c, float* workspace, void* stream);
int eg_se_gate_train_backward(const float* s1, const float* s2raw, const float* clip_sum, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, const float* gate, const float* h, const float* w1, const float* w2,
                              float* dz2, float* dz1, float* dgap_hw, float* u1, float* u2, int32_t batch, int32_t hw, int32_t c, void* stream);
int eg_se_tail_backward_finish(const float* u1, const float* u2, const float* dz2, const float* dz1, const float* h, const float* pooled,
                               float* dgamma, float* dbeta, float* m1, float* m2, float* dw1, float* db1, float* dw2, float* db2,
                               int32_t batch, int32_t hw, int32_t c, void* stream);
int eg_se_tail_backward_apply(const float* dout, const float* out, const uint32_t* relu_bits, const float* c2, const float* mean, const float* rstd,
                              const float* gamma, const float* gate, const float* dgap_hw, const float* m1, const float* m2, float* dc2, float* dres,
                              int32_t batch, int32_t hw, int32_t c, void* stream);
/* o0[c] = sum_r a[r,c]; o1[c] = sum_r a[r,c]*b[r,c] (b NULL: sum a^2).  bias / LayerNorm affine gradients. */
int eg_colsum(const float* a, const float* b, float* o0, float* o1, int64_t rows, int32_t c, float* workspace, void* stream);
/* op: 0 relu(a) | 1 a*(b>0) | 2 leaky(a; s) | 3 a*(b>0 ? 1 : s) | 4 a+b | 5 a*s | 6 sigmoid(a) | 7 a*b*(1-b) | 8 a*b | 9 a+s*b | 10 exp(s*a) | 11 a*b[0] */
int eg_elementwise(const float* a, const float* b, float* y, int64_t n, int32_t op, float s, void* stream);
/* The same weight gradient for stride 1 on the split-bf16 matrix pipe (x and dy split to hi/lo bf16 while staged, 3 MFMA terms, fp32
 * accumulation, fixed-order partial sums): cin % 32 == 0, cout % 32 == 0.  workspace >= eg_conv3x3_wgrad_mfma_workspace_floats(...). */
int64_t eg_conv3x3_wgrad_mfma_workspace_floats(int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout);
int eg_conv3x3_wgrad_mfma(const float* x, const float* dy, float* dw_mat, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                          float* workspace, int64_t workspace_floats, void* stream);
/* The same gradient written in nn.Conv2d's weight layout dw[cout][cin][3][3] by the final fixed-order reduction (no permute pass behind it). */
int eg_conv3x3_wgrad_mfma_oihw(const float* x, const float* dy, float* dw, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                               float* workspace, int64_t workspace_floats, void* stream);
/* ... of a convolution whose input was x' = x * in_scale[ci] + in_shift[ci] (eg_conv3x3_sq_in_affine): the affine is re-applied while x is staged. */
int eg_conv3x3_wgrad_mfma_oihw_in_affine(const float* x, const float* in_scale, const float* in_shift, const float* dy, float* dw, int32_t batch, int32_t h,
                                         int32_t w, int32_t cin, int32_t cout, float* workspace, int64_t workspace_floats, void* stream);
/* Input gradient of nn.Conv2d(cin -> cout, k = 3, pad = 1, stride = 2) -- the `_make_layer` entry convolutions, Full_model/ResNetSE34V2.py:40-55 (F.conv2d's
 * dgrad under autograd) -- on the split-bf16 matrix pipe, phase-decomposed: each dx pixel (2i + py, 2j + px) only receives the taps whose parity
 * matches (9 tap products per four pixels).  dy [batch][ho][wo][cout] NHWC with ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1; w_flip = the packed
 * images of the rotated, transposed filter (eg_pack_conv3x3 with flip = 1: what the stride-1 input gradients read); dx [batch][h][w][cin], every
 * element written.  res_q (optional, [batch][ho][wo][cin]): a gradient that belongs to the pixels (2i, 2j) of dx -- the input gradient of the
 * stride-2 1x1 shortcut (:43-47), which reads exactly those pixels -- added in the epilogue instead of being scattered into a zero map first.
 * precision: EG_PREC_BF16X3 only; (cin, cout) in {(32, 64), (64, 128), (128, 256)}; else EG_ERR_UNSUPPORTED. */
int eg_conv3x3_dgrad_s2(const float* dy, const float* w_flip, const float* res_q, float* dx, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                        int32_t precision, void* stream);
/* Weight and bias gradient of nn.Linear on the split-bf16 matrix pipe (3 x v_mfma_f32_16x16x32_bf16 per product, fp32 accumulate):
 *   dw[n][k] = sum_r dy[r][n] * x[r][k]   (dy [rows, n] at pitch ldy, x [rows, k] at pitch ldx, dw at pitch lddw);  db[n] = sum_r dy[r][n] (db may be NULL).
 * F.linear's parameter gradients under autograd (every nn.Linear of Full_model/Models_spatial_memory.py, SubLayers.py:30-84).  Deterministic: rows
 * are split over workgroups only through fixed-order partials in `workspace` (eg_linear_wgrad_mfma_workspace_floats floats; 0 = not needed). */
int64_t eg_linear_wgrad_mfma_workspace_floats(int32_t rows, int32_t n, int32_t k);
int eg_linear_wgrad_mfma(const float* dy, int32_t ldy, const float* x, int32_t ldx, float* dw, int32_t lddw, float* db, int32_t rows, int32_t n, int32_t k,
                         float* workspace, int64_t workspace_floats, void* stream);
/* Weight gradient of nn.Conv2d(cin -> cout, k = 3, pad = 1, stride 1 | 2) on the split-bf16 matrix pipe as dW = dY^T im2col(x), the im2col matrix gathered
 * from the NHWC map while the operand is staged (no column buffer): the stride-2 stage-entry convolutions, Full_model/ResNetSE34V2.py:40-55 (the stride-1
 * body convolutions keep eg_conv3x3_wgrad_mfma).  dy [batch][ho][wo][cout], dw_mat [cout][9 * cin] ((kh, kw, ci) fastest to slowest, as eg_conv3x3_wgrad).
 * cin % 4 == 0.  workspace >= eg_linear_wgrad_mfma_workspace_floats(batch * ho * wo, cout, 9 * cin).  Deterministic (fixed-order partials). */
int eg_conv3x3_wgrad_gather_mfma(const float* x, const float* dy, float* dw_mat, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                 int32_t stride, float* workspace, int64_t workspace_floats, void* stream);
/* Device-side build of the weight image eg_conv3x3 reads (eg_conv3x3_packed_floats(cin', round_up(cout',16)) floats: fp32 image, then the
 * bf16 hi / lo images), for weights that change every step.  flip_transpose = 0: conv weight [cout][cin][3][3] as in the state_dict
 * (cin' = cin, cout' = cout).  flip_transpose = 1: the filter of the input-gradient convolution, w'[ci][co][kh][kw] = w[co][ci][2-kh][2-kw]
 * (cin' = cout, cout' = cin) -- F.conv2d's dgrad for stride 1 is eg_conv3x3 of dy with that image. */
int eg_pack_conv3x3_device(const float* w_oihw, int32_t cout, int32_t cin, int32_t flip_transpose, float* image, void* stream);
/* Device-side build of the weight image eg_linear's split-bf16 modes read (EG_PACK_LINEAR layout; eg_linear_packed_floats(n, k) floats,
 * ldw = k rounded up to 64) from w [n][k] fp32 with row stride ld.  transpose != 0: the image of w^T (w is then [k][n]): dX = dY W. */
int64_t eg_linear_packed_floats(int32_t n, int32_t k);
int eg_pack_linear_device(const float* w, int32_t ld, int32_t n, int32_t k, int32_t transpose, float* image, void* stream);
/* Every weight image of a training step in one launch.  `table`: device array of `count` 40-byte entries
 *   { const float* src; float* image; int32_t kind, a, b, c, flag, first_block; }
 * kind 0 = eg_pack_linear_device(src, ld = c, n = a, k = b, transpose = flag & 1, image); kind 1 = eg_pack_conv3x3_device(src, cout = a, cin = b, flip = flag & 1, image);
 * flag bit 1 set: the fp32 head of that image is left unwritten (only the split-bf16 kernels may read it);
 * first_block = sum of eg_pack_table_blocks(kind, a, b, flag) over the preceding entries, total_blocks = the sum over all. */
int32_t eg_pack_table_blocks(int32_t kind, int32_t a, int32_t b, int32_t flag);
int eg_pack_table(const void* table, int32_t count, int32_t total_blocks, void* stream);
/* nn.Dropout in train() mode with a counter-based mask (nothing stored): keep(i) = hash(seed, offset + i) >= p, y = keep ? x/(1-p) : 0;
 * the backward pass is the same call on dy.  The mask stream is this library's own (not torch's RNG). */
int eg_dropout(const float* x, float* y, int64_t n, float p, uint32_t seed, uint64_t offset, void* stream);
/* The same with a device-resident step counter mixed into the seed (epoch_dev may be NULL = eg_dropout): a training step replayed from a
 * captured hipGraph freezes the host scalars (seed, offset), the counter (incremented once per step by eg_counter_add inside the graph)
 * still gives every replay a fresh mask.  Forward and backward of one step read the same counter value. */
int eg_dropout_dev(const float* x, float* y, int64_t n, float p, uint32_t seed, uint64_t offset, const int32_t* epoch_dev, void* stream);
/* SELayer pieces (ResNetBlocks.py:92-96) on x [B, HW, C]: pooled mean (x scale), per-(clip, channel) dot, gate scaling (+ add[b,c]);
 * workspace (eg_colreduce_workspace_floats(c) floats) selects the two-level reduction, NULL the one-block-per-clip kernel */
int eg_seg_mean(const float* x, float* out, int32_t batch, int32_t hw, int32_t c, float scale, float* workspace, void* stream);
int eg_seg_dot(const float* dy, const float* x, float* out, int32_t batch, int32_t hw, int32_t c, float* workspace, void* stream);
int eg_se_scale(const float* a, const float* gate, const float* add, float* y, int32_t batch, int32_t hw, int32_t c, void* stream);
/* LayerNorm backward (SubLayers.py:55-57,80-82): dx and xhat; the affine gradients are one column reduction,
 * (dbeta, dgamma) = (sum dy, sum dy*xhat) = eg_colsum(dy, xhat, dbeta, dgamma, ...) */
int eg_layernorm_backward(const float* x, const float* dy, const float* gamma, float* dx, float* xhat, int32_t rows, int32_t d, float eps,
                          void* stream);
/* LayerNorm backward for the fused transformer blocks of the training path: dx as eg_layernorm_backward, the affine gradients from the SAME pass
 * (per-workgroup partial column sums + one fixed-order fold: no xhat round trip, no separate column reduction), and -- drop_p > 0 -- a second
 * output dx_dropped = nn.Dropout's backward applied to dx (the gradient that continues into the Dropout'ed branch `dropout(fc(.))` while dx itself
 * goes to the residual; mask = eg_dropout_dev's on the flat [rows, d] index).  d % 64 == 0, d <= 1024.  workspace: eg_layernorm_backward_ex_workspace_floats. */
int64_t eg_layernorm_backward_ex_workspace_floats(int32_t rows, int32_t d);
int eg_layernorm_backward_ex(const float* x, const float* dy, const float* gamma, float* dx, float* dx_dropped, float* dgamma, float* dbeta,
                             int32_t rows, int32_t d, float eps, float drop_p, uint32_t drop_seed, uint64_t drop_offset, const int32_t* epoch_dev,
                             float* workspace, void* branch_images, void* stream);      /* branch_images (may be NULL): the gradient that enters the
                             Dropout'ed branch (dx_dropped, or dx when drop_p == 0) also as bf16 (hi, lo) images for a pre-split input-gradient product */
/* ScaledDotProductAttention backward (Modules.py:13-23) from the forward's probabilities; Lq, Lk <= 64-ish (LDS-resident) */
int eg_attention_backward(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, const float* attn,
                          const float* dout, int32_t ldo, float* dq, int32_t lddq, float* dk, int32_t lddk, float* dv, int32_t lddv,
                          int32_t batch, int32_t heads, int32_t lq, int32_t lk, int32_t dk_dim, void* stream);
/* Training pair on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32) with nn.Dropout(p) on the probabilities (Modules.py:21) from the counter-based
 * mask of eg_dropout (counter = offset + linear index of (clip, head, query, key); epoch_dev as in eg_dropout_dev, may be NULL).  The forward stores
 * the UNMASKED probabilities in `attn` [batch, heads, lq, lk]; the backward recomputes the mask from the same (p, seed, offset, epoch).  The backward
 * walks the queries in chunks with K / V resident in LDS: lk <= 128 (TED 34, BEAT 60, BEAT-long 120), any lq.  p = 0 is plain attention. */
int eg_attention_train(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, float* out, int32_t ldo, float* attn,
                       int32_t batch, int32_t heads, int32_t lq, int32_t lk, int32_t dk, float p, uint32_t seed, uint64_t offset,
                       const int32_t* epoch_dev, void* stream);
int eg_attention_backward_train(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, const float* attn,
                                const float* dout, int32_t ldo, float* dq, int32_t lddq, float* dk, int32_t lddk, float* dv, int32_t lddv,
                                int32_t batch, int32_t heads, int32_t lq, int32_t lk, int32_t dk_dim, float p, uint32_t seed, uint64_t offset,
                                const int32_t* epoch_dev, void* stream);
/* losses: scale * mean smooth-L1 (beta); scale * mean CE / focal(alpha[b] per sample or NULL, gamma >= 0)
 * (train_audio_classifier_K_fold.py:95-105: `alpha * (1-pt)**gamma * ce` broadcasts alpha over the batch axis) */
int eg_smooth_l1(const float* pred, const float* target, float* loss, float* dpred, int64_t n, float beta, float scale, float* workspace,
                 void* stream);
int eg_cross_entropy(const float* logits, const int64_t* labels, const float* alpha, float gamma, float scale, float* loss, float* dlogits,
                     int32_t batch, int32_t classes, float* workspace, void* stream);
/* VAE KL term of MLP_Reconstruct_v3's (mu, logvar) (CAVE/BEAT_CVAE.py:389-399,403-424): scale * mean_b(-0.5 sum_j(1 + lv - mu^2 - e^lv)) and its gradients */
int eg_kld(const float* mu, const float* logvar, float* loss, float* dmu, float* dlogvar, int32_t n, int32_t d, float scale, void* stream);
/* torch.optim.Adam step on a flat buffer (L2 weight decay added to the gradient; train_audio_classifier_K_fold.py:128) */
int eg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int32_t step, void* stream);
/* The same update with the (1-based) step count read from device memory, and the counter's increment, for steps replayed from a captured
 * hipGraph (host scalars are frozen at capture). */
int eg_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, const int32_t* step_dev, void* stream);
int eg_counter_add(int32_t* counter, int32_t delta, void* stream);
/* The memory nets of Full_model/Models_memory.py's Prior_MemoryEncoder under autograd (the MLPs around them are ordinary Linear operators).
 * SP_Memory_Net_v1.forward (:233-251): for the first `chunk` of the `frames` predicted frames  s = sigmoid(<mem_b, pred_bc>),
 * out_bc = s pred_bc + (1 - s) mem_b; later frames pass through (mem [batch, dim], pred / out [batch, frames, dim], gate [batch, chunk] saved
 * for the backward, which returns dpred and dmem).
 * TM_Memory_Net.forward (:288-292) behind its batch-coupled score [batch, chunk] = mem (mem^T pe):  w = softmax(score, dim=1),
 * out_bc = pred_bc (1 + w_bc) for c < chunk; the backward returns dpred and dscore. */
int eg_sp_gate_forward(const float* mem, const float* pred, float* out, float* gate, int32_t batch, int32_t frames, int32_t dim, int32_t chunk, void* stream);
int eg_sp_gate_backward(const float* mem, const float* pred, const float* gate, const float* dout, float* dpred, float* dmem, int32_t batch,
                        int32_t frames, int32_t dim, int32_t chunk, void* stream);
int eg_tm_scale_forward(const float* score, const float* pred, float* out, float* w, int32_t batch, int32_t frames, int32_t dim, int32_t chunk, void* stream);
int eg_tm_scale_backward(const float* w, const float* pred, const float* dout, float* dpred, float* dscore, int32_t batch, int32_t frames, int32_t dim,
                         int32_t chunk, void* stream);
/* Gradient-bucket payload conversion for the data-parallel all-reduce (bf16 buckets, 79 MB instead of 158 MB per step over
 * xGMI): fp32 -> bf16 round-to-nearest-even, and bf16 -> fp32 times `scale`. */
int eg_f32_to_bf16(const float* x, uint16_t* y, int64_t n, void* stream);
int eg_bf16_to_f32(const uint16_t* x, float* y, int64_t n, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EMOGEST_H */
