/*
 * ttsx.h - C ABI of the MI355X (gfx950) synthesis kernels.
 *
 * The reference (a Mozilla-TTS fork) has no FFI: its boundary for this path
 * is a Python API.  The Python classes in tts_amd/ mirror that API and call
 * the entry points below through ctypes; each entry point cites the
 * reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - every call returns 0 (TTSX_OK) or a negative error code and never
 *     throws; ttsx_last_error_string() describes the last failure on the
 *     calling thread;
 *   - handles are opaque, created and destroyed explicitly; one handle may be
 *     used by one thread at a time;
 *   - all work is enqueued on the hipStream_t passed in (a void* here so that
 *     the header needs no HIP include); calls that return host-visible
 *     results synchronise that stream before returning and say so;
 *   - fp32 everywhere; weights are handed over in the reference's own
 *     state_dict layouts and packed once at create time.
 */
#ifndef TTSX_H
#define TTSX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTSX_OK 0
#define TTSX_ERR_ARG (-1)      /* invalid argument (shape, null pointer, unsupported size) */
#define TTSX_ERR_HIP (-2)      /* a HIP runtime call failed */
#define TTSX_ERR_STATE (-3)    /* call sequence error (e.g. step before begin) */

#define TTSX_NORM_SOFTMAX 0
#define TTSX_NORM_SIGMOID 1

#define TTSX_PAD_CONSTANT 0
#define TTSX_PAD_REFLECT 1

const char* ttsx_last_error_string(void);
int ttsx_version(void);

/* ------------------------------------------------------------------ Tacotron2 decoder
 * Replaces Decoder.inference / Decoder.decode and everything they call:
 * Tacotron2/TTS/models.py:643-673 (loop), 578-617 (step), 76-82 (Prenet),
 * 314-361 + 257-267 + 84-104 (OriginalAttention with location layer),
 * 251-252 (inputs_layer pre-projection).  Tensor names are the reference
 * state_dict keys under "decoder.".
 */
typedef struct {
  const float* prenet_w[2];       /* prenet.layers.{0,1}.linear_layer.weight  [256,80], [256,256] */
  const float* prenet_bn_weight[2]; /* prenet.layers.i.bn.{weight,bias,running_mean,running_var} [256]; */
  const float* prenet_bn_bias[2];   /*   all four NULL for prenet_type "original"                       */
  const float* prenet_bn_mean[2];
  const float* prenet_bn_var[2];
  const float* attn_rnn_w_ih;     /* attention_rnn.weight_ih [4096,768]  (input = [prenet 256; context 512]) */
  const float* attn_rnn_w_hh;     /* attention_rnn.weight_hh [4096,1024] */
  const float* attn_rnn_b_ih;     /* [4096] */
  const float* attn_rnn_b_hh;     /* [4096] */
  const float* query_w;           /* attention.query_layer.linear_layer.weight  [128,1024] */
  const float* inputs_w;          /* attention.inputs_layer.linear_layer.weight [128,512]  */
  const float* v_w;               /* attention.v.linear_layer.weight [1,128] */
  const float* v_b;               /* attention.v.linear_layer.bias   [1]     */
  const float* loc_conv_w;        /* attention.location_layer.location_conv.weight [32,2,31] */
  const float* loc_dense_w;       /* attention.location_layer.location_dense.linear_layer.weight [128,32] */
  const float* dec_rnn_w_ih;      /* decoder_rnn.weight_ih [4096,1536] (input = [query 1024; context 512]) */
  const float* dec_rnn_w_hh;      /* decoder_rnn.weight_hh [4096,1024] */
  const float* dec_rnn_b_ih;
  const float* dec_rnn_b_hh;
  const float* proj_w;            /* linear_projection.linear_layer.weight [80*r_init,1536] */
  const float* proj_b;            /* [80*r_init] */
  const float* stop_w;            /* stopnet.1.linear_layer.weight [1,1024+80*r_init] */
  const float* stop_b;            /* [1] */
} ttsx_t2_decoder_weights;

typedef struct ttsx_t2_decoder ttsx_t2_decoder;

/* Packs the weights (BatchNorm folded, gate rows interleaved per hidden unit,
 * MFMA-fragment order).  Synchronises `stream` before returning; the source
 * tensors may be freed afterwards. */
int ttsx_t2_decoder_create(const ttsx_t2_decoder_weights* w, int r_init, int attn_norm,
                           void* stream, ttsx_t2_decoder** out);
int ttsx_t2_decoder_destroy(ttsx_t2_decoder* dec);

/* Greedy decode of B utterances (Decoder.inference, models.py:643-673).
 *   enc      [B,T,512]  encoder outputs
 *   enc_len  [B] int32 valid encoder length per utterance, or NULL (= T for all);
 *            positions >= enc_len get zero attention, as a batch-1 run of the
 *            unpadded utterance would give
 *   r        frames kept per step (Decoder.set_r, models.py:527-528), 1..r_init
 *   max_steps  the reference's max_decoder_steps guard (models.py:663-665)
 *   stop_threshold  0.7 in the reference (models.py:661), strict '>'
 *   fixed_steps  non-zero: ignore the stop token and run exactly max_steps
 *   mel      [B, max_steps*r, 80]   decoder_output as Tacotron2.inference returns it
 *   align    [B, max_steps, T]
 *   stop     [B, max_steps]         post-sigmoid stop tokens
 *   n_steps  [B] int32              steps until (and including) the first stop frame,
 *                                   or the number of steps run if it never fired
 *   steps_run_host  host int: number of decoder steps executed (max over b)
 * Rows past an utterance's n_steps are whatever a run that kept stepping
 * produced; rows past *steps_run_host are not written.  Synchronises `stream`. */
int ttsx_t2_decoder_run(ttsx_t2_decoder* dec, const float* enc, const int32_t* enc_len, int B, int T,
                        int r, int max_steps, float stop_threshold, int fixed_steps, float* mel,
                        float* align, float* stop, int32_t* n_steps, int* steps_run_host, void* stream);

/* Step-level access for teacher-forced parity checks (Decoder.decode,
 * models.py:578-617).  begin() zero-initialises the recurrent state
 * (models.py:536-552, 240-249) and computes processed_inputs; load_state()
 * overwrites it; step() runs ONE decode step from `memory` [B,80] (the last
 * mel frame of the previous step) and writes frames [B,r,80], alignment [B,T]
 * and the post-sigmoid stop token [B]; read_state() copies the state out.
 * State tensors are plain row-major: h_a,c_a,h_d,c_d [B,1024]; ctx [B,512];
 * a_prev,a_cum [B,T].  Any state pointer may be NULL (left untouched / not read). */
int ttsx_t2_decoder_begin(ttsx_t2_decoder* dec, const float* enc, const int32_t* enc_len, int B, int T,
                          void* stream);
int ttsx_t2_decoder_load_state(ttsx_t2_decoder* dec, const float* h_a, const float* c_a, const float* h_d,
                               const float* c_d, const float* ctx, const float* a_prev, const float* a_cum,
                               void* stream);
int ttsx_t2_decoder_step(ttsx_t2_decoder* dec, const float* memory, int r, float* frames, float* align,
                         float* stop, void* stream);
int ttsx_t2_decoder_read_state(ttsx_t2_decoder* dec, float* h_a, float* c_a, float* h_d, float* c_d,
                               float* ctx, float* a_prev, float* a_cum, void* stream);

/* ------------------------------------------------------------------ Tacotron2 postnet
 * Replaces Postnet.forward + the residual add: Tacotron2/TTS/models.py:402-417,
 * 809-810 (five ConvBNBlocks, :380-400; dropout is identity in eval).
 * Tensor names are the state_dict keys under "postnet.convolutions.i.".
 */
typedef struct {
  const float* conv_w[5];   /* net.0.weight [Cout,Cin,5]: (512,80,5), 3x(512,512,5), (80,512,5) */
  const float* conv_b[5];   /* net.0.bias   [Cout] */
  const float* bn_weight[5];/* net.1.weight / bias / running_mean / running_var [Cout] */
  const float* bn_bias[5];
  const float* bn_mean[5];
  const float* bn_var[5];
} ttsx_postnet_weights;

typedef struct ttsx_postnet ttsx_postnet;

int ttsx_postnet_create(const ttsx_postnet_weights* w, void* stream, ttsx_postnet** out);
int ttsx_postnet_destroy(ttsx_postnet* p);
/*   mel       [B,F,80] decoder output (frame-major, as Tacotron2.inference returns it)
 *   n_frames  [B] int32 valid frames per utterance or NULL (= F); frames >= n_frames are
 *             treated as absent (zero padding of the convolutions starts there)
 *   out       [B,F,80] mel + postnet(mel)                                          */
int ttsx_postnet_run(ttsx_postnet* p, const float* mel, const int32_t* n_frames, int B, int F, float* out,
                     void* stream);

/* ------------------------------------------------------------------ Tacotron2 encoder convolutions
 * Replaces the three ConvBNBlocks of Encoder.inference: Tacotron2/TTS/models.py:449-451 with
 * :380-400 (conv1d k=5, 512->512, + BatchNorm (eval) + ReLU; dropout is identity in eval).
 * Tensor names are the state_dict keys under "encoder.convolutions.i.".  The embedding lookup
 * and the BiLSTM around it stay stock PyTorch ops.
 */
typedef struct {
  const float* conv_w[3];   /* net.0.weight [512,512,5] */
  const float* conv_b[3];   /* net.0.bias   [512] */
  const float* bn_weight[3];/* net.1.weight / bias / running_mean / running_var [512] */
  const float* bn_bias[3];
  const float* bn_mean[3];
  const float* bn_var[3];
} ttsx_encoder_conv_weights;

typedef struct ttsx_encoder_convs ttsx_encoder_convs;

int ttsx_encoder_convs_create(const ttsx_encoder_conv_weights* w, void* stream, ttsx_encoder_convs** out);
int ttsx_encoder_convs_destroy(ttsx_encoder_convs* p);
/*   x        [B,T,512] embedded text, position-major (embedding output as it is)
 *   lengths  [B] int32 valid positions per utterance or NULL (= T); positions >= length are
 *            treated as absent (zero padding), as a batch-1 run of the unpadded text gives
 *   out      [B,T,512]                                                                  */
int ttsx_encoder_convs_run(ttsx_encoder_convs* p, const float* x, const int32_t* lengths, int B, int T, float* out,
                           void* stream);

/* ------------------------------------------------------------------ vocoder (Griffin-Lim)
 * Replaces the inverse path of AudioProcessor: Tacotron2/TTS/utils.py:834-851
 * (_denormalize), 866-868 (_db_to_amp), 802-804 (_mel_to_linear), 922-929
 * (_griffin_lim), 931-942 (_stft/_istft via librosa), 875-878
 * (apply_inv_preemphasis); Tacotron/TTS/utils.py:109-125, 169-208 for the
 * linear-spectrogram variant.
 */
typedef struct {
  int n_fft;          /* 2048 (only value supported) */
  int hop_length;     /* 275 */
  int win_length;     /* 1100 (Tacotron2) or 1102 (Tacotron); <= n_fft */
  int pad_mode;       /* TTSX_PAD_CONSTANT (Tacotron2) or TTSX_PAD_REFLECT (Tacotron) */
} ttsx_stft_params;

typedef struct {
  int signal_norm;    /* config "signal_norm" */
  int symmetric_norm; /* config "symmetric_norm" */
  int clip_norm;      /* config "clip_norm" */
  float max_norm;     /* config "max_norm" */
  float min_level_db; /* config "min_level_db" */
  float ref_level_db; /* config "ref_level_db" */
  float power;        /* config "power" */
} ttsx_audio_params;

/* spec -> magnitude for Griffin-Lim.
 *   spec      [B,F,C] normalised spectrogram, frame-major (C = 80 mel bins with inv_basis,
 *             or C = n_freq linear bins with inv_basis == NULL)
 *   inv_basis [n_freq,80] pinv of the mel filterbank (an explicit input), or NULL
 *   mag       [B,F,n_freq] max(1e-10, inv_basis . 10^((denorm(spec)+ref)/20)) ^ power      */
int ttsx_spec_to_mag(const float* spec, const float* inv_basis, int B, int F, int C, int n_freq,
                     const ttsx_audio_params* ap, float* mag, void* stream);

/* Batched Griffin-Lim + de-emphasis.
 *   mag        [B,F,n_freq] target magnitudes (frame-major)
 *   init_phase [B,F,n_freq] initial phase as a fraction of a turn in [0,1) (the reference
 *              draws it from the global NumPy RNG, utils.py:923)
 *   n_frames   [B] int32 valid frames per utterance or NULL (= F)
 *   iters      griffin_lim_iters (60)
 *   preemph    de-emphasis coefficient (0.98); 0 skips the filter
 *   wav        [B, hop*(F-1)] ; utterance b holds hop*(n_frames[b]-1) samples, rest zero   */
int ttsx_griffin_lim(const float* mag, const float* init_phase, const int32_t* n_frames, int B, int F,
                     const ttsx_stft_params* sp, int iters, float preemph, float* wav, void* stream);

/* Unit-level entry points (librosa.stft / librosa.istft semantics, center=True).
 *   y  [B, hop*(F-1)]   spec_re/spec_im [B,F,n_freq]                                      */
int ttsx_stft(const float* y, int B, int F, const ttsx_stft_params* sp, float* spec_re, float* spec_im,
              void* stream);
int ttsx_istft(const float* spec_re, const float* spec_im, int B, int F, const ttsx_stft_params* sp,
               float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TTSX_H */
