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
 *     state_dict layouts and packed once at create time;
 *   - per-utterance length arrays (enc_len, lengths, n_frames, n_samples) are
 *     read on the device and clamped there to the range the call's shape
 *     allows (0 or 1 .. T / F / pitch): an out-of-range entry shortens or
 *     lengthens nothing beyond the buffers of the call.
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
/* Interface version of the loaded library; a caller compares it with the TTSX_VERSION it was built against.
 *   1  first interface
 *   2  ttsx_request_tail's `offsets` buffer holds B + 2 ints (the device keeps the peak word at [B + 1]; it was
 *      B + 1); ttsx_t2_decoder_run zero-fills rows between an utterance's own stop and *steps_run_host when one or
 *      two utterances end at their own stop frames */
#define TTSX_VERSION 2
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

/* Optional attention variants of the reference (all off in the repo's config.json:84-92):
 *   location_attn = false  OriginalAttention.get_attention (models.py:269-273): hand NULL for
 *                          loc_conv_w / loc_dense_w in ttsx_t2_decoder_weights;
 *   windowing              apply_windowing at inference (models.py:275-288; window = [idx-2, idx+6),
 *                          idx = argmax of the windowed energies, per utterance);
 *   forward_attn           apply_forward_attention (models.py:290-312), with forward_attn_mask its
 *                          inference-time masking (the index arithmetic is Python's: n-1 / n-2 may
 *                          wrap to the end of the utterance) and with trans_agent the transition
 *                          probability u = sigmoid(ta [context; query]) (models.py:357-360);
 *   graves_K > 0           GravesAttention with K mixtures (models.py:106-183) instead of the
 *                          original attention: query_w / inputs_w / v_w / v_b / loc_* may be NULL.
 * Call after create and before begin/run; tensors are copied.  Synchronises `stream`. */
typedef struct {
  int windowing;          /* config "windowing" */
  int forward_attn;       /* config "use_forward_attn" */
  int forward_attn_mask;  /* config "forward_attn_mask" */
  int trans_agent;        /* config "transition_agent" */
  const float* ta_w;      /* attention.ta.weight [1,1536] (input = [context 512; query 1024]) */
  const float* ta_b;      /* attention.ta.bias   [1] */
  int graves_K;           /* config "attention_heads" when attention_type is "graves", else 0; <= 16 */
  const float* graves_w1; /* attention.N_a.0.weight [1024,1024] */
  const float* graves_b1; /* attention.N_a.0.bias   [1024] */
  const float* graves_w2; /* attention.N_a.2.weight [3K,1024] */
  const float* graves_b2; /* attention.N_a.2.bias   [3K] */
} ttsx_t2_attention_options;
int ttsx_t2_decoder_set_attention(ttsx_t2_decoder* dec, const ttsx_t2_attention_options* opt, void* stream);

/* One-shot settings of the NEXT begin()/run() (Decoder.inference_truncated, models.py:675-709):
 *   keep_states  non-zero: the LSTM states and the context of the previous run survive
 *                (Decoder._init_states(keep_states=True), models.py:536-552); the attention state
 *                is reset as always.  The previous run must have had the same B.
 *   memory       [B,80] frame fed to the first step instead of the all-zero go frame
 *                (memory_truncated, models.py:690,703), copied at call time; or NULL. */
int ttsx_t2_decoder_set_initial(ttsx_t2_decoder* dec, int keep_states, const float* memory, int B, void* stream);

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
 * Rows past an utterance's n_steps and below *steps_run_host are either what a
 * run that kept stepping produced (batches that step together) or zero (one or
 * two utterances on the persistent kernels, which end at their own stop
 * frame); rows past *steps_run_host are not written.  Synchronises `stream`. */
int ttsx_t2_decoder_run(ttsx_t2_decoder* dec, const float* enc, const int32_t* enc_len, int B, int T,
                        int r, int max_steps, float stop_threshold, int fixed_steps, float* mel,
                        float* align, float* stop, int32_t* n_steps, int* steps_run_host, void* stream);

/* The launch plan of the following runs (no reference counterpart).  By default the library picks, by batch
 * size and text length, (a) one fused attention launch per step or three (query, energies, normalise + context),
 * (b) whole LSTM launches or an early / late split along K whose early part rides along other launches, (c) at
 * batch 1-2 without the fused kernel, whether the energies launch adds up the query shares itself.  All choices
 * compute the same step (rounding differs at 1e-6).  Each argument: -1 by shape (default), 0 off, 1 on wherever
 * the shapes allow.  For A/B timing and for tests that hold the alternatives against each other. */
int ttsx_t2_decoder_set_plan(ttsx_t2_decoder* dec, int fused_attention, int split_lstm, int share_query);

/* Batches padded to 64 columns with the early / late LSTM split and the fused attention launch: how the early
 * columns of the decoder LSTM, which run beside the attention, are tiled (no reference counterpart).  0: 32 x 32
 * register tiles over two halves of K, two workgroups per CU; 1: 128 rows x 64 columns x one of six K parts per
 * workgroup with the activation slice shared through LDS, every workgroup of the launch on a CU of its own (the
 * attention workgroups no longer share theirs); -1 (default): staged wherever the shapes allow.  The step is the same
 * (rounding differs at 1e-6: six partial sums instead of two). */
int ttsx_t2_decoder_set_pieces(ttsx_t2_decoder* dec, int staged);

/* Batches on the MFMA engine (B > 2): launches of a step folded into the launch that produces their input, their
 * workgroups waiting for step tags that the producers raise (one launch boundary less per step each).  No
 * reference counterpart.  1: prenet layer 2 of a step inside the projection launch of the step before it, on every
 * batch above 2; 2: also the late columns of the decoder LSTM inside the attention launch, where that launch runs
 * staged pieces (ttsx_t2_decoder_set_pieces); 0: neither; -1 (default): both where they measured faster (padded
 * batch 64 with the early / late LSTM split).  The results are bit-equal in every setting.  A waiting workgroup
 * that sees no tag within its bound makes the run start over without these launches (default setting) or
 * return TTSX_ERR_STATE (settings 1, 2). */
int ttsx_t2_decoder_set_fusion(ttsx_t2_decoder* dec, int prenet_in_projection);

/* Whether a run of one or two utterances may use the persistent team kernel (no reference counterpart): texts of up to
 * 1024 symbols, reduction factor 1-2, plain location-sensitive attention, no state carried over by
 * ttsx_t2_decoder_set_initial.  One launch decodes a whole utterance (two utterances: two launches) with all decoder weights held in the
 * registers and LDS of 216 workgroups (240 beyond 256 symbols), which must be resident together (one CU each): meant for a GPU that serves one
 * request stream.  -1 by shape (default; a run whose team cannot become resident falls back to the launch chain and
 * the handle stays with it), 0 never, 1 wherever the shapes allow (such a run fails with TTSX_ERR_STATE instead). */
int ttsx_t2_decoder_set_team(ttsx_t2_decoder* dec, int team);

/* How a free run at batch 1-2 watches for its stop (no reference counterpart; the loop of models.py:651-668 asks
 * the device after every step).  The host enqueues graphs of `chunk` decoder steps and keeps up to `ahead`
 * steps outstanding while it polls a word the device writes into pinned host memory; at most ahead - 1 steps
 * are enqueued past the stop frame.  Short chunks notice the stop sooner but cost more per step.  0 / 0 restores
 * the defaults.  chunk: even, 2..32; ahead: chunk..128. */
int ttsx_t2_decoder_set_polling(ttsx_t2_decoder* dec, int chunk, int ahead);

/* Decoder steps the last ttsx_t2_decoder_run enqueued.  A free run notices the stop token of its last
 * utterance a few steps late (the loop of models.py:651-668 asks the device after every step; here the
 * host looks at a flag the device raises); the steps enqueued past it do no work that is kept.  No reference
 * counterpart: a diagnostic for latency accounting. */
int ttsx_t2_decoder_steps_launched(const ttsx_t2_decoder* dec);

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

/* Encoder BiLSTM: nn.LSTM(512, 256, bidirectional, batch_first) of Encoder (Tacotron2/TTS/models.py:419-427)
 * as run by Encoder.inference (:449-454).  Index 0 = forward (encoder.lstm.*_l0), 1 = reverse (*_l0_reverse). */
typedef struct {
  const float* w_ih[2]; /* encoder.lstm.weight_ih_l0[_reverse] [1024,512] (gate order i,f,g,o) */
  const float* w_hh[2]; /* encoder.lstm.weight_hh_l0[_reverse] [1024,256] */
  const float* b_ih[2]; /* encoder.lstm.bias_ih_l0[_reverse]   [1024] */
  const float* b_hh[2]; /* encoder.lstm.bias_hh_l0[_reverse]   [1024] */
} ttsx_encoder_lstm_weights;

typedef struct ttsx_encoder_lstm ttsx_encoder_lstm;

int ttsx_encoder_lstm_create(const ttsx_encoder_lstm_weights* w, void* stream, ttsx_encoder_lstm** out);
int ttsx_encoder_lstm_destroy(ttsx_encoder_lstm* p);
/*   x        [B,T,512] output of the conv blocks, position-major
 *   lengths  [B] int32 or NULL (= T): an utterance is processed as its own unpadded sequence
 *            (what pack_padded_sequence gives); rows at positions >= length are zero
 *   out      [B,T,512] = [forward 256 | backward 256]                                    */
int ttsx_encoder_lstm_run(ttsx_encoder_lstm* p, const float* x, const int32_t* lengths, int B, int T, float* out,
                          void* stream);

/* Encoder.inference_truncated (models.py:456-464): the BiLSTM starts from a carried state and
 * returns the final one.  h_state / c_state [2][B][256] (index 0 forward, 1 reverse: nn.LSTM's
 * (h_n, c_n) layout), read and overwritten; both NULL = ttsx_encoder_lstm_run.  lengths must be NULL. */
int ttsx_encoder_lstm_run_stateful(ttsx_encoder_lstm* p, const float* x, const int32_t* lengths, int B, int T,
                                   float* out, float* h_state, float* c_state, void* stream);

/* How the recurrence of the following runs is executed (no reference counterpart): 1 = as one persistent launch
 * over all T steps of both directions (a workgroup per row block keeps its slice of W_hh and its cell states in
 * registers; h(t) travels between them through step counters), wherever all of its workgroups fit the chip at once
 * (up to 128 padded utterance columns); 0 = as a chain of T step launches; -1 (default) = the persistent launch at
 * batch 1-2, where it measured faster (3.0 against 4.2 ms for 1000 symbols), and the chain after a launch whose
 * workgroups were not resident together (bounded waits).  With the persistent launch ttsx_encoder_lstm_run*
 * synchronise `stream`. */
int ttsx_encoder_lstm_set_plan(ttsx_encoder_lstm* lstm, int scan);

/* Tacotron2._add_speaker_embedding (models.py:838-849): enc[b][t][:] += table[speaker_ids[b]][:].
 *   enc [B,T,512] in place, table = speaker_embedding.weight [n_speakers,512], speaker_ids [B] int32. */
int ttsx_add_speaker_embedding(float* enc, const float* table, const int32_t* speaker_ids, int n_speakers, int B,
                               int T, void* stream);

/* ------------------------------------------------------------------ Tacotron (v1)
 * Replaces Tacotron.forward in eval mode, Tacotron/TTS/models.py:692-700:
 *   encoder   Prenet (:250-253) + CBHG K=16 (:412-434, BatchNormConv1d :307-313, Highway :331-334)
 *   decoder   greedy Decoder.forward (:586-645) with AttentionRNNCell (:124-156),
 *             LocationSensitiveAttention (:71-88), StopNet (:663-667), learned initial states (:538-556)
 *   postnet   PostCBHG K=8 (:471-484) + last_linear + sigmoid (:690, :699)
 * Tensor names are the reference state_dict keys.  The embedding lookup stays a stock op.
 */
typedef struct {
  int K;                          /* number of conv banks: 16 (encoder) or 8 (postnet) */
  int in_features;                /* 128 (encoder) or 80 (postnet) */
  int proj_dims[2];               /* conv_projections: {128,128} or {256,80} */
  const float* bank_w[16];        /* conv1d_banks.k.conv1d.weight [128,in,k+1] */
  const float* bank_bn_weight[16];/* conv1d_banks.k.bn.{weight,bias,running_mean,running_var} [128] */
  const float* bank_bn_bias[16];
  const float* bank_bn_mean[16];
  const float* bank_bn_var[16];
  const float* proj_w[2];         /* conv1d_projections.i.conv1d.weight [out,in,3] */
  const float* proj_bn_weight[2];
  const float* proj_bn_bias[2];
  const float* proj_bn_mean[2];
  const float* proj_bn_var[2];
  const float* pre_highway_w;     /* pre_highway.weight [128,proj_dims[1]] or NULL (encoder) */
  const float* hw_H_w[4];         /* highways.i.H.weight [128,128], .bias [128] */
  const float* hw_H_b[4];
  const float* hw_T_w[4];         /* highways.i.T.weight / .bias */
  const float* hw_T_b[4];
  const float* gru_w_ih[2];       /* gru.weight_ih_l0 [384,128], gru.weight_ih_l0_reverse */
  const float* gru_w_hh[2];       /* gru.weight_hh_l0 [384,128], ..._reverse */
  const float* gru_b_ih[2];
  const float* gru_b_hh[2];
} ttsx_cbhg_weights;

typedef struct {
  const float* enc_prenet_w[2];   /* encoder.prenet.layers.{0,1}.weight [256,256], [128,256]; .bias */
  const float* enc_prenet_b[2];
  ttsx_cbhg_weights enc_cbhg;     /* encoder.cbhg.cbhg.* */
  const float* dec_prenet_w[2];   /* decoder.prenet.layers.{0,1}.weight [256,80*memory_size], [128,256]; .bias */
  const float* dec_prenet_b[2];
  const float* att_rnn_w_ih;      /* decoder.attention_rnn.rnn_cell.weight_ih [768,384] (input [prenet 128; context 256]) */
  const float* att_rnn_w_hh;      /* [768,256] */
  const float* att_rnn_b_ih;
  const float* att_rnn_b_hh;
  const float* loc_conv_w;        /* ...alignment_model.loc_conv.1.weight [32,2,31] */
  const float* loc_linear_w;      /* ...loc_linear.weight [128,32], .bias [128] */
  const float* loc_linear_b;
  const float* query_w;           /* ...query_layer.weight [128,256], .bias */
  const float* query_b;
  const float* annot_w;           /* ...annot_layer.weight [128,256], .bias */
  const float* annot_b;
  const float* v_w;               /* ...v.weight [1,128] (no bias) */
  const float* proj_dec_in_w;     /* decoder.project_to_decoder_in.weight [256,512], .bias */
  const float* proj_dec_in_b;
  const float* dec_rnn_w_ih[2];   /* decoder.decoder_rnns.{0,1}.weight_ih [768,256] ... */
  const float* dec_rnn_w_hh[2];
  const float* dec_rnn_b_ih[2];
  const float* dec_rnn_b_hh[2];
  const float* proj_mel_w;        /* decoder.proj_to_mel.weight [80*r,256], .bias */
  const float* proj_mel_b;
  const float* att_rnn_init;      /* decoder.attention_rnn_init.weight [1,256] */
  const float* memory_init;       /* decoder.memory_init.weight [1,80*memory_size] */
  const float* dec_rnn_inits;     /* decoder.decoder_rnn_inits.weight [2,256] */
  const float* stop_w;            /* decoder.stopnet.linear.weight [1,256+80*r], .bias [1] */
  const float* stop_b;
  ttsx_cbhg_weights post_cbhg;    /* postnet.cbhg.* */
  const float* last_linear_w;     /* last_linear.0.weight [1025,256], .bias */
  const float* last_linear_b;
} ttsx_t1_weights;

typedef struct ttsx_t1 ttsx_t1;

int ttsx_t1_create(const ttsx_t1_weights* w, int r, int memory_size, int linear_dim, void* stream, ttsx_t1** out);
int ttsx_t1_destroy(ttsx_t1* m);
/* emb [B,T,256] embedded characters -> enc [B,T,256] (Encoder.forward, :459-469).
 * lengths [B] int32 valid positions per utterance or NULL (= T): positions >= length are treated
 * as absent, so every utterance is encoded as its unpadded batch-1 run would be; their rows of
 * enc are zero. */
int ttsx_t1_encoder_run(ttsx_t1* m, const float* emb, const int32_t* lengths, int B, int T, float* enc, void* stream);
/* Greedy decode (Decoder.forward with memory=None in eval mode).  Stops after step t (1-based)
 * when t > T/4 and (stop > 0.6 or attention on the last position > 0.6), or when
 * t > max_decoder_steps (:635-640), so at most max_decoder_steps+1 steps run.
 *   mel   [B, (max_decoder_steps+1)*r, 80]  sigmoid-bounded frames
 *   align [B, max_decoder_steps+1, T]       stop [B, max_decoder_steps+1]
 *   n_steps [B] int32 steps each utterance ran; *steps_run_host = their maximum.
 * lengths [B] int32 or NULL as above: attention runs over the first lengths[b] positions and the
 * stop rule uses that length; align columns beyond it are zero.  Rows of mel / align / stop past
 * n_steps[b] are not written.  T <= 512.  Synchronises `stream`.
 * At batch 1-2 with T <= 128 the whole loop is one persistent launch of a 36-workgroup team per utterance that
 * keeps every decoder weight in registers / LDS; otherwise seven launches per step over the whole chip (a hipGraph
 * per 32 steps).  ttsx_t1_set_plan overrides the choice. */
int ttsx_t1_decoder_run(ttsx_t1* m, const float* enc, const int32_t* lengths, int B, int T, int max_decoder_steps,
                        float* mel, float* align, float* stop, int32_t* n_steps, int* steps_run_host, void* stream);
/* Decoder form of the following runs (no reference counterpart): -1 by shape (default), 0 the launch chain,
 * 1 the on-chip team kernel where the shapes allow (B <= 2, T <= 128, memory queue of 400 floats, r = 2), 2 the
 * single-workgroup-per-utterance kernel.  All compute the same steps (rounding differs at 1e-6). */
int ttsx_t1_set_plan(ttsx_t1* m, int decoder);

/* mel [B,F,80] -> linear [B,F,linear_dim] = sigmoid(last_linear(PostCBHG(mel))) (:698-699).
 * n_frames [B] int32 valid frames per utterance or NULL (= F); rows past it are unspecified. */
int ttsx_t1_postnet_run(ttsx_t1* m, const float* mel, const int32_t* n_frames, int B, int F, float* linear,
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

/* Vocoder handle: the twiddle / window tables of one STFT geometry and the scratch signals Griffin-Lim ping-pongs
 * between, on the device that is current at create.  Calls on a handle are enqueued on the caller's stream and
 * do NOT synchronise (results are stream-ordered); scratch grows on demand (growing waits for the stream).  One
 * handle serves one caller at a time; handles share nothing, so callers with a handle each can overlap.
 * The library keeps no vocoder state outside handles. */
typedef struct ttsx_vocoder ttsx_vocoder;
int ttsx_vocoder_create(const ttsx_stft_params* sp, void* stream, ttsx_vocoder** out);
int ttsx_vocoder_destroy(ttsx_vocoder* v);

/* Batched Griffin-Lim + de-emphasis (AudioProcessor._griffin_lim, utils.py:922-929, + apply_inv_preemphasis).
 *   mag        [B,F,n_freq] target magnitudes (frame-major)
 *   init_phase [B,F,n_freq] initial phase as a fraction of a turn in [0,1) (the reference
 *              draws it from the global NumPy RNG, utils.py:923)
 *   n_frames   [B] int32 valid frames per utterance or NULL (= F); values outside 0..F are clamped
 *   iters      griffin_lim_iters (60)
 *   preemph    de-emphasis coefficient (0.98); 0 skips the filter; |preemph| <= 0.9899 (the filter runs as a blocked
 *              scan that cuts its impulse response off after 2048 taps: |a|^2048 < 1e-9)
 *   wav        [B, hop*(F-1)] ; utterance b holds hop*(n_frames[b]-1) samples, rest zero   */
int ttsx_vocoder_griffin_lim(ttsx_vocoder* v, const float* mag, const float* init_phase, const int32_t* n_frames,
                             int B, int F, int iters, float preemph, float* wav, void* stream);

/* Unit-level entry points (librosa.stft / librosa.istft semantics, center=True; utils.py:931-942).
 *   y  [B, hop*(F-1)]   spec_re/spec_im [B,F,n_freq]                                      */
int ttsx_vocoder_stft(ttsx_vocoder* v, const float* y, int B, int F, float* spec_re, float* spec_im, void* stream);
int ttsx_vocoder_istft(ttsx_vocoder* v, const float* spec_re, const float* spec_im, int B, int F, float* y,
                       void* stream);

/* One-shot forms of the three calls above: tables and scratch are allocated for the call and released before it
 * returns, which synchronises `stream`.  For occasional calls; a serving loop keeps a handle. */
int ttsx_griffin_lim(const float* mag, const float* init_phase, const int32_t* n_frames, int B, int F,
                     const ttsx_stft_params* sp, int iters, float preemph, float* wav, void* stream);
int ttsx_stft(const float* y, int B, int F, const ttsx_stft_params* sp, float* spec_re, float* spec_im,
              void* stream);
int ttsx_istft(const float* spec_re, const float* spec_im, int B, int F, const ttsx_stft_params* sp,
               float* y, void* stream);

/* ttsx_vocoder_griffin_lim without any device allocation (no counterpart in the reference).  For callers that
 * cannot allocate: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Tables
 * and the two padded signals live in a static device array of the library, so
 *   B * round_up(n_fft + hop*(F-1), 4)  must not exceed 4 194 304 floats (e.g. 16 utterances of 900 frames);
 * larger calls are rejected with TTSX_ERR_ARG.  The array (32 MiB) exists once per GPU for the life of the
 * process and every call on that GPU uses it: calls that overlap on one GPU race (serialise them, or use
 * handles).  Arguments and results as ttsx_vocoder_griffin_lim; does not synchronise. */
int ttsx_griffin_lim_noalloc(const float* mag, const float* init_phase, const int32_t* n_frames, int B, int F,
                             const ttsx_stft_params* sp, int iters, float preemph, float* wav, void* stream);

/* ------------------------------------------------------------------ post-vocoder tail of a request
 * Replaces the per-sentence tail of Synthesizer.tts (Tacotron2/TTS/synthesizer.py:141-174):
 * trim_silence -> AudioProcessor.find_endpoint (Tacotron2/TTS/utils.py:944-951), concatenation with
 * `gap` (10 000) zero samples after every sentence, and the int16 scaling of save_wav (utils.py:794-796).
 *   wav        [B,pitch] float32 waveforms, n_samples [B] int32 valid samples per sentence or NULL (= pitch)
 *   window_length = int(sample_rate * 0.8), hop_length = int(window_length / 4), threshold = 10^(-40/20)
 *   do_trim    0: keep every sentence whole
 *   endpoints  [B] int32 out, offsets [B+2] int32 out (start of every sentence in pcm; total at [B]; [B+1] is
 *              working space of the call: no device memory is allocated)
 *   pcm        [capacity] int16 out, capacity >= B*(pitch+gap), at most 2^31-1;  *total_host = samples written,
 *              *peak_host = max |sample| before scaling.  Synchronises `stream`.                       */
int ttsx_request_tail(const float* wav, const int32_t* n_samples, int B, int pitch, int window_length, int hop_length,
                      float threshold, int gap, int do_trim, int32_t* endpoints, int32_t* offsets, int16_t* pcm,
                      long capacity, long* total_host, float* peak_host, void* stream);

/* ------------------------------------------------------------------ testing hook
 *
 * No counterpart in the reference.  Some launches here hold workgroups that wait - for a bounded time - for other
 * workgroups of the same launch (the persistent teams of batch 1-2, the encoder scan, the fused launches of the
 * batched decoder step).  A wait that expires makes the run start over on the plain launch chain (default plans) or
 * fail with TTSX_ERR_STATE (a plan forced by its setter).  That only happens when other work holds the CUs, so a
 * test cannot provoke it: this call makes the NEXT run of each named kind, on any handle of the process, report an
 * expired wait once its launches have ended.  `kinds` is a sum of the values below.  For tests; a server has no use
 * for it. */
#define TTSX_WAIT_T2_STEP_FUSION 1 /* ttsx_t2_decoder_run, batched launch chain (ttsx_t2_decoder_set_fusion) */
#define TTSX_WAIT_T2_TEAM 2        /* ttsx_t2_decoder_run, batch 1-2 (ttsx_t2_decoder_set_team) */
#define TTSX_WAIT_T1_TEAM 4        /* ttsx_t1_decoder_run, batch 1-2 (ttsx_t1_set_plan) */
#define TTSX_WAIT_ENCODER_SCAN 8   /* ttsx_encoder_lstm_run, batch 1-2 (ttsx_encoder_lstm_set_plan) */
int ttsx_debug_fail_next_wait(unsigned kinds);

#ifdef __cplusplus
}
#endif
#endif /* TTSX_H */
