// Tacotron (v1) greedy decoder as a chain of launches over the whole chip (row-block engine of
// t2_decoder.hip) instead of one workgroup per utterance.  Internal interface between
// t1_model.hip (owner of the ttsx_t1 handle and of the weights) and t2_decoder.hip (the engine).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

// Device pointers in the reference's state_dict layouts (Tacotron/TTS/models.py:487-536).
struct T1StepWeights {
  const float *pre_w0, *pre_b0, *pre_w1, *pre_b1;           // decoder.prenet.layers.{0,1}: [256,mem_dim], [128,256]
  const float *att_w_ih, *att_w_hh, *att_b_ih, *att_b_hh;   // attention_rnn.rnn_cell: [768,384], [768,256]
  const float *loc_conv_w, *loc_lin_w, *loc_lin_b;          // alignment_model.loc_conv.1 [32,2,31], loc_linear [128,32]
  const float *query_w, *query_b, *v_w;                     // query_layer [128,256], v [1,128]
  const float *pd_w, *pd_b;                                 // project_to_decoder_in [256,512]
  const float *rnn_w_ih[2], *rnn_w_hh[2], *rnn_b_ih[2], *rnn_b_hh[2];  // decoder_rnns.{0,1}: [768,256] x2
  const float *mel_w, *mel_b, *stop_w, *stop_b;             // proj_to_mel [80r,256], stopnet.linear [1,256+80r]
  const float *att_init, *mem_init, *rnn_inits;             // learned initial states
  int mem_dim, odim;                                        // 80*memory_size, 80*r
};

struct t1_steps;

int t1_steps_create(const T1StepWeights* w, hipStream_t st, t1_steps** out);
void t1_steps_destroy(t1_steps* s);
// Same contract as ttsx_t1_decoder_run (include/ttsx.h); `annot` [B,T,128] = annot_layer(enc) with its bias.
int t1_steps_run(t1_steps* s, const float* enc, const float* annot, const int32_t* lengths, int B, int T,
                 int max_decoder_steps, float* mel, float* align, float* stop, int32_t* n_steps, int* steps_run_host,
                 hipStream_t st);
