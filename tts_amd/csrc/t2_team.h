// Tacotron2 greedy decoder at batch 1 as ONE persistent launch with every weight resident on chip.
// Internal interface between t2_decoder.hip (owner of the ttsx_t2_decoder handle) and t2_team.hip.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

// Device pointers; the LSTM / attention tensors in the reference's state_dict layouts (Tacotron2/TTS/models.py:
// 466-525), the prenet layers with eval-mode BatchNorm folded in.
struct T2TeamWeights {
  const float *attn_w_ih, *attn_w_hh, *attn_b_ih, *attn_b_hh;  // attention_rnn: [4096,768], [4096,1024], [4096] x2
  const float *dec_w_ih, *dec_w_hh, *dec_b_ih, *dec_b_hh;      // decoder_rnn: [4096,1536], [4096,1024], [4096] x2
  const float *query_w, *v_w, *v_b;                            // [128,1024], [128], [1]
  const float *loc_conv_w, *loc_dense_w;                       // [32,2,31], [128,32]; both null: no location term
  const float *a2f, *b2f;                                      // prenet layer 2 folded: [256,256], [256]
  const float* b1f;                                            // prenet layer 1 folded bias [256] (layer 1 of a zero frame)
};

struct t2_team;

// Final state of a run in the layouts ttsx_t2_decoder keeps for batch 1 (plain vectors), so that
// ttsx_t2_decoder_set_initial(keep_states) works after a team run; any pointer may be null.
struct T2TeamState {
  float *h_a, *c_a, *h_d, *c_d, *ctx, *x1;
  float *h_a_alt, *h_d_alt;  // second copies of the hidden states (the launch chain ping-pongs them by step parity)
};

int t2_team_create(const T2TeamWeights* w, hipStream_t st, t2_team** out);
void t2_team_destroy(t2_team* t);
// Projection rows for reduction factor r, row-major [80 r + 16 + 256][1536] over [h_d; ctx] with their biases:
// the 80 r mel rows, the composed stop-logit row (first of 16), the 256 composed prenet-layer-1 rows.
int t2_team_set_proj(t2_team* t, const float* rows_w, const float* rows_b, int r, hipStream_t st);
// Shapes the team kernel covers: one utterance, encoder length up to 1024, reduction factor 1-2.
bool t2_team_supports(const t2_team* t, int B, int T, int r);
// One greedy decode (the contract of ttsx_t2_decoder_run, include/ttsx.h, for B = 1).  enc [T,512], P [T,128]
// processed inputs, len [1] on the device.  TTSX_ERR_STATE: the team's workgroups did not become resident together.
int t2_team_run(t2_team* t, const float* enc, const float* P, const int32_t* len, int T, int r, int max_steps,
                float stop_threshold, int fixed_steps, int norm, float* mel, float* align, float* stop,
                int32_t* n_steps, const T2TeamState* state_out, int* steps_run_host, hipStream_t st);
