// Persistent scan of the encoder BiLSTM (enc_scan.hip), called by ttsx_encoder_lstm_run* in t2_decoder.hip.
#pragma once
#include "common.h"

struct EncScanArgs {
  const float* w_hh[2];  // packed recurrent weights per direction (pack_layer with interleaved gate rows)
  const float* bias[2];  // b_ih + b_hh per direction, packed row order
  const float* gi[2];    // input projections of all steps per direction: float4 [256][Ct], column t * Bc + b
  float* hx;             // workspace, enc_scan_hx_floats(Bc) floats
  unsigned* steps;       // workspace, enc_scan_step_words(Bc) words
  int* abort_word;       // workspace, one int
  float* out;            // [B][T][512]
  const int32_t* len;    // [B] valid lengths or null
  float* h_state;        // [2][B][256] state to start from and to leave, or null (zero start)
  float* c_state;
  int T, B, Bc, Ct;
};

bool enc_scan_supports(int Bc);        // column counts the scan covers (all of its workgroups resident at once)
size_t enc_scan_hx_floats(int Bc);
size_t enc_scan_step_words(int Bc);
// Runs all T steps of both directions; synchronises `st`.  TTSX_ERR_STATE: the workgroups were not resident
// together and their bounded waits gave up (outputs undefined: the caller runs the launch chain instead).
int enc_scan_run(const EncScanArgs& a, hipStream_t st);
