// Tacotron (v1) greedy decoder at batch 1-2 as ONE persistent launch with every weight resident on chip.
// Internal interface between t1_model.hip (owner of the ttsx_t1 handle) and t1_team.hip.
#pragma once
#include "t1_steps.h"

struct t1_team;

// Packs the decoder weights (reference state_dict layouts, device pointers) into the per-thread register images
// and the LDS image of the team kernel.  Only the repo's geometry is supported (memory queue of 400 floats,
// 160 outputs per step); returns TTSX_OK with *out = nullptr for any other.
int t1_team_create(const T1StepWeights* w, hipStream_t st, t1_team** out);
void t1_team_destroy(t1_team* t);
// Shapes the team kernel covers: batch 1-2, encoder length up to 128.
bool t1_team_supports(const t1_team* t, int B, int T);
// Same contract as ttsx_t1_decoder_run (include/ttsx.h); `annot` [B,T,128] = annot_layer(enc) with its bias.
int t1_team_run(t1_team* t, const float* enc, const float* annot, const int32_t* lengths, int B, int T,
                int max_decoder_steps, float* mel, float* align, float* stop, int32_t* n_steps, int* steps_run_host,
                hipStream_t st);
