// Arguments of the two chained per-class launches of the synthetic match (gs_match.hip), shared with their caller gs_engine.hip.
#pragma once
#include "gs_common.h"

struct gs_match1_args {
  const float* Q; const float* T; const float* mask; const float* X;
  const long* lo; const int* nc;       // per local class: first synthetic row, row count
  float* dHp; long s_dhp;              // [Cl][n x h]
  float* gW1; long s_gw;               // gW1_c = gW1 + c * s_gw : [d x h]
  float* gb1;                          // optional, gb1_c = gb1 + c * s_gw : [h]
  int n, h, d, ldx;
};

struct gs_match2_args {
  const float* X; const float* D; const float* mask; const float* Wl; const float* dHp;
  long sD, oW1, ob1, s_dhp;            // D1_c = D + c sD + oW1 : [d x h] ; Db1_c = D + c sD + ob1 : [h]
  float* Bc; long s_bc;                // [Cl][n x C]
  float* slabs; long s_slab;           // [Cl][n x d]
  int n, h, d, ldx, C;
};

bool gs_match_class_eligible(int n, int h, int d, int C);
int gs_match_first(const gs_match1_args*, int n_classes, hipStream_t);
int gs_match_second(const gs_match2_args*, int n_classes, hipStream_t);
