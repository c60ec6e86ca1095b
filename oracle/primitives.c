/*
 * oracle/primitives.c -- TEST INFRASTRUCTURE.  Exports the pinned arithmetic
 * primitives so tests can compare the GPU's LUT/Newton implementation and, on
 * Intel hosts, the real RCPPS/RSQRTPS instructions against the model.
 */
#include "bc_common.h"

float   oracle_rcp(float v)     { return ispc_rcp(v); }
float   oracle_rsqrt(float v)   { return ispc_rsqrt(v); }
float   oracle_rcpps(float v)   { return x86_rcpps(v); }
float   oracle_rsqrtps(float v) { return x86_rsqrtps(v); }
int32_t oracle_f2i(float v)     { return f2i_x86(v); }
