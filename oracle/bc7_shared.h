/*
 * oracle/bc7_shared.h -- TEST INFRASTRUCTURE.  BC7 helpers that the BC6H
 * restatement reuses, mirroring kernel.ispc:2174-2300 / 2982-3031 calling into
 * kernel.ispc:688-1262 / 1694-1805.
 */
#ifndef ORACLE_BC7_SHARED_H
#define ORACLE_BC7_SHARED_H
#include <stdint.h>

const int32_t* get_unquant_table(int bits);
uint32_t get_pattern(int part_id);
int32_t  get_pattern_mask(int part_id, int j);
void     get_skips(int32_t skips[3], int part_id);

void  compute_stats_masked(float stats[15], const float block[64], int32_t mask, int channels);
void  covar_from_stats(float covar[10], const float stats[15], int channels);
void  block_segment_core(float ep[], const float block[64], int32_t mask, int channels);
float block_pca_bound_split(const float block[64], int32_t mask, const float full_stats[15], int channels);
float block_quant(uint32_t qblock[2], const float block[64], int bits, const float ep[], uint32_t pattern, int channels);
void  opt_endpoints(float ep[], const float block[64], int bits, const uint32_t qblock[2], int32_t mask, int channels);
void  partial_sort_list(int32_t list[], int length, int partial_count);

void    bc7_code_apply_swap_mode456(int32_t qep[], int channels, uint32_t qblock[2], int bits);
int32_t bc7_code_apply_swap_mode01237(int32_t qep[], uint32_t qblock[2], int mode, int part_id);
void    put_bits(uint32_t data[5], int* pos, int bits, int32_t v);
void    bc7_code_qblock(uint32_t data[5], int* pPos, const uint32_t qblock[2], int bits, int32_t flips);
void    bc7_code_adjust_skip_mode01237(uint32_t data[5], int mode, int part_id);

#endif
