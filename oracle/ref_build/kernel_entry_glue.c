/*
 * oracle/ref_build/kernel_entry_glue.c -- TEST INFRASTRUCTURE.
 *
 * The five symbols the ispc compiler would emit for kernel.ispc's `export` functions (kernel.ispc:598-614, 2030-2037,
 * 3132-3139, 3683-3691).  ispc is absent, so they forward to the oracle's C restatement of the same functions; ETC1 is
 * outside the path and traps.  Together with the reference's unmodified ispc_texcomp.cpp this forms
 * oracle/_ref/libispc_texcomp_ref.so = "reference host TU (presets + ABI wrappers) over the restated kernel".
 */
#include <stdio.h>
#include <stdlib.h>
#include "../oracle.h"

void CompressBlocksBC1_ispc(void* src, uint8_t* dst) { oracle_CompressBlocksBC1((const oracle_surface*)src, dst); }
void CompressBlocksBC3_ispc(void* src, uint8_t* dst) { oracle_CompressBlocksBC3((const oracle_surface*)src, dst); }
void CompressBlocksBC7_ispc(void* src, uint8_t* dst, void* s)
{
    oracle_CompressBlocksBC7((const oracle_surface*)src, dst, (const oracle_bc7_settings*)s);
}
void CompressBlocksBC6H_ispc(void* src, uint8_t* dst, void* s)
{
    oracle_CompressBlocksBC6H((const oracle_surface*)src, dst, (const oracle_bc6h_settings*)s);
}
void CompressBlocksETC1_ispc(void* src, uint8_t* dst, void* s)
{
    (void)src; (void)dst; (void)s;
    fprintf(stderr, "oracle/_ref: ETC1 is outside the path this project restates (kernel.ispc:3141-3691)\n");
    abort();
}
