/*
 * oracle/ref_build/find_closest_glue.cpp -- TEST INFRASTRUCTURE.
 *
 * oracle/_ref/libdxtex_findclosest_ref.so: the reference's FindClosestUNORM (BC4BC5.cpp:314-337), which is `static` in its
 * translation unit, made callable by compiling that file -- from where it lies, unmodified -- INTO this one (the #include
 * below names a source file on purpose).  The function is a pure map (red_0, red_1, texel) -> 3-bit index; its whole domain
 * with 8-bit texels is 256^3 = 16.7 M cases, which tests/test_bc45_index_table.py walks exhaustively to pin both
 * oracle/bc4_bc5.c's restatement and the run-length (threshold) form csrc/bc4_bc5.hip evaluates.
 */
#include "directxtexp.h"
#include "BC.h"
#include "BC4BC5.cpp"

using namespace DirectX;

extern "C" {

/* For one endpoint pair and the 256 texel values v * (1/255.f) (the loader's SSE form, oracle/bc4_bc5.c header note 1):
 * out[v] = the index the reference stores.  16 texels per call of the reference function. */
void dxtex_ref_find_closest_row(int r0, int r1, uint8_t* out)
{
    BC4_UNORM bc;
    for (int base = 0; base < 256; base += 16) {
        float t[16];
        for (int i = 0; i < 16; i++) t[i] = (float)(base + i) * (1.0f / 255.0f);
        bc.data = 0;
        bc.red_0 = (uint8_t)r0;
        bc.red_1 = (uint8_t)r1;
        FindClosestUNORM(&bc, t);
        for (int i = 0; i < 16; i++) out[base + i] = (uint8_t)bc.GetIndex((size_t)i);
    }
}

/* the eight decoded levels of an endpoint pair, as the reference computes them */
void dxtex_ref_decode_levels(int r0, int r1, float* out)
{
    BC4_UNORM bc;
    bc.data = 0;
    bc.red_0 = (uint8_t)r0;
    bc.red_1 = (uint8_t)r1;
    for (int i = 0; i < 8; i++) out[i] = bc.DecodeFromIndex((size_t)i);
}

}
