/*
 * oracle/prepass.c -- TEST INFRASTRUCTURE.  CPU restatement of the Photoshop-buffer -> encoder-surface conversions that
 * run immediately before the ABI (IntelPlugin.cpp:741-810 ConvertToBCFrom8/16/32Bit, :291-366 ConvertToBC6From8/16/32Bit;
 * scalar helpers IntelPlugin.h:31-96).  Checker for csrc/convert.hip; never linked by the product.
 *
 * F32toF16 is DirectXMath's XMConvertFloatToHalf (not vendored in the reference tree): round to nearest even incl.
 * denormals for everything a half can hold -- restated below.  What it returns above 65504 differs between DirectXMath
 * releases (0x7FFF in the 2012-2015 ones, +-inf later); this restatement saturates to +-inf like the hardware conversion
 * and the fixtures' generator, and the parity tests stay inside the finite range ("parity unpinned" beyond it).
 * The 32-bit -> 8-bit path applies pow(v, 1/2.2) in double precision with the C library's pow -- the same call the reference's own
 * ConvertTo8Bit makes where it is compiled here (tests/test_reference_pins.py).  The kernel does not call pow: it counts the code thresholds
 * of that function (csrc/gamma_thresholds.h, tools/gen_gamma_thresholds.py) and is compared bit for bit.  (On another platform's C library the
 * thresholds could sit one float further: the pin is "the reference's function under this image's libm".)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static uint16_t f32_to_f16(float value)
{
    uint32_t x; memcpy(&x, &value, 4);
    const uint32_t sign = (x & 0x80000000u) >> 16;
    x &= 0x7fffffffu;
    uint32_t r;
    if (x > 0x477fe000u) {                                   /* too large for a half (or inf / NaN) */
        r = ((x & 0x7f800000u) == 0x7f800000u && (x & 0x7fffffu)) ? 0x7fffu : 0x7c00u;
    } else {
        if (x < 0x38800000u) {                               /* becomes a half denormal */
            const uint32_t shift = 113u - (x >> 23);
            x = shift < 32u ? (0x800000u | (x & 0x7fffffu)) >> shift : 0u;
        } else {
            x += 0xc8000000u;                                /* rebias the exponent */
        }
        r = ((x + 0x0fffu + ((x >> 13) & 1u)) >> 13) & 0x7fffu;
    }
    return (uint16_t)(r | sign);
}

static uint8_t float_to_byte(double v)                       /* IntelPlugin.h:41-48 */
{
    if (v > 1) return 255;
    else if (v < 0) return 0;
    return (uint8_t)(v * 255);
}

/* ConvertToBCFrom{8,16,32}Bit: `planes` interleaved source planes per pixel -> RGBA8.  Missing colour planes are 0,
 * alpha is 255 unless has_alpha (then plane 3).  depth 16: Photoshop's 0..32768 range; depth 32: optional 1/2.2 gamma. */
void oracle_convert_rgba8(const void* src, int depth, int planes, int has_alpha, int gamma, int width, int height, uint8_t* dst)
{
    for (long i = 0; i < (long)width * height; i++) {
        uint8_t px[4] = {0, 0, 0, 255};
        for (int c = 0; c < 4; c++) {
            const int present = (c < 3) ? (c < planes) : has_alpha;
            if (!present) continue;
            const long idx = i * planes + c;
            if (depth == 8) px[c] = ((const uint8_t*)src)[idx];
            else if (depth == 16) px[c] = float_to_byte(((const uint16_t*)src)[idx] / 32768.0);
            else {
                double v = ((const float*)src)[idx];
                if (gamma) v = pow(v, 1 / 2.2);
                px[c] = float_to_byte(v);
            }
        }
        memcpy(dst + 4 * i, px, 4);
    }
}

/* ConvertToBC6From{8,16,32}Bit -> RGBA16F bit patterns.  Alpha is 1.0 unless has_alpha; the 32-bit variant reads the
 * alpha from plane 2 (IntelPlugin.cpp:361: index+2, a reference quirk kept here). */
void oracle_convert_rgba16f(const void* src, int depth, int planes, int has_alpha, int width, int height, uint16_t* dst)
{
    for (long i = 0; i < (long)width * height; i++) {
        uint16_t px[4] = {0, 0, 0, 0x3c00};
        for (int c = 0; c < 4; c++) {
            const int present = (c < 3) ? (c < planes) : has_alpha;
            if (!present) continue;
            const long idx = i * planes + ((c == 3 && depth == 32) ? 2 : c);
            if (depth == 8) px[c] = f32_to_f16(((const uint8_t*)src)[idx] / 255.f);
            else if (depth == 16) px[c] = f32_to_f16((float)(((const uint16_t*)src)[idx] / 32768.0));
            else px[c] = f32_to_f16(((const float*)src)[idx]);
        }
        memcpy(dst + 4 * i, px, 8);
    }
}
