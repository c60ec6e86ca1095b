/*
 * oracle/ref_build/ref_convert_glue.cpp -- TEST INFRASTRUCTURE.
 *
 * C face of oracle/_ref/libintelplugin_convert_ref.so = the reference's own scalar pixel conversions, compiled from where
 * they lie: lines 31-96 of /root/reference/IntelCompressionPlugin/IntelPlugin.h (F16toF32, F32toF16, FloatToByte, F16toByte,
 * the three ConvertTo8Bit and the three ConvertTo16Bit overloads) are cut out by the Makefile into a build intermediate
 * (REF_CONVERT_INC, deleted after linking; the header as a whole needs the Photoshop SDK) and included here behind a typedef
 * shim: the SDK's `unsigned8` / `unsigned16` and the two DirectXMath half conversions (dxmath_stub/, as for the block
 * codecs).  tests/test_reference_pins.py compares them exhaustively with oracle/prepass.c, the checker of csrc/convert.hip
 * (the last oracle piece no reference source pinned).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "directxpackedvector.h"

typedef unsigned char unsigned8;        /* PITypes.h of the Photoshop SDK */
typedef unsigned short unsigned16;

#include REF_CONVERT_INC

extern "C" {
uint8_t ref_convert8_from8(uint8_t v) { return ConvertTo8Bit((unsigned8)v); }
uint8_t ref_convert8_from16(uint16_t v) { return ConvertTo8Bit((unsigned16)v); }
uint8_t ref_convert8_from32(float v, int gamma) { return ConvertTo8Bit((double)v, gamma != 0); }
uint16_t ref_convert16_from8(uint8_t v) { return ConvertTo16Bit((unsigned8)v); }
uint16_t ref_convert16_from16(uint16_t v) { return ConvertTo16Bit((unsigned16)v); }
uint16_t ref_convert16_from32(float v) { return ConvertTo16Bit(v); }
}
