/*
 * oracle/ref_build/stub/kernel_ispc.h -- TEST INFRASTRUCTURE.
 *
 * Stand-in for the header `ispc --header-outfile` would generate from kernel.ispc (the real one is git-ignored
 * upstream and needs the ispc compiler, which this image does not have).  It declares exactly what the reference's
 * own translation unit /root/reference/3rdParty/Intel/Source/ispc_texcomp.cpp:18,417-440 needs from it: the
 * `ispc::` struct names it casts to and the five exported kernel entry points (kernel.ispc:598, 607, 2030, 3132,
 * 3683), with C linkage as ISPC emits them.  With this on the include path the reference's ispc_texcomp.cpp compiles
 * UNMODIFIED from where it lies; the entry points are supplied by kernel_entry_glue.c (-> the oracle's restatement:
 * libispc_texcomp_ref.so) or by kernel.ispc itself built as a scalar program (ispc_as_cpp/: libispc_texcomp_ref_full.so).
 */
#pragma once
#include <stdint.h>

namespace ispc {
struct rgba_surface;
struct bc7_enc_settings;
struct bc6h_enc_settings;
struct etc_enc_settings;

extern "C" {
void CompressBlocksBC1_ispc(rgba_surface* src, uint8_t* dst);
void CompressBlocksBC3_ispc(rgba_surface* src, uint8_t* dst);
void CompressBlocksBC7_ispc(rgba_surface* src, uint8_t* dst, bc7_enc_settings* settings);
void CompressBlocksBC6H_ispc(rgba_surface* src, uint8_t* dst, bc6h_enc_settings* settings);
void CompressBlocksETC1_ispc(rgba_surface* src, uint8_t* dst, etc_enc_settings* settings);
}
}
