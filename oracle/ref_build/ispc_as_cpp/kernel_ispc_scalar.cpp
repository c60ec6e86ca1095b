// TEST INFRASTRUCTURE: the reference's kernel.ispc, compiled as ONE scalar program instance (see ispc_prelude.h).
// KERNEL_CPP is the translated file under oracle/_ref/ (Makefile).  Everything lands in namespace ispc, which is where
// the ispc-generated header declares the entry points that ispc_texcomp.cpp:419-439 calls.
#include "ispc_prelude.h"
namespace ispc {
#include "ispc_stdlib.inc"
#include KERNEL_CPP
}
