// pending.hip -- formats whose kernels are not written yet abort loudly (no CPU path exists).
#include <cstdio>
#include <cstdlib>
#include "kernels.hpp"
namespace itw {
void launch_bc6h(const uint8_t*, int64_t, int, int, uint8_t*, const bc6h_enc_settings&, hipStream_t)
{ std::fprintf(stderr, "libispc_texcomp (itw-amd): BC6H kernel not built\n"); std::abort(); }
}
