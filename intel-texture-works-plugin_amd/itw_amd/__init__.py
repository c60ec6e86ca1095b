"""itw_amd -- thin host binding of libispc_texcomp.so (the MI355X-native BCn encoder).

The product is the C-ABI shared library built from ../csrc (extern "C"
CompressBlocksBC1/BC3/BC7/BC6H + GetProfile_*, the reference's own entry points,
/root/reference/3rdParty/Intel/Source/ispc_texcomp.h:67-107).  This package only
binds it with ctypes for tests and benchmarks and offers torch-tensor
conveniences (device memory, streams, torch.distributed are plumbing here).

There is no CPU implementation in this package: if the library is missing or no
GPU is present, calls raise / the library aborts.  Nothing here imports oracle/.
"""
from .abi import (  # noqa: F401
    lib, test_lib, TEST_HOOK_SYMBOLS, lib_path, RgbaSurface, Bc7Settings, Bc6hSettings, BC7_PROFILES, BC6H_PROFILES,
    bc7_profile, bc6h_profile, compress, compress_numpy, band_for_part, version, device_info,
    BYTES_PER_BLOCK, EXPORTED_SYMBOLS, DXGI_FORMAT, DdsDesc, image_func, compress_image, PROGRESS_FUNC, pad_to_multiple_of_4, dds_file, decode, block_count, KEEPS_PARTIAL_BLOCKS,
    available, set_error_mode, last_error, ON_ERROR_ABORT, ON_ERROR_RETURN, set_bc7_path, set_bc7_pilot, compress_image_multigpu, multigpu_sub_bands, MultiGpuStats, source_sha256, bc7_two_subset_bounds,
)
