"""Host-pointer call time of BC7 slow / BC6H slow for ITW_HOST_CHUNKS = 1, 2, 4, 8 and pageable vs pinned host memory."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
    import numpy as np, torch, ctypes as C
    import itw_amd
    from itw_amd import surfaces
    torch.cuda.set_device(0)
    for fmt, prof in (("bc7", "slow"), ("bc7", "basic"), ("bc6h", "slow"), ("bc1", None)):
        img = surfaces.hdr_smooth(4096, 4096) if fmt == "bc6h" else surfaces.ldr_smooth(4096, 4096)
        res = []
        for pinned in (False, True):
            src = torch.from_numpy(img.view(np.int16) if fmt == "bc6h" else img)
            out = torch.empty(1024 * 1024 * itw_amd.BYTES_PER_BLOCK[fmt], dtype=torch.uint8)
            if pinned:
                src = src.pin_memory(); out = out.pin_memory()
            a = src.numpy(); a = a.view(np.uint16) if fmt == "bc6h" else a
            surf = itw_amd.RgbaSurface(a.ctypes.data, 4096, 4096, a.strides[0])
            ts = []
            for _ in range(6):
                t0 = time.perf_counter(); itw_amd.abi._call(fmt, surf, out.data_ptr(), prof); ts.append(time.perf_counter() - t0)
            res.append(min(ts) * 1e3)
        print(f"chunks={os.environ.get('ITW_HOST_CHUNKS','default')} {fmt} {prof}: pageable {res[0]:.2f} ms  pinned {res[1]:.2f} ms", flush=True)
else:
    for n in ("1", "2", "4", "8"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, ITW_HOST_CHUNKS=n))
