#!/usr/bin/env python3
"""Cycle sums of a METRO_DBG_G4_CLOCK build of conv_gemm4w (patch -p0 < tools/knockouts_r06.patch; tools/build_dbg_variants.sh conv_gemm4w.hip G4_CLOCK;
METRO_HIP_LIB=.../ab/libmetro_G4_CLOCK.so python tools/gemm4w_clock.py [batch]): per K tile, waves 0 and 3 -- k steps 0-2 (with the staging of the next
tile), the `s_waitcnt lgkmcnt(0) + s_barrier`, k step 3; and the tile's prologue + loop and epilogue."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.')
from metro_pose3d_amd import _lib
from tests import helpers as H
lib = _lib.load(); dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device=dev); g.manual_seed(0)
for c_in, c_out in ((1024, 2048), (2048, 512), (1024, 256)):
    x = torch.randn((n, 16, 16, c_in), generator=g, device=dev).half()
    w = (torch.randn((c_out, c_in), generator=g, device=dev) * (2.0 / c_in) ** 0.5).half()
    b = torch.zeros(c_out, dtype=torch.float32, device=dev)
    sc = (torch.rand(c_in, generator=g, device=dev) + 0.5).half(); sh = (torch.randn(c_in, generator=g, device=dev) * 0.3).half()
    out = torch.zeros((n, 16, 16, c_out), dtype=torch.float16, device=dev)
    d = H.conv_desc(n, 16, c_in, 16, c_out, 1, prologue=True)
    lib.metro_kernel_notes(1)
    for _ in range(3):
        assert lib.metro_conv_f16_gemm4w(C.byref(d), H.ptr(x), H.ptr(w), H.ptr(b), H.ptr(sc), H.ptr(sh), None, H.ptr(out), 0, None, C.c_void_p(0)) == 0, lib.metro_last_error()
    torch.cuda.synchronize()
    kid = lib.metro_last_kernel_id().decode()
    tn = 256 if '256x256' in kid else 128 if '256x128' in kid else 64
    rows = out.view(-1, c_out)
    recs = []
    for m0 in range(0, rows.shape[0], tn):
        for n0 in range(0, c_out, 256):
            v = rows[m0, n0:n0 + 64].contiguous().view(torch.int64).cpu().numpy()
            if v[6] == 0x600DC10C and v[14] == 0x600DC10C:
                recs.append(np.concatenate([v[:6], v[8:14]]))
    if not recs:
        print(f'batch {n} {c_in} -> {c_out}: {kid}: no records'); continue
    r = np.array(recs, dtype=np.float64)
    nk = r[:, 5].mean()
    print(f'batch {n}  {c_in} -> {c_out}  {kid}: {len(recs)} tiles, {nk:.0f} K tiles each; MFMA floor per K tile {tn // 4 * 32} cycles')
    for wv, off in ((0, 0), (3, 6)):
        s012, bar, s3, loop, epi = (r[:, off + i].mean() for i in range(5))
        print(f'   wave {wv}: per K tile: k steps 0-2 {s012 / nk:6.0f}  wait + barrier {bar / nk:5.0f}  k step 3 {s3 / nk:5.0f}  = {(s012 + bar + s3) / nk:6.0f} cycles | '
              f'prologue + loop {loop:8.0f}  epilogue {epi:7.0f}')
