"""Row N3 kernels against the HBM roofline: pt_tonemap (16 B read + 4 B written per pixel) and pt_accumulate (32 B read +
16 B written per pixel) on 1080p and 4K frames; HIP events on the context's stream."""
import sys, os; sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
import dxrs_amd_loader, dxrs_amd
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
r = dxrs_amd.Renderer(stream=ts.cuda_stream)
p = dxrs_amd.types.tonemap_params()
p10 = dxrs_amd.types.tonemap_params(dxrs_amd.types.TONE_NONE, dxrs_amd.types.TRANSFER_ST2084)
for (w, h) in ((1920, 1080), (3840, 2160)):
    n = w * h
    hdr = torch.rand((n, 4), device="cuda"); acc = torch.zeros((n, 4), device="cuda"); out = torch.zeros(n, dtype=torch.int32, device="cuda")
    def timed(fn, reps=200):
        for _ in range(10): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(ts)
        for _ in range(reps): fn()
        b.record(ts); b.synchronize()
        return a.elapsed_time(b) / reps * 1e-3
    t1 = timed(lambda: r.tonemap(hdr.data_ptr(), n, p, out.data_ptr()))
    t2 = timed(lambda: r.tonemap(hdr.data_ptr(), n, p10, out.data_ptr()))
    t3 = timed(lambda: r.accumulate(acc.data_ptr(), hdr.data_ptr(), n, 5))
    print(f"{w}x{h}: tonemap ACES+sRGB {t1*1e6:.1f} us = {20*n/t1/1e9:.0f} GB/s ({20*n/t1/8e12:.2f} of 8 TB/s) | "
          f"HDR10 {t2*1e6:.1f} us = {20*n/t2/1e9:.0f} GB/s | accumulate {t3*1e6:.1f} us = {48*n/t3/1e9:.0f} GB/s ({48*n/t3/8e12:.2f})")
