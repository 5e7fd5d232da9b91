"""Driver of the projector-form GEMM probe (tools/probe/pform_gemm.hip): time per launch and MFMA rate."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libpform.so"))
lib.pform_probe_launch.restype = ctypes.c_int
lib.pform_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
n, KS = 10240, 34
tiles = n // 32
A = (torch.randn(tiles * KS * 64 * 8, device=dev) * 0.05).half()
B = (torch.randn(tiles * KS * 64 * 8, device=dev) * 0.05).half()
out = torch.empty(tiles * tiles * 256, device=dev)
st = torch.cuda.current_stream().cuda_stream
flops = 2.0 * n * n * KS * 16
for variant, name in [(0, "wave 128x128, wg 256x256"), (1, "wave 128x64, wg 256x128"), (2, "wave 64x64, wg 128x128")]:
    for _ in range(3):
        rc = lib.pform_probe_launch(A.data_ptr(), B.data_ptr(), tiles, tiles, KS, out.data_ptr(), variant, st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.pform_probe_launch(A.data_ptr(), B.data_ptr(), tiles, tiles, KS, out.data_ptr(), variant, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}: {ms * 1e3:.1f} us per launch, {flops / ms / 1e9:.0f} TFLOP/s  (10240 x 10240 x {KS * 16}); checksum {float(out.max()):.3f}")
