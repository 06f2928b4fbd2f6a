"""Shader clock while cmf_gemm runs back to back (rocm-smi sampled from a child process); optional CMF_GEMM_DIAG_RT."""
import os, sys, subprocess, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")
M, N, K = 131072, 512, 512
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
for _ in range(5):
    gemm(A, W)
torch.cuda.synchronize()
t0 = time.time()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 6000
for _ in range(n):
    gemm(A, W)
e1.record()
time.sleep(1.5)                                     # queue is ~3.8 s deep: sample in the middle of it
out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print("diag=%s  %.1f us/launch  %.1f TF" % (os.environ.get("CMF_GEMM_DIAG_RT", "0"), ms * 1e3, 2.0 * M * N * K / ms / 1e9))
print("\n".join(l for l in out.splitlines() if "sclk" in l or "ower" in l or "mclk" in l))
