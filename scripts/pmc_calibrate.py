"""rocprofv3 PMC calibration: gather-copy N random 512-B rows (k=128) out of a 2 GiB table (> the 256 MiB
Infinity Cache) with the step kernels' access pattern.  Known traffic per launch: N*512 B read + N*512 B
written (+ 4N B of row ids).  Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import torch, tkr_hip
k, n_rows, N = 128, 1 << 22, 1 << 20
src = torch.randn(n_rows, k, device='cuda'); dst = torch.empty_like(src)
g = torch.Generator(device='cuda'); g.manual_seed(0)
for rep in range(5):
    rows = torch.randperm(n_rows, device='cuda', generator=g)[:N].to(torch.int32)
    rc = tkr_hip.lib().tkr_calib_rowcopy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_void_p(rows.data_ptr()), N, k,
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
torch.cuda.synchronize()
print('known bytes per launch: read', N * k * 4 + N * 4, 'write', N * k * 4)
