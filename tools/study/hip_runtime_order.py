"""Which HIP / HSA runtime serves a process that uses both libcolate_amd.so and PyTorch-ROCm (GPU box)?  `ours_first` loads the
library through ctypes directly (not through colate_amd/_lib.py, which imports torch first for this very reason), then torch;
`torch_first` the other way round.  Observed: ours first -> torch.cuda.is_available() is False (two runtimes in the process)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l})
if order == "ours_first":
    import ctypes
    lib = ctypes.CDLL(os.path.join(sys.path[0], "colate_amd", "lib", "libcolate_amd.so"))
    print("ours:", lib.colate_device_count(), maps())
    import torch
    print("torch avail:", torch.cuda.is_available(), maps())
else:
    import torch
    print("torch avail:", torch.cuda.is_available(), maps())
    import colate_amd
    print("ours:", colate_amd.device_count(), maps())
