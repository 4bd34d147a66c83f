import sys, os, ctypes
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime' in l})
if order == "torch_first":
    import torch
    print("torch cuda avail", torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.ones(4, device="cuda"); print(x.sum().item())
    from weiner_slamit_v2_amd import api
    print("slamit devices", api.device_count())
else:
    from weiner_slamit_v2_amd import api
    print("slamit devices", api.device_count())
    import torch
    print("torch cuda avail", torch.cuda.is_available(), torch.cuda.device_count())
print(maps())
import numpy as np
from weiner_slamit_v2_amd import synth
ext = api.ORBextractor(1000)
k, d = ext(synth.synth_frame(640, 480, 0))
print("kps", len(k))
if order == "torch_first":
    frames = torch.from_numpy(np.stack([synth.synth_frame(640,480,i) for i in range(2)])).cuda()
    ext2 = api.ORBextractor(1000, max_batch=2); ext2._bind(640,480,2); cap = ext2.max_keypoints
    dk = torch.zeros((2,cap,7), device="cuda"); dd = torch.zeros((2,cap,32), dtype=torch.uint8, device="cuda"); dn = torch.zeros(2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ext2.extract_batch_dev(frames, dk, dd, dn, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); print("dev n", dn.cpu().numpy())
