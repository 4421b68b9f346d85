import subprocess, sys, os, time
sys.path.insert(0, '/root/repo')
mode = sys.argv[1]
if mode in ("lib", "both"):
    from __graft_entry__ import load_package
    pkg = load_package(); pkg.init(0)
if mode in ("torch", "both"):
    import torch
    torch.cuda.set_device(0); x = torch.zeros(1024, device="cuda"); torch.cuda.synchronize()
if mode == "torchlib":
    import torch, numpy as np
    torch.cuda.set_device(0); device = torch.device("cuda", 0)
    from __graft_entry__ import load_package
    pkg = load_package(); pkg.init(0)
    pts = pkg.synth_points(0, 1, 1, 1 << 20); sc = pkg.synth_scalars(0, 2, 1 << 20)
    bs = pkg.BaseSet(0, 1, pts)
    d_sc = torch.from_numpy(sc.view(np.int64)).to(device)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3): bs.msm(d_sc.data_ptr(), n=1 << 20, on_device=True, stream=stream)
    torch.cuda.synchronize()
    if len(sys.argv) > 2: bs.close(); del d_sc; torch.cuda.empty_cache()
if mode == "libmsm":
    from __graft_entry__ import load_package
    pkg = load_package(); pkg.init(0)
    pts = pkg.synth_points(0, 1, 1, 1 << 20); sc = pkg.synth_scalars(0, 2, 1 << 20)
    bs = pkg.BaseSet(0, 1, pts); bs.msm(sc); bs.close()
r = subprocess.run(["/root/repo/snark-challenge-prover-reference_amd/main_hip", "MNT4753", "compute", "/tmp/p20", "/tmp/i20", "/tmp/o20"], capture_output=True, text=True)
print(mode, [l for l in r.stdout.splitlines() if "Total time from" in l or "enqueued" in l or "remaining" in l])
