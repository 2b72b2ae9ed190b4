import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from audio_denoising_amd.pipeline import HopPipeline
dev = torch.device('cuda', 0)
dn = bench.build_denoiser(dev)
B = 256
frames = (0.1 * torch.randn(B, 1024)).to(dev); hx = dn.init_hx(B); out = torch.empty_like(frames)
pipe = HopPipeline(dn, B)
for i in range(20): pipe.submit(frames, hx, out, seed=0)
pipe.flush(); torch.cuda.synchronize()
for K in (200,):
    t0 = time.perf_counter()
    for i in range(K): pipe.submit(frames, hx, out, seed=0)
    t1 = time.perf_counter()
    pipe.flush(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"pipe: enqueue {1e6*(t1-t0)/K:.1f} us/step, total {1e6*(t2-t0)/K:.1f} us/step")
    t0 = time.perf_counter()
    for i in range(K): dn.process_frame_(frames, hx, out, seed=i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"serial: enqueue {1e6*(t1-t0)/K:.1f} us/step, total {1e6*(t2-t0)/K:.1f} us/step")
