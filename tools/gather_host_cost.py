"""Host time of FramePipeline.submit alone (world 1 over RCCL, no encoder): dev tool, GPU box"""
import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import grok_amd.dist as D
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
stream, comm = torch.cuda.Stream(), torch.cuda.Stream()
used = torch.tensor([1000], dtype=torch.int64, device=dev)
offs = torch.zeros(49152, dtype=torch.int64, device=dev); lens = torch.zeros(49152, dtype=torch.int32, device=dev)
arena = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
import cProfile, pstats
for depth, lag in ((1, 1), (1, 2), (4, 2)):
    pipe = D.FramePipeline(dev, (stream, comm), depth=depth, lag=lag)
    for f in range(50):
        with torch.cuda.stream(stream):
            pipe.submit(f, used, offs, lens, arena)
    pipe.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 400
    for f in range(50, 50 + n):
        with torch.cuda.stream(stream):
            pipe.submit(f, used, offs, lens, arena)
    t1 = time.perf_counter()
    pipe.flush(); torch.cuda.synchronize()
    print("depth %d lag %d: %.1f us of host time per submit" % (depth, lag, (t1 - t0) / n * 1e6))
    if depth == 4:
        pr = cProfile.Profile(); pr.enable()
        for f in range(1000, 1200):
            with torch.cuda.stream(stream):
                pipe.submit(f, used, offs, lens, arena)
        pr.disable(); pipe.flush()
        pstats.Stats(pr).sort_stats("cumtime").print_stats(18)
dist.destroy_process_group()
