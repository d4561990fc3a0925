"""The per-frame exchange step at world size 1 over RCCL (dev tool, GPU box): host time to submit a frame vs the period, by the
pieces of the step -- which piece costs what under GPU_MAX_HW_QUEUES=4 / 8.  python tools/dist_host_time.py"""
import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist, grok_amd as G, synth
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
S = 8192
p = G.TileParams.make(S, S, 3, 8, 5)
ctx = G.Context(0)
stream, comm = torch.cuda.Stream(), torch.cuda.Stream()
ctx.set_stream(stream.cuda_stream)
d = torch.from_numpy(synth.g2(3, S, S, 8).reshape(-1)).cuda()
used_ptr = None
cbuf = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(2)]
src1 = torch.zeros(1, dtype=torch.int64, device=dev)
import ctypes as C


def as_tensor(ptr):
    class _A:                                   # __cuda_array_interface__ view of the encoder's own word
        __cuda_array_interface__ = {"shape": (1,), "typestr": "<i8", "data": (ptr, False), "version": 2}
    return torch.as_tensor(_A(), device=dev)


def run(mode, n=60):
    ctx.set_pipelining(True)
    host = 0.0
    for phase in (0, 1):
        if phase:
            torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0.0
        for f in range(n):
            h0 = time.perf_counter()
            with torch.cuda.stream(stream):
                ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
                if mode >= 1:
                    ctx.stream_wait_results(comm.cuda_stream)
                if mode == 2:
                    with torch.cuda.stream(comm):
                        cbuf[f & 1].copy_(src1)
                if mode == 3:
                    with torch.cuda.stream(comm):
                        dist.all_gather_into_tensor(cbuf[f & 1], as_tensor(ctx.table_device_ptr(2)))
            host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.set_pipelining(False)
    return dt / n * 1e3, host / n * 1e3


names = {0: "encode only", 1: "+ comm stream waits for the results", 2: "+ a copy on the comm stream", 3: "+ all_gather_into_tensor on the comm stream"}
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for m in (0, 1, 2, 3, 0):
    per, host = run(m)
    print("%-46s %.4f ms per frame, host %.4f ms per frame" % (names[m], per, host))
dist.destroy_process_group()
