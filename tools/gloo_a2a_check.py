"""Does torch's gloo backend order all_to_all_single / all_gather on DEVICE tensors after work queued on the current (non-default)
stream?  Two or more ranks on one GPU (torch.distributed.run, gloo).  Each iteration queues a long kernel, then a kernel that
writes the send buffer, then the collective, and checks what arrived.  Prints the number of iterations with wrong data per
collective.  (Used to root-cause the rare numeric failure of the multi-rank GPU tests)"""
import os, sys
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = 1 << 16
bad = {"a2a_even": 0, "a2a_uneven": 0, "all_gather": 0, "all_reduce": 0}
supported = {}
big = torch.randn(4096, 4096, device=dev)
for it in range(iters):
    for kind in bad:
        send = torch.zeros(world * n, dtype=torch.float64, device=dev)
        _ = big @ big                                  # ~10 ms of work ahead of the write on the same stream
        send += float(1000 * it + rank + 1)            # the value the peers must see
        try:
            if kind == "a2a_even":
                recv = torch.empty_like(send)
                dist.all_to_all_single(recv, send)
                want = torch.cat([torch.full((n,), float(1000 * it + s + 1), dtype=torch.float64) for s in range(world)])
            elif kind == "a2a_uneven":
                cs = [n // 2 + (d * 17 + it) % 97 for d in range(world)]               # what this rank sends to d
                cr = [n // 2 + (rank * 17 + it) % 97 for s in range(world)]            # what it receives from s
                recv = torch.empty(sum(cr), dtype=torch.float64, device=dev)
                dist.all_to_all_single(recv, send[:sum(cs)], cr, cs)
                want = torch.cat([torch.full((c,), float(1000 * it + s + 1), dtype=torch.float64) for s, c in enumerate(cr)])
            elif kind == "all_gather":
                parts = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(world)]
                dist.all_gather(parts, send[:n])
                recv = torch.cat(parts)
                want = torch.cat([torch.full((n,), float(1000 * it + s + 1), dtype=torch.float64) for s in range(world)])
            else:
                recv = send[:n].clone()
                dist.all_reduce(recv)
                want = torch.full((n,), float(sum(1000 * it + s + 1 for s in range(world))), dtype=torch.float64)
            supported[kind] = True
        except (RuntimeError, NotImplementedError) as e:
            supported[kind] = "unsupported: " + str(e)[:80]
            continue
        got = recv.cpu()
        if not torch.equal(got, want):
            bad[kind] += 1
torch.cuda.synchronize()
print("rank %d of %d, %d iterations: wrong results %s; supported %s" % (rank, world, iters, bad, supported), flush=True)
dist.barrier()
dist.destroy_process_group()
