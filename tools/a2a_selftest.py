"""RCCL all_to_all_single in a one-rank group against the identity, for a list of element counts (float64): found the large-block
failure that mp-gadget_amd/dist.py::TorchComm works around (A2A_MAX_BYTES).  usage: python tools/a2a_selftest.py [n ...]"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
s=torch.cuda.Stream(); torch.cuda.set_stream(s)
import sys
sizes = [int(a) for a in sys.argv[1:]] or [1000, 1 << 20, 2 * 33816576, 2 * 67633152]
for n in sizes:
    send=torch.arange(n, dtype=torch.float64, device="cuda"); recv=torch.full_like(send, -1.0)
    dist.all_to_all_single(recv, send)
    torch.cuda.synchronize()
    print("a2a", n, bool(torch.equal(recv, send)), int((recv != send).sum()))
    out=torch.full((5, n//5), -1.0, dtype=torch.float64, device="cuda"); src=torch.arange(5*(n//5), dtype=torch.float64, device="cuda").reshape(5,-1)
    dist.all_to_all_single(out, src, [5], [5])
    torch.cuda.synchronize()
    print("a2a uneven", n, bool(torch.equal(out, src)))
dist.destroy_process_group()
