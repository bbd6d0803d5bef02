"""FOF timing experiments: uniform vs clumped sets."""
import importlib, sys, time, math
sys.path.insert(0, '.')
import torch
pkg = importlib.import_module("mp-gadget_amd")
dev = torch.device("cuda", 0)
n = 256; N = n ** 3; box = 1000.0 * n; LL = 0.2 * box / n
g = torch.Generator(device=dev).manual_seed(7)
f8 = torch.float64
def clumps(total, mmin, mmax):
    parts, left = [], total
    cpu = torch.Generator().manual_seed(3)
    while left > 0:
        m = min(left, int(torch.exp(torch.empty(1).uniform_(math.log(mmin), math.log(mmax), generator=cpu)).item()))
        c = torch.rand(3, dtype=f8, device=dev, generator=g) * box
        parts.append(torch.remainder(c + torch.randn(m, 3, dtype=f8, device=dev, generator=g) * (0.25 * LL * m ** (1. / 3)), box))
        left -= m
    return torch.cat(parts)
sets = {"uniform": lambda: torch.rand(N, 3, dtype=f8, device=dev, generator=g) * box,
        "clumps 20..20000": lambda: clumps(N, 20, 20000),
        "clumps 20..200": lambda: clumps(N, 20, 200),
        "4 clumps of 1M + uniform": lambda: torch.cat([clumps(4 << 20, 1 << 20, (1 << 20) + 1), torch.rand(N - (4 << 20), 3, dtype=f8, device=dev, generator=g) * box])}
eng = pkg.Engine(0); eng.use_torch_stream()
for name, mk in sets.items():
    pos = mk().contiguous(); pos.clamp_(min=1e-9)
    mass = torch.ones(N, dtype=torch.float32, device=dev)
    ids = torch.randperm(N, device=dev, generator=g).to(torch.int64)
    grnr = torch.zeros(N, dtype=torch.int64, device=dev)
    eng.dev_bind_particles(pos, mass, box)
    ng = eng.dev_fof_fof(ids, LL, 32, grnr=grnr); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ng = eng.dev_fof_fof(ids, LL, 32, grnr=grnr)
    torch.cuda.synchronize()
    print("%-18s %.1f ms  groups %d  in groups %d" % (name, (time.perf_counter() - t0) / 3 * 1e3, ng, int((grnr >= 0).sum())), flush=True)
