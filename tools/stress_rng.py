#!/usr/bin/env python3
"""Is torch.randn(generator=<a CUDA generator seeded s>) reproducible when two host threads draw at once (each from its own generators, on its own stream)?"""
import sys, threading, torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lock = threading.Lock() if len(sys.argv) > 2 else None
dev = torch.device("cuda", 0)
def draws(seeds):
    gens = [torch.Generator(device=dev).manual_seed(s) for s in seeds]
    out = []
    for shape in ((5, 4, 8, 8), (4, 16, 16), (4, 16, 16), (4, 8, 8)):
        if lock:
            with lock:
                out.append(torch.stack([torch.randn(shape, generator=g, device=dev) for g in gens]))
        else:
            out.append(torch.stack([torch.randn(shape, generator=g, device=dev) for g in gens]))
    return torch.cat([o.flatten() for o in out])
ref = [draws([21, 22]), draws([23, 24])]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bad = [0, 0]
def work(t):
    with torch.cuda.stream(streams[t]):
        for _ in range(N):
            o = draws([21, 22] if t == 0 else [23, 24])
            streams[t].synchronize()
            if not torch.equal(o, ref[t]):
                bad[t] += 1
ths = [threading.Thread(target=work, args=(t,)) for t in range(2)]
[t.start() for t in ths]; [t.join() for t in ths]
print("two threads, %d repetitions each%s: %d / %d draws differ from the single-threaded reference" % (N, " (draws under a lock)" if lock else "", bad[0], bad[1]))
