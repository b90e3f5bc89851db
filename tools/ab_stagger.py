"""Does the relative placement of q / xk / sj / y matter?  Vectors carved from one arena with a stagger between their
start offsets (modulo a large power of two), ShiftedNormL1Box n = 1e8, interleaved rounds."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
n = 100_000_000
chi = s.NormLinf(1.0)
seg = 1 << 27                        # 2^27 doubles = 1 GiB per vector slot
arena = torch.empty(4 * seg + (1 << 22), dtype=torch.float64, device=dev)
base = (-(arena.data_ptr() // 8)) % (1 << 21)   # align slot 0 to 16 MiB
staggers = [0, 32, 128, 512, 2048, 8192, 32768, 131072]   # doubles
res = {st: [] for st in staggers}
g = torch.Generator(device=dev).manual_seed(1)
src = [torch.randn(n, dtype=torch.float64, device=dev, generator=g), torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5,
       torch.randn(n, dtype=torch.float64, device=dev, generator=g)]
for rnd in range(5):
    for st in staggers:
        vs = [arena[base + k * seg + k * st: base + k * seg + k * st + n] for k in range(4)]
        for v, t in zip(vs[:3], src): v.copy_(t)
        xk, sj, q, y = vs
        psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, chi), sj)
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(20): s.prox_bang(y, psi, q, 1.0)
        L.spx_timer_stop(ctx, ctypes.byref(ms)); res[st].append(ms.value / 20)
for st in staggers:
    t = sorted(res[st][1:]); med = t[len(t) // 2]
    print("stagger %7d B  median %.4f ms  -> %.0f GB/s" % (st * 8, med, 32 * n / med / 1e6))
