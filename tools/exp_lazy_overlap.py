"""Can the lazy optimizer's update of step t (memory-bound) and the catch-up of step t + 1 (arithmetic-bound) share the
GPU?  Two independent lazy states at the configs[3] shard size; update on one, catch-up on the other: each alone,
back to back on one stream, concurrently on two streams, and as even / odd workgroups of ONE launch
(hiprec_debug_lazy_dual, libhiprec_test.so).   python tools/exp_lazy_overlap.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from beta_recsys_amd import _lib
from beta_recsys_amd.mf import _new_stats

lib = _lib.load()
tlib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libhiprec_test.so"))
tlib.hiprec_debug_lazy_dual.restype = ctypes.c_int
tlib.hiprec_debug_lazy_dual.argtypes = [ctypes.c_void_p] * 7
dev = torch.device("cuda:0")
U, I, D, B = 1_250_000, 125_000, 128, 65536
n_flat = (U + I) * (D + 1) + 1


class State:
    def __init__(self, seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        self.w = torch.randn(n_flat, device=dev, generator=g) * 0.1
        self.g = torch.zeros(n_flat, device=dev)
        self.m = torch.zeros(n_flat, device=dev)
        self.v = torch.zeros(n_flat, device=dev)
        self.su = torch.full((U,), -1, dtype=torch.int32, device=dev)
        self.si = torch.full((I,), -1, dtype=torch.int32, device=dev)
        self.sc = torch.zeros((1 << 12, 2), device=dev)
        self.c = _lib.LazyState(self.w.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), U, I, D, 1,
                                self.su.data_ptr(), self.si.data_ptr(), self.sc.data_ptr(), 1 << 12, 0, 0.01, 0.9, 0.999, 1e-8)
        self.stats = _new_stats(dev)
        _lib.check(lib.hiprec_stats_reset(_lib.ptr(self.stats), 0.9, 0.999, _lib.stream_ptr(dev)))
        self.gen = g

    def lists(self):
        u = torch.randint(0, U, (B,), device=dev, generator=self.gen)
        a = torch.randint(0, I, (B,), device=dev, generator=self.gen)
        b = torch.randint(0, I, (B,), device=dev, generator=self.gen)
        return u, a, b

    def rows(self, l):
        self._keep = l
        return _lib.LazyRows(l[0].data_ptr(), B, l[1].data_ptr(), B, l[2].data_ptr(), B, None, 0)

    def update(self, l, stream):
        r = self.rows(l)
        _lib.check(lib.hiprec_lazy_update(ctypes.byref(self.c), ctypes.byref(r), None, _lib.ptr(self.stats), stream))

    def catchup(self, l, stream):
        r = self.rows(l)
        _lib.check(lib.hiprec_lazy_catchup(ctypes.byref(self.c), ctypes.byref(r), _lib.ptr(self.stats), stream))

    def advance(self, stream):
        _lib.check(lib.hiprec_stats_advance_step(_lib.ptr(self.stats), stream))


s1, s2 = State(1), State(2)
st = _lib.stream_ptr(dev)
for s in (s1, s2):            # 40 steps with real gradients: stamps of every age, non-zero moments
    for _ in range(40):
        l = s.lists()
        s.g.normal_(generator=s.gen)
        s.advance(st)
        s.update(l, st)
    s.g.zero_()
torch.cuda.synchronize()
l1, l2 = s1.lists(), s2.lists()
save = (s2.su.clone(), s2.si.clone())
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)


def run(mode, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(iters):
        s2.su.copy_(save[0]); s2.si.copy_(save[1])
        s1.advance(st)
        torch.cuda.synchronize()
        e0.record(main)
        if mode == "update":
            s1.update(l1, st)
        elif mode == "catchup":
            s2.catchup(l2, st)
        elif mode == "serial":
            s1.update(l1, st); s2.catchup(l2, st)
        elif mode == "dual":
            r1, r2 = s1.rows(l1), s2.rows(l2)
            rc = tlib.hiprec_debug_lazy_dual(ctypes.addressof(s1.c), ctypes.addressof(r1), _lib.ptr(s1.stats),
                                             ctypes.addressof(s2.c), ctypes.addressof(r2), _lib.ptr(s2.stats), st)
            assert rc == 0
        else:
            side.wait_stream(main)
            s2.catchup(l2, side.cuda_stream)
            s1.update(l1, st)
            main.wait_stream(side)
        e1.record(main)
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3


for mode in ("update", "catchup", "serial", "concurrent", "dual", "update", "catchup", "serial", "concurrent", "dual"):
    print(f"{mode:10s} {run(mode):8.1f} us")
