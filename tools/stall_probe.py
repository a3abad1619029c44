"""Where does host time go in bench.py's timed loop?  Wraps the timer and launch calls with host clocks and
runs bench.main(); prints every wrapped call that took more than a millisecond.  (Debugging aid.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                    # noqa: E402
from graphsage_amd import kernels               # noqa: E402


def wrap(cls, name):
    fn = getattr(cls, name)

    def timed(*a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            dt = time.perf_counter() - t0
            if dt > 1e-3:
                print("[stall] %s.%s took %.2f ms" % (cls.__name__, name, dt * 1e3), file=sys.stderr, flush=True)
    setattr(cls, name, timed)


wrap(kernels.LaunchTimer, '__init__')
wrap(kernels.LaunchTimer, 'arm')
wrap(kernels.LaunchTimer, 'elapsed_us')
wrap(kernels.Sage2Call, 'run')
import torch                                    # noqa: E402
_sync = torch.cuda.synchronize


def sync(*a, **kw):
    t0 = time.perf_counter()
    _sync(*a, **kw)
    dt = time.perf_counter() - t0
    if dt > 1e-3:
        print("[stall] synchronize took %.2f ms" % (dt * 1e3), file=sys.stderr, flush=True)


torch.cuda.synchronize = sync
sys.argv = ['bench.py', '--no-cpu-baseline', '--train-steps', '0']
bench.main()
