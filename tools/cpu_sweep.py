import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle_py
from tools import fmsig_py
FS, D, N = 2.4e6, 11, 65536
p = fmsig_py.default_params(FS, noise_sigma=0.01)
blocks = np.stack([fmsig_py.generate_f32(p, b * N, N) for b in range(16)])
params = oracle_py.FmoParams(FS, -0.15 * FS, 48000.0, 15000.0, D, 0, 0, 0, 0, 0)
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "n/a")
print("affinity", len(os.sched_getaffinity(0)), "loadavg", open("/proc/loadavg").read().strip())
for t in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    r, c, w = oracle_py.bench_threads(params, t, 1.5, blocks)
    print("threads %3d  %8.1f MS/s  per thread %6.2f  calls %d" % (t, r / 1e6, r / 1e6 / t, c))
