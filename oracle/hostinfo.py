"""How many host cores this process may really use (affinity mask and cgroup CPU quota), so CPU-side oracle runs
do not oversubscribe a container that reports the whole machine in os.cpu_count()."""
import math
import os


def usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, math.ceil(q / p)))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)
