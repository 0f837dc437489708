"""The processors this process may really use: the affinity mask cut to the control group's CPU quota.

Round 6: the measurement boxes report 256 hardware threads and run jobs in a container with `cpu.max` = "1600000 100000" -- 16 CPUs.
A team sized from the 256 spends the quota early in every 100 ms period and is then stopped as a whole until the period ends; the
reference binary run with `-t 256` as the CPU baseline was measured under exactly that.  (The CLI does the same in C++: cm_cli.cpp,
cpu_budget.)"""
import os


def cpu_budget():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 8
    quota = 0.0
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota us | max> <period us>"
            q, per = f.read().split()[:2]
            if q != "max" and float(per) > 0:
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:  # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                qu = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if qu > 0 and per > 0:
                quota = qu / per
        except (OSError, ValueError):
            pass
    if 1 <= quota < n:
        n = int(quota)
    env = os.environ.get("CM_CPU_BUDGET")
    if env and env.isdigit() and int(env) > 0:
        n = int(env)
    return max(n, 1)
